from .caption_distill_double import Caption_distill_double, CustomCLIP, PromptLearner, TextEncoder  # noqa: F401
