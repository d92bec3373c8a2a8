from .caption_distill_double import Caption_distill_double, CustomCLIP, DenseCLIP, PromptLearner, TextEncoder  # noqa: F401
