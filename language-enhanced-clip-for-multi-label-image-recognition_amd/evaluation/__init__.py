from .evaluator import MLClassification, average_precision, mAP  # noqa: F401
