from .clip import available_models, load, tokenize, encode_text, _MODELS  # noqa: F401
from .model import CLIP, VisionTransformer, Transformer, LayerNorm, build_model, convert_weights  # noqa: F401
