"""``clip`` front end: model names, path-based loading, ``tokenize`` (reference project/my_code/clip/clip.py).

Differences from the reference, all forced by the offline / HIP-only setting:
* ``load`` takes a local checkpoint path (the reference's URL download, clip.py:39-68,108-109, is not reproduced;
  the trainer itself uses a fixed local path, trainers/Caption_distill_double.py:42);
* there is no torchvision ``_transform`` (clip.py:71-78): preprocessing is host-side data work outside the hot path;
* ``tokenize`` uses the BPE tokenizer when a merge table is available and otherwise the offline prompt cache that
  ships with the package (token ids of the COCO-80 prompt families, produced by the reference tokenizer).
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Union

import torch

from .model import build_model
from .simple_tokenizer import SimpleTokenizer, default_bpe

__all__ = ["available_models", "load", "tokenize"]

# model name -> conventional checkpoint file name (clip.py:29-36 lists the same names against download URLs)
_MODELS = {
    "RN50": "RN50.pt", "RN101": "RN101.pt", "RN50x4": "RN50x4.pt", "RN50x16": "RN50x16.pt",
    "ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt",
    "ViT-L/14@336px": "ViT-L-14-336px.pt",
}
SOT_TOKEN, EOT_TOKEN = 49406, 49407
_CACHE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prompt_cache.json")
_tokenizer: Optional[SimpleTokenizer] = None
_prompt_cache: Optional[dict] = None


def available_models() -> List[str]:
    return list(_MODELS.keys())


def get_tokenizer() -> Optional[SimpleTokenizer]:
    """The BPE tokenizer if its merge table can be found, else None."""
    global _tokenizer
    if _tokenizer is None and os.path.exists(default_bpe()):
        _tokenizer = SimpleTokenizer()
    return _tokenizer


class NativeTokenizer:
    """The C++ BPE tokenizer of libleclip_hip.so (csrc/bpe_tokenizer.hip) on the same merge table.  Host code: usable without a
    GPU.  Texts it refuses (html entities, context-dependent case mapping) raise ``NotImplementedError`` for the caller to route
    through ``SimpleTokenizer``."""

    def __init__(self, bpe_path: str = None):
        import ctypes
        from ..hip import _capi
        self._lib = _capi.load()
        self._ct = ctypes
        path = bpe_path or default_bpe()
        if not os.path.exists(path):
            raise FileNotFoundError(f"BPE merge table not found at {path}; set LECLIP_BPE_VOCAB")
        self._h = self._lib.leclip_bpe_open(path.encode())
        if not self._h:
            raise RuntimeError(self._lib.leclip_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.leclip_bpe_close(self._h)
            self._h = None

    def encode(self, text: str) -> List[int]:
        buf = (self._ct.c_int64 * 512)()
        n = self._lib.leclip_bpe_encode(self._h, text.encode("utf-8"), buf, 512)
        if n == -2:
            raise NotImplementedError(self._lib.leclip_last_error().decode())
        if n < 0:
            raise ValueError(self._lib.leclip_last_error().decode())
        if n > 512:
            buf = (self._ct.c_int64 * n)()
            n = self._lib.leclip_bpe_encode(self._h, text.encode("utf-8"), buf, n)
        return list(buf[:n])

    def tokenize(self, texts: List[str], context_length: int = 77, truncate: bool = False) -> torch.LongTensor:
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        arr = (self._ct.c_char_p * len(texts))(*[t.encode("utf-8") for t in texts])
        rc = self._lib.leclip_bpe_tokenize(self._h, arr, len(texts), context_length, int(truncate), out.data_ptr())
        if rc == -2:
            raise NotImplementedError(self._lib.leclip_last_error().decode())
        if rc != 0:
            msg = self._lib.leclip_last_error().decode()
            raise RuntimeError(msg) if "too long" in msg else ValueError(msg)
        return out


_native: Optional["NativeTokenizer"] = None
_native_failed = False


def get_native_tokenizer() -> Optional["NativeTokenizer"]:
    """The native tokenizer when both the library and the merge table are there; None otherwise (never an error)."""
    global _native, _native_failed
    if _native is None and not _native_failed and os.path.exists(default_bpe()):
        try:
            _native = NativeTokenizer()
        except Exception:
            _native_failed = True
    return _native


def _cached_ids(text: str) -> List[int]:
    global _prompt_cache
    if _prompt_cache is None:
        with open(_CACHE_PATH) as f:
            _prompt_cache = json.load(f)
    key = " ".join(text.strip().lower().split())
    if key not in _prompt_cache:
        raise FileNotFoundError(
            f"no BPE merge table ({default_bpe()}; set LECLIP_BPE_VOCAB) and {text!r} is not in the offline prompt cache")
    return list(_prompt_cache[key])


def encode_text(text: str) -> List[int]:
    """BPE ids of ``text`` without SOT/EOT (``_tokenizer.encode`` in the reference)."""
    nat = get_native_tokenizer()
    if nat is not None and "&" not in text and _fix_is_identity(text):
        try:
            return nat.encode(text)
        except NotImplementedError:
            pass
    tok = get_tokenizer()
    return tok.encode(text) if tok is not None else _cached_ids(text)


def _fix_is_identity(text: str) -> bool:
    """ftfy (when installed) may rewrite non-ASCII text; the native tokenizer does not run it."""
    try:
        import ftfy  # noqa: F401
    except Exception:
        return True
    return text.isascii()


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.LongTensor:
    """[n, context_length] int64: SOT + BPE ids + EOT, zero padded; over-long input raises RuntimeError unless
    ``truncate`` (then the last kept id becomes EOT) - clip.py:185-221."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, text in enumerate(texts):
        ids = [SOT_TOKEN] + encode_text(text) + [EOT_TOKEN]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = EOT_TOKEN
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
    return out


def load(name: str, device: Union[str, torch.device] = "cuda", jit: bool = False, precision: str = "fp16"):
    """Load a CLIP checkpoint from a local path (a state-dict file or a TorchScript archive) and build the
    HIP-backed model.  ``precision``: "fp16" (reference GPU behaviour), "bf16", or "fp32" (reference CPU behaviour,
    clip.py:128-129)."""
    if jit:
        raise NotImplementedError("TorchScript execution is not supported; the model runs on hand-written HIP kernels")
    if not os.path.isfile(name):
        if name in _MODELS:
            raise RuntimeError(f"Model {name} must be given as a local checkpoint path (expected file name "
                               f"{_MODELS[name]}); this build never downloads")
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    try:
        state_dict = torch.jit.load(name, map_location="cpu").eval().state_dict()
    except RuntimeError:
        state_dict = torch.load(name, map_location="cpu")
    if isinstance(state_dict, dict) and "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]
    model = build_model(state_dict)
    from .model import convert_weights
    if precision == "fp32":
        model.float()
    elif precision == "bf16":
        model.float()
        convert_weights(model, torch.bfloat16)
    elif precision != "fp16":
        raise ValueError(f"precision must be fp16, bf16 or fp32, got {precision}")
    return model.to(device)
