"""CLIP module surface backed by the HIP engines (ViT image tower + text tower).

Mirrors the object protocol the reference's trainers consume: ``build_model(state_dict)``
-> ``CLIP`` with ``.visual`` (``input_resolution``, ``conv1``, ``proj``), ``.transformer``,
``.token_embedding``, ``.positional_embedding``, ``.ln_final``, ``.text_projection``, ``.logit_scale``,
``.dtype``, ``.context_length`` and ``encode_image / encode_text / forward``
(reference project/my_code/clip/model.py:193-472).  Parameter names equal the OpenAI state-dict keys,
so released checkpoints and the reference's ``load_state_dict`` round-trip unchanged.

The modules only *hold* parameters; every forward runs hand-written gfx950 kernels through
``leclip_amd.hip`` on the caller's HIP stream.  There is no CPU forward: calling a tower with CPU
tensors raises.  ``model.dtype`` (= dtype of ``visual.conv1.weight``, model.py:372-374) selects the
compute precision exactly as in the reference: ``build_model`` converts GEMM weights to fp16
(model.py:411-432, 470), ``.float()`` selects the fp32 parity path, ``.bfloat16()`` the bf16 MFMA path.
Only the ViT image tower is built (north-star scope); a ResNet state-dict raises NotImplementedError.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import numpy as np
import torch
from torch import nn


class LayerNorm(nn.Module):
    """fp32-statistics LayerNorm (model.py:193-199); parameters stay fp32 in every precision mode."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.normalized_shape = (dim,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ..hip import ops
        return ops.layernorm(x.contiguous(), self.weight.detach().float(), self.bias.detach().float(), self.eps)


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) (model.py:202-204).  On the HIP path it is the epilogue of the c_fc GEMM;
    the module is a structural marker so ``mlp.gelu`` exists as in the reference."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("QuickGELU is fused into the c_fc GEMM epilogue on the HIP path; run the enclosing block")


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's names/shapes/default init (weight [out,in], bias [out])."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        nn.init.uniform_(self.bias, -in_features ** -0.5, in_features ** -0.5)


class MultiheadAttention(nn.Module):
    """Parameter holder with nn.MultiheadAttention's packed names (in_proj q|k|v, out_proj)."""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class ResidualAttentionBlock(nn.Module):
    """x += attn(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))  (model.py:207-228)."""

    def __init__(self, d_model: int, n_head: int, attn_mask: Optional[torch.Tensor] = None):
        super().__init__()
        self.attn = MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", _Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", _Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask


class Transformer(nn.Module):
    """Stack of residual attention blocks (model.py:231-239).  ``forward`` takes and returns the reference's
    LND layout [T, N, d]; the mask is either None or the causal mask of ``CLIP.build_attention_mask``."""

    def __init__(self, width: int, layers: int, heads: int, attn_mask: Optional[torch.Tensor] = None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.causal = attn_mask is not None
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])
        self._packed = None

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ..hip import engine
        t, n, d = x.shape
        dtype = self.resblocks[0].attn.in_proj_weight.dtype
        key = (dtype, x.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, engine.pack_blocks(self.resblocks, dtype, x.device))
        xb = x.permute(1, 0, 2).to(dtype).contiguous().view(n * t, d)
        ws = engine._Workspace(n * t, d, dtype, x.device)
        engine.run_blocks(xb, self._packed[1], ws, n, t, self.heads, self.causal)
        return xb.view(n, t, d).permute(1, 0, 2)


class VisionTransformer(nn.Module):
    """ViT image tower (model.py:242-276): patch conv (no bias) -> [cls | patches] + pos -> ln_pre -> blocks ->
    ln_post(cls) @ proj.  Returns fp32 features [B, output_dim]."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)  # parameter holder
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self, device):
        from ..hip.engine import VisionEngine
        dtype = self.conv1.weight.dtype
        if self._engine is None or self._engine.dtype != dtype or self._engine.device != device:
            self._engine = VisionEngine(self, dtype, device)
        return self._engine

    def forward(self, x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer.forward needs a HIP device tensor (no CPU fallback on the product path)")
        return self.engine(x.device).forward(x, taps)

    def forward_beside(self, x: torch.Tensor, fn):
        """(forward(x), fn()): fn's kernels are enqueued on the caller's stream while the tower's stream parts run on theirs, so work that
        does not depend on the image features (the prompt-tuning step's text tower) overlaps the tower instead of following it."""
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer.forward_beside needs a HIP device tensor (no CPU fallback on the product path)")
        eng = self.engine(x.device)
        feats = eng.forward(x, beside=fn)
        res, eng.beside_result = eng.beside_result, None
        return feats, res

    def dense_features(self, x: torch.Tensor) -> torch.Tensor:
        """[B, T, E] fp32 features of every token (class token first): ln_post + proj applied to all rows of the last block."""
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer.dense_features needs a HIP device tensor (no CPU fallback on the product path)")
        return self.engine(x.device).dense_features(x)

    def score(self, x: torch.Tensor, text_features: torch.Tensor, scale: float) -> torch.Tensor:
        """scale * normalize(self(x)) @ normalize(text_features).T (model.py:399-404) with the contraction folded into the
        tower's tail kernel: logits [B, C] fp32."""
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer.score needs a HIP device tensor (no CPU fallback on the product path)")
        return self.engine(x.device).score(x, text_features, scale)[1]


class CLIP(nn.Module):
    def __init__(self, embed_dim: int, image_resolution: int, vision_layers, vision_width: int, vision_patch_size: int,
                 context_length: int, vocab_size: int, transformer_width: int, transformer_heads: int,
                 transformer_layers: int):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet image towers are outside the scope of this build (ViT only)")
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads,
                                       attn_mask=self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self._text_engine = None
        self.initialize_parameters()

    def initialize_parameters(self):
        """Same initial distributions as model.py:335-362 (text tower stds, 0.02 / 0.01 embeddings)."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        width, layers = self.transformer.width, self.transformer.layers
        proj_std, attn_std, fc_std = (width ** -0.5) * ((2 * layers) ** -0.5), width ** -0.5, (2 * width) ** -0.5
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=width ** -0.5)

    def build_attention_mask(self) -> torch.Tensor:
        """Additive causal mask, -inf strictly above the diagonal (model.py:364-370).  The HIP attention kernel
        applies the same mask analytically (key > query => excluded); the tensor is kept for API parity."""
        mask = torch.full((self.context_length, self.context_length), float("-inf"))
        return torch.triu(mask, diagonal=1)

    def _apply(self, fn, *a, **k):
        self._text_engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._text_engine = None
        self.visual._engine = None
        return super().load_state_dict(*a, **k)

    @property
    def dtype(self) -> torch.dtype:
        return self.visual.conv1.weight.dtype

    def text_engine(self, device):
        from ..hip.engine import TextEngine
        if self._text_engine is None or self._text_engine.dtype != self.dtype or self._text_engine.device != device:
            self._text_engine = TextEngine(self, self.dtype, device)
        return self._text_engine

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        return self.visual(image)

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        if not text.is_cuda:
            raise RuntimeError("CLIP.encode_text needs a HIP device tensor (no CPU fallback on the product path)")
        return self.text_engine(text.device).encode_tokens(text)

    def forward(self, image: torch.Tensor, text: torch.Tensor):
        from ..hip import ops
        fi, ft = self.encode_image(image), self.encode_text(text)
        scale = float(self.logit_scale.detach().float().exp())
        return ops.l2norm_logits(fi, ft, scale), ops.l2norm_logits(ft, fi, scale)


def convert_weights(model: nn.Module, dtype: torch.dtype = torch.float16):
    """Cast GEMM operands (conv / linear / packed attention weights and biases, ``proj``, ``text_projection``) to
    ``dtype`` and leave LayerNorm, embeddings and positional embeddings in fp32 - model.py:411-432."""
    def _cast(m):
        if isinstance(m, (nn.Conv2d, _Linear)):
            m.weight.data = m.weight.data.to(dtype)
            if getattr(m, "bias", None) is not None:
                m.bias.data = m.bias.data.to(dtype)
        if isinstance(m, MultiheadAttention):
            m.in_proj_weight.data = m.in_proj_weight.data.to(dtype)
            m.in_proj_bias.data = m.in_proj_bias.data.to(dtype)
        for name in ("text_projection", "proj"):
            p = getattr(m, name, None)
            if isinstance(p, torch.Tensor):
                p.data = p.data.to(dtype)
    model.apply(_cast)
    for m in model.modules():
        if hasattr(m, "_engine"):
            m._engine = None
        if hasattr(m, "_text_engine"):
            m._text_engine = None
        if hasattr(m, "_packed"):
            m._packed = None


def arch_from_state_dict(sd: dict) -> dict:
    """Shape-driven architecture inference, model.py:436-458 (ViT branch)."""
    if "visual.proj" not in sd:
        raise NotImplementedError("state-dict has no 'visual.proj': ResNet CLIP towers are outside this build's scope")
    width = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    t_width = sd["ln_final.weight"].shape[0]
    return dict(
        embed_dim=sd["text_projection"].shape[1], image_resolution=patch * grid,
        vision_layers=len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")]),
        vision_width=width, vision_patch_size=patch, context_length=sd["positional_embedding"].shape[0],
        vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=t_width, transformer_heads=t_width // 64,
        transformer_layers=len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")}))


def build_model(state_dict: dict) -> CLIP:
    """state-dict -> CLIP in eval mode with fp16 GEMM weights, as model.py:435-472."""
    state_dict = dict(state_dict)
    model = CLIP(**arch_from_state_dict(state_dict))
    for key in ("input_resolution", "context_length", "vocab_size"):
        state_dict.pop(key, None)
    convert_weights(model)
    model.load_state_dict(state_dict)
    return model.eval()
