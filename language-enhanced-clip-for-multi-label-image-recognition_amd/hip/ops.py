"""Tensor-level wrappers over the C ABI: validate torch tensors, pass raw pointers + the current HIP stream.

PyTorch is plumbing here (device memory, streams); every arithmetic step is a kernel in
``csrc/*.hip``.  Each wrapper mirrors one ``leclip_*_fwd`` entry point of ``include/leclip_hip.h``.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _capi
from ._capi import ACT_NONE, ACT_QUICKGELU, MASK_CAUSAL, MASK_NONE

_DT = {torch.float32: _capi.F32, torch.float16: _capi.F16, torch.bfloat16: _capi.BF16}


# Optional per-launch timing hook (bench.py): when a list is installed, gemm()/attention()/layernorm() bracket
# their launch with HIP events on the launch stream and append (kernel family, algorithmic flops, bytes, e0, e1, shape label).
_PROFILE = None


def set_profile(sink):
    global _PROFILE
    _PROFILE = sink


class _Timed:
    __slots__ = ("name", "flops", "nbytes", "e0", "shape")

    def __init__(self, name, flops, nbytes, shape=""):
        self.name, self.flops, self.nbytes, self.shape = name, flops, nbytes, shape

    def __enter__(self):
        if _PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _PROFILE.append((self.name, self.flops, self.nbytes, self.e0, e1, self.shape))
        return False


def set_gemm_family(family: int) -> int:
    """GEMM kernel-family override for the calling thread's next launches (leclip_set_gemm_family): 128 / 256 / 384, anything else = the
    library's rate heuristic.  Results never depend on it (the families are bit-identical); tests and A/B timings use it.  Returns the previous override."""
    return _capi.load().leclip_set_gemm_family(int(family))


def set_walk_order(order: int) -> int:
    """Walk-order hint for the calling thread's next launches (leclip_set_walk_order): 0 ascending rows, 1 descending, -1 library default.
    Which rows a workgroup takes first, never what it computes.  Returns the previous hint."""
    return _capi.load().leclip_set_walk_order(int(order))


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}; the HIP path computes in float32, float16 or bfloat16") from None


def _dev(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the CLIP scoring path runs only on a HIP device "
                           f"(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rows2d(t: torch.Tensor, name: str):
    """View a [..., d] tensor as rows; returns (rows, d, leading dimension)."""
    _dev(t, name)
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")
    if t.dim() == 1:
        return 1, t.shape[0], t.shape[0]
    if t.dim() > 2 and not t.is_contiguous():
        raise ValueError(f"{name}: tensors with more than 2 dims must be contiguous")
    rows = t.numel() // t.shape[-1]
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    return rows, t.shape[-1], ld


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows, dim, ldx = _rows2d(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    _, _, ldy = _rows2d(out, "out")
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == dim == beta.numel()
    with _Timed("layernorm", 0, rows * dim * (x.element_size() + out.element_size())):
        _capi.check(_capi.load().leclip_layernorm_fwd(_ptr(x), _ptr(_dev(gamma, "gamma")), _ptr(_dev(beta, "beta")), _ptr(out),
                                                      rows, dim, ldx, ldy, eps, dtype_code(x.dtype), dtype_code(out.dtype),
                                                      _stream()), "layernorm")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(a @ w.T + bias) + residual ; a [..., K], w [N, K] (nn.Linear layout)."""
    m, k, lda = _rows2d(a, "a")
    n, kw, ldw = _rows2d(w, "w")
    if k != kw or a.dtype != w.dtype:
        raise ValueError(f"gemm: a [.., {k}] {a.dtype} vs w [{n}, {kw}] {w.dtype}")
    if out is None:
        out = torch.empty(a.shape[:-1] + (n,), dtype=out_dtype or a.dtype, device=a.device)
    mo, no, ldy = _rows2d(out, "out")
    assert (mo, no) == (m, n)
    ldr = 0
    rdt = _capi.F32
    if residual is not None:
        mr, nr, ldr = _rows2d(residual, "residual")
        assert (mr, nr) == (m, n)
        rdt = dtype_code(residual.dtype)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == n
        _dev(bias, "bias")
    nbytes = (m * k + n * k) * a.element_size() + m * n * out.element_size() + (m * n * residual.element_size() if residual is not None else 0)
    with _Timed("gemm", 2 * m * n * k, nbytes, f"M{m} N{n} K{k}"):
        _capi.check(_capi.load().leclip_gemm_bias_act_res_fwd(_ptr(a), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out), m, n, k,
                                                              lda, ldw, ldr, ldy, act, dtype_code(a.dtype), rdt,
                                                              dtype_code(out.dtype), _stream()), "gemm")
    return out


def patch_embed_workspace(batch: int, resolution: int, patch: int, w_dtype: torch.dtype, device) -> torch.Tensor:
    nbytes = _capi.load().leclip_patch_embed_workspace_bytes(batch, resolution, patch, dtype_code(w_dtype))
    if nbytes < 0:
        raise ValueError("patch_embed: bad geometry")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def patch_embed(image: torch.Tensor, wp: torch.Tensor, class_emb: torch.Tensor, pos: torch.Tensor, patch: int,
                x_dtype: torch.dtype, workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    _dev(image, "image")
    b, c, r, r2 = image.shape
    if c != 3 or r != r2 or not image.is_contiguous():
        raise ValueError("patch_embed: image must be contiguous [B,3,R,R]")
    width = wp.shape[0]
    t = (r // patch) ** 2 + 1
    assert pos.shape == (t, width) and pos.dtype == torch.float32 and class_emb.dtype == torch.float32
    if workspace is None:
        workspace = patch_embed_workspace(b, r, patch, wp.dtype, image.device)
    if out is None:
        out = torch.empty((b, t, width), dtype=x_dtype, device=image.device)
    npatch = b * (t - 1)
    with _Timed("patch_embed", 2 * npatch * width * 3 * patch * patch,
                image.numel() * image.element_size() + wp.numel() * wp.element_size() + b * t * width * out.element_size()):   # ALGORITHMIC bytes: image in, weights, tokens out
        _capi.check(_capi.load().leclip_patch_embed_fwd(_ptr(image), _ptr(_dev(wp, "wp")), _ptr(class_emb), _ptr(pos), _ptr(out), b, r,
                                                        patch, width, dtype_code(image.dtype), dtype_code(wp.dtype),
                                                        dtype_code(out.dtype), _ptr(workspace), _stream()), "patch_embed")
    return out


def patch_embed_ln(image: torch.Tensor, wp: torch.Tensor, class_emb: torch.Tensor, pos: torch.Tensor, gamma: torch.Tensor,
                   beta: torch.Tensor, patch: int, x_dtype: torch.dtype, workspace: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, eps: float = 1e-5, stats_out: Optional[torch.Tensor] = None):
    """ln_pre(patch embedding) (leclip_patch_embed_ln_fwd): conv GEMM (plain fast epilogue), then the fused class-token / positional add /
    LayerNorm pass.  Images in the compute dtype with 16 x 16 patches and a batch that fills the 256 x 256 GEMM kernel are gathered by that
    kernel's LDS-DMA straight from NCHW (two launches, no patch matrix); anything else goes through a patch-extraction kernel first."""
    _dev(image, "image")
    b, c, r, r2 = image.shape
    if c != 3 or r != r2 or not image.is_contiguous():
        raise ValueError("patch_embed_ln: image must be contiguous [B,3,R,R]")
    width = wp.shape[0]
    t = (r // patch) ** 2 + 1
    assert pos.shape == (t, width) and pos.dtype == torch.float32 and class_emb.dtype == torch.float32
    if workspace is None:
        nbytes = _capi.load().leclip_patch_embed_ln_workspace_bytes(b, r, patch, width, dtype_code(wp.dtype))
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=image.device)
    if out is None:
        out = torch.empty((b, t, width), dtype=x_dtype, device=image.device)
    npatch = b * (t - 1)
    with _Timed("patch_embed", 2 * npatch * width * 3 * patch * patch,
                image.numel() * image.element_size() + wp.numel() * wp.element_size() + b * t * width * out.element_size()):   # ALGORITHMIC bytes only (no intermediates)
        _capi.check(_capi.load().leclip_patch_embed_ln_fwd(_ptr(image), _ptr(_dev(wp, "wp")), _ptr(class_emb), _ptr(pos), _ptr(_dev(gamma, "gamma")),
                                                           _ptr(_dev(beta, "beta")), _ptr(out), _ptr(stats_out), b, r, patch, width, dtype_code(image.dtype),
                                                           dtype_code(wp.dtype), dtype_code(out.dtype), eps, _ptr(workspace), _stream()),
                    "patch_embed_ln")
    return out


def attention(qkv: torch.Tensor, batch: int, tokens: int, heads: int, causal: bool = False,
              out: Optional[torch.Tensor] = None, q_rows: int = 0) -> torch.Tensor:
    """q_rows > 0: only the first q_rows query rows of every (batch, head) are computed (leclip_attention_prefix_fwd); the rest of
    ``out`` is left as it is."""
    rows, width3, ld = _rows2d(qkv, "qkv")
    d = heads * 64
    if rows != batch * tokens or width3 != 3 * d:
        raise ValueError(f"attention: qkv {tuple(qkv.shape)} vs B={batch} T={tokens} heads={heads}")
    if out is None:
        out = torch.empty((rows, d), dtype=qkv.dtype, device=qkv.device)
    _, _, ldo = _rows2d(out, "out")
    nq = q_rows if q_rows > 0 else tokens
    with _Timed("attention", 4 * batch * heads * nq * tokens * 64, (rows * 2 * d + batch * nq * 2 * d) * qkv.element_size()):
        _capi.check(_capi.load().leclip_attention_prefix_fwd(_ptr(qkv), _ptr(out), batch, tokens, heads, 64, ld, ldo,
                                                             MASK_CAUSAL if causal else MASK_NONE, 0.125, int(q_rows),
                                                             dtype_code(qkv.dtype), _stream()), "attention")
    return out


def gather_ln_proj(x: torch.Tensor, row_index: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                   proj: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    rows, dim, ldx = _rows2d(x, "x")
    assert proj.dim() == 2 and proj.shape[0] == dim and proj.is_contiguous()
    n = rows if row_index is None else row_index.numel()
    if row_index is not None:
        assert row_index.dtype == torch.int64 and row_index.is_contiguous()
        _dev(row_index, "row_index")
    out = torch.empty((n, proj.shape[1]), dtype=torch.float32, device=x.device)
    _capi.check(_capi.load().leclip_gather_ln_proj_fwd(_ptr(x), _ptr(row_index), _ptr(_dev(gamma, "gamma")), _ptr(_dev(beta, "beta")),
                                                       _ptr(_dev(proj, "proj")), _ptr(out), n, dim, proj.shape[1], ldx, eps,
                                                       dtype_code(x.dtype), dtype_code(proj.dtype), _stream()),
                "gather_ln_proj")
    return out


def l2norm_logits(img: torch.Tensor, txt: torch.Tensor, scale: float) -> torch.Tensor:
    _dev(img, "image_features"), _dev(txt, "text_features")
    if img.dtype != torch.float32 or txt.dtype != torch.float32:
        raise TypeError("l2norm_logits: features must be float32")
    img, txt = img.contiguous(), txt.contiguous()
    b, d = img.shape
    c, d2 = txt.shape
    assert d == d2
    out = torch.empty((b, c), dtype=torch.float32, device=img.device)
    _capi.check(_capi.load().leclip_l2norm_logits_fwd(_ptr(img), _ptr(txt), _ptr(out), b, c, d, float(scale), _stream()),
                "l2norm_logits")
    return out


def embed_tokens(tokens: torch.Tensor, table: torch.Tensor, pos: torch.Tensor, x_dtype: torch.dtype) -> torch.Tensor:
    _dev(tokens, "tokens")
    assert tokens.dtype == torch.int64 and tokens.dim() == 2 and tokens.is_contiguous()
    assert table.dtype == torch.float32 and pos.dtype == torch.float32 and table.is_contiguous() and pos.is_contiguous()
    n, t = tokens.shape
    dim = table.shape[1]
    assert pos.shape[0] >= t and pos.shape[1] == dim
    out = torch.empty((n, t, dim), dtype=x_dtype, device=tokens.device)
    _capi.check(_capi.load().leclip_embed_tokens_fwd(_ptr(tokens), _ptr(_dev(table, "table")), _ptr(_dev(pos, "pos")), _ptr(out), n, t,
                                                     dim, table.shape[0], dtype_code(x_dtype), _stream()), "embed_tokens")
    return out


def prompt_assemble(prefix: torch.Tensor, ctx: torch.Tensor, suffix: torch.Tensor, pos: Optional[torch.Tensor],
                    x_dtype: torch.dtype) -> torch.Tensor:
    for name, t in (("prefix", prefix), ("ctx", ctx), ("suffix", suffix)):
        _dev(t, name)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError(f"prompt_assemble: {name} must be contiguous float32")
    n_cls, one, dim = prefix.shape
    per_class = ctx.dim() == 3
    n_ctx = ctx.shape[-2]
    t = 1 + n_ctx + suffix.shape[1]
    assert one == 1 and suffix.shape[0] == n_cls and ctx.shape[-1] == dim
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.shape[0] >= t and pos.is_contiguous()
    out = torch.empty((n_cls, t, dim), dtype=x_dtype, device=prefix.device)
    _capi.check(_capi.load().leclip_prompt_assemble_fwd(_ptr(prefix), _ptr(ctx), _ptr(suffix), _ptr(pos), _ptr(out), n_cls, n_ctx, t,
                                                        dim, int(per_class), dtype_code(x_dtype), _stream()), "prompt_assemble")
    return out


def add_pos(x: torch.Tensor, pos: torch.Tensor, x_dtype: torch.dtype) -> torch.Tensor:
    _dev(x, "prompts")
    x = x.contiguous()
    assert x.dtype == torch.float32 and x.dim() == 3 and pos.dtype == torch.float32 and pos.is_contiguous()
    n, t, dim = x.shape
    out = torch.empty((n, t, dim), dtype=x_dtype, device=x.device)
    _capi.check(_capi.load().leclip_add_pos_fwd(_ptr(x), _ptr(_dev(pos, "pos")), _ptr(out), n, t, dim, dtype_code(x_dtype), _stream()),
                "add_pos")
    return out


def eot_index(tokens: torch.Tensor):
    """(argmax over the token axis, flat row n*T + argmax) - both int64 device tensors."""
    _dev(tokens, "tokens")
    assert tokens.dtype == torch.int64 and tokens.dim() == 2 and tokens.is_contiguous()
    n, t = tokens.shape
    eot = torch.empty(n, dtype=torch.int64, device=tokens.device)
    flat = torch.empty(n, dtype=torch.int64, device=tokens.device)
    _capi.check(_capi.load().leclip_eot_index_fwd(_ptr(tokens), _ptr(eot), _ptr(flat), n, t, _stream()), "eot_index")
    return eot, flat


def gemm_ln(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, ln_stats: Optional[torch.Tensor] = None,
            ln_colsum: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, act: int = ACT_NONE,
            stats_out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
            out: Optional[torch.Tensor] = None, ln_partials: Optional[torch.Tensor] = None, ln_stats_ws: Optional[torch.Tensor] = None,
            eps: float = 1e-5) -> torch.Tensor:
    """GEMM with LayerNorm folded around it (leclip_gemm_ln_partials_fwd).  The A-side LayerNorm statistics come either as
    ``ln_stats`` [M,2] (mean, rstd) or as the producer's block partials ``ln_partials`` [K/64, M, 2] (slot-major); the C entry point then ALWAYS
    launches the separate merge kernel (ln_stats_finalize_kernel, same stream, in front of the GEMM) that turns the partials into
    (mean, rstd) in ``ln_stats_ws`` [M,2], which the caller provides - an in-kernel merge was built twice and rejected for register
    spills.  ``stats_out`` [N/64, M, 2] receives the output rows' block partials (sum, M2 about the block
    mean) for the next LayerNorm."""
    m, k, lda = _rows2d(a, "a")
    n, kw, ldw = _rows2d(w, "w")
    if k != kw or a.dtype != w.dtype:
        raise ValueError(f"gemm_ln: a [.., {k}] {a.dtype} vs w [{n}, {kw}] {w.dtype}")
    if out is None:
        out = torch.empty(a.shape[:-1] + (n,), dtype=out_dtype or a.dtype, device=a.device)
    _, _, ldy = _rows2d(out, "out")
    ldr, rdt = 0, _capi.F32
    if residual is not None:
        _, _, ldr = _rows2d(residual, "residual")
        rdt = dtype_code(residual.dtype)
    slots = 0
    if ln_partials is not None:
        slots = ln_partials.shape[0]
        if ln_stats_ws is None:
            ln_stats_ws = torch.empty((m, 2), dtype=torch.float32, device=a.device)
    for name, t, shape in (("ln_stats", ln_stats, (m, 2)), ("ln_colsum", ln_colsum, (n,)), ("stats_out", stats_out, (n // 64, m, 2)),
                           ("bias", bias, (n,)), ("ln_partials", ln_partials, (k // 64, m, 2)), ("ln_stats_ws", ln_stats_ws, (m, 2))):
        if t is not None:
            _dev(t, name)
            assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape, (name, tuple(t.shape), shape)
    nbytes = (m * k + n * k) * a.element_size() + m * n * out.element_size() + (m * n * residual.element_size() if residual is not None else 0)
    with _Timed("gemm", 2 * m * n * k, nbytes, f"M{m} N{n} K{k}"):
        _capi.check(_capi.load().leclip_gemm_ln_partials_fwd(_ptr(a), _ptr(w), _ptr(bias), _ptr(ln_stats), _ptr(ln_partials), slots,
                                                             _ptr(ln_stats_ws), eps, _ptr(ln_colsum), _ptr(residual), _ptr(out), _ptr(stats_out),
                                                             m, n, k, lda, ldw, ldr, ldy, act, dtype_code(a.dtype), rdt,
                                                             dtype_code(out.dtype), _stream()), "gemm_ln")
    return out


def gemm_res_stats(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: torch.Tensor, partials: torch.Tensor,
                   stats_out: torch.Tensor, tickets: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """out = a @ w.T + bias + residual and stats_out [M, 2] = (mean, rstd) of out's rows (leclip_gemm_res_stats_fwd): the residual-stream update
    of a block with the next LayerNorm's statistics finished by the producer - on the 384 x 256 kernel inside the launch (``tickets``: int32 zeros,
    one per 384-row block, left zero), otherwise by the merge kernel behind it.  ``partials`` [N/64, M, 2] is scratch (it keeps the block partials)."""
    m, k, lda = _rows2d(a, "a")
    n, kw, ldw = _rows2d(w, "w")
    if k != kw or a.dtype != w.dtype:
        raise ValueError(f"gemm_res_stats: a [.., {k}] {a.dtype} vs w [{n}, {kw}] {w.dtype}")
    if out is None:
        out = torch.empty(a.shape[:-1] + (n,), dtype=a.dtype, device=a.device)
    _, _, ldy = _rows2d(out, "out")
    _, _, ldr = _rows2d(residual, "residual")
    for name, t, shape in (("partials", partials, (n // 64, m, 2)), ("stats_out", stats_out, (m, 2)), ("bias", bias, (n,))):
        if t is not None:
            _dev(t, name)
            assert t.dtype == torch.float32 and tuple(t.shape) == shape and t.is_contiguous(), (name, tuple(t.shape), shape)
    if tickets is not None:
        _dev(tickets, "tickets")
        assert tickets.dtype == torch.int32 and tickets.is_contiguous() and tickets.numel() >= (m + 383) // 384
    nbytes = (m * k + n * k) * a.element_size() + 2 * m * n * out.element_size()
    with _Timed("gemm", 2 * m * n * k, nbytes, f"M{m} N{n} K{k}"):
        _capi.check(_capi.load().leclip_gemm_res_stats_fwd(_ptr(a), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out), _ptr(partials), _ptr(stats_out),
                                                           _ptr(tickets), eps, m, n, k, lda, ldw, ldr, ldy, dtype_code(a.dtype),
                                                           dtype_code(residual.dtype), dtype_code(out.dtype), _stream()), "gemm_res_stats")
    return out


def ln_stats_finalize(partials: torch.Tensor, dim: int, eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(partials, "partials")
    assert partials.dtype == torch.float32 and partials.dim() == 3 and partials.shape[2] == 2 and partials.is_contiguous()
    slots, rows, _ = partials.shape      # slot-major [slots][rows][2]
    if out is None:
        out = torch.empty((rows, 2), dtype=torch.float32, device=partials.device)
    with _Timed("ln_stats", 0, partials.numel() * 4 + rows * 8):
        _capi.check(_capi.load().leclip_ln_stats_finalize_fwd(_ptr(partials), _ptr(out), rows, slots, dim, eps, _stream()),
                    "ln_stats_finalize")
    return out


def row_stats(x: torch.Tensor, eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows, dim, ldx = _rows2d(x, "x")
    if out is None:
        out = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    with _Timed("ln_stats", 0, rows * dim * x.element_size()):
        _capi.check(_capi.load().leclip_row_stats_fwd(_ptr(x), _ptr(out), rows, dim, ldx, eps, dtype_code(x.dtype), _stream()), "row_stats")
    return out


# ------------------------------------------------------------------------------------------- backward (prompt tuning)
def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, add: Optional[torch.Tensor] = None,
                  eps: float = 1e-5) -> torch.Tensor:
    """dx of LayerNorm (gamma frozen) applied to dy, plus ``add`` (upstream gradient of the residual branch)."""
    _dev(dy, "dy"), _dev(x, "x")
    assert dy.shape == x.shape and dy.dtype == x.dtype and dy.is_contiguous() and x.is_contiguous()
    assert add is None or (add.shape == x.shape and add.dtype == x.dtype and add.is_contiguous())
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    _capi.check(_capi.load().leclip_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(_dev(gamma, "gamma")), _ptr(add), _ptr(dx), rows, dim, eps,
                                                  dtype_code(x.dtype), _stream()), "layernorm_bwd")
    return dx


def quickgelu(pre: torch.Tensor) -> torch.Tensor:
    _dev(pre, "pre")
    assert pre.is_contiguous()
    out = torch.empty_like(pre)
    _capi.check(_capi.load().leclip_quickgelu_fwd(_ptr(pre), _ptr(out), pre.numel(), dtype_code(pre.dtype), _stream()), "quickgelu_fwd")
    return out


def quickgelu_bwd(pre: torch.Tensor, du: torch.Tensor) -> torch.Tensor:
    _dev(pre, "pre"), _dev(du, "du")
    assert pre.shape == du.shape and pre.dtype == du.dtype and pre.is_contiguous() and du.is_contiguous()
    out = torch.empty_like(pre)
    _capi.check(_capi.load().leclip_quickgelu_bwd(_ptr(pre), _ptr(du), _ptr(out), pre.numel(), dtype_code(pre.dtype), _stream()),
                "quickgelu_bwd")
    return out


def attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, batch: int, tokens: int, heads: int, causal: bool) -> torch.Tensor:
    rows, width3, ld = _rows2d(qkv, "qkv")
    d = heads * 64
    assert rows == batch * tokens and width3 == 3 * d and dout.shape == (rows, d) and dout.dtype == qkv.dtype and dout.is_contiguous()
    dqkv = torch.empty_like(qkv)
    _capi.check(_capi.load().leclip_attention_bwd(_ptr(qkv), _ptr(_dev(dout, "dout")), _ptr(dqkv), batch, tokens, heads, 64, ld, d,
                                                  MASK_CAUSAL if causal else MASK_NONE, 0.125, dtype_code(qkv.dtype), _stream()),
                "attention_bwd")
    return dqkv


# ------------------------------------------------------------------------------------- score post-processing (N2 / N3)
def window_aggregate(global_logits: torch.Tensor, window_logits: torch.Tensor, threshold: float = 0.3, weight: float = 1.4) -> torch.Tensor:
    """out = weight * s_ag + global with s_ag = max over windows if it exceeds ``threshold`` else min (CDD.py:654-660)."""
    _dev(global_logits, "global_logits"), _dev(window_logits, "window_logits")
    g, w = global_logits.float().contiguous(), window_logits.float().contiguous()
    b, c = g.shape
    assert w.dim() == 3 and w.shape[0] == b and w.shape[2] == c
    out = torch.empty_like(g)
    _capi.check(_capi.load().leclip_window_aggregate_fwd(_ptr(g), _ptr(w), _ptr(out), b, w.shape[1], c, threshold, weight, _stream()),
                "window_aggregate")
    return out


def cooccurrence_matrix(adj, nums) -> torch.Tensor:
    """Row-normalised conditional co-occurrence matrix of CDD.py:632-634 from ``freq_stats.pkl``'s {'adj', 'nums'} (host side)."""
    import numpy as np
    p = np.asarray(adj, dtype=np.float64) / np.asarray(nums, dtype=np.float64)[:, None]
    p = p / p.sum(-1)[:, None]
    return torch.from_numpy(p.astype(np.float32))


def cooccurrence_adjust(p: torch.Tensor, mn: torch.Tensor, weight: float = 0.5) -> torch.Tensor:
    """p + weight * (p @ Mn) (adjust_predictions, CDD.py:614-618)."""
    _dev(p, "p"), _dev(mn, "Mn")
    p, mn = p.float().contiguous(), mn.float().contiguous()
    b, c = p.shape
    assert mn.shape == (c, c)
    out = torch.empty_like(p)
    _capi.check(_capi.load().leclip_cooccurrence_adjust_fwd(_ptr(p), _ptr(mn), _ptr(out), b, c, weight, _stream()), "cooccurrence_adjust")
    return out


# ------------------------------------------------------------------------------------------- fused image tail / helpers
def image_tail(x: torch.Tensor, batch: int, row_stride: int, gamma: torch.Tensor, beta: torch.Tensor, proj_t: torch.Tensor,
               txt: Optional[torch.Tensor] = None, scale: float = 4.0, want_features: bool = False, eps: float = 1e-5):
    """ln_post(class rows) @ proj -> (features [B,E] fp32 or None, logits [B,C] fp32 or None) in one launch
    (leclip_image_tail_fwd).  ``x`` holds image b's class-token row at element offset ``b * row_stride``."""
    _dev(x, "x"), _dev(proj_t, "proj_t")
    e, dim = proj_t.shape
    assert proj_t.is_contiguous() and proj_t.dtype == x.dtype and x.is_contiguous()
    feat = torch.empty((batch, e), dtype=torch.float32, device=x.device) if (want_features or txt is None) else None
    logits, c = None, 0
    if txt is not None:
        _dev(txt, "text_features")
        assert txt.dtype == torch.float32 and txt.is_contiguous() and txt.shape[1] == e
        c = txt.shape[0]
        logits = torch.empty((batch, c), dtype=torch.float32, device=x.device)
    with _Timed("image_tail", 2 * batch * e * (dim + c), batch * dim * x.element_size() + e * dim * x.element_size() + c * e * 4):
        _capi.check(_capi.load().leclip_image_tail_fwd(_ptr(x), _ptr(_dev(gamma, "gamma")), _ptr(_dev(beta, "beta")), _ptr(proj_t), _ptr(txt),
                                                       _ptr(feat), _ptr(logits), batch, row_stride, dim, e, c, eps, float(scale),
                                                       dtype_code(x.dtype), _stream()), "image_tail")
    return feat, logits


def l2norm_logits_bwd(img: torch.Tensor, txt: torch.Tensor, dlogits: torch.Tensor, scale: float) -> torch.Tensor:
    """d(text features) of ``l2norm_logits`` (image features frozen)."""
    for name, t in (("image_features", img), ("text_features", txt), ("dlogits", dlogits)):
        _dev(t, name)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError(f"l2norm_logits_bwd: {name} must be contiguous float32")
    b, d = img.shape
    c = txt.shape[0]
    assert txt.shape == (c, d) and dlogits.shape == (b, c)
    out = torch.empty_like(txt)
    _capi.check(_capi.load().leclip_l2norm_logits_bwd(_ptr(img), _ptr(txt), _ptr(dlogits), _ptr(out), b, c, d, float(scale), _stream()),
                "l2norm_logits_bwd")
    return out


def gather_rows(src: torch.Tensor, index: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src[index] for a 2-D row-major tensor (device-side row copy, no ATen indexing kernel)."""
    rows, dim, ld = _rows2d(src, "src")
    assert index.dtype == torch.int64 and index.is_contiguous()
    if out is None:
        out = torch.empty((index.numel(), dim), dtype=src.dtype, device=src.device)
    _, odim, ldo = _rows2d(out, "out")
    assert odim == dim and out.dtype == src.dtype and out.shape[0] == index.numel()
    _capi.check(_capi.load().leclip_gather_rows_fwd(_ptr(src), _ptr(_dev(index, "index")), _ptr(out), index.numel(), dim, ld, ldo,
                                                    dtype_code(src.dtype), _stream()), "gather_rows")
    return out


def scatter_rows(src: torch.Tensor, index: torch.Tensor, n_rows: int) -> torch.Tensor:
    """zeros([n_rows, dim]) with row index[i] = src[i] (indices distinct)."""
    n, dim, ld = _rows2d(src, "src")
    assert index.dtype == torch.int64 and index.is_contiguous() and index.numel() == n
    out = torch.empty((n_rows, dim), dtype=src.dtype, device=src.device)
    _capi.check(_capi.load().leclip_scatter_rows_fwd(_ptr(src), _ptr(_dev(index, "index")), _ptr(out), n, n_rows, dim, ld, dim,
                                                     dtype_code(src.dtype), _stream()), "scatter_rows")
    return out


def crop_resize(src_u8: torch.Tensor, windows: torch.Tensor, size: int, mean, std, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """[B,3,H,W] uint8 images x [NW,5] int32 windows (y0, x0, rows, cols, pad_top) -> [B, NW, 3, size, size]: the reference's
    per-window test transform (bicubic Resize on the smaller edge, CenterCrop, ToTensor, Normalize) on the device,
    bit-compatible with Pillow's resampler (leclip_crop_resize_fwd)."""
    import ctypes
    _dev(src_u8, "src"), _dev(windows, "windows")
    if src_u8.dim() == 3:
        src_u8 = src_u8.unsqueeze(0)
    src_u8 = src_u8.contiguous()
    assert src_u8.dtype == torch.uint8 and src_u8.shape[1] == 3
    assert windows.dtype == torch.int32 and windows.is_contiguous() and windows.dim() == 2 and windows.shape[1] == 5
    b, _, h, w = src_u8.shape
    nw = windows.shape[0]
    out = torch.empty((b, nw, 3, size, size), dtype=out_dtype, device=src_u8.device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    _capi.check(_capi.load().leclip_crop_resize_fwd(_ptr(src_u8), b, h, w, _ptr(windows), nw, _ptr(out), size, m3, s3, dtype_code(out_dtype),
                                                    _stream()), "crop_resize")
    return out


# ---------------------------------------------------------------------------------------------- local (dense) branch, N4
def l2norm_rows_(x: torch.Tensor) -> torch.Tensor:
    """In-place row-wise L2 normalisation of an fp32 matrix."""
    rows, dim, ld = _rows2d(x, "x")
    assert x.dtype == torch.float32
    _capi.check(_capi.load().leclip_l2norm_rows_fwd(_ptr(x), rows, dim, ld, _stream()), "l2norm_rows")
    return x


def local_pool(sim: torch.Tensor, batch: int, tokens: int, first: int, n_cls: int, evidence_offset: int, spatial_scale: float,
               logit_scale: float, mask_tokens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sim [batch * tokens, ld] fp32 (positions ``first`` .. tokens-1 of every image are pooled) -> logits_local [batch, n_cls].
    ``mask_tokens`` [batch, tokens] int64 (caption-as-image branch): positions whose token id is 0 get the reference's -10000."""
    _dev(sim, "sim")
    assert sim.dtype == torch.float32 and sim.is_contiguous() and sim.shape[0] == batch * tokens
    ld = sim.shape[1]
    out = torch.empty((batch, n_cls), dtype=torch.float32, device=sim.device)
    view = sim[first:]          # skip the class-token row of image 0; image stride stays tokens * ld
    mask = _mask_arg(mask_tokens, batch, tokens, first, sim.device)
    _capi.check(_capi.load().leclip_local_pool_masked_fwd(_ptr(view), _ptr(mask), tokens, _ptr(out), batch, tokens - first, n_cls, ld, tokens * ld,
                                                          evidence_offset, float(spatial_scale), float(logit_scale), _stream()), "local_pool")
    return out


def topk_mix(img: torch.Tensor, feats: torch.Tensor, k: int = 10) -> torch.Tensor:
    """(img + mean of the k rows of ``feats`` most similar to img) / 2 per row of ``img`` - the caption-feature mixing of
    Caption_distill_double.py:444-448.  img [B, E] and feats [N, E] fp32, both already normalised; the similarity panel runs on the
    exact-fp32 MFMA GEMM (its N granularity is 64 rows: a table that is a view of a zero-padded one - as DenseCLIP.set_caption_text_feats
    keeps it - is used in place, anything else is padded per call)."""
    _dev(img, "img")
    _dev(feats, "feats")
    if img.dtype != torch.float32 or feats.dtype != torch.float32 or not img.is_contiguous() or not feats.is_contiguous() or img.shape[1] != feats.shape[1]:
        raise TypeError("topk_mix: img [B, E] and feats [N, E] must be contiguous float32")
    n, e = feats.shape
    npad = (n + 63) // 64 * 64
    base = feats._base if feats._base is not None else None
    if npad == n:
        w = feats
    elif base is not None and base.dim() == 2 and base.shape == (npad, e) and base.data_ptr() == feats.data_ptr() and base.is_contiguous():
        w = base       # the caller's table is a view of one already padded to the GEMM's N granularity (DenseCLIP.set_caption_text_feats)
    else:              # ad-hoc call: pad here; the padded columns are never selected (the kernel scans [0, n))
        w = torch.zeros((npad, e), dtype=torch.float32, device=feats.device)
        w[:n] = feats
    sim = gemm(img, w, out_dtype=torch.float32)                   # [B, npad]
    out = torch.empty_like(img)
    _capi.check(_capi.load().leclip_topk_mix_fwd(_ptr(sim), _ptr(feats), _ptr(img), _ptr(out), img.shape[0], n, e, int(k), npad, _stream()), "topk_mix")
    return out


def _mask_arg(mask_tokens, batch, tokens, first, device):
    if mask_tokens is None:
        return None
    _dev(mask_tokens, "mask_tokens")
    if mask_tokens.dtype != torch.int64 or not mask_tokens.is_contiguous() or mask_tokens.shape != (batch, tokens) or mask_tokens.device != device:
        raise TypeError("local_pool: mask_tokens must be a contiguous int64 [batch, tokens] tensor on the similarity panel's device")
    return mask_tokens.view(-1)[first:]


def local_pool_bwd(sim: torch.Tensor, dout: torch.Tensor, batch: int, tokens: int, first: int, n_cls: int, evidence_offset: int,
                   spatial_scale: float, logit_scale: float, mask_tokens: Optional[torch.Tensor] = None, transposed: bool = False):
    """Gradient of ``local_pool`` w.r.t. the two similarity panels: (dneg, devi or None), each [batch * (tokens - first), n_cls] fp32 -
    or, ``transposed``, [n_cls, rows padded to a multiple of 32] with zero pad columns (the K-contiguous GEMM operand)."""
    _dev(sim, "sim")
    _dev(dout, "dout")
    assert sim.dtype == torch.float32 and sim.is_contiguous() and sim.shape[0] == batch * tokens
    if dout.dtype != torch.float32 or not dout.is_contiguous() or dout.shape != (batch, n_cls):
        raise TypeError("local_pool_bwd: dout must be contiguous float32 [batch, n_cls]")
    ld = sim.shape[1]
    p = tokens - first
    rows = batch * p
    if transposed:
        t_ld = (rows + 31) // 32 * 32
        make = lambda: (torch.zeros if t_ld != rows else torch.empty)((n_cls, t_ld), dtype=torch.float32, device=sim.device)
    else:
        t_ld = 0
        make = lambda: torch.empty((rows, n_cls), dtype=torch.float32, device=sim.device)
    dneg = make()
    devi = make() if evidence_offset >= 0 else None
    mask = _mask_arg(mask_tokens, batch, tokens, first, sim.device)
    _capi.check(_capi.load().leclip_local_pool_bwd(_ptr(sim[first:]), _ptr(mask), tokens, _ptr(dout), _ptr(dneg), _ptr(devi), batch, p, n_cls, ld, tokens * ld,
                                                   evidence_offset, float(spatial_scale), float(logit_scale), t_ld, _stream()), "local_pool_bwd")
    return dneg, devi


def transpose_f32(src: torch.Tensor, pad_cols_to: int = 32) -> torch.Tensor:
    """[cols, rows padded to a multiple of ``pad_cols_to``] = src^T for a contiguous fp32 [rows, cols] matrix (pad columns zero)."""
    _dev(src, "src")
    if src.dtype != torch.float32 or src.dim() != 2 or not src.is_contiguous():
        raise TypeError("transpose_f32: contiguous float32 [rows, cols] expected")
    rows, cols = src.shape
    ld = (rows + pad_cols_to - 1) // pad_cols_to * pad_cols_to
    dst = (torch.zeros if ld != rows else torch.empty)((cols, ld), dtype=torch.float32, device=src.device)
    _capi.check(_capi.load().leclip_transpose_f32_fwd(_ptr(src), _ptr(dst), rows, cols, cols, ld, _stream()), "transpose_f32")
    return dst


def l2norm_rows_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """Backward of the row normalisation y = x / |x|: dx = (dy - y <y, dy>) / |x| (fp32 [rows, dim], contiguous)."""
    for name, t in (("x", x), ("dy", dy)):
        _dev(t, name)
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2:
            raise TypeError(f"l2norm_rows_bwd: {name} must be contiguous float32 [rows, dim]")
    assert x.shape == dy.shape
    dx = torch.empty_like(x)
    _capi.check(_capi.load().leclip_l2norm_rows_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.shape[0], x.shape[1], _stream()), "l2norm_rows_bwd")
    return dx
