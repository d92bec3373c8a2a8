"""Autograd bridges for prompt tuning: gradients flow from the text features back to the learnable context vectors
through the frozen text tower (reference trainers/Caption_distill_double.py:762-765, 789-897).

Only ACTIVATION gradients are formed (every CLIP weight is frozen), so the backward of each linear layer is the
forward TN GEMM kernel on a transposed weight copy; LayerNorm, QuickGELU and attention have their own backward
kernels (``csrc/backward.hip``).  ``torch.autograd.Function`` is the seam: above it ordinary autograd (loss, SGD on
``ctx``), below it only ``leclip_*`` kernels.
"""
from __future__ import annotations

import torch

from . import ops


class _BwdWeights:
    """Transposed copies of one block's weights ([in, out] with `out` contiguous), packed on first backward."""
    __slots__ = ("qkv_t", "o_t", "fc_t", "pr_t")

    def __init__(self, p):
        self.qkv_t = p.w_qkv.t().contiguous()   # [d, 3d]
        self.o_t = p.w_o.t().contiguous()       # [d, d]
        self.fc_t = p.w_fc.t().contiguous()     # [d, 4d]
        self.pr_t = p.w_pr.t().contiguous()     # [4d, d]


def _bwd_weights(engine):
    if getattr(engine, "_bwd", None) is None:
        engine._bwd = [_BwdWeights(p) for p in engine.blocks]
    return engine._bwd


class TextTowerFunction(torch.autograd.Function):
    """features = TextEncoder(prompts) (CDD.py:82-101) with a hand-written backward w.r.t. ``prompts``."""

    @staticmethod
    def forward(ctx, prompts: torch.Tensor, engine, tokenized_prompts: torch.Tensor):
        n, t, d = prompts.shape
        heads = engine.heads
        x = ops.add_pos(prompts.detach().to(dtype=torch.float32), engine.pos, engine.dtype).view(n * t, d)
        saved = []
        for p in engine.blocks:
            h1 = ops.layernorm(x, p.ln1_w, p.ln1_b)
            qkv = ops.gemm(h1, p.w_qkv, p.b_qkv)
            att = ops.attention(qkv, n, t, heads, True)
            x_mid = ops.gemm(att, p.w_o, p.b_o, residual=x)
            h2 = ops.layernorm(x_mid, p.ln2_w, p.ln2_b)
            pre = ops.gemm(h2, p.w_fc, p.b_fc)
            u = ops.quickgelu(pre)
            x_out = ops.gemm(u, p.w_pr, p.b_pr, residual=x_mid)
            saved.append((x, qkv, x_mid, pre))
            x = x_out
        _, flat = ops.eot_index(tokenized_prompts.to(prompts.device).contiguous())
        feats = ops.gather_ln_proj(x, flat, engine.ln_w, engine.ln_b, engine.proj)
        ctx.engine, ctx.saved, ctx.x_last, ctx.flat, ctx.shape = engine, saved, x, flat, (n, t, d)
        return feats

    @staticmethod
    def backward(ctx, dfeat: torch.Tensor):
        engine, (n, t, d) = ctx.engine, ctx.shape
        dt = engine.dtype
        bw = _bwd_weights(engine)
        # features = LN(x[eot]) @ proj: d(LN out) = dfeat @ proj^T  (proj [d, E] is already the [N'=d, K'=E] layout)
        dln = ops.gemm(dfeat.to(dt).contiguous(), engine.proj)
        rows = ops.gather_rows(ctx.x_last, ctx.flat)
        drows = ops.layernorm_bwd(dln, rows, engine.ln_w)
        dx = ops.scatter_rows(drows, ctx.flat, n * t)        # zero everywhere but the EOT rows
        for p, w, (x_in, qkv, x_mid, pre) in zip(reversed(engine.blocks), reversed(bw), reversed(ctx.saved)):
            du = ops.gemm(dx, w.pr_t)                                   # through c_proj
            dpre = ops.quickgelu_bwd(pre, du)
            dh2 = ops.gemm(dpre, w.fc_t)                                # through c_fc
            dxm = ops.layernorm_bwd(dh2, x_mid, p.ln2_w, add=dx)        # ln_2 + the residual branch
            dctx = ops.gemm(dxm, w.o_t)                                 # through out_proj
            dqkv = ops.attention_bwd(qkv, dctx, n, t, engine.heads, True)
            dh1 = ops.gemm(dqkv, w.qkv_t)                               # through in_proj
            dx = ops.layernorm_bwd(dh1, x_in, p.ln1_w, add=dxm)         # ln_1 + the residual branch
        return dx.view(n, t, d).float(), None, None


class PromptAssembleFunction(torch.autograd.Function):
    """prompts = cat(prefix, ctx, suffix) (CDD.py:206-225); backward: the context slice, summed over classes for a
    generic (class-shared) context."""

    @staticmethod
    def forward(ctx, ctx_vectors: torch.Tensor, prefix: torch.Tensor, suffix: torch.Tensor):
        ctx.per_class = ctx_vectors.dim() == 3
        ctx.n_ctx = ctx_vectors.shape[-2]
        return ops.prompt_assemble(prefix, ctx_vectors.detach().float().contiguous(), suffix, None, torch.float32)

    @staticmethod
    def backward(ctx, dprompts: torch.Tensor):
        g = dprompts[:, 1:1 + ctx.n_ctx, :]
        return (g.contiguous() if ctx.per_class else g.sum(dim=0)), None, None


class CosineLogitsFunction(torch.autograd.Function):
    """logits = scale * normalize(img) @ normalize(txt).T (CDD.py:330-335) with the gradient w.r.t. the text features
    (image features come from frozen encoders: no gradient).  Backward: g = scale * dlogits^T . normalize(img) is a [C, B] x [B, D]
    contraction - the exact-fp32 MFMA GEMM on the two transposed operands (it splits a long B over workgroups by itself) - followed by the
    row-normalisation backward, the same three kernels the local head's backward uses; the one-kernel form (leclip_l2norm_logits_bwd:
    one workgroup per class walking every image row) remains for feature widths the GEMM does not take."""

    @staticmethod
    def forward(ctx, img: torch.Tensor, txt: torch.Tensor, scale: float):
        img = img.detach().float().contiguous()
        txt_d = txt.detach().float().contiguous()
        ctx.save_for_backward(img, txt_d)
        ctx.scale = float(scale)
        return ops.l2norm_logits(img, txt_d, scale)

    @staticmethod
    def backward(ctx, dlogits: torch.Tensor):
        img, txt = ctx.saved_tensors
        if img.shape[1] % 64 != 0:
            return None, ops.l2norm_logits_bwd(img, txt, dlogits.float().contiguous(), ctx.scale), None
        fhat_t = ops.transpose_f32(ops.l2norm_rows_(img.clone()))                          # [D, B padded to 32]
        dl_t = ops.transpose_f32(dlogits.float().mul(ctx.scale).contiguous())               # [C, B padded to 32]
        return None, ops.l2norm_rows_bwd(txt, ops.gemm(dl_t, fhat_t, out_dtype=torch.float32)), None


class LocalPoolFunction(torch.autograd.Function):
    """logits_local of the caption-as-image training branch (CDD.py:493-513): normalise the sequence features and the "negative"
    (and evidence) prompt features, similarity panels on the exact-fp32 MFMA GEMM, ``text_mask``, spatial softmax over the 77
    positions (winner-take-all weighting with evidence prompts), weighted sum.  Backward w.r.t. the two sets of text features: the
    pooling's own backward kernel (leclip_local_pool_bwd), then - per panel - the cosine-similarity backward already used for the
    global logits (leclip_l2norm_logits_bwd with the 77 positions of every caption as its "images").  The sequence features come
    from the frozen text tower: no gradient."""

    @staticmethod
    def forward(ctx, seq: torch.Tensor, txt_neg: torch.Tensor, txt_evi, tokens: torch.Tensor, spatial_scale: float, logit_scale: float):
        b, t, e = seq.shape
        flat = seq.detach().float().contiguous().view(b * t, e)
        tn = txt_neg.detach().float().contiguous()
        te = txt_evi.detach().float().contiguous() if txt_evi is not None else None
        sim, c_pad, fhat = local_similarity(flat, tn, te)
        c = tn.shape[0]
        evi = c_pad if te is not None else -1
        toks = tokens.to(device=seq.device, dtype=torch.int64).contiguous()
        out = ops.local_pool(sim, b, t, 0, c, evi, spatial_scale, logit_scale, mask_tokens=toks)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(fhat, tn, te if te is not None else tn, sim, toks)
            ctx.meta = (b, t, c, evi, float(spatial_scale), float(logit_scale), te is not None)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        fhat, tn, te, sim, toks = ctx.saved_tensors
        b, t, c, evi, spatial_scale, logit_scale, has_evi = ctx.meta
        dneg_t, devi_t = ops.local_pool_bwd(sim, dout.float().contiguous(), b, t, 0, c, evi, spatial_scale, logit_scale, mask_tokens=toks,
                                            transposed=True)                       # [C, rows_pad]
        fhat_t = ops.transpose_f32(fhat)                                           # [E, rows_pad]
        d_tn = ops.l2norm_rows_bwd(tn, ops.gemm(dneg_t, fhat_t, out_dtype=torch.float32))
        d_te = ops.l2norm_rows_bwd(te, ops.gemm(devi_t, fhat_t, out_dtype=torch.float32)) if has_evi else None
        return None, d_tn, d_te, None, None, None


def local_similarity(flat: torch.Tensor, txt_neg: torch.Tensor, txt_evi=None):
    """[rows, c_pad (x2)] fp32 cosine similarities of every (unnormalised) feature row against the negative (| evidence) prompt
    features: rows and prompts normalised by the row-norm kernel, contraction on the exact-fp32 MFMA GEMM; the prompt panel is
    zero-padded to 64-row blocks (the GEMM's N granularity).  Returns (sim, c_pad, normalised rows)."""
    rows = [ops.l2norm_rows_(txt_neg.clone())]
    if txt_evi is not None:
        rows.append(ops.l2norm_rows_(txt_evi.clone()))
    c = rows[0].shape[0]
    cp = (c + 63) // 64 * 64
    w = torch.zeros((cp * len(rows), rows[0].shape[1]), dtype=torch.float32, device=flat.device)
    for i, r in enumerate(rows):
        w[i * cp:i * cp + c] = r
    fhat = ops.l2norm_rows_(flat.clone())
    return ops.gemm(fhat, w, out_dtype=torch.float32), cp, fhat
