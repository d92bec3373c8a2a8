"""CPU oracle: fp32 restatement of the reference's CLIP scoring path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may (the product path fails loudly when the HIP library is missing).

Every function restates, in plain ``torch`` CPU tensor algebra on a state-dict
(no ``nn.Module``), the arithmetic of one piece of the reference and cites it.
Pinned (parity is NOT unpinned): ``oracle/make_golden.py`` imports the reference's
own ``project/my_code/clip/model.py`` in the build container, loads the same
synthetic state-dicts, and writes its outputs to ``tests/golden/``;
``tests/test_oracle_golden.py`` holds this file to those vectors (<=2e-5 abs,
label indices exact) on every run, with or without ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """model.py:193-199 - LayerNorm evaluated in fp32 (eps = nn.LayerNorm default 1e-5)."""
    xf = x.float()
    mu = xf.mean(dim=-1, keepdim=True)
    var = ((xf - mu) ** 2).mean(dim=-1, keepdim=True)
    return ((xf - mu) * torch.rsqrt(var + eps) * w.float() + b.float()).to(x.dtype)


def quick_gelu(x: Tensor) -> Tensor:
    """model.py:202-204 - x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def causal_mask(n: int) -> Tensor:
    """model.py:364-370 - additive mask, -inf strictly above the diagonal."""
    m = torch.full((n, n), float("-inf"))
    return torch.triu(m, diagonal=1)


def multi_head_attention(x: Tensor, sd: Dict[str, Tensor], p: str, heads: int,
                         mask: Optional[Tensor], taps: Optional[dict] = None) -> Tensor:
    """model.py:211,221-223 (nn.MultiheadAttention, packed in-proj, q|k|v order).
    x: [N, T, d] (batch-major; the reference's LND layout is a pure permutation)."""
    n, t, d = x.shape
    dh = d // heads
    qkv = x @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(n, t, heads, dh).transpose(1, 2)
    k = k.reshape(n, t, heads, dh).transpose(1, 2)
    v = v.reshape(n, t, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if mask is not None:
        s = s + mask
    pr = torch.softmax(s, dim=-1)
    o = (pr @ v).transpose(1, 2).reshape(n, t, d)
    if taps is not None:
        taps["qkv"] = qkv
        taps["attn_ctx"] = o
    return o @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]


def residual_block(x: Tensor, sd: Dict[str, Tensor], p: str, heads: int, mask: Optional[Tensor],
                   taps: Optional[dict] = None) -> Tensor:
    """model.py:225-228 - x += attn(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    if taps is not None:
        taps["ln_1"] = h
    x = x + multi_head_attention(h, sd, p, heads, mask, taps)
    if taps is not None:
        taps["after_attn"] = x
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    u = quick_gelu(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
    if taps is not None:
        taps["ln_2"] = h
        taps["gelu"] = u
    x = x + (u @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"])
    if taps is not None:
        taps["out"] = x
    return x


def _n_layers(sd: Dict[str, Tensor], prefix: str) -> int:
    return len([k for k in sd if k.startswith(prefix) and k.endswith(".attn.in_proj_weight")])


def patch_embed(image: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """model.py:260-264 - stride-p conv without bias, flatten, prepend class token, add pos-emb."""
    w = sd["visual.conv1.weight"]
    width, _, p, _ = w.shape
    b, c, hh, ww = image.shape
    g = hh // p
    # conv with kernel == stride is a GEMM over non-overlapping patches, k = (c, py, px)
    patches = image.reshape(b, c, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, c * p * p)
    x = patches @ w.reshape(width, -1).t()
    cls = sd["visual.class_embedding"].expand(b, 1, width)
    return torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]


def encode_image(image: Tensor, sd: Dict[str, Tensor], taps: Optional[dict] = None) -> Tensor:
    """VisionTransformer.forward, model.py:259-276 (CLIP.encode_image, 376-377)."""
    image = image.float()
    width = sd["visual.conv1.weight"].shape[0]
    heads = width // 64
    x = patch_embed(image, sd)
    if taps is not None:
        taps["embed"] = x
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    if taps is not None:
        taps["ln_pre"] = x
    for i in range(_n_layers(sd, "visual.transformer.")):
        bt = {} if taps is not None else None
        x = residual_block(x, sd, f"visual.transformer.resblocks.{i}.", heads, None, bt)
        if taps is not None:
            taps[f"block{i}"] = bt
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    if taps is not None:
        taps["ln_post"] = x
    return x @ sd["visual.proj"]


def text_transformer(x: Tensor, sd: Dict[str, Tensor], taps: Optional[dict] = None) -> Tensor:
    """The shared middle of CLIP.encode_text (model.py:382-387) and TextEncoder.forward
    (Caption_distill_double.py:86-90): + positional embedding, causal blocks, ln_final."""
    d = sd["ln_final.weight"].shape[0]
    heads = d // 64
    t = x.shape[1]
    x = x.float() + sd["positional_embedding"][:t]
    mask = causal_mask(t)
    for i in range(_n_layers(sd, "transformer.")):
        bt = {} if taps is not None else None
        x = residual_block(x, sd, f"transformer.resblocks.{i}.", heads, mask, bt)
        if taps is not None:
            taps[f"block{i}"] = bt
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    if taps is not None:
        taps["ln_final"] = x
    return x


def text_encoder(prompts: Tensor, tokenized_prompts: Optional[Tensor], sd: Dict[str, Tensor],
                 if_embedding: bool = True, if_sequence: bool = False,
                 taps: Optional[dict] = None) -> Tensor:
    """TextEncoder.forward, Caption_distill_double.py:82-101.  With ``if_embedding=False`` the
    first argument holds token ids (83-85) and this is CLIP.encode_text (model.py:379-392)."""
    if not if_embedding:
        tokenized_prompts = prompts
        prompts = sd["token_embedding.weight"][prompts]
    x = text_transformer(prompts, sd, taps)
    if if_sequence:
        return x @ sd["text_projection"]
    eot = tokenized_prompts.argmax(dim=-1)  # EOT has the largest id (model.py:390)
    return x[torch.arange(x.shape[0]), eot] @ sd["text_projection"]


def encode_text(tokens: Tensor, sd: Dict[str, Tensor], taps: Optional[dict] = None) -> Tensor:
    return text_encoder(tokens, None, sd, if_embedding=False, taps=taps)


def prompt_learner_forward(ctx: Tensor, prefix: Tensor, suffix: Tensor) -> Tensor:
    """PromptLearner.forward, class_token_position == 'end' (Caption_distill_double.py:206-225):
    expand a generic [n_ctx,d] context over classes and concatenate [SOS | ctx | class.. EOT pad]."""
    n_cls = prefix.shape[0]
    if ctx.dim() == 2:
        ctx = ctx.unsqueeze(0).expand(n_cls, -1, -1)
    return torch.cat([prefix, ctx, suffix], dim=1)


def prompt_buffers(tokenized_prompts: Tensor, sd: Dict[str, Tensor], n_ctx: int):
    """PromptLearner.__init__, Caption_distill_double.py:177-184 - frozen SOS / class+EOT embeddings."""
    emb = sd["token_embedding.weight"][tokenized_prompts]
    return emb[:, :1, :], emb[:, 1 + n_ctx:, :]


def l2_normalize(f: Tensor) -> Tensor:
    """model.py:399-400 - f / ||f||_2, no epsilon."""
    return f / f.norm(dim=-1, keepdim=True)


def cosine_logits(image_features: Tensor, text_features: Tensor, scale: float) -> Tensor:
    """model.py:399-404 / Caption_distill_double.py:330-335.  Python precedence makes
    ``scale * img @ txt.t()`` equal ``(scale * img) @ txt.t()``."""
    return (scale * l2_normalize(image_features)) @ l2_normalize(text_features).t()


def clip_forward(image: Tensor, tokens: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """CLIP.forward -> logits_per_image, model.py:394-408 (scale = exp(logit_scale))."""
    return cosine_logits(encode_image(image, sd), encode_text(tokens, sd), float(sd["logit_scale"].exp()))


def custom_clip_forward(image: Tensor, sd: Dict[str, Tensor], ctx: Tensor, prefix: Tensor, suffix: Tensor,
                        tokenized_prompts: Tensor, scale: float = 4.0) -> Tensor:
    """CustomCLIP.forward(if_test=True) as intended (Caption_distill_double.py:323-337; the shipped
    5-way unpack of a 6-tuple at :326 is the bug SURVEY.md notes - only ``prompts`` is used)."""
    img = encode_image(image, sd)
    txt = text_encoder(prompt_learner_forward(ctx, prefix, suffix), tokenized_prompts, sd)
    return cosine_logits(img, txt, scale)


def custom_clip_forward_captions(captions: Tensor, sd: Dict[str, Tensor], ctx: Tensor, prefix: Tensor,
                                 suffix: Tensor, tokenized_prompts: Tensor, scale: float = 4.0) -> Tensor:
    """CustomCLIP.forward(if_test=False): captions stand in for images (:338-352)."""
    img = text_encoder(captions, None, sd, if_embedding=False)
    txt = text_encoder(prompt_learner_forward(ctx, prefix, suffix), tokenized_prompts, sd)
    return cosine_logits(img, txt, scale)


def encode_image_tokens(image: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """[B, T, E]: every token of the last block through ln_post and proj (model.py:271-274 applied to all rows, not only
    x[:, 0, :]).  Row 0 equals encode_image; rows 1.. are the ViT's per-position features of the local branch (the reference
    defines per-position features only for the ResNet, Caption_distill_double.py:409-410 - SURVEY.md §8f N4)."""
    image = image.float()
    heads = sd["visual.conv1.weight"].shape[0] // 64
    x = layer_norm(patch_embed(image, sd), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    for i in range(_n_layers(sd, "visual.transformer.")):
        x = residual_block(x, sd, f"visual.transformer.resblocks.{i}.", heads, None)
    return layer_norm(x, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]


def local_pool(logits_neg: Tensor, logits_evidence: Optional[Tensor], tmp_scale: float, logit_scale: float) -> Tensor:
    """Caption_distill_double.py:447-462 on similarity panels [positions, batch, classes] (pinned: make_golden.py executes those
    reference lines, tests/golden/postprocess.npz n4.*)."""
    if logits_evidence is not None:
        w = torch.softmax(tmp_scale * logits_neg * (logits_neg.max(-1)[0].unsqueeze(-1) + 1), -1)   # winner-take-all
        logits_neg = logits_neg * w
        prob_spatial = torch.softmax(logits_evidence * tmp_scale, dim=0)
    else:
        prob_spatial = torch.softmax(logits_neg * tmp_scale, dim=0)
    return torch.sum(logit_scale * logits_neg * prob_spatial, dim=0)


def mix_caption_features(image_feature_: Tensor, caption_text_feats: Tensor, topk: int = 10) -> Tensor:
    """Caption_distill_double.py:444-448: the normalised global feature averaged with the mean of its top-k most similar caption
    features (the reference hard-codes the RN50 width in `.view(-1, topk, 1024)`; the width in use here).  Pinned:
    tests/golden/caption_branch.npz mix.* (those reference lines executed by make_golden.py)."""
    sim_caption = image_feature_ @ caption_text_feats.float().t()
    _, idx = sim_caption.topk(topk, -1)
    selected = caption_text_feats[idx.view(-1)].view(-1, topk, caption_text_feats.shape[1]).mean(1)
    return torch.cat([image_feature_[:, None], selected[:, None]], 1).mean(1)


def caption_text_features(captions: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """generate_caption_text_features.py:82-88: normalised EOT-row text features of tokenised captions."""
    return l2_normalize(text_encoder(captions, None, sd, if_embedding=False))


def dense_clip_forward(image: Tensor, sd: Dict[str, Tensor], ctx: Tensor, ctx_double: Tensor, ctx_evidence: Optional[Tensor],
                       prefix: Tensor, suffix: Tensor, tokenized_prompts: Tensor, tmp_scale: float = 40.0, scale: float = 4.0,
                       caption_text_feats: Optional[Tensor] = None):
    """DenseCLIP.forward(if_test=True) (:401-462) with the ViT's patch tokens as positions: (logits_, logits_local); with
    ``caption_text_feats`` the global feature is mixed with its top-10 caption features first (:444-448)."""
    feats = encode_image_tokens(image, sd)                                   # [B, T, E]
    enc = lambda c: l2_normalize(text_encoder(prompt_learner_forward(c, prefix, suffix), tokenized_prompts, sd))
    text, text_neg = enc(ctx), enc(ctx_double)
    glob = l2_normalize(feats[:, 0])
    if caption_text_feats is not None:
        glob = mix_caption_features(glob, caption_text_feats)
    pos = l2_normalize(feats[:, 1:]).permute(1, 0, 2)                        # [P, B, E]
    logits_ = scale * glob @ text.t()
    logits_neg = pos @ text_neg.t()
    logits_evi = pos @ enc(ctx_evidence).t() if ctx_evidence is not None else None
    return logits_, local_pool(logits_neg, logits_evi, tmp_scale, scale)


def dense_clip_forward_captions(captions: Tensor, sd: Dict[str, Tensor], ctx: Tensor, ctx_double: Tensor, ctx_evidence: Optional[Tensor],
                                prefix: Tensor, suffix: Tensor, tokenized_prompts: Tensor, tmp_scale: float = 50.0, scale: float = 4.0):
    """DenseCLIP.forward(captions=..., if_test=False), Caption_distill_double.py:473-513 (fixed scales: IF_LEARN_SCALE and
    IF_LEARN_spatial_SCALE are False in every shipped config): (logits_, logits_local).  Differentiable w.r.t. the three contexts
    (torch autograd is the gradient oracle of the tuning step).  Pinned: make_golden.py executes those reference lines on the
    reference's own TextEncoder (tests/golden/caption_branch.npz)."""
    image_feat = text_encoder(captions, None, sd, if_embedding=False, if_sequence=True)                  # :474  [B, L, E]
    image_feature_ = image_feat[torch.arange(image_feat.shape[0]), captions.argmax(dim=-1)]             # :476
    image_features = image_feat.permute(1, 0, 2)                                                         # :477  [L, B, E]
    enc = lambda c: l2_normalize(text_encoder(prompt_learner_forward(c, prefix, suffix), tokenized_prompts, sd))
    text_features, text_features_neg = enc(ctx), enc(ctx_double)                                          # :480-488
    image_feature_ = l2_normalize(image_feature_)
    image_features = l2_normalize(image_features)
    text_mask = (captions == 0).long() * (-10000)                                                         # :491  [B, L]
    logits_ = scale * image_feature_ @ text_features.t()                                                  # :494
    logits_neg = image_features @ text_features_neg.t()                                                   # :495  [L, B, C]
    logits_neg = (logits_neg.permute(2, 1, 0) + text_mask[None, :, :]).permute(2, 1, 0)                  # :496-497
    logits_evi = None
    if ctx_evidence is not None:                                                                          # :500-509
        logits_evi = image_features @ enc(ctx_evidence).t()
        logits_evi = (logits_evi.permute(2, 1, 0) + text_mask[None, :, :]).permute(2, 1, 0)
    return logits_, local_pool(logits_neg, logits_evi, tmp_scale, scale)


def double_ranking_loss(output: Tensor, output_local: Optional[Tensor], label: Tensor, output_m: Optional[Tensor] = None,
                        output_local_m: Optional[Tensor] = None) -> Tensor:
    """forward_backward's LOSSFUNC == 'double_ranking' branch, Caption_distill_double.py:805-815 with trainers/utils.py:85-93
    (ranking_loss, scale_ = 1, margin_ = 1): both heads, plus - with the momentum copy's scores - the distillation term."""
    def rank(y_pred, y_true):
        y_true_ = y_true.float()
        tmp = 1 - y_pred[:, None, :] + y_pred[:, :, None]
        part = torch.maximum(torch.zeros_like(tmp), tmp) * y_true_[:, None, :] * (1 - y_true_[:, :, None])
        return part.sum(dim=-1).sum(dim=-1).mean()
    loss = rank(output, label)
    if output_local is not None:
        loss = loss + rank(output_local, label)
    if output_m is not None:
        kl = torch.nn.KLDivLoss(reduction="batchmean")
        loss = loss + kl(F.log_softmax(output, dim=-1), F.softmax(output_m, dim=-1)) \
            + kl(F.log_softmax(output_local, dim=-1), F.softmax(output_local_m, dim=-1)) * 10000
    return loss


def flatten_taps(taps: dict, prefix: str = "") -> Dict[str, np.ndarray]:
    out = {}
    for k, v in taps.items():
        if isinstance(v, dict):
            out.update(flatten_taps(v, prefix + k + "."))
        else:
            out[prefix + k] = v.detach().cpu().numpy()
    return out
