#!/usr/bin/env python3
"""Where the 16-bit modes spend their mantissa (CPU experiment, dev-only; evidence for DESIGN.md §3/§6).

Emulates the HIP image tower's rounding points on the CPU oracle's arithmetic (fp32 matmuls on operands rounded to the
16-bit type = MFMA with fp32 accumulate) and reports max |dlogit| and the mAP gap against the fp32 oracle on the same
256 images / labels the GPU test `test_map_against_oracle` uses.  Variants:

    act      activations (qkv, probabilities, attention output, MLP hidden) and GEMM A operands rounded to this type
    stream   storage type of the residual stream x (what the out-proj / c_proj epilogues write back)

Run:  python profiles/r02_lowprec_error_budget.py [n_images]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leclip_amd import synth                      # noqa: E402
from leclip_amd.evaluation import mAP             # noqa: E402
from oracle import clip_oracle as co              # noqa: E402


def rnd(t, dt):
    return t if dt is None else t.to(dt).float()


def tower(img, sd, act, stream):
    width = sd["visual.conv1.weight"].shape[0]
    heads = width // 64
    x = co.patch_embed(rnd(img, act), {**sd, "visual.conv1.weight": rnd(sd["visual.conv1.weight"], act)})
    x = rnd(x, stream)
    x = rnd(co.layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]), stream)
    n, t, d = x.shape
    for i in range(co._n_layers(sd, "visual.transformer.")):
        p = f"visual.transformer.resblocks.{i}."
        W = lambda k: rnd(sd[p + k], act)
        # LayerNorm is folded into the GEMM: statistics from the stored x, A operand = stored x rounded to the MFMA type
        xa = rnd(x, act)
        mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
        h = (xa - mu) * torch.rsqrt(var + 1e-5) * sd[p + "ln_1.weight"] + sd[p + "ln_1.bias"]
        qkv = rnd(h @ W("attn.in_proj_weight").t() + sd[p + "attn.in_proj_bias"], act)
        q, k, v = [z.reshape(n, t, heads, 64).transpose(1, 2) for z in qkv.split(d, dim=-1)]
        pr = torch.softmax((q @ k.transpose(-1, -2)) / 8.0, dim=-1)
        den = pr.sum(-1, keepdim=True)
        o = rnd((rnd(pr, act) @ v) / den, act).transpose(1, 2).reshape(n, t, d)
        x = rnd(x + o @ W("attn.out_proj.weight").t() + sd[p + "attn.out_proj.bias"], stream)
        xa = rnd(x, act)
        mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
        h = (xa - mu) * torch.rsqrt(var + 1e-5) * sd[p + "ln_2.weight"] + sd[p + "ln_2.bias"]
        u = rnd(co.quick_gelu(h @ W("mlp.c_fc.weight").t() + sd[p + "mlp.c_fc.bias"]), act)
        x = rnd(x + u @ W("mlp.c_proj.weight").t() + sd[p + "mlp.c_proj.bias"], stream)
    x = co.layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return rnd(x, act) @ rnd(sd["visual.proj"], act)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.set_num_threads(8)
    sd = synth.make_state_dict(synth.VIT_B16, seed=0, dist="cond")
    ctx = torch.from_numpy(synth.make_ctx(16, 512, seed=0))
    toks = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "tokens_coco80.npz"))["tokens_ctx16"])
    prefix, suffix = co.prompt_buffers(toks, sd, 16)
    with torch.no_grad():
        txt = co.text_encoder(co.prompt_learner_forward(ctx, prefix, suffix), toks, sd)
        img = torch.from_numpy(synth.make_images(n, 224, seed=4321))
        run = lambda a, s: torch.cat([co.cosine_logits(tower(img[i:i + 32], sd, a, s), txt, 4.0) for i in range(0, n, 32)]).numpy()
        ref = run(None, None)
        labels = synth.make_labels_from_logits(ref, seed=7, pos_frac=0.1, noise=0.5)
        m_ref = mAP(labels, ref)
        print(f"fp32 oracle arithmetic: mAP {m_ref:.3f} on {n} images")
        bf, hf = torch.bfloat16, torch.float16
        for name, a, s in (("bf16 act, bf16 stream (shipped bf16 mode)", bf, bf), ("bf16 act, fp32 stream", bf, None),
                           ("bf16 act, fp16 stream", bf, hf), ("fp16 act, fp16 stream (shipped fp16 mode)", hf, hf),
                           ("fp32 act, bf16 stream", None, bf)):
            lg = run(a, s)
            print(f"{name:46s} max|dlogit| {np.abs(lg - ref).max():.2e}  rms {np.sqrt(((lg - ref) ** 2).mean()):.2e}  "
                  f"mAP {mAP(labels, lg):.3f} (d {mAP(labels, lg) - m_ref:+.3f})  top1 agree {(lg.argmax(1) == ref.argmax(1)).mean():.4f}")


if __name__ == "__main__":
    main()
