#!/usr/bin/env python3
"""Round 5 (VERDICT r4 task 7): ONE measurement of the variant the bf16 error budget priced but nobody ran - bf16 GEMM / attention operands with
an fp32 RESIDUAL STREAM - on the HIP kernels, against the reference's own logits at N = 2 048 (tests/golden/vitb16_cfg4_logits.npz), next to the
product's bf16 mode (16-bit stream) on the same images.

The variant is assembled here from the library's operators (it is not an engine mode): x lives in fp32; LayerNorm reads fp32 and writes the bf16 GEMM
operand (its own kernel: the LayerNorm fold needs the stream in the operand dtype); out-proj and c_proj add their bf16-operand, fp32-accumulated
result to the fp32 stream (generic epilogue: fp32 residual in, fp32 out); ln_post / projection / logits as in the product.  2.5 x the residual traffic and
four extra row-wise passes per block, so the rate printed beside it is that of an unfused path.

    gpurun -- python profiles/bf16_fp32_stream.py > gpurun_out/r05_bf16_fp32_stream.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leclip_amd import synth                                          # noqa: E402
from leclip_amd.clip import build_model, convert_weights             # noqa: E402
from leclip_amd.config import get_cfg_default                        # noqa: E402
from leclip_amd.datasets import coco_object_categories               # noqa: E402
from leclip_amd.evaluation import mAP                                # noqa: E402
from leclip_amd.hip import ops                                       # noqa: E402
from leclip_amd.trainers import CustomCLIP                           # noqa: E402

DEV = torch.device("cuda", 0)


def tower_fp32_stream(eng, image):
    """The image tower with bf16 operands and an fp32 residual stream -> class-row features [B, E] fp32."""
    b = image.shape[0]
    t, d = eng.tokens, eng.width
    x = ops.patch_embed(image.to(eng.dtype).contiguous(), eng.wp, eng.cls, eng.pos, eng.patch, torch.float32)       # [B, T, d] fp32
    x = x.view(b * t, d)
    ops.layernorm(x, eng.ln_pre_w, eng.ln_pre_b, out=x)
    h = torch.empty((b * t, d), dtype=eng.dtype, device=DEV)
    for p in eng.blocks:
        ops.layernorm(x, p.ln1_w, p.ln1_b, out=h)
        qkv = ops.gemm(h, p.w_qkv, p.b_qkv)
        ctx = ops.attention(qkv, b, t, eng.heads, False)
        ops.gemm(ctx, p.w_o, p.b_o, residual=x, out=x)
        ops.layernorm(x, p.ln2_w, p.ln2_b, out=h)
        u = ops.gemm(h, p.w_fc, p.b_fc, act=ops.ACT_QUICKGELU)
        ops.gemm(u, p.w_pr, p.b_pr, residual=x, out=x)
    rows = (torch.arange(b, device=DEV, dtype=torch.int64) * t).contiguous()
    return ops.gather_ln_proj(x, rows, eng.ln_post_w, eng.ln_post_b, eng.proj)


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "vitb16_cfg4_logits.npz"))
    ref = g["logits"]
    labels = np.unpackbits(g["labels"], axis=1)[:, :int(g["n_classes"])].astype(np.int64)
    m_ref = float(g["mAP_reference"])
    model = build_model(synth.make_state_dict(synth.VIT_B16, seed=0, dist="cond")).float()
    convert_weights(model, torch.bfloat16)
    cc = CustomCLIP(get_cfg_default(), coco_object_categories, model)
    with torch.no_grad():
        cc.prompt_learner.ctx.copy_(torch.from_numpy(synth.make_ctx(16, 512, seed=0)))
    cc = cc.to(DEV).eval()
    eng = cc.image_encoder.engine(DEV)
    out = {"bf16 (16-bit residual stream: the product's bf16 mode)": [], "bf16 operands / fp32 residual stream": []}
    with torch.no_grad():
        txt = cc.class_text_features().float().contiguous()
        for r in range(8):
            img = torch.from_numpy(synth.make_images(256, 224, seed=1234, start=256 * r)).to(DEV)
            out["bf16 (16-bit residual stream: the product's bf16 mode)"].append(cc(img, if_test=True)[0].float().cpu().numpy())
            out["bf16 operands / fp32 residual stream"].append(ops.l2norm_logits(tower_fp32_stream(eng, img), txt, 4.0).cpu().numpy())
        for name, parts in out.items():
            hip = np.concatenate(parts)
            r1, h1 = ref.argmax(1), hip.argmax(1)
            dis = np.nonzero(r1 != h1)[0]
            worst = max([float(ref[i, r1[i]] - ref[i, h1[i]]) for i in dis], default=0.0)
            m = mAP(labels, hip)
            print(f"{name}: N = {len(hip)}, max |dlogit| {np.abs(hip - ref).max():.3e}, {len(dis)} top-1 flips (worst reference margin {worst:.2e}), "
                  f"mAP {m:.3f} vs reference {m_ref:.3f}: delta {m - m_ref:+.3f}  -> +-0.2 clause {'MET' if abs(m - m_ref) <= 0.2 else 'missed'}")
        img = torch.from_numpy(synth.make_images(256, 224, seed=1234)).to(DEV)
        for name, fn in (("product bf16", lambda: cc(img, if_test=True)[0]), ("fp32-stream variant", lambda: ops.l2norm_logits(tower_fp32_stream(eng, img), txt, 4.0))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            print(f"rate, {name}: {256 * 10 / (time.perf_counter() - t0):.0f} img/s at B = 256")


if __name__ == "__main__":
    main()
