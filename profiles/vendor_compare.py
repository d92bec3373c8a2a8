#!/usr/bin/env python3
"""Same tensors, same process, interleaved rounds: the block GEMMs of ViT-B/16 at B=256 through this build's kernels (with their
fused epilogues) and through torch.nn.functional.linear (hipBLASLt; comparator only - never on the product path).
Run on the GPU box:  python profiles/vendor_compare.py [bf16|fp16]        (rocprofv3 --kernel-trace --stats on it names the vendor kernels)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leclip_amd.hip import ops  # noqa: E402

dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
dev = "cuda:0"
M = 256 * 197
g = torch.Generator(device="cpu").manual_seed(1)
shapes = [("qkv", 2304, 768, False, False), ("out_proj", 768, 768, True, False), ("c_fc", 3072, 768, False, True), ("c_proj", 768, 3072, True, False)]
rows = []
for name, N, K, res, gelu in shapes:
    # activations with the statistics of the model's (LayerNorm-scale values, a common offset), weights ~ 2/sqrt(K)
    a = (torch.randn(M, K, generator=g) * 0.8 + 0.1).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * (2.0 / K ** 0.5)).to(dt).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dt).to(dev) if res else None
    out = torch.empty(M, N, dtype=dt, device=dev)
    bl = b.to(dt)

    def ours():
        ops.gemm(a, w, b, residual=r, act=ops.ACT_QUICKGELU if gelu else ops.ACT_NONE, out=out)

    def vendor():
        torch.nn.functional.linear(a, w, bl)

    def t(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res_t = []
    for rnd in range(3):
        res_t.append((t(ours), t(vendor)))
    o = min(x[0] for x in res_t)
    v = min(x[1] for x in res_t)
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K} {dt}: ours (fused epilogue) {o:7.1f} us {fl / o * 1e-6:7.1f} TF/s | hipBLASLt (bias only) {v:7.1f} us {fl / v * 1e-6:7.1f} TF/s"
          f" | all rounds {[(round(x, 1), round(y, 1)) for x, y in res_t]}")
