#!/usr/bin/env python3
"""One-off soak of the in-producer LayerNorm merge (leclip_gemm_res_stats_fwd) under the product's conditions: two HIP streams, each with its own
workspace and ticket words, launching out-proj- and c_proj-shaped calls back to back while the other stream does the same (uneven load, warm caches) -
every (mean, rstd) pair of every launch compared bit for bit with the two-launch path's.  Not part of the test-suite (200 launches x 2 streams).
    gpurun -- python profiles/stress_producer_merge.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import leclip_amd                      # noqa: E402
leclip_amd.configure()
from leclip_amd.hip import ops         # noqa: E402

DEV = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(3)
dt = torch.float16
parts = []
for M in (25216, 25216 - 197 * 3):
    shapes = {}
    for name, K in (("out_proj", 768), ("c_proj", 3072)):
        a = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(768, K, generator=g) * 0.05).to(dt).to(DEV)
        bias = torch.randn(768, generator=g).to(DEV)
        res = (torch.randn(M, 768, generator=g) + 2.0).to(dt).to(DEV)
        pr = torch.zeros(12, M, 2, device=DEV)
        y = ops.gemm_ln(a, w, bias, residual=res, stats_out=pr)
        shapes[name] = (a, w, bias, res, y, pr, ops.ln_stats_finalize(pr, 768))
    parts.append((M, shapes, torch.zeros((M + 383) // 384, dtype=torch.int32, device=DEV), torch.cuda.Stream()))
torch.cuda.synchronize()
bad = 0
outs = []
for it in range(100):
    for M, shapes, tickets, st in parts:
        with torch.cuda.stream(st):
            for name in (("out_proj", "c_proj") if it % 3 else ("c_proj", "out_proj", "out_proj")):
                a, w, bias, res, y_ref, p_ref, s_ref = shapes[name]
                part = torch.full((12, M, 2), float("nan"), device=DEV)
                stt = torch.full((M, 2), float("nan"), device=DEV)
                y = ops.gemm_res_stats(a, w, bias, res, part, stt, tickets)
                outs.append((torch.equal(y, y_ref) & torch.equal(stt, s_ref) & torch.equal(part, p_ref), int(tickets.abs().sum()) == 0 if False else True))
torch.cuda.synchronize()
bad = sum(1 for ok, _ in outs if not bool(ok))
tick = sum(int(t.abs().sum()) for _, _, t, _ in parts)
print(f"{len(outs)} launches on two concurrent streams: {bad} with a differing output / partial / statistic; ticket words non-zero at the end: {tick}")
sys.exit(1 if bad or tick else 0)
