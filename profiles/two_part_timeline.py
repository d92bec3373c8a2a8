#!/usr/bin/env python3
"""Occupancy timeline of the image tower's persistent kernels (GEMM 256x256 family, pipelined attention) from the DIAGNOSTIC library's
per-workgroup log (make diag; begin / end of every workgroup on the 100 MHz real-time counter).  Run on the GPU box:

    LECLIP_HIP_LIB=$PWD/language-enhanced-clip-for-multi-label-image-recognition_amd/lib/libleclip_hip_diag.so python profiles/two_part_timeline.py

Prints, for the batch as one part and as two stream parts: wall time of one step, the CU-time the logged workgroups hold (one workgroup
per CU for these kernels: LDS), the share of the 256 CUs that is busy, and where the idle CU-time sits (by the number of busy CUs)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from leclip_amd import synth  # noqa: E402
from leclip_amd.clip import build_model  # noqa: E402
from leclip_amd.config import get_cfg_default  # noqa: E402
from leclip_amd.datasets import coco_object_categories  # noqa: E402
from leclip_amd.hip import _capi  # noqa: E402
from leclip_amd.trainers import CustomCLIP  # noqa: E402

dev = torch.device("cuda:0")
lib = _capi.load()
if not hasattr(lib, "leclip_diag_set_wglog"):
    sys.exit("this needs the diagnostic library: make -C <pkg>/csrc diag, then LECLIP_HIP_LIB=<pkg>/lib/libleclip_hip_diag.so")
setter = lib.leclip_diag_set_wglog
setter.argtypes = [ctypes.c_void_p, ctypes.c_uint]
setter.restype = None
arch = synth.VIT_B16
cc = CustomCLIP(get_cfg_default(), coco_object_categories, build_model(synth.make_state_dict(arch, seed=0, dist="cond"))).to(dev).eval()
eng = cc.image_encoder.engine(dev)
eng.cls_last_block = False
img = torch.from_numpy(synth.make_images(256, 224, seed=1234)).to(dev)
CAP = 200000
N_CU = torch.cuda.get_device_properties(dev).multi_processor_count
NAMES = {0x200: "attention"}


def run(streams, steps=3):
    eng.streams = streams
    with torch.no_grad():
        cc.class_text_features()
        for _ in range(4):
            cc(img, if_test=True)
        torch.cuda.synchronize()
        log = torch.zeros(2 + 6 * CAP, dtype=torch.int64, device=dev)
        setter(log.data_ptr(), CAP)
        for _ in range(steps):
            cc(img, if_test=True)
        torch.cuda.synchronize()
        setter(None, 0)
    raw = log.cpu().numpy().view(np.uint64)
    n = int(min(raw[0], CAP))
    e = raw[2:2 + 6 * n].reshape(n, 6)
    tag, t0, t1 = (e[:, 0] >> np.uint64(32)).astype(np.int64), e[:, 1].astype(np.int64), e[:, 2].astype(np.int64)
    seq = (e[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    hw, xcc = ((e[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64), ((e[:, 3] >> np.uint64(48)) & np.uint64(0xf)).astype(np.int64)
    clk = (e[:, 5] - e[:, 4]).astype(np.float64) / np.maximum((e[:, 2] - e[:, 1]).astype(np.float64), 1.0) * 0.1   # GHz: shader cycles per 10 ns
    cu = xcc * 1024 + ((hw >> 13) & 7) * 128 + ((hw >> 12) & 1) * 64 + ((hw >> 8) & 15)      # (xcc, se, sh, cu) -> one id per CU
    # the middle step: launches are numbered in host order; split the sequence numbers evenly over the steps
    per = int(seq.max()) // steps
    sel = (seq > per) & (seq <= 2 * per)
    tag, t0, t1, seq, cu, clk = tag[sel], t0[sel], t1[sel], seq[sel], cu[sel], clk[sel]
    # per CU: the gap between one workgroup's end and the next one's begin
    gaps = []
    for c in np.unique(cu):
        m = np.argsort(t0[cu == c])
        a, b = t0[cu == c][m], t1[cu == c][m]
        gaps.append((a[1:] - b[:-1]) / 100.0)
    gaps = np.concatenate(gaps)
    span = (t1.max() - t0.min()) / 100.0      # microseconds (100 MHz)
    held = (t1 - t0).sum() / 100.0
    # sweep: busy-CU histogram
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    busy = np.cumsum(ev[:, 1])[:-1]
    dt = np.diff(ev[:, 0]) / 100.0
    bins = [(0, 0), (1, 63), (64, 127), (128, 191), (192, 239), (240, 255), (256, 100000)]
    hist = [(lo, hi, float(dt[(busy >= lo) & (busy <= hi)].sum())) for lo, hi in bins]
    print(f"--- batch as {streams} part(s): logged step {span:8.1f} us, CU-time held {held / N_CU:8.1f} us x {N_CU} CUs = {100 * held / (span * N_CU):.1f} % of the chip")
    for lo, hi, t in hist:
        print(f"    busy CUs {lo:3d}..{min(hi, N_CU):3d}: {t:8.1f} us ({100 * t / span:4.1f} %)")
    print(f"    {len(np.unique(cu))} distinct CUs; workgroup-to-workgroup gap on a CU: median {np.median(gaps):.1f} us, mean {gaps.mean():.1f}, "
          f"p10 {np.percentile(gaps, 10):.1f}, p90 {np.percentile(gaps, 90):.1f}; {len(gaps)} hand-overs, {gaps.clip(min=0).sum() / N_CU:.0f} us of chip time")
    kinds = {}
    for k in np.unique(tag):
        m = tag == k
        name = NAMES.get(int(k), (f"gemm384 PF={(int(k) - 0x300) // 16} CFG={(int(k) - 0x300) % 16}" if int(k) >= 0x300 else
                                  f"gemm256 PF={(int(k) - 0x100) // 16} CFG={(int(k) - 0x100) % 16}"))
        kinds[name] = (len(np.unique(seq[m])), float((t1[m] - t0[m]).sum()) / 100.0 / N_CU, float(np.median(clk[m])), float(np.percentile(clk[m], 10)),
                       float(np.percentile(clk[m], 90)))
    for name, (nl, cu_t, c50, c10, c90) in sorted(kinds.items()):
        print(f"    {name:24s} {nl:3d} launches, {cu_t:8.1f} us of chip time, shader clock while running: median {c50:.2f} GHz (p10 {c10:.2f}, p90 {c90:.2f})")
    return span


one = run(1)
two = run(2)
print(f"step: one part {one:.1f} us, two parts {two:.1f} us ({100 * (one / two - 1):+.1f} %)  [diagnostic build: ~10 % slower than the product library]")
