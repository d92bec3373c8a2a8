#!/bin/bash
# End-to-end sweep of the ViT-B attention kernel's workgroup count (variant library built with -DLECLIP_ATTN_WG_CAP, VSRC=attention NAME=cap): does a
# kernel that is close to the HBM rate need all 256 CUs, or can it leave some to the other stream part's GEMM?
P=language-enhanced-clip-for-multi-label-image-recognition_amd/lib/exp
mkdir -p gpurun_out
for round in 1 2; do
  for w in 256 224 192 160 128; do
    LECLIP_XATTN_WGS=$w LECLIP_HIP_LIB=$PWD/$P/lib_cap.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-dtype --steps 40 --profile-every 20 > gpurun_out/ab_cap.json 2> gpurun_out/ab_cap.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_cap.json"))
k=d["kernels"]
print("wgs $w round $round: %.0f img/s  %.3f ms  gemm %.1f us  attn %.1f us" % (d["value"], d["ms_per_step"], k["gemm"]["avg_us"], k["attention"]["avg_us"]))
PY
  done
done
