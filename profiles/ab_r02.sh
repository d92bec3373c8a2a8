#!/bin/bash
# A/B of library variants end to end (same process conditions, interleaved rounds): usage  bash profiles/ab_r02.sh base SC1 NSPLIT2
P=language-enhanced-clip-for-multi-label-image-recognition_amd/lib/exp
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    LECLIP_HIP_LIB=$PWD/$P/lib_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-dtype --steps 40 --profile-every 20 > gpurun_out/ab_$v.$round.json 2> gpurun_out/ab_$v.$round.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.$round.json"))
k=d["kernels"]
print("$v round $round: %.0f img/s  %.3f ms  gemm %.1f TF/s (%.1f us)  attn %.1f us" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], k["gemm"]["avg_us"], k["attention"]["avg_us"]))
PY
  done
done
