#!/bin/bash
# A/B of the engine's walk-order policies (same library): bash profiles/ab_walk.sh [policy ...]
mkdir -p gpurun_out
for round in 1 2 3; do
  for v in ${@:-default c_proj c_fc}; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-dtype --steps 40 --profile-every 20 --walk $v > gpurun_out/ab_walk_$v.$round.json 2> gpurun_out/ab_walk_$v.$round.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_walk_$v.$round.json"))
k=d["kernels"]
print("$v round $round: %.0f img/s  %.3f ms  gemm %.1f TF/s (%.1f us)  attn %.1f us  " % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], k["gemm"]["avg_us"], k["attention"]["avg_us"]) + str({kk.split()[1]+kk.split()[2]: round(x["avg_us"],1) for kk,x in d["gemm_shapes"].items()}))
PY
  done
done
