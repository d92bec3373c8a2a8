#!/bin/bash
# A/B of bench.py flag sets end to end (interleaved rounds, one box): usage  bash profiles/ab_flags.sh "name1:flags1" "name2:flags2" ...
mkdir -p gpurun_out
for round in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; f=${spec#*:}
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-dtype --steps 40 --profile-every 20 $f > gpurun_out/abf_$v.$round.json 2> gpurun_out/abf_$v.$round.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/abf_$v.$round.json"))
k=d["kernels"]
print("$v round $round: %.0f img/s  %.3f ms  gemm %.1f TF/s (%.1f us)  attn %.1f us" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], k["gemm"]["avg_us"], k["attention"]["avg_us"]))
PY
  done
done
