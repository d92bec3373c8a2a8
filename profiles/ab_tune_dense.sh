#!/bin/bash
# Round 5: the shipped DenseCLIP caption-as-image tuning step with its three text-tower passes side by side (product) against one after the other
# (bench.py --mode tune --tune-model DenseCLIP --no-text-beside), interleaved rounds on one box.
set -o pipefail
mkdir -p gpurun_out
for r in 1 2 3; do for v in behind beside; do
  f=""; [ $v = behind ] && f="--no-text-beside"
  timeout -k 10 300 python bench.py --mode tune --tune-model DenseCLIP --dtype fp16 --steps 12 --warmup 3 --no-cpu-baseline $f > gpurun_out/ab_dense_$v.$r.json 2> gpurun_out/ab_dense_$v.$r.err || { tail -5 gpurun_out/ab_dense_$v.$r.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_dense_$v.$r.json"))
print("$v round $r: %.0f captions/s  %.3f ms  loss %.6f" % (d["value"], d["ms_per_step"], d["last_loss"]))
PY
done; done
