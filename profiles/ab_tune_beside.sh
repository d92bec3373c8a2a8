#!/bin/bash
# Round 5: prompt-tuning step (cfg3, B = 512 bf16) with the text tower's forward enqueued beside the image tower's stream parts (product) against behind it
# (bench.py --mode tune --no-text-beside), interleaved rounds on one box.
set -o pipefail
mkdir -p gpurun_out
for r in 1 2 3; do for v in ${VARIANTS:-behind beside}; do
  f=""; [ $v = behind ] && f="--no-text-beside --no-pipeline"; [ $v = beside ] && f="--no-pipeline"   # (pipelined = the product: text tower beside + next batch's image tower beside the backward)
  timeout -k 10 300 python bench.py --mode tune --dtype bf16 --steps 12 --warmup 3 --no-cpu-baseline $f > gpurun_out/ab_tune_$v.$r.json 2> gpurun_out/ab_tune_$v.$r.err || { tail -5 gpurun_out/ab_tune_$v.$r.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_tune_$v.$r.json"))
print("$v round $r: %.0f img/s  %.3f ms  loss %.6f" % (d["value"], d["ms_per_step"], d["last_loss"]))
PY
done; done
