#!/bin/bash
# A/B of stream-part splits of the B=256 batch (same library): bash profiles/ab_split.sh "128,128" "110,146" ...
# Why uneven: a part of 110 images is 85 tile rows of 256 -> 255 / 765 / 1020 tiles for N = 768 / 2304 / 3072: full rounds on 256 CUs for every block GEMM.
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    tag=${v//,/_}
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-dtype --steps 40 --profile-every 0 --split $v > gpurun_out/ab_split_$tag.$round.json 2> gpurun_out/ab_split_$tag.$round.err || { echo "$v failed"; tail -3 gpurun_out/ab_split_$tag.$round.err; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_split_$tag.$round.json"))
print("split $v round $round: %.0f img/s  %.3f ms" % (d["value"], d["ms_per_step"]))
PY
  done
done
