#!/bin/bash
# A/B of library variants on the two tuning steps: usage  bash profiles/ab_tune.sh t192 t128 ...   (lib/exp/lib_<name>.so)
P=language-enhanced-clip-for-multi-label-image-recognition_amd/lib/exp
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    LECLIP_HIP_LIB=$PWD/$P/lib_$v.so timeout -k 10 200 python bench.py --mode tune --tune-model DenseCLIP --dtype fp16 --steps 10 --warmup 3 > gpurun_out/abt_$v.json 2> gpurun_out/abt_$v.err || exit 1
    LECLIP_HIP_LIB=$PWD/$P/lib_$v.so timeout -k 10 200 python bench.py --mode tune --dtype bf16 --steps 10 --warmup 3 > gpurun_out/abt2_$v.json 2> gpurun_out/abt2_$v.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/abt_$v.json")); e=json.load(open("gpurun_out/abt2_$v.json"))
print("$v round $round: DenseCLIP %.0f captions/s (%.2f ms)   cfg3 %.0f img/s (%.2f ms)" % (d["value"], d["ms_per_step"], e["value"], e["ms_per_step"]))
PY
  done
done
