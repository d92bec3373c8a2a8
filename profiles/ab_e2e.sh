#!/bin/bash
# Runs ON THE GPU BOX: end-to-end bench against two builds of the library (lib_old/ vs lib/), interleaved, same process settings.
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/language-enhanced-clip-for-multi-label-image-recognition_amd
cd $R
for rep in 1 2 3; do
for d in lib_old lib; do
  LECLIP_HIP_LIB=$P/$d/libleclip_hip.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --profile-every 0 > gpurun_out/ab_e2e.log 2>&1 || exit 1
  python - "$d" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_e2e.log") if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"]), "img/s", round(d["ms_per_step"], 3), "ms")
PY
done
done
