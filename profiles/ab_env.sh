#!/bin/bash
# Runs ON THE GPU BOX: end-to-end bench under different environment switches, interleaved with the default.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --profile-every 0 > gpurun_out/ab_env.log 2>&1 || { echo "FAILED $*"; return; }
  python - "$*" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_env.log") if l.startswith("{")][-1])
print(f"{sys.argv[1]:40s} {round(d['value'])} img/s {d['ms_per_step']:.3f} ms")
PY
}
for rep in 1 2; do
  run LECLIP_DEFAULT=1
  for v in "$@"; do run $v; done
done
