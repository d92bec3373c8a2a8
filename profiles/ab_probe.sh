#!/bin/bash
# Runs ON THE GPU BOX: A/B of an env-switchable GEMM behaviour with the kernel-check bench.  usage: ab_probe.sh VAR val1 val2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/language-enhanced-clip-for-multi-label-image-recognition_amd/lib/leclip_kernel_check
export LECLIP_BENCH_QUICK=1
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 120 $K bench 2>&1 | grep "bench gemm" || exit 1
done
done
