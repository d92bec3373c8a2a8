#!/bin/bash
# Runs ON THE GPU BOX: same kernel_check bench against two builds of the library (lib_old/ vs lib/), interleaved.
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/language-enhanced-clip-for-multi-label-image-recognition_amd
export LECLIP_BENCH_QUICK=1
for rep in 1 2; do
for d in lib_old lib; do
  echo "== $d $*"
  env "$@" timeout -k 10 120 $P/$d/leclip_kernel_check bench 2>&1 | grep "bench gemm" || exit 1
done
done
