#!/bin/bash
# Runs ON THE GPU BOX: start-up stagger of the persistent GEMM workgroups (groups*256 + step, step ~ 512 cycles).
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/language-enhanced-clip-for-multi-label-image-recognition_amd/lib/leclip_kernel_check
export LECLIP_BENCH_QUICK=1
for d in 0 1026 1028 1032 1040 2050 2052 2056 4098 4100 0; do
  echo "== desync $d"
  LECLIP_GEMM_DESYNC=$d timeout -k 10 120 $K bench 2>&1 | grep "bench gemm" || exit 1
done
