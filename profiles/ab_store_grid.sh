#!/bin/bash
# does the store cost of the 256x256 GEMM depend on how many CUs store at once?  grid capped to G workgroups (diagnostic build), with / without the stores
cd "$(dirname "$0")/../language-enhanced-clip-for-multi-label-image-recognition_amd/lib"
for g in 0 128 64 32; do
  for d in 0 2; do
    echo "== LECLIP_GEMM_GRID=$g LECLIP_GEMM_DEBUG=$d"
    LECLIP_BENCH_DT=f16 LECLIP_GEMM_GRID=$g LECLIP_GEMM_DEBUG=$d timeout -k 10 200 ./leclip_kernel_check_diag bench 2>&1 | grep "bench gemm"
  done
done
