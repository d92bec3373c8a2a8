#!/bin/bash
# A/B: start-up staggering of the 256x256 GEMM's workgroups (diagnostic build; LECLIP_GEMM_DESYNC = groups * 256 + step, step in 512-cycle units)
cd "$(dirname "$0")/../language-enhanced-clip-for-multi-label-image-recognition_amd/lib"
for d in 0 562 1049 1036 1074 2060 2054 0; do
  g=$((d / 256)); s=$((d % 256))
  echo "== LECLIP_GEMM_DESYNC=$d groups $g step $s"
  LECLIP_BENCH_DT=f16 LECLIP_GEMM_DESYNC=$d timeout -k 10 120 ./leclip_kernel_check_diag bench 2>&1 | grep "bench gemm"
done
