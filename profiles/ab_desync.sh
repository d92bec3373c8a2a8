#!/bin/bash
# A/B: start-up staggering of the 256x256 GEMM's workgroups (diagnostic build `make diag`; LECLIP_GEMM_DESYNC = groups * 256 + step: workgroup b starts
# ((b >> 3) % groups) * step * 512 cycles late).  Why: every workgroup of a launch reaches its epilogue at the same time, so the residual reads and output
# stores of a whole tile round (64 MB for out-proj / c_proj) hit the memory system as one burst while the matrix cores idle (r04_epilogue_experiments.txt).
cd "$(dirname "$0")/../language-enhanced-clip-for-multi-label-image-recognition_amd/lib"
for d in 0 522 532 552 1029 1034 1044 2053 0; do
  g=$((d / 256)); s=$((d % 256))
  echo "== LECLIP_GEMM_DESYNC=$d groups $g step $s (x512 cycles)"
  LECLIP_BENCH_DT=f16 LECLIP_GEMM_DESYNC=$d timeout -k 10 120 ./leclip_kernel_check_diag bench 2>&1 | grep "bench gemm"
done
