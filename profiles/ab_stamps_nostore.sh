#!/bin/bash
# per-tile timeline (diagnostic build) with and without the epilogue's global stores: where does the store cost land?
cd "$(dirname "$0")/../language-enhanced-clip-for-multi-label-image-recognition_amd/lib"
for d in 0 2; do
  echo "== LECLIP_GEMM_DEBUG=$d"
  LECLIP_GEMM_DEBUG=$d timeout -k 10 200 ./leclip_kernel_check_diag stamps 2>&1 | grep -v "^    kt"
done
