#!/bin/bash
# Runs ON THE GPU BOX.  Is the 256x256 GEMM epilogue bound by one CU's store path or by the chip-wide write rate?
# Same GEMM shapes with the persistent grid capped at 256 / 64 / 32 workgroups, shipped kernel vs epilogue skipped.
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$R/language-enhanced-clip-for-multi-label-image-recognition_amd/lib/leclip_kernel_check_diag   # the diagnostic build (make diag): the product library reads no LECLIP_GEMM_* switch
export LECLIP_BENCH_QUICK=1
for grid in 256 64 32; do
  for dbg in 0 2 1; do
    echo "== grid $grid debug $dbg"
    LECLIP_GEMM_GRID=$grid LECLIP_GEMM_DEBUG=$dbg timeout -k 10 120 $K bench 2>&1 | grep "bench gemm" || exit 1
  done
done
