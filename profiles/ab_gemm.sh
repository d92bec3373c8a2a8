#!/bin/bash
# A/B of GEMM library variants with the standalone harness: correctness (kernel_check) then the four ViT-B/16 block shapes,
# interleaved rounds.  usage: bash profiles/ab_gemm.sh fr pp ...   (variants built with `make -C csrc variant NAME=.. DEFS=..`;
# "base" = the product library)
L=language-enhanced-clip-for-multi-label-image-recognition_amd/lib
mkdir -p gpurun_out
for v in "$@"; do
  exe=$L/exp/kernel_check_$v; [ "$v" = base ] && exe=$L/leclip_kernel_check
  if ! timeout -k 10 120 $exe > gpurun_out/check_$v.txt 2>&1; then echo "variant $v: kernel_check FAILED"; tail -5 gpurun_out/check_$v.txt; exit 1; fi
  echo "variant $v: $(tail -1 gpurun_out/check_$v.txt)"
done
for round in 1 2 3; do
  for v in "$@"; do
    exe=$L/exp/kernel_check_$v; [ "$v" = base ] && exe=$L/leclip_kernel_check
    echo "== $v round $round"; LECLIP_BENCH_DT=f16 timeout -k 10 120 $exe bench 2>&1 | grep "bench gemm" || exit 1
  done
done
