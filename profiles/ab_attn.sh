#!/bin/bash
# A/B of attention kernel variants with the standalone harness (kernel_check attn: correctness on the many-heads grids, then timings).
# usage: bash profiles/ab_attn.sh aplain anew ...   (variants built with `make -C csrc variant NAME=.. VSRC=attention DEFS=..`)
L=language-enhanced-clip-for-multi-label-image-recognition_amd/lib/exp
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  timeout -k 10 400 $L/kernel_check_$v attn > gpurun_out/.ab_attn_$v.log 2>&1; rc=$?; grep -v "^ok " gpurun_out/.ab_attn_$v.log; [ $rc -eq 0 ] || { echo "variant $v failed (rc $rc): stopping"; exit 1; }
done
