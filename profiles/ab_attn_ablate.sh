#!/bin/bash
# timing-only runs of attention ablation builds (results are garbage by construction): usage bash profiles/ab_attn_ablate.sh L s3 nok nov ...
L=language-enhanced-clip-for-multi-label-image-recognition_amd/lib/exp
shapes=$1; shift
for v in "$@"; do
  echo "== $v"
  LECLIP_ATTN_BENCH_ONLY=1 LECLIP_ATTN_SHAPES=$shapes timeout -k 10 300 $L/kernel_check_$v attn 2>&1 | grep "bench attn" || exit 1
done
