#!/bin/bash
# Round 5: the many-heads attention kernel without the dead 8-key groups of its last key tile (adead = product) against the same library with
# -DLECLIP_ATTN_NO_DEAD_GROUPS (afull): harness timings, then the step end to end, interleaved.
set -o pipefail
L=language-enhanced-clip-for-multi-label-image-recognition_amd/lib/exp
mkdir -p gpurun_out
for r in 1 2; do for v in ${VARIANTS:-afull adead}; do
  echo "== $v round $r"; timeout -k 10 300 $L/kernel_check_$v attn > gpurun_out/.ab_attn_$v.log 2>&1 || { tail -5 gpurun_out/.ab_attn_$v.log; exit 1; }
  grep "bench attn ViT-B" gpurun_out/.ab_attn_$v.log
done; done
bash profiles/ab_r02.sh ${VARIANTS:-afull adead}
