#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  bash profiles/collect_issue.sh r03
# Instruction-issue counters of the GEMM / attention kernels (their own rocprofv3 --pmc passes, 8 SQ counters each, --kernel-trace only):
# how many instructions of each class a dispatch issues and for how many cycles each class is busy.  `python profiles/summarize.py <tag>`
# folds them into <tag>_pmc_summary.json ("issue" blocks).
set -o pipefail
TAG=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype --streams 1"
VITL="python3 $R/bench.py --arch ViT-L/14@336px --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-second-dtype --streams 1"
P1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
P2="SQ_INSTS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/pmc_issue1 -- $CMD > $OUT/pmc_issue1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/pmc_issue2 -- $CMD > $OUT/pmc_issue2.log 2>&1 || exit 1
echo vitb issue done
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/pmc_issue1_vitl -- $VITL > $OUT/pmc_issue1_vitl.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/pmc_issue2_vitl -- $VITL > $OUT/pmc_issue2_vitl.log 2>&1 || exit 1
echo vitl issue done
