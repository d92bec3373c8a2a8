#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  bash profiles/collect.sh r04 main   then   bash profiles/collect.sh r04 extra   (one call each: a call is limited to 20 minutes)
# kernel-trace stats + separate PMC passes (never combined with other trace domains) for the benchmark command, the
# ViT-L/14@336 large-model point (BASELINE configs[4], one GPU's share B=128) and the prompt-tuning step (configs[2]).
# Raw output under gpurun_out/prof_<tag>/ ; `python profiles/summarize.py <tag>` turns it into the committed files.
set -o pipefail
TAG=${1:-r05}
PART=${2:-all}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
if [ "$PART" = all ] || [ "$PART" = main ]; then
# --streams 1: each kernel has the chip to itself, as on the sampled steps bench.py takes its roofline block from (in the default
# two-part mode a dispatch's begin..end spans the time it shares the CUs with the other stream's kernel); the two-part trace
# follows as trace2 for the record.
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype --streams 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || exit 1
CMD2="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- $CMD2 > $OUT/trace2.log 2>&1 || exit 1
echo trace done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
echo pmc mem done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $CMD > $OUT/pmc_grbm.log 2>&1 || exit 1
echo pmc sq done
cd $R
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo main collected
if [ "$PART" = main ]; then exit 0; fi
fi
if [ "$PART" = all ] || [ "$PART" = extra ]; then
cd /tmp
# ViT-L/14@336 (BASELINE configs[4], one GPU's share): --streams 1 so that a dispatch's duration is the kernel alone on the chip (the
# round-2 trace was taken in two-part mode, where a small kernel's begin..end includes queueing behind the other part's persistent grid:
# ln_stats_finalize_kernel read 112 us there)
VITL="python3 $R/bench.py --arch ViT-L/14@336px --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-second-dtype --streams 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vitl -- $VITL > $OUT/vitl.log 2>&1 || exit 1
# attention counters (MFMA busy, LDS conflicts, issue stalls): ViT-L stream kernel and ViT-B heads kernel, their own pass
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq_vitl -- $VITL > $OUT/pmc_sq_vitl.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm_vitl -- $VITL > $OUT/pmc_grbm_vitl.log 2>&1 || exit 1
echo vitl done
TUNE="python3 $R/bench.py --mode tune --dtype bf16 --steps 4 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tune -- $TUNE > $OUT/tune.log 2>&1 || exit 1
# the reference's shipped tuning step (TRAIN.MODEL = DenseCLIP): captions as images, global + local head, three prompt sets, EMA loss
DENSE="python3 $R/bench.py --mode tune --tune-model DenseCLIP --dtype fp16 --steps 4 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tune_dense -- $DENSE > $OUT/tune_dense.log 2>&1 || exit 1
# round 5: counter passes of the tuning step (HBM bytes per GEMM launch for its roofline block) and the multi-crop inference path (N2)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_tune -- $TUNE > $OUT/pmc_fetch_tune.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write_tune -- $TUNE > $OUT/pmc_write_tune.log 2>&1 || exit 1
MC="python3 $R/bench.py --mode multicrop --steps 5 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/multicrop -- $MC > $OUT/multicrop.log 2>&1 || exit 1
cd $R
timeout -k 10 300 python3 bench.py --mode multicrop > $OUT/multicrop_bench.json 2> $OUT/multicrop_bench.err || exit 1
timeout -k 10 300 python3 bench.py --mode tune --tune-model DenseCLIP --dtype fp16 --steps 10 --warmup 3 > $OUT/tune_dense_bench.json 2> $OUT/tune_dense_bench.err || exit 1
timeout -k 10 300 python3 bench.py --arch ViT-L/14@336px --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-second-dtype > $OUT/vitl_bench.json 2> $OUT/vitl_bench.err || exit 1
timeout -k 10 300 python3 bench.py --mode tune --dtype bf16 --steps 10 --warmup 3 > $OUT/tune_bench.json 2> $OUT/tune_bench.err || exit 1
echo extra collected
fi
