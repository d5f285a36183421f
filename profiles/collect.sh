#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect.sh r2 [ycbv|lmo|hires]
# Kernel-trace stats and PMC counters are collected in SEPARATE rocprofv3 runs (guide: never combine --pmc with the
# sys/runtime/hip trace domains).  Raw output goes to gpurun_out/ (scratch); profiles/summarize.py turns it into the
# committed summaries profiles/round<N>_<config>_*.csv.
set -x
TAG=${1:-r2}
CFG=${2:-ycbv}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -o p -- $CMD > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -30
