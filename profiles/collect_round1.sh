#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect_round1.sh
# Kernel-trace stats and PMC counters are collected in SEPARATE rocprofv3 runs (guide: never combine --pmc with the
# sys/runtime/hip trace domains).  Raw output goes to gpurun_out/ (scratch); the summaries are copied by hand into profiles/.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r1 -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r1 -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r1 -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o r1 -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
