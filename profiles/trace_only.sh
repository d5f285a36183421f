#!/bin/bash
# kernel-trace only (quick): bash profiles/trace_only.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
