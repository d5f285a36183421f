#!/bin/bash
# does bucket k's RCCL all-reduce (comm stream) run CONCURRENTLY with the backward segment of bucket k+1 (compute
# stream)?  1-rank `nccl` group with POET_FORCE_COLLECTIVES=1 (the whole N-rank machinery: segmented backward graphs, one
# all-reduce per bucket on the comm stream), kernel trace only; profiles/probes/overlap_analyze.py measures the overlap.
# Run on the GPU box from the repo root:  bash profiles/probes/overlap_trace.sh r4
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/overlap_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export POET_FORCE_COLLECTIVES=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
  $R/bench.py --gpus 1 --steps 4 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-300
python $R/profiles/probes/overlap_analyze.py $OUT | tee $OUT/overlap.txt
