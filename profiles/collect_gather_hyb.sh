#!/bin/bash
# Round 4: cache / LDS / SQ counters of the forward gather, L1-served default (l1) against the hybrid kernel with level 3 staged in LDS
# (hyb3: same occupancy, a quarter of the L1 look-ups gone) and levels 2-3 staged (hyb2: half of them gone, half the occupancy).
# Run on the GPU box from the repo root: bash profiles/collect_gather_hyb.sh r4   (-> profiles/round4_ycbv_pmc_gather.csv)
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gather_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
setv() { unset POET_MSDA_HYBRID POET_HYB_LC; case $1 in hyb3) export POET_MSDA_HYBRID=1 POET_HYB_LC=3;; hyb2) export POET_MSDA_HYBRID=1 POET_HYB_LC=2;; esac; }
for v in l1 hyb3 hyb2; do
  setv $v
  timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp_$v -o p -- $CMD > $OUT/tcp_$v.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq_$v -o p -- $CMD > $OUT/sq_$v.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o p -- $CMD > $OUT/trace_$v.log 2>&1
done
unset POET_MSDA_HYBRID POET_HYB_LC
python $R/profiles/summarize_gather.py $TAG
