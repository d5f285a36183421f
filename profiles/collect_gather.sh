#!/bin/bash
# cache / texture-addresser / LDS counters of the MSDA gather kernels, L1-served (default) against the
# LDS-window variant (POET_WIN_GATHER=1).  Run on the GPU box from the repo root: bash profiles/collect_gather.sh r3
# Counter passes only (--pmc, no trace domains); names are intersected with what `rocprofv3 -L` offers on this box.
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gather_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
pass() {   # name, counters...
  local name=$1; shift
  local have=""
  for c in "$@"; do if grep -qw "$c" $OUT/avail.txt; then have="$have $c"; else echo "missing counter $c" >> $OUT/missing.txt; fi; done
  [ -z "$have" ] && return
  for v in l1 win; do
    if [ $v = win ]; then export POET_WIN_GATHER=1; else unset POET_WIN_GATHER; fi
    timeout 600 rocprofv3 --pmc $have --output-format csv -d $OUT/${name}_$v -o p -- $CMD > $OUT/${name}_$v.log 2>&1
  done
  unset POET_WIN_GATHER
}
pass tcp  TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass ta   TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
pass sq   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2  SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD
pass tcc  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_l1 -o p -- $CMD > $OUT/trace_l1.log 2>&1
POET_WIN_GATHER=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_win -o p -- $CMD > $OUT/trace_win.log 2>&1
python $R/profiles/summarize_gather.py $TAG
