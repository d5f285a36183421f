#!/bin/bash
# SQ counters of the value-gradient scatter, LDS-tiled DPP kernel (default) against the matrix-core form
# (POET_DV_MFMA=1), YCB-V geometry, on the same inputs (profiles/probes/dv_mfma_ab.py with NOISE=0: the benchmark's sampling
# pattern).  Run on the GPU box from the repo root: bash profiles/collect_dv_mfma.sh r4.  Counter passes only (--pmc).
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dvmfma_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
CMD="python $R/profiles/probes/dv_mfma_ab.py ycbv"
export NOISE=0 AB_QUICK=1
pass() {
  local name=$1; shift
  local have=""
  for c in "$@"; do if grep -qw "$c" $OUT/avail.txt; then have="$have $c"; else echo "missing counter $c" >> $OUT/missing.txt; fi; done
  [ -z "$have" ] && return
  timeout 600 rocprofv3 --pmc $have --output-format csv -d $OUT/${name} -o p -- $CMD > $OUT/${name}.log 2>&1
}
pass sq   SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2  SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- $CMD > $OUT/trace.log 2>&1
python $R/profiles/summarize_dv_mfma.py $TAG
