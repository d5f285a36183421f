#!/bin/bash
# SQ counters of the wide split forward products (gemm_wr / gemm_ws) in isolation:  bash profiles/probes/wr_pmc.sh <tag>
TAG=${1:-wr}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -o p -- python $R/profiles/probes/ws_split_bench.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -o p -- python $R/profiles/probes/ws_split_bench.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("OUT_DIR")
PY
python $R/profiles/probes/wr_pmc_sum.py $OUT
