cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
for v in "0 0" "1 0" "0 32" "1 32"; do
  set -- $v
  echo "== cfg $1 dbg $2"
  POET_PIPE_CFG=$1 POET_PIPE_DBG=$2 timeout 300 python profiles/probes/pipe_probe.py quick 2>&1 | grep "M=\|rror"
done > gpurun_out/r3a/pipe_cfg.log 2>&1
cat gpurun_out/r3a/pipe_cfg.log
