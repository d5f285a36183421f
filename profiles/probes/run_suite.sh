cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "site-packages\|dist-packages" | tail -15 | cut -c1-700 > gpurun_out/r3d/pytest.log
cat gpurun_out/r3d/pytest.log
