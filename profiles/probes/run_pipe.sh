cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 300 python profiles/probes/pipe_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/pipe_prof.log
