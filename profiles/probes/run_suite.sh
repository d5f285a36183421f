cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
echo NO_PIPE_SPLIT > gpurun_out/r3b/arena.log
POET_NO_PIPE_SPLIT=1 timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k test_arena_paths_match_plain 2>&1 | grep "bf16 gradients vs\|largest per\|passed\|failed\|Error" >> gpurun_out/r3b/arena.log
cat gpurun_out/r3b/arena.log
