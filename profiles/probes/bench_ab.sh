#!/bin/bash
# step-level A/B of two library builds on one box: LIBS="head new" (bin_probe/libpoet_<name>.bin (git-ignored, travels to the GPU box); "new" = the tree's build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp poet_amd/csrc/libpoet_hip.so /tmp/new.so
for rep in 1 2; do
for which in ${LIBS:-head new}; do
  if [ $which = new ]; then cp /tmp/new.so poet_amd/csrc/libpoet_hip.so; else cp bin_probe/libpoet_$which.bin poet_amd/csrc/libpoet_hip.so; fi
  echo -n "$which: "; python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
cp /tmp/new.so poet_amd/csrc/libpoet_hip.so
