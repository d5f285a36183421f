#!/bin/bash
# step-level A/B of two library builds on one box: LIBS="head new" (scratch/libpoet_<name>.bin; "new" = the tree's build)
cd /root/repo
cp poet_amd/csrc/libpoet_hip.so /tmp/new.so
for rep in 1 2; do
for which in ${LIBS:-head new}; do
  if [ $which = new ]; then cp /tmp/new.so poet_amd/csrc/libpoet_hip.so; else cp scratch/libpoet_$which.bin poet_amd/csrc/libpoet_hip.so; fi
  echo -n "$which: "; python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
cp /tmp/new.so poet_amd/csrc/libpoet_hip.so
