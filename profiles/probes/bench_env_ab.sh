#!/bin/bash
# step-level A/B of an environment switch on one box: ENVVAR=NAME (A: NAME=1, B: unset)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  echo -n "$ENVVAR=1: "; env $ENVVAR=1 python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  echo -n "default: "; python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
