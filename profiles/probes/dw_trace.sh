#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for b in ${BLOCKS:-256 512}; do
  POET_DW_BLOCKS=$b rocprofv3 --kernel-trace -d /tmp/p$b -o o --output-format csv -- python profiles/probes/dw_bench.py > /tmp/log$b 2>&1 || tail -5 /tmp/log$b
  echo "== blocks $b"; python profiles/probes/dw_trace.py /tmp/p$b
done
