#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
cp poet_amd/csrc/libpoet_hip.so /tmp/keep.so
for v in keep dwNO_MFMA dwNO_BOTH; do
  if [ $v = keep ]; then cp /tmp/keep.so poet_amd/csrc/libpoet_hip.so; else cp scratch/libpoet_$v.bin poet_amd/csrc/libpoet_hip.so; fi
  rm -rf /tmp/pv; rocprofv3 --kernel-trace -d /tmp/pv -o o --output-format csv -- python profiles/probes/dw_bench.py > /tmp/logv 2>&1 || tail -3 /tmp/logv
  echo "== $v"; python profiles/probes/dw_trace.py /tmp/pv
done
cp /tmp/keep.so poet_amd/csrc/libpoet_hip.so
