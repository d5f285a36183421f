#!/bin/bash
cd /root/repo
cp poet_amd/csrc/libpoet_hip.so /tmp/new.so
for cfg in ${CFGS:-ycbv lmo hires}; do
  for skip in ${SKIPS:-0}; do
  for which in ${LIBS:-old new old new}; do
    if [ $which = new ]; then cp /tmp/new.so poet_amd/csrc/libpoet_hip.so; else cp bin_probe/libpoet_$which.bin poet_amd/csrc/libpoet_hip.so; fi
    echo -n "$cfg skip=$skip $which: "; POET_DV_SKIP=$skip python profiles/probes/dv_tiles_bench.py run $cfg | tail -n 1
  done; done
done
cp /tmp/new.so poet_amd/csrc/libpoet_hip.so
