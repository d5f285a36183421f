# Final-build evidence of a round in one GPU call: bash profiles/collect_final.sh  (kernel stats + counters of all three configs through
# collect.sh / summarize.py, then the bench lines profiles/README.md lists; raw output under gpurun_out/final/)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
for c in ycbv lmo hires; do
  bash $R/profiles/collect.sh r6 $c > $R/gpurun_out/final/collect_$c.log 2>&1
  cd $R && python profiles/summarize.py r6 $c >> $R/gpurun_out/final/collect_$c.log 2>&1
done
cd $R
python bench.py --steps 40 --warmup 10 > gpurun_out/final/round6_ycbv_bench.json 2> gpurun_out/final/ycbv.err
python bench.py --config lmo --steps 30 --warmup 8 > gpurun_out/final/round6_lmo_bench.json 2> gpurun_out/final/lmo.err
python bench.py --config hires --steps 20 --warmup 5 > gpurun_out/final/round6_hires_bench.json 2> gpurun_out/final/hires.err
python bench.py --config hires --batch 1 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/final/round6_hires_b1_bench.json 2> gpurun_out/final/hires_b1.err
python bench.py --infer --no-cpu-baseline > gpurun_out/final/round6_infer_bench.json 2> gpurun_out/final/infer.err
python bench.py --infer --batch 16 --no-cpu-baseline >> gpurun_out/final/round6_infer_bench.json 2>> gpurun_out/final/infer.err
POET_BENCH_SELF_LAUNCH=1 POET_FORCE_COLLECTIVES=1 python bench.py --gpus 1 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/final/round6_ycbv_rccl1_bench.json 2> gpurun_out/final/rccl1.err
POET_BENCH_SELF_LAUNCH=1 POET_FORCE_COLLECTIVES=1 python bench.py --config hires --batch 1 --gpus 1 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/final/round6_hires_b1_rccl1_bench.json 2> gpurun_out/final/rccl1b.err
ls -la profiles/round6_* gpurun_out/final/ | head -50
cp profiles/round6_*_kernel_stats.csv profiles/round6_*_pmc_*.csv gpurun_out/final/ 2>/dev/null
grep -o "ms_per_step\": [0-9.]*" gpurun_out/final/*.json | awk 'NR%8==1'
