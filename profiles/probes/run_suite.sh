cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
for c in lmo hires; do
  timeout 900 python bench.py --config $c > gpurun_out/r3e/bench_$c.json 2> gpurun_out/r3e/bench_$c.err
  bash profiles/collect.sh r3 $c > /dev/null 2>&1
done
timeout 600 python bench.py > gpurun_out/r3e/bench_ycbv.json 2> gpurun_out/r3e/bench_ycbv.err
timeout 600 python bench.py --infer --batch 1 > gpurun_out/r3e/bench_infer1.json 2>&1
timeout 600 python bench.py --infer --batch 16 > gpurun_out/r3e/bench_infer16.json 2>&1
du -sh gpurun_out/prof_r3_*; python - <<'PY'
import json
for c in ("ycbv","lmo","hires"):
    d=json.loads(open(f"gpurun_out/r3e/bench_{c}.json").read().strip().splitlines()[-1]); print(c, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
tail -2 gpurun_out/r3e/bench_infer1.json gpurun_out/r3e/bench_infer16.json | cut -c1-300
