cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3c/pytest.log
timeout 600 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
POET_DV_VIA_MAP=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3c/bench_via_map.json 2> gpurun_out/r3c/bench_via_map.err
cat gpurun_out/r3c/pytest.log; python - <<'PY'
import json
for f in ("bench", "bench_via_map"):
    try:
        d = json.loads(open(f"gpurun_out/r3c/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], {k: v for k, v in list(d["kernel_breakdown_ms_per_step"].items())[:14]})
        if "cpu_baseline" in d: print(d["cpu_baseline"]); print(d["roofline"]["top3"]); print(d["config"])
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/r3c/{f}.err").read()[-1500:])
PY
