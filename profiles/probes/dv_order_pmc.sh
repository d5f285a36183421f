#!/bin/bash
# Counter traffic of the value-gradient scatter under the two work orders (VERDICT r4 #6a): POET_DV_ORDER=0 = the (tiles, M, N) grid dealt
# round-robin over the XCDs, 1 (default) = head fastest + contiguous ranges per XCD.  FETCH_SIZE / WRITE_SIZE in separate --pmc passes,
# the scatter kernel only.  Run on the GPU box from the repo root:  bash profiles/probes/dv_order_pmc.sh r5   -> gpurun_out/dv_order_<tag>/summary.csv
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dv_order_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in ycbv lmo hires; do
  for o in 0 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      POET_DV_ORDER=$o rocprofv3 --pmc $c --kernel-include-regex "msda_bwd_dv_tiled" --output-format csv -d $OUT/${cfg}_o${o}_$c -o p -- \
        python $R/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/${cfg}_o${o}_$c.log 2>&1
    done
  done
done
python - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
geo = {"ycbv": (16, 6380), "lmo": (32, 1600), "hires": (8, 25500)}
rows = ["config,order,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,read_bytes,algorithmic_read_bytes,read_over_algorithmic"]
for cfg in ("ycbv", "lmo", "hires"):
    n, s = geo[cfg]
    alg = n * s * 256 * (2 + 6)            # grad_out (bf16) + offsets | logits (fp16): bytes per (row, channel) x rows x 256 channels
    for o in (0, 1):
        v = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = []
            for f in glob.glob(os.path.join(out, f"{cfg}_o{o}_{c}", "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == c and "msda_bwd_dv_tiled" in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            v[c] = (sum(vals) / len(vals), len(vals)) if vals else (float("nan"), 0)
        rd = v["FETCH_SIZE"][0] * 1024
        rows.append(f"{cfg},{o},{v['FETCH_SIZE'][1]},{v['FETCH_SIZE'][0]:.0f},{v['WRITE_SIZE'][0]:.0f},{rd:.0f},{alg},{rd / alg:.3f}")
open(os.path.join(out, "summary.csv"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
