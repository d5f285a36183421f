#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (profiles/probes/fetch_calib.hip) -> gpurun_out/calib_<tag>/fetch_calib.csv, committed as
# profiles/round<N>_fetch_calib.csv: per access pattern the KNOWN bytes read per launch, the raw counter (KiB) and their ratio.
# Run on the GPU box from the repo root: bash profiles/probes/collect_calib.sh r5
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/calib_${TAG}
mkdir -p $OUT $R/profiles/probes/_bin
hipcc --offload-arch=gfx950 -O2 $R/profiles/probes/fetch_calib.hip -o $R/profiles/probes/_bin/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
$R/profiles/probes/_bin/fetch_calib > $OUT/bytes.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $R/profiles/probes/_bin/fetch_calib > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $R/profiles/probes/_bin/fetch_calib > $OUT/write.log 2>&1
python - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
# launch order inside one repetition (fetch_calib.hip main()): the two segread<uint16_t> launches share a symbol, so patterns are
# told apart by dispatch order
pats = ["stream16", "seg32_s512", "seg64_s1536", "seg32_s1536", "lane16_s1536"]
known = dict(zip(pats, [int(x) for x in open(os.path.join(out, "bytes.txt")).read().split() if x.isdigit()]))
rows = {}
for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    recs = []
    for f in glob.glob(os.path.join(out, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname and any(k in r["Kernel_Name"] for k in ("stream16", "segread", "lane16")):
                recs.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    recs.sort()
    for i, (_, name, val) in enumerate(recs):
        rows.setdefault((pats[i % 5], name), {}).setdefault(kind, []).append(val)
with open(os.path.join(out, "fetch_calib.csv"), "w") as fh:
    fh.write("# FETCH_SIZE / WRITE_SIZE of rocprofv3 on gfx950 against KNOWN bytes (profiles/probes/fetch_calib.hip: every kernel reads its bytes exactly once\n"
             "# from a 1.5 GiB buffer = 6x the Infinity Cache; 3 repetitions, separate --pmc passes).  ratio = FETCH_SIZE KiB x 1024 / known bytes:\n"
             "# 1.0 = the counter is exact for the pattern, 0.5 = it reports half (the guide's gfx950 correction: multiply by 2)\n")
    fh.write("pattern,kernel,known_bytes_per_launch,FETCH_SIZE_KiB_mean,FETCH_SIZE_KiB_min,FETCH_SIZE_KiB_max,ratio_fetch_over_known,WRITE_SIZE_KiB_mean\n")
    for p in pats:
        for (pp, name), v in rows.items():
            if pp != p:
                continue
            f, w = v.get("fetch", [0.0]), v.get("write", [0.0])
            fm = sum(f) / len(f)
            fh.write(f"{p},{name},{known[p]},{fm:.1f},{min(f):.1f},{max(f):.1f},{fm * 1024 / known[p]:.4f},{sum(w) / len(w):.1f}\n")
print(open(os.path.join(out, "fetch_calib.csv")).read())
PY
