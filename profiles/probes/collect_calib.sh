#!/bin/bash
# FETCH_SIZE calibration (profiles/probes/fetch_calib.hip) -> gpurun_out/calib_<tag>/fetch.csv; summarised by hand into DESIGN section 6.
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/calib_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/profiles/probes/_bin/fetch_calib > $OUT/bytes.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $R/profiles/probes/_bin/fetch_calib > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $R/profiles/probes/_bin/fetch_calib > $OUT/write.log 2>&1
python - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for kind in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(kind, k[0][:40], k[1], "per launch:", [round(x) for x in v])
print(open(os.path.join(out, "bytes.txt")).read())
PY
