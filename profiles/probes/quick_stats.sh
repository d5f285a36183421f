#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats -d /tmp/qs -o o --output-format csv -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/qs.log 2>&1 || tail -5 /tmp/qs.log
f=$(find /tmp/qs -name "*kernel_stats.csv" | head -1)
python - "$f" $FILTER <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in [r for r in rows if (len(sys.argv) < 3 or sys.argv[2] in r["Name"])][int(__import__("os").environ.get("SKIP","0")):int(__import__("os").environ.get("SKIP","0"))+int(__import__("os").environ.get("ROWS","14"))]:
    n = r["Name"]; n = n[:100]
    print('%5d x %8.1f us (min %.1f max %.1f)  tot %7.2f ms  %s' % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['TotalDurationNs'])/1e6, n))
PY
