#!/bin/bash
# which kernels sit around the small D2D copies / fills of a replayed step?  (kernel trace of the last steps, by start time)
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
rm -rf /tmp/ct; rocprofv3 --kernel-trace -d /tmp/ct -o o --output-format csv -- python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/ct.log 2>&1 || tail -5 /tmp/ct.log
f=$(find /tmp/ct -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last replayed step: find the last adamw kernel and the one before it
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
step = rows[lo:hi]
print("kernels in last step:", len(step), "span ms", (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6)
def short(n):
    n = n.replace("poet::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]
cnt = collections.Counter()
for i, r in enumerate(step):
    n = r["Kernel_Name"]
    if "copyBuffer" in n or "FillFunctor" in n or "elementwise" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = short(step[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(step[i + 1]["Kernel_Name"]) if i + 1 < len(step) else "-"
        g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        print(f"{i:4d} {d:7.1f} us grid {g:>9} {short(n)[:40]:40s} | prev {prev[:45]:45s} | next {nxt[:45]}")
        cnt[short(n)[:40]] += 1
print(cnt)
PY
