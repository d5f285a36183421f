#!/bin/bash
# every kernel of the last replayed step in launch order: duration, gap to the previous kernel's end, short name
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
rm -rf /tmp/st; rocprofv3 --kernel-trace -d /tmp/st -o o --output-format csv -- python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS} > /tmp/st.log 2>&1 || tail -5 /tmp/st.log
f=$(find /tmp/st -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
def short(n):
    n = n.replace("poet::", "").replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16")
    n = re.sub(r"at::native::", "", n)
    return n[:90]
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0
tot_gap = 0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3; tot_gap += max(gap, 0)
    print(f"{i:4d} t={(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {gap:6.1f} {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
print("step span us", (prev_end - t0) / 1e3, "sum of gaps", tot_gap)
PY
