#!/usr/bin/env python3
"""Kernel-trace CSV (rocprofv3 --kernel-trace) -> for every RCCL kernel (ncclDevKernel*): its interval, the queue it ran on,
and how much of it overlaps kernels of OTHER queues (the compute stream's backward segments).  Prints one line per collective of
the last traced step and the totals."""
import csv, glob, os, sys
src = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
if not rows:
    sys.exit("no kernel trace found under " + src)
nccl = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
other = [r for r in rows if not ("nccl" in r[2].lower() or "rccl" in r[2].lower())]
print(f"{len(rows)} kernels, {len(nccl)} RCCL kernels; queues: {sorted({r[3] for r in rows})}, RCCL queues: {sorted({r[3] for r in nccl})}")
import bisect
starts = [r[0] for r in other]
tot = ov = 0
per = []
for s, e, name, q, st in nccl:
    i = bisect.bisect_left(starts, s) - 1
    o = 0
    names = set()
    j = max(i, 0)
    while j < len(other) and other[j][0] < e:
        a, b = max(other[j][0], s), min(other[j][1], e)
        if b > a and other[j][3] != q:
            o += b - a
            names.add(other[j][2].split("(")[0][:40])
        j += 1
    tot += e - s; ov += min(o, e - s)
    per.append((s, e - s, min(o, e - s), q, sorted(names)[:2]))
for s, d, o, q, names in per[-16:]:
    print(f"  RCCL kernel on queue {q}: {d / 1e3:8.1f} us, {o / 1e3:8.1f} us of it concurrent with compute-queue kernels {names}")
print(f"RCCL kernel time {tot / 1e3:.1f} us, of which {ov / 1e3:.1f} us ({100.0 * ov / max(tot, 1):.1f} %) ran while a kernel of another queue was executing")
# the reverse view: gaps on the compute queue while RCCL runs alone
