#!/usr/bin/env python3
"""gpurun_out/gather_<tag>/ (profiles/collect_gather.sh) -> profiles/round<N>_ycbv_pmc_gather.csv: per MSDA kernel and gather
variant (l1 = default L1-served gathers, win = LDS value windows, POET_WIN_GATHER=1) the average counter values per launch."""
import csv, os, re, sys
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", f"gather_{tag}")
def short(name):
    name = name.replace("unsigned short", "bf16").replace("void ", "").replace("poet::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)
table = defaultdict(dict)           # (kernel, variant) -> counter -> avg per launch
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d)
    if not os.path.isdir(p):
        continue
    variant = d.rsplit("_", 1)[1]
    for root, _, files in os.walk(p):
        for f in files:
            if f.endswith("counter_collection.csv"):
                acc, n = defaultdict(float), defaultdict(int)
                for r in csv.DictReader(open(os.path.join(root, f))):
                    k = short(r["Kernel_Name"])
                    if not k.startswith("msda"):
                        continue
                    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
                for (k, c), v in acc.items():
                    table[(k, variant)][c] = v / n[(k, c)]
            elif f.endswith("kernel_stats.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    k = short(r["Name"])
                    if k.startswith("msda"):
                        table[(k, variant)]["avg_us"] = float(r["AverageNs"]) / 1e3
                        table[(k, variant)]["calls"] = float(r["Calls"])
cols = sorted({c for v in table.values() for c in v})
out = os.path.join(here, f"round{tag.lstrip('r')}_ycbv_pmc_gather.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "variant"] + cols)
    for (k, v), row in sorted(table.items()):
        w.writerow([k, v] + [("%.4g" % row[c]) if c in row else "" for c in cols])
print(open(out).read())
