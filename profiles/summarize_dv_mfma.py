#!/usr/bin/env python3
"""gpurun_out/dvmfma_<tag>/ (profiles/collect_dv_mfma.sh) -> profiles/round<N>_ycbv_pmc_dv_mfma.csv: per scatter kernel the average
counter values per launch (the probe launches both kernels on the same inputs)."""
import csv, os, re, sys
from collections import defaultdict
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", f"dvmfma_{tag}")
def short(name):
    name = name.replace("unsigned short", "bf16").replace("void ", "").replace("poet::", "")
    return re.sub(r"\(.*$", "", name)
table = defaultdict(dict)
for root, _, files in os.walk(src):
    for f in files:
        if f.endswith("counter_collection.csv"):
            acc, n = defaultdict(float), defaultdict(int)
            for r in csv.DictReader(open(os.path.join(root, f))):
                k = short(r["Kernel_Name"])
                if "msda_bwd_dv" not in k:
                    continue
                acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
            for (k, c), v in acc.items():
                table[k][c] = v / n[(k, c)]
        elif f.endswith("kernel_stats.csv"):
            for r in csv.DictReader(open(os.path.join(root, f))):
                k = short(r["Name"])
                if "msda_bwd_dv" in k:
                    table[k]["avg_us"] = float(r["AverageNs"]) / 1e3
                    table[k]["calls"] = float(r["Calls"])
cols = sorted({c for v in table.values() for c in v})
out = os.path.join(here, f"round{tag.lstrip('r')}_ycbv_pmc_dv_mfma.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel"] + cols)
    for k, row in sorted(table.items()):
        w.writerow([k] + [("%.4g" % row[c]) if c in row else "" for c in cols])
print(open(out).read())
