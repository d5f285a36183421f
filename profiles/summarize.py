#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect_round1.sh (gpurun_out/prof_r1/) into the two summaries that are
committed and that bench.py / DESIGN.md cite:
  profiles/round1_kernel_stats.csv  per-kernel calls / total / average duration (kernel-trace pass)
  profiles/round1_pmc_hbm.csv       per-kernel average FETCH_SIZE / WRITE_SIZE per launch (two separate --pmc passes)
Usage: python profiles/summarize.py [gpurun_out/prof_r1]"""
import csv, os, re, shutil, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "prof_r1")
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace("unsigned short", "bf16").replace("void ", "").replace("poet::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


shutil.copy(os.path.join(src, "trace", "r1_kernel_stats.csv"), os.path.join(here, "round1_kernel_stats.csv"))
fetch = per_kernel(os.path.join(src, "pmc_fetch", "r1_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(src, "pmc_write", "r1_counter_collection.csv"), "WRITE_SIZE")
rows = []
for k, (n, f) in fetch.items():
    if not (k.startswith("gemm") or k.startswith("msda") or k.startswith("ln_") or k.startswith("gn_") or k.startswith("colsum")):
        continue
    wn, w = write.get(k, (0, 0.0))
    af, aw = f / n, (w / wn if wn else 0.0)
    rows.append((2 * af + aw, k, n, af, aw))
rows.sort(reverse=True)
with open(os.path.join(here, "round1_pmc_hbm.csv"), "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/collect_round1.sh), same bench command;\n")
    fo.write("# counter unit = KiB at the L2<->fabric boundary, averaged per launch over ALL launches of the kernel symbol\n")
    fo.write("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x -> fetch_corrected = 2 * FETCH_SIZE\n")
    fo.write("kernel,launches,avg_FETCH_SIZE_KiB,avg_fetch_corrected_MB,avg_WRITE_SIZE_KiB,avg_write_MB\n")
    for _, k, n, af, aw in rows:
        fo.write(f'"{k}",{n},{af:.0f},{2 * af / 1024:.1f},{aw:.0f},{aw / 1024:.1f}\n')
print(f"wrote {len(rows)} kernels")
