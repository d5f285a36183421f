#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<tag>_<config>/) into the committed summaries that
bench.py / DESIGN.md cite:
  profiles/round<N>_<config>_kernel_stats.csv  per-kernel calls / total / average duration (kernel-trace pass)
  profiles/round<N>_<config>_pmc_hbm.csv       per-kernel average FETCH_SIZE / WRITE_SIZE per launch (two separate --pmc passes)
  profiles/round<N>_<config>_pmc_sq.csv        per-kernel MFMA / VALU / LDS counters per launch + the MFMA rate they imply
Usage: python profiles/summarize.py r2 ycbv"""
import csv, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
cfg = sys.argv[2] if len(sys.argv) > 2 else "ycbv"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", f"prof_{tag}_{cfg}")
rnd = "round" + tag.lstrip("r")


def short(name):
    if name.startswith("Cijk_"):                               # hipBLASLt / Tensile kernel: keep the macro tile, the MFMA shape and the stream-K tag
        m = re.search(r"(MT\d+x\d+x\d+)_(MI\d+x\d+x\d+)", name)
        sk = re.search(r"_(SK\d+)_", name)
        return "hipblaslt_Cijk_" + ("_".join(m.groups()) if m else "") + ("_" + sk.group(1) if sk else "")
    name = name.replace("unsigned short", "bf16").replace("void ", "").replace("poet::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def find(sub, suffix):
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def per_kernel(path, counters):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    if not path:
        return acc, n
    for r in csv.DictReader(open(path)):
        c = r["Counter_Name"]
        if c in counters:
            k = short(r["Kernel_Name"])
            acc[k][c] += float(r["Counter_Value"])
            n[k][c] += 1
    return acc, n


stats = find("trace", "kernel_stats.csv")
shutil.copy(stats, os.path.join(here, f"{rnd}_{cfg}_kernel_stats.csv"))
avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(stats))}
keep = lambda k: k.startswith(("gemm", "hipblaslt", "msda", "ln_", "gn_", "colsum", "lsa", "adamw", "vgrad", "add_", "mha"))

fa, fn = per_kernel(find("pmc_fetch", "counter_collection.csv"), {"FETCH_SIZE"})
wa, wn = per_kernel(find("pmc_write", "counter_collection.csv"), {"WRITE_SIZE"})
rows = []
for k in fa:
    if not keep(k):
        continue
    af = fa[k]["FETCH_SIZE"] / max(fn[k]["FETCH_SIZE"], 1)
    aw = wa[k]["WRITE_SIZE"] / max(wn[k]["WRITE_SIZE"], 1) if k in wa else 0.0
    rows.append((2 * af + aw, k, fn[k]["FETCH_SIZE"], af, aw))
rows.sort(reverse=True)
with open(os.path.join(here, f"{rnd}_{cfg}_pmc_hbm.csv"), "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/collect.sh), same bench command;\n")
    fo.write("# counter unit = KiB at the L2<->fabric boundary, averaged per launch over ALL launches of the kernel symbol\n")
    fo.write("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by 2x -> fetch_corrected = 2 * FETCH_SIZE\n")
    fo.write("kernel,launches,avg_FETCH_SIZE_KiB,avg_fetch_corrected_MB,avg_WRITE_SIZE_KiB,avg_write_MB\n")
    for _, k, n, af, aw in rows:
        fo.write(f'"{k}",{n},{af:.0f},{2 * af / 1024:.1f},{aw:.0f},{aw / 1024:.1f}\n')

SQ = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
SQ2 = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
sa, sn = per_kernel(find("pmc_sq", "counter_collection.csv"), set(SQ))
sb, sbn = per_kernel(find("pmc_sq2", "counter_collection.csv"), set(SQ2))
with open(os.path.join(here, f"{rnd}_{cfg}_pmc_sq.csv"), "w") as fo:
    fo.write("# rocprofv3 --pmc SQ_* (two passes of <= 8 counters, profiles/collect.sh); values are per-launch averages over all launches of the symbol.\n")
    fo.write("# mfma_TFLOPs = SQ_INSTS_MFMA * 16384 flop (v_mfma_f32_16x16x32_bf16; the fp32 16x16x4 form of gemm_small is 2048) / average kernel-trace duration;\n")
    fo.write("# mfma_frac = mfma_TFLOPs / 2500 (dense bf16 peak).  SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES = matrix-pipe busy share of the busy CU time.\n")
    fo.write("kernel,avg_us," + ",".join(SQ + SQ2) + ",mfma_TFLOPs,mfma_frac,mfma_busy_share\n")
    for k in sorted(sa, key=lambda k: -avg_ns.get(k, 0) * sn[k]["SQ_WAVES"]):
        if not keep(k):
            continue
        v = [sa[k][c] / max(sn[k][c], 1) for c in SQ] + [(sb[k][c] / max(sbn[k][c], 1) if k in sb else 0.0) for c in SQ2]
        ns = avg_ns.get(k, 0.0)
        fl = 2048 if k.startswith("gemm_small") or "<float, float, float, float" in k else 16384
        tf = v[2] * fl / ns / 1e3 if ns else 0.0
        share = v[7] / v[8] if v[8] else 0.0
        fo.write(f'"{k}",{ns / 1e3:.1f},' + ",".join(f"{x:.0f}" for x in v) + f",{tf:.1f},{tf / 2500:.4f},{share:.3f}\n")
print("wrote", rnd, cfg)
