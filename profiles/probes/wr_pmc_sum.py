import csv, glob, os, sys, collections
src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_w" not in k: continue
        k = k.split("(")[0].replace("void poet::(anonymous namespace)::", "")[:60] + " grid " + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    print("   " + "  ".join(f"{c}={acc[k][c] / max(n[k][c], 1):.3g}" for c in sorted(acc[k])))
