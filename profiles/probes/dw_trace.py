import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_dw_kernel" in n or "dw_reduce" in n:
        seq.append(("dw" if "gemm_dw" in n else "red", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
per = len(seq) // 5
for shape in range(5):
    part = seq[shape * per:(shape + 1) * per]
    dw = sorted(t for k, t in part if k == "dw"); rd = sorted(t for k, t in part if k == "red")
    print("shape", shape, "gemm_dw %.1f us  reduce %.1f us" % (dw[len(dw) // 2], rd[len(rd) // 2] if rd else 0))
