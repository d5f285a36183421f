"""poet_linear_bwd (dW + db + dX in one pass, gemm_dwx_kernel) against the two poet_gemm calls, per launch pair, at the encoder's two shapes
(FFN linear2: n2 = 1024 gated; output_proj: n2 = 256).  Usage: python profiles/probes/dwx_bench.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 102080
for n2, gate in ((1024, True), (256, False)):
    dy = torch.randn(rows, 256, device="cuda").to(torch.bfloat16)
    x = torch.relu(torch.randn(rows, n2, device="cuda")).to(torch.bfloat16)
    w = (torch.randn(256, n2, device="cuda") / 16).to(torch.bfloat16)
    dw, db = torch.zeros(256, n2, device="cuda"), torch.zeros(256, device="cuda")
    dx = torch.empty(rows, n2, dtype=torch.bfloat16, device="cuda")
    tf = timeit(lambda: ops.linear_bwd(dy, x, w, dw, db, dx, rows=rows, gate=gate, gate_scale=1.1))
    t1 = timeit(lambda: ops.linear_dw(dy, x, dw, rows=rows, db=db))
    t2 = timeit(lambda: ops.linear_dx(dy, w, dx, rows=rows, gate_ref=x if gate else None, gate_scale=1.1))
    byt = rows * (256 * 2 + n2 * 2 + n2 * 2)
    print(f"rows {rows} n2 {n2} gate {gate}: fused {tf:7.1f} us ({byt / tf / 1e3:5.0f} GB/s)   dW {t1:6.1f} + dX {t2:6.1f} = {t1 + t2:6.1f} us", flush=True)
