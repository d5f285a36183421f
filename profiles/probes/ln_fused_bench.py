"""poet_linear_ln_fwd (projection + dropout + residual + LayerNorm in the pipeline kernel) against poet_gemm + poet_ln_fwd, per launch,
at the encoder's two shapes (FFN2: K = 1024, output_proj: K = 256; 102 080 rows).  Usage: python profiles/probes/ln_fused_bench.py [rows]"""
import os, sys, math, torch
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


rows, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 102080), 256
for K in (1024, 256):
    for with_q in (True, False):
        x = torch.randn(rows, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(d, K, device="cuda") / math.sqrt(K)
        hi = w.to(torch.bfloat16); lo = (w - hi.float()).to(torch.bfloat16)
        b = torch.randn(d, device="cuda"); res = torch.randn(rows, d, device="cuda")
        g, bt = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
        pos = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
        y, tmp = torch.empty(rows, d, device="cuda"), torch.empty(rows, d, device="cuda")
        y16, z, q = (torch.empty(rows, d, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        kw = dict(y16=y16, pos16=pos if with_q else None, q16=q if with_q else None)
        tf = timeit(lambda: ops.linear_ln_fwd(x, hi, lo, b, res, g, bt, y, z, mean, rstd, rows, 1e-5, 0.1, 5, **kw))
        if K >= 512:
            tg = timeit(lambda: ops.linear_fwd(x, hi, b, tmp, W_lo=lo))
        else:
            tg = timeit(lambda: ops.linear_fwd(x, w, b, tmp, split=True))
        tl = timeit(lambda: ops.ln_fwd(tmp, res, g, bt, y, z, mean, rstd, rows, d, 1e-5, 0.1, 5, **kw))
        print(f"rows {rows} K {K} q16 {with_q}: fused {tf:7.1f} us   gemm {tg:6.1f} + ln {tl:6.1f} = {tg + tl:6.1f} us", flush=True)
