"""256 -> N forward products at 102 080 rows: single bf16 weight image against split (hi + lo) weights, with / without ReLU + dropout.
Usage: python profiles/probes/ws_nosplit_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops
rows, K = 102080, 256
x = torch.randn(rows, K, device="cuda").to(torch.bfloat16)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (1024, 768, 256):
    w = torch.randn(N, K, device="cuda") / 16
    w16 = w.to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.empty(rows, N, dtype=torch.bfloat16, device="cuda")
    for act, dp in ((0, 0.0), (1, 0.0), (1, 0.1)):
        ts = timeit(lambda: ops.linear_fwd(x, w, b, out, split=True, act=act, drop_p=dp, seed=7))
        tn = timeit(lambda: ops.linear_fwd(x, w16, b, out, split=False, act=act, drop_p=dp, seed=7))
        print(f"256 -> {N} relu={act} drop={dp}: split {ts:7.1f} us   single image {tn:7.1f} us", flush=True)
