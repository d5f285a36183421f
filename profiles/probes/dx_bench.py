import os, sys, torch
sys.path.insert(0, os.getcwd())
from poet_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 16 * 6380
bf = torch.bfloat16
for n_out, n_in in [(1024, 256), (768, 256), (1280, 256), (512, 256)]:
    w = (torch.randn(n_out, n_in, device="cuda") / 16).to(bf)
    wt = w.t().contiguous()
    dy = torch.randn(M, n_out, device="cuda").to(bf)
    acc = torch.randn(M, n_in, device="cuda")
    o1 = torch.empty(M, n_in, device="cuda"); o2 = torch.empty_like(o1)
    t1 = timeit(lambda: ops.linear_dx(dy, w, o1, rows=M, add_src=acc))
    t2 = timeit(lambda: ops.linear_fwd(dy, wt, None, o2, add_src=acc))
    err = (o1 - o2).abs().max().item()
    byt = dy.numel() * 2 + o1.numel() * 8
    print(f"dX {n_in}<-{n_out}: k-major weight {t1:7.1f} us   transposed shadow {t2:7.1f} us   (floor {byt/6e6:5.1f} us) maxdiff {err:.2e}", flush=True)
