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
for name, n_out, k_in in [("dW 256x256", 256, 256), ("dW 768x256", 768, 256), ("dW 1024x256", 1024, 256), ("dW 256x1024", 256, 1024), ("dW 1280x256", 1280, 256)]:
    dy = torch.randn(M, n_out, device="cuda").to(bf)
    xx = torch.randn(M, k_in, device="cuda").to(bf)
    dw = torch.zeros(n_out, k_in, device="cuda")
    t = timeit(lambda: ops.linear_dw(dy, xx, dw, rows=M))
    dw.zero_(); ops.linear_dw(dy, xx, dw, rows=M)
    ref = dy.float().T @ xx.float()
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    byt = dy.numel() * 2 + xx.numel() * 2
    print(f"{name:32s} {t:7.1f} us  {2.0*M*n_out*k_in/t/1e6:6.0f} TF/s  {byt/t/1e3:6.0f} GB/s (floor {byt/6e6:5.1f} us) relerr {err:.2e}", flush=True)
