"""Isolated timing + check of the weight-gradient GEMM (gemm_dw.hip) on the encoder's shapes.  Variants by environment:
POET_DW_OLD=1 (register-staged kernel), POET_DWP_CFG=0/1/2 (DMA-ring kernel: 4+4 loader waves x 8 stages, 2+2 x 4 stages with two
workgroups per CU, 2+2 x 8 stages), POET_DW_BLOCKS=<target workgroups>."""
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
bf = torch.bfloat16
Ms = [16 * 6380] + ([int(a) for a in sys.argv[1:]] if len(sys.argv) > 1 else [16 * 6380 - 13])
for M in Ms:
    for name, n_out, k_in in [("dW 256x256", 256, 256), ("dW 768x256", 768, 256), ("dW 1024x256", 1024, 256), ("dW 256x1024", 256, 1024), ("dW 1280x256", 1280, 256)]:
        torch.manual_seed(1)
        dy = torch.randn(M, n_out, device="cuda").to(bf)
        xx = torch.randn(M, k_in, device="cuda").to(bf)
        dw = torch.zeros(n_out, k_in, device="cuda")
        db = torch.zeros(n_out, device="cuda")
        t = timeit(lambda: ops.linear_dw(dy, xx, dw, rows=M, db=db))
        t0 = timeit(lambda: ops.linear_dw(dy, xx, dw, rows=M))            # without the fused bias gradient
        dw.zero_(); db.zero_(); ops.linear_dw(dy, xx, dw, rows=M, db=db)
        ref = dy.float().T @ xx.float()
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
        rb = dy.float().sum(0)
        errb = ((db - rb).abs().max() / rb.abs().max()).item()
        byt = dy.numel() * 2 + xx.numel() * 2
        print(f"M={M} {name:14s} {t:7.1f} us (no db {t0:6.1f})  {2.0*M*n_out*k_in/t/1e6:6.0f} TF/s  {byt/t/1e3:6.0f} GB/s (floor {byt/6e6:5.1f} us) relerr {err:.2e} db {errb:.2e}", flush=True)
