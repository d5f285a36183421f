"""gemm_pipe.hip (deep-pipeline long-K GEMM) -- correctness against torch fp32 matmul and timing, per shape.
Run on the GPU box:  python profiles/probes/pipe_probe.py            (POET_PIPE_CFG=0|1|2, POET_GEMM_NO_PIPE=1 for the A/B arms)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from poet_amd import ops, _lib

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

bf = torch.bfloat16
torch.manual_seed(0)
lib = _lib.load()
cfg = os.environ.get("POET_PIPE_CFG", "0")
print(f"cfg={cfg} no_pipe={os.environ.get('POET_GEMM_NO_PIPE', '0')}", flush=True)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(16 * 6380, 1024), (16 * 6380, 1280), (16 * 6380, 512), (51200, 1024), (204000, 1024), (5000 + 7, 1024), (4096, 512)]
if quick:
    shapes = shapes[:1]
for M, K in shapes:
    N = 256
    A = torch.randn(M, K, device="cuda").to(bf)
    Wkn = (torch.randn(K, N, device="cuda") / 16).to(bf)               # [K][N]: the input-gradient form (b_kmajor)
    Wnk = Wkn.t().contiguous()                                          # [N][K]: nn.Linear layout
    C0 = torch.randn(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    ref = A.float() @ Wkn.float()
    # 1) dX form, accumulate
    out = C0.clone()
    ops.gemm(A, Wkn, out, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, add_src=out, ld_add=N)
    path = ops._GEMM_PATHS[lib.poet_gemm_last_path()]
    e1 = ((out - (C0 + ref)).abs().max() / ref.abs().max()).item()
    t1 = timeit(lambda: ops.gemm(A, Wkn, out, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, add_src=out, ld_add=N))
    # 2) dX form, write
    out2 = torch.full_like(C0, float("nan"))
    ops.gemm(A, Wkn, out2, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True)
    e2 = ((out2 - ref).abs().max() / ref.abs().max()).item()
    t2 = timeit(lambda: ops.gemm(A, Wkn, out2, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True))
    # 3) forward layout + bias
    out3 = torch.full_like(C0, float("nan"))
    ops.linear_fwd(A, Wnk, bias, out3)
    e3 = ((out3 - (ref + bias)).abs().max() / ref.abs().max()).item()
    t3 = timeit(lambda: ops.linear_fwd(A, Wnk, bias, out3))
    # 4) forward layout, split weights (hi + lo) + bias
    Wf = torch.randn(N, K, device="cuda") / 16
    hi = Wf.to(bf); lo = (Wf - hi.float()).to(bf)
    ref4 = A.float() @ (hi.float() + lo.float()).t() + bias
    out4 = torch.full_like(C0, float("nan"))
    e4 = t4 = float("nan")
    if os.environ.get("POET_GEMM_NO_PIPE", "0") in ("", "0"):
        ops.linear_fwd(A, hi, bias, out4, W_lo=lo)
        e4 = ((out4 - ref4).abs().max() / ref4.abs().max()).item()
        t4 = timeit(lambda: ops.linear_fwd(A, hi, bias, out4, W_lo=lo))
    # 5) forward layout accumulate (STSLACK A/B arm lives here)
    out5 = C0.clone()
    ops.linear_fwd(A, Wnk, None, out5, add_src=out5)
    e5 = ((out5 - (C0 + ref)).abs().max() / ref.abs().max()).item()
    t5 = timeit(lambda: ops.linear_fwd(A, Wnk, None, out5, add_src=out5))
    fl = (M * K * 2 + M * N * 8) / 6.0e6
    print(f"M={M:6d} K={K:4d} [{path}] dX+= {t1:6.1f} us (floor@6TB/s {fl:5.1f}) err {e1:.1e} | dX= {t2:6.1f} err {e2:.1e} | fwd+b {t3:6.1f} err {e3:.1e} | "
          f"split {t4:6.1f} err {e4:.1e} | fwd+= {t5:6.1f} err {e5:.1e}", flush=True)
    del A, Wkn, Wnk, C0, ref, out, out2, out3, out4, out5, ref4
