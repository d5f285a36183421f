"""Where do the waves of gemm_pipe_kernel spend their cycles?  (POET_PIPE_DBG=16 build: s_memtime around the phases of a step)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
prof = torch.zeros(256 * 3 * 4, dtype=torch.int64, device="cuda")
os.environ["POET_PIPE_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "16"
os.environ["POET_PIPE_PROF_PTR"] = hex(prof.data_ptr())
from poet_amd import ops
bf = torch.bfloat16
M, K, N = 16 * 6380, 1024, 256
A = torch.randn(M, K, device="cuda").to(bf)
W = (torch.randn(K, N, device="cuda") / 16).to(bf)
C = torch.randn(M, N, device="cuda")
for _ in range(3):
    ops.gemm(A, W, C, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, add_src=C, ld_add=N)
torch.cuda.synchronize()
p = prof.view(256, 3, 4).cpu().double()
for r, name, cols in ((0, "W loader", ("vmcnt wait", "barrier", "issue")), (1, "A loader", ("vmcnt wait", "barrier", "issue")), (2, "compute ", ("barrier", "mfma+lds", "total"))):
    st = p[:, r, 3]
    print(name, " per step (mean cycles over workgroups):", ", ".join(f"{c} {(p[:, r, i] / st).mean().item():7.0f}" for i, c in enumerate(cols)),
          f"| totals: " + ", ".join(f"{c} {p[:, r, i].mean().item():8.0f}" for i, c in enumerate(cols)))
