"""Where do the waves of gemm_pipe_kernel spend their cycles?  (POET_PIPE_DBG=16 build: s_memtime around the phases of a step)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
prof = torch.zeros(256 * 2 * 8, dtype=torch.int64, device="cuda")
os.environ["POET_PIPE_DBG"] = "16"
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
p = prof.view(256, 2, 8).cpu().double()
names = ["vmcnt wait", "barrier", "issue(+flush,Cld)", "compute", "issue at PH0", "total", "steps", "start"]
for wv in (0, 1):
    print("wave", 0 if wv == 0 else 7)
    for i, n in enumerate(names[:7]):
        v = p[:, wv, i]
        print(f"  {n:20s} mean {v.mean().item():10.0f}  min {v.min().item():10.0f}  max {v.max().item():10.0f}")
st = p[:, 0, 7]
print("start skew (cycles): max-min", (st.max() - st.min()).item())
tot = p[:, 0, 5]
print("per step (mean cycles): wait %.0f barrier %.0f issue %.0f compute %.0f total/step %.0f" % tuple((p[:, 0, i] / p[:, 0, 6]).mean().item() for i in (0, 1, 2, 3, 5)))
