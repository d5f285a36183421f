"""LayerNorm backward at the encoder shape (102 080 x 256, fp32 stream, bf16 saved pre-norm sum, dropout 0.1), 100 launches.
POET_LN_BWD_NB (probe switch of this round's experiment) = the grid cap."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops
rows, d = 102080, 256
dy = torch.randn(rows, d, device="cuda"); z = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
mean = torch.randn(rows, device="cuda"); rstd = torch.rand(rows, device="cuda") + 0.5
gamma = torch.randn(d, device="cuda")
dz = torch.empty(rows, d, device="cuda"); dx = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
def run(): ops.ln_bwd(dy, z, mean, rstd, gamma, dz, dx, dg, db, rows, d, 0.1, 77)
for _ in range(10): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 10
print("cap", os.environ.get("POET_LN_BWD_NB", "512"), f"ln_bwd {t:6.1f} us  ({(rows * d * (4 + 2 + 4 + 2)) / t / 1e6:.2f} TB/s of 312 MB)")
