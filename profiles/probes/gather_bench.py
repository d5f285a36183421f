"""Forward and d(offsets | logits) gathers of the encoder shape (shared-geometry kernels), per launch.  A/B by environment switches read
once per process (run one process per variant).  Usage: python profiles/probes/gather_bench.py [ycbv|lmo|hires] [bf16|f16]
(second argument, round 6: storage type of the value maps)"""
import os, sys, torch, numpy as np
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


name = sys.argv[1] if len(sys.argv) > 1 else "ycbv"
shapes, n = {"ycbv": ([(60, 80), (30, 40), (15, 20), (8, 10)], 16), "hires": ([(120, 160), (60, 80), (30, 40), (15, 20)], 8),
             "lmo": ([(30, 40), (15, 20), (8, 10), (4, 5)], 32)}[name]
m, d, p = 16, 16, 4
geom = ops.LevelGeom(shapes); S = geom.S; L = 4; mlp = m * L * p
g = torch.Generator(device="cuda").manual_seed(0)
vdt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.bfloat16
value = torch.randn(n, m, S, d, device="cuda", generator=g).to(vdt)
th = np.arange(m) * (2 * np.pi / m)
grid = np.stack([np.cos(th), np.sin(th)], -1); grid = grid / np.abs(grid).max(-1, keepdims=True)
base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1).reshape(-1)
gout = torch.randn(n, S, m * d, device="cuda", generator=g).to(torch.bfloat16)
ref = torch.empty(n, S, L, 2, device="cuda")
ops.enc_ref_points(torch.ones(n, L, 2, device="cuda"), geom, ref, n)
vstr = (m * S * d, d, S * d)
off = torch.from_numpy(base.astype(np.float32)).cuda()[None, None] + 0.3 * torch.randn(n, S, 2 * mlp, device="cuda", generator=g)
oa = torch.cat([off, torch.randn(n, S, mlp, device="cuda", generator=g)], -1).to(torch.float16).contiguous()
goa = torch.empty(n, S, 3 * mlp, dtype=torch.bfloat16, device="cuda")
gv = torch.zeros(n, m, S, d, device="cuda", dtype=torch.bfloat16)
out = torch.empty(n, S, m * d, dtype=torch.bfloat16, device="cuda")
def fwd(): ops.msda_fused_fwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True)
def bwd(): ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=1)
def dv(): ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)
tf, tb, tv = timeit(fwd), timeit(bwd), timeit(dv)
print(f"{name} V={str(vdt)[6:]}: fwd {tf:7.1f} us   d(off|logit) {tb:7.1f} us   dV scatter {tv:7.1f} us   checksums out {out.float().abs().sum().item():.6e} goa {goa.float().abs().sum().item():.6e}", flush=True)
