import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes, n = [(60, 80), (30, 40), (15, 20), (8, 10)], 16
m, d, p = 16, 16, 4
geom = ops.LevelGeom(shapes); S = geom.S; L = 4; mlp = m * L * p
g = torch.Generator(device="cuda").manual_seed(0)
value = torch.randn(n, m, S, d, device="cuda", generator=g).to(torch.bfloat16)
th = np.arange(m) * (2 * np.pi / m)
grid = np.stack([np.cos(th), np.sin(th)], -1); grid = grid / np.abs(grid).max(-1, keepdims=True)
base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1).reshape(-1)
gout = torch.randn(n, S, m * d, device="cuda", generator=g).to(torch.bfloat16)
ref = torch.empty(n, S, L, 2, device="cuda")
ops.enc_ref_points(torch.ones(n, L, 2, device="cuda"), geom, ref, n)
vstr = (m * S * d, d, S * d)
for noise in (0.0, 0.3):
    off = torch.from_numpy(base.astype(np.float32)).cuda()[None, None] + noise * torch.randn(n, S, 2 * mlp, device="cuda", generator=g)
    oa = torch.cat([off, torch.randn(n, S, mlp, device="cuda", generator=g)], -1).to(torch.bfloat16).contiguous()
    goa = torch.empty_like(oa)
    gv = torch.zeros(n, m, S, d, device="cuda", dtype=torch.bfloat16)
    def dv(): ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)
    for mf, sk in (("0", 0), ("1", 0), ("1", 8), ("1", 8 | 1), ("1", 2), ("1", 2 | 1)):
        os.environ["POET_DV_MFMA"] = mf
        os.environ["POET_DV_SKIP"] = str(sk)
        print(f"noise {noise} mfma={mf} skip={sk}: {timeit(dv):7.1f} us", flush=True)
