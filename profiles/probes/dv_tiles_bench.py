import os, sys, torch, numpy as np, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    cfgname = sys.argv[2]
    cands = ["", "4,4,4", "8,8,4", "8,4,4", "4,8,4", "8,8,5", "8,8,3", "8,6,4", "6,6,4", "6,8,4", "10,8,4", "16,8,4", "8,8,6", "5,5,4", "6,5,4", "4,4,5", "4,4,6", "5,4,4", "4,3,4", "4,4,3"]
    for c in cands:
        env = dict(os.environ)
        if c: env["POET_DV_TILES"] = c
        r = subprocess.run([sys.executable, __file__, "run", cfgname], env=env, capture_output=True, text=True)
        print(f"tiles {c or 'default':10s}: ", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
    sys.exit(0)
sys.path.insert(0, os.getcwd())
from poet_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
name = sys.argv[2]
shapes, n = {"ycbv": ([(60, 80), (30, 40), (15, 20), (8, 10)], 16), "hires": ([(120, 160), (60, 80), (30, 40), (15, 20)], 8), "lmo": ([(30, 40), (15, 20), (8, 10), (4, 5)], 32)}[name]
m, d, p = 16, 16, 4
geom = ops.LevelGeom(shapes); S = geom.S; L = 4; mlp = m * L * p
g = torch.Generator(device="cuda").manual_seed(0)
value = torch.randn(n, m, S, d, device="cuda", generator=g).to(torch.bfloat16)
th = np.arange(m) * (2 * np.pi / m)
grid = np.stack([np.cos(th), np.sin(th)], -1); grid = grid / np.abs(grid).max(-1, keepdims=True)
base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1).reshape(-1)
off = torch.from_numpy(base.astype(np.float32)).cuda()[None, None] + 0.3 * torch.randn(n, S, 2 * mlp, device="cuda", generator=g)
oa = torch.cat([off, torch.randn(n, S, mlp, device="cuda", generator=g)], -1).to(torch.bfloat16).contiguous()
gout = torch.randn(n, S, m * d, device="cuda", generator=g).to(torch.bfloat16)
ref = torch.empty(n, S, L, 2, device="cuda")
ops.enc_ref_points(torch.ones(n, L, 2, device="cuda"), geom, ref, n)
vstr = (m * S * d, d, S * d)
gv = torch.zeros(n, m, S, d, device="cuda", dtype=torch.bfloat16 if os.environ.get("GV", "bf16") == "bf16" else torch.float32); goa = torch.empty_like(oa)   # (GV=f32: the fp32 map)
def dv(): ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)
print(f"dV {timeit(dv):.1f} us")
