import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops
shapes, n = [(60, 80), (30, 40), (15, 20), (8, 10)], 1
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
noise = float(os.environ.get("NOISE", "0.3"))
off = torch.from_numpy(base.astype(np.float32)).cuda()[None, None] + noise * torch.randn(n, S, 2 * mlp, device="cuda", generator=g)
oa = torch.cat([off, torch.randn(n, S, mlp, device="cuda", generator=g)], -1).to(torch.bfloat16).contiguous()
goa = torch.empty_like(oa)
def run(mf, skip=None):
    os.environ["POET_DV_MFMA"] = mf
    if skip is None: os.environ.pop("POET_DV_SKIP", None)
    else: os.environ["POET_DV_SKIP"] = str(skip)
    gv = torch.zeros(n, m, S, d, device="cuda", dtype=torch.float32)
    ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)
    torch.cuda.synchronize()
    return gv
a = run("0"); b = run("1", 8); c = run("1")
for sk in (16, 32, 48):
    cc = run("1", sk)
    print("skip", sk, "mfma vs tiled:", ((a - cc).abs().max() / a.abs().max()).item())
mx = a.abs().max().item()
print("slow-only vs tiled:", ((a - b).abs().max() / mx).item())
print("mfma vs tiled:", ((a - c).abs().max() / mx).item())
# per level error of c
st = 0
for (h, w) in shapes:
    e = (a[:, :, st:st + h * w] - c[:, :, st:st + h * w]).abs().max().item() / mx
    e2 = (a[:, :, st:st + h * w] - b[:, :, st:st + h * w]).abs().max().item() / mx
    print("level", (h, w), "mfma err", e, "slow err", e2)
    st += h * w
# ratio statistics
idx = (a.abs() > 0.3 * mx)
print("median ratio c/a on big entries:", (c[idx] / a[idx]).median().item(), "b/a", (b[idx] / a[idx]).median().item())
