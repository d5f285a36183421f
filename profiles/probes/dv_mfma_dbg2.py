import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
shapes, n = [(8, 8)], 1
m, d, p = 2, 16, 4
geom = ops.LevelGeom(shapes); S = geom.S; L = 1; mlp = m * L * p
value = torch.zeros(n, m, S, d, device="cuda", dtype=torch.bfloat16)
off = torch.zeros(n, S, 2 * mlp, device="cuda")           # all 4 points at the query's own pixel centre
lg = torch.zeros(n, S, mlp, device="cuda")
oa = torch.cat([off, lg], -1).to(torch.bfloat16).contiguous()
gout = torch.zeros(n, S, m * d, device="cuda")
gout[0, :, 0] = torch.arange(1, S + 1, device="cuda").float()       # channel 0: query index + 1
gout[0, :, 1] = 1.0
gout = gout.to(torch.bfloat16)
ref = torch.empty(n, S, L, 2, device="cuda")
ops.enc_ref_points(torch.ones(n, L, 2, device="cuda"), geom, ref, n)
vstr = (m * S * d, d, S * d)
goa = torch.empty_like(oa)
def run(mf, skip=None):
    os.environ["POET_DV_MFMA"] = mf
    if skip is None: os.environ.pop("POET_DV_SKIP", None)
    else: os.environ["POET_DV_SKIP"] = str(skip)
    gv = torch.zeros(n, m, S, d, device="cuda", dtype=torch.float32)
    ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)
    torch.cuda.synchronize()
    return gv
a = run("0"); c = run("1")
cs = run("1", 128 + 4 + 1)
print("W tile of sub-block 0, (l, pt) = (0, 0): rows = tile pixel, cols = query")
wt = cs.flatten()[:1024].view(64, 16).cpu()
for px in range(64):
    if wt[px].abs().sum() > 0: print(px, wt[px].tolist())
print("lane info (fit + 2 tcx + 32 tcy + 1024 mine):", cs.flatten()[1024:1088].cpu().tolist())
for sb in range(0):
    cs = run("1", 64 + 256 * sb)
    print("only sub-block", sb, "ch1:\n", cs[0, 0, :, 1].view(8, 8).cpu())
print("tiled ch0:\n", a[0, 0, :, 0].view(8, 8).cpu())
print("mfma ch0:\n", c[0, 0, :, 0].view(8, 8).cpu())
print("tiled ch1:\n", a[0, 0, :, 1].view(8, 8).cpu())
print("mfma ch1:\n", c[0, 0, :, 1].view(8, 8).cpu())
