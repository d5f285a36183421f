"""A/B of the encoder's value-gradient scatter: LDS-tiled DPP kernel (msda_bwd_dv_tiled_kernel) against the matrix-core form
(msda_bwd_dv_mfma_kernel, POET_DV_MFMA=1), same inputs, per launch, at several offset-noise levels (0 = the reference's initial
sampling pattern, identical for all queries: what the benchmark runs; 0.3 px = the kernel tests' noise) and with phases switched
off (POET_DV_SKIP bits: 1 max pass, 2 accumulate, 4 flush).  Usage: python profiles/probes/dv_mfma_ab.py [ycbv|lmo|hires]"""
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


name = sys.argv[1] if len(sys.argv) > 1 else "ycbv"
shapes, n = {"ycbv": ([(60, 80), (30, 40), (15, 20), (8, 10)], 16), "hires": ([(120, 160), (60, 80), (30, 40), (15, 20)], 8),
             "lmo": ([(30, 40), (15, 20), (8, 10), (4, 5)], 32)}[name]
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
noise_levels = [float(x) for x in os.environ.get("NOISE", "0,0.05,0.3").split(",")]
for noise in noise_levels:
    off = torch.from_numpy(base.astype(np.float32)).cuda()[None, None] + noise * torch.randn(n, S, 2 * mlp, device="cuda", generator=g)
    oa = torch.cat([off, torch.randn(n, S, mlp, device="cuda", generator=g)], -1).to(torch.bfloat16).contiguous()
    goa = torch.empty_like(oa)
    res = {}
    for gvdt in ((torch.bfloat16,) if os.environ.get("AB_QUICK") else (torch.bfloat16, torch.float32)):
        gv = torch.zeros(n, m, S, d, device="cuda", dtype=gvdt)
        def dv(): ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)
        for mf in ("0", "1"):
            os.environ["POET_DV_MFMA"] = mf
            os.environ.pop("POET_DV_SKIP", None)
            gv.zero_(); dv(); torch.cuda.synchronize()
            res[(gvdt, mf)] = gv.float().clone()
            t = timeit(dv)
            ph = []
            if gvdt == torch.bfloat16 and not os.environ.get("AB_QUICK"):
                for sk in (7, 6, 5, 3, 2):
                    os.environ["POET_DV_SKIP"] = str(sk)
                    ph.append(f"skip{sk}={timeit(dv, 10):.0f}")
                os.environ.pop("POET_DV_SKIP", None)
            print(f"{name} noise {noise}: {'mfma ' if mf == '1' else 'tiled'} gv {str(gvdt)[6:]:8s} {t:7.1f} us  {' '.join(ph)}", flush=True)
        a, b = res[(gvdt, "0")], res[(gvdt, "1")]
        print(f"    mfma vs tiled: max |diff| / max |dV| = {(a - b).abs().max().item() / a.abs().max().item():.2e}", flush=True)
os.environ["POET_DV_MFMA"] = "0"
