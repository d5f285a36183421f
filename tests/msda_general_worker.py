"""Helper of test_msda_encoder_kernels_vs_explicit_full_geometry: runs the encoder's fused MSDA forward and d(offsets | logits) on
the GENERAL gather kernels.  The library reads its kernel-selection switches ONCE per process (no getenv in the launch path), so
the variant runs in its own process with POET_MSDA_NO_SHARED=1.  argv: in.npz out.npz.  Not a test module."""
import os
import sys

import numpy as np
import torch

os.environ["POET_MSDA_NO_SHARED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from poet_amd import ops
    z = np.load(sys.argv[1])
    shapes = [tuple(int(v) for v in s) for s in z["shapes"]]
    n, m, d, p = (int(v) for v in z["nmdp"])
    geom = ops.LevelGeom(shapes)
    S, L = geom.S, len(shapes)
    mlp = m * L * p
    bf = lambda k: torch.from_numpy(z[k]).cuda().to(torch.bfloat16)         # (stored as the exact fp32 images of the bf16 values)
    vdev, oa, gout = bf("value_hm"), bf("oa"), bf("gout")
    ref = torch.from_numpy(z["ref"]).cuda()
    vstr = (m * S * d, d, S * d)
    out = torch.empty(n, S, m * d, dtype=torch.bfloat16, device="cuda")
    goa = torch.empty_like(oa)
    ops.msda_fused_fwd(vdev, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True)
    ops.msda_fused_bwd(vdev, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, torch.zeros(n, m, S, d, device="cuda"), goa,
                       n, m, d, p, S, grid_queries=True, parts=1)
    torch.cuda.synchronize()
    np.savez(sys.argv[2], out=out.float().cpu().numpy(), goa=goa.float().cpu().numpy())


if __name__ == "__main__":
    main()
