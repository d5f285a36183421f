"""Helper of test_gemm_split_pipelined_vs_plain_streaming: the wide split-weight forward products on gemm_ws_kernel (POET_WS_PIPE=0; the
library reads its kernel-selection switches once per process, so the variant runs in a process of its own).  argv: in.npz out.npz.
Not a test module."""
import os
import sys

import numpy as np
import torch

os.environ["POET_WS_PIPE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from poet_amd import ops
    z = np.load(sys.argv[1])
    x = torch.from_numpy(z["x"]).cuda().to(torch.bfloat16)
    w, b = torch.from_numpy(z["w"]).cuda(), torch.from_numpy(z["b"]).cuda()
    act, seed, f16 = int(z["act"]), int(z["seed"]), int(z["f16"])
    dp = float(z["dp"])
    out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float16 if f16 else torch.bfloat16, device="cuda")
    ops.linear_fwd(x, w, b, out, split=True, act=act, drop_p=dp, seed=seed)
    np.savez(sys.argv[2], out=out.float().cpu().numpy())


if __name__ == "__main__":
    main()
