"""LayerNorm launch timing at the encoder's shape (102 080 x 256): the two forward forms of a layer (after output_proj; after the
FFN with the next layer's query copy) and the backward, per storage of the residual / gradient stream.  One process per environment
setting (POET_LN_FWD_R / POET_LN_FWD_NB / POET_LN_BWD_R / POET_LN_BWD_NB are read once):
    python profiles/probes/ln_probe.py [rows]
Prints us per launch (median of 5 x 20 launches, HIP events) and the GB/s of the launch's own byte stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poet_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 102080
only = sys.argv[2] if len(sys.argv) > 2 else ""            # "fwd" / "bwd": one half only
d = 256
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
x16 = torch.randn(rows, d, generator=g).to(torch.float16).to(dev)
gamma = torch.ones(d, device=dev); beta = torch.zeros(d, device=dev)
pos = torch.randn(rows, d, generator=g).to(torch.bfloat16).to(dev)
mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
z = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
y16 = torch.empty_like(z); q16 = torch.empty_like(z)


def timeit(fn, reps=5, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


for rdt, ydt in ((torch.float32, torch.float32), (torch.float32, torch.float16), (torch.float16, torch.float16), (torch.float16, torch.float32)) if only != "bwd" else ():
    res = torch.randn(rows, d, generator=g).to(rdt).to(dev)
    y = torch.empty(rows, d, dtype=ydt, device=dev)
    es = lambda t: t.element_size()
    for q in (False, True):
        r16 = y16.clone() if rdt == torch.float16 else None                # (split stream: bf16 head + fp16 remainder)
        fn = lambda: ops.ln_fwd(x16, res, gamma, beta, y, z, mean, rstd, rows, d, 1e-5, 0.1, 7, y16=y16, pos16=pos if q else None, q16=q16 if q else None, res16=r16)
        us = timeit(fn)
        nbytes = rows * d * (2 + es(res) + (2 if r16 is not None else 0) + es(y) + 2 + 2 + (4 if q else 0))
        print(f"ln_fwd res {str(rdt)[6:]:8s} y {str(ydt)[6:]:8s} q16 {int(q)}: {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s")

zb = torch.randn(rows, d, generator=g).to(torch.bfloat16).to(dev)
dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
dx = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
for ddt, zdt in ((torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)) if only != "fwd" else ():
    dy = torch.randn(rows, d, generator=g).to(ddt).to(dev)
    dz = torch.empty(rows, d, dtype=zdt, device=dev)
    fn = lambda: ops.ln_bwd(dy, zb, mean, rstd, gamma, dz, dx, dg, db, rows, d, 0.1, 7)
    us = timeit(fn)
    nbytes = rows * d * (dy.element_size() + 2 + dz.element_size() + 2)
    print(f"ln_bwd dy {str(ddt)[6:]:8s} dz {str(zdt)[6:]:8s}: {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s")
