"""Round 6: WHICH pairs of the step's kernels overlap at all, and does the way they are launched matter?
A = a VALU / LDS / L1-bound MSDA kernel (forward gather, d(offsets | logits) gather, LDS-tiled d(value) scatter); B = an HBM-bound
kernel (streaming copy of 400 MB, LayerNorm forward over 102 080 rows, the weight-gradient DMA ring).  Each pair four ways:
  serial      A then B on one stream (eager)
  2 streams   B on a second stream forked / joined by events (eager)
  1 graph     the same fork captured as two branches of ONE replayed graph
  2 graphs    A and B captured as two graphs, replayed on two streams between two events
overlap = (t_A + t_B - t_pair) / min(t_A, t_B): 1 = the shorter kernel is hidden completely, 0 = none, < 0 = slower than serial.
Co-residency is decided by LDS: the scatter holds 157 KB of the CU's 160, the gathers 33 KB per 256-thread workgroup, the ring 144 KB."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops

shapes = [(60, 80), (30, 40), (15, 20), (8, 10)]
n, m, d, p = 16, 16, 16, 4
geom = ops.LevelGeom(shapes); S = geom.S; L = 4; mlp = m * L * p
g = torch.Generator(device="cuda").manual_seed(0)
value = torch.randn(n, m, S, d, device="cuda", generator=g).to(torch.float16)
th = np.arange(m) * (2 * np.pi / m)
grid = np.stack([np.cos(th), np.sin(th)], -1); grid = grid / np.abs(grid).max(-1, keepdims=True)
base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1).reshape(-1)
off = torch.from_numpy(base.astype(np.float32)).cuda()[None, None] + 0.3 * torch.randn(n, S, 2 * mlp, device="cuda", generator=g)
oa = torch.cat([off, torch.randn(n, S, mlp, device="cuda", generator=g)], -1).to(torch.float16).contiguous()
gout = torch.randn(n, S, m * d, device="cuda", generator=g).to(torch.bfloat16)
ref = torch.empty(n, S, L, 2, device="cuda")
ops.enc_ref_points(torch.ones(n, L, 2, device="cuda"), geom, ref, n)
vstr = (m * S * d, d, S * d)
gv = torch.zeros(n, m, S, d, device="cuda", dtype=torch.bfloat16); goa = torch.empty(n, S, 3 * mlp, dtype=torch.bfloat16, device="cuda")
out = torch.empty(n, S, m * d, dtype=torch.bfloat16, device="cuda")
M = n * S
src32 = torch.randn(M, 512, device="cuda"); dst32 = torch.empty_like(src32)                     # 209 MB in + 209 MB out
x16 = torch.randn(M, 256, device="cuda").to(torch.float16); res = torch.randn(M, 256, device="cuda")
gam, bet = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
y = torch.empty(M, 256, device="cuda"); z = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda"); y16 = torch.empty_like(z)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dy = torch.randn(M, 256, device="cuda").to(torch.bfloat16); hh = torch.randn(M, 1024, device="cuda").to(torch.bfloat16)
dw1 = torch.zeros(256, 1024, device="cuda")

A = {"gather fwd": lambda: ops.msda_fused_fwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True),
     "gather d(off|logit)": lambda: ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=1),
     "scatter d(value)": lambda: ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True, parts=2)}
B = {"copy 418 MB": lambda: ops.cast(src32, dst32),
     "LayerNorm fwd": lambda: ops.ln_fwd(x16, res, gam, bet, y, z, mean, rstd, M, 256, 1e-5, 0.0, 0, y16=y16),
     "dW ring 256x1024": lambda: ops.linear_dw(dy, hh, dw1, rows=M)}


def timeit(fn, nrep=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep * 1e3


side = torch.cuda.Stream()


def on_side(fn):
    prev = ops._STREAM_OVERRIDE[0]
    with torch.cuda.stream(side):
        ops._STREAM_OVERRIDE[0] = side.cuda_stream
        try:
            fn()
        finally:
            ops._STREAM_OVERRIDE[0] = prev


def forked(a, b):
    def run():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        on_side(b)
        a()
        main.wait_stream(side)
    return run


def capture(fn, stream=None):
    st = stream or torch.cuda.Stream()
    with torch.cuda.stream(st):
        prev = ops._STREAM_OVERRIDE[0]
        ops._STREAM_OVERRIDE[0] = st.cuda_stream
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            ops._STREAM_OVERRIDE[0] = torch.cuda.current_stream().cuda_stream
            fn()
        ops._STREAM_OVERRIDE[0] = prev
    return gr


for an, a in A.items():
    ta = timeit(a)
    for bn, b in B.items():
        tb = timeit(b)
        t_ser = timeit(lambda: (a(), b()))
        t_2s = timeit(forked(a, b))
        g1 = capture(forked(a, b))
        t_1g = timeit(g1.replay)
        ga, gb = capture(a), capture(b)
        def two_graphs():
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                gb.replay()
            ga.replay()
            main.wait_stream(side)
        t_2g = timeit(two_graphs)
        ov = lambda t: (ta + tb - t) / min(ta, tb)
        print(f"{an:22s} ({ta:6.1f} us) || {bn:18s} ({tb:6.1f} us): serial {t_ser:6.1f}  2 streams {t_2s:6.1f} (overlap {ov(t_2s):+.2f})  "
              f"1 graph forked {t_1g:6.1f} ({ov(t_1g):+.2f})  2 graphs {t_2g:6.1f} ({ov(t_2g):+.2f})", flush=True)
        del g1, ga, gb
