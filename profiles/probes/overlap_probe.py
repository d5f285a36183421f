"""Do two branches of a replayed HIP graph overlap?  Branch A: MSDA dV scatter + dq (VALU / LDS / L1 bound);
branch B: dW GEMMs (HBM / MFMA bound).  Compares serial capture, forked capture, and eager two-stream timing."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from poet_amd import ops
shapes = [(60, 80), (30, 40), (15, 20), (8, 10)]
n, m, d, p = 16, 16, 16, 4
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
gv = torch.zeros(n, m, S, d, device="cuda"); goa = torch.empty_like(oa)
M = n * S
dy = torch.randn(M, 256, device="cuda").to(torch.bfloat16); hh = torch.randn(M, 1024, device="cuda").to(torch.bfloat16)
xx = torch.randn(M, 256, device="cuda").to(torch.bfloat16)
dw1 = torch.zeros(256, 1024, device="cuda"); dw2 = torch.zeros(1024, 256, device="cuda"); dw3 = torch.zeros(256, 256, device="cuda")
def A(): ops.msda_fused_bwd(value, vstr, geom, oa, 3 * mlp, 2 * mlp, ref, S * L * 2, gout, gv, goa, n, m, d, p, S, grid_queries=True)
def B():
    ops.linear_dw(dy, hh, dw1, rows=M); ops.linear_dw(hh, xx, dw2, rows=M); ops.linear_dw(dy, xx, dw3, rows=M)
def timeit(fn, nrep=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep * 1e3
print(f"eager A {timeit(A):.1f} us  B {timeit(B):.1f} us  A+B serial {timeit(lambda: (A(), B())):.1f} us", flush=True)
side = torch.cuda.Stream()
def forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
    prev = ops._STREAM_OVERRIDE[0]
    with torch.cuda.stream(side):
        ops._STREAM_OVERRIDE[0] = side.cuda_stream
        B()
        ops._STREAM_OVERRIDE[0] = prev
    A()
    main.wait_stream(side)
print(f"eager forked {timeit(forked):.1f} us", flush=True)
for name, fn in (("serial", lambda: (A(), B())), ("forked", forked)):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        prev = ops._STREAM_OVERRIDE[0]
        with torch.cuda.graph(gr, stream=st):
            ops._STREAM_OVERRIDE[0] = torch.cuda.current_stream().cuda_stream
            fn()
            ops._STREAM_OVERRIDE[0] = prev
    print(f"graph {name}: {timeit(gr.replay):.1f} us", flush=True)
