"""Do TWO chains of small dependent kernels overlap when they are parallel branches of a replayed HIP graph?  (overlap_small_probe.py: a small
chain under chip-filling kernels hides 18 % of itself.)  Chain A and chain B: 100 dependent 800-row Linears each (gemm_small 800 x 256 x 256 =
the decoder's Linears at the benched batch: 800 single-wave workgroups); serial on one stream against forked, 2 and 4 branches, as graphs."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from poet_amd import ops

dev = "cuda"
NB = 4
xs = [torch.randn(800, 256, device=dev) for _ in range(NB)]
w = torch.randn(256, 256, device=dev) / 16; b = torch.zeros(256, device=dev)
ys = [[torch.empty(800, 256, device=dev) for _ in range(2)] for _ in range(NB)]

def chain(j, n=100):
    cur = xs[j]
    for i in range(n):
        out = ys[j][i & 1]
        ops.linear_fwd(cur, w, b, out)
        cur = out

def serial(nb):
    for j in range(nb): chain(j)

sides = [torch.cuda.Stream() for _ in range(NB)]
def forked(nb):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    prev = ops._STREAM_OVERRIDE[0]
    for j in range(1, nb):
        sides[j].wait_event(ev)
        with torch.cuda.stream(sides[j]):
            ops._STREAM_OVERRIDE[0] = sides[j].cuda_stream
            chain(j)
            ops._STREAM_OVERRIDE[0] = prev
    chain(0)
    for j in range(1, nb): main.wait_stream(sides[j])

def graphed(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        prev = ops._STREAM_OVERRIDE[0]
        with torch.cuda.graph(gr, stream=st):
            ops._STREAM_OVERRIDE[0] = torch.cuda.current_stream().cuda_stream
            fn()
            ops._STREAM_OVERRIDE[0] = prev
    return gr

def timeit(g, nrep=20):
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep * 1e3

for nb in (1, 2, 4):
    gs = graphed(lambda: serial(nb)); gf = graphed(lambda: forked(nb))
    print(f"{nb} chains of 100 dependent 800-row Linears: serial {timeit(gs):8.1f} us   forked {timeit(gf):8.1f} us")
