"""Can a chain of SMALL dependent kernels (the decoder + pose heads: ~450 launches of 1-40 workgroups, ~1.2 ms of the step, GPU almost
idle) hide under chip-filling kernels of another queue (the encoder of another micro-batch)?  round-4's overlap_probe.py forked two
branches that EACH fill every CU and saw no overlap; this one forks a small chain against a big stream.
Branch S: 300 dependent 320-row kernels (gemm_small 320 x 256 x 256 = the decoder's Linears, LayerNorm rows);
branch B: LayerNorm forward over 102 080 x 256 rows (HBM-bound, every CU), 30 launches.  Serial vs forked, as replayed HIP graphs."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from poet_amd import ops

dev = "cuda"
M = 102080
xs = torch.randn(320, 256, device=dev); w = torch.randn(256, 256, device=dev) / 16; b = torch.zeros(256, device=dev)
ys = [torch.empty(320, 256, device=dev) for _ in range(2)]
big_in = torch.randn(M, 256, device=dev); big_out = torch.empty(M, 256, device=dev)
big_in2 = torch.randn(M, 256, device=dev); big_out2 = torch.empty(M, 256, device=dev)

def S(n=300):
    cur = xs
    for i in range(n):
        out = ys[i & 1]
        ops.linear_fwd(cur, w, b, out)          # 320-row Linear: gemm_small, ~20 workgroups
        cur = out

def B(n=30):
    for i in range(n):
        torch.add(big_in, 1.0, out=big_out)     # 209 MB of traffic per launch, fills the chip
        torch.add(big_in2, 1.0, out=big_out2)

def timeit(fn, nrep=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep * 1e3

side = torch.cuda.Stream()
def forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
    prev = ops._STREAM_OVERRIDE[0]
    with torch.cuda.stream(side):
        ops._STREAM_OVERRIDE[0] = side.cuda_stream
        S()
        ops._STREAM_OVERRIDE[0] = prev
    B()
    main.wait_stream(side)

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

gS, gB, gSB, gF = graphed(S), graphed(B), graphed(lambda: (S(), B())), graphed(forked)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}")
print(f"graph S alone {timeit(gS.replay):.1f} us   B alone {timeit(gB.replay):.1f} us   S then B (one branch) {timeit(gSB.replay):.1f} us   S || B (forked) {timeit(gF.replay):.1f} us", flush=True)
print(f"eager two streams: {timeit(forked):.1f} us", flush=True)
