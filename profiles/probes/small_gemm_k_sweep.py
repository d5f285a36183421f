"""Is the 320-row decoder Linear bound by its matrix instructions?  (VERDICT r5 #3 asked for the 3-product split-bf16 form "or the bisect
that shows the floor is the dependent round trip and not the matrix rate".)  The same launch at K = 32 ... 1024: the fp32 MFMA count and the
operand bytes scale with K, the launch, the dependent load -> MFMA -> store round trip and the epilogue do not.  Replayed from a HIP graph
of 64 dependent launches (each reads the previous one's output where shapes allow, else the same input), us per launch.
    python profiles/probes/small_gemm_k_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poet_amd import ops

dev = "cuda"
rows, N = 320, 256
g = torch.Generator().manual_seed(0)
print(f"{'K':>5s} {'us / launch (graph of 64)':>26s} {'MFMA 16x16x4 per launch':>24s} {'weight bytes':>13s}")
for K in (32, 64, 128, 256, 512, 1024):
    x = torch.randn(rows, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    outs = [torch.empty(rows, N, device=dev) for _ in range(2)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.linear_fwd(x, w, b, outs[0])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(64):
                ops.linear_fwd(x, w, b, outs[i & 1])
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(5):
                gr.replay()
            e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / (5 * 64))
    print(f"{K:5d} {sorted(ts)[3]:26.2f} {(rows // 16) * (N // 16) * (K // 4):24d} {N * K * 4:13d}")
