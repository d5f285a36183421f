import os, sys, torch
sys.path.insert(0, os.getcwd())
from poet_amd import ops
def graph_time(fn, reps=50, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / reps * 1e3
N, Q, M, hd = 16, 20, 16, 16
d = M * hd
qkv = torch.randn(N * Q, 3 * d, device="cuda")
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
out = torch.empty(N * Q, d, device="cuda"); dout = torch.randn(N * Q, d, device="cuda")
dqkv = torch.empty(N * Q, 3 * d, device="cuda")
print("mha_fwd:", round(graph_time(lambda: ops.mha_fwd(q, k, v, 3 * d, out, d, N, Q, M, hd, drop_p=0.1, seed=3)), 2), "us")
print("mha_bwd:", round(graph_time(lambda: ops.mha_bwd(q, k, v, 3 * d, dout, d, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], 3 * d, N, Q, M, hd, drop_p=0.1, seed=3)), 2), "us")
