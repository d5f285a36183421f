import os, sys, torch, math
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
R = 320
x = torch.randn(R, 256, device="cuda"); w = torch.randn(256, 256, device="cuda") / 16; b = torch.randn(256, device="cuda")
out = torch.empty(R, 256, device="cuda"); dy = torch.randn(R, 256, device="cuda"); dw = torch.zeros(256, 256, device="cuda"); db = torch.zeros(256, device="cuda")
w4 = torch.randn(1024, 256, device="cuda") / 16; h = torch.empty(R, 1024, device="cuda"); dh = torch.randn(R, 1024, device="cuda"); dw4 = torch.zeros(1024, 256, device="cuda"); db4 = torch.zeros(1024, device="cuda")
print("fwd 256->256 :", round(graph_time(lambda: ops.linear_fwd(x, w, b, out, act=1)), 2), "us")
print("fwd 256->1024:", round(graph_time(lambda: ops.linear_fwd(x, w4, None, h, act=1)), 2), "us")
print("fwd 1024->256:", round(graph_time(lambda: ops.linear_fwd(h, w4.t().contiguous(), b, out)), 2), "us")
print("dX  256<-256 :", round(graph_time(lambda: ops.linear_dx(dy, w, out, rows=R)), 2), "us")
print("dX  256<-1024:", round(graph_time(lambda: ops.linear_dx(dh, w4, out, rows=R)), 2), "us")
print("dW  256x256  :", round(graph_time(lambda: ops.linear_dw(dy, x, dw, rows=R, db=db)), 2), "us")
print("dW  1024x256 :", round(graph_time(lambda: ops.linear_dw(dh, x, dw4, rows=R, db=db4)), 2), "us")
