"""Timing of the wide split-weight forward products at 102 080 rows (FFN1: 256 -> 1024, ReLU + dropout, bf16; offsets | logits: 256 -> 384 fp16;
256 -> 768 bf16), 200 launches each.  LIB=path loads another build (same ABI) for same-box A/B; POET_WS_PIPE=0 = gemm_ws_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
from poet_amd import ops
rows, K = 102080, 256
x = torch.randn(rows, K, device="cuda").to(torch.bfloat16)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for N, odt, act, dp in ((1024, torch.bfloat16, 1, 0.1), (768, torch.bfloat16, 0, 0.0), (384, torch.float16, 0, 0.0)):
    w = torch.randn(N, K, device="cuda") / 16
    b = torch.randn(N, device="cuda")
    o = torch.empty(rows, N, dtype=odt, device="cuda")
    out.append("%d: %6.1f" % (N, timeit(lambda: ops.linear_fwd(x, w, b, o, split=True, act=act, drop_p=dp, seed=1234))))
print(os.environ.get("LIB", "default")[-14:], "pipe=" + os.environ.get("POET_WS_PIPE", "1"), " us  ", "   ".join(out))
