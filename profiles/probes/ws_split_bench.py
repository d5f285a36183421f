"""Isolated timing of the split-weight streaming forward GEMM (offsets | logits: N = 768, FFN1: N = 1024; 102 080 rows, K = 256).
LIB=path loads another build of the library (ABI check relaxed) for same-box A/B.  Usage: python profiles/probes/ws_split_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
    _lib.ABI_VERSION = int(os.environ.get("LIB_ABI", "2"))
    for k in ("poet_nchw_to_tokens_split", "poet_split_rows"):      # (entry points newer than the other build)
        _lib._PROTOS.pop(k, None)
from poet_amd import ops
rows, K = 102080, 256
x = torch.randn(rows, K, device="cuda").to(torch.bfloat16)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (768, 1024, 256):
    w = torch.randn(N, K, device="cuda") / 16
    b = torch.randn(N, device="cuda")
    for odt in (torch.bfloat16,) + ((torch.float16,) if not os.environ.get("LIB") else ()):
        out = torch.empty(rows, N, dtype=odt, device="cuda")
        dp = float(os.environ.get("DROP", "0.1")) if (N == 1024 and odt == torch.bfloat16) else 0.0      # FFN1 runs with its dropout in the step
        t = timeit(lambda: ops.linear_fwd(x, w, b, out, split=True, act=1 if N == 1024 else 0, drop_p=dp, seed=1234))
        ref = torch.relu(x[:4096].float() @ w.t() + b) if N == 1024 else x[:4096].float() @ w.t() + b
        got = out[:4096].float()
        keep = got != 0 if dp > 0 else torch.ones_like(got, dtype=torch.bool)
        err = ((got * (1 - dp) - ref).abs() * keep).max().item() / ref.abs().max().item()
        print(f"split fwd 256 -> {N} out {str(odt)[6:]:9s} drop {dp}: {t:7.1f} us   rel err (first 4096 rows, kept elements) {err:.2e}")
