"""What does the second MFMA of the split-bf16 forward products cost today?  The encoder's five forward Linears at 102 080 rows, split
weights (fp32 master, bf16 hi + lo: the policy) against ONE bf16 product per fragment (`bf16_nosplit`) -- the matrix work an fp16-weight
form (single IEEE fp16 weight image, the bf16 activations converted to fp16 in registers: exact) has, and that form itself
(gemm_ws.hip WM = 2, the K = 256 products).  Operand sets rotate through > 256 MB: with one set the infinity cache serves the re-read
activations and absorbs the stores, and every single-product (memory-leaning) form looks 10-25 us faster than it is in the step.
    python profiles/probes/split_cost_probe.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poet_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 102080
dev = "cuda"
g = torch.Generator().manual_seed(0)


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


for name, K, N, odt, kw in (("value / output proj (fp16 out)", 256, 256, torch.float16, {}),
                            ("offsets | logits (fp16 out)", 256, 768, torch.float16, {}),
                            ("FFN1 (bf16 out, relu + dropout)", 256, 1024, torch.bfloat16, dict(act=1, drop_p=0.1, seed=3)),
                            ("FFN2 (fp16 out, K = 1024)", 1024, 256, torch.float16, {})):
    NB = 6                                                              # rotating operand sets (> the 256 MB infinity cache): every launch streams from / to HBM
    xs = [torch.randn(rows, K, generator=g).to(torch.bfloat16).to(dev) for _ in range(NB)]
    x = xs[0]
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    w16 = w.to(torch.bfloat16)
    wlo = (w - w16.float()).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    outs = [torch.empty(rows, N, dtype=odt, device=dev) for _ in range(NB)]
    out = outs[0]
    ctr = [0]

    def rot():
        ctr[0] = (ctr[0] + 1) % NB
        return xs[ctr[0]], outs[ctr[0]]
    t_h = float("nan")
    if K == 1024:
        t_split = timeit(lambda: ops.linear_fwd(*rot()[:1], w16, b, outs[ctr[0]], W_lo=wlo))
        t_one = timeit(lambda: ops.linear_fwd(*rot()[:1], w16, b, outs[ctr[0]]))
    else:
        ops._W16 = False
        t_split = timeit(lambda: ops.linear_fwd(*rot()[:1], w, b, outs[ctr[0]], split=True, **kw))
        t_one = timeit(lambda: ops.linear_fwd(*rot()[:1], w16, b, outs[ctr[0]], **kw))
        ops._W16 = True                                              # the fp16-weight form (gemm_ws.hip WM = 2)
        t_h = timeit(lambda: ops.linear_fwd(*rot()[:1], w, b, outs[ctr[0]], split=True, **kw))
        ops._W16 = False
    print(f"{name:36s} split {t_split:7.1f} us   one bf16 product {t_one:7.1f} us   one fp16 product of the fp32 master {t_h:7.1f} us")
