"""Software-pipelined split-weight forward (gemm_wsp_kernel) against gemm_ws_kernel: run once per POET_WS_PIPE setting, the second run compares.
python profiles/probes/wsp_check.py  (POET_WS_PIPE=0 first, then =1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poet_amd import ops
tag = os.environ.get("POET_WS_PIPE", "1")
torch.manual_seed(0)
res = {}
for rows in (102080, 51200 + 24, 4096 + 7):
    x = torch.randn(rows, 256, device="cuda").to(torch.bfloat16)
    for N, odt, act, dp in ((1024, torch.bfloat16, 1, 0.1), (1024, torch.bfloat16, 1, 0.0), (384, torch.float16, 0, 0.0), (768, torch.bfloat16, 0, 0.0), (512, torch.float16, 1, 0.1)):
        w = torch.randn(N, 256, device="cuda") / 16
        b = torch.randn(N, device="cuda")
        out = torch.full((rows + 8, N), 7.0, dtype=odt, device="cuda")
        ops.linear_fwd(x, w, b, out[:rows], split=True, act=act, drop_p=dp, seed=1234)
        assert (out[rows:] == 7.0).all(), "wrote past the last row"
        res[(rows, N, str(odt), act, dp)] = out[:rows].float().cpu()
path = "/tmp/wsp_check_%s.pt" % tag
torch.save(res, path)
other = "/tmp/wsp_check_%s.pt" % ("0" if tag != "0" else "1")
if os.path.exists(other):
    ref = torch.load(other)
    for k, v in res.items():
        r = ref[k]
        mask_same = ((v == 0) == (r == 0)).float().mean().item()
        err = (v - r).abs().max().item() / r.abs().max().item()
        print(k, "zero pattern equal on %.6f of the outputs, max |diff| / max |ref| = %.2e" % (mask_same, err))
