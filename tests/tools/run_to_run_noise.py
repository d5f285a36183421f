import os, sys, torch
sys.path.insert(0, os.getcwd())
import poet_amd
from tests.product_runner import build_product
rot = "quat"
# first part of the test: fp32 model, eval forward, criterion, backward
r = build_product("tiny", 2, True, torch.float32, rotation_mode=rot)
model, crit = r["model"], r["crit"]; model.eval()
out, n_boxes = model(r["samples"], r["targets"])
losses = crit(out, r["targets"], n_boxes)
total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
model.zero_grad(); total.backward()
def run(mode, lr, steps):
    rr = build_product("tiny", 2, True, "bf16", dropout=0.0, rotation_mode=rot)
    rr["model"].train()
    if mode == "eager":
        tr = poet_amd.Trainer(rr["model"], rr["crit"], lr=lr, weight_decay=1e-4, max_norm=0.1)
    else:
        tr = poet_amd.GraphedTrainer(rr["model"], rr["crit"], lr=lr, weight_decay=1e-4, max_norm=0.1, warm=1, segment_backward=(mode == "segmented"))
    snaps = []
    for s in range(steps):
        l = float(tr.step(rr["samples"], rr["targets"])[0]); torch.cuda.synchronize()
        snaps.append((l, tr.arena.flat.clone(), tr.arena.grad.clone(), float(tr.arena.sq[0])))
    return tr, snaps
res = {}
for mode in ("eager", "graph", "segmented"):
    run(mode, 0.0, 4)
    res[mode] = run(mode, 2e-4, 2)
te, se = res["eager"]
for mode in ("graph", "segmented"):
    t, s = res[mode]
    for k in range(2):
        dp = (s[k][1] - se[k][1]).abs(); dg = (s[k][2] - se[k][2]).abs()
        print(f"{mode} step {k}: loss {s[k][0]:.6f}/{se[k][0]:.6f} sqnorm {s[k][3]:.6e}/{se[k][3]:.6e}  param frac>1e-5 {(dp > 1e-5).float().mean().item():.4f} max {dp.max().item():.2e}; grad max diff {dg.max().item():.2e} (gmax {se[k][2].abs().max().item():.2e}) frac grad rel>1e-3 {((dg > 1e-3 * se[k][2].abs()) & (se[k][2].abs() > 1e-6)).float().mean().item():.4f}")
