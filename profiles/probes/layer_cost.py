"""Marginal cost of one decoder / encoder layer in the REPLAYED step (wall clock, no profiler): the kernel trace serialises
kernels and cannot show how much of the decoder's ~90 small launches per layer overlap."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, poet_amd
CONFIGS = bench.CONFIGS
device = torch.device("cuda:0")
def step_ms(enc, dec, batch=16, steps=30):
    cfg = dict(CONFIGS["ycbv"], enc_layers=enc, dec_layers=dec)
    torch.manual_seed(1); poet_amd.manual_seed(1)
    feats, targets = bench.synth_batch(cfg, batch, 1234, device)
    model, crit = bench.build_model(cfg, feats, "bf16", device)
    model.train()
    tr = poet_amd.GraphedTrainer(model, crit, lr=2e-4, weight_decay=1e-4, max_norm=0.1, warm=2)
    ih, iw = cfg["image_hw"]
    samples = poet_amd.NestedTensor(None, torch.zeros((batch, ih, iw), dtype=torch.bool, device=device))
    for _ in range(8): tr.step(samples, targets)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): tr.step(samples, targets)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
res = {}
for enc, dec in ((5, 5), (5, 1), (5, 3), (1, 5), (3, 5), (5, 5)):
    res[(enc, dec)] = step_ms(enc, dec)
    print(f"enc {enc} dec {dec}: {res[(enc, dec)]:.3f} ms/step", flush=True)
print(f"per decoder layer (fwd + bwd + its heads / losses): {(res[(5, 5)] - res[(5, 1)]) / 4:.3f} ms; per encoder layer: {(res[(5, 5)] - res[(1, 5)]) / 4:.3f} ms")
