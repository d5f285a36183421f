import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, poet_amd
from oracle.formula import CONFIGS
cfg = CONFIGS["ycbv"]; batch = 16; device = torch.device("cuda:0")
torch.manual_seed(1234); poet_amd.manual_seed(1234)
model = None
feats, targets = bench.synth_batch(cfg, batch, 1234, device)
model, crit = bench.build_model(cfg, feats, "bf16", device)
model.train()
trainer = poet_amd.GraphedTrainer(model, crit, lr=2e-4, weight_decay=1e-4, max_norm=0.1, warm=2)
ih, iw = cfg["image_hw"]
samples = poet_amd.NestedTensor(None, torch.zeros((batch, ih, iw), dtype=torch.bool, device=device))
alt = [bench.synth_batch(cfg, batch, 2000 + i, device)[1] for i in range(8)]       # 8 different target sets
for _ in range(6): trainer.step(samples, targets)
torch.cuda.synchronize()
import resource
for block in range(20):
    t0 = time.perf_counter()
    for i in range(50):
        total, _ = trainer.step(samples, alt[(block * 50 + i) % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    print(f"block {block}: {dt:.2f} ms/step  loss {float(total):.4f}  cuda_alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  host_rss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/2**20:.2f} GiB", flush=True)
