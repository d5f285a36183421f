"""Helper for the data-parallel GPU tests: one rank.  argv: rank world port out.npz graph|eager [backend=gloo] [arith]
("arith": fp32 policy, the rank's own fixed batch for 2 steps -- the run test_data_parallel_arithmetic_vs_single_process compares
with ONE process stepping both ranks' batches)
(gloo: several ranks share cuda:0; nccl: RCCL, one rank per GPU -- world 1 with POET_FORCE_COLLECTIVES=1).  Not a test module."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    graphed = sys.argv[5] in ("graph", "graph1")
    # "graph1": ONE backward graph + ONE all-reduce of the whole gradient arena (the default at world > 1);
    # "graph": one backward segment + one all-reduce per bucket on the comm stream (POET_DP_SINGLE_COLLECTIVE=0)
    if graphed:
        os.environ["POET_DP_SINGLE_COLLECTIVE"] = "1" if sys.argv[5] == "graph1" else "0"
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    arith = len(sys.argv) > 7 and sys.argv[7] == "arith"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import poet_amd
    from oracle.formula import CONFIGS, make_inputs
    from tests.product_runner import build_product
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    r = build_product("tiny", 2, True, "fp32" if arith else "bf16", dropout=0.0, seed=1234 + rank)     # different data, same formula weights
    r["model"].train()
    with torch.no_grad():                                                        # rank 1 starts perturbed: the broadcast must fix it
        if rank:
            for p in r["model"].parameters():
                p.add_(0.01)
    cls = poet_amd.GraphedTrainer if graphed else poet_amd.Trainer
    kw = dict(warm=1) if graphed else {}
    tr = cls(r["model"], r["crit"], lr=2e-4, weight_decay=1e-4, max_norm=0.1, **kw)
    if graphed:
        assert tr.world == world and tr.segment_backward and tr.reducer is not None
    cfg = CONFIGS["tiny"]
    losses = []
    for step in range(2 if arith else 4):
        if arith:
            gt = r["targets"]
        else:
            _, _, targets = make_inputs(cfg, seed=100 + 10 * rank + step, batch=2, pad=True)
            gt = [{k: (v.cuda() if k.startswith("relative") else v) for k, v in t.items()} for t in targets]
        total, _ = tr.step(r["samples"], gt)
        losses.append(float(total))
    torch.cuda.synchronize()
    if graphed:
        n_enc = len(r["model"].transformer.encoder.layers)          # one backward segment per gradient bucket: heads, decoder, encoder layers, input_proj
        if sys.argv[5] == "graph1":
            assert tr.segs is not None and len(tr.segs) == 1 and tr.seg_tags == [[]]
            assert tr.reducer.collectives == [(tr.arena.buckets[0][1], tr.arena.buckets[-1][2])], tr.reducer.collectives
        else:
            assert tr.segs is not None and len(tr.segs) == len(tr.seg_tags) == 3 + n_enc
            assert [t for tags in tr.seg_tags for t in tags] == [b[0] for b in tr.arena.buckets]
    assert tr.reducer is not None and tr.reducer.active
    flat = torch.cat([p.detach().float().flatten() for p in r["model"].parameters()]).cpu().numpy()
    # the clip norm the optimiser saw in the last step: |mean over ranks of the gradient| (AdamW is invariant to the gradient's scale,
    # so a wrong 1 / world fold is invisible in the parameters -- it shows HERE)
    np.savez(out, flat=flat, losses=np.array(losses), gnorm=float(tr.arena.grad_norm()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
