"""Data-parallel plumbing on CPU: two `gloo` processes (world_size 2).  The HIP kernels cannot run here, so the
gradient arena is filled with known per-rank values; what is checked is what the N>1 path adds on top of the
single-GPU step: bucket layout, bucketed sum all-reduce in backward-completion order, the 1/world average folded
into the optimiser scale, the logging all-reduce (util/misc.py:168-195) and the sampler sharding
(data_utils/samplers.py:48-66)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Toy(nn.Module):
    """Parameter names chosen to land in every arena bucket and LR group."""

    def __init__(self):
        super().__init__()
        self.translation_head = nn.Linear(8, 3)
        self.rotation_head = nn.Linear(8, 6)
        self.transformer = nn.Module()
        self.transformer.decoder = nn.Linear(8, 8)
        self.transformer.encoder = nn.Module()
        self.transformer.encoder.sampling_offsets = nn.Linear(8, 4)
        self.transformer.encoder.linear1 = nn.Linear(8, 16)
        self.transformer.level_embed = nn.Parameter(torch.zeros(2, 8))
        self.transformer.reference_points = nn.Linear(8, 2)      # never gets a gradient: stays out of the arena
        self.input_proj = nn.Linear(5, 8)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from poet_amd.engine import BucketReducer, ParamArena, reduce_dict, set_reducer, announce, shard_indices
        torch.manual_seed(0)
        model = _Toy()
        arena = ParamArena(model)
        names = [n for n, _, _ in arena.entries]
        assert not any("reference_points" in n for n in names)
        # buckets are contiguous, ordered by backward completion, and cover the arena exactly
        tags = [b[0] for b in arena.buckets]
        from poet_amd.engine import segment_tags          # the graphed trainer cuts backward at exactly these buckets
        assert segment_tags(arena) == tags
        assert tags == ["0_heads", "1_decoder", "2_encoder_99", "3_input_proj"], tags      # (the toy encoder has no `layers.N`)
        assert arena.buckets[0][1] == 0 and arena.buckets[-1][2] == arena.total
        for (_, _, e), (_, s, _) in zip(arena.buckets[:-1], arena.buckets[1:]):
            assert e == s
        # the 0.1x LR group (main.py:41) is a per-64-element multiplier table over ONE range: 0.1 exactly on sampling_offsets
        assert len(arena.groups) == 1 and arena.groups[0][:2] == (0, arena.total) and len(arena.lr_scale) == arena.total // 64
        for n, p, o in arena.entries:
            blk = arena.lr_scale[o // 64: (o + p.numel() + 63) // 64]
            assert bool((blk == (0.1 if "sampling_offsets" in n else 1.0)).all()), n
        # parameters are views of the flat buffer
        p0 = model.input_proj.weight
        assert p0.data_ptr() >= arena.flat.data_ptr() and p0._grad_view.data_ptr() >= arena.grad.data_ptr()

        red = BucketReducer(arena)
        assert red.world == world and arena.world == world
        set_reducer(red)
        arena.zero_grad()
        for n, p, _ in arena.entries:                      # "kernels" write rank-dependent gradients in place
            p._grad_view.fill_(float(rank + 1))
        for tag in ["0_heads", "1_decoder", "2_encoder_99", "3_input_proj"]:   # order of the autograd nodes' announce()
            announce(tag)
        # the tail buckets (level_embed, input_proj) are not reduced when announced: finish() sends them as ONE collective
        rng = {t: (a, b) for t, a, b in arena.buckets}
        assert red.collectives == [rng["0_heads"], rng["1_decoder"]], red.collectives
        red.finish()                                       # (also joins the comm stream)
        assert red.collectives == [rng["0_heads"], rng["1_decoder"], (rng["2_encoder_99"][0], rng["3_input_proj"][1])], red.collectives
        set_reducer(None)
        expect = float(sum(r + 1 for r in range(world)))
        for n, p, _ in arena.entries:
            assert torch.all(p._grad_view == expect), n
        # bf16 transport (POET_DP_GRAD_DTYPE=bf16 / grad_dtype): same sums wherever bf16 holds them exactly, the arena stays fp32,
        # and nothing announced + finish() = the whole arena in one collective
        red16 = BucketReducer(arena, grad_dtype=torch.bfloat16)
        for n, p, _ in arena.entries:
            p._grad_view.fill_(0.5 * (rank + 1))
        set_reducer(red16); announce("1_decoder"); red16.finish(); set_reducer(None)
        assert red16.collectives == [rng["1_decoder"], rng["0_heads"], (rng["2_encoder_99"][0], rng["3_input_proj"][1])], red16.collectives
        assert arena.grad.dtype == torch.float32 and all(bool(torch.all(p._grad_view == 0.5 * expect)) for _, p, _ in arena.entries)
        arena.grad.fill_(1.0 + 2.0 ** -10 * (rank + 1))    # not representable in bf16: the transport rounds, fp32 would not
        red16.finish()
        assert red16.collectives == [(0, arena.total)]
        assert torch.all(arena.grad == float(world))
        # a second step must reduce again (done-set is cleared)
        for n, p, _ in arena.entries:
            p._grad_view.fill_(1.0)
        set_reducer(red); announce("0_heads"); red.finish(); set_reducer(None)
        assert torch.all(arena.grad[arena.buckets[0][1]: arena.buckets[0][2]][:3] == float(world))

        # logging all-reduce keeps the reference's semantics (sorted keys, averaged)
        out = reduce_dict({"loss_b": torch.tensor(float(rank)), "loss_a": torch.tensor(2.0 * rank)})
        assert float(out["loss_b"]) == pytest.approx(sum(range(world)) / world)
        assert float(out["loss_a"]) == pytest.approx(2.0 * sum(range(world)) / world)

        # sampler sharding == the reference's DistributedSampler (contiguous slice of the epoch-seeded permutation;
        # note: torch's own sampler strides instead -- the reference's is the contract here)
        g = torch.Generator(); g.manual_seed(3)
        perm = torch.randperm(37, generator=g).tolist()
        per = 19
        perm += perm[: per * world - 37]
        mine = shard_indices(37, rank, world, epoch=3)
        assert mine == perm[per * rank: per * (rank + 1)] and len(mine) == per
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        assert sorted(set(sum(gathered, []))) == list(range(37))          # every sample is covered
        q.put((rank, "ok"))
    except Exception as e:                                  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_input_side_batching_and_prefetch_protocol():
    """util/misc.py:326-343 (zero padding + mask) against the oracle's restatement, and the prefetcher's iteration protocol
    (data_utils/data_prefetcher.py: batches in order, (None, None) at the end) on the CPU."""
    import poet_amd
    from oracle import poet_ref
    g = torch.Generator().manual_seed(0)
    ims = [torch.randn(3, 5, 7, generator=g), torch.randn(3, 4, 9, generator=g), torch.randn(3, 6, 2, generator=g)]
    a, b = poet_amd.nested_tensor_from_tensor_list(ims), poet_ref.nested_from_list(ims)
    assert torch.equal(a.tensors, b.tensors) and torch.equal(a.mask, b.mask)
    assert a.tensors.shape == (3, 3, 6, 9) and bool(a.mask[0, 5:, :].all()) and not bool(a.mask[0, :5, :7].any())
    batches = [(poet_amd.nested_tensor_from_tensor_list([im]), [{"boxes": torch.full((2, 4), float(i))}]) for i, im in enumerate(ims)]
    for prefetch in (True, False):
        pf = poet_amd.DataPrefetcher(batches, "cpu", prefetch=prefetch)
        seen = []
        while True:
            s, t = pf.next()
            if s is None:
                break
            seen.append(float(t[0]["boxes"][0, 0]))
        assert seen == [0.0, 1.0, 2.0] and pf.next() == (None, None)


def test_arena_refuses_trainable_parameters_outside_the_hot_path():
    """A trainable backbone (the reference trains it at lr_backbone, main.py:253-271) must not land in the flat arena, where it
    would be weight-decayed at the transformer's learning rate without ever receiving a gradient."""
    from poet_amd.engine import ParamArena
    m = _Toy()
    m.backbone = nn.Linear(4, 4)
    with pytest.raises(ValueError, match="outside the hot path"):
        ParamArena(m)
    for p in m.backbone.parameters():
        p.requires_grad_(False)
    ParamArena(m)                                            # frozen: fine


def test_arena_lr_schedule_and_state_dict_roundtrip():
    """StepLR semantics on the device-side LR table (main.py:277-278) and optimizer-state checkpointing (main.py:293-301):
    the 0.1x sampling_offsets ratio survives set_lr; moments / step / lr round-trip through state_dict; loading weights into
    the model refreshes the bf16 operand shadow."""
    from poet_amd.engine import ParamArena
    torch.manual_seed(1)
    m = _Toy()
    a = ParamArena(m, lr=2e-4)
    base = a.lr_scale.clone()
    a.step_lr(epoch=7, lr_drop=5)                             # one decay step: lr = 2e-5
    assert abs(a.lr - 2e-5) < 1e-12
    assert torch.allclose(a.lr_scale, base * 0.1)
    so = [o for n, p, o in a.entries if "sampling_offsets.weight" in n][0]
    lin = [o for n, p, o in a.entries if "linear1.weight" in n][0]
    assert abs(float(a.lr_scale[so // 64]) / float(a.lr_scale[lin // 64]) - 0.1) < 1e-6
    a.m.uniform_(); a.v.uniform_(); a.step_count = 12
    sd = a.state_dict()
    m2 = _Toy()
    b = ParamArena(m2, lr=2e-4)
    b.load_state_dict(sd)
    assert torch.equal(b.m, a.m) and torch.equal(b.v, a.v) and b.step_count == 12 and int(b.step_word) == 12
    assert abs(b.lr - 2e-5) < 1e-12 and torch.allclose(b.lr_scale, a.lr_scale)
    # load_state_dict on the MODEL copies into the arena views in place; the bf16 shadow must follow
    m2.load_state_dict(m.state_dict())
    for (n, p), (_, q) in zip(m2.named_parameters(), m.named_parameters()):
        assert torch.equal(p, q)
        if hasattr(p, "_bf16"):
            assert torch.equal(p._bf16, q.detach().to(torch.bfloat16)), n


def test_encoder_layers_get_their_own_buckets_in_backward_order():
    """One gradient bucket per encoder layer, laid out in backward-completion order (last layer first), then level_embed /
    the rest of the encoder, then input_proj -- the reference's DDP bucket order (main.py:280-283): a layer's all-reduce can
    start as soon as ITS backward program ends.  functional.enc_bucket_tag names the same buckets."""
    from poet_amd.engine import ParamArena
    from poet_amd.functional import enc_bucket_tag
    m = _Toy()
    m.transformer.encoder.layers = nn.ModuleList([nn.Linear(8, 8) for _ in range(3)])
    a = ParamArena(m)
    tags = [b[0] for b in a.buckets]
    assert tags == ["0_heads", "1_decoder", "2_encoder_00", "2_encoder_01", "2_encoder_02", "2_encoder_99", "3_input_proj"], tags
    assert [enc_bucket_tag(3, i) for i in (2, 1, 0)] == tags[2:5]
    where = {n: o for n, _, o in a.entries}
    rng = {t: (s0, e0) for t, s0, e0 in a.buckets}
    for i in range(3):
        s0, e0 = rng[enc_bucket_tag(3, i)]
        assert s0 <= where[f"transformer.encoder.layers.{i}.weight"] < e0
    s0, e0 = rng["2_encoder_99"]
    assert s0 <= where["transformer.level_embed"] < e0
    assert a.buckets[0][1] == 0 and a.buckets[-1][2] == a.total and all(x[2] == y[1] for x, y in zip(a.buckets[:-1], a.buckets[1:]))
