"""Shared helper (tests only): build the HIP-backed PoET on a named config with formula weights."""
import torch

import poet_amd
from poet_amd.synthetic import SyntheticBackbone, image_mask
from oracle.formula import CONFIGS, formula_fill, make_inputs


def build_product(name, batch, pad, precision="fp32", seed=1234, dropout=None, feat_dtype=None, default_init=False,
                  bbox_mode="gt", predictions=None, class_mode="specific", rotation_mode="6d", aleatoric=False,
                  ref_points_mode="bbox", query_embedding_mode="bbox", position_embedding="sine", replicate=None, perturb=0):
    """replicate=B: the batch is B copies of image 0 of the (batch, pad) inputs -- features, size and targets alike.  Images of
    a batch are independent and every loss term is a sum over matched objects divided by the batch's object count, so every
    per-image output, every loss value and every parameter gradient equals the single-image run's: a golden of the real
    reference at batch 1 pins a launch of B times the token rows."""
    if not isinstance(precision, str):
        precision = "fp32" if precision == torch.float32 else "bf16"
    cfg = CONFIGS[name]
    feats, sizes, targets = make_inputs(cfg, seed=seed, batch=batch, pad=pad)
    if perturb:
        # member `perturb` of a NOISE ENSEMBLE of the same problem (tests/test_model_gpu.py, the bf16 gradient gate): every feature
        # value moved by a random-sign 2^-10 relative step -- a quarter of a bf16 ulp: it re-rolls the rounding decisions of a 16-bit
        # policy from the first operand on while the exact gradient moves by ~1e-3 of itself
        gen = torch.Generator().manual_seed(9000 + int(perturb))
        feats = [f * (1.0 + (2.0 ** -10) * (torch.randint(0, 2, f.shape, generator=gen).to(f.dtype) * 2.0 - 1.0)) for f in feats]
    if replicate:
        feats = [f[:1].repeat(replicate, 1, 1, 1).contiguous() for f in feats]
        sizes = [sizes[0]] * replicate
        targets = [{k: v.clone() for k, v in targets[0].items()} for _ in range(replicate)]
    fd = feat_dtype or torch.float32
    gfeats = [f.cuda().to(fd) for f in feats]
    bb = SyntheticBackbone(gfeats, cfg["strides"], cfg["num_channels"], cfg["d_model"] // 2, predictions=predictions,
                           position_embedding=position_embedding)
    p = cfg["dropout"] if dropout is None else dropout
    if default_init:
        torch.manual_seed(4321)          # oracle/gen_golden.py INIT_SEED: the reference's own init order
    tr = poet_amd.DeformableTransformer(d_model=cfg["d_model"], nhead=cfg["nheads"], num_encoder_layers=cfg["enc_layers"],
                                        num_decoder_layers=cfg["dec_layers"], dim_feedforward=cfg["d_ffn"], dropout=p,
                                        activation=cfg.get("activation", "relu"), return_intermediate_dec=True,
                                        num_feature_levels=cfg["n_levels"], dec_n_points=cfg["n_points"],
                                        enc_n_points=cfg["n_points"])
    tr.set_precision(precision)
    model = poet_amd.PoET(bb, tr, num_queries=cfg["num_queries"], num_feature_levels=cfg["n_levels"],
                          n_classes=cfg["n_classes"], bbox_mode=bbox_mode, class_mode=class_mode, aux_loss=True,
                          rotation_mode=rotation_mode, aleatoric=aleatoric, ref_points_mode=ref_points_mode,
                          query_embedding_mode=query_embedding_mode)
    if not default_init:
        formula_fill(model)
    model = model.cuda()
    crit = poet_amd.SetCriterion(poet_amd.PoseMatcher(bbox_mode="jitter" if bbox_mode == "jitter" else "gt"),
                                 poet_amd.build_weight_dict(cfg["dec_layers"]), poet_amd.losses_for(rotation_mode, aleatoric))
    samples = poet_amd.NestedTensor(None, image_mask(sizes, "cuda"))
    gt = [{k: v.cuda() for k, v in t.items()} for t in targets]
    return dict(cfg=cfg, model=model, crit=crit, samples=samples, targets=gt, cpu_targets=targets, feats=feats, sizes=sizes)
