"""Shared helper: run the CPU oracle on a named config (tests only)."""
import numpy as np
import torch

from oracle import poet_ref
from oracle.formula import CONFIGS, formula_fill, make_inputs, make_samples


INIT_SEED = 4321      # oracle/gen_golden.py: seed of the reference's own default initialisation


def run_oracle(name, batch, pad, seed=1234, backward=True, train=False, default_init=False, bbox_mode="gt", class_mode="specific",
               rotation_mode="6d", aleatoric=False, ref_points_mode="bbox", query_embedding_mode="bbox", position_embedding="sine"):
    cfg = CONFIGS[name]
    feats, sizes, targets = make_inputs(cfg, seed=seed, batch=batch, pad=pad)
    if default_init:
        torch.manual_seed(INIT_SEED)
    model, crit = poet_ref.build_poet(cfg, feats, bbox_mode=bbox_mode, class_mode=class_mode, rotation_mode=rotation_mode, aleatoric=aleatoric,
                                      ref_points_mode=ref_points_mode, query_embedding_mode=query_embedding_mode,
                                      position_embedding=position_embedding)
    if not default_init:
        formula_fill(model)
    model.train(train)
    samples = poet_ref.nested_from_list(make_samples(cfg, sizes))
    cap = {}
    h1 = model.transformer.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("memory", o.detach()))
    h2 = model.transformer.register_forward_hook(lambda m, i, o: cap.__setitem__("hs", o[0].detach()))
    out, n_boxes = model(samples, targets)
    h1.remove(); h2.remove()
    losses = crit(out, targets, n_boxes)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    if backward:
        model.zero_grad()
        total.backward()
    return dict(cfg=cfg, model=model, crit=crit, out=out, n_boxes=n_boxes, losses=losses, total=total,
                memory=cap["memory"], hs=cap["hs"], feats=feats, sizes=sizes, targets=targets, samples=samples)


def run_oracle_inference(name, batch=3):
    """The inference path (bbox_mode='backbone', eval(), no targets) on the seeded detector rows of formula.make_predictions."""
    from oracle.formula import make_predictions
    cfg = CONFIGS[name]
    feats, sizes, _ = make_inputs(cfg, seed=1234, batch=batch, pad=False)
    preds = make_predictions(cfg, seed=77, batch=batch)
    model, _ = poet_ref.build_poet(cfg, feats, bbox_mode="backbone", predictions=preds)
    formula_fill(model)
    model.eval()
    samples = poet_ref.nested_from_list(make_samples(cfg, sizes))
    with torch.no_grad():
        out, n_boxes = model(samples, None)
    return dict(cfg=cfg, model=model, out=out, n_boxes=n_boxes, feats=feats, sizes=sizes, preds=preds, samples=samples)
