"""Golden-vector generator (TEST INFRASTRUCTURE; runs in the BUILD CONTAINER only).

Imports the REAL reference from /root/reference (never copied, never shipped), fills it with the
closed-form weights of oracle/formula.py, runs it in eval() on the seeded synthetic inputs and
writes the outputs to tests/golden/*.npz.  Re-run:  python -m oracle.gen_golden

Two import shims are injected into sys.modules:
  * ``torchvision``  -- import surface only (util/misc.py:33-63, util/box_ops.py:18,
    models/backbone_maskrcnn.py:9-11); no arithmetic on the hot path comes from it.
  * ``deformable_attention`` -- the un-vendored CUDA op (models/deformable_transformer.py:24);
    its stand-in is oracle.poet_ref.MSDeformAttn (upstream's grid_sample formulation), so the
    goldens pin everything AROUND the MSDA core; the core itself is cross-checked against HF
    transformers' independent restatement in a separate process (``--hf``).
"""
from __future__ import annotations

import os
import subprocess
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _install_shims():
    import torch.nn as nn
    from oracle import poet_ref

    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.25.0"
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    misc = types.ModuleType("torchvision.ops.misc")
    misc.interpolate = torch.nn.functional.interpolate
    models = types.ModuleType("torchvision.models")
    det = types.ModuleType("torchvision.models.detection")
    mrcnn = types.ModuleType("torchvision.models.detection.mask_rcnn")
    mrcnn.MaskRCNN = type("MaskRCNN", (nn.Module,), {})
    bbu = types.ModuleType("torchvision.models.detection.backbone_utils")
    bbu.resnet_fpn_backbone = None
    rpn = types.ModuleType("torchvision.models.detection.rpn")
    rpn.AnchorGenerator = None
    rpn.concat_box_prediction_layers = None
    tv.ops, ops.boxes, ops.misc = ops, boxes, misc
    tv.models, models.detection = models, det
    det.mask_rcnn, det.backbone_utils, det.rpn = mrcnn, bbu, rpn
    for m in (tv, ops, boxes, misc, models, det, mrcnn, bbu, rpn):
        sys.modules[m.__name__] = m

    da = types.ModuleType("deformable_attention")
    da.MSDeformAttn = poet_ref.MSDeformAttn
    sys.modules["deformable_attention"] = da
    sys.path.insert(0, REF)


def _ref_model(cfg, feats, bbox_mode="gt", predictions=None, class_mode="specific", rotation_mode="6d", aleatoric=False,
               ref_points_mode="bbox", query_embedding_mode="bbox", position_embedding="sine"):
    """Build the reference's own PoET around a Joiner-like synthetic backbone."""
    import torch.nn as nn
    import torch.nn.functional as F
    from models.deformable_transformer import DeformableTransformer
    from models.pose_estimation_transformer import PoET, SetCriterion
    from models.matcher import PoseMatcher
    from models.position_encoding import PositionEmbeddingSine, PositionEmbeddingLearned
    from util.misc import NestedTensor
    from oracle.poet_ref import build_weight_dict, losses_for

    class Joinerish(nn.Module):
        def __init__(self):
            super().__init__()
            self.strides, self.num_channels = cfg["strides"], cfg["num_channels"]
            # (registered as "1" like the second entry of the reference's Joiner = nn.Sequential(backbone, position_embedding))
            self.add_module("1", PositionEmbeddingLearned(cfg["d_model"] // 2) if position_embedding == "learned"
                            else PositionEmbeddingSine(cfg["d_model"] // 2, normalize=True))

        @property
        def pe(self):
            return self._modules["1"]

        def __getitem__(self, i):
            return self if i == 0 else self.pe

        def forward(self, samples):
            out, pos = [], []
            for f in feats:
                m = F.interpolate(samples.mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
                out.append(NestedTensor(f, m))
            for x in out:
                pos.append(self.pe(x).to(x.tensors.dtype))
            return out, pos, predictions

    tr = DeformableTransformer(d_model=cfg["d_model"], nhead=cfg["nheads"], num_encoder_layers=cfg["enc_layers"],
                               num_decoder_layers=cfg["dec_layers"], dim_feedforward=cfg["d_ffn"],
                               dropout=cfg["dropout"], activation=cfg.get("activation", "relu"), return_intermediate_dec=True,
                               num_feature_levels=cfg["n_levels"], dec_n_points=cfg["n_points"],
                               enc_n_points=cfg["n_points"])
    model = PoET(Joinerish(), tr, num_queries=cfg["num_queries"], num_feature_levels=cfg["n_levels"],
                 n_classes=cfg["n_classes"], bbox_mode=bbox_mode, ref_points_mode=ref_points_mode, query_embedding_mode=query_embedding_mode,
                 rotation_mode=rotation_mode, class_mode=class_mode, aleatoric=aleatoric, aux_loss=True, backbone_type="yolo")
    crit = SetCriterion(PoseMatcher(bbox_mode=bbox_mode, class_mode=class_mode), build_weight_dict(cfg["dec_layers"]),
                        list(losses_for(rotation_mode, aleatoric)))
    return model, crit


INIT_SEED = 4321


def _run_model(name, batch, pad, full, default_init=False, bbox_mode="gt", class_mode="specific", rotation_mode="6d", aleatoric=False,
               ref_points_mode="bbox", query_embedding_mode="bbox", position_embedding="sine"):
    from oracle.formula import CONFIGS, formula_fill, make_inputs, make_samples, checksum
    from util.misc import nested_tensor_from_tensor_list

    cfg = CONFIGS[name]
    feats, sizes, targets = make_inputs(cfg, seed=1234, batch=batch, pad=pad)
    if default_init:
        # the reference's own initialisation under a fixed seed; the oracle / HIP modules consume the RNG in the
        # same order, which is asserted here against the oracle and re-checked in the tests via param checksums
        from oracle import poet_ref
        torch.manual_seed(INIT_SEED)
        model, crit = _ref_model(cfg, feats)
        torch.manual_seed(INIT_SEED)
        omodel, _ = poet_ref.build_poet(cfg, feats)
        osd = omodel.state_dict()
        for k, v in model.state_dict().items():
            assert torch.equal(v, osd[k]), f"default init differs at {k}"
    else:
        model, crit = _ref_model(cfg, feats, bbox_mode=bbox_mode, class_mode=class_mode, rotation_mode=rotation_mode, aleatoric=aleatoric,
                                 ref_points_mode=ref_points_mode, query_embedding_mode=query_embedding_mode,
                                 position_embedding=position_embedding)
        formula_fill(model)
    model.eval()
    crit.eval()
    samples = nested_tensor_from_tensor_list(make_samples(cfg, sizes))

    captured = {}
    hook = model.transformer.encoder.register_forward_hook(lambda m, i, o: captured.__setitem__("memory", o.detach()))
    hook2 = model.transformer.register_forward_hook(lambda m, i, o: captured.__setitem__("hs", o[0].detach()))
    out, n_boxes = model(samples, targets)
    hook.remove(); hook2.remove()
    losses = crit(out, targets, n_boxes)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses if k in crit.weight_dict)
    model.zero_grad()
    total.backward()

    rec = {
        "pred_translation": out["pred_translation"].detach().numpy(),
        "pred_rotation": out["pred_rotation"].detach().numpy(),
        "aux_translation": np.stack([a["pred_translation"].detach().numpy() for a in out["aux_outputs"]])
        if out["aux_outputs"] else np.zeros((0,)),
        "aux_rotation": np.stack([a["pred_rotation"].detach().numpy() for a in out["aux_outputs"]])
        if out["aux_outputs"] else np.zeros((0,)),
        "n_boxes": np.asarray(n_boxes),
        "loss_total": np.asarray(float(total)),
        "loss_names": np.asarray(sorted(losses)),
        "loss_values": np.asarray([float(losses[k]) for k in sorted(losses)]),
    }
    if full:
        rec["memory"] = captured["memory"].numpy()
        rec["hs"] = captured["hs"].numpy()
    else:
        rec["memory_checksum"] = checksum(captured["memory"])
        rec["hs_checksum"] = checksum(captured["hs"])
    names, sums = [], []
    for n, p in model.named_parameters():
        names.append(n)
        sums.append(checksum(p.grad) if p.grad is not None else np.full(9, np.nan))
    rec["grad_names"] = np.asarray(names)
    rec["grad_checksums"] = np.stack(sums)
    if default_init:
        rec["param_names"] = np.asarray([n for n, _ in model.named_parameters()])
        rec["param_checksums"] = np.stack([checksum(p) for _, p in model.named_parameters()])
        # Encoder sampling_offsets at the reference's own init: every query's offsets are the bias alone, the axis / diagonal
        # heads' bias components are exact integers and grid queries sit on pixel centres, so those OUTPUT CHANNELS take a
        # one-sided derivative of the bilinear interpolation (decided by the rounding of the reference's own 2 * loc - 1):
        # checksums of the gradient restricted to the channels whose bias component is NOT an integer pin everything else.
        kn, ks, kc = [], [], []
        params = dict(model.named_parameters())
        for n, p in model.named_parameters():
            if "encoder" in n and "sampling_offsets" in n and p.grad is not None:
                b = params[n.rsplit(".", 1)[0] + ".bias"].detach()
                kink = (b - b.round()).abs() < 1e-3
                if not 0 < int(kink.sum()) < b.numel():
                    continue              # (`tiny`: 4 heads = the four axis directions, EVERY channel of the bias is an integer -- there is
                                          # no kink-free restriction to pin: that fixture carries the whole-tensor checksums only)
                kn.append(n)
                ks.append(checksum(p.grad[~kink]))
                kc.append(int(kink.sum()))
        if kn:
            rec["nokink_names"], rec["nokink_checksums"], rec["kink_channels"] = np.asarray(kn), np.stack(ks), np.asarray(kc)
    tag = f"{name}_b{batch}{'_pad' if pad else ''}{'_init' if default_init else ''}"
    if (bbox_mode, class_mode) != ("gt", "specific"):
        tag += f"_{bbox_mode}_{class_mode}"
    if (ref_points_mode, query_embedding_mode) != ("bbox", "bbox"):
        tag += f"_q{query_embedding_mode}_r{ref_points_mode}"
    if rotation_mode != "6d" or aleatoric:
        tag += f"_{rotation_mode}{'_aleatoric' if aleatoric else ''}"
        if aleatoric:
            rec["aux_translation_aleatoric"] = np.stack([a["pred_translation_aleatoric"].detach().numpy() for a in out["aux_outputs"]])
            rec["pred_translation_aleatoric"] = out["pred_translation_aleatoric"].detach().numpy()
            rec["pred_rotation_aleatoric"] = out["pred_rotation_aleatoric"].detach().numpy()
    if position_embedding != "sine":
        tag += f"_pe{position_embedding}"
    np.savez_compressed(os.path.join(GOLD, f"poet_{tag}.npz"), **rec)
    print("wrote", tag, "loss", float(total))


def _run_inference(name, batch=3):
    """The inference path (bbox_mode='backbone', eval, no targets): queries come from the detector rows."""
    from oracle.formula import CONFIGS, formula_fill, make_inputs, make_predictions, make_samples
    from util.misc import nested_tensor_from_tensor_list

    cfg = CONFIGS[name]
    feats, sizes, _ = make_inputs(cfg, seed=1234, batch=batch, pad=False)
    preds = make_predictions(cfg, seed=77, batch=batch)
    model, _ = _ref_model(cfg, feats, bbox_mode="backbone", predictions=preds)
    formula_fill(model)
    model.eval()
    samples = nested_tensor_from_tensor_list(make_samples(cfg, sizes))
    with torch.no_grad():
        out, n_boxes = model(samples, None)
    rec = {"pred_translation": out["pred_translation"].numpy(), "pred_rotation": out["pred_rotation"].numpy(),
           "pred_boxes": out["pred_boxes"].numpy(), "pred_classes": out["pred_classes"].numpy(),
           "aux_translation": np.stack([a["pred_translation"].numpy() for a in out["aux_outputs"]]) if out["aux_outputs"] else np.zeros((0,)),
           "aux_rotation": np.stack([a["pred_rotation"].numpy() for a in out["aux_outputs"]]) if out["aux_outputs"] else np.zeros((0,)),
           "n_boxes": np.asarray(n_boxes)}
    np.savez_compressed(os.path.join(GOLD, f"poet_{name}_b{batch}_infer.npz"), **rec)
    print("wrote", f"{name}_b{batch}_infer", "n_boxes", list(n_boxes))


def _matcher_cases(seed=5):
    """Inputs for the evaluation-side matcher (bbox_mode='backbone'): per image a set of detector boxes (queries) and
    ground-truth boxes -- jittered copies that overlap (GIoU above and below the 0.5 threshold), wrong-class detections,
    spurious detections, an image without detections and one without targets."""
    rng = np.random.default_rng(seed)
    nq = 12
    imgs = []
    for n_t, n_d, jit, flip in ((5, 7, 0.02, 1), (8, 8, 0.10, 2), (3, 12, 0.25, 0), (0, 4, 0.0, 0), (6, 0, 0.0, 0), (9, 5, 0.05, 3)):
        tb = np.concatenate([rng.uniform(0.2, 0.8, (n_t, 2)), rng.uniform(0.05, 0.3, (n_t, 2))], 1).astype(np.float32)
        tl = rng.integers(1, 6, n_t).astype(np.int64)
        k = min(n_t, n_d)
        perm = rng.permutation(n_t)[:k]
        db = tb[perm] + (rng.standard_normal((k, 4)) * jit * np.array([1, 1, 0.5, 0.5])).astype(np.float32)
        db[:, 2:] = np.abs(db[:, 2:]) + 1e-3
        dl = tl[perm].copy()
        dl[:flip] = (dl[:flip] % 5) + 1                    # wrong class on the first `flip` detections
        extra = n_d - k
        db = np.concatenate([db, np.concatenate([rng.uniform(0.1, 0.9, (extra, 2)), rng.uniform(0.05, 0.3, (extra, 2))], 1)]).astype(np.float32)
        dl = np.concatenate([dl, rng.integers(1, 6, extra)]).astype(np.int64)
        order = rng.permutation(n_d)
        pb = -np.ones((nq, 4), np.float32)
        pc = -np.ones((nq,), np.int64)
        pb[:n_d], pc[:n_d] = db[order], dl[order]
        imgs.append(dict(pred_boxes=pb, pred_classes=pc, n_boxes=n_d, boxes=tb, labels=tl))
    return imgs


def _run_matcher():
    """matcher.py:104-229 in 'backbone' mode (the mode pose_evaluate / bop_evaluate use, engine.py:127,212), both class modes."""
    from models.matcher import PoseMatcher
    imgs = _matcher_cases()
    outputs = {"pred_boxes": torch.from_numpy(np.stack([i["pred_boxes"] for i in imgs])),
               "pred_classes": torch.from_numpy(np.stack([i["pred_classes"] for i in imgs]))}
    targets = [{"boxes": torch.from_numpy(i["boxes"]), "labels": torch.from_numpy(i["labels"])} for i in imgs]
    n_boxes = [i["n_boxes"] for i in imgs]
    rec = {"pred_boxes": outputs["pred_boxes"].numpy(), "pred_classes": outputs["pred_classes"].numpy(), "n_boxes": np.asarray(n_boxes),
           "n_targets": np.asarray([len(i["boxes"]) for i in imgs]),
           "tgt_boxes": np.concatenate([i["boxes"] for i in imgs]), "tgt_labels": np.concatenate([i["labels"] for i in imgs])}
    for cm in ("specific", "agnostic"):
        for thr in (0.5, 0.0):
            res = PoseMatcher(bbox_mode="backbone", class_mode=cm)(outputs, targets, n_boxes, giou_thresh=thr)
            flat = np.concatenate([np.stack([np.full(len(s), b), s.numpy(), t.numpy()], 1).reshape(-1, 3) for b, (s, t) in enumerate(res)]).astype(np.int64)
            rec[f"match_{cm}_{int(thr * 10)}"] = flat
            print("matcher", cm, thr, [len(s) for s, _ in res])
    np.savez_compressed(os.path.join(GOLD, "matcher_backbone.npz"), **rec)


def _small_units():
    from models.position_encoding import PositionEmbeddingSine, BoundingBoxEmbeddingSine
    from models.pose_estimation_transformer import PoET
    from util.misc import NestedTensor

    rng = np.random.default_rng(7)
    # G1: sine position embedding on right/bottom padded masks
    mask = torch.zeros(3, 12, 16, dtype=torch.bool)
    mask[1, 9:, :] = True
    mask[1, :, 13:] = True
    mask[2, :, 10:] = True
    pe = PositionEmbeddingSine(128, normalize=True)(NestedTensor(torch.zeros(3, 1, 12, 16), mask))
    # G2: bbox embedding incl. the dummy box
    boxes = torch.from_numpy(np.concatenate([rng.uniform(0, 1, (7, 4)), -np.ones((1, 4))]).astype(np.float32))
    be = BoundingBoxEmbeddingSine(num_pos_feats=256 / 8)(boxes)
    # G10: 6D -> R on random and near-degenerate inputs
    r6 = rng.standard_normal((2, 6, 6)).astype(np.float32)
    r6[0, 0, 3:] = r6[0, 0, :3] * 2.0 + 1e-4          # m2 almost parallel to m1
    r6[0, 1, :3] *= 1e-6                               # tiny m1
    rm = PoET.rotation_6d_to_matrix(None, torch.from_numpy(r6))
    np.savez_compressed(os.path.join(GOLD, "units.npz"), pe_mask=mask.numpy(), pe_out=pe.numpy(),
                        bbox_in=boxes.numpy(), bbox_out=be.numpy(), rot6d_in=r6, rot6d_out=rm.numpy())
    print("wrote units")


def _hf_msda():
    """Separate process (no torchvision stub on the path): HF transformers' own pure-PyTorch MSDA."""
    from transformers.models.deformable_detr.modeling_deformable_detr import MultiScaleDeformableAttention
    rng = np.random.default_rng(11)
    shapes = [(6, 8), (3, 4)]
    n, m, d, lq, p = 2, 4, 16, 9, 4
    s = sum(h * w for h, w in shapes)
    value = torch.from_numpy(rng.standard_normal((n, s, m, d)).astype(np.float32)).requires_grad_()
    loc = torch.from_numpy(rng.uniform(-0.3, 1.3, (n, lq, m, len(shapes), p, 2)).astype(np.float32)).requires_grad_()
    w = torch.softmax(torch.from_numpy(rng.standard_normal((n, lq, m, len(shapes) * p)).astype(np.float32)), -1)
    w = w.view(n, lq, m, len(shapes), p).detach().requires_grad_()
    core = MultiScaleDeformableAttention()
    out = core(value, torch.tensor(shapes), shapes, None, loc, w, 64)
    gout = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
    (out * gout).sum().backward()
    np.savez_compressed(os.path.join(GOLD, "msda_core_hf.npz"), shapes=np.asarray(shapes), value=value.detach().numpy(),
                        loc=loc.detach().numpy(), attn=w.detach().numpy(), out=out.detach().numpy(), grad_out=gout.numpy(),
                        d_value=value.grad.numpy(), d_loc=loc.grad.numpy(), d_attn=w.grad.numpy())
    print("wrote msda_core_hf")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--hf" in sys.argv:
        _hf_msda()
        return
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden needs /root/reference (build container only)")
    _install_shims()
    if "--matcher" in sys.argv:               # the evaluation-side matcher (bbox_mode='backbone')
        _run_matcher()
        return
    if "--infer" in sys.argv:                 # only the inference goldens (added after the training ones)
        _run_inference("tiny")
        _run_inference("cfg0")
        return
    if "--rot" in sys.argv:                   # only the quaternion / aleatoric goldens
        for rm, al in (("quat", False), ("silho_quat", False), ("6d", True)):
            _run_model("tiny", 2, True, True, rotation_mode=rm, aleatoric=al)
        return
    if "--big" in sys.argv:                   # BASELINE.json configs[3] / configs[4] at bs 1: outputs + checksums only
        _run_model("lmo", 1, False, False)
        _run_model("lmo", 2, True, False)
        _run_model("hires", 1, False, False)
        return
    if "--pelearned" in sys.argv:             # --position_embedding learned (main.py:67, position_encoding.py:87-112)
        _run_model("tiny", 2, True, True, position_embedding="learned")
        return
    if "--learned" in sys.argv:               # --query_embedding learned / --reference_points learned (main.py:76-79)
        _run_model("tiny", 2, True, True, query_embedding_mode="learned")
        _run_model("tiny", 2, True, True, query_embedding_mode="learned", ref_points_mode="learned")
        return
    if "--round3" in sys.argv:                # the reference's own init at every full-size config; LM-O at >= 4096 token rows (bs 3)
        _run_model("ycbv", 1, False, False, default_init=True)
        _run_model("lmo", 1, False, False, default_init=True)
        _run_model("hires", 1, False, False, default_init=True)
        _run_model("lmo", 3, True, False)
        return
    if "--queries" in sys.argv:               # --num_queries 100 (main.py:98): more object queries than one wave holds
        _run_model("tiny100", 2, True, True)
        return
    if "--gelu" in sys.argv:                  # activation="gelu", 8 heads of dim 8, 130 queries (the generic self-attention kernels)
        _run_model("tinyg", 2, True, True)
        return
    if "--levels" in sys.argv:                # 5 feature levels (two chained extra levels) x 3 sampling points
        _run_model("tiny5", 2, True, True)
        return
    if "--modes" in sys.argv:                 # only the jitter / class-agnostic goldens
        _run_model("tiny", 2, True, True, bbox_mode="jitter", class_mode="specific")
        _run_model("tiny", 2, True, True, bbox_mode="gt", class_mode="agnostic")
        return
    _small_units()
    _run_model("tiny", 2, True, True)
    _run_model("tiny", 2, False, True)
    _run_model("cfg0", 2, False, True)
    _run_model("cfg0", 2, True, False)
    _run_model("ycbv", 1, False, False)
    _run_model("ycbv", 1, False, False, default_init=True)
    _run_model("lmo", 1, False, False)
    _run_model("lmo", 2, True, False)
    _run_model("hires", 1, False, False)
    _run_model("lmo", 1, False, False, default_init=True)
    _run_model("hires", 1, False, False, default_init=True)
    _run_model("lmo", 3, True, False)
    _run_model("tiny", 2, True, True, default_init=True)
    _run_model("tiny", 2, True, True, query_embedding_mode="learned")
    _run_model("tiny", 2, True, True, query_embedding_mode="learned", ref_points_mode="learned")
    _run_model("tiny", 2, True, True, position_embedding="learned")
    _run_model("tiny5", 2, True, True)
    _run_model("tiny100", 2, True, True)
    _run_model("tinyg", 2, True, True)
    _run_inference("tiny")
    _run_inference("cfg0")
    _run_matcher()
    _run_model("tiny", 2, True, True, bbox_mode="jitter", class_mode="specific")
    _run_model("tiny", 2, True, True, bbox_mode="gt", class_mode="agnostic")
    for rm, al in (("quat", False), ("silho_quat", False), ("6d", True)):
        _run_model("tiny", 2, True, True, rotation_mode=rm, aleatoric=al)
    subprocess.check_call([sys.executable, "-m", "oracle.gen_golden", "--hf"], cwd=ROOT)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
