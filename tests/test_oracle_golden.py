"""The CPU oracle (oracle/poet_ref.py) must reproduce the imported reference's outputs
(tests/golden/*.npz, written by oracle/gen_golden.py from /root/reference itself)."""
import os

import numpy as np
import pytest
import torch

from oracle import poet_ref, msda_explicit
from oracle.formula import checksum
from tests.oracle_runner import run_oracle

ATOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_position_embedding_sine(golden_dir):
    g = _load(golden_dir, "units.npz")
    mask = torch.from_numpy(g["pe_mask"])
    pe = poet_ref.PositionEmbeddingSine(128, normalize=True)(poet_ref.NestedTensor(torch.zeros(3, 1, 12, 16), mask))
    np.testing.assert_allclose(pe.numpy(), g["pe_out"], atol=1e-6, rtol=0)


def test_bbox_embedding_sine(golden_dir):
    g = _load(golden_dir, "units.npz")
    be = poet_ref.BoundingBoxEmbeddingSine(256 / 8)(torch.from_numpy(g["bbox_in"]))
    np.testing.assert_allclose(be.numpy(), g["bbox_out"], atol=1e-6, rtol=0)


def test_rotation_6d(golden_dir):
    g = _load(golden_dir, "units.npz")
    r = poet_ref.rotation_6d_to_matrix(torch.from_numpy(g["rot6d_in"]))
    np.testing.assert_allclose(r.numpy(), g["rot6d_out"], atol=1e-6, rtol=0)


def test_msda_core_vs_hf(golden_dir):
    """grid_sample restatement and float64 closed form vs HF transformers' independent restatement."""
    g = _load(golden_dir, "msda_core_hf.npz")
    shapes = [tuple(int(x) for x in r) for r in g["shapes"]]
    value = torch.from_numpy(g["value"]).requires_grad_()
    loc = torch.from_numpy(g["loc"]).requires_grad_()
    attn = torch.from_numpy(g["attn"]).requires_grad_()
    out = poet_ref.msda_core(value, shapes, loc, attn)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=1e-6)
    (out * torch.from_numpy(g["grad_out"])).sum().backward()
    np.testing.assert_allclose(value.grad.numpy(), g["d_value"], atol=1e-5)
    np.testing.assert_allclose(loc.grad.numpy(), g["d_loc"], atol=2e-4)
    np.testing.assert_allclose(attn.grad.numpy(), g["d_attn"], atol=1e-5)
    # closed form
    o2 = msda_explicit.msda_forward(g["value"], shapes, g["loc"], g["attn"])
    np.testing.assert_allclose(o2, g["out"], atol=2e-6)
    dv, dl, da = msda_explicit.msda_backward(g["value"], shapes, g["loc"], g["attn"], g["grad_out"])
    np.testing.assert_allclose(dv, g["d_value"], atol=1e-5)
    np.testing.assert_allclose(dl, g["d_loc"], atol=2e-4)
    np.testing.assert_allclose(da, g["d_attn"], atol=1e-5)


@pytest.mark.parametrize("name,batch,pad,full,init", [("tiny", 2, True, True, False), ("tiny", 2, False, True, False),
                                                      ("cfg0", 2, False, True, False), ("cfg0", 2, True, False, False),
                                                      ("tiny", 2, True, True, True),
                                                      # 5 feature levels (two chained 3x3-s2 extra levels) x 3 sampling points
                                                      ("tiny5", 2, True, True, False),
                                                      ("tiny100", 2, True, True, False),
                                                      # activation="gelu", 8 heads of dim 8, 130 queries
                                                      ("tinyg", 2, True, True, False),
                                                      # BASELINE.json configs[3] (LM-O geometry) and configs[4] (1280x960, 6/6, Q=50)
                                                      ("lmo", 1, False, False, False), ("lmo", 2, True, False, False),
                                                      ("hires", 1, False, False, False),
                                                      # round 3: the reference's own init at the full-size configs; LM-O at bs 3 (4800 rows)
                                                      ("ycbv", 1, False, False, True), ("lmo", 1, False, False, True),
                                                      ("hires", 1, False, False, True), ("lmo", 3, True, False, False)])
def test_poet_vs_reference(golden_dir, name, batch, pad, full, init):
    g = _load(golden_dir, f"poet_{name}_b{batch}{'_pad' if pad else ''}{'_init' if init else ''}.npz")
    r = run_oracle(name, batch, pad, default_init=init)
    if init:        # the oracle consumed the RNG exactly like the reference's constructors did
        for (n, p), ref_sum in zip(r["model"].named_parameters(), g["param_checksums"]):
            np.testing.assert_allclose(checksum(p), ref_sum, atol=0, rtol=0, err_msg=n)
    np.testing.assert_allclose(r["out"]["pred_translation"].detach().numpy(), g["pred_translation"], atol=ATOL)
    np.testing.assert_allclose(r["out"]["pred_rotation"].detach().numpy(), g["pred_rotation"], atol=ATOL)
    if g["aux_translation"].size:
        aux_t = np.stack([a["pred_translation"].detach().numpy() for a in r["out"]["aux_outputs"]])
        aux_r = np.stack([a["pred_rotation"].detach().numpy() for a in r["out"]["aux_outputs"]])
        np.testing.assert_allclose(aux_t, g["aux_translation"], atol=ATOL)
        np.testing.assert_allclose(aux_r, g["aux_rotation"], atol=ATOL)
    assert list(g["n_boxes"]) == list(r["n_boxes"])
    names = sorted(r["losses"])
    assert names == [str(x) for x in g["loss_names"]]
    np.testing.assert_allclose([float(r["losses"][k]) for k in names], g["loss_values"], rtol=1e-5, atol=1e-6)
    if full:
        np.testing.assert_allclose(r["memory"].numpy(), g["memory"], atol=ATOL)
        np.testing.assert_allclose(r["hs"].numpy(), g["hs"], atol=ATOL)
    else:
        np.testing.assert_allclose(checksum(r["memory"]), g["memory_checksum"], rtol=1e-5, atol=1e-5)
    grads = dict(r["model"].named_parameters())
    for n, ref_sum in zip(g["grad_names"], g["grad_checksums"]):
        p = grads[str(n)]
        if np.isnan(ref_sum).all():
            assert p.grad is None, n
            continue
        got = checksum(p.grad)
        scale = max(1.0, abs(ref_sum[0]))
        np.testing.assert_allclose(got, ref_sum, atol=2e-4 * scale, err_msg=str(n))
    if "nokink_names" in g.files:       # gradient of the encoder's sampling_offsets restricted to the channels without a kink
        for n, ref_sum, nk in zip(g["nokink_names"], g["nokink_checksums"], g["kink_channels"]):
            b = grads[str(n).rsplit(".", 1)[0] + ".bias"].detach()
            kink = (b - b.round()).abs() < 1e-3
            assert int(kink.sum()) == int(nk) and 0 < int(nk) < b.numel(), (n, int(kink.sum()), int(nk))
            np.testing.assert_allclose(checksum(grads[str(n)].grad[~kink]), ref_sum, atol=2e-4 * max(1.0, abs(ref_sum[0])), err_msg=str(n))


def test_state_dict_keys_match_reference(golden_dir):
    """The compatibility contract: parameter names equal the reference's."""
    g = _load(golden_dir, "poet_tiny_b2.npz")
    r = run_oracle("tiny", 2, False, backward=False)
    assert [str(x) for x in g["grad_names"]] == [n for n, _ in r["model"].named_parameters()]


@pytest.mark.parametrize("name", ["tiny", "cfg0"])
def test_oracle_inference_path_matches_reference(golden_dir, name):
    """bbox_mode='backbone' (pose_estimation_transformer.py:240-305): more detections than queries (top-k by score), fewer
    (dummy padding) and none at all (None) -- the oracle's query assembly and outputs equal the real reference's."""
    from tests.oracle_runner import run_oracle_inference
    g = np.load(os.path.join(golden_dir, f"poet_{name}_b3_infer.npz"))
    o = run_oracle_inference(name)
    assert list(o["n_boxes"]) == list(g["n_boxes"])
    assert o["n_boxes"][0] == o["cfg"]["num_queries"] and o["n_boxes"][2] == 0
    np.testing.assert_array_equal(o["out"]["pred_classes"].numpy(), g["pred_classes"])
    np.testing.assert_allclose(o["out"]["pred_boxes"].numpy(), g["pred_boxes"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(o["out"]["pred_translation"].numpy(), g["pred_translation"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["out"]["pred_rotation"].numpy(), g["pred_rotation"], rtol=1e-5, atol=1e-6)
    if o["out"]["aux_outputs"]:                         # (cfg0 has a single decoder layer: no auxiliary outputs)
        np.testing.assert_allclose(np.stack([a["pred_rotation"].numpy() for a in o["out"]["aux_outputs"]]), g["aux_rotation"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bbox_mode,class_mode", [("jitter", "specific"), ("gt", "agnostic")])
def test_poet_modes_vs_reference(golden_dir, bbox_mode, class_mode):
    """bbox_mode='jitter' (queries from the perturbed boxes, matching by class equality: pose_estimation_transformer.py:
    208-209, matcher.py:175-181) and class_mode='agnostic' (3 / 6-wide heads, no per-class gather: :85-96,365): outputs,
    losses and gradient checksums of the oracle equal the real reference's."""
    g = _load(golden_dir, f"poet_tiny_b2_pad_{bbox_mode}_{class_mode}.npz")
    r = run_oracle("tiny", 2, True, bbox_mode=bbox_mode, class_mode=class_mode)
    np.testing.assert_allclose(r["out"]["pred_translation"].detach().numpy(), g["pred_translation"], rtol=1e-5, atol=ATOL)
    np.testing.assert_allclose(r["out"]["pred_rotation"].detach().numpy(), g["pred_rotation"], rtol=1e-5, atol=ATOL)
    names = sorted(r["losses"])
    assert names == list(g["loss_names"])
    np.testing.assert_allclose([float(r["losses"][k]) for k in names], g["loss_values"], rtol=1e-5, atol=1e-6)
    grads = dict(r["model"].named_parameters())
    for n, cs in zip(g["grad_names"], g["grad_checksums"]):
        p = grads[str(n)]
        if np.isnan(cs[0]):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
        else:
            np.testing.assert_allclose(checksum(p.grad), cs, rtol=2e-4, atol=2e-6, err_msg=str(n))


@pytest.mark.parametrize("qmode,rmode", [("learned", "bbox"), ("learned", "learned")])
def test_poet_learned_queries_vs_reference(golden_dir, qmode, rmode):
    """--query_embedding learned (nn.Embedding rows as (query_pos | tgt), pose_estimation_transformer.py:149-150,342-343) and
    --reference_points learned (sigmoid(Linear(query_pos)), deformable_transformer.py:157-158): outputs, losses and gradient
    checksums -- incl. query_embed.weight and transformer.reference_points.* -- equal the real reference's."""
    g = _load(golden_dir, f"poet_tiny_b2_pad_q{qmode}_r{rmode}.npz")
    r = run_oracle("tiny", 2, True, query_embedding_mode=qmode, ref_points_mode=rmode)
    np.testing.assert_allclose(r["out"]["pred_translation"].detach().numpy(), g["pred_translation"], rtol=1e-5, atol=ATOL)
    np.testing.assert_allclose(r["out"]["pred_rotation"].detach().numpy(), g["pred_rotation"], rtol=1e-5, atol=ATOL)
    names = sorted(r["losses"])
    assert names == list(g["loss_names"])
    np.testing.assert_allclose([float(r["losses"][k]) for k in names], g["loss_values"], rtol=1e-5, atol=1e-6)
    grads = dict(r["model"].named_parameters())
    assert sorted(grads) == sorted(str(n) for n in g["grad_names"])
    got_ref_grad = False
    for n, cs in zip(g["grad_names"], g["grad_checksums"]):
        p = grads[str(n)]
        if np.isnan(cs[0]):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            np.testing.assert_allclose(checksum(p.grad), cs, rtol=2e-4, atol=2e-6, err_msg=str(n))
            got_ref_grad |= "reference_points" in str(n)
    assert float(grads["query_embed.weight"].grad.abs().max()) > 0
    assert got_ref_grad == (rmode == "learned")


def test_poet_learned_position_embedding_vs_reference(golden_dir):
    """--position_embedding learned (main.py:67; position_encoding.py:87-112: [col_embed(x) | row_embed(y)], the second entry of
    the reference's Joiner, state_dict keys backbone.1.*): outputs, losses and gradient checksums -- incl. the two embedding
    tables' -- equal the real reference's."""
    g = _load(golden_dir, "poet_tiny_b2_pad_pelearned.npz")
    r = run_oracle("tiny", 2, True, position_embedding="learned")
    np.testing.assert_allclose(r["out"]["pred_translation"].detach().numpy(), g["pred_translation"], rtol=1e-5, atol=ATOL)
    np.testing.assert_allclose(r["out"]["pred_rotation"].detach().numpy(), g["pred_rotation"], rtol=1e-5, atol=ATOL)
    names = sorted(r["losses"])
    assert names == list(g["loss_names"])
    np.testing.assert_allclose([float(r["losses"][k]) for k in names], g["loss_values"], rtol=1e-5, atol=1e-6)
    grads = dict(r["model"].named_parameters())
    assert sorted(grads) == sorted(str(n) for n in g["grad_names"])
    for n, cs in zip(g["grad_names"], g["grad_checksums"]):
        p = grads[str(n)]
        if np.isnan(cs[0]):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            np.testing.assert_allclose(checksum(p.grad), cs, rtol=2e-4, atol=2e-6, err_msg=str(n))
    assert float(grads["backbone.1.row_embed.weight"].grad.abs().max()) > 0 and float(grads["backbone.1.col_embed.weight"].grad.abs().max()) > 0


@pytest.mark.parametrize("rotation_mode,aleatoric", [("quat", False), ("silho_quat", False), ("6d", True)])
def test_poet_rotation_modes_vs_reference(golden_dir, rotation_mode, aleatoric):
    """Quaternion representations (4-wide rotation heads, L2-normalised; losses -log(<q,q*>^2 + eps) and log(1 - |<q,q*>| + eps),
    pose_estimation_transformer.py:420-432,564-609) and the aleatoric extension (two more heads predicting log-variances, Gaussian
    losses with the so(3) log map, :490-513,536-562): outputs, losses and gradient checksums equal the real reference's."""
    g = _load(golden_dir, f"poet_tiny_b2_pad_{rotation_mode}{'_aleatoric' if aleatoric else ''}.npz")
    r = run_oracle("tiny", 2, True, rotation_mode=rotation_mode, aleatoric=aleatoric)
    np.testing.assert_allclose(r["out"]["pred_translation"].detach().numpy(), g["pred_translation"], rtol=1e-5, atol=ATOL)
    np.testing.assert_allclose(r["out"]["pred_rotation"].detach().numpy(), g["pred_rotation"], rtol=1e-5, atol=ATOL)
    assert r["out"]["pred_rotation"].shape[-1] == (3 if rotation_mode == "6d" else 4)
    if aleatoric:
        np.testing.assert_allclose(r["out"]["pred_rotation_aleatoric"].detach().numpy(), g["pred_rotation_aleatoric"], rtol=1e-5, atol=ATOL)
        np.testing.assert_allclose(r["out"]["pred_translation_aleatoric"].detach().numpy(), g["pred_translation_aleatoric"], rtol=1e-5, atol=ATOL)
    names = sorted(r["losses"])
    assert names == list(g["loss_names"])
    np.testing.assert_allclose([float(r["losses"][k]) for k in names], g["loss_values"], rtol=1e-5, atol=1e-6)
    grads = dict(r["model"].named_parameters())
    assert sorted(grads) == sorted(str(n) for n in g["grad_names"])
    for n, cs in zip(g["grad_names"], g["grad_checksums"]):
        p = grads[str(n)]
        if np.isnan(cs[0]):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
        else:
            np.testing.assert_allclose(checksum(p.grad), cs, rtol=2e-4, atol=2e-6 * max(1.0, abs(cs[0])), err_msg=str(n))


def _matcher_inputs(g):
    outputs = {"pred_boxes": torch.from_numpy(g["pred_boxes"]), "pred_classes": torch.from_numpy(g["pred_classes"])}
    nt = [int(x) for x in g["n_targets"]]
    tb, tl = np.split(g["tgt_boxes"], np.cumsum(nt)[:-1]), np.split(g["tgt_labels"], np.cumsum(nt)[:-1])
    targets = [{"boxes": torch.from_numpy(b.reshape(-1, 4)), "labels": torch.from_numpy(l)} for b, l in zip(tb, tl)]
    return outputs, targets, [int(x) for x in g["n_boxes"]]


def _flat_matches(res):
    rows = [np.stack([np.full(len(s), b), s.numpy(), t.numpy()], 1).reshape(-1, 3) for b, (s, t) in enumerate(res)]
    return np.concatenate(rows).astype(np.int64)


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_matcher_backbone_mode_vs_reference(golden_dir, which):
    """PoseMatcher(bbox_mode='backbone') -- centre-L1 + class cost, then the class / GIoU filter (matcher.py:183-229; called
    by pose_evaluate / bop_evaluate, engine.py:127,212) -- against the imported reference's matches: jittered, wrong-class and
    spurious detections, an image without detections, one without targets; both class modes, two thresholds.  The product's
    matcher is host code (numpy + SciPy), so it is checked here on CPU tensors as well."""
    g = _load(golden_dir, "matcher_backbone.npz")
    outputs, targets, n_boxes = _matcher_inputs(g)
    if which == "oracle":
        cls = poet_ref.PoseMatcher
    else:
        import poet_amd
        cls = poet_amd.PoseMatcher
    total = 0
    for cm in ("specific", "agnostic"):
        for thr in (0.5, 0.0):
            res = cls(bbox_mode="backbone", class_mode=cm)(outputs, targets, n_boxes, giou_thresh=thr)
            want = g[f"match_{cm}_{int(thr * 10)}"]
            np.testing.assert_array_equal(_flat_matches(res), want)
            total += len(want)
    assert total >= 20                       # the cases do exercise kept AND removed matches
