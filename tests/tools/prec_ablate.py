"""CPU ablation: which bf16 rounding events of the encoder produce the output error at YCB-V geometry (CONFIG=hires|lmo: another)?
Emulates the HIP bf16 policy on the oracle (fp32 math, operands/stores rounded to bf16 where the policy does).
Usage: python tests/tools/prec_ablate.py [input_seed init_seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import poet_ref
from oracle.formula import CONFIGS, make_inputs, make_samples

FL = {}
ALL = ["pos", "q", "src16", "Wv", "V", "Woa", "off", "logit", "out_m", "Wo", "tmp1", "x16", "W1", "Hd", "W2", "tmp2",
       "inprojA", "inprojW", "decA", "decW"]


def R(x, key):
    if not FL.get(key, False):
        return x
    if FL.get(key) == "f16":                  # stored in fp16 instead of bf16 (11 significant bits at the same 2 bytes)
        return x.half().float()
    if key in WKEYS and FL.get("split_w", False) and key not in FL.get("nosplit", ()):          # weight = bf16 hi + bf16 lo (two MFMAs)
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    return x.bfloat16().float()


WKEYS = ("Wv", "Woa", "Wo", "W1", "W2", "inprojW", "decW")


def msda_fwd(self, query_parts, ref, inp, shapes, lsi, mask, enc):
    n, s, _ = inp.shape
    m, l, p = self.n_heads, self.n_levels, self.n_points
    if enc:
        src, pos = query_parts
        q = R(src + pos, "q")
        value = F.linear(R(inp, "src16"), R(self.value_proj.weight, "Wv"), self.value_proj.bias)
    else:
        q = query_parts
        value = F.linear(R(inp, "decA"), R(self.value_proj.weight, "decW"), self.value_proj.bias)
    lq = q.shape[1]
    if mask is not None:
        value = value.masked_fill(mask[..., None], 0.0)
    value = R(value, "V" if enc else "decA").view(n, s, m, -1)
    if enc:
        off = R(F.linear(q, R(self.sampling_offsets.weight, "Woa"), self.sampling_offsets.bias), "off")
        lg = R(F.linear(q, R(self.attention_weights.weight, "Woa"), self.attention_weights.bias), "logit")
    else:
        off = self.sampling_offsets(q); lg = self.attention_weights(q)
    off = off.view(n, lq, m, l, p, 2)
    w = F.softmax(lg.view(n, lq, m, l * p), -1).view(n, lq, m, l, p)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1).float()
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = poet_ref.msda_core(value, shapes.tolist(), loc, w)
    if enc:
        out = R(out, "out_m")
        return R(F.linear(out, R(self.output_proj.weight, "Wo"), self.output_proj.bias), "tmp1")
    return self.output_proj(out)


def enc_layer_fwd(self, src, pos, ref, shapes, lsi, padding_mask=None):
    a = msda_fwd(self.self_attn, (src, pos), ref, src, shapes, lsi, padding_mask, True)
    src = self.norm1(src + a)
    h = R(F.relu(F.linear(R(src, "x16"), R(self.linear1.weight, "W1"), self.linear1.bias)), "Hd")
    f = R(F.linear(h, R(self.linear2.weight, "W2"), self.linear2.bias), "tmp2")
    return self.norm2(src + f)


def dec_layer_fwd(self, tgt, query_pos, ref, src, shapes, lsi, padding_mask=None):
    qk = tgt + query_pos
    a = self.self_attn(qk.transpose(0, 1), qk.transpose(0, 1), tgt.transpose(0, 1))[0].transpose(0, 1)
    tgt = self.norm2(tgt + a)
    c = msda_fwd(self.cross_attn, tgt + query_pos, ref, src, shapes, lsi, padding_mask, False)
    tgt = self.norm1(tgt + c)
    f = self.linear2(F.relu(self.linear1(tgt)))
    return self.norm3(tgt + f)


_orig_tr_fwd = poet_ref.DeformableTransformer.forward


def tr_fwd(self, srcs, masks, pos_embeds, query_embed=None, reference_points=None):
    pos_embeds = [R(p + self.level_embed[i].view(1, -1, 1, 1), "pos") - self.level_embed[i].view(1, -1, 1, 1) for i, p in enumerate(pos_embeds)]
    return _orig_tr_fwd(self, srcs, masks, pos_embeds, query_embed, reference_points)


class RoundedProj(torch.nn.Module):
    def __init__(self, seq):
        super().__init__()
        self.seq = seq

    def forward(self, x):
        conv, gn = self.seq[0], self.seq[1]
        y = F.conv2d(R(x, "inprojA"), R(conv.weight, "inprojW"), conv.bias, conv.stride, conv.padding)
        return gn(R(y, "inprojA"))


def run(model, samples, targets):
    """translations / rotations of ALL decoder layers (auxiliary outputs first, the model output last)"""
    with torch.no_grad():
        out, _ = model(samples, targets)
    aux = out.get("aux_outputs", [])
    return (torch.stack([a["pred_translation"] for a in aux] + [out["pred_translation"]]),
            torch.stack([a["pred_rotation"] for a in aux] + [out["pred_rotation"]]))


def main():
    iseed, wseed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1234, 4321)
    torch.set_num_threads(8)
    cfg = CONFIGS[os.environ.get("CONFIG", "ycbv")]
    feats, sizes, targets = make_inputs(cfg, seed=iseed, batch=int(os.environ.get("BATCH", "1")), pad=bool(os.environ.get("PAD")))
    torch.manual_seed(wseed)
    model, _ = poet_ref.build_poet(cfg, feats)
    if os.environ.get("FORMULA"):
        from oracle.formula import formula_fill
        formula_fill(model)
    model.eval()
    samples = poet_ref.nested_from_list(make_samples(cfg, sizes))
    t0 = time.time()
    cap = {}
    model.transformer.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("mem", o.detach()))
    t_ref, r_ref = run(model, samples, targets)
    mem_ref = cap["mem"]
    print("fp32 ref %.1fs" % (time.time() - t0))
    poet_ref.EncoderLayer.forward = enc_layer_fwd
    poet_ref.DecoderLayer.forward = dec_layer_fwd
    poet_ref.DeformableTransformer.forward = tr_fwd
    model.input_proj = torch.nn.ModuleList([RoundedProj(s) for s in model.input_proj])

    def measure(tag):
        nonlocal t_ref, r_ref, mem_ref
        t, r = run(model, samples, targets)
        dm = (cap["mem"] - mem_ref)
        print(f"{tag:34s} mem rms {dm.pow(2).mean().sqrt():.2e}  dt max {(t - t_ref).abs().max():.2e}  dR rms {(r - r_ref).pow(2).mean().sqrt():.2e} "
              f"max final layer {(r[-1] - r_ref[-1]).abs().max():.2e} all layers {(r - r_ref).abs().max():.2e}", flush=True)

    pols = {"P0 splitW": (),
            "P1 splitW+LNfused": ("tmp1", "tmp2"),
            "P2 P1 + stream operands exact": ("tmp1", "tmp2", "src16", "x16", "q"),
            "P3 P2 + Hd exact": ("tmp1", "tmp2", "src16", "x16", "q", "Hd"),
            "P4 P2 + decA exact": ("tmp1", "tmp2", "src16", "x16", "q", "decA"),
            "P5 P1 + decA exact": ("tmp1", "tmp2", "decA")}
    if os.environ.get("NSITE"):
        # n-site table: today's policy (P1) with one, two, three ... rounding sites made exact ON TOP of each other
        base = ("tmp1", "tmp2")
        pols = {"Q0 policy (P1)": base,
                "Q1 + offsets fp32": base + ("off",),
                "Q2 + offsets, logits fp32": base + ("off", "logit"),
                "Q3 + offsets + input_proj operands": base + ("off", "inprojA"),
                "Q4 + offsets + input_proj + value maps": base + ("off", "inprojA", "V"),
                "Q5 + offsets + input_proj + stream operands": base + ("off", "inprojA", "src16", "x16", "q"),
                "Q6 + offsets + input_proj + FFN hidden": base + ("off", "inprojA", "Hd"),
                "Q7 + offsets + input_proj + MSDA out": base + ("off", "inprojA", "out_m"),
                "Q8 input_proj operands only": base + ("inprojA",),
                "Q9 + all encoder activations (dec only)": base + ("off", "logit", "inprojA", "V", "src16", "x16", "q", "Hd", "out_m", "pos"),
                "QA offsets, logits in fp16": base + ("off:f16", "logit:f16"),
                "QB offsets, logits fp16 + input_proj": base + ("off:f16", "logit:f16", "inprojA"),
                "QC QB + MSDA out, FFN hidden in fp16": base + ("off:f16", "logit:f16", "inprojA", "out_m:f16", "Hd:f16"),
                "QD QB + value maps fp16": base + ("off:f16", "logit:f16", "inprojA", "V:f16")}
        if os.environ.get("QSEL"):
            pols = {k: v for k, v in pols.items() if k.split()[0] in os.environ["QSEL"].split(",")}
        for rep in range(int(os.environ.get("REPS", "3"))):
            if rep:
                g = torch.Generator().manual_seed(rep)
                for f in feats:
                    f.mul_(1 + 3e-7 * torch.randn(f.shape, generator=g))
                FL.clear()
                t_ref, r_ref = run(model, samples, targets)
                mem_ref = cap["mem"]
            for tag, off in pols.items():
                FL.clear()
                for k in ALL: FL[k] = k not in off
                for k in off:
                    if k.endswith(":f16"): FL[k[:-4]] = "f16"
                FL["split_w"] = True
                measure(f"rep{rep} {tag}")
        return
    if os.environ.get("ONLY"):
        for k in ALL:
            FL.clear(); FL[k] = True; FL["split_w"] = False
            measure("only " + k)
        return
    if os.environ.get("W16TEST"):
        # (round 6) what would ONE fp16 product per fragment cost in parity?  A = today's policy (split-bf16 weights, fp16 offsets | logits /
        # value maps / LayerNorm branch operands); B = the encoder's forward weights as single IEEE fp16 values (the bf16 activation
        # operands convert to fp16 exactly, so an f16 MFMA would take them as they are stored); C = B with the activation operands
        # stored in fp16 as well.  `tmp1` / `tmp2`: the projection outputs the LayerNorms read.
        f16_today = ("off", "logit", "V", "tmp1", "tmp2")
        encw = ("Wv", "Woa", "Wo", "W1", "W2")
        pols = {"A today (split weights)": (f16_today, ()),
                "B encoder weights single fp16": (f16_today + encw, ()),
                "C B + activation operands fp16": (f16_today + encw + ("src16", "x16", "q", "Hd", "out_m"), ())}
        for rep in range(int(os.environ.get("REPS", "3"))):
            if rep:
                g = torch.Generator().manual_seed(rep)
                for f in feats:
                    f.mul_(1 + 3e-7 * torch.randn(f.shape, generator=g))
                FL.clear()
                t_ref, r_ref = run(model, samples, targets)
                mem_ref = cap["mem"]
            for tag, (f16s, exact) in pols.items():
                FL.clear()
                for k in ALL: FL[k] = k not in exact
                for k in f16s: FL[k] = "f16"
                FL["split_w"] = True
                measure(f"rep{rep} {tag}")
        return
    if os.environ.get("W2TEST"):
        for rep in range(3):
            if rep:
                g = torch.Generator().manual_seed(rep)
                for f in feats:
                    f.mul_(1 + 3e-7 * torch.randn(f.shape, generator=g))
                FL.clear()
                t_ref, r_ref = run(model, samples, targets)
                mem_ref = cap["mem"]
            for tag, ns in (("split all", ()), ("split all but W2", ("W2",)), ("split all but W2,inprojW", ("W2", "inprojW"))):
                FL.clear()
                for k in ALL: FL[k] = True
                FL["split_w"] = True; FL["nosplit"] = ns
                measure(f"rep{rep} {tag}")
        return
    for rep in range(int(os.environ.get("REPS", "3"))):
        if rep:
            g = torch.Generator().manual_seed(rep)
            for f in feats:
                f.mul_(1 + 3e-7 * torch.randn(f.shape, generator=g))          # new rounding realisation, same problem
            FL.clear()
            t_ref, r_ref = run(model, samples, targets)
            mem_ref = cap["mem"]
        FL.clear()
        for k in ALL: FL[k] = True
        measure(f"rep{rep} ALL nosplit")
        for tag, off in pols.items():
            FL.clear()
            for k in ALL: FL[k] = k not in off
            FL["split_w"] = True
            measure(f"rep{rep} {tag}")


if __name__ == "__main__":
    main()
