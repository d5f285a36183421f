"""Forward/backward kernel programs of the PoET hot path (no autograd here).

Each function is a straight-line sequence of libpoet_hip.so launches on torch-owned buffers.
Reference arithmetic restated (paths relative to aau-cns/poet):
  value_proj_* / sample_*   MSDeformAttn.forward (external op; call sites
                            models/deformable_transformer.py:201,283; spec SURVEY.md App. A)
  proj_ln_*                 `x = norm(x + dropout(proj(y)))`  deformable_transformer.py:202-203,279-287
  ffn_*                     forward_ffn  deformable_transformer.py:193-197,269-273
"""
from __future__ import annotations

import torch

from . import ops

_seed_state = {"base": 0x1234567, "count": 0}


def manual_seed(seed: int):
    _seed_state["base"] = (int(seed) * 2654435761 + 12345) & 0xFFFFFFFF
    _seed_state["count"] = 0


def next_seed() -> int:
    _seed_state["count"] += 1
    return (_seed_state["base"] + _seed_state["count"] * 0x9E3779B1) & 0xFFFFFFFF


_ENV_HEAD_LOOP = __import__("os").environ.get("POET_HEAD_DX_LOOP", "0") not in ("", "0")      # (A/B aid, read at import)
_ENV_SEG_FUSE = __import__("os").environ.get("POET_NO_SEG_FUSE", "0") in ("", "0")      # (A/B aid, read at import)
_ENV_DW_MERGE = __import__("os").environ.get("POET_NO_DW_MERGE", "0") in ("", "0")
# POET_FSTREAM_SPLIT=1: the residual stream between the encoder's LayerNorms as bf16 head + fp16 remainder instead of fp32 (norm.hip;
# -0.04 ms per step at YCB-V).  OFF by default: 2^-20 against fp32 still flips a few bf16 roundings downstream, and the goldens of the
# reference's own random init sit on a coin toss there (YCB-V `_init`: two of five members of the policy's noise ensemble are 13 % off in
# every gradient because one nearly degenerate 6D output dominates the loss) -- the fp32 stream keeps the realisation of rounds 3-6.
_ENV_FSTREAM = __import__("os").environ.get("POET_FSTREAM_SPLIT", "0") not in ("", "0")
_ENV_GSTREAM = __import__("os").environ.get("POET_GSTREAM_F32", "0") in ("", "0")       # (POET_GSTREAM_F32=1: fp32 gradient stream, rounds 1-5)


def empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


class GradSink:
    """Hands out fp32 gradient buffers for parameters.  A parameter that carries `_grad_view`
    (a slice of the flat gradient arena, see poet_amd.engine.ParamArena) is written in place and
    autograd gets None for it; otherwise a zero-filled tensor is created and returned to autograd."""

    def __init__(self, names, params):
        self.index = {n: i for i, n in enumerate(names)}
        self.params = params
        self.ret = [None] * len(params)

    def __call__(self, name):
        i = self.index[name]
        p = self.params[i]
        gv = getattr(p, "_grad_view", None)
        if gv is not None:
            return gv
        if self.ret[i] is None:
            self.ret[i] = torch.zeros_like(p, dtype=torch.float32)
        return self.ret[i]


def Wb(W, like):
    """bf16 shadow of a weight (kept up to date by the AdamW kernel, see engine.ParamArena) when the other GEMM operand
    is stored in bf16 -- then both MFMA operands stream in at 2 B/element; otherwise the fp32 master itself."""
    if like.dtype == torch.bfloat16:
        sh = getattr(W, "_bf16", None)
        if sh is not None:
            return sh
    return W


def Wf(W, like, split):
    """Forward weight operand: (tensor, b_split).  bf16 policy with split weights: the fp32 master itself, which the GEMM
    takes as bf16 hi + bf16 lo (PoetGemmDesc.b_split); otherwise whatever Wb gives."""
    if split and like.dtype == torch.bfloat16 and W.dtype == torch.float32:
        return W, True
    return Wb(W, like), False


def vstrides(M, S, D):
    """element strides (n, s, m) of a contiguous head-major (N,M,S,D) value map"""
    return (M * S * D, D, S * D)


def vstrides_of(V):
    """element strides (n, s, m) of a head-major value map that may be a head slice of a larger (N, M_all, S, D) tensor (the
    decoder's value maps of all layers come out of ONE projection)"""
    return (V.stride(0), V.stride(2), V.stride(1))


# ---- (a) value projection ------------------------------------------------------------------------
def value_proj_fwd(inp2d, W, b, row_mask, N, S, M, D, act=None, split=False):
    V = empty((N, M, S, D), act or inp2d.dtype, inp2d)
    Wt, sp = Wf(W, inp2d, split)
    ops.linear_fwd(inp2d, Wt, b, V, row_mask=row_mask, head_major=(M, S, D), split=sp)
    return V


def value_proj_bwd(dV, inp2d, W, row_mask, N, S, M, D, gW, gb, dinp, accumulate, act=None, wb_is_operand=False, dVr=None):
    """wb_is_operand: W is already the GEMM operand (a stacked view chosen by the caller), not a parameter to look a shadow up for.
    dVr: optional caller-owned (rows, M*D) buffer for the gradient rows (a column block of a wider one); with dV=None it already
    HOLDS the gradient rows (scattered in place, masked rows not yet zeroed)."""
    rows, d = N * S, M * D
    if dV is None:                                                     # the scatter wrote dVr itself (token-major rows): only the
        if row_mask is not None:                                       # masked_fill backward is left
            ops.zero_masked_rows(dVr, row_mask, rows, d)
    else:
        if dVr is None:
            dVr = empty((rows, d), act or inp2d.dtype, inp2d)
        ops.vgrad_to_rows(dV, vstrides(M, S, D), row_mask, dVr, N, S, M, D, ld_out=dVr.stride(0))
    ops.linear_dw(dVr, inp2d, gW, rows=rows, ldy=dVr.stride(0), db=gb)
    if dinp is not None:
        ops.linear_dx(dVr, W if wb_is_operand else Wb(W, dVr), dinp, rows=rows, ldy=dVr.stride(0), add_src=dinp if accumulate else None)


# ---- (b) offsets/logits projection + fused deformable sampling ----------------------------------
def _pair(so_w, like):
    """(weight, bias, grad weight, grad bias) of the stacked [sampling_offsets | attention_weights] Linear when the arena
    keeps the two adjacent (engine.ParamArena._link_projection_pairs); None otherwise (plain modules outside an arena)."""
    pr = getattr(so_w, "_pair", None)
    if pr is None:
        return None
    return (pr["w16"] if like.dtype == torch.bfloat16 else pr["w"]), pr["b"], pr["gw"], pr["gb"]


def sample_fwd(q2d, so_w, so_b, aw_w, aw_b, V, geom, ref, ref_bs, N, Lq, M, D, P, act=None, split=False, grid_queries=False):
    mlp = M * geom.L * P
    ldq, rows = 3 * mlp, N * Lq
    odt = act or q2d.dtype
    # offsets | logits in fp16 where their readers take it (ops.oa_f16): the sampling offsets are the largest single rounding site of
    # the bf16 policy (1/32 px at |offset| ~ 4), fp16 stores them 8x finer in the same bytes
    oa16 = odt == torch.bfloat16 and V.dtype in (torch.bfloat16, torch.float16) and ops.oa_f16(M, D, geom.L, P, grid_queries)
    OA = empty((rows, ldq), torch.float16 if oa16 else odt, q2d)
    pr = getattr(so_w, "_pair", None)
    if pr is not None:                        # offsets | logits = ONE Linear over the adjacent parameters
        Wt, sp = Wf(pr["w"], q2d, split)
        if not sp:
            Wt = pr["w16"] if q2d.dtype == torch.bfloat16 else pr["w"]
        ops.linear_fwd(q2d, Wt, pr["b"], OA, split=sp)
    else:
        Wt, sp = Wf(so_w, q2d, split)
        ops.linear_fwd(q2d, Wt, so_b, OA, ldc=ldq, split=sp)
        Wt, sp = Wf(aw_w, q2d, split)
        ops.linear_fwd(q2d, Wt, aw_b, OA[:, 2 * mlp:], ldc=ldq, split=sp)
    out = empty((rows, M * D), odt, q2d)
    ops.msda_fused_fwd(V, vstrides_of(V), geom, OA, ldq, 2 * mlp, ref, ref_bs, out, N, M, D, P, Lq, grid_queries=grid_queries)
    return out, OA


def sample_bwd(d_out, q2d, OA, so_w, aw_w, V, geom, ref, ref_bs, N, Lq, M, D, P, dV, g_so_w, g_so_b, g_aw_w, g_aw_b,
               dq, dq_accumulate, seg_sums=None, grid_queries=False, dOA=None):
    """dOA: optional caller-owned (rows, >= 3 M L P) buffer for d(offsets | logits) (a column block of a wider one: row stride
    dOA.stride(0)); with dq=None the input-gradient product is left to the caller."""
    mlp = M * geom.L * P
    ldq, rows = 3 * mlp, N * Lq
    if dOA is None:
        dOA = torch.empty(OA.shape, dtype=torch.bfloat16 if OA.dtype == torch.float16 else OA.dtype, device=OA.device)    # (gradients of an fp16 buffer are bf16)
    ldg = dOA.stride(0)
    gvs = vstrides_of(dV)                                              # (the decoder scatters into token-major rows: its own strides)
    ops.msda_fused_bwd(V, vstrides_of(V), geom, OA, ldq, 2 * mlp, ref, ref_bs, d_out, dV, dOA, N, M, D, P, Lq,
                       grid_queries=grid_queries, ld_grad=ldg, gv_strides=None if gvs == vstrides_of(V) else gvs)
    plain = seg_sums is None
    pr = _pair(so_w, dOA)
    seg_done = False
    if pr is not None and pr[2].data_ptr() == g_so_w.data_ptr():      # (the gradient sink is the arena: stacked dW + db)
        # the per-level column sums ride in the weight-gradient kernel's own pass over d(offsets | logits) (PoetGemmDesc.seg_sums):
        # no 157 MB column-sum launch per layer
        seg_done = seg_sums is not None and seg_sums.stride(0) == ldq and Lq >= 64 and geom.L <= 8 and _ENV_SEG_FUSE
        ops.linear_dw(dOA, q2d, pr[2], rows=rows, ldy=ldg, db=pr[3] if plain else None,
                      seg=(seg_sums, geom.c_segs, Lq) if seg_done else None)
    else:
        ops.linear_dw(dOA, q2d, g_so_w, rows=rows, ldy=ldg, db=g_so_b if plain else None)
        ops.linear_dw(dOA[:, 2 * mlp:], q2d, g_aw_w, rows=rows, ldy=ldg, db=g_aw_b if plain else None)
    if seg_sums is not None:      # per-level column sums (encoder): feeds both the biases and level_embed
        if not seg_done:
            ops.colsum(dOA, ldg, seg_sums, N, Lq, ldq, geom.c_segs, geom.L)
        ops.colsum(seg_sums, ldq, g_so_b, 1, geom.L, 2 * mlp)
        ops.colsum(seg_sums[:, 2 * mlp:], ldq, g_aw_b, 1, geom.L, mlp)
    if dq is not None:
        if pr is not None:                    # d(query) (+)= dOA [W_so ; W_aw]: one product with K = 3 M L P
            ops.linear_dx(dOA, pr[0], dq, rows=rows, ldy=ldg, add_src=dq if dq_accumulate else None)
        else:
            ops.linear_dx(dOA, Wb(so_w, dOA), dq, rows=rows, ldy=ldg, add_src=dq if dq_accumulate else None)
            ops.linear_dx(dOA[:, 2 * mlp:], Wb(aw_w, dOA), dq, rows=rows, ldy=ldg, add_src=dq)


# ---- (c) projection + residual + dropout + LayerNorm ----------------------------------------------
def proj_ln_fwd(x_in, W, b, res, gamma, beta, p, seed, split=False, pos_next=None, y_out=None, stream16=False, res16=None):
    """Returns (y, y16, saved): y in the residual-stream dtype; y16 = bf16 copy for the next GEMM when the stream is
    fp32 but the branch is bf16 (else y itself).  With split weights the projection hands its fp32 accumulators to the
    LayerNorm unrounded (the branch is never stored in bf16); only the pre-norm sum saved for backward is bf16.
    stream16: hand y over as the SPLIT stream where this launch can (an fp16 branch, see below): y16 is its bf16 head and the returned
    y the fp16 remainder y - float(y16).  res may arrive in that form from such a launch: then res16 is its head."""
    rows, d = res.shape[0], W.shape[0]
    mixed = x_in.dtype == torch.bfloat16 and res.dtype in (torch.float32, torch.float16)
    Wt, sp = Wf(W, x_in, split)
    lo = getattr(W, "_bf16_lo", None) if (sp and mixed and rows >= 4096 and W.shape[1] >= 512 and W.shape[0] == 256) else None
    # (the long-K kernel's own admission rules, gemm_pipe.hip:gemm_pipe_try: K a multiple of 64, 16-byte aligned operands, row
    # strides of 8 / 4 elements -- anything else, e.g. dim_feedforward = 1000 or a sliced x_in, keeps the fp32-master split kernel)
    if lo is not None and not (getattr(W, "_bf16", None) is not None and W.shape[1] % 64 == 0 and x_in.stride(-1) == 1 and x_in.stride(0) % 8 == 0 and x_in.data_ptr() % 16 == 0
                               and lo.data_ptr() % 16 == 0 and W._bf16.data_ptr() % 16 == 0):
        lo = None
    use_pipe = lo is not None and getattr(W, "_bf16", None) is not None and ops.pipe_split()
    # (round 6) ... as IEEE fp16 where the LayerNorm launch is the only reader (ops.ln_f16: >= 4096 rows, d = 256): 2^-11 relative, a
    # quarter of the rounding the LayerNorm's own bf16 operand copy applies next, and 2 bytes per element less on both sides -- on the
    # kernels that store it: the K = 256 streaming kernel and the long-K pipeline (a K >= 512 product WITHOUT the arena's two weight
    # images -- plain modules, GraphedInference -- runs the K-chunked fp32-master kernel, which writes fp32)
    f16_ok = ops.ln_f16(rows, d) and (W.shape[1] == 256 or use_pipe)
    tmp = empty((rows, d), (torch.float16 if f16_ok else torch.float32) if (sp and mixed) else x_in.dtype, res)
    if use_pipe:
        # long-K split product on the arena's two bf16 images, tmp = x (hi + lo)^T + b in ONE pass over x (gemm_pipe.hip):
        # every activation fragment staged in LDS meets both weight images
        ops.linear_fwd(x_in, W._bf16, b, tmp, W_lo=lo)
    else:
        ops.linear_fwd(x_in, Wt, b, tmp, split=sp)
    if res.dtype == torch.float16 and tmp.dtype != torch.float16:      # (a split stream only travels between fp16-branch launches)
        res = res16.float() + res.float()
    # The residual stream between the encoder's LayerNorms as bf16 head + fp16 remainder (bf16 policy, fp16 branch; norm.hip): the head
    # IS the operand copy y16 this launch writes anyway, so the stream costs 52 MB per launch to write instead of 104 (and the same 104
    # to read), at 2^-20 relative -- the realisation of every rounding downstream stays the fp32 stream's.
    ydt = torch.float16 if (stream16 and tmp.dtype == torch.float16 and y_out is None and mixed) else (torch.float32 if res.dtype == torch.float16 else res.dtype)
    y = torch.empty(res.shape, dtype=ydt, device=res.device) if y_out is None else y_out      # (y_out: the caller's buffer, e.g. a row of hs)
    z = empty((rows, d), x_in.dtype, res)            # branch dtype
    mean = empty((rows,), torch.float32, res)
    rstd = empty((rows,), torch.float32, res)
    y16 = empty((rows, d), torch.bfloat16, res) if mixed else None
    q_next = None
    if pos_next is not None and mixed and pos_next.dtype == torch.bfloat16:       # the next layer's `src + pos`, written by this LayerNorm
        q_next = empty((rows, d), torch.bfloat16, res)
    ops.ln_fwd(tmp, res, gamma, beta, y, z, mean, rstd, rows, d, 1e-5, p, seed, y16=y16,
               pos16=pos_next if q_next is not None else None, q16=q_next, res16=res16 if res.dtype == torch.float16 else None)
    if pos_next is not None:
        return y, (y16 if y16 is not None else y), (z, mean, rstd), q_next
    return y, (y16 if y16 is not None else y), (z, mean, rstd)


def proj_ln_bwd(dy, x_in, W, gamma, saved, p, seed, gW, gb, ggamma, gbeta, gate_ref=None, gate_scale=1.0, stream_dtype=None):
    """stream_dtype: storage of the returned stream gradient dz (default: dy's; torch.bfloat16 = the encoder's bf16 gradient
    stream, see enc_layer_bwd)."""
    z, mean, rstd = saved
    rows, d = z.shape
    dz = torch.empty(dy.shape, dtype=stream_dtype or dy.dtype, device=dy.device)
    # (a deferred weight gradient reads dxo after this function returned: it must not alias dz, which the callers go on
    # accumulating into)
    separate = p > 0 or z.dtype != dy.dtype or dz.dtype != dy.dtype or ops.defer_small_dw.active is not None
    dxo = torch.empty_like(z) if separate else dz
    ops.ln_bwd(dy, z, mean, rstd, gamma, dz, dxo if separate else None, ggamma, gbeta, rows, d, p, seed)
    dx_in = empty((rows, W.shape[1]), x_in.dtype, x_in)
    Wop = Wb(W, dxo)
    if (gate_ref is None or gate_ref is x_in) and ops.linear_bwd_ok(dxo, x_in, Wop, gW, rows):
        # weight, bias and input gradient of the 256-wide-output Linear in ONE pass over dxo and x_in (poet_linear_bwd): the hidden
        # activation of the FFN is read once instead of twice (as the dW operand and as the ReLU / dropout gate)
        ops.linear_bwd(dxo, x_in, Wop, gW, gb, dx_in, rows=rows, gate=gate_ref is not None, gate_scale=gate_scale)
        return dz, dx_in
    ops.linear_dw(dxo, x_in, gW, rows=rows, db=gb)
    ops.linear_dx(dxo, Wop, dx_in, rows=rows, gate_ref=gate_ref, gate_scale=gate_scale)
    return dz, dx_in


# ---- (d) FFN block -------------------------------------------------------------------------------
def ffn_fwd(x, x16, W1, b1, W2, b2, gamma, beta, p_h, p_o, seed_h, seed_o, act=None, split=False, pos_next=None, y_out=None, fn="relu",
            stream16=False):
    """x: residual stream; x16: the GEMM operand copy of it (== x in the pure modes).  fn: the FFN's activation
    (deformable_transformer.py:347-355): "relu" rides in the Linear's epilogue together with the dropout; "gelu" keeps the
    pre-activation (its derivative needs it) and runs activation + dropout as one element-wise pass (poet_gelu_fwd)."""
    rows = x.shape[0]
    Hd = empty((rows, W1.shape[0]), act or x.dtype, x)
    Wt, sp = Wf(W1, x16, split)
    extra = ()
    if fn == "relu":
        ops.linear_fwd(x16, Wt, b1, Hd, act=1, drop_p=p_h, seed=seed_h, split=sp)
    elif fn == "gelu":
        pre = empty(Hd.shape, torch.float32, Hd)      # (fp32 also under bf16 storage: one rounding -- gelu's output -- instead of two)
        ops.linear_fwd(x16, Wt, b1, pre, split=sp)
        ops.gelu_fwd(pre, Hd, p_h, seed_h)
        extra = (pre, seed_h)
    else:
        # ("glu": F.glu halves the hidden width and the reference's linear2 then fails on the shape, deformable_transformer.py:193-197)
        raise RuntimeError(f"activation {fn!r}: the FFN's second Linear takes d_ffn columns, glu leaves d_ffn / 2 (the reference fails here too)")
    if pos_next is not None:
        y, y16, ln_saved, q_next = proj_ln_fwd(Hd, W2, b2, x, gamma, beta, p_o, seed_o, split=split, pos_next=pos_next, stream16=stream16, res16=x16)
        return y, y16, (Hd, ln_saved) + extra, q_next
    y, y16, ln_saved = proj_ln_fwd(Hd, W2, b2, x, gamma, beta, p_o, seed_o, split=split, y_out=y_out, stream16=stream16, res16=x16)
    return y, y16, (Hd, ln_saved) + extra


def ffn_bwd(dy, x, W1, W2, gamma, saved, p_h, p_o, seed_o, gW1, gb1, gW2, gb2, ggamma, gbeta, stream_dtype=None):
    Hd, ln_saved = saved[:2]
    rows, f = Hd.shape
    if len(saved) > 2:                        # GELU: d(hidden) ungated, then the element-wise backward on the kept pre-activation
        pre, seed_h = saved[2:]
        dz, dh = proj_ln_bwd(dy, Hd, W2, gamma, ln_saved, p_o, seed_o, gW2, gb2, ggamma, gbeta, stream_dtype=stream_dtype)
        ops.gelu_bwd(dh, pre, dh, p_h, seed_h)
    else:
        dz, dh = proj_ln_bwd(dy, Hd, W2, gamma, ln_saved, p_o, seed_o, gW2, gb2, ggamma, gbeta,
                             gate_ref=Hd, gate_scale=1.0 / (1.0 - p_h) if p_h > 0 else 1.0, stream_dtype=stream_dtype)
    ops.linear_dw(dh, x, gW1, rows=rows, db=gb1)
    ops.linear_dx(dh, Wb(W1, dh), dz, rows=rows, add_src=dz)
    return dz


# ---- encoder layer ---------------------------------------------------------------------------------
ENC_PARAMS = ("self_attn.sampling_offsets.weight", "self_attn.sampling_offsets.bias",
              "self_attn.attention_weights.weight", "self_attn.attention_weights.bias",
              "self_attn.value_proj.weight", "self_attn.value_proj.bias",
              "self_attn.output_proj.weight", "self_attn.output_proj.bias",
              "norm1.weight", "norm1.bias", "linear1.weight", "linear1.bias",
              "linear2.weight", "linear2.bias", "norm2.weight", "norm2.bias")


def enc_layer_fwd(src, src16, pos, P_, ref, ref_bs, mask, geom, N, M, npts, p, training, act=None, split=False, q_in=None, emit_q=False,
                  ffn_act="relu", stream_out16=False):
    """src (N*S,d): residual stream; src16: its GEMM-operand copy (== src in the pure modes); pos (N*S,d).
    q_in: `src + pos` when the previous layer's LayerNorm already produced it; emit_q: have this layer's last LayerNorm
    produce it for the next layer.  Returns (out, out16, saved[, q_next])."""
    S, d = geom.S, src.shape[1]
    D = d // M
    pd = p if training else 0.0
    seeds = [next_seed() for _ in range(3)]
    if q_in is not None:
        q = q_in
    else:
        q = empty(src.shape, src16.dtype, src)
        ops.add(src16.float() + src.float() if src.dtype == torch.float16 else src, pos, q)      # (a split stream normally arrives with its q_in)
    # value maps in fp16 where their only readers take it (ops.v_f16: the shared-geometry gathers): 11 mantissa bits instead of 8 in
    # the same bytes, and the forward gather multiplies the halves straight out of the packed pair (v_fma_mix_f32: no unpack)
    v16 = (act or src16.dtype) == torch.bfloat16 and ops.v_f16(M, D, geom.L, npts, True)
    V = value_proj_fwd(src16, P_["self_attn.value_proj.weight"], P_["self_attn.value_proj.bias"], mask, N, S, M, D,
                       torch.float16 if v16 else act, split)
    out_m, OA = sample_fwd(q, P_["self_attn.sampling_offsets.weight"], P_["self_attn.sampling_offsets.bias"],
                           P_["self_attn.attention_weights.weight"], P_["self_attn.attention_weights.bias"],
                           V, geom, ref, ref_bs, N, S, M, D, npts, act, split, grid_queries=True)
    # (stream_out16: the caller takes the layer's output stream as fp16 -- every layer but the last; inside the layer the stream between
    # the two LayerNorms is fp16 whenever the first one can store it)
    x1, x1_16, ln1 = proj_ln_fwd(out_m, P_["self_attn.output_proj.weight"], P_["self_attn.output_proj.bias"], src,
                                 P_["norm1.weight"], P_["norm1.bias"], pd, seeds[0], split, stream16=_ENV_FSTREAM, res16=src16)
    q_next = None
    if emit_q:
        x2, x2_16, ffn, q_next = ffn_fwd(x1, x1_16, P_["linear1.weight"], P_["linear1.bias"], P_["linear2.weight"], P_["linear2.bias"],
                                         P_["norm2.weight"], P_["norm2.bias"], pd, pd, seeds[1], seeds[2], act, split, pos_next=pos, fn=ffn_act,
                                         stream16=_ENV_FSTREAM and stream_out16)
    else:
        x2, x2_16, ffn = ffn_fwd(x1, x1_16, P_["linear1.weight"], P_["linear1.bias"], P_["linear2.weight"], P_["linear2.bias"],
                                 P_["norm2.weight"], P_["norm2.bias"], pd, pd, seeds[1], seeds[2], act, split, fn=ffn_act,
                                 stream16=_ENV_FSTREAM and stream_out16)
    saved = dict(src=src16, q=q, V=V, OA=OA, out_m=out_m, ln1=ln1, x1=x1_16, ffn=ffn, seeds=seeds, pd=pd)
    if emit_q:
        return x2, x2_16, saved, q_next
    return x2, x2_16, saved


def enc_layer_bwd(dx2, sv, P_, G, pre, ref, ref_bs, mask, geom, N, M, npts, g_level, dpos=None, out_f32=False):
    """G: GradSink; pre: name prefix of this layer's params.  Returns d(src).  dpos: optional (N*S, d) fp32 OUTPUT that receives
    d(src + pos) of this layer's offsets | logits projection -- the gradient of a learned position encoding (then the query
    gradient is its own product instead of a block of the stacked K = 1024 one).  out_f32: the caller needs d(src) in fp32 (layer 0:
    the input projection's backward reads it) -- a bf16 gradient stream then ends at this layer's first LayerNorm."""
    S, d = geom.S, dx2.shape[1]
    D = d // M
    pd, seeds = sv["pd"], sv["seeds"]
    g = lambda n: G(pre + n)
    v2b = sv["V"].dtype in (torch.bfloat16, torch.float16)             # 2-byte value maps (fp16: ops.v_f16)
    mlp = M * geom.L * npts
    so_w = P_["self_attn.sampling_offsets.weight"]
    tri = getattr(so_w, "_triple", None) if sv["OA"].dtype in (torch.bfloat16, torch.float16) and v2b else None
    if tri is not None and (tri["n_oa"] != 3 * mlp or tri["w16"].shape[0] != 3 * mlp + d or dpos is not None or (3 * mlp) % 8 != 0):
        tri = None
    # The bf16 GRADIENT STREAM (bf16 policy, the arena's stacked weights): d(src) travels through the layer -- and on to the layer
    # below -- as bf16 rows instead of fp32.  Its writers / read-modify-writers are the two LayerNorm backwards and the two long-K
    # input-gradient products (gemm_pipe, bf16 C accumulated in place); every one of them is HBM-bound, and the stream is 2 x 104 MB
    # of each one's ~310-420 MB.  All arithmetic stays fp32 (LayerNorm statistics, MFMA accumulators); what changes is the rounding
    # of the stored sum, 2^-9 relative per hand-over, against gradient OPERANDS (d(hidden), d(offsets | logits), d(value) rows, the
    # LayerNorm branch gradients) that are bf16 already.  Conditions = what gemm_pipe takes (256 columns, >= 4096 rows, K % 64 == 0).
    d_ffn = P_["linear1.weight"].shape[0]
    stream = None
    if (_ENV_GSTREAM and tri is not None and d == 256 and N * S >= 4096 and d_ffn >= 512 and d_ffn % 64 == 0
            and (3 * mlp + d) >= 512 and (3 * mlp + d) % 64 == 0 and sv["x1"].dtype == torch.bfloat16 and len(sv["ffn"]) == 2):
        stream = torch.bfloat16
    elif dx2.dtype != torch.float32:                                   # (a bf16 hand-over into a layer that cannot continue it)
        dx2 = dx2.float()
    dx1 = ffn_bwd(dx2, sv["x1"], P_["linear1.weight"], P_["linear2.weight"], P_["norm2.weight"], sv["ffn"], pd, pd,
                  seeds[2], g("linear1.weight"), g("linear1.bias"), g("linear2.weight"), g("linear2.bias"),
                  g("norm2.weight"), g("norm2.bias"), stream_dtype=stream)
    dsrc, d_out_m = proj_ln_bwd(dx1, sv["out_m"], P_["self_attn.output_proj.weight"], P_["norm1.weight"], sv["ln1"], pd,
                                seeds[0], g("self_attn.output_proj.weight"), g("self_attn.output_proj.bias"),
                                g("norm1.weight"), g("norm1.bias"), stream_dtype=torch.float32 if (stream is not None and out_f32) else stream)
    # grid queries + bf16 storage + D = 16: the LDS-tiled scatter can hand over the value gradient in bf16 (packed bf16x2
    # atomics; its consumer, the value projection's backward, rounds it to bf16 anyway) -- half the atomics, zero-fill and read
    gv16 = (v2b and sv["OA"].dtype in (torch.bfloat16, torch.float16) and D == 16 and npts == 4 and geom.L * npts <= 16
            and ops.tiled_scatter_bf16())
    dV = ops.zeros(sv["V"].shape, torch.bfloat16 if gv16 else torch.float32, dx2.device)
    G2 = None
    if tri is not None:
        # d(src) += [d(offsets|logits) | d(value) rows] [W_so ; W_aw ; W_v]: both gradient blocks are written into ONE
        # (rows, 3 M L P + d) buffer and the two input-gradient products (K = 768 and K = 256, each a read-modify-write of the
        # fp32 stream) become one with K = 1024
        G2 = torch.empty((N * S, 3 * mlp + d), dtype=torch.bfloat16, device=dx2.device)
    g_so_w = g("self_attn.sampling_offsets.weight")
    if (G2 is not None and _ENV_DW_MERGE and tri.get("gw") is not None and tri["gw"].data_ptr() == g_so_w.data_ptr() and S >= 64
            and geom.L <= 8 and sv["src"].dtype == torch.bfloat16 and sv["q"].dtype == torch.bfloat16):
        # The weight gradients of the three stacked Linears [sampling_offsets ; attention_weights ; value_proj] as ONE launch: their
        # gradient rows are the column blocks of G2 = [d(offsets | logits) | d(value) rows]; the first two pair with the query src + pos,
        # the third with src (PoetGemmDesc.B_alt).  The per-level column sums of ALL 3 M L P + d columns ride in the same pass: bias
        # gradients of the three Linears and d(level_embed).  8 tiles x 32 row ranges instead of 6 x 40 + (2 x 128 on the slow shape).
        ldg = G2.stride(0)
        ops.msda_fused_bwd(sv["V"], vstrides_of(sv["V"]), geom, sv["OA"], 3 * mlp, 2 * mlp, ref, ref_bs, d_out_m, dV, G2, N, M, D, npts, S,
                           grid_queries=True, ld_grad=ldg)
        ops.vgrad_to_rows(dV, vstrides(M, S, D), mask, G2[:, 3 * mlp:], N, S, M, D, ld_out=ldg)
        segw = ops.zeros((geom.L, 3 * mlp + d), torch.float32, dx2.device)
        ops.linear_dw(G2, sv["q"], tri["gw"], rows=N * S, ldy=ldg, seg=(segw, geom.c_segs, S), x_alt=(sv["src"], 3 * mlp))
        ops.colsum(segw, ldg, g("self_attn.sampling_offsets.bias"), 1, geom.L, 2 * mlp)
        ops.colsum(segw[:, 2 * mlp:], ldg, g("self_attn.attention_weights.bias"), 1, geom.L, mlp)
        ops.colsum(segw[:, 3 * mlp:], ldg, g("self_attn.value_proj.bias"), 1, geom.L, d)
        if g_level is not None:               # d(level_embed)[l] += colsum_l(dOA) @ [W_so ; W_aw]   (pos = sine + level_embed, q = src + pos)
            aw_w = P_["self_attn.attention_weights.weight"]
            ops.gemm(segw, so_w, g_level, geom.L, d, 2 * mlp, lda=ldg, ldb=d, ldc=d, b_kmajor=True, add_src=g_level, ld_add=d)
            ops.gemm(segw[:, 2 * mlp:], aw_w, g_level, geom.L, d, mlp, lda=ldg, ldb=d, ldc=d, b_kmajor=True, add_src=g_level, ld_add=d)
        ops.linear_dx(G2, tri["w16"], dsrc, rows=N * S, add_src=dsrc)
        return dsrc
    seg = ops.zeros((geom.L, 3 * mlp), torch.float32, dx2.device)
    sample_bwd(d_out_m, sv["q"], sv["OA"], so_w, P_["self_attn.attention_weights.weight"],
               sv["V"], geom, ref, ref_bs, N, S, M, D, npts, dV,
               g_so_w, g("self_attn.sampling_offsets.bias"),
               g("self_attn.attention_weights.weight"), g("self_attn.attention_weights.bias"),
               None if G2 is not None else (dsrc if dpos is None else dpos), dpos is None, seg_sums=seg, grid_queries=True,
               dOA=None if G2 is None else G2[:, : 3 * mlp])
    if dpos is not None:                      # d(q) went to its own buffer: q = src + pos, so it is d(pos) and one more term of d(src)
        ops.add(dsrc, dpos, dsrc)
    # d(level_embed)[l] += colsum_l(dOA) @ [W_so ; W_aw]   (pos = sine + level_embed, q = src + pos)
    if g_level is not None:
        so_w, aw_w = P_["self_attn.sampling_offsets.weight"], P_["self_attn.attention_weights.weight"]
        ops.gemm(seg, so_w, g_level, geom.L, d, 2 * mlp, lda=3 * mlp, ldb=d, ldc=d, b_kmajor=True, add_src=g_level, ld_add=d)
        ops.gemm(seg[:, 2 * mlp:], aw_w, g_level, geom.L, d, mlp, lda=3 * mlp, ldb=d, ldc=d, b_kmajor=True, add_src=g_level, ld_add=d)
    vact = torch.bfloat16 if sv["V"].dtype == torch.float16 else sv["V"].dtype      # (gradient rows of an fp16 map are bf16)
    if G2 is not None:
        value_proj_bwd(dV, sv["src"], P_["self_attn.value_proj.weight"], mask, N, S, M, D,
                       g("self_attn.value_proj.weight"), g("self_attn.value_proj.bias"), None, True, vact, dVr=G2[:, 3 * mlp:])
        ops.linear_dx(G2, tri["w16"], dsrc, rows=N * S, add_src=dsrc)
    else:
        value_proj_bwd(dV, sv["src"], P_["self_attn.value_proj.weight"], mask, N, S, M, D,
                       g("self_attn.value_proj.weight"), g("self_attn.value_proj.bias"), dsrc, True, vact)
    return dsrc


# ---- decoder layer ---------------------------------------------------------------------------------
DEC_PARAMS = ("cross_attn.sampling_offsets.weight", "cross_attn.sampling_offsets.bias",
              "cross_attn.attention_weights.weight", "cross_attn.attention_weights.bias",
              "cross_attn.value_proj.weight", "cross_attn.value_proj.bias",
              "cross_attn.output_proj.weight", "cross_attn.output_proj.bias",
              "norm1.weight", "norm1.bias",
              "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
              "norm2.weight", "norm2.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
              "norm3.weight", "norm3.bias")


def dec_layer_fwd(tgt, qpos, V, P_, ref_in, geom, N, Q, M, npts, p, training, y_out=None, ffn_act="relu"):
    """tgt,qpos (N*Q,d) fp32; V head-major value map of the encoder memory for this layer."""
    d = tgt.shape[1]
    D, rows = d // M, N * Q
    pd = p if training else 0.0
    seeds = [next_seed() for _ in range(5)]
    Win, bin_ = P_["self_attn.in_proj_weight"], P_["self_attn.in_proj_bias"]
    qk = torch.empty_like(tgt)
    ops.add(tgt, qpos, qk)
    packed = empty((rows, 3 * d), torch.float32, tgt)
    ops.linear_fwd(qk, Win[: 2 * d], bin_[: 2 * d], packed, ldc=3 * d)
    ops.linear_fwd(tgt, Win[2 * d:], bin_[2 * d:], packed[:, 2 * d:], ldc=3 * d)
    att = empty((rows, d), torch.float32, tgt)
    ops.mha_fwd(packed, packed[:, d:], packed[:, 2 * d:], 3 * d, att, d, N, Q, M, D, pd, seeds[0])
    t1, _, ln2 = proj_ln_fwd(att, P_["self_attn.out_proj.weight"], P_["self_attn.out_proj.bias"], tgt,
                             P_["norm2.weight"], P_["norm2.bias"], pd, seeds[1])
    q2 = torch.empty_like(t1)
    ops.add(t1, qpos, q2)
    out_m, OA = sample_fwd(q2, P_["cross_attn.sampling_offsets.weight"], P_["cross_attn.sampling_offsets.bias"],
                           P_["cross_attn.attention_weights.weight"], P_["cross_attn.attention_weights.bias"],
                           V, geom, ref_in, Q * geom.L * 2, N, Q, M, D, npts)
    t2, _, ln1 = proj_ln_fwd(out_m, P_["cross_attn.output_proj.weight"], P_["cross_attn.output_proj.bias"], t1,
                             P_["norm1.weight"], P_["norm1.bias"], pd, seeds[2])
    t3, _, ffn = ffn_fwd(t2, t2, P_["linear1.weight"], P_["linear1.bias"], P_["linear2.weight"], P_["linear2.bias"],
                         P_["norm3.weight"], P_["norm3.bias"], pd, pd, seeds[3], seeds[4], y_out=y_out, fn=ffn_act)
    saved = dict(tgt=tgt, qk=qk, packed=packed, att=att, ln2=ln2, t1=t1, q2=q2, OA=OA, out_m=out_m, ln1=ln1, t2=t2,
                 ffn=ffn, seeds=seeds, pd=pd, V=V)
    return t3, saved


def dec_layer_bwd(dt3, sv, P_, G, pre, ref_in, geom, N, Q, M, npts, dV, dqp=None, dOA_out=None):
    """Returns d(tgt).  dV (fp32, zeroed) receives the value-map gradient of this layer.
    dqp (rows, d), optional: += d(query_pos) of this layer (learned query embeddings: q = tgt + query_pos enters the self-
    attention's q / k and the cross-attention's query, deformable_transformer.py:277-284); dOA_out: optional caller buffer that
    receives d(offsets | logits) (learned reference points need the offsets' gradient: d(loc) = d(offset) * (W_l, H_l))."""
    d = dt3.shape[1]
    D, rows = d // M, N * Q
    pd, seeds = sv["pd"], sv["seeds"]
    g = lambda n: G(pre + n)
    dt2 = ffn_bwd(dt3, sv["t2"], P_["linear1.weight"], P_["linear2.weight"], P_["norm3.weight"], sv["ffn"], pd, pd,
                  seeds[4], g("linear1.weight"), g("linear1.bias"), g("linear2.weight"), g("linear2.bias"),
                  g("norm3.weight"), g("norm3.bias"))
    dt1, d_out_m = proj_ln_bwd(dt2, sv["out_m"], P_["cross_attn.output_proj.weight"], P_["norm1.weight"], sv["ln1"], pd,
                               seeds[2], g("cross_attn.output_proj.weight"), g("cross_attn.output_proj.bias"),
                               g("norm1.weight"), g("norm1.bias"))
    dq2 = torch.empty_like(dt1) if dqp is not None else None
    sample_bwd(d_out_m, sv["q2"], sv["OA"], P_["cross_attn.sampling_offsets.weight"], P_["cross_attn.attention_weights.weight"],
               sv["V"], geom, ref_in, Q * geom.L * 2, N, Q, M, D, npts, dV,
               g("cross_attn.sampling_offsets.weight"), g("cross_attn.sampling_offsets.bias"),
               g("cross_attn.attention_weights.weight"), g("cross_attn.attention_weights.bias"),
               dt1 if dqp is None else dq2, dqp is None, dOA=dOA_out)
    if dqp is not None:                       # d(q2) goes to t1 AND to query_pos
        ops.add(dt1, dq2, dt1)
        ops.add(dqp, dq2, dqp)
    dtgt, datt = proj_ln_bwd(dt1, sv["att"], P_["self_attn.out_proj.weight"], P_["norm2.weight"], sv["ln2"], pd, seeds[1],
                             g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), g("norm2.weight"), g("norm2.bias"))
    packed = sv["packed"]
    dpk = torch.empty_like(packed)
    ops.mha_bwd(packed, packed[:, d:], packed[:, 2 * d:], 3 * d, datt, d, dpk, dpk[:, d:], dpk[:, 2 * d:], 3 * d, N, Q, M, D,
                pd, seeds[0])
    Win = P_["self_attn.in_proj_weight"]
    gWin, gbin = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
    ops.linear_dw(dpk, sv["qk"], gWin[: 2 * d], rows=rows, ldy=3 * d, db=gbin[: 2 * d])
    ops.linear_dw(dpk[:, 2 * d:], sv["tgt"], gWin[2 * d:], rows=rows, ldy=3 * d, db=gbin[2 * d:])
    # d(tgt) += d(q|k) W_qk + d(v) W_v: q = k = tgt + query_pos and v = tgt (deformable_transformer.py:277-278), and query_pos
    # has no gradient consumer, so the two products collapse into ONE GEMM over the packed in_proj_weight (K = 3d)
    if dqp is None:
        ops.linear_dx(dpk, Win, dtgt, rows=rows, add_src=dtgt)
    else:                                     # d(qk) = d(q|k) W_qk goes to tgt AND to query_pos; d(v) W_v to tgt only
        dqk = torch.empty_like(dtgt)
        ops.linear_dx(dpk, Win[: 2 * d], dqk, rows=rows, ldy=3 * d)
        ops.linear_dx(dpk[:, 2 * d:], Win[2 * d:], dtgt, rows=rows, ldy=3 * d, add_src=dtgt)
        ops.add(dtgt, dqk, dtgt)
        ops.add(dqp, dqk, dqp)
    return dtgt


# ---- pose heads -------------------------------------------------------------------------------------
def mlp3_fwd(h, Ws, bs):
    a = empty((h.shape[0], Ws[0].shape[0]), torch.float32, h)
    ops.linear_fwd(h, Ws[0], bs[0], a, act=1)
    b = empty((h.shape[0], Ws[1].shape[0]), torch.float32, h)
    ops.linear_fwd(a, Ws[1], bs[1], b, act=1)
    c = empty((h.shape[0], Ws[2].shape[0]), torch.float32, h)
    ops.linear_fwd(b, Ws[2], bs[2], c)
    return c, (a, b)


def uniform_stride(ts):
    """Element stride between consecutive tensors of a list when they are fp32, contiguous, equally shaped and equally
    spaced in memory (the per-decoder-layer copies of one head parameter inside the flat arena); None otherwise."""
    if len(ts) < 2:
        return None
    t0 = ts[0]
    step = ts[1].data_ptr() - t0.data_ptr()
    if step <= 0 or step % 4:
        return None
    for i, t in enumerate(ts):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != t0.shape or t.data_ptr() - t0.data_ptr() != i * step:
            return None
    return step // 4


def mlp3_fwd_batched(h, Ws, bs, sW, sb):
    """mlp3_fwd for `nb` INDEPENDENT heads of one shape in three launches: h (nb, rows, d); Ws[k] / bs[k] = layer k's weight
    / bias of head 0, head j's copy sW[k] / sb[k] elements further (pose_estimation_transformer.py:357-363 applies
    head l to decoder output l: nothing couples them)."""
    nb, rows, d = h.shape
    outs, x = [], h
    for k in range(3):
        n_out, k_in = Ws[k].shape
        y = empty((nb, rows, n_out), torch.float32, h)
        ops.gemm(x, Ws[k], y, rows, n_out, k_in, lda=k_in, ldb=k_in, ldc=n_out, bias=bs[k], act=1 if k < 2 else 0, batch=nb,
                 strideA=rows * k_in, strideB=sW[k], strideC=rows * n_out, stride_bias=sb[k])
        outs.append(y)
        x = y
    return outs[2], (outs[0], outs[1])


def mlp3_bwd_batched(dc, h, Ws, saved, gWs, gbs, sW, sgW, sgb, dh, accumulate, Ws_all):
    """Backward of mlp3_fwd_batched: per MLP layer one launch for all dW + db and one for all dX.  sW: head-to-head stride of
    the weights, sgW / sgb: of the gradient buffers (the arena packs parameters and gradients differently)."""
    a, b = saved
    nb, rows, d = h.shape
    acts = (h, a, b)
    dy = dc
    for k in (2, 1, 0):
        n_out, k_in = Ws[k].shape
        x = acts[k]
        ops.gemm(dy, x, gWs[k], n_out, k_in, rows, lda=n_out, ldb=k_in, ldc=k_in, a_kmajor=True, b_kmajor=True, atomic=True,
                 bias=gbs[k], batch=nb, strideA=rows * n_out, strideB=rows * k_in, strideC=sgW[k], stride_bias=sgb[k])
        if k > 0:
            dx = torch.empty_like(x)
            # ((n_classes+1) * 6 or * 3 wide outputs are outside the <= 1024-row kernel's K % 16 == 0: the generic tiled kernel takes
            # the batch as well -- one launch instead of one per decoder layer)
            if n_out % 16 == 0 or not _ENV_HEAD_LOOP:
                ops.gemm(dy, Ws[k], dx, rows, k_in, n_out, lda=n_out, ldb=k_in, ldc=k_in, b_kmajor=True, gate_ref=x, batch=nb,
                         strideA=rows * n_out, strideB=sW[k], strideC=rows * k_in)
            else:                                   # (POET_HEAD_DX_LOOP=1: one launch per decoder layer, A/B aid)
                for j in range(nb):
                    ops.linear_dx(dy[j], Ws_all[k][j], dx[j], rows=rows, gate_ref=x[j])
            dy = dx
        else:
            ops.gemm(dy, Ws[0], dh, rows, k_in, n_out, lda=n_out, ldb=k_in, ldc=k_in, b_kmajor=True,
                     add_src=dh if accumulate else None, ld_add=k_in, batch=nb, strideA=rows * n_out, strideB=sW[0], strideC=rows * k_in)


def mlp3_bwd(dc, h, Ws, saved, gWs, gbs, dh, accumulate):
    a, b = saved
    rows = h.shape[0]
    ops.linear_dw(dc, b, gWs[2], rows=rows, db=gbs[2])
    db_ = torch.empty_like(b)
    ops.linear_dx(dc, Ws[2], db_, rows=rows, gate_ref=b)
    ops.linear_dw(db_, a, gWs[1], rows=rows, db=gbs[1])
    da = torch.empty_like(a)
    ops.linear_dx(db_, Ws[1], da, rows=rows, gate_ref=a)
    ops.linear_dw(da, h, gWs[0], rows=rows, db=gbs[0])
    ops.linear_dx(da, Ws[0], dh, rows=rows, add_src=dh if accumulate else None)
