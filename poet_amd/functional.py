"""torch.autograd.Function wrappers: four coarse nodes (input_proj, encoder, decoder, heads) plus the
stand-alone MSDeformAttn node.  Each forward/backward is a hand-written kernel program
(poet_amd.blocks) -- autograd only carries the tensors across the few node boundaries."""
from __future__ import annotations

import os
from typing import List, Sequence

import torch

from . import blocks as B
from . import ops
from .engine import announce


def _pdict(names: Sequence[str], params: Sequence[torch.Tensor], prefix: str):
    n = len(prefix)
    return {k[n:]: p for k, p in zip(names, params) if k.startswith(prefix)}


class CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
        ops.cast(x.contiguous(), out)
        return out

    @staticmethod
    def backward(ctx, g):
        out = torch.empty(g.shape, dtype=ctx.src_dtype, device=g.device)
        ops.cast(g.contiguous(), out)
        return out, None


# ====================================================================================================
# Encoder: one autograd node per layer (a gradient bucket each: the layer's parameter gradients are complete -- and their
# all-reduce can start -- as soon as its backward program ends, main.py:280-283's DDP overlap)
# ====================================================================================================
class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the HIP GEMMs for a stand-alone fp32 nn.Linear (forward, dX, dW + db) -- the learned reference-point
    projection `self.reference_points(query_embed)` (deformable_transformer.py:157-158); parameter gradients go through the
    arena's views when there are any (GradSink), else back to autograd."""

    @staticmethod
    def forward(ctx, x, W, b):
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        y = torch.empty((x2.shape[0], W.shape[0]), dtype=torch.float32, device=x.device)
        ops.linear_fwd(x2, W.detach(), b.detach(), y)
        ctx.save_for_backward(x2)
        ctx.W, ctx.b, ctx.shape = W, b, x.shape
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        dy2 = dy.reshape(-1, W.shape[0]).contiguous().float()
        G = B.GradSink(["w", "b"], [W, b])
        ops.linear_dw(dy2, x2, G("w"), rows=x2.shape[0], db=G("b"))
        dx = torch.empty_like(x2)
        ops.linear_dx(dy2, W.detach(), dx, rows=x2.shape[0])
        return dx.view(ctx.shape), G.ret[0], G.ret[1]


def enc_bucket_tag(n_layers: int, i: int) -> str:
    """Bucket name of encoder layer i (engine._bucket_of): backward completes layer n-1 first, so it sorts first."""
    return f"2_encoder_{n_layers - 1 - i:02d}"


# The side channels of the encoder's streams live in the cfg dict all layers of ONE forward share (encoder_forward makes it):
#   cfg["_fstream"][i]: fp16 remainder of layer i's input stream, left by the forward of layer i - 1 (POET_FSTREAM_SPLIT=1);
#   cfg["_gstream"][i]: bf16 d(output) of layer i, left by the backward of layer i + 1.


class EncoderLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x16, q_in, pos2, level_embed, ref, mask, geom, cfg, idx, names, *params):
        """x (N*S,d) residual stream; x16 its GEMM-operand copy (non-differentiable; == x in the pure modes); q_in: x + pos when
        the previous layer's last LayerNorm already produced it (else None); pos2 (N*S,d); ref (N,S,L,2) fp32; mask (N*S) uint8
        or None; cfg dict(M,P,p,training,n_layers,act,split,N).  Returns (y, y16, q_next): q_next = y + pos for the next layer
        (an empty tensor when not produced)."""
        N, S = cfg["N"], geom.S
        ctx.need_pos, ctx.pos_dtype = pos2.requires_grad, pos2.dtype          # a learned position encoding (position_encoding.py:87-112): backward returns d(pos)
        P_ = _pdict(names, params, "")
        emit = idx + 1 < cfg["n_layers"]
        # The split residual stream (blocks.proj_ln_fwd: bf16 head = x16, fp16 remainder) travels from layer to layer beside autograd,
        # like the bf16 gradient stream in backward: autograd sees an unwritten fp32 placeholder of the stream's shape (graph
        # connectivity only), the next layer picks the remainder rows up here.  The last layer writes fp32 (`memory`).
        x_real = cfg.get("_fstream", {}).pop(idx, None)
        out = B.enc_layer_fwd(x if x_real is None else x_real, x16, pos2, P_, ref, S * geom.L * 2, mask, geom, N, cfg["M"], cfg["P"], cfg["p"],
                              cfg["training"], cfg.get("act"), cfg.get("split", False), q_in=q_in, emit_q=emit, ffn_act=cfg.get("ffn_act", "relu"),
                              stream_out16=emit)
        y, y16, sv = out[:3]
        if y.dtype == torch.float16:
            cfg.setdefault("_fstream", {})[idx + 1] = y
            y = torch.empty(y.shape, dtype=torch.float32, device=y.device)            # placeholder: never written, never read
        q_next = out[3] if emit and out[3] is not None else x.new_empty(0)
        ctx.saved, ctx.geom, ctx.cfg, ctx.idx, ctx.names, ctx.params = sv, geom, cfg, idx, names, params
        ctx.ref, ctx.mask, ctx.level_embed = ref, mask, level_embed
        ctx.need_x = x.requires_grad
        if y16 is y or y16.data_ptr() == y.data_ptr():
            y16 = y.detach()
        ctx.mark_non_differentiable(y16, q_next)
        ctx.set_materialize_grads(False)                  # else autograd zero-fills two (N*S,d) bf16 gradients per layer for y16 / q_next
        return y, y16, q_next

    @staticmethod
    def backward(ctx, dy, _unused=None, _unused2=None):
        cfg, geom, names, params, i = ctx.cfg, ctx.geom, ctx.names, ctx.params, ctx.idx
        N, S = cfg["N"], geom.S
        if dy is None:
            raise RuntimeError("EncoderLayerFn.backward: no gradient for the layer output")
        G = B.GradSink(list(names) + ["level_embed"], list(params) + [ctx.level_embed])
        g_level = G("level_embed") if ctx.level_embed.requires_grad else None
        dpos = torch.empty((N * S, dy.shape[1]), dtype=torch.float32, device=dy.device) if ctx.need_pos else None
        # The bf16 gradient stream (blocks.enc_layer_bwd) is handed from layer to layer beside autograd: the layer above left its
        # d(input) here and gave autograd an unwritten fp32 placeholder of the same shape (autograd would cast a bf16 gradient of an
        # fp32 tensor back to fp32: a 156 MB pass per layer).  Layer 0 converts once for the input projection.
        dy_real = cfg.get("_gstream", {}).pop(i, None)
        if dy_real is None:
            dy_real = dy.contiguous()
        dx = B.enc_layer_bwd(dy_real, ctx.saved, _pdict(names, params, ""), G, "", ctx.ref, S * geom.L * 2, ctx.mask, geom,
                             N, cfg["M"], cfg["P"], g_level, dpos=dpos, out_f32=(i == 0 or not ctx.need_x))
        if dx.dtype != torch.float32:
            if i > 0 and ctx.need_x:
                cfg.setdefault("_gstream", {})[i - 1] = dx
                dx = torch.empty(dx.shape, dtype=torch.float32, device=dx.device)        # placeholder: never written, never read
            else:
                dx32 = torch.empty(dx.shape, dtype=torch.float32, device=dx.device)
                ops.cast(dx, dx32)
                dx = dx32
        ctx.saved = None
        ops.SIDE.join()
        announce(enc_bucket_tag(cfg["n_layers"], i))
        if i == 0:
            announce("2_encoder_99")                      # level_embed (+ anything of the encoder outside its layers)
        if dpos is not None and dpos.dtype != ctx.pos_dtype:
            dpos = dpos.to(ctx.pos_dtype)
        return (dx if ctx.need_x else None, None, None, dpos, G.ret[-1], None, None, None, None, None, None, *G.ret[:-1])


_ADD_CAST = __import__("os").environ.get("POET_NO_ADD_CAST", "0") in ("", "0")       # (A/B aid, read at import)


def encoder_forward(src, pos, level_embed, ref, mask, geom, cfg, layers_named):
    """src (N,S,d) residual-stream dtype; pos (N,S,d).  layers_named: per layer (names, params).  Returns (memory, memory16,
    per-layer stream tensors [input, out_0, ..., out_{n-1}]) -- the handles the graphed trainer segments backward at."""
    N, S, d = src.shape
    x = src.reshape(N * S, d)
    act = cfg.get("act") or src.dtype
    pos2 = pos.reshape(N * S, d)
    q = None
    if act != x.dtype:
        x16 = torch.empty((N * S, d), dtype=act, device=src.device)
        if (_ADD_CAST and x.dtype == torch.float32 and act == torch.bfloat16 and pos2.dtype == torch.bfloat16 and pos2.is_contiguous() and not pos2.requires_grad
                and x.is_contiguous() and x.data_ptr() % 32 == 0 and pos2.data_ptr() % 16 == 0):
            # the operand copy of the stream and the first layer's `src + pos` in one pass over the fp32 stream
            q = torch.empty((N * S, d), dtype=act, device=src.device)
            ops.add_cast(x.detach(), pos2, q, x16)
        else:
            ops.cast(x.detach(), x16)
    else:
        x16 = x.detach()
    cfg = dict(cfg, N=N)
    outs = [x]
    for i, (names, params) in enumerate(layers_named):
        x, x16, q = EncoderLayerFn.apply(x, x16, q, pos2, level_embed, ref, mask, geom, cfg, i, names, *params)
        if q.numel() == 0:
            q = None
        outs.append(x)
    return x.view(N, S, d), x16.view(N, S, d), outs


# ====================================================================================================
# Decoder: memory value projections + all layers in one node
# ====================================================================================================
class QueryEmbedFn(torch.autograd.Function):
    """Learned query embeddings (pose_estimation_transformer.py:149-150,342-343; deformable_transformer.py:150-155): the rows of
    nn.Embedding.weight (Q, 2d) split into (query_pos | tgt) and repeated for every image.  A node of its own so that the
    parameter's gradient -- the batch sums of d(query_pos) and d(tgt) -- is written straight into its arena view like every
    other parameter gradient of the path (an AccumulateGrad node kept alive from the eager warm-up steps would run on another
    stream and break the HIP-graph capture of backward)."""

    @staticmethod
    def forward(ctx, weight, N):
        d = weight.shape[1] // 2
        ctx.weight = weight
        w = weight.detach()
        return w[:, :d][None].expand(N, -1, -1).contiguous(), w[:, d:][None].expand(N, -1, -1).contiguous()

    @staticmethod
    def backward(ctx, dqpos, dtgt):
        G = B.GradSink(["w"], [ctx.weight])
        gw = G("w")
        d = gw.shape[1] // 2
        if dqpos is not None:
            gw[:, :d] += dqpos.sum(0)
        if dtgt is not None:
            gw[:, d:] += dtgt.sum(0)
        return G.ret[0], None


class PosEmbedFn(torch.autograd.Function):
    """Learned position encoding (models/position_encoding.py:87-112) as token rows: pos[n, start_l + y W_l + x] =
    [col_embed[x] | row_embed[y]] + level_embed[l] for every image.  A node of its own for the reason QueryEmbedFn is one: the
    tables' gradients -- d(pos) summed over the batch, then over the rows / columns of every level -- go straight into their
    gradient views (arena or autograd) instead of through AccumulateGrad nodes."""

    @staticmethod
    def forward(ctx, row_w, col_w, level_embed, shapes, N, dtype):
        ctx.row_w, ctx.col_w, ctx.shapes = row_w, col_w, shapes
        rw, cw = row_w.detach(), col_w.detach()
        rows = []
        for l, (h, w) in enumerate(shapes):
            if h > rw.shape[0] or w > cw.shape[0]:
                raise ValueError(f"PositionEmbeddingLearned: a {h} x {w} feature map exceeds the {rw.shape[0]}-entry tables "
                                 "(models/position_encoding.py:93-94)")
            rows.append(torch.cat([cw[:w][None].expand(h, -1, -1), rw[:h][:, None].expand(-1, w, -1)], -1).reshape(h * w, -1) + level_embed[l])
        return torch.cat(rows, 0)[None].expand(N, -1, -1).to(dtype).contiguous()

    @staticmethod
    def backward(ctx, dpos):
        G = B.GradSink(["row", "col"], [ctx.row_w, ctx.col_w])
        gr, gc = G("row"), G("col")
        F_ = gc.shape[1]
        d = dpos.float().sum(0)                                   # (S, 2F)
        o = 0
        for h, w in ctx.shapes:
            blk = d[o:o + h * w].view(h, w, 2 * F_)
            gc[:w] += blk[..., :F_].sum(0)
            gr[:h] += blk[..., F_:].sum(1)
            o += h * w
        return G.ret[0], G.ret[1], None, None, None, None


_WH_CACHE = {}


def _level_wh(geom, device):
    """(L, 2) fp32 tensor of (W_l, H_l) on `device`, cached per geometry: the upload happens in the eager warm-up steps, never
    inside a graph capture (a host -> device copy there is illegal)."""
    key = (geom.key(), str(device))
    t = _WH_CACHE.get(key)
    if t is None:
        t = _WH_CACHE[key] = torch.tensor([[w, h] for h, w in geom.shapes], dtype=torch.float32, device=device)
    return t


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, memory, memory16, tgt, qpos, ref_in, mask, geom, cfg, names, *params):
        """memory (N,S,d): differentiable handle in the residual-stream dtype; memory16: the copy the value projections
        actually read (bf16 in the bf16 policy); tgt,qpos (N,Q,d) fp32; ref_in (N,Q,L,2) fp32.
        Returns hs (n_layers,N,Q,d) fp32."""
        N, S, d = memory.shape
        Q, M = tgt.shape[1], cfg["M"]
        D = d // M
        mem2 = memory16.reshape(N * S, d)
        x = tgt.reshape(N * Q, d).contiguous()
        qp = qpos.reshape(N * Q, d).contiguous()
        nl = cfg["n_layers"]
        hs = torch.empty((nl, N, Q, d), dtype=torch.float32, device=memory.device)
        saved = []
        # value maps of all layers from ONE projection of the encoder memory when the arena keeps the layers' value_proj
        # parameters adjacent (engine.ParamArena._link_value_stack): memory is read once instead of n_layers times
        vs = getattr(_pdict(names, params, "layers.0.")["cross_attn.value_proj.weight"], "_vstack", None)
        if vs is not None and vs["n"] != nl:
            vs = None
        V_all = None
        if vs is not None:
            Wt, sp = B.Wf(vs["w"], mem2, cfg.get("split", False))
            if not sp:
                Wt = vs["w16"] if mem2.dtype == torch.bfloat16 else vs["w"]
            V_all = B.empty((N, M * nl, S, D), cfg.get("act") or mem2.dtype, mem2)
            ops.linear_fwd(mem2, Wt, vs["b"], V_all, row_mask=mask, head_major=(M * nl, S, D), split=sp)
        for i in range(nl):
            P_ = _pdict(names, params, f"layers.{i}.")
            if V_all is not None:
                V = V_all[:, i * M:(i + 1) * M]
            else:
                V = B.value_proj_fwd(mem2, P_["cross_attn.value_proj.weight"], P_["cross_attn.value_proj.bias"], mask, N, S, M, D,
                                     cfg.get("act"), cfg.get("split", False))
            # the layer's last LayerNorm writes its row of hs directly (hs[i] is also the next layer's input)
            x, sv = B.dec_layer_fwd(x, qp, V, P_, ref_in, geom, N, Q, M, cfg["P"], cfg["p"], cfg["training"], y_out=hs[i].view(N * Q, d),
                                    ffn_act=cfg.get("ffn_act", "relu"))
            saved.append(sv)
        ctx.vstack = vs if V_all is not None else None
        ctx.saved, ctx.geom, ctx.cfg, ctx.names, ctx.params = saved, geom, cfg, names, params
        ctx.mem2, ctx.ref_in, ctx.mask, ctx.dims = mem2, ref_in, mask, (N, S, d, Q)
        ctx.mem_dtype = memory.dtype
        ctx.need_mem, ctx.need_tgt = memory.requires_grad, tgt.requires_grad
        ctx.need_qpos, ctx.need_ref = qpos.requires_grad, ref_in.requires_grad       # learned query embeddings / reference points
        return hs

    @staticmethod
    def backward(ctx, dhs):
        N, S, d, Q = ctx.dims
        cfg, geom, names, params = ctx.cfg, ctx.geom, ctx.names, ctx.params
        M = cfg["M"]
        D = d // M
        G = B.GradSink(names, params)
        dhs = dhs.contiguous()
        dmem = torch.empty((N * S, d), dtype=ctx.mem_dtype, device=dhs.device) if ctx.need_mem else None
        dx = None
        first = True
        nl = cfg["n_layers"]
        vs = ctx.vstack
        # bf16 policy with stacked value projections: the scatters of all layers write bf16 token-major rows (N*S, nl*M*D) in place --
        # what the stacked weight / input-gradient GEMMs read; no fp32 head-major staging map (522 MB zero-fill + transposing pass)
        rows16 = vs is not None and ctx.mem2.dtype == torch.bfloat16 and os.environ.get("POET_DEC_DV_F32", "0") in ("", "0")
        dVr_all = ops.zeros((N * S, nl * M * D), torch.bfloat16, dhs.device) if rows16 else None
        dV_all = ops.zeros((N, M * nl, S, D), torch.float32, dhs.device) if (vs is not None and not rows16) else None
        dqp = ops.zeros((N * Q, d), torch.float32, dhs.device) if ctx.need_qpos else None
        mlp = M * geom.L * cfg["P"]
        dref = ops.zeros((N, Q, geom.L, 2), torch.float32, dhs.device) if ctx.need_ref else None
        if ctx.need_ref:
            wh = _level_wh(geom, dhs.device)                                                                   # (L, 2): x scales with W, y with H
        with ops.defer_small_dw() as deferred:           # the 320-row dW + db of all layers: one launch per Linear, at the end
            for i in reversed(range(nl)):
                deferred.next_layer()
                pre = f"layers.{i}."
                P_ = _pdict(names, params, pre)
                dcur = dhs[i].view(N * Q, d)
                if dx is not None:
                    ops.add(dx, dcur, dx)
                else:
                    dx = dcur.clone()
                if rows16:
                    dV = dVr_all.view(N, S, nl, M, D)[:, :, i].permute(0, 2, 1, 3)      # (N, M, S, D) view of this layer's column block
                else:
                    dV = dV_all[:, i * M:(i + 1) * M] if vs is not None else ops.zeros((N, M, S, D), torch.float32, dhs.device)
                # (a fresh buffer per layer: the deferred weight-gradient launches read it after the loop)
                dOA = torch.empty((N * Q, 3 * mlp), dtype=torch.float32, device=dhs.device) if ctx.need_ref else None
                dx = B.dec_layer_bwd(dx, ctx.saved[i], P_, G, pre, ctx.ref_in, geom, N, Q, M, cfg["P"], dV, dqp=dqp, dOA_out=dOA)
                if ctx.need_ref:
                    # loc = ref_in + offset / (W_l, H_l): d(ref_in)[n, q, l] = (W_l, H_l) * sum over heads and points of d(offset)
                    # -- 320 rows x 512 offsets: elementwise torch on the kernel's d(offsets) block, as for the quaternion extras
                    dref += dOA[:, : 2 * mlp].view(N, Q, M, geom.L, cfg["P"], 2).sum((2, 4)) * wh
                if vs is None:
                    B.value_proj_bwd(dV, ctx.mem2, P_["cross_attn.value_proj.weight"], ctx.mask, N, S, M, D,
                                     G(pre + "cross_attn.value_proj.weight"), G(pre + "cross_attn.value_proj.bias"), dmem, not first,
                                     cfg.get("act"))
                first = False
                ctx.saved[i] = None
        if vs is not None:                                # one stacked backward for the value projections of all layers
            B.value_proj_bwd(dV_all, ctx.mem2, vs["w16"] if ctx.mem2.dtype == torch.bfloat16 else vs["w"], ctx.mask, N, S, M * nl, D,
                             vs["gw"], vs["gb"], dmem, False, cfg.get("act"), wb_is_operand=True, dVr=dVr_all)
        ops.SIDE.join()
        announce("1_decoder")
        dmemory = dmem.view(N, S, d) if ctx.need_mem else None
        dtgt = dx.view(N, Q, d) if ctx.need_tgt else None
        dqpos = dqp.view(N, Q, d) if ctx.need_qpos else None
        return (dmemory, None, dtgt, dqpos, dref, None, None, None, None, *G.ret)


# ====================================================================================================
# Pose heads: per decoder layer two 3-layer MLPs + class-slot gather + 6D->R
# ====================================================================================================
def _head_stack(names, params, prefix, nl, grads=None):
    """Layer-0 tensors and uniform per-decoder-layer strides of one head family's 3 weights and 3 biases (None when the
    copies are not equally spaced, e.g. outside the flat arena).  grads: a GradSink -> the same for the gradient buffers."""
    Ws, bs, sW, sb = [], [], [], []
    for k in range(3):
        for kind, lst, strides in (("weight", Ws, sW), ("bias", bs, sb)):
            ts = [(_pdict(names, params, f"{prefix}{i}.layers.")[f"{k}.{kind}"] if grads is None else grads(f"{prefix}{i}.layers.{k}.{kind}"))
                  for i in range(nl)]
            st = B.uniform_stride(ts)
            if st is None:
                return None
            lst.append(ts[0])
            strides.append(st)
    return Ws, bs, sW, sb


_BATCH_HEADS = os.environ.get("POET_GEMM_NO_SMALL", "0") in ("", "0") and os.environ.get("POET_NO_BATCHED_HEADS", "0") in ("", "0")


class HeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hs, cls, ncls, names, *params):
        """hs (L,N,Q,d) fp32; cls (N*Q) int32.  Returns rot (L,N,Q,3,3), trans (L,N,Q,3).
        The heads of different decoder layers are independent (pose_estimation_transformer.py:357-363): when their
        parameters sit equally spaced in the flat arena, every MLP layer of all L heads is ONE batched launch (7 launches
        instead of 7 L forward, 13 instead of 13 L backward)."""
        nl, N, Q, d = hs.shape
        R = N * Q
        hs = hs.contiguous()
        rot = torch.empty((nl, N, Q, 3, 3), dtype=torch.float32, device=hs.device)
        trans = torch.empty((nl, N, Q, 3), dtype=torch.float32, device=hs.device)
        ctx.batched = None
        # (the batch-capable kernels are the <= 1024-row fp32 ones: larger per-GPU batches take the per-layer loop)
        st_r = _head_stack(names, params, "rotation_head.", nl) if (_BATCH_HEADS and nl > 1 and R <= 1024 and d % 16 == 0) else None
        st_t = _head_stack(names, params, "translation_head.", nl) if st_r is not None else None
        if st_t is not None:
            h = hs.view(nl, R, d)
            r_all, rs = B.mlp3_fwd_batched(h, st_r[0], st_r[1], st_r[2], st_r[3])
            t_all, ts = B.mlp3_fwd_batched(h, st_t[0], st_t[1], st_t[2], st_t[3])
            cls_rep = cls.repeat(nl)
            ops.pose_finish_fwd(r_all, t_all, cls_rep, rot, trans, nl * R, ncls)
            ctx.batched = (r_all, rs, ts, cls_rep, st_r, st_t)
            saved = None
        else:
            saved = []
            for i in range(nl):
                h = hs[i].view(R, d)
                Pr = _pdict(names, params, f"rotation_head.{i}.layers.")
                Pt = _pdict(names, params, f"translation_head.{i}.layers.")
                r_all, rs = B.mlp3_fwd(h, [Pr[f"{k}.weight"] for k in range(3)], [Pr[f"{k}.bias"] for k in range(3)])
                t_all, ts = B.mlp3_fwd(h, [Pt[f"{k}.weight"] for k in range(3)], [Pt[f"{k}.bias"] for k in range(3)])
                ops.pose_finish_fwd(r_all, t_all, cls, rot[i], trans[i], R, ncls)
                saved.append((r_all, rs, ts))
        ctx.saved, ctx.hs, ctx.cls, ctx.ncls, ctx.names, ctx.params = saved, hs, cls, ncls, names, params
        ctx.do_announce = True
        return rot, trans

    @staticmethod
    def backward(ctx, drot, dtrans):
        hs, cls, ncls, names, params = ctx.hs, ctx.cls, ctx.ncls, ctx.names, ctx.params
        nl, N, Q, d = hs.shape
        R = N * Q
        G = B.GradSink(names, params)
        drot, dtrans = drot.contiguous(), dtrans.contiguous()
        dhs = torch.empty_like(hs)
        gst_r = gst_t = None
        if ctx.batched is not None:
            gst_r = _head_stack(names, params, "rotation_head.", nl, grads=G)
            gst_t = _head_stack(names, params, "translation_head.", nl, grads=G) if gst_r is not None else None
        if gst_t is not None:
            r_all, rs, ts, cls_rep, st_r, st_t = ctx.batched
            dr_all = torch.empty_like(r_all)
            dt_all = torch.empty((nl, R, ncls * 3), dtype=torch.float32, device=hs.device)
            ops.pose_finish_bwd(r_all, cls_rep, drot, dtrans, dr_all, dt_all, nl * R, ncls)
            h = hs.view(nl, R, d)
            dh = dhs.view(nl, R, d)
            allw = lambda pre: [[_pdict(names, params, f"{pre}{i}.layers.")[f"{k}.weight"] for i in range(nl)] for k in range(3)]
            B.mlp3_bwd_batched(dr_all, h, st_r[0], rs, gst_r[0], gst_r[1], st_r[2], gst_r[2], gst_r[3], dh, False, allw("rotation_head."))
            B.mlp3_bwd_batched(dt_all, h, st_t[0], ts, gst_t[0], gst_t[1], st_t[2], gst_t[2], gst_t[3], dh, True, allw("translation_head."))
        else:
            for i in range(nl):
                h = hs[i].view(R, d)
                if ctx.batched is not None:          # forward ran batched but the gradient buffers are not equally spaced
                    r_all, (ra, rb), (ta, tb) = ctx.batched[0][i], ctx.batched[1], ctx.batched[2]
                    rs, ts = (ra[i], rb[i]), (ta[i], tb[i])
                else:
                    r_all, rs, ts = ctx.saved[i]
                pr, pt = f"rotation_head.{i}.layers.", f"translation_head.{i}.layers."
                Pr, Pt = _pdict(names, params, pr), _pdict(names, params, pt)
                dr_all = torch.empty_like(r_all)
                dt_all = torch.empty((R, ncls * 3), dtype=torch.float32, device=hs.device)
                ops.pose_finish_bwd(r_all, cls, drot[i], dtrans[i], dr_all, dt_all, R, ncls)
                dh = dhs[i].view(R, d)
                B.mlp3_bwd(dr_all, h, [Pr[f"{k}.weight"] for k in range(3)], rs, [G(pr + f"{k}.weight") for k in range(3)],
                           [G(pr + f"{k}.bias") for k in range(3)], dh, False)
                B.mlp3_bwd(dt_all, h, [Pt[f"{k}.weight"] for k in range(3)], ts, [G(pt + f"{k}.weight") for k in range(3)],
                           [G(pt + f"{k}.bias") for k in range(3)], dh, True)
        ops.SIDE.join()
        if ctx.do_announce:
            announce("0_heads")
        return (dhs, None, None, None, *G.ret)


class HeadsQuietFn(HeadsFn):
    """HeadsFn that leaves the bucket announcement to another head function of the same step (HeadsRawFn, applied BEFORE it
    in forward and therefore run AFTER it in backward: autograd orders independent nodes by descending sequence number)."""

    @staticmethod
    def forward(ctx, hs, cls, ncls, names, *params):
        out = HeadsFn.forward(ctx, hs, cls, ncls, names, *params)
        ctx.do_announce = False
        return out


class HeadsRawFn(torch.autograd.Function):
    """The 3-layer MLP heads named by `prefixes` (e.g. "rotation_head.", "translation_head_aleatoric.") of every decoder
    layer, WITHOUT the per-class slice / rotation post-processing: returns one (L, N*Q, out_dim) tensor per prefix.  Serves
    the representations the fused head kernel does not cover (pose_estimation_transformer.py:85-96,357-384: quaternion
    heads, aleatoric log-variance heads); the slice / normalisation are a handful of elementwise ops on 320 rows."""

    @staticmethod
    def forward(ctx, hs, prefixes, names, *params):
        nl, N, Q, d = hs.shape
        R = N * Q
        hs = hs.contiguous()
        outs, saved = [], []
        for pre in prefixes:
            per_layer, sv = [], []
            for i in range(nl):
                P = _pdict(names, params, f"{pre}{i}.layers.")
                o, st = B.mlp3_fwd(hs[i].view(R, d), [P[f"{k}.weight"] for k in range(3)], [P[f"{k}.bias"] for k in range(3)])
                per_layer.append(o)
                sv.append(st)
            outs.append(torch.stack(per_layer))
            saved.append(sv)
        ctx.saved, ctx.hs, ctx.prefixes, ctx.names, ctx.params = saved, hs, prefixes, names, params
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        hs, names, params = ctx.hs, ctx.names, ctx.params
        nl, N, Q, d = hs.shape
        R = N * Q
        G = B.GradSink(names, params)
        dhs = torch.empty_like(hs)
        for j, pre in enumerate(ctx.prefixes):
            dout = douts[j].contiguous()
            for i in range(nl):
                P = _pdict(names, params, f"{pre}{i}.layers.")
                pl = f"{pre}{i}.layers."
                B.mlp3_bwd(dout[i], hs[i].view(R, d), [P[f"{k}.weight"] for k in range(3)], ctx.saved[j][i],
                           [G(pl + f"{k}.weight") for k in range(3)], [G(pl + f"{k}.bias") for k in range(3)], dhs[i].view(R, d), j > 0)
        ops.SIDE.join()
        announce("0_heads")
        return (dhs, None, None, *G.ret)


# ====================================================================================================
# input_proj: 1x1 conv (+3x3 s2 conv for extra levels) + GroupNorm(32), written token-major
# ====================================================================================================
class InputProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, geom, n_groups, dtypes, names, *params):
        """feats: list of NCHW backbone maps (no grad).  Levels beyond len(feats) come from the 3x3 s2 conv of the
        last backbone map (first extra level) or of the previous projected map (further ones)."""
        N = feats[0].shape[0]
        d = params[0].shape[0]
        S = geom.S
        act_dtype, stream_dtype = dtypes[:2]      # branch (conv output) dtype, residual-stream dtype
        split = len(dtypes) > 2 and bool(dtypes[2]) and act_dtype == torch.bfloat16     # fp32 conv weights as bf16 hi + lo
        src = torch.empty((N, S, d), dtype=stream_dtype, device=feats[0].device)
        saved = []
        for lvl in range(geom.L):
            W, b = params[names.index(f"{lvl}.0.weight")], params[names.index(f"{lvl}.0.bias")]
            gw, gb = params[names.index(f"{lvl}.1.weight")], params[names.index(f"{lvl}.1.bias")]
            H, Wd = geom.shapes[lvl]
            HW = H * Wd
            pre = torch.empty((N, HW, d), dtype=act_dtype, device=src.device)
            if lvl < len(feats):
                f = feats[lvl].contiguous()
                C = f.shape[1]
                exact = split and f.dtype == torch.float32 and C % 8 == 0 and ops.inproj_exact()
                if (act_dtype == torch.bfloat16 and (exact or (N * HW >= 4096 and C % 128 == 0 and d % 128 == 0))
                        and os.environ.get("POET_INPROJ_NCHW_GEMM", "0") in ("", "0")):
                    # 1x1 conv = Linear over token rows: transpose + round the NCHW map once (the GEMM rounds it to bf16 anyway) and
                    # run the streaming kernels on row-major operands -- forward here, weight gradient in backward -- instead of
                    # the tiled kernel's K-major fp32 loads (94 us for the 118 MB of the 60 x 80 level)
                    if exact:
                        # the operand with 16 significant bits: rows [hi | lo] against [W | W] (K = 2 C, split weights), fp32
                        # accumulators kept up to the GroupNorm (the input projection was the second largest
                        # rounding site of the bf16 policy); the bf16 copy of the output is what backward reads
                        f2 = torch.empty((N * HW, 2 * C), dtype=torch.bfloat16, device=src.device)
                        ops.nchw_to_tokens_split(f, f2, N, C, HW)
                        pre32 = torch.empty((N, HW, d), dtype=torch.float32, device=src.device)
                        ops.linear_fwd(f2, torch.cat([W.view(d, C), W.view(d, C)], 1), b, pre32.view(N * HW, d), split=True)
                        stats = torch.empty((N, n_groups, 2), dtype=torch.float32, device=src.device)
                        ops.groupnorm_fwd(pre32, gw, gb, src, stats, N, HW, d, n_groups, 0, HW, geom.starts[lvl], S, x16=pre)      # (+ the bf16 copy backward reads)
                        saved.append((f2, None, pre, stats))
                        continue
                    f16 = torch.empty((N * HW, C), dtype=torch.bfloat16, device=src.device)
                    ops.nchw_to_tokens(f, f16, N, C, HW, 0, HW)
                    ops.linear_fwd(f16, W.view(d, C), b, pre.view(N * HW, d), split=split)
                    f = f16
                else:
                    ops.gemm(f, W, pre, HW, d, C, lda=HW, ldb=C, ldc=d, a_kmajor=True, bias=b, batch=N, strideA=C * HW, strideC=HW * d,
                             b_split=split, compute=ops.BF16 if split else None)
                col = None
            else:
                if lvl == len(feats):
                    inp = feats[-1].contiguous()
                else:       # further extra levels: previous projected level back to NCHW
                    Hp, Wp = geom.shapes[lvl - 1]
                    inp = torch.empty((N, d, Hp, Wp), dtype=act_dtype, device=src.device)
                    ops.tokens_to_nchw(src, inp, N, d, Hp * Wp, geom.starts[lvl - 1], S)
                C, Hi, Wi = inp.shape[1:]
                if split and inp.dtype == torch.float32 and C % 16 == 0 and 9 * C <= 4096 and ops.inproj_exact():
                    # (9 C <= 4096: gemm_small's K limit -- a real backbone's C = 512 .. 2048 takes the bf16 [hi | lo] path below instead of
                    # the generic fp32 tiled kernel with a very long K)
                    # the 3x3 stride-2 convolution of the extra level in fp32 end to end: a few hundred token rows (80 per image at
                    # 640 x 480) with K = 9 C -- the latency-oriented fp32 kernels of the decoder (gemm_small.hip: <= 1024 rows per
                    # call, the reduction split over the 4 waves of a workgroup) run it in ~30 us where the tiled bf16 kernel walks
                    # 18 (with [hi | lo] rows: 36) dependent K stages on 80 workgroups (78-146 us)
                    col32 = torch.empty((N * HW, C * 9), dtype=torch.float32, device=src.device)
                    ops.im2col3x3s2(inp, col32, N, C, Hi, Wi, H, Wd)
                    pre32 = torch.empty((N, HW, d), dtype=torch.float32, device=src.device)
                    p2, Wc = pre32.view(N * HW, d), W.view(d, C * 9)
                    for r0 in range(0, N * HW, 1024):
                        ops.linear_fwd(col32[r0:r0 + 1024], Wc, b, p2[r0:r0 + 1024])
                    col = torch.empty((N * HW, C * 9), dtype=torch.bfloat16, device=src.device)     # (backward's operand copy)
                    ops.cast(col32, col)
                    stats = torch.empty((N, n_groups, 2), dtype=torch.float32, device=src.device)
                    ops.groupnorm_fwd(pre32, gw, gb, src, stats, N, HW, d, n_groups, 0, HW, geom.starts[lvl], S, x16=pre)
                    saved.append((None, col, pre, stats))
                    continue
                col = torch.empty((N * HW, C * 9), dtype=act_dtype, device=src.device)
                ops.im2col3x3s2(inp, col, N, C, Hi, Wi, H, Wd)
                ops.linear_fwd(col, W.view(d, C * 9), b, pre, split=split)
                f = None
            stats = torch.empty((N, n_groups, 2), dtype=torch.float32, device=src.device)
            ops.groupnorm_fwd(pre, gw, gb, src, stats, N, HW, d, n_groups, 0, HW, geom.starts[lvl], S)
            saved.append((f, col, pre, stats))
        ctx.saved, ctx.geom, ctx.n_groups, ctx.names, ctx.params, ctx.dims = saved, geom, n_groups, names, params, (N, S, d)
        ctx.n_feats = len(feats)
        return src

    @staticmethod
    def backward(ctx, dsrc):
        geom, names, params = ctx.geom, ctx.names, ctx.params
        N, S, d = ctx.dims
        G = B.GradSink(names, params)
        dsrc = dsrc.contiguous()
        chained = geom.L > ctx.n_feats + 1        # a second (third, ...) extra level reads the previous PROJECTED level: its input
        if chained:                                # gradient joins that level's rows of d(src) before their GroupNorm backward runs
            dsrc = dsrc.clone()                    # (autograd owns the incoming tensor)
        for lvl in reversed(range(geom.L)):
            f, col, pre, stats = ctx.saved[lvl]
            H, Wd = geom.shapes[lvl]
            HW = H * Wd
            gw = params[names.index(f"{lvl}.1.weight")]
            dpre = torch.empty_like(pre)
            ops.groupnorm_bwd(dsrc, pre, stats, gw, dpre, G(f"{lvl}.1.weight"), G(f"{lvl}.1.bias"), N, HW, d, ctx.n_groups,
                              0, HW, geom.starts[lvl], S)
            gW = G(f"{lvl}.0.weight")
            if lvl > ctx.n_feats:                  # pose_estimation_transformer.py:327-330: input_proj[lvl](srcs[-1])
                W = params[names.index(f"{lvl}.0.weight")]
                Wc = getattr(W, "_bf16", None) if dpre.dtype == torch.bfloat16 else W
                if Wc is None:
                    Wc = W.to(torch.bfloat16)
                Hp, Wp = geom.shapes[lvl - 1]
                dcol = torch.empty((N * HW, d * 9), dtype=dpre.dtype, device=dpre.device)
                ops.linear_dx(dpre.view(N * HW, d), Wc.view(d, d * 9), dcol, rows=N * HW)
                ops.col2im3x3s2_add(dcol, dsrc, N, d, Hp, Wp, H, Wd, geom.starts[lvl - 1], S)
            if f is not None and f.dim() == 2:                 # token-major bf16 copy of the feature map (see forward): dW + db in one pass
                # ([hi | lo] rows: the weight gradient takes the hi half, row stride 2 C)
                ops.linear_dw(dpre.view(N * HW, d), f, gW.view(d, -1), rows=N * HW, db=G(f"{lvl}.0.bias"), ldx=f.shape[1])
                continue
            ops.colsum(dpre, d, G(f"{lvl}.0.bias"), 1, N * HW, d)
            if f is not None:
                C = f.shape[1]
                ops.gemm(dpre, f, gW, d, C, HW, lda=d, ldb=HW, ldc=C, a_kmajor=True, b_kmajor=False, batch=N,
                         strideA=HW * d, strideB=C * HW, strideC=0, atomic=True, splitk=max(1, min(8, HW // 512)))
            else:
                ops.linear_dw(dpre.view(N * HW, d), col, gW.view(d, -1), rows=N * HW, ldx=col.shape[1])     # ([hi | lo] rows: the hi half)
        ops.SIDE.join()
        announce("3_input_proj")
        return (None, None, None, None, None, *G.ret)


# ====================================================================================================
# Stand-alone MSDeformAttn (the drop-in for `deformable_attention.MSDeformAttn`)
# ====================================================================================================
MSDA_PARAMS = ("sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight", "attention_weights.bias",
               "value_proj.weight", "value_proj.bias", "output_proj.weight", "output_proj.bias")


class PoseLossFn(torch.autograd.Function):
    """losses (L, 2) = per-layer [translation, rotation] loss of the matched pairs (poet_pose_loss); the kernel also
    returns d losses / d predictions, so backward is two broadcast multiplies."""

    @staticmethod
    def forward(ctx, trans, rot, qi, tt, tr, n_obj):
        trans, rot = trans.contiguous(), rot.contiguous()
        L = trans.shape[0]
        losses = torch.empty((L, 2), dtype=torch.float32, device=trans.device)
        gt, gr = torch.empty_like(trans), torch.empty_like(rot)
        ops.pose_loss(trans, rot, qi, tt.contiguous(), tr.contiguous(), int(n_obj), losses, gt, gr)
        ctx.save_for_backward(gt, gr)
        return losses

    @staticmethod
    def backward(ctx, g):
        gt, gr = ctx.saved_tensors
        L = gt.shape[0]
        return (gt * g[:, 0].reshape(L, *([1] * (gt.dim() - 1))), gr * g[:, 1].reshape(L, *([1] * (gr.dim() - 1))),
                None, None, None, None)


class MSDeformAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, ref, inp, mask, geom, M, npts, *params):
        P_ = dict(zip(MSDA_PARAMS, params))
        N, Lq, d = query.shape
        S = inp.shape[1]
        D = d // M
        q2 = query.reshape(N * Lq, d).contiguous()
        in2 = inp.reshape(N * S, d).contiguous()
        refc = ref.contiguous().float()
        V = B.value_proj_fwd(in2, P_["value_proj.weight"], P_["value_proj.bias"], mask, N, S, M, D)
        out_m, OA = B.sample_fwd(q2, P_["sampling_offsets.weight"], P_["sampling_offsets.bias"], P_["attention_weights.weight"],
                                 P_["attention_weights.bias"], V, geom, refc, Lq * geom.L * 2, N, Lq, M, D, npts)
        out = torch.empty((N * Lq, d), dtype=query.dtype, device=query.device)
        ops.linear_fwd(out_m, P_["output_proj.weight"], P_["output_proj.bias"], out)
        ctx.sv = (q2, in2, refc, V, OA, out_m, mask, geom, M, npts, params, (N, Lq, S, d))
        ctx.need = (query.requires_grad, inp.requires_grad)
        return out.view(N, Lq, d)

    @staticmethod
    def backward(ctx, dout):
        q2, in2, refc, V, OA, out_m, mask, geom, M, npts, params, (N, Lq, S, d) = ctx.sv
        P_ = dict(zip(MSDA_PARAMS, params))
        D = d // M
        G = B.GradSink(MSDA_PARAMS, params)
        dy = dout.contiguous().view(N * Lq, d)
        rows = N * Lq
        ops.linear_dw(dy, out_m, G("output_proj.weight"), rows=rows, db=G("output_proj.bias"))
        d_out_m = torch.empty_like(out_m)
        ops.linear_dx(dy, P_["output_proj.weight"], d_out_m, rows=rows)
        dV = ops.zeros(V.shape, torch.float32, dy.device)
        dq = torch.empty_like(q2) if ctx.need[0] else None
        B.sample_bwd(d_out_m, q2, OA, P_["sampling_offsets.weight"], P_["attention_weights.weight"], V, geom, refc,
                     Lq * geom.L * 2, N, Lq, M, D, npts, dV, G("sampling_offsets.weight"), G("sampling_offsets.bias"),
                     G("attention_weights.weight"), G("attention_weights.bias"), dq, False)
        dinp = torch.empty_like(in2) if ctx.need[1] else None
        B.value_proj_bwd(dV, in2, P_["value_proj.weight"], mask, N, S, M, D, G("value_proj.weight"), G("value_proj.bias"), dinp, False)
        ops.SIDE.join()
        return (dq.view(N, Lq, d) if dq is not None else None, None, dinp.view(N, S, d) if dinp is not None else None,
                None, None, None, None, *G.ret)
