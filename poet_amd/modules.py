"""nn.Module mirror of the reference's hot-path classes, running on libpoet_hip.so.

Same constructor arguments, forward signatures and `state_dict` keys as
  deformable_attention.MSDeformAttn            (external op; models/deformable_transformer.py:24)
  models/deformable_transformer.py             DeformableTransformer{,Encoder,Decoder}{,Layer}
  models/position_encoding.py                  PositionEmbeddingSine, BoundingBoxEmbeddingSine
  models/pose_estimation_transformer.py        PoET, MLP
so reference checkpoints load with strict=True (minus backbone.*).  torch.nn.Linear / LayerNorm /
Conv2d / GroupNorm / MultiheadAttention objects appear below ONLY as parameter containers (names,
shapes, default init): their forward() is never called -- every FLOP runs in the HIP kernels, and
there is no CPU or eager fallback (CPU tensors raise).
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import functional as Fn
from . import ops
from .ops import LevelGeom


class NestedTensor:
    """util/misc.py:346-371 (two-field struct on the hot-path signature)."""

    def __init__(self, tensors, mask):
        self.tensors, self.mask = tensors, mask

    def decompose(self):
        return self.tensors, self.mask

    def to(self, device, non_blocking=False):
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking),
                            None if self.mask is None else self.mask.to(device, non_blocking=non_blocking))


def _named(module: nn.Module, prefix: str = ""):
    names, params = [], []
    for n, p in module.named_parameters():
        names.append(prefix + n)
        params.append(p)
    return tuple(names), params


def _u8(mask: Optional[torch.Tensor]):
    if mask is None:
        return None
    return mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous().to(torch.uint8)


# ====================================================================================================
class MSDeformAttn(nn.Module):
    """Drop-in for `deformable_attention.MSDeformAttn(d_model, n_levels, n_heads, n_points)`."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        geom = input_spatial_shapes if isinstance(input_spatial_shapes, LevelGeom) else LevelGeom(
            input_spatial_shapes.tolist())
        if reference_points.shape[-1] != 2:
            raise ValueError("Last dim of reference_points must be 2 (the 4-d variant is unreachable from PoET)")
        if geom.S != input_flatten.shape[1]:
            raise ValueError("sum(H*W) of input_spatial_shapes does not match input_flatten")
        params = [self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight,
                  self.attention_weights.bias, self.value_proj.weight, self.value_proj.bias,
                  self.output_proj.weight, self.output_proj.bias]
        mask = None if input_padding_mask is None else _u8(input_padding_mask).view(-1)
        return Fn.MSDeformAttnFn.apply(query, reference_points, input_flatten, mask, geom, self.n_heads, self.n_points, *params)


# ====================================================================================================
class PositionEmbeddingSine(nn.Module):
    """models/position_encoding.py:24-60 (normalize=True path); returns NCHW like the reference."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if not normalize or temperature != 10000 or (scale is not None and scale != 2 * math.pi):
            raise NotImplementedError("only the reference's build (normalize=True, T=1e4, scale=2pi) is implemented")
        self.num_pos_feats = num_pos_feats

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        N, H, W = mask.shape
        F_ = self.num_pos_feats
        tok = torch.empty((N, H * W, 2 * F_), dtype=torch.float32, device=mask.device)
        ops.pos_sine(_u8(mask), tok, None, N, H, W, F_, 0, H * W)
        out = torch.empty((N, 2 * F_, H, W), dtype=torch.float32, device=mask.device)
        ops.tokens_to_nchw(tok, out, N, 2 * F_, H * W, 0, H * W)
        return out


class PositionEmbeddingLearned(nn.Module):
    """models/position_encoding.py:87-112 (`--position_embedding learned`, position_encoding.py:115-127): [col_embed(x) | row_embed(y)]
    per pixel; nn.Embedding(50, .) tables, so feature maps up to 50 x 50.  `forward` returns NCHW like the reference; PoET itself
    takes `tokens(h, w)` -- the (h w, 2 F) token-major rows, differentiable -- and the encoder's backward returns d(pos) (the sum
    over layers of d(src + pos), blocks.enc_layer_bwd) so that the tables train."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def tokens(self, h: int, w: int):
        if h > self.row_embed.num_embeddings or w > self.col_embed.num_embeddings:
            raise ValueError(f"PositionEmbeddingLearned: a {h} x {w} feature map exceeds the {self.row_embed.num_embeddings}-entry tables "
                             "(models/position_encoding.py:93-94)")
        x_emb, y_emb = self.col_embed.weight[:w], self.row_embed.weight[:h]
        return torch.cat([x_emb[None].expand(h, -1, -1), y_emb[:, None].expand(-1, w, -1)], dim=-1).reshape(h * w, -1)

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors
        h, w = x.shape[-2:]
        return self.tokens(h, w).view(h, w, -1).permute(2, 0, 1)[None].expand(x.shape[0], -1, -1, -1).contiguous()


class BoundingBoxEmbeddingSine(nn.Module):
    """models/position_encoding.py:63-84."""

    def __init__(self, num_pos_feats=32):
        super().__init__()
        self.num_pos_feats = int(num_pos_feats)

    def forward(self, bboxes: torch.Tensor):
        n = bboxes.shape[0]
        out = torch.empty((n, 8 * self.num_pos_feats), dtype=torch.float32, device=bboxes.device)
        if n:
            ops.bbox_sine(bboxes.contiguous().float(), out, n, self.num_pos_feats)
        return out


def _check_activation(name):
    """deformable_transformer.py:347-355 (`_get_activation_fn`): "relu" / "gelu" / "glu", anything else raises RuntimeError at
    construction.  "glu" halves the hidden width, so the reference's own forward fails in linear2 (d_ffn / 2 columns against a
    d_ffn-wide weight): here it constructs and raises RuntimeError on the first forward as well (blocks.ffn_fwd)."""
    if name not in ("relu", "gelu", "glu"):
        raise RuntimeError(f"activation should be relu/gelu, not {name}.")
    return name


# ====================================================================================================
class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.activation = _check_activation(activation)
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    def run(self, src, pos, level_embed, ref, mask_u8, geom: LevelGeom, act=None, split=False):
        l0 = self.layers[0]
        cfg = dict(M=l0.self_attn.n_heads, P=l0.self_attn.n_points, p=l0.dropout1.p, training=self.training,
                   n_layers=self.num_layers, act=act, split=split, ffn_act=l0.activation)
        memory, memory16, self._layer_outs = Fn.encoder_forward(src, pos, level_embed, ref, mask_u8, geom, cfg,
                                                                [_named(l) for l in self.layers])
        return memory, memory16


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.activation = _check_activation(activation)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)      # parameter container only
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.bbox_embed = None
        self.class_embed = None

    def run(self, memory, memory16, tgt, qpos, ref_in, mask_u8, geom: LevelGeom, act=None, split=False):
        l0 = self.layers[0]
        cfg = dict(M=l0.cross_attn.n_heads, P=l0.cross_attn.n_points, p=l0.dropout1.p, training=self.training,
                   n_layers=self.num_layers, act=act, split=split, ffn_act=l0.activation)
        names, params = _named(self.layers, "layers.")
        return Fn.DecoderFn.apply(memory, memory16, tgt, qpos, ref_in, mask_u8, geom, cfg, names, *params)


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4,
                 dec_n_points=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = DeformableTransformerEncoder(
            DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead,
                                              enc_n_points), num_encoder_layers)
        self.decoder = DeformableTransformerDecoder(
            DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead,
                                              dec_n_points), num_decoder_layers, return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)      # unused in PoET's bbox mode (kept for state_dict parity)
        self.set_precision("fp32")
        self._reset_parameters()

    def set_precision(self, mode: str):
        """'fp32' : fp32 storage, v_mfma_f32_16x16x4_f32 -- the 1e-3 parity path.
        'bf16' : the training path.  bf16 storage + v_mfma_f32_16x16x32_bf16 for everything that feeds or leaves
                 a GEMM / the sampling kernel in the ENCODER (offsets+logits, value maps, sampled output, FFN
                 hidden, pre-norm branch), fp32 for the residual stream (LayerNorm outputs) and the whole
                 320-row decoder/head stream (only the value maps it samples are bf16); fp32 accumulate everywhere.
                 FORWARD weights enter every 102k-row GEMM (input_proj, the encoder's four Linears per layer, the decoder's
                 value projections) as bf16 hi + bf16 lo of the fp32 master (two MFMAs per fragment pair, PoetGemmDesc.b_split):
                 weight rounding is one fixed perturbation shared by all 6380 tokens of an image, so unlike per-token
                 activation rounding it does not average out in the decoder's sampling -- it was 4/5 of the rotation error.
        'bf16_nosplit' : the same with single bf16 weights (the round-1 policy; A/B and ablation).
        'bf16_pure' : residual stream in bf16 too, single bf16 weights (fastest, misses the 1e-2 bound at full size)."""
        self.split_w = False
        if mode == "fp32":
            self.act_dtype, self.stream_dtype = torch.float32, torch.float32
        elif mode in ("bf16", "bf16_nosplit"):
            self.act_dtype, self.stream_dtype = torch.bfloat16, torch.float32
            self.split_w = mode == "bf16"
        elif mode == "bf16_pure":
            self.act_dtype, self.stream_dtype = torch.bfloat16, torch.bfloat16
        else:
            raise ValueError(mode)
        self.precision = mode
        return self

    def _reset_parameters(self):
        """deformable_transformer.py:52-62."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        nn.init.constant_(self.reference_points.bias.data, 0.0)
        nn.init.normal_(self.level_embed)

    # ---- fused entry: everything already token-major ------------------------------------------------
    def forward_flat(self, src, pos, masks_u8: List[torch.Tensor], geom: LevelGeom, tgt, qpos, reference_points):
        """src,pos (N,S,d) act dtype (pos WITH level_embed added); masks_u8: per-level (N,H,W) uint8;
        tgt,qpos (N,Q,d) fp32; reference_points (N,Q,2) fp32.  Returns hs (n_dec,N,Q,d) fp32."""
        N, S, d = src.shape
        dev = src.device
        vr = torch.empty((N, geom.L, 2), dtype=torch.float32, device=dev)
        for l, (h, w) in enumerate(geom.shapes):
            ops.valid_ratio(masks_u8[l], vr[:, l], 2 * geom.L, N, h, w)
        mask_flat = torch.cat([m.reshape(N, -1) for m in masks_u8], 1).contiguous().view(-1)
        ref = torch.empty((N, S, geom.L, 2), dtype=torch.float32, device=dev)
        ops.enc_ref_points(vr, geom, ref, N)
        memory, memory16 = self.encoder.run(src, pos, self.level_embed, ref, mask_flat, geom, self.act_dtype, self.split_w)
        Q = tgt.shape[1]
        if reference_points is None:      # learned reference points (deformable_transformer.py:157-158): sigmoid(Linear(query_pos))
            reference_points = torch.sigmoid(Fn.LinearFn.apply(qpos, self.reference_points.weight, self.reference_points.bias))
        self._last_reference_points = reference_points
        if reference_points.requires_grad:  # differentiable form of dec_ref_points (deformable_transformer.py:317): 320 x L x 2 values
            ref_in = reference_points[:, :, None, :] * vr[:, None, :, :]
        else:
            ref_in = torch.empty((N, Q, geom.L, 2), dtype=torch.float32, device=dev)
            ops.dec_ref_points(reference_points.contiguous(), vr, ref_in, N, Q, geom.L)
        hs = self.decoder.run(memory, memory16, tgt, qpos, ref_in, mask_flat, geom, self.act_dtype, self.split_w)
        self._last_memory = memory
        return hs

    # ---- reference-compatible entry (deformable_transformer.py:120-166) -----------------------------
    def forward(self, srcs, masks, pos_embeds, query_embed=None, reference_points=None):
        assert query_embed is not None
        geom = LevelGeom([s.shape[-2:] for s in srcs])
        N, d = srcs[0].shape[0], srcs[0].shape[1]
        src = Fn_flatten(srcs, geom, self.stream_dtype)
        pos = torch.empty((N, geom.S, d), dtype=self.act_dtype, device=src.device)
        for l, p in enumerate(pos_embeds):
            h, w = geom.shapes[l]
            ops.nchw_to_tokens(p.contiguous(), pos, N, d, h * w, geom.starts[l], geom.S)
            ops.add_rowvec(pos, self.level_embed.detach()[l].contiguous(), N, geom.S, geom.starts[l], h * w, d)
        if self.level_embed.requires_grad:
            pass    # d(level_embed) is produced inside EncoderFn from the per-level sums (see blocks.enc_layer_bwd)
        if query_embed.dim() == 2:
            qpos, tgt = torch.split(query_embed, d, dim=1)
            qpos = qpos[None].expand(N, -1, -1)
            tgt = tgt[None].expand(N, -1, -1)
        else:
            qpos, tgt = torch.split(query_embed, d, dim=2)
        hs = self.forward_flat(src, pos, [_u8(m) for m in masks], geom, tgt.contiguous().float(),
                               qpos.contiguous().float(), None if reference_points is None else reference_points.float())
        reference_points = self._last_reference_points
        inter_refs = reference_points[None].expand(hs.shape[0], -1, -1, -1)
        return hs, reference_points, inter_refs, None, None


class _FlattenFn(torch.autograd.Function):
    """NCHW levels -> (N,S,d) token-major (deformable_transformer.py:128-141) and back for the gradient."""

    @staticmethod
    def forward(ctx, geom, act_dtype, *srcs):
        N, d = srcs[0].shape[:2]
        out = torch.empty((N, geom.S, d), dtype=act_dtype, device=srcs[0].device)
        for l, s in enumerate(srcs):
            h, w = geom.shapes[l]
            ops.nchw_to_tokens(s.contiguous(), out, N, d, h * w, geom.starts[l], geom.S)
        ctx.geom, ctx.meta = geom, [(s.shape, s.dtype) for s in srcs]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        grads = []
        for l, (shape, dt) in enumerate(ctx.meta):
            g = torch.empty(shape, dtype=dt, device=dout.device)
            ops.tokens_to_nchw(dout, g, shape[0], shape[1], shape[2] * shape[3], ctx.geom.starts[l], ctx.geom.S)
            grads.append(g)
        return (None, None, *grads)


def Fn_flatten(srcs, geom, act_dtype):
    return _FlattenFn.apply(geom, act_dtype, *srcs)


# ====================================================================================================
class MLP(nn.Module):
    """pose_estimation_transformer.py:677-689 (parameter container; run by Fn.HeadsFn)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        if num_layers != 3:
            raise NotImplementedError("PoET's heads are 3-layer MLPs")
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


def _to_host_list(tensors, dtype):
    """Per-image arrays on the host.  Device tensors are concatenated and copied with a single (blocking) transfer --
    a `.cpu()` per image would drain the GPU queue N times per step."""
    if not tensors:
        return []
    if all(torch.is_tensor(t) and t.is_cuda for t in tensors):
        sizes = [int(t.shape[0]) for t in tensors]
        width = [int(np.prod(t.shape[1:])) for t in tensors]          # explicit: reshape(0, -1) of an object-less image is ambiguous
        flat = torch.cat([t.detach().reshape(t.shape[0], w) for t, w in zip(tensors, width)], 0).cpu().numpy().astype(dtype, copy=False)
        out, o = [], 0
        for t, n in zip(tensors, sizes):
            out.append(flat[o:o + n].reshape((n,) + tuple(t.shape[1:])))
            o += n
        return out
    return [(t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)).astype(dtype, copy=False) for t in tensors]


class PoET(nn.Module):
    """pose_estimation_transformer.py:32-451 for bbox_mode in {'gt','jitter'} (training: queries from the targets) and
    'backbone' (inference: queries from the detector rows the backbone returns), rotation_mode '6d',
    class_mode in {'specific','agnostic'}, query/ref-point mode 'bbox' (the reference's defaults)."""

    def __init__(self, backbone, transformer, num_queries, num_feature_levels, n_classes, bbox_mode="gt",
                 ref_points_mode="bbox", query_embedding_mode="bbox", rotation_mode="6d", class_mode="agnostic",
                 aleatoric=False, aux_loss=True, backbone_type="yolo"):
        super().__init__()
        if bbox_mode not in ("gt", "jitter", "backbone"):
            raise NotImplementedError("PoET Bounding Box Mode not implemented!")
        if query_embedding_mode not in ("bbox", "learned"):
            raise NotImplementedError("This query embedding mode is not implemented.")
        if ref_points_mode not in ("bbox", "learned"):
            raise NotImplementedError("This reference point mode is not implemented.")
        self.ref_points_mode, self.query_embedding_mode = ref_points_mode, query_embedding_mode
        if rotation_mode not in ("6d", "quat", "silho_quat"):
            raise NotImplementedError("Rotational representation is not supported.")
        if aleatoric and rotation_mode != "6d":        # pose_estimation_transformer.py:72-73
            raise NotImplementedError("Aleatoric uncertainty estimation not implemented for quaternion rotation representation.")
        self.transformer = transformer
        d = transformer.d_model
        self.hidden_dim, self.backbone, self.backbone_type = d, backbone, backbone_type
        self.aux_loss, self.n_queries, self.n_classes = aux_loss, num_queries, n_classes + 1
        self.bbox_mode, self.class_mode, self.rotation_mode, self.aleatoric = bbox_mode, class_mode, rotation_mode, aleatoric
        self.t_dim, self.rot_dim, self.aleatoric_dim = 3, (6 if rotation_mode == "6d" else 4), 3
        mult = self.n_classes if class_mode == "specific" else 1
        self.num_feature_levels = num_feature_levels
        # construction order == RNG consumption order of pose_estimation_transformer.py:85-144
        self.translation_head = th = MLP(d, d, self.t_dim * mult, 3)    # registered first, replaced by ModuleLists
        self.rotation_head = rh = MLP(d, d, self.rot_dim * mult, 3)      # below: keeps the reference's parameter order
        if aleatoric:                                                    # :88-90,95-96: log-variance heads
            self.translation_head_aleatoric = tah = MLP(d, d, self.aleatoric_dim * mult, 3)
            self.rotation_head_aleatoric = rah = MLP(d, d, self.aleatoric_dim * mult, 3)
        n_bb = len(backbone.strides)
        projs, cin = [], None
        for n in range(n_bb):
            cin = backbone.num_channels[n]
            projs.append(nn.Sequential(nn.Conv2d(cin, d, kernel_size=1), nn.GroupNorm(32, d)))
        for _ in range(num_feature_levels - n_bb):
            projs.append(nn.Sequential(nn.Conv2d(cin, d, kernel_size=3, stride=2, padding=1), nn.GroupNorm(32, d)))
            cin = d
        self.input_proj = nn.ModuleList(projs)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        n_pred = transformer.decoder.num_layers
        self.translation_head = nn.ModuleList([copy.deepcopy(th) for _ in range(n_pred)])
        self.rotation_head = nn.ModuleList([copy.deepcopy(rh) for _ in range(n_pred)])
        if aleatoric:
            self.translation_head_aleatoric = nn.ModuleList([copy.deepcopy(tah) for _ in range(n_pred)])
            self.rotation_head_aleatoric = nn.ModuleList([copy.deepcopy(rah) for _ in range(n_pred)])
        self.bbox_embedding = BoundingBoxEmbeddingSine(num_pos_feats=d / 8)
        if query_embedding_mode == "learned":           # pose_estimation_transformer.py:149-150 (after the heads: RNG order)
            self.query_embed = nn.Embedding(num_queries, d * 2)

    # ---- query assembly (pose_estimation_transformer.py:203-239,309-311,337-338), batched on the host ----
    def host_queries(self, targets):
        """Pure host work: pad every image's boxes/labels to n_queries (dummy box -1, dummy class -1).
        Returns numpy (boxes (N,Q,4) f32, classes (N,Q) i64, valid (N,Q) u8) and n_boxes."""
        N, Q = len(targets), self.n_queries
        boxes = np.full((N, Q, 4), -1.0, np.float32)
        classes = np.full((N, Q), -1, np.int64)
        valid = np.zeros((N, Q), np.uint8)
        n_boxes = []
        key = "jitter_boxes" if self.bbox_mode == "jitter" else "boxes"
        bl = _to_host_list([t[key] for t in targets], np.float32)       # ONE device->host copy per field, not one per image
        cl = _to_host_list([t["labels"] for t in targets], np.int64)
        # the matcher works on the ground-truth boxes: hand it this host copy instead of letting it sync again
        self._tgt_boxes_host = bl if key == "boxes" else _to_host_list([t["boxes"] for t in targets], np.float32)
        self._tgt_labels_host, self._pred_classes_host = cl, classes       # ('jitter' matching is by class)
        for i, (b, c) in enumerate(zip(bl, cl)):
            nb = len(b)
            if nb > Q:
                raise ValueError(f"image {i} has {nb} boxes > num_queries={Q} (the reference assumes n <= Q in gt mode)")
            n_boxes.append(nb)
            boxes[i, :nb], classes[i, :nb], valid[i, :nb] = b, c, 1
        return boxes, classes, valid, n_boxes

    def host_queries_backbone(self, pred_objects, image_hw):
        """'backbone' mode (pose_estimation_transformer.py:240-305), on the host like host_queries: per image the detector
        rows (x0, y0, x1, y1, score, class) in pixels -> cxcywh normalised by the batch image size (util/box_ops.py:24-40;
        the reference uses image_sizes[0] for the whole batch); more rows than queries: the n_queries best scores, in
        descending order; fewer, or None: dummy padding (box -1, class -1)."""
        N, Q = len(pred_objects), self.n_queries
        ih, iw = float(image_hw[0]), float(image_hw[1])
        boxes = np.full((N, Q, 4), -1.0, np.float32)
        classes = np.full((N, Q), -1, np.int64)
        valid = np.zeros((N, Q), np.uint8)
        n_boxes = []
        have = [i for i, p in enumerate(pred_objects) if p is not None and len(p)]
        rows = dict(zip(have, _to_host_list([pred_objects[i] for i in have], np.float32)))   # one device->host copy for the batch
        for i in range(N):
            r = rows.get(i)
            if r is None:
                n_boxes.append(0)
                continue
            if len(r) > Q:
                r = r[np.argsort(-r[:, 4], kind="stable")[:Q]]
            nb = len(r)
            x0, y0, x1, y1 = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
            b = np.stack([(x0 + x1) / np.float32(2), (y0 + y1) / np.float32(2), x1 - x0, y1 - y0], -1)
            boxes[i, :nb] = b / np.array([iw, ih, iw, ih], np.float32)
            classes[i, :nb] = r[:, 5].astype(np.int64)
            valid[i, :nb] = 1
            n_boxes.append(nb)
        self._tgt_boxes_host = self._tgt_labels_host = None
        self._pred_classes_host = classes
        return boxes, classes, valid, n_boxes

    def forward_core(self, feats, feat_masks, image_mask, boxes, valid, classes):
        """Device-only part of forward (capturable in a hipGraph: no host sync, no host-dependent shapes).
        feats: list of NCHW maps; feat_masks: list of (N,h,w) uint8; image_mask (N,H,W) uint8;
        boxes (N,Q,4) f32; valid (N,Q) uint8; classes (N,Q) int64.  Returns (rot, trans, hs)."""
        tr = self.transformer
        act, stream = tr.act_dtype, tr.stream_dtype
        dev = feats[0].device
        N, Q = classes.shape
        emb = torch.empty((N * Q, self.hidden_dim), dtype=torch.float32, device=dev)
        ops.bbox_sine(boxes.view(N * Q, 4), emb, N * Q, self.hidden_dim // 8, valid=valid.view(-1), fill=-10.0)
        emb = emb.view(N, Q, -1)
        # per-level geometry and masks (extra levels: nearest resize of the image mask, :328-329)
        masks = list(feat_masks)
        shapes = [tuple(f.shape[-2:]) for f in feats]
        for lvl in range(len(feats), self.num_feature_levels):
            h, w = shapes[-1]
            shapes.append(((h - 1) // 2 + 1, (w - 1) // 2 + 1))
            m = torch.empty((N, *shapes[-1]), dtype=torch.uint8, device=dev)
            ops.mask_nearest(image_mask, m, N, image_mask.shape[1], image_mask.shape[2], *shapes[-1])
            masks.append(m)
        geom = LevelGeom(shapes)
        names, params = _named(self.input_proj)
        src = Fn.InputProjFn.apply(feats, geom, 32, (act, stream, tr.split_w), names, *params)
        self._last_src = src                          # autograd-node boundaries: the graphed trainer splits backward here
        lvl_embed = tr.level_embed.detach().contiguous()
        pe = self.backbone[1] if hasattr(self.backbone, "__getitem__") else None
        if isinstance(pe, PositionEmbeddingLearned):
            # learned encoding (position_encoding.py:87-112): token rows from the embedding tables (Fn.PosEmbedFn); every encoder layer
            # hands back d(pos), level_embed keeps its own gradient path (added here as a constant)
            pos = Fn.PosEmbedFn.apply(pe.row_embed.weight, pe.col_embed.weight, lvl_embed, tuple(geom.shapes), N, act)
        else:
            pos = torch.empty((N, geom.S, self.hidden_dim), dtype=act, device=dev)
            for l, (h, w) in enumerate(geom.shapes):
                ops.pos_sine(masks[l], pos, lvl_embed[l], N, h, w, self.hidden_dim // 2, geom.starts[l], geom.S)
        ref_pts = boxes[:, :, :2].contiguous() if self.ref_points_mode == "bbox" else None      # :337-340
        if self.query_embedding_mode == "learned":     # :342-343 and deformable_transformer.py:150-155: (query_pos | tgt) rows, same for every image
            qpos, tgt = Fn.QueryEmbedFn.apply(self.query_embed.weight, N)
            hs = tr.forward_flat(src, pos, masks, geom, tgt, qpos, ref_pts)
        else:
            hs = tr.forward_flat(src, pos, masks, geom, emb, emb, ref_pts)
        if self.class_mode == "specific":
            cls32 = classes.to(torch.int32).view(-1)
            ncls = self.n_classes
        else:
            cls32, ncls = torch.zeros(N * Q, dtype=torch.int32, device=dev), 1
        names, params = _named(self.translation_head, "translation_head.")
        n2, p2 = _named(self.rotation_head, "rotation_head.")

        def pick(raw):                                 # (L, N*Q, ncls*k) -> (L, N, Q, k): per query the slice of its class (:365-374)
            L = raw.shape[0]
            if ncls == 1:
                return raw.view(L, N, Q, -1)
            idx = cls32.clamp(min=0).long().view(1, N * Q, 1, 1).expand(L, N * Q, 1, raw.shape[-1] // ncls)
            return raw.view(L, N * Q, ncls, -1).gather(2, idx).view(L, N, Q, -1)

        self._last_aleatoric = None
        if self.rotation_mode != "6d":                 # quaternion heads: MLPs on the HIP path, slice + L2 normalisation elementwise (:429)
            r_raw, t_raw = Fn.HeadsRawFn.apply(hs, ("rotation_head.", "translation_head."), names + n2, *(params + p2))
            return F.normalize(pick(r_raw), p=2, dim=3), pick(t_raw), hs
        if self.aleatoric:                             # applied BEFORE HeadsQuietFn: it runs last in backward and announces the bucket
            n3, p3 = _named(self.translation_head_aleatoric, "translation_head_aleatoric.")
            n4, p4 = _named(self.rotation_head_aleatoric, "rotation_head_aleatoric.")
            ra_raw, ta_raw = Fn.HeadsRawFn.apply(hs, ("rotation_head_aleatoric.", "translation_head_aleatoric."), n3 + n4, *(p3 + p4))
            self._last_aleatoric = (pick(ra_raw), pick(ta_raw))
            rot, trans = Fn.HeadsQuietFn.apply(hs, cls32, ncls, names + n2, *(params + p2))
            return rot, trans, hs
        rot, trans = Fn.HeadsFn.apply(hs, cls32, ncls, names + n2, *(params + p2))
        return rot, trans, hs

    def make_outputs(self, rot, trans, pred_boxes, pred_classes, boxes_host):
        out = {"pred_translation": trans[-1], "pred_rotation": rot[-1], "pred_boxes": pred_boxes, "pred_classes": pred_classes}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_translation": t, "pred_rotation": r, "pred_boxes": pred_boxes,
                                   "pred_classes": pred_classes} for t, r in zip(trans[:-1], rot[:-1])]
        out["_pred_boxes_host"] = boxes_host          # lets the matcher run without a device->host sync
        out["_tgt_boxes_host"] = getattr(self, "_tgt_boxes_host", None)
        out["_tgt_labels_host"] = getattr(self, "_tgt_labels_host", None)
        out["_pred_classes_host"] = getattr(self, "_pred_classes_host", None)
        out["_stacked"] = (trans, rot)                # (L, N, Q, 3) / (L, N, Q, 3, 3): lets the criterion take all layers at once
        al = getattr(self, "_last_aleatoric", None)
        if al is not None:                            # :402-411
            ra, ta = al
            out["pred_translation_aleatoric"], out["pred_rotation_aleatoric"] = ta[-1], ra[-1]
            for a, aux in enumerate(out.get("aux_outputs", [])):
                aux["pred_translation_aleatoric"], aux["pred_rotation_aleatoric"] = ta[a], ra[a]
        return out

    def forward(self, samples, targets=None):
        features, _pos, pred_objects = self.backbone(samples)
        dev = features[0].tensors.device
        if self.bbox_mode == "backbone":
            if pred_objects is None:
                raise ValueError("bbox_mode 'backbone' needs the backbone's detections (third output of backbone(samples))")
            hw = samples.tensors.shape[-2:] if getattr(samples, "tensors", None) is not None else samples.mask.shape[-2:]
            boxes, classes, valid, n_boxes = self.host_queries_backbone(pred_objects, hw)
        else:
            if targets is None:
                raise ValueError("bbox_mode gt/jitter needs targets")
            boxes, classes, valid, n_boxes = self.host_queries(targets)
        pred_boxes = torch.from_numpy(boxes).to(dev, non_blocking=True)
        pred_classes = torch.from_numpy(classes).to(dev, non_blocking=True)
        val = torch.from_numpy(valid).to(dev, non_blocking=True)
        rot, trans, hs = self.forward_core([f.tensors for f in features], [_u8(f.mask) for f in features], _u8(samples.mask),
                                           pred_boxes, val, pred_classes)
        self._last_hs = hs
        return self.make_outputs(rot, trans, pred_boxes, pred_classes, boxes), n_boxes
