"""fp32 PyTorch-CPU restatement of the PoET hot path (TEST INFRASTRUCTURE, see __init__).

Every class cites the reference lines it restates (paths relative to /root/reference).
state_dict keys are identical to the reference's so one formula fill drives both.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------------------------
# util/misc.py:346-371 NestedTensor, :326-343 nested_tensor_from_tensor_list
# ----------------------------------------------------------------------------------------------
class NestedTensor:
    def __init__(self, tensors, mask):
        self.tensors, self.mask = tensors, mask

    def decompose(self):
        return self.tensors, self.mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))


def nested_from_list(images: Sequence[torch.Tensor]) -> NestedTensor:
    c = images[0].shape[0]
    hmax = max(int(im.shape[1]) for im in images)
    wmax = max(int(im.shape[2]) for im in images)
    batch = torch.zeros(len(images), c, hmax, wmax, dtype=images[0].dtype)
    mask = torch.ones(len(images), hmax, wmax, dtype=torch.bool)
    for i, im in enumerate(images):
        batch[i, :, : im.shape[1], : im.shape[2]] = im
        mask[i, : im.shape[1], : im.shape[2]] = False
    return NestedTensor(batch, mask)


# ----------------------------------------------------------------------------------------------
# MSDeformAttn -- EXTERNAL op (not in /root/reference).  Spec: SURVEY.md Appendix A.
# Reference call sites: models/deformable_transformer.py:24,58-59,177,201,248,283.
# ----------------------------------------------------------------------------------------------
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """value (N,S,M,D); spatial_shapes [(H,W)..]; loc (N,Lq,M,L,P,2) in [0,1]; w (N,Lq,M,L,P).

    Upstream's documented CPU formulation: per level a bilinear grid_sample with zero
    padding and align_corners=False, multiplied by the weights and summed over (L,P).
    """
    n, s, m, d = value.shape
    _, lq, _, l, p, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes]
    chunks = value.split([h * w for h, w in shapes], dim=1)
    grids = 2.0 * sampling_locations - 1.0
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = chunks[lvl].flatten(2).transpose(1, 2).reshape(n * m, d, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)          # (N*M, Lq, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    wts = attention_weights.transpose(1, 2).reshape(n * m, 1, lq, l * p)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * wts).sum(-1)      # (N*M, D, Lq)
    return out.view(n, m * d, lq).transpose(1, 2).contiguous()


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads:
            raise ValueError("d_model must be divisible by n_heads")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.0)
        theta = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([theta.cos(), theta.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid = grid.repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.0)
        nn.init.constant_(self.attention_weights.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        n, lq, _ = query.shape
        _, s, _ = input_flatten.shape
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == s
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        m, l, p = self.n_heads, self.n_levels, self.n_points
        value = value.view(n, s, m, self.d_model // m)
        off = self.sampling_offsets(query).view(n, lq, m, l, p, 2)
        w = F.softmax(self.attention_weights(query).view(n, lq, m, l * p), -1).view(n, lq, m, l, p)
        if reference_points.shape[-1] != 2:
            raise ValueError("only 2-d reference points are reachable from PoET")
        norm = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1).to(query.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        out = msda_core(value, input_spatial_shapes.tolist(), loc, w)
        return self.output_proj(out)


# ----------------------------------------------------------------------------------------------
# models/position_encoding.py:24-60 and :63-84
# ----------------------------------------------------------------------------------------------
class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=128, temperature=10000, normalize=True, scale=None):
        super().__init__()
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor):
        keep = ~tensor_list.mask
        ey = keep.cumsum(1, dtype=torch.float32)
        ex = keep.cumsum(2, dtype=torch.float32)
        if self.normalize:
            ey = (ey - 0.5) / (ey[:, -1:, :] + 1e-6) * self.scale
            ex = (ex - 0.5) / (ex[:, :, -1:] + 1e-6) * self.scale
        idx = torch.arange(self.num_pos_feats, dtype=torch.float32)
        div = self.temperature ** (2 * (idx // 2) / self.num_pos_feats)
        px = ex[..., None] / div
        py = ey[..., None] / div
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


class BoundingBoxEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=32):
        super().__init__()
        self.num_pos_feats = num_pos_feats

    def forward(self, boxes):
        mult = 2 ** torch.arange(self.num_pos_feats, dtype=torch.float32)
        parts = []
        for c in range(4):
            arg = boxes[:, c, None] * mult
            parts += [arg.sin(), arg.cos()]
        return torch.cat(parts, dim=-1)


# ----------------------------------------------------------------------------------------------
# models/deformable_transformer.py:169-238 encoder, :241-340 decoder, :27-166 wrapper
# ----------------------------------------------------------------------------------------------
def activation_fn(name):
    """deformable_transformer.py:347-355 (`_get_activation_fn`): relu / gelu / glu, anything else raises.  (The reference's own
    builder passes "relu", :358-372; `glu` halves the hidden width and fails in linear2 there as well.)"""
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {name}.")


class EncoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, dropout, n_levels, n_heads, n_points, activation="relu"):
        super().__init__()
        self.activation = activation_fn(activation)
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, ref, shapes, lsi, padding_mask=None):
        a = self.self_attn(src + pos, ref, src, shapes, lsi, padding_mask)
        src = self.norm1(src + self.dropout1(a))
        f = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        return self.norm2(src + self.dropout3(f))


class Encoder(nn.Module):
    def __init__(self, layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def reference_grid(shapes, valid_ratios):
        """deformable_transformer.py:217-230."""
        pieces = []
        for lvl, (h, w) in enumerate(shapes.tolist()):
            ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32)
            xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32)
            gy, gx = torch.meshgrid(ys, xs, indexing="ij")
            gy = gy.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            gx = gx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            pieces.append(torch.stack((gx, gy), -1))
        ref = torch.cat(pieces, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, shapes, lsi, valid_ratios, pos=None, padding_mask=None):
        ref = self.reference_grid(shapes, valid_ratios)
        out = src
        for layer in self.layers:
            out = layer(out, pos, ref, shapes, lsi, padding_mask)
        return out


class DecoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, dropout, n_levels, n_heads, n_points, activation="relu"):
        super().__init__()
        self.activation = activation_fn(activation)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    def forward(self, tgt, query_pos, ref, src, shapes, lsi, padding_mask=None):
        qk = tgt + query_pos
        a = self.self_attn(qk.transpose(0, 1), qk.transpose(0, 1), tgt.transpose(0, 1))[0].transpose(0, 1)
        tgt = self.norm2(tgt + self.dropout2(a))
        c = self.cross_attn(tgt + query_pos, ref, src, shapes, lsi, padding_mask)
        tgt = self.norm1(tgt + self.dropout1(c))
        f = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout4(f))


class Decoder(nn.Module):
    def __init__(self, layer, num_layers, return_intermediate=True):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate

    def forward(self, tgt, ref, src, shapes, lsi, valid_ratios, query_pos=None, padding_mask=None):
        assert ref.shape[-1] == 2
        out, inter, inter_ref = tgt, [], []
        for layer in self.layers:
            ref_in = ref[:, :, None] * valid_ratios[:, None]          # :317
            out = layer(out, query_pos, ref_in, src, shapes, lsi, padding_mask)
            if self.return_intermediate:
                inter.append(out)
                inter_ref.append(ref)
        if self.return_intermediate:
            return torch.stack(inter), torch.stack(inter_ref)
        return out, ref


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=1024, dropout=0.1, return_intermediate_dec=True,
                 num_feature_levels=4, dec_n_points=4, enc_n_points=4, activation="relu"):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = Encoder(EncoderLayer(d_model, dim_feedforward, dropout, num_feature_levels,
                                            nhead, enc_n_points, activation), num_encoder_layers)
        self.decoder = Decoder(DecoderLayer(d_model, dim_feedforward, dropout, num_feature_levels,
                                            nhead, dec_n_points, activation), num_decoder_layers,
                               return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()

    def _reset_parameters(self):
        """deformable_transformer.py:52-62."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for mod in self.modules():
            if isinstance(mod, MSDeformAttn):
                mod._reset_parameters()
        nn.init.xavier_uniform_(self.reference_points.weight, gain=1.0)
        nn.init.constant_(self.reference_points.bias, 0.0)
        nn.init.normal_(self.level_embed)

    @staticmethod
    def valid_ratio(mask):
        _, h, w = mask.shape
        vh = (~mask[:, :, 0]).sum(1).float() / h
        vw = (~mask[:, 0, :]).sum(1).float() / w
        return torch.stack([vw, vh], -1)

    def forward(self, srcs, masks, pos_embeds, query_embed=None, reference_points=None):
        assert query_embed is not None
        flat_src, flat_mask, flat_pos, shapes = [], [], [], []
        for lvl, (src, mask, pos) in enumerate(zip(srcs, masks, pos_embeds)):
            shapes.append(tuple(src.shape[-2:]))
            flat_src.append(src.flatten(2).transpose(1, 2))
            flat_mask.append(mask.flatten(1))
            flat_pos.append(pos.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1))
        src = torch.cat(flat_src, 1)
        mask = torch.cat(flat_mask, 1)
        pos = torch.cat(flat_pos, 1)
        shapes = torch.as_tensor(shapes, dtype=torch.long)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        valid_ratios = torch.stack([self.valid_ratio(m) for m in masks], 1)
        memory = self.encoder(src, shapes, lsi, valid_ratios, pos, mask)

        c = memory.shape[-1]
        if query_embed.dim() == 2:
            query_pos, tgt = torch.split(query_embed, c, dim=1)
            query_pos = query_pos[None].expand(memory.shape[0], -1, -1)
            tgt = tgt[None].expand(memory.shape[0], -1, -1)
        else:
            query_pos, tgt = torch.split(query_embed, c, dim=2)
        if reference_points is None:
            reference_points = self.reference_points(query_pos).sigmoid()
        hs, inter_refs = self.decoder(tgt, reference_points, memory, shapes, lsi, valid_ratios, query_pos, mask)
        return hs, reference_points, inter_refs, None, None


# ----------------------------------------------------------------------------------------------
# models/pose_estimation_transformer.py:677-689 MLP, :434-451 6D->R, :32-451 PoET
# ----------------------------------------------------------------------------------------------
class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = lin(x)
            if i + 1 < self.num_layers:
                x = F.relu(x)
        return x


def rotation_6d_to_matrix(rot_6d):
    bs, nq, _ = rot_6d.shape
    r = rot_6d.reshape(-1, 6)
    x = F.normalize(r[:, 0:3], p=2, dim=1)
    z = F.normalize(torch.cross(x, r[:, 3:6], dim=1), p=2, dim=1)
    y = torch.cross(z, x, dim=1)
    return torch.stack((x, y, z), dim=2).view(bs, nq, 3, 3)          # x,y,z are the COLUMNS


class PositionEmbeddingLearned(nn.Module):
    """models/position_encoding.py:87-112: learned absolute encoding, [col_embed(x) | row_embed(y)] per pixel (selected by
    `--position_embedding learned`, position_encoding.py:115-127; nn.Embedding(50, .): feature maps up to 50 x 50)."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, tensor_list):
        x = tensor_list.tensors
        h, w = x.shape[-2:]
        x_emb = self.col_embed(torch.arange(w, device=x.device))
        y_emb = self.row_embed(torch.arange(h, device=x.device))
        pos = torch.cat([x_emb.unsqueeze(0).repeat(h, 1, 1), y_emb.unsqueeze(1).repeat(1, w, 1)], dim=-1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(x.shape[0], 1, 1, 1)


class SyntheticBackbone(nn.Module):
    """Stands at the backbone interface (models/backbone.py:26-50 ``Joiner``): frozen, returns
    pre-made multi-scale feature maps as NestedTensors, their sine encodings, and the detections given at
    construction (``predictions``: per image None or rows (x0, y0, x1, y1, score, class) in pixels; None = a training
    backbone).  ``self[1]`` is the position embedding (pose_estimation_transformer.py:332)."""

    def __init__(self, features: List[torch.Tensor], strides, num_channels, pos_feats=128, predictions=None, position_embedding="sine"):
        super().__init__()
        self.features = features
        self.predictions = predictions
        self.strides, self.num_channels = list(strides), list(num_channels)
        # registered as "1": the reference's Joiner is nn.Sequential(backbone, position_embedding), so a learned encoding's
        # parameters are `backbone.1.row_embed.weight` / `backbone.1.col_embed.weight` (backbone.py:26-33)
        self.add_module("1", PositionEmbeddingLearned(pos_feats) if position_embedding == "learned" else PositionEmbeddingSine(pos_feats, normalize=True))
        self.train_backbone = False

    @property
    def position_embedding(self):
        return self._modules["1"]

    def __getitem__(self, idx):
        return self if idx == 0 else self.position_embedding

    def forward(self, samples: NestedTensor):
        outs, pos = [], []
        for f in self.features:
            m = F.interpolate(samples.mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
            outs.append(NestedTensor(f, m))
        for x in outs:
            pos.append(self.position_embedding(x).to(x.tensors.dtype))
        return outs, pos, self.predictions


class PoET(nn.Module):
    def __init__(self, backbone, transformer, num_queries, num_feature_levels, n_classes,
                 bbox_mode="gt", class_mode="specific", aux_loss=True, rotation_mode="6d", aleatoric=False,
                 ref_points_mode="bbox", query_embedding_mode="bbox"):
        super().__init__()
        assert ref_points_mode in ("bbox", "learned") and query_embedding_mode in ("bbox", "learned")
        self.ref_points_mode, self.query_embedding_mode = ref_points_mode, query_embedding_mode
        self.transformer, self.backbone = transformer, backbone
        d = transformer.d_model
        self.hidden_dim, self.n_queries, self.n_classes = d, num_queries, n_classes + 1
        self.bbox_mode, self.class_mode, self.aux_loss = bbox_mode, class_mode, aux_loss
        self.rotation_mode, self.aleatoric = rotation_mode, aleatoric
        assert rotation_mode in ("6d", "quat", "silho_quat") and not (aleatoric and rotation_mode != "6d")      # :72-82
        rot_dim = 6 if rotation_mode == "6d" else 4
        self.num_feature_levels = num_feature_levels
        mult = self.n_classes if class_mode == "specific" else 1
        n_pred = transformer.decoder.num_layers
        # construction order (and hence RNG consumption) as pose_estimation_transformer.py:85-144
        self.translation_head = t_head = MLP(d, d, 3 * mult, 3)    # registered first, replaced by the ModuleList below
        self.rotation_head = r_head = MLP(d, d, rot_dim * mult, 3) # (keeps the reference's parameter order)
        if aleatoric:                                              # :88-90,95-96: log-variances, 3 per head
            self.translation_head_aleatoric = ta_head = MLP(d, d, 3 * mult, 3)
            self.rotation_head_aleatoric = ra_head = MLP(d, d, 3 * mult, 3)
        projs = []
        n_bb = len(backbone.strides)
        cin = None
        for i in range(n_bb):
            cin = backbone.num_channels[i]
            projs.append(nn.Sequential(nn.Conv2d(cin, d, kernel_size=1), nn.GroupNorm(32, d)))
        for _ in range(num_feature_levels - n_bb):
            projs.append(nn.Sequential(nn.Conv2d(cin, d, kernel_size=3, stride=2, padding=1), nn.GroupNorm(32, d)))
            cin = d
        self.input_proj = nn.ModuleList(projs)
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.translation_head = nn.ModuleList([copy.deepcopy(t_head) for _ in range(n_pred)])
        self.rotation_head = nn.ModuleList([copy.deepcopy(r_head) for _ in range(n_pred)])
        if aleatoric:
            self.translation_head_aleatoric = nn.ModuleList([copy.deepcopy(ta_head) for _ in range(n_pred)])
            self.rotation_head_aleatoric = nn.ModuleList([copy.deepcopy(ra_head) for _ in range(n_pred)])
        if query_embedding_mode == "learned":           # pose_estimation_transformer.py:149-150 (nn.Embedding: normal_ init)
            self.query_embed = nn.Embedding(num_queries, d * 2)
        self.bbox_embedding = BoundingBoxEmbeddingSine(num_pos_feats=d / 8)

    def assemble_queries(self, targets):
        """pose_estimation_transformer.py:203-239,309-311 ('gt' mode)."""
        boxes_all, cls_all, emb_all, n_boxes = [], [], [], []
        for t in targets:
            boxes = t["jitter_boxes"] if self.bbox_mode == "jitter" else t["boxes"]      # :206-209
            nb = len(boxes)
            n_boxes.append(nb)
            cls = t["labels"]
            emb = self.bbox_embedding(boxes).repeat(1, 2)
            pad = self.n_queries - nb
            if pad > 0:
                boxes = torch.vstack((boxes, torch.full((pad, 4), -1.0)))
                emb = torch.cat([emb, torch.full((pad, 2 * self.hidden_dim), -10.0)], 0)
                cls = torch.cat((cls, torch.full((pad,), -1, dtype=torch.int32)))
            boxes_all.append(boxes)
            emb_all.append(emb)
            cls_all.append(cls)
        return torch.stack(emb_all), torch.stack(boxes_all), torch.stack(cls_all), n_boxes

    def assemble_queries_backbone(self, pred_objects, image_hw):
        """pose_estimation_transformer.py:240-305 ('backbone' mode, the inference path): per image the detector's
        rows (x0, y0, x1, y1, score, class) in pixels -> cxcywh normalised by the batch's image size (util/box_ops.py:
        24-40); more rows than queries: the top n_queries by score; fewer (or None): dummy padding as in 'gt' mode."""
        ih, iw = image_hw
        boxes_all, cls_all, emb_all, n_boxes = [], [], [], []
        for pred in pred_objects:
            if pred is None:
                nb, boxes, cls = 0, torch.zeros((0, 4)), torch.zeros((0,), dtype=torch.int64)
                emb = torch.zeros((0, 2 * self.hidden_dim))
            else:
                x0, y0, x1, y1 = pred[:, :4].unbind(-1)
                boxes = torch.stack([(x0 + x1) / 2 / iw, (y0 + y1) / 2 / ih, (x1 - x0) / iw, (y1 - y0) / ih], -1)
                scores, cls = pred[:, 4], pred[:, 5].to(torch.int64)
                emb = self.bbox_embedding(boxes).repeat(1, 2)
                nb = len(boxes)
                if nb > self.n_queries:
                    order = torch.sort(scores, dim=0, descending=True)[1][: self.n_queries]
                    boxes, cls, emb, nb = boxes[order], cls[order], emb[order], self.n_queries
            pad = self.n_queries - nb
            if pad > 0:
                boxes = torch.vstack((boxes, torch.full((pad, 4), -1.0)))
                emb = torch.cat([emb, torch.full((pad, 2 * self.hidden_dim), -10.0)], 0)
                cls = torch.cat((cls, torch.full((pad,), -1, dtype=torch.int64)))
            n_boxes.append(nb)
            boxes_all.append(boxes)
            emb_all.append(emb)
            cls_all.append(cls)
        return torch.stack(emb_all), torch.stack(boxes_all), torch.stack(cls_all), n_boxes

    def forward(self, samples: NestedTensor, targets=None):
        features, pos, pred_objects = self.backbone(samples)
        if self.bbox_mode == "backbone":
            query_embeds, pred_boxes, pred_classes, n_boxes = self.assemble_queries_backbone(pred_objects, samples.tensors.shape[-2:])
        else:
            query_embeds, pred_boxes, pred_classes, n_boxes = self.assemble_queries(targets)
        srcs, masks = [], []
        for lvl, feat in enumerate(features):
            src, mask = feat.decompose()
            srcs.append(self.input_proj[lvl](src))
            masks.append(mask)
        for lvl in range(len(srcs), self.num_feature_levels):
            inp = features[-1].tensors if lvl == len(features) else srcs[-1]
            src = self.input_proj[lvl](inp)
            mask = F.interpolate(samples.mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
            pos.append(self.backbone[1](NestedTensor(src, mask)).to(src.dtype))
            srcs.append(src)
            masks.append(mask)
        ref = pred_boxes[:, :, :2] if self.ref_points_mode == "bbox" else None      # :337-340
        if self.query_embedding_mode == "learned":                                  # :342-343
            query_embeds = self.query_embed.weight
        hs, _, _, _, _ = self.transformer(srcs, masks, pos, query_embeds, ref)

        bs = pred_classes.shape[0]
        idx = torch.where(pred_classes > 0, pred_classes, 0).view(-1).long()
        rows = torch.arange(bs * self.n_queries)
        def pick(x):                                    # :365-374: per query the slice of its class
            return x.view(bs * self.n_queries, self.n_classes, -1)[rows, idx].view(bs, self.n_queries, -1) if self.class_mode == "specific" else x

        rots, trans, rots_a, trans_a = [], [], [], []
        for lvl in range(hs.shape[0]):
            r = pick(self.rotation_head[lvl](hs[lvl]))
            trans.append(pick(self.translation_head[lvl](hs[lvl])))
            rots.append(rotation_6d_to_matrix(r) if self.rotation_mode == "6d" else F.normalize(r, p=2, dim=2))      # :420-432
            if self.aleatoric:
                rots_a.append(pick(self.rotation_head_aleatoric[lvl](hs[lvl])))
                trans_a.append(pick(self.translation_head_aleatoric[lvl](hs[lvl])))
        rots, trans = torch.stack(rots), torch.stack(trans)
        out = {"pred_translation": trans[-1], "pred_rotation": rots[-1],
               "pred_boxes": pred_boxes, "pred_classes": pred_classes}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_translation": t, "pred_rotation": r, "pred_boxes": pred_boxes,
                                   "pred_classes": pred_classes} for t, r in zip(trans[:-1], rots[:-1])]
        if self.aleatoric:                              # :402-411
            out["pred_translation_aleatoric"], out["pred_rotation_aleatoric"] = trans_a[-1], rots_a[-1]
            if self.aux_loss:
                for a, aux in enumerate(out["aux_outputs"]):
                    aux["pred_translation_aleatoric"], aux["pred_rotation_aleatoric"] = trans_a[a], rots_a[a]
        return out, n_boxes


# ----------------------------------------------------------------------------------------------
# models/matcher.py:104-229 PoseMatcher ('gt' mode) and pose_estimation_transformer.py:454-674
# ----------------------------------------------------------------------------------------------
class PoseMatcher(nn.Module):
    def __init__(self, cost_bbox=1.0, cost_class=1.0, bbox_mode="gt", class_mode="specific"):
        super().__init__()
        assert bbox_mode in ("gt", "jitter", "backbone")
        self.cost_bbox, self.cost_class, self.bbox_mode, self.class_mode = cost_bbox, cost_class, bbox_mode, class_mode

    @torch.no_grad()
    def forward(self, outputs, targets, n_boxes, giou_thresh=0.5):
        from scipy.optimize import linear_sum_assignment
        bs, nq = outputs["pred_boxes"].shape[:2]
        out_bbox = outputs["pred_boxes"].flatten(0, 1)
        tgt_bbox = torch.cat([t["boxes"] for t in targets])
        if self.bbox_mode == "gt":          # matcher.py:169-173: L1 between the (identical) boxes
            cost = (self.cost_bbox * torch.cdist(out_bbox, tgt_bbox, p=1)).view(bs, nq, -1).cpu()
        else:                               # matcher.py:175-181 ('jitter'): class equality only, 0 = same class
            out_class = outputs["pred_classes"].flatten(0, 1)
            tgt_class = torch.cat([t["labels"].type(torch.float32) for t in targets])
            cost = self.cost_class * torch.where(out_class[:, None] == tgt_class[None, :], 0.0, 1.0)
            if self.bbox_mode == "backbone":    # matcher.py:183-192: + L1 between the box CENTRES
                cost = self.cost_bbox * torch.cdist(out_bbox[:, 0:2], tgt_bbox[:, 0:2], p=1) + cost
            cost = cost.view(bs, nq, -1).cpu()
        sizes = [len(t["boxes"]) for t in targets]
        res = []
        for i, c in enumerate(cost.split(sizes, -1)):
            r, cidx = linear_sum_assignment(c[i][: n_boxes[i]])
            res.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(cidx, dtype=torch.int64)))
        if self.bbox_mode == "backbone":    # matcher.py:204-227: drop matches of the wrong class / with GIoU below the threshold
            kept = []
            for b, ((src, tgt), t) in enumerate(zip(res, targets)):
                ob, oc = outputs["pred_boxes"][b, : n_boxes[b]], outputs["pred_classes"][b]
                gious = generalized_box_iou(box_cxcywh_to_xyxy(ob), box_cxcywh_to_xyxy(t["boxes"]))
                keep = [k for k, (i, j) in enumerate(zip(src.tolist(), tgt.tolist()))
                        if not (self.class_mode == "specific" and oc[i] != t["labels"][j]) and not (gious[i, j] < giou_thresh)]
                kept.append((src[keep], tgt[keep]))
            res = kept
        return res


def box_cxcywh_to_xyxy(x):
    """util/box_ops.py:21-25."""
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


def generalized_box_iou(b1, b2):
    """util/box_ops.py:68-105 (xyxy boxes -> [N, M] GIoU)."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    wh = (torch.max(b1[:, None, 2:], b2[:, 2:]) - torch.min(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def acos_linear_extrapolation(x, lo, hi):
    """util/rotation_utils.py:13-67: acos inside (lo, hi), first-order Taylor continuation outside."""
    out = torch.empty_like(x)
    up, low = x >= hi, x <= lo
    mid = ~up & ~low
    out[mid] = torch.acos(x[mid])
    for m, b in ((up, hi), (low, lo)):
        out[m] = math.acos(b) - (x[m] - b) / math.sqrt(1.0 - b * b)
    return out


def so3_log_map(R, eps=1e-4, cos_bound=1e-4):
    """util/rotation_utils.py:150-192,244-316: rotation matrices (n,3,3) -> rotation vectors (n,3)."""
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    if ((tr < -1.0 - eps) | (tr > 3.0 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    phi = acos_linear_extrapolation((tr - 1.0) * 0.5, -(1.0 - cos_bound), 1.0 - cos_bound)
    sin = torch.sin(phi)
    fac = torch.empty_like(phi)
    ok = sin.abs() > 0.5 * eps
    fac[~ok] = 0.5 + (phi[~ok] ** 2) * (1.0 / 12)
    fac[ok] = phi[ok] / (2.0 * sin[ok])
    A = fac[:, None, None] * (R - R.permute(0, 2, 1))
    return torch.stack((A[:, 2, 1], A[:, 0, 2], A[:, 1, 0]), dim=1)


class SetCriterion(nn.Module):
    def __init__(self, matcher, weight_dict, losses=("translation", "rotation")):
        super().__init__()
        self.matcher, self.weight_dict, self.losses = matcher, weight_dict, tuple(losses)

    @staticmethod
    def _src_idx(indices):
        b = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        s = torch.cat([src for (src, _) in indices])
        return b, s

    def _losses(self, outputs, targets, indices):
        """pose_estimation_transformer.py:472-609, one entry of `self.losses` per term."""
        idx = self._src_idx(indices)
        gather = lambda key: torch.cat([t[key][j] for t, (_, j) in zip(targets, indices)], 0)
        st, tt = outputs["pred_translation"][idx], gather("relative_position")
        n_obj = len(tt)
        res = {}
        for name in self.losses:
            if name == "translation":
                res["loss_trans"] = torch.sqrt(((st - tt) ** 2).sum(1)).sum() / n_obj
            elif name == "aleatoric_translation":        # s = log(sigma^2) per axis
                sa = outputs["pred_translation_aleatoric"][idx]
                res["loss_trans"] = ((torch.exp(-sa) * torch.square(tt - st)).sum(1) + sa.sum(1)).sum() / (2 * n_obj)
            elif name in ("rotation", "aleatoric_rotation"):
                sr, tr = outputs["pred_rotation"][idx], gather("relative_rotation")
                prod = torch.bmm(sr, tr.transpose(1, 2))
                if name == "rotation":
                    trace = prod[:, torch.eye(3).bool()].sum(1)
                    theta = torch.clamp(0.5 * (trace - 1), -1 + 1e-6, 1 - 1e-6)
                    res["loss_rot"] = torch.acos(theta).sum() / n_obj
                else:
                    sa = outputs["pred_rotation_aleatoric"][idx]
                    res["loss_rot"] = ((torch.exp(-sa) * torch.square(so3_log_map(prod))).sum(1) + sa.sum(1)).sum() / (2 * n_obj)
            elif name in ("quaternion", "silho_quaternion"):   # [w, x, y, z]; eps 1e-4
                dp = (outputs["pred_rotation"][idx] * gather("relative_quaternions")).sum(1)
                lq = -torch.log(torch.square(dp) + 1e-4) if name == "quaternion" else torch.log(1 - torch.abs(dp) + 1e-4)
                res["loss_rot"] = lq.sum() / n_obj
            else:
                raise ValueError(name)
        return res

    def forward(self, outputs, targets, n_boxes):
        main = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        losses = dict(self._losses(outputs, targets, self.matcher(main, targets, n_boxes)))
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            part = self._losses(aux, targets, self.matcher(aux, targets, n_boxes))
            losses.update({f"{k}_{i}": v for k, v in part.items()})
        return losses


def losses_for(rotation_mode="6d", aleatoric=False):
    """main.py's choice of loss terms for a rotation representation / the aleatoric extension."""
    if aleatoric:
        return ("aleatoric_translation", "aleatoric_rotation")
    return ("translation", {"6d": "rotation", "quat": "quaternion", "silho_quat": "silho_quaternion"}[rotation_mode])


def build_weight_dict(dec_layers, t_coef=1.0, r_coef=1.0, aux=True):
    wd = {"loss_trans": t_coef, "loss_rot": r_coef}
    if aux:
        for i in range(dec_layers - 1):
            wd.update({f"loss_trans_{i}": t_coef, f"loss_rot_{i}": r_coef})
    return wd


def param_groups(model, lr=2e-4, lr_backbone=2e-5, proj_names=("reference_points", "sampling_offsets"), proj_mult=0.1):
    """main.py:253-271."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    is_bb = lambda n: "backbone.0" in n
    is_pr = lambda n: any(k in n for k in proj_names)
    return [
        {"params": [p for n, p in named if not is_bb(n) and not is_pr(n)], "lr": lr},
        {"params": [p for n, p in named if is_bb(n)], "lr": lr_backbone},
        {"params": [p for n, p in named if is_pr(n)], "lr": lr * proj_mult},
    ]


def build_poet(cfg, features, bbox_mode="gt", predictions=None, class_mode="specific", rotation_mode="6d", aleatoric=False,
               ref_points_mode="bbox", query_embedding_mode="bbox", position_embedding="sine"):
    """cfg: dict(d_model, nheads, enc_layers, dec_layers, d_ffn, n_levels, n_points, num_queries,
    n_classes, dropout, strides, num_channels)."""
    bb = SyntheticBackbone(features, cfg["strides"], cfg["num_channels"], cfg["d_model"] // 2, predictions=predictions,
                           position_embedding=position_embedding)
    tr = DeformableTransformer(cfg["d_model"], cfg["nheads"], cfg["enc_layers"], cfg["dec_layers"],
                               cfg["d_ffn"], cfg["dropout"], True, cfg["n_levels"], cfg["n_points"], cfg["n_points"],
                               activation=cfg.get("activation", "relu"))
    model = PoET(bb, tr, cfg["num_queries"], cfg["n_levels"], cfg["n_classes"], bbox_mode, class_mode, True,
                 rotation_mode=rotation_mode, aleatoric=aleatoric, ref_points_mode=ref_points_mode, query_embedding_mode=query_embedding_mode)
    crit = SetCriterion(PoseMatcher(bbox_mode="jitter" if bbox_mode == "jitter" else "gt"), build_weight_dict(cfg["dec_layers"]),
                        losses_for(rotation_mode, aleatoric))
    return model, crit
