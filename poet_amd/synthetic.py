"""Synthetic stand-in at the backbone interface (models/backbone.py:26-50 `Joiner`): the reference's
detector backbone is frozen, external and out of scope, so benchmarks and tests feed pre-made
multi-scale feature maps here.  Returns (features: list[NestedTensor], pos: None, predictions) where `predictions` are the
detector rows given at construction (None for training; inference: per image None or (n, 6) rows x0,y0,x1,y1,score,class);
level masks are the nearest-neighbour resize of the image padding mask, as real backbones produce."""
from __future__ import annotations

from typing import List, Sequence

import torch
from torch import nn

from . import ops
from .modules import NestedTensor, PositionEmbeddingLearned, PositionEmbeddingSine


class SyntheticBackbone(nn.Module):
    def __init__(self, features: List[torch.Tensor], strides: Sequence[int], num_channels: Sequence[int], pos_feats: int = 128,
                 predictions=None, position_embedding: str = "sine"):
        super().__init__()
        self.features = features
        self.predictions = predictions
        self.strides, self.num_channels = list(strides), list(num_channels)
        # registered as "1" -- the reference's Joiner is nn.Sequential(backbone, position_embedding) (backbone.py:26-33), so the
        # state_dict keys of a learned encoding are backbone.1.row_embed.weight / backbone.1.col_embed.weight
        self.add_module("1", PositionEmbeddingLearned(pos_feats) if position_embedding == "learned" else PositionEmbeddingSine(pos_feats, normalize=True))
        self.train_backbone = False

    @property
    def position_embedding(self):
        return self._modules["1"]

    def __getitem__(self, idx):
        return self if idx == 0 else self.position_embedding

    def forward(self, samples: NestedTensor):
        im = samples.mask
        im8 = im.contiguous().view(torch.uint8) if im.dtype == torch.bool else im.contiguous()
        N, H, W = im8.shape
        out = []
        for f in self.features:
            h, w = f.shape[-2:]
            m = torch.empty((N, h, w), dtype=torch.uint8, device=f.device)
            ops.mask_nearest(im8, m, N, H, W, h, w)
            out.append(NestedTensor(f, m))
        return out, None, self.predictions


def image_mask(sizes, device):
    """Padding mask of a batch of images of the given (h,w) sizes (util/misc.py:326-343)."""
    H = max(s[0] for s in sizes)
    W = max(s[1] for s in sizes)
    m = torch.ones((len(sizes), H, W), dtype=torch.bool)
    for i, (h, w) in enumerate(sizes):
        m[i, :h, :w] = False
    return m.to(device)


class FrozenConvBackbone(nn.Module):
    """A real (small) frozen convolutional backbone at the same interface: images in, three feature levels at strides
    8 / 16 / 32 with `channels` channels each, level masks by nearest resize of the image padding mask (backbone.py:36-50:
    `requires_grad_(False)`, `F.interpolate(mask[None].float(), size=x.shape[-2:])`).  It is NOT part of the hot path -- it
    runs in plain PyTorch like the reference's backbone does and exists so that the path is exercised end to end from padded
    images: non-trivial masks, valid ratios != 1 and feature maps with spatial structure.  `nested_cls` lets the
    CPU oracle reuse it with its own NestedTensor type."""

    def __init__(self, channels: int = 256, seed: int = 7, nested_cls=NestedTensor, pos_embed=None):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        chans = [3, 32, 64, channels, channels, channels]
        self.convs = nn.ModuleList(nn.Conv2d(chans[i], chans[i + 1], 3, stride=2, padding=1) for i in range(5))
        with torch.no_grad():
            for c in self.convs:
                fan_in = c.weight[0].numel()
                c.weight.copy_(torch.randn(c.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.1)
        for p in self.parameters():
            p.requires_grad_(False)
        self.strides, self.num_channels = [8, 16, 32], [channels] * 3
        self.nested_cls = nested_cls
        self.train_backbone = False
        self.position_embedding = pos_embed               # None here: PoET computes the level encodings itself (poet_pos_sine)

    def __getitem__(self, idx):
        return self if idx == 0 else self.position_embedding

    @torch.no_grad()
    def forward(self, samples):
        c = self.convs
        x = torch.relu(c[1](torch.relu(c[0](samples.tensors))))
        f0 = c[2](x)
        f1 = c[3](torch.relu(f0))
        f2 = c[4](torch.relu(f1))
        out = []
        for f in (f0, f1, f2):
            m = torch.nn.functional.interpolate(samples.mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
            out.append(self.nested_cls(f.contiguous(), m))
        pos = None if self.position_embedding is None else [self.position_embedding(o).to(o.tensors.dtype) for o in out]
        return out, pos, None
