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
from .modules import NestedTensor, PositionEmbeddingSine


class SyntheticBackbone(nn.Module):
    def __init__(self, features: List[torch.Tensor], strides: Sequence[int], num_channels: Sequence[int], pos_feats: int = 128,
                 predictions=None):
        super().__init__()
        self.features = features
        self.predictions = predictions
        self.strides, self.num_channels = list(strides), list(num_channels)
        self.position_embedding = PositionEmbeddingSine(pos_feats, normalize=True)
        self.train_backbone = False

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
