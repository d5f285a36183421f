"""Closed-form MSDA core forward/backward in float64 numpy (TEST INFRASTRUCTURE, see __init__).

Restates SURVEY.md Appendix A (the un-vendored fundamentalvision/Deformable-DETR
``ms_deform_attn`` semantics, as used at models/deformable_transformer.py:201,283): no
grid_sample, no autograd -- exactly the arithmetic the HIP kernels implement, so every kernel
quantity (out, dValue, dLoc, dWeight) has an independent float64 statement.
"""
from __future__ import annotations

import numpy as np


def _corners(loc, h, w):
    px = loc[..., 0] * w - 0.5
    py = loc[..., 1] * h - 0.5
    x0 = np.floor(px)
    y0 = np.floor(py)
    return px - x0, py - y0, x0.astype(np.int64), y0.astype(np.int64)


def msda_forward(value, shapes, loc, attn):
    """value (N,S,M,D), shapes [(H,W)], loc (N,Lq,M,L,P,2), attn (N,Lq,M,L,P) -> (N,Lq,M*D)."""
    value = np.asarray(value, np.float64)
    loc = np.asarray(loc, np.float64)
    attn = np.asarray(attn, np.float64)
    n, s, m, d = value.shape
    lq = loc.shape[1]
    out = np.zeros((n, lq, m, d))
    ni = np.arange(n)[:, None, None, None]
    mi = np.arange(m)[None, None, :, None]
    start = 0
    for lvl, (h, w) in enumerate(shapes):
        fx, fy, x0, y0 = _corners(loc[:, :, :, lvl], h, w)              # (N,Lq,M,P)
        for dy, dx, wt in ((0, 0, (1 - fy) * (1 - fx)), (0, 1, (1 - fy) * fx),
                           (1, 0, fy * (1 - fx)), (1, 1, fy * fx)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            idx = start + np.clip(yy, 0, h - 1) * w + np.clip(xx, 0, w - 1)
            v = value[ni, idx, mi]                                        # (N,Lq,M,P,D)
            out += ((wt * ok * attn[:, :, :, lvl])[..., None] * v).sum(3)
        start += h * w
    return out.reshape(n, lq, m * d)


def msda_backward(value, shapes, loc, attn, grad_out):
    """Returns (dValue, dLoc, dAttn) for loss = <out, grad_out>."""
    value = np.asarray(value, np.float64)
    loc = np.asarray(loc, np.float64)
    attn = np.asarray(attn, np.float64)
    n, s, m, d = value.shape
    lq = loc.shape[1]
    g = np.asarray(grad_out, np.float64).reshape(n, lq, m, 1, d)
    dvalue = np.zeros_like(value)
    dloc = np.zeros_like(loc)
    dattn = np.zeros_like(attn)
    ni = np.arange(n)[:, None, None, None]
    mi = np.arange(m)[None, None, :, None]
    start = 0
    for lvl, (h, w) in enumerate(shapes):
        fx, fy, x0, y0 = _corners(loc[:, :, :, lvl], h, w)
        a = attn[:, :, :, lvl]
        vals = {}
        for dy in (0, 1):
            for dx in (0, 1):
                yy, xx = y0 + dy, x0 + dx
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                idx = start + np.clip(yy, 0, h - 1) * w + np.clip(xx, 0, w - 1)
                vals[(dy, dx)] = (value[ni, idx, mi] * ok[..., None], idx, ok)
        wts = {(0, 0): (1 - fy) * (1 - fx), (0, 1): (1 - fy) * fx, (1, 0): fy * (1 - fx), (1, 1): fy * fx}
        sample = sum(wts[k][..., None] * vals[k][0] for k in wts)
        dattn[:, :, :, lvl] = (g * sample).sum(-1)
        for k in wts:
            v, idx, ok = vals[k]
            contrib = (a * wts[k] * ok)[..., None] * g                   # (N,Lq,M,P,D)
            np.add.at(dvalue, (np.broadcast_to(ni, idx.shape), idx, np.broadcast_to(mi, idx.shape)), contrib)
        v00, v01, v10, v11 = (vals[k][0] for k in ((0, 0), (0, 1), (1, 0), (1, 1)))
        dpx = a * (g * ((1 - fy)[..., None] * (v01 - v00) + fy[..., None] * (v11 - v10))).sum(-1)
        dpy = a * (g * ((1 - fx)[..., None] * (v10 - v00) + fx[..., None] * (v11 - v01))).sum(-1)
        dloc[:, :, :, lvl, :, 0] = dpx * w
        dloc[:, :, :, lvl, :, 1] = dpy * h
        start += h * w
    return dvalue, dloc, dattn
