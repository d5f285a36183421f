"""Kernel-level parity: every C-ABI kernel vs the CPU oracle / an fp32 torch statement of the
same arithmetic, on seeded inputs.  -m gpu only."""
import contextlib
import os
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# kernels that were built, measured and lost are not in the product library; a probe build (POET_BUILD_PROBES=1) carries them
PROBES_BUILT = os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "poet_amd", "csrc", ".probes_on"))

from oracle import poet_ref, msda_explicit  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from poet_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


TOL = {torch.float32: dict(atol=2e-4, rtol=2e-4), torch.bfloat16: dict(atol=6e-2, rtol=3e-2)}


def _close(a, b, dtype, scale=1.0, msg=""):
    a = a.float().cpu()
    b = b.float().cpu()
    tol = TOL[dtype]
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol["atol"] * scale + tol["rtol"] * ref, f"{msg}: max err {err} (ref max {ref})"


# ---------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 768, 256), (640, 256, 1024), (70, 66, 256), (320, 132, 256)])
def test_gemm_nt_bias_relu(ops, dtype, M, N, K):
    x = _rand(M, K, seed=1).to(dtype)
    w = _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    b = _rand(N, seed=3)
    out = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out, act=1)
    wq = w.to(dtype).float() if dtype == torch.bfloat16 else w
    ref = F.relu(x.float() @ wq.t() + b)
    _close(out, ref, dtype, msg=f"NT {M}x{N}x{K}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1000, 256, 768), (320, 256, 66), (640, 1024, 256)])
def test_gemm_dx_gate_add(ops, dtype, M, N, K):
    dy = _rand(M, K, seed=4).to(dtype)
    w = _rand(K, N, seed=5, scale=1 / math.sqrt(K))          # [N_out=K, K_in=N]
    gate = _rand(M, N, seed=6).to(dtype)
    addend = _rand(M, N, seed=7).to(dtype)
    out = dev(addend.clone())
    ops.linear_dx(dev(dy), dev(w), out, rows=M, add_src=out, gate_ref=dev(gate), gate_scale=1.25)
    wq = w.to(dtype).float() if dtype == torch.bfloat16 else w
    ref = (dy.float() @ wq) * (gate.float() > 0) * 1.25 + addend.float()
    _close(out, ref, dtype, msg="dX")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,nout,kin", [(5000, 256, 256), (3000, 768, 256), (320, 66, 256), (20, 256, 1024), (2048, 1024, 256)])
def test_gemm_dw_splitk(ops, dtype, rows, nout, kin):
    dy = _rand(rows, nout, seed=8).to(dtype)
    x = _rand(rows, kin, seed=9).to(dtype)
    dw = torch.zeros(nout, kin, device="cuda")
    ops.linear_dw(dev(dy), dev(x), dw, rows=rows)
    ref = dy.float().t() @ x.float()
    _close(dw, ref, dtype, scale=math.sqrt(rows), msg="dW")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nchw_batched_and_headmajor(ops, dtype):
    # 1x1 conv on NCHW features: A K-major per image
    N, Cin, HW, d = 3, 48, 100, 64
    feat = _rand(N, Cin, HW, seed=10).to(dtype)
    w = _rand(d, Cin, seed=11, scale=0.2)
    b = _rand(d, seed=12)
    out = torch.empty(N, HW, d, dtype=dtype, device="cuda")
    ops.gemm(dev(feat), dev(w), out, HW, d, Cin, lda=HW, ldb=Cin, ldc=d, a_kmajor=True, bias=dev(b), batch=N,
             strideA=Cin * HW, strideC=HW * d)
    wq = w.to(dtype).float() if dtype == torch.bfloat16 else w
    ref = torch.einsum("nch,dc->nhd", feat.float(), wq) + b
    _close(out, ref, dtype, msg="nchw conv1x1")
    # weight grad of the same conv: A = dout K-major, B = feat k-contiguous, batch accumulates atomically
    dout = _rand(N, HW, d, seed=13).to(dtype)
    dw = torch.zeros(d, Cin, device="cuda")
    ops.gemm(dev(dout), dev(feat), dw, d, Cin, HW, lda=d, ldb=HW, ldc=Cin, a_kmajor=True, b_kmajor=False, batch=N,
             strideA=HW * d, strideB=Cin * HW, strideC=0, atomic=True)
    ref = torch.einsum("nhd,nch->dc", dout.float(), feat.float())
    _close(dw, ref, dtype, scale=10, msg="conv dW")
    # head-major value epilogue with row mask
    Nn, S, M, D = 2, 150, 4, 16
    x = _rand(Nn * S, 64, seed=14).to(dtype)
    wv = _rand(M * D, 64, seed=15, scale=0.15)
    mask = (torch.arange(Nn * S) % 7 == 0).to(torch.uint8)
    v = torch.empty(Nn, M, S, D, dtype=dtype, device="cuda")
    ops.linear_fwd(dev(x), dev(wv), None, v, row_mask=dev(mask), head_major=(M, S, D))
    wq = wv.to(dtype).float() if dtype == torch.bfloat16 else wv
    ref = (x.float() @ wq.t()).masked_fill(mask.bool()[:, None], 0).view(Nn, S, M, D).permute(0, 2, 1, 3)
    _close(v, ref, dtype, msg="head-major")


def test_gemm_dropout_statistics(ops):
    M, N, K = 512, 256, 64
    x = torch.ones(M, K)
    w = torch.ones(N, K) / K
    out = torch.empty(M, N, device="cuda")
    ops.linear_fwd(dev(x), dev(w), None, out, drop_p=0.1, seed=123)
    o = out.cpu()
    kept = (o != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01
    assert torch.allclose(o[o != 0], torch.full_like(o[o != 0], 1 / 0.9), atol=1e-5)
    out2 = torch.empty(M, N, device="cuda")
    ops.linear_fwd(dev(x), dev(w), None, out2, drop_p=0.1, seed=123)
    assert torch.equal(out, out2)


# ---------------------------------------------------------------------------------------------- MSDA
def _msda_inputs(seed, shapes, n=2, m=4, d=16, lq=37, p=4):
    rng = np.random.default_rng(seed)
    s = sum(h * w for h, w in shapes)
    value = rng.standard_normal((n, s, m, d)).astype(np.float32)
    loc = rng.uniform(-0.25, 1.25, (n, lq, m, len(shapes), p, 2)).astype(np.float32)
    attn = torch.softmax(torch.from_numpy(rng.standard_normal((n, lq, m, len(shapes) * p)).astype(np.float32)), -1)
    attn = attn.view(n, lq, m, len(shapes), p).numpy()
    gout = rng.standard_normal((n, lq, m * d)).astype(np.float32)
    return value, loc, attn, gout


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shapes,m,d", [([(6, 8), (3, 4)], 4, 16), ([(12, 16), (6, 8), (3, 4), (2, 2)], 8, 32), ([(5, 7)], 2, 64),
                                        ([(9, 11), (5, 6), (3, 3)], 4, 16), ([(7, 5), (4, 3)], 3, 8), ([(1, 1), (2, 3)], 5, 8),
                                        ([(13, 17), (7, 9), (4, 5), (2, 3)], 1, 128)])
def test_msda_boundary(ops, dtype, shapes, m, d):
    value, loc, attn, gout = _msda_inputs(3, shapes, m=m, d=d)
    geom = ops.LevelGeom(shapes)
    tv, tl, ta, tg = (torch.from_numpy(a).to(dtype) for a in (value, loc, attn, gout))
    out = torch.empty(tg.shape, dtype=dtype, device="cuda")
    ops.msda_fwd(dev(tv), geom, dev(tl), dev(ta), out)
    ref = msda_explicit.msda_forward(tv.float().numpy(), shapes, tl.float().numpy(), ta.float().numpy())
    _close(out, torch.from_numpy(ref), dtype, msg="msda fwd")
    gv = torch.empty(tv.shape, dtype=torch.float32, device="cuda")
    gl = torch.empty(tl.shape, dtype=dtype, device="cuda")
    ga = torch.empty(ta.shape, dtype=dtype, device="cuda")
    ops.msda_bwd(dev(tv), geom, dev(tl), dev(ta), dev(tg), gv, gl, ga)
    rv, rl, ra = msda_explicit.msda_backward(tv.float().numpy(), shapes, tl.float().numpy(), ta.float().numpy(), tg.float().numpy())
    _close(gv, torch.from_numpy(rv), dtype, scale=3, msg="msda dvalue")
    _close(ga, torch.from_numpy(ra), dtype, scale=3, msg="msda dattn")
    _close(gl, torch.from_numpy(rl), dtype, scale=30, msg="msda dloc")


@pytest.mark.parametrize("npts", [1, 2])
def test_msda_fewer_points(ops, npts):
    """n_points 1 and 2 (the other instantiations of the gather / scatter kernels) against the float64 closed form."""
    shapes = [(9, 11), (5, 6), (3, 3)]
    value, loc, attn, gout = _msda_inputs(11, shapes, m=4, d=16, p=npts)
    geom = ops.LevelGeom(shapes)
    tv, tl, ta, tg = (torch.from_numpy(a) for a in (value, loc, attn, gout))
    out = torch.empty(tg.shape, device="cuda")
    ops.msda_fwd(dev(tv), geom, dev(tl), dev(ta), out)
    _close(out, torch.from_numpy(msda_explicit.msda_forward(value, shapes, loc, attn)), torch.float32, msg="msda fwd")
    gv = torch.empty(tv.shape, device="cuda"); gl = torch.empty(tl.shape, device="cuda"); ga = torch.empty(ta.shape, device="cuda")
    ops.msda_bwd(dev(tv), geom, dev(tl), dev(ta), dev(tg), gv, gl, ga)
    rv, rl, ra = msda_explicit.msda_backward(value, shapes, loc, attn, gout)
    _close(gv, torch.from_numpy(rv), torch.float32, scale=3, msg="msda dvalue")
    _close(ga, torch.from_numpy(ra), torch.float32, scale=3, msg="msda dattn")
    _close(gl, torch.from_numpy(rl), torch.float32, scale=30, msg="msda dloc")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shapes,m,d,npts", [([(9, 11), (5, 6), (3, 3), (2, 2), (1, 2)], 4, 16, 3), ([(7, 5), (4, 3)], 3, 8, 8),
                                             ([(6, 8), (5, 7), (4, 6), (3, 4), (3, 3), (2, 3), (2, 2), (1, 1)], 2, 32, 1),
                                             ([(9, 11), (5, 6), (3, 3)], 4, 16, 3), ([(5, 7)], 2, 64, 5), ([(4, 4), (2, 2)], 1, 128, 6)])
def test_msda_generic_levels_and_points(ops, dtype, shapes, m, d, npts):
    """n_levels up to 8 and ANY n_points (`--num_feature_levels`, `--enc_n_points`, `--dec_n_points`: main.py:71,100-101) -- the
    generic kernels (msda_gen_*) behind the upstream-shaped boundary, against the float64 closed form."""
    value, loc, attn, gout = _msda_inputs(17, shapes, m=m, d=d, p=npts)
    geom = ops.LevelGeom(shapes)
    tv, tl, ta, tg = (torch.from_numpy(a).to(dtype) for a in (value, loc, attn, gout))
    out = torch.empty(tg.shape, dtype=dtype, device="cuda")
    ops.msda_fwd(dev(tv), geom, dev(tl), dev(ta), out)
    ref = msda_explicit.msda_forward(tv.float().numpy(), shapes, tl.float().numpy(), ta.float().numpy())
    _close(out, torch.from_numpy(ref), dtype, msg="msda generic fwd")
    gv = torch.empty(tv.shape, dtype=torch.float32, device="cuda")
    gl = torch.empty(tl.shape, dtype=dtype, device="cuda")
    ga = torch.empty(ta.shape, dtype=dtype, device="cuda")
    ops.msda_bwd(dev(tv), geom, dev(tl), dev(ta), dev(tg), gv, gl, ga)
    rv, rl, ra = msda_explicit.msda_backward(tv.float().numpy(), shapes, tl.float().numpy(), ta.float().numpy(), tg.float().numpy())
    _close(gv, torch.from_numpy(rv), dtype, scale=3, msg="msda generic dvalue")
    _close(ga, torch.from_numpy(ra), dtype, scale=3, msg="msda generic dattn")
    _close(gl, torch.from_numpy(rl), dtype, scale=30, msg="msda generic dloc")


@pytest.mark.parametrize("vdt,qdt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("shapes,npts", [([(10, 12), (5, 6), (3, 3), (2, 2), (1, 1)], 3), ([(7, 9), (4, 5)], 5)])
def test_msda_fused_generic(ops, vdt, qdt, shapes, npts):
    """The fused form (softmax over L x P logits, loc = ref + offset / (W, H), softmax Jacobian) of the generic kernels: rows of
    3 M L P = 180 / 120 elements (no 16-byte alignment), head-major value maps, against autograd through the oracle's core."""
    n, m, d, lq, p = 2, 4, 16, 29, npts
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(7)
    value = torch.from_numpy(rng.standard_normal((n, S, m, d)).astype(np.float32)).to(vdt)
    mlp = m * L * p
    oa = torch.from_numpy(np.concatenate([rng.standard_normal((n, lq, 2 * mlp)) * 2.0, rng.standard_normal((n, lq, mlp))], -1)
                          .astype(np.float32)).to(qdt)
    ref_pts = torch.from_numpy(rng.uniform(-0.1, 1.1, (n, lq, L, 2)).astype(np.float32))
    gout = torch.from_numpy(rng.standard_normal((n, lq, m * d)).astype(np.float32)).to(qdt)
    v32 = value.float().requires_grad_()
    oa32 = oa.float().requires_grad_()
    off = oa32[..., : 2 * mlp].view(n, lq, m, L, p, 2)
    w = torch.softmax(oa32[..., 2 * mlp:].view(n, lq, m, L * p), -1).view(n, lq, m, L, p)
    norm = torch.tensor([[wd, ht] for ht, wd in shapes], dtype=torch.float32)
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out_ref = poet_ref.msda_core(v32, shapes, loc, w)
    (out_ref * gout.float()).sum().backward()
    vdev = dev(value.permute(0, 2, 1, 3).contiguous())
    vstr = (m * S * d, d, S * d)
    out = torch.empty(n, lq, m * d, dtype=qdt, device="cuda")
    ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, dev(ref_pts), lq * L * 2, out, n, m, d, p, lq)
    _close(out, out_ref.detach(), vdt if vdt == torch.bfloat16 else qdt, msg="fused generic fwd")
    gv = torch.zeros(vdev.shape, dtype=torch.float32, device="cuda")
    goa = torch.empty_like(dev(oa))
    ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, dev(ref_pts), lq * L * 2, dev(gout), gv, goa, n, m, d, p, lq)
    worst = torch.bfloat16 if torch.bfloat16 in (vdt, qdt) else torch.float32
    _close(gv, v32.grad.permute(0, 2, 1, 3), worst, scale=3, msg="fused generic dvalue")
    _close(goa, oa32.grad, worst, scale=3, msg="fused generic d(off|logit)")


@pytest.mark.parametrize("sdt,ddt", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("H,W", [(3, 4), (6, 8), (5, 7), (1, 1)])
def test_col2im3x3s2_add_is_the_adjoint_of_the_extra_level_conv(ops, sdt, ddt, H, W):
    """poet_col2im3x3s2_add == d(input) of nn.Conv2d(C, d, 3, stride 2, padding 1) given d(im2col matrix), added into the token
    rows of a wider stream gradient (the chained extra levels of pose_estimation_transformer.py:327-330)."""
    N, C = 2, 8
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    rng = np.random.default_rng(3)
    dcol = torch.from_numpy(rng.standard_normal((N * Ho * Wo, C * 9)).astype(np.float32)).to(sdt)
    S, off = H * W + 11, 5
    base = torch.from_numpy(rng.standard_normal((N, S, C)).astype(np.float32)).to(ddt)
    x = torch.zeros(N, C, H, W, requires_grad=True)
    cols = torch.nn.functional.unfold(x, 3, padding=1, stride=2)                   # (N, C*9, Ho*Wo), k = c*9 + ky*3 + kx
    (cols.transpose(1, 2).reshape(N * Ho * Wo, C * 9) * dcol.float()).sum().backward()
    want = base.float().clone()
    want[:, off:off + H * W] += x.grad.permute(0, 2, 3, 1).reshape(N, H * W, C)
    got = dev(base)
    ops.col2im3x3s2_add(dev(dcol), got, N, C, H, W, Ho, Wo, off, S)
    _close(got, want, ddt, scale=3, msg="col2im")


@pytest.mark.parametrize("vdt,qdt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("head_major", [False, True])
def test_msda_fused(ops, vdt, qdt, head_major):
    shapes = [(10, 12), (5, 6), (3, 3)]
    n, m, d, lq, p = 2, 4, 16, 29, 4
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(5)
    value = torch.from_numpy(rng.standard_normal((n, S, m, d)).astype(np.float32)).to(vdt)
    mlp = m * L * p
    oa = torch.from_numpy(np.concatenate([rng.standard_normal((n, lq, 2 * mlp)) * 2.0, rng.standard_normal((n, lq, mlp))], -1)
                          .astype(np.float32)).to(qdt)
    ref_pts = torch.from_numpy(rng.uniform(-0.1, 1.1, (n, lq, L, 2)).astype(np.float32))
    gout = torch.from_numpy(rng.standard_normal((n, lq, m * d)).astype(np.float32)).to(qdt)
    # oracle: compose softmax + loc arithmetic + grid_sample core with autograd
    v32 = value.float().requires_grad_()
    oa32 = oa.float().requires_grad_()
    off = oa32[..., : 2 * mlp].view(n, lq, m, L, p, 2)
    w = torch.softmax(oa32[..., 2 * mlp:].view(n, lq, m, L * p), -1).view(n, lq, m, L, p)
    norm = torch.tensor([[wd, ht] for ht, wd in shapes], dtype=torch.float32)
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out_ref = poet_ref.msda_core(v32, shapes, loc, w)
    (out_ref * gout.float()).sum().backward()

    if head_major:
        vdev = dev(value.permute(0, 2, 1, 3).contiguous())
        vstr = (m * S * d, d, S * d)
    else:
        vdev = dev(value)
        vstr = (S * m * d, m * d, d)
    out = torch.empty(n, lq, m * d, dtype=qdt, device="cuda")
    ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, dev(ref_pts), lq * L * 2, out, n, m, d, p, lq)
    _close(out, out_ref.detach(), vdt if vdt == torch.bfloat16 else qdt, msg="fused fwd")
    gv = torch.zeros(vdev.shape, dtype=torch.float32, device="cuda")
    goa = torch.empty_like(dev(oa))
    ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, dev(ref_pts), lq * L * 2, dev(gout), gv, goa, n, m, d, p, lq)
    gv_ref = v32.grad.permute(0, 2, 1, 3) if head_major else v32.grad
    worst = torch.bfloat16 if torch.bfloat16 in (vdt, qdt) else torch.float32
    _close(gv, gv_ref, worst, scale=3, msg="fused dvalue")
    _close(goa, oa32.grad, worst, scale=3, msg="fused d(off|logit)")


@pytest.mark.parametrize("gdt", [torch.bfloat16, torch.float32])
def test_msda_fused_bwd_scatter_into_token_rows(ops, gdt):
    """The decoder's value-gradient path (functional.DecoderFn.backward, bf16 policy): head-major value maps, but grad_value is a column
    block of token-major rows (N*S, layers*M*D) addressed by its own strides (poet_msda_fused_bwd gv_strides) and, in bf16, accumulated
    with packed bf16x2 atomics; masked rows are zeroed afterwards (poet_zero_masked_rows).  Reference: autograd of the explicit
    sampling (models/deformable_transformer.py:300-340 cross-attention, ops/functions/ms_deform_attn_func.py:41-61)."""
    shapes = [(10, 12), (5, 6), (3, 3)]
    n, m, d, lq, p, nl, layer = 2, 4, 16, 23, 4, 3, 1
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(11)
    value = torch.from_numpy(rng.standard_normal((n, S, m, d)).astype(np.float32)).to(torch.bfloat16)
    mlp = m * L * p
    oa = torch.from_numpy(np.concatenate([rng.standard_normal((n, lq, 2 * mlp)) * 2.0, rng.standard_normal((n, lq, mlp))], -1).astype(np.float32))
    ref_pts = torch.from_numpy(rng.uniform(-0.1, 1.1, (n, lq, L, 2)).astype(np.float32))
    gout = torch.from_numpy(rng.standard_normal((n, lq, m * d)).astype(np.float32))
    v32 = value.float().requires_grad_()
    oa32 = oa.clone().requires_grad_()
    off = oa32[..., : 2 * mlp].view(n, lq, m, L, p, 2)
    w = torch.softmax(oa32[..., 2 * mlp:].view(n, lq, m, L * p), -1).view(n, lq, m, L, p)
    norm = torch.tensor([[wd, ht] for ht, wd in shapes], dtype=torch.float32)
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    (poet_ref.msda_core(v32, shapes, loc, w) * gout).sum().backward()

    vdev = dev(value.permute(0, 2, 1, 3).contiguous())                   # head-major (N, M, S, D)
    vstr = (m * S * d, d, S * d)
    rows = torch.full((n * S, nl * m * d), 7.0, dtype=gdt, device="cuda")  # the other layers' columns must stay untouched
    blk = rows.view(n, S, nl, m, d)[:, :, layer]
    blk.zero_()
    gv = blk.permute(0, 2, 1, 3)                                         # (N, M, S, D) view of the column block
    goa = torch.empty_like(dev(oa))
    ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, dev(ref_pts), lq * L * 2, dev(gout), gv, goa, n, m, d, p, lq,
                       gv_strides=(gv.stride(0), gv.stride(2), gv.stride(1)))
    got = rows.view(n, S, nl, m, d)[:, :, layer].float().cpu()
    _close(got, v32.grad, torch.bfloat16, scale=3, msg="dvalue rows")
    _close(goa, oa32.grad, torch.bfloat16, scale=3, msg="d(off|logit)")
    assert (rows.view(n, S, nl, m, d)[:, :, [0, 2]] == 7.0).all()
    mask = torch.from_numpy(rng.uniform(size=n * S) < 0.3).cuda()
    ops.zero_masked_rows(rows, mask, n * S, nl * m * d)
    out = rows.float().cpu()
    assert (out[mask.cpu()] == 0).all()
    keep = (~mask).cpu()
    assert torch.equal(out.view(n * S, nl, m * d)[keep][:, layer], got.reshape(n * S, m * d)[keep])
    assert (out.view(n * S, nl, m * d)[keep][:, [0, 2]] == 7.0).all()


# ---------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d", [256, 64])
def test_layernorm(ops, dtype, d):
    rows = 203
    x = _rand(rows, d, seed=20).to(dtype)
    r = _rand(rows, d, seed=21).to(dtype)
    gamma = 1 + 0.1 * _rand(d, seed=22)
    beta = 0.1 * _rand(d, seed=23)
    y = torch.empty(rows, d, dtype=dtype, device="cuda")
    z = torch.empty_like(y)
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    ops.ln_fwd(dev(x), dev(r), dev(gamma), dev(beta), y, z, mean, rstd, rows, d)
    zr = (x.float() + r.float()).requires_grad_()
    g32, b32 = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr = F.layer_norm(zr, (d,), g32, b32, 1e-5)
    _close(y, yr.detach(), dtype, msg="ln fwd")
    dy = _rand(rows, d, seed=24).to(dtype)
    yr.backward(dy.float())
    dz = torch.empty_like(y)
    dg = torch.zeros(d, device="cuda")
    db = torch.zeros(d, device="cuda")
    zsaved = z if dtype == torch.float32 else dev(zr.detach().to(dtype))
    ops.ln_bwd(dev(dy), zsaved, mean, rstd, dev(gamma), dz, None, dg, db, rows, d)
    _close(dz, zr.grad, dtype, msg="ln dz")
    _close(dg, g32.grad, dtype, scale=10, msg="ln dgamma")
    _close(db, b32.grad, dtype, scale=10, msg="ln dbeta")


def test_layernorm_dropout_consistency(ops):
    rows, d = 64, 256
    x = torch.ones(rows, d)
    r = torch.zeros(rows, d)
    g, b = torch.ones(d), torch.zeros(d)
    y = torch.empty(rows, d, device="cuda"); z = torch.empty_like(y)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    ops.ln_fwd(dev(x), dev(r), dev(g), dev(b), y, z, mean, rstd, rows, d, drop_p=0.25, seed=7)
    zc = z.cpu()
    keep = zc != 0
    assert abs(keep.float().mean().item() - 0.75) < 0.03
    dy = torch.ones(rows, d)
    dz = torch.empty_like(y); dx = torch.empty_like(y)
    dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
    ops.ln_bwd(dev(dy), z, mean, rstd, dev(g), dz, dx, dg, db, rows, d, drop_p=0.25, seed=7)
    dzc, dxc = dz.cpu(), dx.cpu()
    assert torch.allclose(dxc[keep], dzc[keep] / 0.75, atol=1e-5)
    assert (dxc[~keep] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_groupnorm(ops, dtype):
    N, HW, C, G, S, off = 2, 50, 64, 32, 80, 20
    x = _rand(N, HW, C, seed=30).to(dtype)
    gamma = 1 + 0.1 * _rand(C, seed=31)
    beta = 0.1 * _rand(C, seed=32)
    y = torch.zeros(N, S, C, dtype=dtype, device="cuda")
    stats = torch.empty(N, G, 2, device="cuda")
    x16 = torch.empty(N, HW, C, dtype=torch.bfloat16, device="cuda") if dtype == torch.float32 else None
    ops.groupnorm_fwd(dev(x), dev(gamma), dev(beta), y, stats, N, HW, C, G, 0, HW, off, S, x16=x16)
    if x16 is not None:                                                 # (the bf16 copy of an fp32 input, written in the same pass)
        assert torch.equal(x16.cpu(), x.to(torch.bfloat16))
    xr = x.float().permute(0, 2, 1).contiguous().requires_grad_()       # (N,C,HW)
    g32, b32 = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr = F.group_norm(xr, G, g32, b32, 1e-5)
    _close(y[:, off:off + HW], yr.detach().permute(0, 2, 1), dtype, msg="gn fwd")
    dy = torch.zeros(N, S, C)
    dy[:, off:off + HW] = _rand(N, HW, C, seed=33)
    dy = dy.to(dtype)
    yr.backward(dy[:, off:off + HW].float().permute(0, 2, 1))
    dx = torch.empty(N, HW, C, dtype=dtype, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    ops.groupnorm_bwd(dev(dy), dev(x), stats, dev(gamma), dx, dg, db, N, HW, C, G, 0, HW, off, S)
    _close(dx, xr.grad.permute(0, 2, 1), dtype, msg="gn dx")
    _close(dg, g32.grad, dtype, scale=5, msg="gn dgamma")
    _close(db, b32.grad, dtype, scale=5, msg="gn dbeta")


# ---------------------------------------------------------------------------------------------- MHA
@pytest.mark.parametrize("Q,M,hd", [(20, 16, 16), (10, 4, 64), (6, 4, 16), (50, 8, 32), (64, 16, 16), (65, 4, 16), (100, 16, 16), (128, 16, 16), (100, 8, 32),
                                    (70, 4, 64), (128, 8, 32), (114, 4, 64)])
def test_mha(ops, Q, M, hd):
    """(Q > 64: two waves per (image, head); backward with k / v and q / d(out) sharing their LDS; `--num_queries`, main.py:98.
    (128, 8, 32) = the reference's default head geometry (hidden 256 / nheads 8) at the largest query count: the backward's score
    matrices fill the CU's 160 KB exactly (swizzled, no pad column); (114, 4, 64): the largest Q the backward takes at head dim 64.)"""
    N, d = 3, M * hd
    packed = _rand(N * Q, 3 * d, seed=40)
    pk = dev(packed)
    out = torch.empty(N * Q, d, device="cuda")
    ops.mha_fwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, out, d, N, Q, M, hd)
    p32 = packed.clone().requires_grad_()
    q, k, v = (p32[:, i * d:(i + 1) * d].view(N, Q, M, hd).transpose(1, 2) for i in range(3))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    ref = (att @ v).transpose(1, 2).reshape(N * Q, d)
    _close(out, ref.detach(), torch.float32, msg="mha fwd")
    dout = _rand(N * Q, d, seed=41)
    ref.backward(dout)
    dpk = torch.empty_like(pk)
    ops.mha_bwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, dev(dout), d, dpk, dpk[:, d:], dpk[:, 2 * d:], 3 * d, N, Q, M, hd)
    _close(dpk, p32.grad, torch.float32, msg="mha bwd")


def test_mha_limits_are_refused_not_misrun(ops):
    """Beyond the generic form's row buffers (Q > 2048) the call returns POET_ERR_UNSUPPORTED instead of misrunning."""
    from poet_amd._lib import PoetHipError
    Q, M, hd = 2049, 1, 16
    N, d = 1, M * hd
    pk = dev(_rand(N * Q, 3 * d, seed=44))
    out = torch.empty(N * Q, d, device="cuda")
    with pytest.raises(PoetHipError):
        ops.mha_fwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, out, d, N, Q, M, hd)
    dpk = torch.empty_like(pk)
    with pytest.raises(PoetHipError):
        ops.mha_bwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, out, d, dpk, dpk[:, d:], dpk[:, 2 * d:], 3 * d, N, Q, M, hd)


@pytest.mark.parametrize("Q,M,hd", [(129, 4, 16), (300, 8, 32), (20, 4, 24), (115, 4, 64), (40, 2, 128), (7, 3, 5), (1, 2, 16), (600, 2, 48)])
def test_mha_generic_any_query_count_and_head_dim(ops, Q, M, hd):
    """`--num_queries`, `--hidden_dim` and `--nheads` are free parameters of the reference (main.py:94-98; nn.MultiheadAttention at
    deformable_transformer.py:253 has no limit): what the LDS-resident kernels do not take (Q > 128, head dims outside {16, 32, 64},
    the backward beyond 114 queries at head dim 64) runs the generic kernels -- against torch fp32."""
    N, d = 2, M * hd
    packed = _rand(N * Q, 3 * d, seed=46)
    pk = dev(packed)
    out = torch.empty(N * Q, d, device="cuda")
    ops.mha_fwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, out, d, N, Q, M, hd)
    p32 = packed.clone().requires_grad_()
    q, k, v = (p32[:, i * d:(i + 1) * d].view(N, Q, M, hd).transpose(1, 2) for i in range(3))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    ref = (att @ v).transpose(1, 2).reshape(N * Q, d)
    _close(out, ref.detach(), torch.float32, msg="mha generic fwd")
    dout = _rand(N * Q, d, seed=47)
    ref.backward(dout)
    dpk = torch.empty_like(pk)
    ops.mha_bwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, dev(dout), d, dpk, dpk[:, d:], dpk[:, 2 * d:], 3 * d, N, Q, M, hd)
    _close(dpk, p32.grad, torch.float32, msg="mha generic bwd")


@pytest.mark.parametrize("Q,M,hd", [(24, 2, 24), (130, 2, 130), (20, 4, 32)])
def test_mha_dropout_forward_and_backward_share_one_mask(ops, Q, M, hd):
    """Dropout on the attention probabilities (nn.MultiheadAttention(dropout=...), deformable_transformer.py:253): nothing is stored,
    forward and backward redraw the mask from the same counter.  With v = identity (hd >= Q) the forward output IS the dropped
    probability row, so the mask can be read off it; the backward must then equal autograd through softmax * mask / (1 - p).
    Generic kernels (head dim 24 / 130) and the LDS-resident ones (head dim 32) alike."""
    N, d, pdrop, seed = 2, M * hd, 0.25, 1234
    packed = _rand(N * Q, 3 * d, seed=48)
    assert hd >= Q
    eye = torch.zeros(Q, hd); eye[:, :Q] = torch.eye(Q)
    packed.view(N, Q, 3, M, hd)[:, :, 2] = eye[None, :, None, :]
    pk = dev(packed)
    out = torch.empty(N * Q, d, device="cuda")
    ops.mha_fwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, out, d, N, Q, M, hd, pdrop, seed)
    p32 = packed.clone().requires_grad_()
    q, k, v = (p32[:, i * d:(i + 1) * d].view(N, Q, M, hd).transpose(1, 2) for i in range(3))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)                  # (N, M, Q, Q)
    dropped = out.cpu().view(N, Q, M, hd).transpose(1, 2)[..., :Q]
    mask = (dropped != 0).float()
    frac = 1.0 - mask.mean().item()
    assert abs(frac - pdrop) < 0.05, frac
    _close(dev(dropped), (att * mask / (1 - pdrop)).detach(), torch.float32, msg="dropped probabilities")
    ref = ((att * mask / (1 - pdrop)) @ v).transpose(1, 2).reshape(N * Q, d)
    dout = _rand(N * Q, d, seed=49)
    ref.backward(dout)
    dpk = torch.empty_like(pk)
    ops.mha_bwd(pk, pk[:, d:], pk[:, 2 * d:], 3 * d, dev(dout), d, dpk, dpk[:, d:], dpk[:, 2 * d:], 3 * d, N, Q, M, hd, pdrop, seed)
    _close(dpk, p32.grad, torch.float32, msg="mha dropout bwd")


# ---------------------------------------------------------------------------------------------- encodings & misc
def test_pos_sine_and_valid_ratio(ops, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "units.npz"))
    mask = torch.from_numpy(g["pe_mask"])
    N, H, W = mask.shape
    out = torch.empty(N, H * W, 256, device="cuda")
    ops.pos_sine(dev(mask.to(torch.uint8)), out, None, N, H, W, 128, 0, H * W)
    ref = torch.from_numpy(g["pe_out"]).flatten(2).transpose(1, 2)      # reference's own output
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    vr = torch.empty(N, 1, 2, device="cuda")
    ops.valid_ratio(dev(mask.to(torch.uint8)), vr, 2, N, H, W)
    ref_vr = poet_ref.DeformableTransformer.valid_ratio(mask)
    assert torch.allclose(vr.cpu()[:, 0], ref_vr, atol=1e-7)


def test_bbox_sine_full_range(ops, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "units.npz"))
    boxes = torch.from_numpy(g["bbox_in"])
    n = boxes.shape[0]
    out = torch.empty(n, 256, device="cuda")
    ops.bbox_sine(dev(boxes), out, n, 32)
    err = (out.cpu() - torch.from_numpy(g["bbox_out"])).abs().max().item()
    assert err < 5e-6, err                                               # args reach 2^31: needs exact range reduction
    valid = torch.ones(n, dtype=torch.uint8); valid[-1] = 0
    ops.bbox_sine(dev(boxes), out, n, 32, valid=dev(valid))
    assert (out.cpu()[-1] == -10).all()


def test_ref_points_and_misc(ops):
    shapes = [(6, 8), (3, 4)]
    geom = ops.LevelGeom(shapes)
    vr = torch.tensor([[[1.0, 1.0], [1.0, 1.0]], [[0.75, 0.5], [0.75, 0.6667]]])
    ref = torch.empty(2, geom.S, 2, 2, device="cuda")
    ops.enc_ref_points(dev(vr), geom, ref, 2)
    want = poet_ref.Encoder.reference_grid(torch.tensor(shapes), vr)
    assert torch.allclose(ref.cpu(), want, atol=1e-6)
    r2 = torch.rand(2, 5, 2)
    o = torch.empty(2, 5, 2, 2, device="cuda")
    ops.dec_ref_points(dev(r2), dev(vr), o, 2, 5, 2)
    assert torch.allclose(o.cpu(), r2[:, :, None] * vr[:, None], atol=1e-7)
    # mask nearest == F.interpolate
    m = (torch.rand(2, 96, 128) > 0.5)
    for size in [(12, 16), (3, 4), (8, 10)]:
        dst = torch.empty(2, *size, dtype=torch.uint8, device="cuda")
        ops.mask_nearest(dev(m.to(torch.uint8)), dst, 2, 96, 128, *size)
        want = F.interpolate(m[None].float(), size=size).to(torch.bool)[0]
        assert torch.equal(dst.cpu().bool(), want)
    # transposes, im2col, add, cast, colsum, rowvec
    x = _rand(2, 24, 7, 9, seed=50)
    tok = torch.zeros(2, 100, 24, device="cuda")
    ops.nchw_to_tokens(dev(x), tok, 2, 24, 63, 10, 100)
    assert torch.allclose(tok.cpu()[:, 10:73], x.flatten(2).transpose(1, 2))
    back = torch.empty(2, 24, 63, device="cuda")
    ops.tokens_to_nchw(tok, back, 2, 24, 63, 10, 100)
    assert torch.allclose(back.cpu(), x.flatten(2))
    col = torch.empty(2 * 4 * 5, 24 * 9, device="cuda")
    ops.im2col3x3s2(dev(x), col, 2, 24, 7, 9, 4, 5)
    want = F.unfold(x, 3, padding=1, stride=2).transpose(1, 2).reshape(2 * 20, 24 * 9)
    assert torch.allclose(col.cpu(), want)
    a, b = _rand(1000, seed=51), _rand(1000, seed=52)
    o = torch.empty(1000, device="cuda")
    ops.add(dev(a), dev(b), o)
    assert torch.allclose(o.cpu(), a + b)
    ob = torch.empty(1000, dtype=torch.bfloat16, device="cuda")
    ops.cast(dev(a), ob)
    assert torch.equal(ob.cpu(), a.to(torch.bfloat16))
    xs = _rand(3, 70, 40, seed=53)
    cs = torch.zeros(2, 40, device="cuda")
    ops.colsum(dev(xs), 40, cs, 3, 70, 40, ops._i64arr([0, 48, 70]), 2)
    want = torch.stack([xs[:, :48].sum((0, 1)), xs[:, 48:].sum((0, 1))])
    assert torch.allclose(cs.cpu(), want, atol=1e-4)
    t = dev(xs.clone())
    vecr = _rand(40, seed=54)
    ops.add_rowvec(t, dev(vecr), 3, 70, 48, 22, 40)
    w2 = xs.clone(); w2[:, 48:] += vecr
    assert torch.allclose(t.cpu(), w2)


def test_pose_finish(ops, golden_dir):
    R, ncls = 40, 6
    rot_all = _rand(R, ncls * 6, seed=60)
    trans_all = _rand(R, ncls * 3, seed=61)
    cls = torch.randint(-1, ncls, (R,), generator=torch.Generator().manual_seed(62), dtype=torch.int32)
    rot = torch.empty(R, 3, 3, device="cuda"); tr = torch.empty(R, 3, device="cuda")
    ops.pose_finish_fwd(dev(rot_all), dev(trans_all), dev(cls), rot, tr, R, ncls)
    ra = rot_all.clone().requires_grad_(); ta = trans_all.clone().requires_grad_()
    idx = torch.where(cls > 0, cls, 0).long()
    r6 = ra.view(R, ncls, 6)[torch.arange(R), idx]
    t3 = ta.view(R, ncls, 3)[torch.arange(R), idx]
    rref = poet_ref.rotation_6d_to_matrix(r6[None])[0]
    assert torch.allclose(rot.cpu(), rref.detach(), atol=1e-5)
    assert torch.allclose(tr.cpu(), t3.detach())
    drot = _rand(R, 3, 3, seed=63); dtr = _rand(R, 3, seed=64)
    ((rref * drot).sum() + (t3 * dtr).sum()).backward()
    dra = torch.empty(R, ncls * 6, device="cuda"); dta = torch.empty(R, ncls * 3, device="cuda")
    ops.pose_finish_bwd(dev(rot_all), dev(cls), dev(drot), dev(dtr), dra, dta, R, ncls)
    assert torch.allclose(dra.cpu(), ra.grad, atol=2e-4)
    assert torch.allclose(dta.cpu(), ta.grad, atol=1e-6)


@pytest.mark.parametrize("n", [5000, 5003, 70])
def test_adamw_vector_path_equals_element_path(ops, n):
    """poet_adamw updates four consecutive elements per thread when its buffers are 16-byte aligned and falls back to one element at
    a time otherwise: the two paths must agree bit for bit, including the per-64-element learning-rate table and the two bf16 weight images (optimizer step of main.py:300-305 / torch.optim.AdamW)."""
    p0 = _rand(n + 1, seed=170); g0 = _rand(n + 1, seed=171) * 0.01
    lrs = dev(torch.linspace(0.1, 1.0, (n + 64) // 64 + 1))
    outs = []
    for off in (0, 1):                                                   # off = 1: every buffer misaligned by one element
        pad = 4 - off                                                    # keep the SAME element <-> learning-rate-group relation in both runs
        def buf(t, dt=torch.float32):
            full = torch.zeros(pad + n + 8, dtype=dt, device="cuda")
            if t is not None:
                full[pad: pad + n] = t[:n].to(dt).cuda()
            return full
        P, G, M_, V_ = buf(p0), buf(g0), buf(None), buf(None)
        H, L = buf(None, torch.bfloat16), buf(None, torch.bfloat16)
        if off == 0:
            views = [t[pad: pad + n] for t in (P, G, M_, V_, H, L)]     # pad = 4 floats: 16-byte aligned
        else:
            views = [t[pad: pad + n] for t in (P, G, M_, V_, H, L)]     # pad = 3 floats: misaligned
        p, g, m, v, h, l = views
        assert (p.data_ptr() % 16 == 0) == (off == 0)
        sq = torch.zeros(1 + 1024, device="cuda")
        for step in (1, 2, 3):
            sq.zero_(); ops.sqnorm(g, sq)
            # (the learning-rate group of element k is k >> 6 relative to the view's first element in both runs)
            ops.adamw(p, g, m, v, n, 2e-4, 0.9, 0.999, 1e-8, 1e-4, step, sqnorm_buf=sq, max_norm=0.1, p_bf16=h, lr_scale=lrs, p_bf16_lo=l)
        outs.append([t.clone().float().cpu() for t in (p, m, v, h, l)])
    for a, b, name in zip(outs[0], outs[1], ("p", "m", "v", "hi", "lo")):
        assert torch.equal(a, b), name
    assert torch.equal(outs[0][3] + outs[0][4], (outs[0][0].to(torch.bfloat16).float() + (outs[0][0] - outs[0][0].to(torch.bfloat16).float()).to(torch.bfloat16).float()))


def test_adamw_and_clip(ops):
    n = 5000
    p0 = _rand(n, seed=70); g = _rand(n, seed=71) * 0.01
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=2e-4, weight_decay=1e-4)
    p = dev(p0.clone()); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gs = g * step
        ref_p.grad = gs.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 0.1)
        opt.step()
        sq = torch.zeros(1 + 1024, device="cuda")      # [0] = sum, [1:] = partials (POET_SQNORM_SCRATCH)
        ops.sqnorm(dev(gs), sq)
        sq2 = torch.zeros(1 + 1024, device="cuda")
        ops.sqnorm(dev(gs), sq2)
        assert torch.equal(sq[:1], sq2[:1])              # bit-reproducible (no atomics)
        ops.adamw(p, dev(gs), m, v, n, 2e-4, 0.9, 0.999, 1e-8, 1e-4, step, sqnorm_buf=sq, max_norm=0.1)
    assert torch.allclose(p.cpu(), ref_p.detach(), atol=1e-6)


# ---------------------------------------------------------------------------------------------- mixed precision
def test_gemm_mixed_dtypes(ops):
    """fp32 residual stream in, bf16 branch out (and back): the 'bf16' training policy's operand mixes."""
    M, N, K = 700, 256, 256
    x32 = _rand(M, K, seed=80)
    w = _rand(N, K, seed=81, scale=1 / math.sqrt(K))
    out_b = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.linear_fwd(dev(x32), dev(w), None, out_b)                              # (f32, f32) -> bf16, bf16 MFMA
    ref = x32.bfloat16().float() @ w.bfloat16().float().t()
    _close(out_b, ref, torch.bfloat16, msg="f32->bf16")
    dy = _rand(M, N, seed=82).bfloat16()
    acc = _rand(M, K, seed=83)
    out32 = dev(acc.clone())
    ops.linear_dx(dev(dy), dev(w), out32, rows=M, add_src=out32)               # bf16 x f32 -> f32 (+= stream grad)
    ref = dy.float() @ w.bfloat16().float() + acc
    _close(out32, ref, torch.bfloat16, msg="bf16->f32 accumulate")
    dw = torch.zeros(N, K, device="cuda")
    ops.linear_dw(dev(dy), dev(x32), dw, rows=M)                               # bf16^T x f32 -> f32 atomics
    ref = dy.float().t() @ x32.bfloat16().float()
    _close(dw, ref, torch.bfloat16, scale=math.sqrt(M), msg="mixed dW")


def test_layernorm_mixed(ops):
    rows, d = 130, 256
    x = _rand(rows, d, seed=90).bfloat16()
    r = _rand(rows, d, seed=91)
    gamma = 1 + 0.1 * _rand(d, seed=92); beta = 0.1 * _rand(d, seed=93)
    y = torch.empty(rows, d, device="cuda"); z = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    ops.ln_fwd(dev(x), dev(r), dev(gamma), dev(beta), y, z, mean, rstd, rows, d)
    zr = (x.float() + r).requires_grad_()
    yr = F.layer_norm(zr, (d,), gamma, beta, 1e-5)
    assert y.dtype == torch.float32 and (y.cpu() - yr.detach()).abs().max().item() < 1e-5
    dy = _rand(rows, d, seed=94)
    yr.backward(dy)
    dz = torch.empty(rows, d, device="cuda"); dx = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
    dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
    ops.ln_bwd(dev(dy), z, mean, rstd, dev(gamma), dz, dx, dg, db, rows, d)
    _close(dz, zr.grad, torch.bfloat16, msg="mixed ln dz")
    _close(dx, zr.grad, torch.bfloat16, msg="mixed ln dx")


@pytest.mark.parametrize("dtype,xdtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
def test_gelu_dropout_forward_backward(ops, dtype, xdtype):
    """poet_gelu_fwd / poet_gelu_bwd (the FFN with activation="gelu", deformable_transformer.py:347-355): F.gelu (erf form) and its
    derivative on the kept pre-activation (stored in the output's type, or fp32 under bf16 storage); the backward redraws the forward's
    dropout mask."""
    n = 5 * 2048 + 13
    x = _rand(n, seed=180, scale=2.5).to(xdtype)
    y = torch.empty(n, dtype=dtype, device="cuda")
    ops.gelu_fwd(dev(x), y)
    xr = x.float().requires_grad_()
    yr = F.gelu(xr)
    _close(y, yr.detach(), dtype, msg="gelu fwd")
    dy = _rand(n, seed=181).to(dtype)
    yr.backward(dy.float())
    dx = torch.empty_like(y)
    ops.gelu_bwd(dev(dy), dev(x), dx)
    _close(dx, xr.grad, dtype, msg="gelu bwd")
    yd = torch.empty_like(y)
    ops.gelu_fwd(dev(x), yd, 0.25, 77)
    keep = (yd != 0) | (y == 0)
    assert abs(keep.float().mean().item() - 0.75) < 0.02
    _close(yd[keep], y[keep].float() / 0.75, dtype, msg="gelu dropout scale")
    dxd = torch.empty_like(y)
    ops.gelu_bwd(dev(dy), dev(x), dxd, 0.25, 77)
    assert bool((dxd[~keep] == 0).all())
    _close(dxd[keep], dx[keep].float() / 0.75, dtype, msg="gelu dropout bwd")


@pytest.mark.parametrize("dy_bf16", [False, True])
def test_layernorm_backward_bf16_stream(ops, dy_bf16):
    """The encoder's bf16 gradient stream (blocks.enc_layer_bwd): LayerNorm backward that stores d(z) -- the gradient of the residual
    stream -- as bf16, from an fp32 d(y) (where the stream starts: the top layer, fed by the decoder) or a bf16 one (below).  Equal to
    the fp32-stream launch on the same operands rounded once; branch gradient (with dropout) and parameter gradients unchanged."""
    rows, d = 4099, 256
    z = _rand(rows, d, seed=190, scale=2.0).bfloat16()
    gamma = 1 + 0.1 * _rand(d, seed=191)
    dy = _rand(rows, d, seed=192)
    if dy_bf16:
        dy = dy.bfloat16()
    zf = z.float()
    mean = zf.mean(1)
    rstd = (zf.var(1, unbiased=False) + 1e-5).rsqrt()
    outs = []
    for stream in (torch.float32, torch.bfloat16):
        dyd = dev(dy.float()) if stream == torch.float32 else dev(dy)
        dz = torch.empty(rows, d, dtype=stream, device="cuda")
        dx = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
        dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
        ops.ln_bwd(dyd, dev(z), dev(mean), dev(rstd), dev(gamma), dz, dx, dg, db, rows, d, drop_p=0.1, seed=5)
        outs.append((dz, dx, dg, db))
    assert torch.equal(outs[1][0].cpu(), outs[0][0].cpu().bfloat16())
    assert torch.equal(outs[1][1], outs[0][1])
    assert torch.allclose(outs[1][2], outs[0][2], rtol=1e-4, atol=1e-3) and torch.allclose(outs[1][3], outs[0][3], rtol=1e-4, atol=1e-3)
    zr = zf.clone().requires_grad_()
    F.layer_norm(zr, (d,), gamma, torch.zeros(d), 1e-5).backward(dy.float())
    _close(outs[1][0], zr.grad, torch.bfloat16, msg="bf16-stream ln dz")


@pytest.mark.parametrize("rows,K,acc", [(8192 + 40, 1024, True), (4096, 1280, True), (5000, 512, False)])
def test_gemm_input_gradient_bf16_stream(ops, rows, K, acc):
    """d(src) (+)= d(y) W with the stream stored as bf16 (gemm_pipe OUT = 2: the accumulators start as the unpacked bf16 C and leave
    rounded to nearest even): against fp64 on the same bf16 operands, within one bf16 rounding of the result; rows past the end untouched."""
    dyv = _rand(rows, K, seed=195).bfloat16()
    w = _rand(K, 256, seed=196, scale=1 / math.sqrt(K)).bfloat16()
    c0 = _rand(rows, 256, seed=197).bfloat16()
    buf = torch.full((rows + 2, 256), 3.0, dtype=torch.bfloat16, device="cuda")
    buf[:rows] = dev(c0)
    out = buf[:rows]
    ops.linear_dx(dev(dyv), dev(w), out, rows=rows, add_src=out if acc else None)
    ref = dyv.double() @ w.double() + (c0.double() if acc else 0.0)
    err = (out.double().cpu() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 1e-3
    assert bool((err <= bound).all()), float((err - bound).max())
    assert bool((buf[rows:] == 3.0).all())


@pytest.mark.parametrize("rows", [4096 + 3, 700])
def test_layernorm_fp16_branch(ops, rows):
    """(round 6) `x = norm(res + dropout(branch))` with the branch stored as IEEE fp16 by the projection before it (dtype_x POET_F16:
    fp32 residual stream, bf16 saved sum, bf16 operand copy and next-layer query copy): equal to the fp32-branch launch on the
    fp16-rounded values BIT FOR BIT (same kernel arithmetic, only the loader differs)."""
    d = 256
    x32 = _rand(rows, d, seed=70, scale=3.0)
    res = _rand(rows, d, seed=71)
    gamma, beta = 1.0 + 0.1 * _rand(d, seed=72), 0.1 * _rand(d, seed=73)
    pos = _rand(rows, d, seed=74).to(torch.bfloat16)
    x16 = x32.to(torch.float16)
    for drop in (0.1, 0.0):
        outs = []
        for xd in (x16, x16.float()):
            y = torch.empty(rows, d, device="cuda")
            z = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
            y16, q16 = torch.empty_like(z), torch.empty_like(z)
            mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
            ops.ln_fwd(dev(xd), dev(res), dev(gamma), dev(beta), y, z, mean, rstd, rows, d, 1e-5, drop, 1234, y16=y16, pos16=dev(pos), q16=q16)
            outs.append((y, z, y16, q16, mean, rstd))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    ref = F.layer_norm(x16.float() + res, (d,), gamma, beta, 1e-5)                   # (the last pass ran without dropout)
    assert (outs[0][0].cpu() - ref).abs().max().item() < 1e-5
    assert torch.equal(outs[0][2].cpu(), outs[0][0].cpu().to(torch.bfloat16))


@pytest.mark.parametrize("N,act,dp,f16out,rows", [(256, 0, 0.0, True, 4096 + 37), (768, 0, 0.0, True, 8192), (1024, 1, 0.0, False, 5000), (1024, 1, 0.1, False, 4096),
                                                  (256, 0, 0.0, False, 4100)])
def test_gemm_fp16_weight_form(ops, N, act, dp, f16out, rows):
    """PoetGemmDesc.b_split = 2 (experimental, POET_W16=1): the fp32 master as ONE IEEE fp16 image, one f16 MFMA per fragment pair, the
    bf16 activation fragments converted to fp16 in registers.  Against fp64 on (bf16 x, fp16-rounded W): within one rounding of the
    2-byte output; the dropout mask equals the split form's; rows past the end untouched; and the WEIGHT error against the fp32
    master is 8x below a single bf16 image's."""
    x = _rand(rows, 256, seed=500 + N).to(torch.bfloat16)
    w = _rand(N, 256, seed=501, scale=1 / 16)
    b = _rand(N, seed=502)
    odt = torch.float16 if f16out else torch.bfloat16
    outs = {}
    for mode in (True, False):
        ops._W16 = mode
        try:
            buf = torch.full((rows + 3, N), 7.0, dtype=odt, device="cuda")
            ops.linear_fwd(dev(x), dev(w), dev(b), buf[:rows], split=True, act=act, drop_p=dp, seed=99)
        finally:
            ops._W16 = False
        assert bool((buf[rows:] == 7.0).all())
        outs[mode] = buf[:rows].float().cpu()
    ref = x.double() @ w.to(torch.float16).double().t() + b.double()
    if act:
        ref = ref.clamp_min(0)
    keep = torch.ones_like(ref, dtype=torch.bool)
    if dp > 0:
        keep = (outs[False] != 0) | (ref.float() <= 1e-3)               # the split form's mask (the counter is a function of seed, row, column)
        assert bool(((outs[True] != 0) | (ref.float() <= 1e-3) == keep).all())
        ref = torch.where(outs[False] != 0, ref / (1 - dp), torch.zeros_like(ref))
    ulp = 2.0 ** (-10 if f16out else -7)
    err = (outs[True].double() - ref).abs()
    assert float(err.max()) <= ulp * float(ref.abs().max()) + 1e-5, float(err.max())
    exact = x.double() @ w.double().t() + b.double()                     # weight rounding only (no output rounding): fp16 against bf16 images
    e16 = ((x.double() @ w.to(torch.float16).double().t() + b.double()) - exact).abs().max()
    eb = ((x.double() @ w.to(torch.bfloat16).double().t() + b.double()) - exact).abs().max()
    assert float(e16) < float(eb) / 4


def test_add_cast_equals_cast_and_add(ops):
    """poet_add_cast (the encoder's head: operand copy of the fp32 stream + the first layer's `src + pos` in one pass) == poet_cast and
    poet_add, bit for bit; ragged tail."""
    n = 37 * 2048 + 24 + 5
    a = _rand(n + 3, seed=600, scale=3.0)[:n].contiguous()
    b = _rand(n, seed=601).to(torch.bfloat16)
    ad, bd = dev(a), dev(b)
    a16, s16 = torch.empty(n, dtype=torch.bfloat16, device="cuda"), torch.empty(n, dtype=torch.bfloat16, device="cuda")
    ops.add_cast(ad, bd, s16, a16)
    ra, rs = torch.empty_like(a16), torch.empty_like(s16)
    ops.cast(ad, ra)
    ops.add(ad, bd, rs)
    assert torch.equal(a16, ra) and torch.equal(s16, rs)


def test_zero_fill_sizes_and_alignments(ops):
    """poet_zero: whole 16 KB blocks, ragged tails, bases that are only 4-byte aligned; nothing written outside [base, base + bytes)."""
    for words, off in [(4096 * 3, 0), (4096 * 3 + 17, 0), (5, 0), (4096 * 2 + 1, 1), (4096 + 4095, 3), (70000, 2)]:
        buf = torch.full((words + off + 8,), 7, dtype=torch.int32, device="cuda")
        ops.zero_(buf[off:off + words])
        assert bool((buf[off:off + words] == 0).all())
        assert bool((buf[:off] == 7).all()) and bool((buf[off + words:] == 7).all())


def test_layernorm_split_stream(ops):
    """The encoder's residual stream between its LayerNorms as bf16 head + IEEE fp16 remainder (poet_ln_fwd: dtype_r / dtype_y =
    POET_F16 behind an fp16 branch; the head is the operand copy y_bf16 / res_bf16).  Output: head + remainder reproduces the fp32
    stream to 2^-19 relative, head, query copy, saved sum and statistics are BIT-identical to the fp32-stream launch; input: a split
    res gives the outputs of the fp32 launch on float(head) + float(remainder) bit for bit."""
    rows, d = 4096 + 5, 256
    x16 = _rand(rows, d, seed=170, scale=3.0).to(torch.float16)
    res32 = _rand(rows, d, seed=171, scale=2.0)
    res_hi = res32.to(torch.bfloat16)
    res_lo = (res32 - res_hi.float()).to(torch.float16)
    res_pair = res_hi.float() + res_lo.float()
    assert (res_pair - res32).abs().max().item() <= 2.0 ** -19 * res32.abs().max().item()
    gamma, beta = 1.0 + 0.1 * _rand(d, seed=172), 0.1 * _rand(d, seed=173)
    pos = _rand(rows, d, seed=174).to(torch.bfloat16)

    def run(res, res16, ydt, drop):
        y = torch.empty(rows, d, dtype=ydt, device="cuda")
        z = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
        y16, q16 = torch.empty_like(z), torch.empty_like(z)
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        ops.ln_fwd(dev(x16), dev(res), dev(gamma), dev(beta), y, z, mean, rstd, rows, d, 1e-5, drop, 4321, y16=y16, pos16=dev(pos), q16=q16,
                   res16=None if res16 is None else dev(res16))
        return y, z, y16, q16, mean, rstd

    for drop in (0.0, 0.1):
        base = run(res_pair, None, torch.float32, drop)
        for res, r16, ydt in ((res_lo, res_hi, torch.float16), (res_lo, res_hi, torch.float32), (res_pair, None, torch.float16)):
            got = run(res, r16, ydt, drop)
            for a_, b_ in zip(got[1:], base[1:]):
                assert torch.equal(a_, b_)
            if ydt == torch.float32:
                assert torch.equal(got[0], base[0])
            else:
                back = got[2].float() + got[0].float()
                assert (back - base[0]).abs().max().item() <= 2.0 ** -19 * base[0].abs().max().item()
    with pytest.raises(Exception):                                      # a split input without its head is refused
        run(res_lo, None, torch.float32, 0.0)


@pytest.mark.parametrize("shapes,m,dt", [([(12, 16), (6, 8), (3, 4)], 4, torch.float32), ([(30, 40), (15, 20), (8, 10), (4, 5)], 2, torch.bfloat16),
                                         ([(60, 80), (30, 40), (15, 20), (8, 10)], 1, torch.bfloat16)])
def test_msda_fused_grid_queries_tiled_scatter(ops, shapes, m, dt):
    """Encoder self-attention case (query q == pixel q): the LDS-privatised value-gradient scatter must equal the
    plain scatter, including samples thrown far outside their tile's halo (global-atomic fallback)."""
    n, d, p = 2, 16, 4
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(17)
    value = torch.from_numpy(rng.standard_normal((n, m, S, d)).astype(np.float32)).to(dt)       # head-major
    mlp = m * L * p
    off = rng.standard_normal((n, S, 2 * mlp)) * 2.0
    off[:, ::7] *= 12.0                                                                           # some far-away samples
    oa = torch.from_numpy(np.concatenate([off, rng.standard_normal((n, S, mlp))], -1).astype(np.float32)).to(dt)
    vr = torch.ones(n, L, 2)
    ref = torch.empty(n, S, L, 2, device="cuda")
    ops.enc_ref_points(dev(vr), geom, ref, n)
    gout = torch.from_numpy(rng.standard_normal((n, S, m * d)).astype(np.float32)).to(dt)
    vstr = (m * S * d, d, S * d)
    outs = []
    for grid in (False, True):
        gv = torch.zeros(n, m, S, d, device="cuda")
        goa = torch.empty_like(dev(oa))
        ops.msda_fused_bwd(dev(value), vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, dev(gout), gv, goa, n, m, d, p, S,
                           grid_queries=grid)
        outs.append((gv.cpu(), goa.float().cpu()))
    scale = outs[0][0].abs().max().item()
    # fp32 storage keeps the exact float scatter; bf16 storage uses the int32 fixed-point LDS windows (resolution
    # 2^-18 of the tile's largest |grad_out| per contribution)
    tol = 2e-4 if dt == torch.float32 else 2e-3
    assert (outs[0][0] - outs[1][0]).abs().max().item() < tol * max(1.0, scale)
    # d(offsets | logits): the grid path stages value windows in LDS (bf16) and sums per lane, the plain path gathers from
    # global memory and reduces across lanes: the same fp32 arithmetic in a different order, then one bf16 rounding
    gs = outs[0][1].abs().max().item()
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= (1e-5 if dt == torch.float32 else 1e-2) * gs


def _explicit_from_fused(shapes, value_nsmd, oa, ref_pts, gout, m, p):
    """float64 closed form (oracle/msda_explicit.py) of the FUSED op: softmax over L*P logits, loc = ref + off / (W, H);
    returns out, d(value) (N,S,M,D), d(offsets|logits) laid out like `oa`."""
    n, lq = oa.shape[:2]
    L = len(shapes)
    mlp = m * L * p
    off = oa[..., : 2 * mlp].double().view(n, lq, m, L, p, 2)
    lg = oa[..., 2 * mlp:].double().view(n, lq, m, L * p)
    w = torch.softmax(lg, -1)
    norm = torch.tensor([[wd, ht] for ht, wd in shapes], dtype=torch.float64)
    loc = ref_pts.double()[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    v = value_nsmd.double().numpy()
    wn = w.view(n, lq, m, L, p).numpy()
    out = msda_explicit.msda_forward(v, shapes, loc.numpy(), wn)
    dv, dl, da = msda_explicit.msda_backward(v, shapes, loc.numpy(), wn, gout.double().numpy())
    doff = torch.from_numpy(dl) / norm[None, None, None, :, None, :]
    da = torch.from_numpy(da).view(n, lq, m, L * p)
    dlg = w * (da - (w * da).sum(-1, keepdim=True))
    doa = torch.cat([doff.reshape(n, lq, 2 * mlp), dlg.reshape(n, lq, mlp)], -1)
    # d/d(offset_x) is a ONE-SIDED derivative where px is an integer (bilinear kink): which side a kernel takes there depends
    # on the last bit of px.  Mark the entries within 1e-3 px of a kink so a comparison can leave them out.
    pix = loc * norm[None, None, None, :, None, :] - 0.5
    kink6 = (pix - pix.round()).abs() < 1e-3
    kink = kink6.reshape(n, lq, 2 * mlp)

    def one_sided(sign):
        """d(offsets) with every kinked coordinate moved 2e-3 px to one side (the derivative along a coordinate is constant inside a
        pixel cell, and the other coordinate's derivative changes by 2e-3 relative at most): the LEFT / RIGHT derivatives."""
        loc_s = (pix + sign * 2e-3 * kink6.double() + 0.5) / norm[None, None, None, :, None, :]
        dl_s = msda_explicit.msda_backward(v, shapes, loc_s.numpy(), wn, gout.double().numpy())[1]
        return (torch.from_numpy(dl_s) / norm[None, None, None, :, None, :]).reshape(n, lq, 2 * mlp)
    kink.one_sided = one_sided                 # (evaluated only by the comparisons that look at the kinked entries)
    return torch.from_numpy(out), torch.from_numpy(dv), doa, kink


def _kink_error(dq_off, kink, scale):
    """Entries on a bilinear kink must equal ONE of the two one-sided derivatives (which one is decided by the last bit of px in the
    kernel's own arithmetic, per query): worst distance to the nearer side, relative to `scale`."""
    if not bool(kink.any()):
        return 0.0
    left, right = kink.one_sided(-1.0), kink.one_sided(+1.0)
    e = torch.minimum((dq_off - left).abs(), (dq_off - right).abs())
    return float((e * kink).max()) / scale


@pytest.mark.parametrize("case", ["ycbv_init_like", "ycbv_wide_offsets", "hires_tiles", "one_pixel_pileup", "odd_small", "lmo_whole_image_windows",
                                  "reference_init_exact"])
def test_msda_encoder_kernels_vs_explicit_full_geometry(ops, case, monkeypatch, tmp_path):
    """The kernels the benchmark runs -- fused forward, d(offsets|logits), and the LDS-tiled int32 fixed-point d(value)
    scatter -- at the benchmark's geometry (M = 16 heads, D = 16, bf16 storage, grid queries, bs 2) against the float64
    closed form, not against each other:
      ycbv_init_like    (60,80)..(8,10): offsets = the reference's initial directional grid (+-1..4 px) + noise
      ycbv_wide_offsets the same with 10 % of the samples thrown 10-40 px away (out of window -> global-atomic pass, out of image)
      hires_tiles       (120,160)..(15,20) at bs 1: the 16x16-tile plan with halo-dominated windows (BASELINE configs[4])
      one_pixel_pileup  every sample of every query aims at ONE pixel per level with same-sign gradients: the worst case for
                        the int32 windows (n_queries_in_tile x 2^18 per word; the tile plan keeps that below 2^31).
      odd_small         (7,9)..(1,2) at bs 3: 273 query rows (not a multiple of the 4 rows of a shared-geometry workgroup), odd
                        map sizes, levels of one and two pixels where most samples leave the map
      lmo_whole_image_windows  (30,40)..(4,5): the tile planner's 1 x 1 plan (every level whole in LDS, no halo, no far pass)
      reference_init_exact     offsets = the reference's initial grid EXACTLY (MSDeformAttn._reset_parameters: bias only): the axis /
                        diagonal heads' components are integers and grid queries sit on pixel centres, so those entries of
                        d(offsets) are one-sided derivatives -- each must equal the left or the right one (_kink_error)"""
    m, d, p = 16, 16, 4
    if case == "hires_tiles":
        shapes, n = [(120, 160), (60, 80), (30, 40), (15, 20)], 1
    elif case == "odd_small":
        shapes, n = [(7, 9), (4, 5), (2, 3), (1, 2)], 3
    elif case == "lmo_whole_image_windows":
        shapes, n = [(30, 40), (15, 20), (8, 10), (4, 5)], 2
    else:
        shapes, n = [(60, 80), (30, 40), (15, 20), (8, 10)], 2
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(23)
    mlp = m * L * p
    value = torch.from_numpy(rng.standard_normal((n, S, m, d)).astype(np.float32)).to(torch.bfloat16)
    th = np.arange(m) * (2 * np.pi / m)
    grid = np.stack([np.cos(th), np.sin(th)], -1)
    grid = grid / np.abs(grid).max(-1, keepdims=True)
    base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1)          # (m, L, p, 2)
    off = base[None, None] + (0.0 if case == "reference_init_exact" else 0.3) * rng.standard_normal((n, S, m, L, p, 2))
    lg = rng.standard_normal((n, S, mlp))
    gout = rng.standard_normal((n, S, m * d))
    if case == "ycbv_wide_offsets":
        far = rng.random((n, S, m, L, p, 1)) < 0.1
        off = np.where(far, off * rng.uniform(5, 20, off.shape), off)
    vr = torch.ones(n, L, 2)
    ref = torch.empty(n, S, L, 2, device="cuda")
    ops.enc_ref_points(dev(vr), geom, ref, n)
    if case == "one_pixel_pileup":
        # offset = (target pixel centre) - (query's reference point in pixels): every corner weight lands on one pixel
        refc = ref.cpu().numpy()                                                               # (n,S,L,2) normalised
        tgt = np.array([[(wd // 2 + 0.5), (ht // 2 + 0.5)] for ht, wd in shapes])              # pixel centres (x, y)
        wh = np.array([[wd, ht] for ht, wd in shapes], np.float64)
        off = (tgt[None, None] - refc * wh[None, None])[:, :, None, :, None, :].repeat(m, 2).repeat(p, 4)
        gout = np.abs(gout) + 1.0                                                              # same sign: magnitudes add up
    oa = torch.from_numpy(np.concatenate([off.reshape(n, S, 2 * mlp), lg], -1).astype(np.float32)).to(torch.bfloat16)
    gout = torch.from_numpy(gout.astype(np.float32)).to(torch.bfloat16)
    out_ref, dv_ref, doa_ref, kink = _explicit_from_fused(shapes, value.float(), oa.float(), ref.cpu(), gout.float(), m, p)

    vdev = dev(value.permute(0, 2, 1, 3).contiguous())                                        # head-major (N,M,S,D)
    vstr = (m * S * d, d, S * d)
    out = torch.empty(n, S, m * d, dtype=torch.bfloat16, device="cuda")
    ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True)
    e_out = (out.double().cpu() - out_ref).abs().max().item() / out_ref.abs().max().item()
    if PROBES_BUILT:          # measured losers, compiled only into probe builds of the library (POET_BUILD_PROBES=1, profiles/probes/kernels/)
        # the opt-in hybrid forward (coarse levels staged in LDS per (image, head quarter), fine levels through L1): the same arithmetic
        # in the same order -- bit-identical to the default kernel
        monkeypatch.setenv("POET_MSDA_HYBRID", "1")
        out_h = torch.empty_like(out)
        ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, out_h, n, m, d, p, S, grid_queries=True)
        monkeypatch.delenv("POET_MSDA_HYBRID")
        assert torch.equal(out_h, out)
        # the opt-in variants that stage bf16 value windows in LDS (one lane per (query, head)), same problem
        monkeypatch.setenv("POET_WIN_GATHER", "1")
        out_w, goa_w = torch.empty_like(out), torch.empty_like(dev(oa))
        ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, out_w, n, m, d, p, S, grid_queries=True)
        ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, dev(gout), torch.zeros(n, m, S, d, device="cuda"), goa_w,
                           n, m, d, p, S, grid_queries=True, parts=1)
        monkeypatch.delenv("POET_WIN_GATHER")
        assert (out_w.double().cpu() - out_ref).abs().max().item() / out_ref.abs().max().item() < 6e-3
        dqw = goa_w.double().cpu()
        assert ((dqw[..., : 2 * mlp] - doa_ref[..., : 2 * mlp]).abs() * (~kink)).max().item() / doa_ref[..., : 2 * mlp].abs().max().item() < 8e-3
        assert (dqw[..., 2 * mlp:] - doa_ref[..., 2 * mlp:]).abs().max().item() / doa_ref[..., 2 * mlp:].abs().max().item() < 8e-3
    # the general gather kernels (every lane of a (query, head) computes the sample geometry itself), same problem: the default
    # at this shape are the shared-geometry kernels (one lane per level prepares the points, LDS records).  The library reads its
    # kernel-selection switches once per process, so the variant runs in a process of its own (tests/msda_general_worker.py)
    import subprocess, sys as _sys
    here = os.path.dirname(os.path.abspath(__file__))
    fin, fout = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(fin, shapes=np.array(shapes), nmdp=np.array([n, m, d, p]), value_hm=vdev.float().cpu().numpy(), oa=oa.float().numpy(),
             gout=gout.float().numpy(), ref=ref.cpu().numpy())
    r_ = subprocess.run([_sys.executable, os.path.join(here, "msda_general_worker.py"), fin, fout], capture_output=True, text=True, timeout=600)
    assert r_.returncode == 0, r_.stderr[-3000:]
    zg = np.load(fout)
    out_g, goa_g = torch.from_numpy(zg["out"]), torch.from_numpy(zg["goa"])
    assert (out_g.double().cpu() - out_ref).abs().max().item() / out_ref.abs().max().item() < 6e-3
    dqg = goa_g.double().cpu()
    assert ((dqg[..., : 2 * mlp] - doa_ref[..., : 2 * mlp]).abs() * (~kink)).max().item() / doa_ref[..., : 2 * mlp].abs().max().item() < 8e-3
    assert (dqg[..., 2 * mlp:] - doa_ref[..., 2 * mlp:]).abs().max().item() / doa_ref[..., 2 * mlp:].abs().max().item() < 8e-3
    gv = torch.zeros(n, m, S, d, device="cuda")
    goa = torch.empty_like(dev(oa))
    ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, dev(gout), gv, goa, n, m, d, p, S, grid_queries=True)
    dv = gv.double().cpu().permute(0, 2, 1, 3)
    e_dv = (dv - dv_ref).abs().max().item() / dv_ref.abs().max().item()
    dq = goa.double().cpu()
    e_off = ((dq[..., : 2 * mlp] - doa_ref[..., : 2 * mlp]).abs() * (~kink)).max().item() / doa_ref[..., : 2 * mlp].abs().max().item()
    e_lg = (dq[..., 2 * mlp:] - doa_ref[..., 2 * mlp:]).abs().max().item() / doa_ref[..., 2 * mlp:].abs().max().item()
    e_kink = _kink_error(dq[..., : 2 * mlp], kink, doa_ref[..., : 2 * mlp].abs().max().item())
    print(f"{case}: rel max err out {e_out:.2e} dV {e_dv:.2e} d(off) {e_off:.2e} ({int(kink.sum())} of {kink.numel()} entries on a bilinear kink: "
          f"{e_kink:.2e} from the nearer one-sided derivative) d(logit) {e_lg:.2e}; max|dV| {dv_ref.abs().max():.3g}")
    assert e_kink < 1e-2, e_kink            # (8e-3 as everywhere else + the 2e-3 the shifted evaluation moves the OTHER coordinate's derivative by)
    # inputs are bf16-exact, accumulation fp32 / int32 fixed point (2^-18 of the tile's max |grad_out| per contribution):
    # only the bf16 rounding of the stored outputs (out, d(off|logit): 2^-9 relative) and the fixed point remain
    assert e_out < 6e-3 and e_off < 8e-3 and e_lg < 8e-3, (e_out, e_off, e_lg)
    assert e_dv < 1e-3, e_dv
    # the same scatter into a bf16 map (what the encoder's backward uses: packed bf16x2 atomics, the consumer rounds to bf16
    # anyway): exact windows, one bf16 rounding per (tile, pixel, channel), at most 4 tiles per pixel
    gv16 = torch.zeros(n, m, S, d, dtype=torch.bfloat16, device="cuda")
    ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, dev(gout), gv16, torch.empty_like(dev(oa)), n, m, d, p, S,
                       grid_queries=True, parts=2)
    dv16 = gv16.double().cpu().permute(0, 2, 1, 3)
    e_dv16 = (dv16 - dv_ref).abs().max().item() / dv_ref.abs().max().item()
    print(f"   bf16 value-gradient map: rel max err {e_dv16:.2e}")
    assert e_dv16 < 2e-2, e_dv16            # a pixel on a tile border: <= 4 partial sums, each rounded to bf16 (<= 2^-8 relative)
    rows16 = torch.empty(n * S, m * d, dtype=torch.bfloat16, device="cuda")
    ops.vgrad_to_rows(gv16, vstr, None, rows16, n, S, m, d)
    assert torch.equal(rows16.view(n, S, m, d).cpu(), gv16.permute(0, 2, 1, 3).cpu())


@pytest.mark.parametrize("N,act,dp,odt", [(1024, 1, 0.1, torch.bfloat16), (768, 0, 0.0, torch.float16), (512, 0, 0.0, torch.bfloat16)])
@pytest.mark.parametrize("rows", [102080, 5007, 4096 + 16])
@pytest.mark.skipif(not PROBES_BUILT, reason="gemm_wr.hip lives under profiles/probes/kernels/: probe builds only (POET_BUILD_PROBES=1)")
def test_gemm_register_stationary_equals_lds_stationary(ops, monkeypatch, N, act, dp, odt, rows):
    """gemm_wr.hip (opt-in POET_GEMM_WR=1: the split weights resident in registers, the activation through an LDS-DMA ring) computes
    the wide split-weight forward products in the accumulation order of gemm_ws.hip: the outputs -- ReLU, the dropout mask, bf16 and
    fp16 storage, ragged row counts and odd unit counts -- must be bit-identical."""
    x = _rand(rows, 256, seed=300 + N).to(torch.bfloat16)
    w = _rand(N, 256, seed=301, scale=1 / 16)
    b = _rand(N, seed=302)
    outs = []
    for wr in ("0", "1"):
        monkeypatch.setenv("POET_GEMM_WR", wr)
        out = torch.full((rows + 3, N), 7.0, dtype=odt, device="cuda")          # (3 guard rows: nothing may be written past M)
        ops.linear_fwd(dev(x), dev(w), dev(b), out[:rows], split=True, act=act, drop_p=dp, seed=99)
        outs.append(out)
    monkeypatch.delenv("POET_GEMM_WR")
    assert torch.equal(outs[0], outs[1])
    assert bool((outs[1][rows:] == 7.0).all())
    ref = x.float() @ w.t() + b
    if act:
        ref = torch.relu(ref)
    got = outs[1][:rows].float().cpu()
    keep = got != 0 if dp > 0 else torch.ones_like(got, dtype=torch.bool)
    assert ((got * (1 - dp) - ref).abs() * keep).max().item() <= 6e-3 * ref.abs().max().item()


@pytest.mark.parametrize("N,act,dp,f16", [(1024, 1, 0.1, 0), (768, 0, 0.0, 1), (512, 1, 0.0, 0)])
@pytest.mark.parametrize("rows", [102080, 4096 + 23])
def test_gemm_split_pipelined_vs_plain_streaming(ops, tmp_path, N, act, dp, f16, rows):
    """gemm_wsp_kernel (the default for the wide split-weight forward products: matrix instructions of one 64-column half interleaved
    with the epilogue of the other, the dropout scale folded into the staged weights) against gemm_ws_kernel (POET_WS_PIPE=0, own
    process): BIT-IDENTICAL without dropout (so the goldens' realisation of the bf16 policy is the same under either kernel); with
    dropout the SAME mask and values within one rounding of the 2-byte output; whole and ragged row counts, nothing written past the
    last row; and against torch."""
    import subprocess, sys as _sys
    x = _rand(rows, 256, seed=400 + N).to(torch.bfloat16)
    w = _rand(N, 256, seed=401, scale=1 / 16)
    b = _rand(N, seed=402)
    odt = torch.float16 if f16 else torch.bfloat16
    out = torch.full((rows + 3, N), 7.0, dtype=odt, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out[:rows], split=True, act=act, drop_p=dp, seed=99)
    assert bool((out[rows:] == 7.0).all())
    got = out[:rows].float().cpu()
    here = os.path.dirname(os.path.abspath(__file__))
    fin, fout = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(fin, x=x.float().numpy(), w=w.numpy(), b=b.numpy(), act=act, dp=dp, seed=99, f16=f16)
    r_ = subprocess.run([_sys.executable, os.path.join(here, "ws_pipe_worker.py"), fin, fout], capture_output=True, text=True, timeout=600)
    assert r_.returncode == 0, r_.stderr[-3000:]
    plain = torch.from_numpy(np.load(fout)["out"])
    scale = plain.abs().max().item()
    ulp = 2.0 ** (-10 if f16 else -7)
    if dp == 0:
        assert torch.equal(got, plain)                                   # same accumulation order, bias last: bit-identical
    assert (got - plain).abs().max().item() <= ulp * scale                       # (dropout: the scale sits in the staged weights) one rounding step
    if dp > 0:                                                           # the mask is a function of (seed, row, column) only
        tiny = plain.abs() < 1e-3 * scale                                # (an output the ReLU may or may not have clipped, by one rounding)
        assert bool((((got == 0) == (plain == 0)) | tiny).all())
        assert abs((got == 0).float().mean().item() - (plain == 0).float().mean().item()) < 1e-5
    ref = x.float() @ w.t() + b
    if act:
        ref = torch.relu(ref)
    keep = got != 0 if dp > 0 else torch.ones_like(got, dtype=torch.bool)
    assert ((got * (1 - dp) - ref).abs() * keep).max().item() <= (1.5e-3 if f16 else 6e-3) * ref.abs().max().item()


@pytest.mark.parametrize("rows", [5000 + 7, 700])
def test_gemm_fp16_output_storage(ops, rows):
    """PoetGemmDesc.c_f16: the 2-byte outputs of a bf16-typed C written as IEEE fp16 (the offsets | logits buffer of the encoder's
    MSDeformAttn: values of a few units that only the MSDA kernels read) -- streaming kernel (>= 4096 rows, split weights) and
    tiled kernel (< 4096 rows), row-major and head-major stores; the result is the fp16 rounding of the bf16-operand product."""
    K, N = 256, 768
    x = _rand(rows, K, seed=190).to(torch.bfloat16)
    w = _rand(N, K, seed=191, scale=1 / math.sqrt(K))
    b = _rand(N, seed=192)
    hi = w.to(torch.bfloat16).float()
    w2 = hi + (w - hi).to(torch.bfloat16).float()                       # split weights: hi + lo
    ref = (x.float().double() @ w2.double().t() + b.double())
    out = torch.empty(rows, N, dtype=torch.float16, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out, split=True)
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 2.0 ** -11 * ref.abs().max().item() + 2e-4, err                 # half an fp16 ulp of the largest value + the fp32 accumulation
    assert (out.cpu() == ref.to(torch.float16)).float().mean().item() > 0.98       # almost everywhere THE fp16 rounding of the exact product
    out_bf = torch.empty(rows, N, dtype=torch.bfloat16, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out_bf, split=True)
    assert (out_bf.double().cpu() - ref).abs().max().item() > 4 * err              # the same call in bf16 storage is >= 4x coarser
    with pytest.raises(TypeError):
        ops.linear_fwd(dev(x).to(torch.float16), dev(w), dev(b), out_bf)


def test_msda_fp16_offsets_logits_vs_explicit(ops):
    """The encoder's MSDA kernels reading the offsets | logits buffer as fp16 (q_dtype POET_F16: shared-geometry forward and
    d(offsets | logits), LDS-tiled d(value) scatter; gradients stay bf16) against the float64 closed form evaluated on the SAME
    fp16 values -- and the accuracy this buys: the sampled output of bf16-stored offsets is several times further from the
    output of the unrounded offsets than that of fp16-stored ones."""
    m, d, p = 16, 16, 4
    shapes, n = [(60, 80), (30, 40), (15, 20), (8, 10)], 2
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(29)
    mlp = m * L * p
    value = torch.from_numpy(rng.standard_normal((n, S, m, d)).astype(np.float32)).to(torch.bfloat16)
    th = np.arange(m) * (2 * np.pi / m)
    grid = np.stack([np.cos(th), np.sin(th)], -1)
    grid = grid / np.abs(grid).max(-1, keepdims=True)
    base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1)
    off = base[None, None] + 0.3 * rng.standard_normal((n, S, m, L, p, 2))
    lg = rng.standard_normal((n, S, mlp))
    oa32 = torch.from_numpy(np.concatenate([off.reshape(n, S, 2 * mlp), lg], -1).astype(np.float32))
    gout = torch.from_numpy(rng.standard_normal((n, S, m * d)).astype(np.float32)).to(torch.bfloat16)
    ref = torch.empty(n, S, L, 2, device="cuda")
    ops.enc_ref_points(dev(torch.ones(n, L, 2)), geom, ref, n)
    vdev = dev(value.permute(0, 2, 1, 3).contiguous())
    vstr = (m * S * d, d, S * d)
    res = {}
    for qdt in (torch.float16, torch.bfloat16):
        oa = oa32.to(qdt)
        out_ref, dv_ref, doa_ref, kink = _explicit_from_fused(shapes, value.float(), oa.float(), ref.cpu(), gout.float(), m, p)
        out = torch.empty(n, S, m * d, dtype=torch.bfloat16, device="cuda")
        ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True)
        gv = torch.zeros(n, m, S, d, device="cuda")
        goa = torch.empty(n, S, 3 * mlp, dtype=torch.bfloat16, device="cuda")
        ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, dev(gout), gv, goa, n, m, d, p, S, grid_queries=True)
        e_out = (out.double().cpu() - out_ref).abs().max().item() / out_ref.abs().max().item()
        e_dv = (gv.double().cpu().permute(0, 2, 1, 3) - dv_ref).abs().max().item() / dv_ref.abs().max().item()
        dq = goa.double().cpu()
        e_off = ((dq[..., : 2 * mlp] - doa_ref[..., : 2 * mlp]).abs() * (~kink)).max().item() / doa_ref[..., : 2 * mlp].abs().max().item()
        e_lg = (dq[..., 2 * mlp:] - doa_ref[..., 2 * mlp:]).abs().max().item() / doa_ref[..., 2 * mlp:].abs().max().item()
        print(f"offsets | logits stored as {qdt}: rel max err out {e_out:.2e} dV {e_dv:.2e} d(off) {e_off:.2e} d(logit) {e_lg:.2e}")
        assert e_out < 6e-3 and e_dv < 1e-3 and e_off < 8e-3 and e_lg < 8e-3, (qdt, e_out, e_dv, e_off, e_lg)
        res[qdt] = out_ref
    exact, _, _, _ = _explicit_from_fused(shapes, value.float(), oa32, ref.cpu(), gout.float(), m, p)
    d16 = (res[torch.float16] - exact).pow(2).mean().sqrt().item()
    dbf = (res[torch.bfloat16] - exact).pow(2).mean().sqrt().item()
    print(f"   rms distance of the sampled output from the one of the unrounded offsets | logits: fp16 storage {d16:.2e}, bf16 storage {dbf:.2e}")
    assert d16 < 0.25 * dbf


@pytest.mark.parametrize("gscale", ["unit", "tiny_mixed"])
def test_msda_fp16_value_maps_vs_explicit(ops, gscale):
    """Round 6: the encoder's value maps stored as IEEE fp16 (v_dtype POET_F16 with q_dtype POET_F16: shared-geometry forward by
    v_fma_mix_f32 on the packed halves, d(offsets | logits) by v_dot2c_f32_f16 with grad_out re-typed bf16 -> fp16 under a per-
    (query, head) power-of-two scale) against the float64 closed form on the SAME fp16 values.  tiny_mixed: grad_out spans 1e-9 ..
    1e-2 across heads with one head all zero and single huge channels -- the rescaling must neither flush the small heads nor
    overflow; what it may drop is below 2^-28 of a head's largest channel."""
    m, d, p = 16, 16, 4
    shapes, n = [(60, 80), (30, 40), (15, 20), (8, 10)], 2
    L = len(shapes)
    geom = ops.LevelGeom(shapes)
    S = geom.S
    rng = np.random.default_rng(31)
    mlp = m * L * p
    v32 = torch.from_numpy((rng.standard_normal((n, S, m, d)) * np.exp(rng.uniform(-3, 2, (n, S, m, 1)))).astype(np.float32))
    th = np.arange(m) * (2 * np.pi / m)
    grid = np.stack([np.cos(th), np.sin(th)], -1)
    grid = grid / np.abs(grid).max(-1, keepdims=True)
    base = (grid[:, None, None, :] * (np.arange(p) + 1)[None, None, :, None]).repeat(L, 1)
    off = base[None, None] + 0.3 * rng.standard_normal((n, S, m, L, p, 2))
    lg = rng.standard_normal((n, S, mlp))
    oa = torch.from_numpy(np.concatenate([off.reshape(n, S, 2 * mlp), lg], -1).astype(np.float32)).to(torch.float16)
    g = rng.standard_normal((n, S, m, d))
    if gscale == "tiny_mixed":
        g = g * (10.0 ** rng.uniform(-9, -2, (n, S, m, 1)))
        g[:, :, 3] = 0.0                                         # a head without gradient
        g[:, ::7, 5, 2] *= 3.0e4                                 # one channel 2^15 above the rest of its head
        g[:, ::5, 9, :8] *= 1.0e-6                               # a channel HALF far below its partner half (the scale is per head, not per lane)
    gout = torch.from_numpy(g.reshape(n, S, m * d).astype(np.float32)).to(torch.bfloat16)
    ref = torch.empty(n, S, L, 2, device="cuda")
    ops.enc_ref_points(dev(torch.ones(n, L, 2)), geom, ref, n)
    vstr = (m * S * d, d, S * d)
    errs = {}
    for vdt in (torch.float16, torch.bfloat16):
        value = v32.to(vdt)
        out_ref, dv_ref, doa_ref, kink = _explicit_from_fused(shapes, value.float(), oa.float(), ref.cpu(), gout.float(), m, p)
        vdev = dev(value.permute(0, 2, 1, 3).contiguous())
        out = torch.empty(n, S, m * d, dtype=torch.bfloat16, device="cuda")
        ops.msda_fused_fwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True)
        gv = torch.zeros(n, m, S, d, device="cuda")
        goa = torch.empty(n, S, 3 * mlp, dtype=torch.bfloat16, device="cuda")
        ops.msda_fused_bwd(vdev, vstr, geom, dev(oa), 3 * mlp, 2 * mlp, ref, S * L * 2, dev(gout), gv, goa, n, m, d, p, S, grid_queries=True)
        e_out = (out.double().cpu() - out_ref).abs().max().item() / out_ref.abs().max().item()
        dq = goa.double().cpu()
        ghead = gout.double().abs().reshape(n, S, m, d).amax(-1, keepdim=True)
        # per-(query, head) scale of the reference gradient: the heads span 7 decades, a global max would hide the small ones
        def rel(a, b, width, mask=None):
            a, b = a.reshape(n, S, m, width), b.reshape(n, S, m, width)
            e = (a - b).abs()
            if mask is not None:
                e = e * (~mask.reshape(n, S, m, width))
            # (floor: 1e-3 of what the head's grad_out and the map's values can produce -- a head whose only in-range corner has a
            # weight of 1e-8 carries a gradient of 1e-8 of the typical one, and fp32 sample positions round that corner away)
            sc = torch.maximum(b.abs().amax(-1, keepdim=True), 1e-3 * ghead * float(value.float().abs().max()))
            ok = ghead > 0
            assert bool((a[~ok.expand_as(e)] == 0).all())       # a head without gradient gets exactly zero
            r = torch.where(ok.expand_as(e), e / sc.clamp_min(1e-300), torch.zeros_like(e))
            worst = int(r.argmax())
            i0, i1, i2 = np.unravel_index(worst // width, (n, S, m))
            rel.info = f"worst head (n {i0}, q {i1}, m {i2}): kernel {a[i0, i1, i2].tolist()} reference {b[i0, i1, i2].tolist()}"
            return float(r.max())
        e_off = rel(dq[..., : 2 * mlp], doa_ref[..., : 2 * mlp], 2 * L * p, kink)
        e_lg = rel(dq[..., 2 * mlp:], doa_ref[..., 2 * mlp:], L * p)
        info_lg = rel.info
        e_dv = (gv.double().cpu().permute(0, 2, 1, 3) - dv_ref).abs().max().item() / dv_ref.abs().max().item()
        print(f"value maps stored as {vdt}, grad_out {gscale}: rel max err out {e_out:.2e} d(off) {e_off:.2e} d(logit) {e_lg:.2e} (per head) dV {e_dv:.2e}")
        assert bool(torch.isfinite(goa.float()).all()) and bool(torch.isfinite(out.float()).all())
        errs[vdt] = (e_off, e_lg, e_out, info_lg)
    for vdt, (e_off, e_lg, e_out, info_lg) in errs.items():
        assert e_out < 6e-3 and e_off < 1.2e-2 and e_lg < 1.2e-2, (vdt, e_out, e_off, e_lg, info_lg)      # (bf16 rounding of the stored results, relative to the HEAD's max)
    # the fp16 path is no worse than the bf16 one on the same problem (both are bound by the bf16 rounding of the stored gradient)
    assert errs[torch.float16][0] < 1.5 * errs[torch.bfloat16][0] + 1e-3 and errs[torch.float16][1] < 1.5 * errs[torch.bfloat16][1] + 1e-3, errs
    # ... and rounding fp32 maps to fp16 loses 8x less than rounding them to bf16
    exact, _, _, _ = _explicit_from_fused(shapes, v32, oa.float(), ref.cpu(), gout.float(), m, p)
    o16, _, _, _ = _explicit_from_fused(shapes, v32.to(torch.float16).float(), oa.float(), ref.cpu(), gout.float(), m, p)
    obf, _, _, _ = _explicit_from_fused(shapes, v32.to(torch.bfloat16).float(), oa.float(), ref.cpu(), gout.float(), m, p)
    d16, dbf = (o16 - exact).pow(2).mean().sqrt().item(), (obf - exact).pow(2).mean().sqrt().item()
    print(f"   rms distance of the sampled output from the one of fp32 maps: fp16 storage {d16:.2e}, bf16 storage {dbf:.2e}")
    assert d16 < 0.2 * dbf
    with pytest.raises(Exception):                               # fp16 maps come with fp16 offsets | logits: anything else is refused, not misread
        v16 = dev(v32.to(torch.float16).permute(0, 2, 1, 3).contiguous())
        ops.msda_fused_fwd(v16, vstr, geom, dev(oa.float().to(torch.bfloat16)), 3 * mlp, 2 * mlp, ref, S * L * 2, out, n, m, d, p, S, grid_queries=True)


# ------------------------------------------------------------------------- streaming (weight-stationary) GEMM
@pytest.mark.parametrize("cdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("wdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [256, 768, 80])
def test_gemm_streaming_fwd(ops, cdtype, wdtype, N):
    """M >= 4096, K = 256, bf16 activations: poet_gemm routes to gemm_ws.hip.  Full epilogue against torch, and the
    dropout pattern against the generic tiled kernel (same call on a < 4096-row prefix: masks are index-based)."""
    M, K = 5000 + 7, 256
    x = _rand(M, K, seed=90).to(torch.bfloat16)
    w = _rand(N, K, seed=91, scale=1 / math.sqrt(K)).to(wdtype)
    b = _rand(N, seed=92)
    add = _rand(M, N, seed=93).to(cdtype)
    mask = (torch.arange(M) % 11 == 3).to(torch.uint8)
    out = torch.empty(M, N, dtype=cdtype, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out, act=1, add_src=dev(add), row_mask=dev(mask))
    ref = (torch.relu(x.float() @ w.to(torch.bfloat16).float().t() + b) + add.float()).masked_fill(mask.bool()[:, None], 0)
    _close(out, ref, torch.bfloat16, msg="streaming fwd")
    # dropout: same (seed, row*N+col) stream as the generic kernel
    o_ws = torch.empty(M, N, dtype=cdtype, device="cuda")
    o_gen = torch.empty(1000, N, dtype=cdtype, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), o_ws, drop_p=0.25, seed=77)
    ops.linear_fwd(dev(x[:1000].contiguous()), dev(w), dev(b), o_gen, drop_p=0.25, seed=77)
    a, g = o_ws[:1000].float().cpu(), o_gen.float().cpu()
    assert torch.equal(a == 0, g == 0)
    assert (a - g).abs().max().item() <= 2e-2 * g.abs().max().item()
    frac = (o_ws.float() == 0).float().mean().item()
    assert abs(frac - 0.25) < 0.01


@pytest.mark.parametrize("M,N,K", [(5007, 256, 256), (5007, 768, 256), (5007, 1024, 256), (4500, 256, 1024), (900, 256, 256), (640, 256, 2304)])
@pytest.mark.parametrize("narrow", ["1", "2"])
def test_gemm_split_weight(ops, M, N, K, narrow, monkeypatch):
    """PoetGemmDesc.b_split: the fp32 weight enters as bf16 hi + bf16 lo.  With the activation already bf16-exact the product
    must match the fp32-weight reference to fp32-accumulation accuracy (1e-5 relative), where the single-bf16-weight GEMM is
    off by 2e-3 -- in the streaming kernel (K = 256, >= 4096 rows; both narrow-output variants) and in the tiled one."""
    if narrow == "2" and not (N <= 256 and K == 256 and M >= 4096):
        pytest.skip("variant 2 only differs for narrow streaming shapes")
    x = _rand(M, K, seed=190).to(torch.bfloat16)
    w = _rand(N, K, seed=191, scale=1 / math.sqrt(K))
    b = _rand(N, seed=192)
    add = _rand(M, N, seed=193).to(torch.bfloat16)
    mask = (torch.arange(M) % 13 == 5).to(torch.uint8)
    ref = (x.double() @ w.double().t() + b.double())
    for cdtype in (torch.float32, torch.bfloat16):
        out = torch.empty(M, N, dtype=cdtype, device="cuda")
        import subprocess, sys, os
        if narrow == "2":       # the variant is latched at first use inside the library: run it in a fresh process
            code = ("import torch, math, sys; sys.path.insert(0, %r); from poet_amd import ops; from tests.test_kernels_gpu import _rand;"
                    "x=_rand(%d,%d,seed=190).to(torch.bfloat16).cuda(); w=_rand(%d,%d,seed=191,scale=1/math.sqrt(%d)).cuda(); b=_rand(%d,seed=192).cuda();"
                    "o=torch.empty(%d,%d,dtype=torch.float32,device='cuda'); ops.linear_fwd(x,w,b,o,split=True);"
                    "r=x.double()@w.double().t()+b.double(); e=(o.double()-r).abs().max().item()/r.abs().max().item(); print('ERR',e); assert e<2e-5"
                    % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), M, K, N, K, K, N, M, N))
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, POET_WS_SPLIT_NARROW="2"), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            return
        ops.linear_fwd(dev(x), dev(w), dev(b), out, split=True)
        err = (out.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < (2e-5 if cdtype == torch.float32 else 5e-3), (cdtype, err)
        if cdtype == torch.float32:
            single = torch.empty(M, N, dtype=cdtype, device="cuda")
            ops.linear_fwd(dev(x), dev(w.to(torch.bfloat16)), dev(b), single)
            err1 = (single.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
            assert err1 > 20 * err, (err, err1)               # the split really buys the weight's low bits
    # epilogue forms of the path: residual add / row mask (bf16 output)
    if K == 256 and M >= 4096:
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.linear_fwd(dev(x), dev(w), dev(b), out, split=True, add_src=dev(add))
        _close(out, (ref + add.double()).float(), torch.bfloat16, msg="split + add")
        ops.linear_fwd(dev(x), dev(w), dev(b), out, split=True, row_mask=dev(mask))
        _close(out, ref.float().masked_fill(mask.bool()[:, None], 0), torch.bfloat16, msg="split + mask")


@pytest.mark.parametrize("cdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("K", [512, 768, 1024])
def test_gemm_streaming_kchunked(ops, cdtype, K):
    """K = 512 / 768 / 1024 at >= 4096 rows: the K-chunked weight-stationary kernel (gemm_wsk_kernel) -- forward with single
    and split weights, and the input-gradient forms (weight stored [K][N]) with ReLU gate / accumulate -- against torch, on a
    ragged row count (partial last 256-row block)."""
    M, N = 4096 + 333, 256
    x = _rand(M, K, seed=290).to(torch.bfloat16)
    w = _rand(N, K, seed=291, scale=1 / math.sqrt(K))
    b = _rand(N, seed=292)
    ref = x.double() @ w.double().t() + b.double()
    out = torch.empty(M, N, dtype=cdtype, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out, split=True)
    err = (out.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (2e-5 if cdtype == torch.float32 else 5e-3), err
    ops.linear_fwd(dev(x), dev(w.to(torch.bfloat16)), dev(b), out)
    ref1 = x.double() @ w.to(torch.bfloat16).double().t() + b.double()
    err = (out.double().cpu() - ref1).abs().max().item() / ref1.abs().max().item()
    assert err < (2e-5 if cdtype == torch.float32 else 5e-3), err
    # dX: dy (M, n_out = K here) @ W (n_out, k_in = 256)
    dy = _rand(M, K, seed=293).to(torch.bfloat16)
    wk = _rand(K, 256, seed=294, scale=1 / math.sqrt(K)).to(torch.bfloat16)
    gate = _rand(M, 256, seed=295).to(cdtype)
    addend = _rand(M, 256, seed=296).to(cdtype)
    base = dy.double() @ wk.double()
    o = torch.empty(M, 256, dtype=cdtype, device="cuda")
    ops.linear_dx(dev(dy), dev(wk), o, rows=M)
    _close(o, base.float(), cdtype if cdtype == torch.bfloat16 else torch.float32, msg=f"wsk dX K={K}")
    o = dev(addend.clone())
    ops.linear_dx(dev(dy), dev(wk), o, rows=M, add_src=o)
    _close(o, (base + addend.double()).float(), torch.bfloat16, msg=f"wsk dX+add K={K}")
    ops.linear_dx(dev(dy), dev(wk), o, rows=M, gate_ref=dev(gate), gate_scale=1.25)
    _close(o, (base * 1.25 * (gate.double() > 0)).float(), torch.bfloat16, msg=f"wsk dX gate K={K}")


@pytest.mark.parametrize("cdtype", [torch.bfloat16, torch.float32])
def test_gemm_streaming_dx_and_headmajor(ops, cdtype):
    M, n_out = 4999, 256
    for k_in in (256, 1024):
        dy = _rand(M, n_out, seed=94).to(torch.bfloat16)
        w = _rand(n_out, k_in, seed=95, scale=1 / 16).to(torch.bfloat16)
        gate = _rand(M, k_in, seed=96).to(cdtype)
        addend = _rand(M, k_in, seed=97).to(cdtype)
        out = dev(addend.clone())
        ops.linear_dx(dev(dy), dev(w), out, rows=M, add_src=out, gate_ref=dev(gate), gate_scale=1.25)
        ref = (dy.float() @ w.float()) * 1.25 * (gate.float() > 0) + addend.float()
        _close(out, ref, torch.bfloat16, msg=f"streaming dX k_in={k_in}")
    Nn, S, Mh, D = 2, 2600, 16, 16
    x = _rand(Nn * S, 256, seed=98).to(torch.bfloat16)
    wv = _rand(Mh * D, 256, seed=99, scale=1 / 16).to(torch.bfloat16)
    mask = (torch.arange(Nn * S) % 7 == 0).to(torch.uint8)
    v = torch.empty(Nn, Mh, S, D, dtype=cdtype, device="cuda")
    ops.linear_fwd(dev(x), dev(wv), None, v, row_mask=dev(mask), head_major=(Mh, S, D))
    ref = (x.float() @ wv.float().t()).masked_fill(mask.bool()[:, None], 0).view(Nn, S, Mh, D).permute(0, 2, 1, 3)
    _close(v, ref, torch.bfloat16, msg="streaming head-major")


@pytest.mark.parametrize("n_out,k_in", [(256, 256), (768, 256), (256, 1024), (128, 384)])
def test_gemm_dw_streaming(ops, n_out, k_in):
    """rows >= 4096, bf16 operands, 128-multiples: poet_gemm routes the dW form to gemm_dw.hip (LDS transpose reads).
    Also with operands that are column slices (ld > width) and a ragged row count."""
    rows = 9000 + 37
    dyf = _rand(rows, n_out + 64, seed=110).to(torch.bfloat16)
    xf = _rand(rows, k_in + 128, seed=111).to(torch.bfloat16)
    dy, x = dyf[:, 64:], xf[:, :k_in]
    dw = torch.zeros(n_out, k_in, device="cuda")
    base = _rand(n_out, k_in, seed=112)
    dw.copy_(base)                                       # accumulates onto existing content
    gdy, gx = dev(dyf), dev(xf)
    ops.linear_dw(gdy[:, 64:], gx[:, :k_in], dw, rows=rows, ldy=n_out + 64, ldx=k_in + 128)
    ref = dy.float().t() @ x.float() + base
    _close(dw, ref, torch.bfloat16, scale=math.sqrt(rows), msg="streaming dW")


@pytest.mark.parametrize("rows,n_out,k_in", [(9037, 256, 384), (9037, 1024, 256), (700, 256, 256), (5000, 66, 256)])
def test_gemm_dw_bias_gradient(ops, rows, n_out, k_in):
    """linear_dw(db=...): db += column sums of dy -- fused in gemm_dw.hip (first two) or a separate launch behind the
    same C call (small / ragged shapes)."""
    dy = _rand(rows, n_out, seed=120).to(torch.bfloat16)
    x = _rand(rows, k_in, seed=121).to(torch.bfloat16)
    dw = torch.zeros(n_out, k_in, device="cuda")
    db0 = _rand(n_out, seed=122)
    db = dev(db0.clone())
    ops.linear_dw(dev(dy), dev(x), dw, rows=rows, db=db)
    _close(dw, dy.float().t() @ x.float(), torch.bfloat16, scale=math.sqrt(rows), msg="dW")
    _close(db, dy.float().sum(0) + db0, torch.float32, scale=math.sqrt(rows), msg="db")


def test_pose_loss_fused_matches_torch(ops):
    """poet_pose_loss (all decoder layers, both terms, gradients) against the PyTorch formulation of
    pose_estimation_transformer.py:635-674 incl. autograd, through SetCriterion.total()."""
    import poet_amd
    from poet_amd.functional import PoseLossFn
    torch.manual_seed(5)
    L, N, Q, n_obj = 5, 4, 20, 37
    trans = torch.randn(L, N, Q, 3)
    A = torch.randn(L, N, Q, 3, 3)
    rot = torch.linalg.qr(A)[0]
    qi = torch.randperm(N * Q)[:n_obj]
    tt = torch.randn(n_obj, 3)
    tr = torch.linalg.qr(torch.randn(n_obj, 3, 3))[0]
    tr[0] = rot[2].reshape(-1, 3, 3)[qi[0]]                     # identical rotation: clamp active, zero gradient
    w = torch.rand(L, 2)
    def torch_form(t, r):
        st = t.reshape(L, -1, 3)[:, qi]; sr = r.reshape(L, -1, 3, 3)[:, qi]
        lt = torch.sqrt(((st - tt) ** 2).sum(-1)).sum(-1) / n_obj
        x = 0.5 * ((sr * tr).sum((-1, -2)) - 1)
        lr = torch.acos(torch.clamp(x, -1 + 1e-6, 1 - 1e-6)).sum(-1) / n_obj
        return torch.stack([lt, lr], 1)
    t0, r0 = trans.clone().requires_grad_(), rot.clone().requires_grad_()
    ref = torch_form(t0, r0)
    (ref * w).sum().backward()
    t1, r1 = dev(trans).requires_grad_(), dev(rot).requires_grad_()
    vec = PoseLossFn.apply(t1, r1, dev(qi), dev(tt), dev(tr), n_obj)
    (vec * dev(w)).sum().backward()
    assert torch.allclose(vec.cpu(), ref.detach(), atol=2e-5, rtol=1e-5), (vec.cpu() - ref).abs().max()
    assert torch.allclose(t1.grad.cpu(), t0.grad, atol=1e-6, rtol=1e-4)
    # the layer-2 pair with an identical target sits exactly on the clamp: torch passes no gradient there either
    assert torch.allclose(r1.grad.cpu(), r0.grad, atol=2e-5, rtol=2e-3), (r1.grad.cpu() - r0.grad).abs().max()
    # (ABI v5) the loss weights inside the kernel: weighted gradients + the weighted total in ONE launch, count read on the device
    # (what GraphedTrainer captures: no framework multiply / reduce kernels behind the loss)
    lv = torch.empty(L, 2, device="cuda"); gt = torch.empty_like(dev(trans)); gr = torch.empty_like(dev(rot)); tot = torch.empty((), device="cuda")
    ops.pose_loss(dev(trans), dev(rot), dev(qi), dev(tt), dev(tr), 0, lv, gt, gr, n_obj_dev=torch.tensor([n_obj], dtype=torch.int32, device="cuda"),
                  weights=dev(w), total=tot)
    assert torch.allclose(lv.cpu(), ref.detach(), atol=2e-5, rtol=1e-5)
    assert abs(float(tot) - float((ref.detach() * w).sum())) < 2e-5 * max(1.0, float((ref.detach() * w).sum()))
    assert torch.allclose(gt.cpu(), t0.grad, atol=1e-6, rtol=1e-4) and torch.allclose(gr.cpu(), r0.grad, atol=2e-5, rtol=2e-3)


# ------------------------------------------------------------------ latency-oriented fp32 kernels (gemm_small.hip)
@pytest.mark.parametrize("M,N,K", [(320, 256, 256), (320, 132, 256), (320, 1024, 256), (320, 256, 1024), (37, 48, 512), (1000, 256, 768), (1600, 256, 256)])
def test_gemm_small_fwd_dx(ops, M, N, K):
    """fp32, <= 1024 rows: poet_gemm routes forward and input-gradient products to gemm_small.hip (K >= 512 and a multiple
    of 64: reduction split over the 4 waves).  Dropout pattern against the tiled kernel (disabled by a K that is not a
    multiple of 16) is index-based, so only its rate is checked here."""
    x = _rand(M, K, seed=130)
    w = _rand(N, K, seed=131, scale=1 / math.sqrt(K))
    b = _rand(N, seed=132)
    add = _rand(M, N, seed=133)
    out = torch.empty(M, N, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), out, act=1, add_src=dev(add))
    _close(out, torch.relu(x @ w.t() + b) + add, torch.float32, msg="small fwd")
    o2 = torch.empty(M, N, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(b), o2, drop_p=0.3, seed=5)
    ref = x @ w.t() + b
    keep = o2.cpu() != 0
    assert abs(keep.float().mean().item() - 0.7) < 0.03
    assert torch.allclose(o2.cpu()[keep], (ref / 0.7)[keep], atol=2e-4, rtol=2e-4)
    # input gradient: dy (M, N_out = N) @ W (N, K) with ReLU gate and in-place accumulate
    dy = _rand(M, N, seed=134)
    gate = _rand(M, K, seed=135)
    acc0 = _rand(M, K, seed=136)
    dx = dev(acc0.clone())
    ops.linear_dx(dev(dy), dev(w), dx, rows=M, add_src=dx, gate_ref=dev(gate), gate_scale=1.5)
    _close(dx, (dy @ w) * 1.5 * (gate > 0) + acc0, torch.float32, msg="small dX")


@pytest.mark.parametrize("rows,n_out,k_in", [(320, 256, 256), (320, 132, 256), (320, 1024, 256), (333, 256, 1024), (1024, 48, 40), (20, 128, 64), (100, 40, 68), (700, 24, 256)])
def test_gemm_small_dw_db(ops, rows, n_out, k_in):
    dy = _rand(rows, n_out, seed=140)
    x = _rand(rows, k_in, seed=141)
    w0, b0 = _rand(n_out, k_in, seed=142), _rand(n_out, seed=143)
    dw, db = dev(w0.clone()), dev(b0.clone())
    ops.linear_dw(dev(dy), dev(x), dw, rows=rows, db=db)
    ops.linear_dw(dev(dy), dev(x), dw, rows=rows, db=db)          # accumulates onto what is there
    _close(dw, 2 * (dy.t() @ x) + w0, torch.float32, scale=math.sqrt(rows), msg="small dW")
    _close(db, 2 * dy.sum(0) + b0, torch.float32, scale=math.sqrt(rows), msg="small db")


def test_gemm_small_batched_matches_loop(ops):
    """batch > 1 in the 320-row kernels (independent problems of one shape at a constant stride: the pose heads of the
    L decoder layers inside the flat parameter arena): forward with bias + ReLU, gated dX accumulated onto a tensor, and
    dW + db must equal the same calls issued one problem at a time, bit for bit."""
    nb, rows, n_out, k_in = 5, 320, 256, 256
    pad = 1000                                                    # the copies are NOT back to back
    Wbuf = dev(_rand(nb * (n_out * k_in + pad), seed=1, scale=0.06))
    bbuf = dev(_rand(nb * (n_out + 24), seed=2))
    sW, sb = n_out * k_in + pad, n_out + 24
    Ws = [Wbuf[j * sW: j * sW + n_out * k_in].view(n_out, k_in) for j in range(nb)]
    bs = [bbuf[j * sb: j * sb + n_out] for j in range(nb)]
    x = dev(_rand(nb, rows, k_in, seed=3))
    dy = dev(_rand(nb, rows, n_out, seed=4))
    # forward
    y1, y2 = torch.empty(nb, rows, n_out, device="cuda"), torch.empty(nb, rows, n_out, device="cuda")
    ops.gemm(x, Ws[0], y1, rows, n_out, k_in, lda=k_in, ldb=k_in, ldc=n_out, bias=bs[0], act=1, batch=nb,
             strideA=rows * k_in, strideB=sW, strideC=rows * n_out, stride_bias=sb)
    for j in range(nb):
        ops.linear_fwd(x[j], Ws[j], bs[j], y2[j], act=1)
    assert torch.equal(y1, y2)
    ref = torch.relu(x[3].cpu() @ Ws[3].cpu().t() + bs[3].cpu())
    assert (y1[3].cpu() - ref).abs().max().item() < 2e-3
    # dX with gate and accumulation
    acc1 = dev(_rand(nb, rows, k_in, seed=5)); acc2 = acc1.clone()
    ops.gemm(dy, Ws[0], acc1, rows, k_in, n_out, lda=n_out, ldb=k_in, ldc=k_in, b_kmajor=True, gate_ref=x, add_src=acc1, ld_add=k_in,
             batch=nb, strideA=rows * n_out, strideB=sW, strideC=rows * k_in)
    for j in range(nb):
        ops.linear_dx(dy[j], Ws[j], acc2[j], rows=rows, gate_ref=x[j], add_src=acc2[j])
    assert torch.equal(acc1, acc2)
    # dW + db
    g1, g2 = torch.zeros_like(Wbuf), torch.zeros_like(Wbuf)
    d1, d2 = torch.zeros_like(bbuf), torch.zeros_like(bbuf)
    ops.gemm(dy, x, g1, n_out, k_in, rows, lda=n_out, ldb=k_in, ldc=k_in, a_kmajor=True, b_kmajor=True, atomic=True, bias=d1, batch=nb,
             strideA=rows * n_out, strideB=rows * k_in, strideC=sW, stride_bias=sb)
    for j in range(nb):
        ops.linear_dw(dy[j], x[j], g2[j * sW: j * sW + n_out * k_in].view(n_out, k_in), rows=rows, db=d2[j * sb: j * sb + n_out])
    assert torch.equal(g1, g2) and torch.equal(d1, d2)
    assert float(g1[n_out * k_in: sW].abs().max()) == 0.0        # the gaps between the copies stay untouched


def test_gemm_tiled_batched_gate_matches_loop(ops):
    """batch > 1 with gate_ref / add_src in the GENERIC tiled kernel (the gated input gradient of the pose heads' last Linear,
    (n_classes + 1) * 3 = 66 or * 6 = 132 outputs wide: K % 16 != 0 keeps it out of the <= 1024-row kernel): one launch for the L
    decoder layers' heads must equal one launch per head, bit for bit, and torch."""
    nb, rows, k_in = 5, 800, 256
    for n_out in (66, 132):
        pad = 1000
        Wbuf = dev(_rand(nb * (n_out * k_in + pad), seed=1, scale=0.06))
        sW = n_out * k_in + pad
        Ws = [Wbuf[j * sW: j * sW + n_out * k_in].view(n_out, k_in) for j in range(nb)]
        x = dev(_rand(nb, rows, k_in, seed=3))
        dy = dev(_rand(nb, rows, n_out, seed=4))
        a1 = dev(_rand(nb, rows, k_in, seed=5)); a2 = a1.clone()
        ops.gemm(dy, Ws[0], a1, rows, k_in, n_out, lda=n_out, ldb=k_in, ldc=k_in, b_kmajor=True, gate_ref=x, add_src=a1, ld_add=k_in,
                 batch=nb, strideA=rows * n_out, strideB=sW, strideC=rows * k_in)
        from poet_amd import _lib
        assert _lib.load().poet_gemm_last_path() == 1                    # POET_GEMM_PATH_TILED
        b1 = torch.empty(nb, rows, k_in, device="cuda"); b2 = torch.empty_like(b1)
        ops.gemm(dy, Ws[0], b1, rows, k_in, n_out, lda=n_out, ldb=k_in, ldc=k_in, b_kmajor=True, gate_ref=x,
                 batch=nb, strideA=rows * n_out, strideB=sW, strideC=rows * k_in)
        for j in range(nb):
            ops.linear_dx(dy[j], Ws[j], a2[j], rows=rows, gate_ref=x[j], add_src=a2[j])
            ops.linear_dx(dy[j], Ws[j], b2[j], rows=rows, gate_ref=x[j])
        assert torch.equal(a1, a2) and torch.equal(b1, b2)
        ref = (dy[2].cpu() @ Ws[2].cpu()) * (x[2].cpu() > 0)
        assert (b1[2].cpu() - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-4


def test_gemm_dw_list_matches_loop(ops):
    """poet_gemm_dw_list (the dW + db of one Linear of every decoder layer in one launch, operands at unrelated addresses)
    through `ops.defer_small_dw`: identical to one launch per layer, also with strided dY (a column slice) and without a
    bias gradient; a record sequence that differs between layers falls back to single launches."""
    rows, n_out, k_in, nl = 320, 256, 256, 5
    res = {}
    for mode in ("loop", "deferred"):
        dys = [dev(_rand(rows, 3 * n_out, seed=10 + i)) for i in range(nl)]
        xs = [dev(_rand(rows, k_in, seed=20 + i)) for i in range(nl)]
        gW = [torch.ones(n_out, k_in, device="cuda") for _ in range(nl)]
        gb = [torch.ones(n_out, device="cuda") for _ in range(nl)]
        gW2 = [torch.zeros(n_out, k_in, device="cuda") for _ in range(nl)]
        ctx = ops.defer_small_dw() if mode == "deferred" else contextlib.nullcontext()
        with ctx as D:
            for i in range(nl):
                if D is not None:
                    D.next_layer()
                ops.linear_dw(dys[i][:, n_out: 2 * n_out], xs[i], gW[i], rows=rows, ldy=3 * n_out, db=gb[i])
                ops.linear_dw(dys[i], xs[i], gW2[i], rows=rows, ldy=3 * n_out)
        res[mode] = [t.cpu() for t in gW + gb + gW2]
    for a, b in zip(res["loop"], res["deferred"]):
        assert torch.equal(a, b)
    dy3, x3 = _rand(rows, 3 * n_out, seed=13), _rand(rows, k_in, seed=23)
    assert (res["deferred"][3] - (1 + dy3[:, n_out: 2 * n_out].t() @ x3)).abs().max().item() < 2e-3
    # ragged record sequences -> fallback
    g1, g2 = torch.zeros(n_out, k_in, device="cuda"), torch.zeros(64, k_in, device="cuda")
    dy_a, dy_b, x_a = dev(_rand(rows, n_out, seed=31)), dev(_rand(rows, 64, seed=32)), dev(_rand(rows, k_in, seed=33))
    with ops.defer_small_dw() as D:
        D.next_layer(); ops.linear_dw(dy_a, x_a, g1, rows=rows)
        D.next_layer(); ops.linear_dw(dy_b, x_a, g2, rows=rows)
    assert (g1.cpu() - dy_a.cpu().t() @ x_a.cpu()).abs().max().item() < 2e-3 and (g2.cpu() - dy_b.cpu().t() @ x_a.cpu()).abs().max().item() < 2e-3


def test_gemm_dw_multi_lists_of_different_shapes(ops):
    """poet_gemm_dw_multi (every deferred weight gradient of the decoder stack as block ranges of ONE launch): the seven Linears of a
    decoder layer with their real shapes (FFN 256 <-> 1024, the 768-wide offsets | logits pair, in-projection slices of a packed
    buffer, one without a bias gradient), 5 layers -- bit-identical to one launch per (layer, Linear); POET_NO_DW_MULTI's per-Linear
    lists are the fallback (k_in % 4 != 0 takes it)."""
    rows, nl = 320, 5
    shapes = [(256, 1024, True), (1024, 256, True), (256, 256, True), (768, 256, True), (256, 256, False), (512, 256, True), (256, 256, True)]
    res = {}
    for mode in ("loop", "deferred"):
        grads = []
        ctx = ops.defer_small_dw() if mode == "deferred" else contextlib.nullcontext()
        with ctx as D:
            for i in range(nl):
                if D is not None:
                    D.next_layer()
                for j, (n_out, k_in, bias) in enumerate(shapes):
                    dy = dev(_rand(rows, n_out + 64, seed=700 + 10 * i + j))
                    x = dev(_rand(rows, k_in, seed=800 + 10 * i + j))
                    gW = torch.full((n_out, k_in), 0.5, device="cuda")
                    gb = torch.full((n_out,), 0.25, device="cuda") if bias else None
                    ops.linear_dw(dy[:, 32:32 + n_out], x, gW, rows=rows, ldy=n_out + 64, db=gb)
                    grads += [gW] + ([gb] if bias else [])
        torch.cuda.synchronize()
        res[mode] = [t.cpu() for t in grads]
    assert len(res["loop"]) == len(res["deferred"])
    for a, b in zip(res["loop"], res["deferred"]):
        assert torch.equal(a, b)
    dy0, x0 = _rand(rows, 256 + 64, seed=700), _rand(rows, 1024, seed=800)
    assert (res["deferred"][0] - (0.5 + dy0[:, 32:32 + 256].t() @ x0)).abs().max().item() < 5e-3


# ------------------------------------------------------------------------------------------- on-device matcher
def test_lsa_boxes_matches_scipy_including_ties(ops):
    """poet_lsa_boxes == scipy.optimize.linear_sum_assignment on the fp32 L1 box cost (models/matcher.py:60-75,158-229): random
    boxes, exact duplicates (zero-cost ties), quantised boxes (many equal costs), fewer / more predictions than targets,
    images without targets or predictions."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    N, Q = 24, 50
    pred = np.full((N, Q, 4), -1.0, np.float32)
    tgts, n_pred, off = [], [], [0]
    for i in range(N):
        nt = int(rng.integers(0, 51)) if i % 7 else 0
        kind = i % 4
        if kind == 0:
            t = rng.uniform(0, 1, (nt, 4))
        elif kind == 1:
            t = np.round(rng.uniform(0, 1, (nt, 4)) * 4) / 4                     # coarse grid: many equal costs
        elif kind == 2:
            t = np.repeat(rng.uniform(0, 1, (max(nt // 3, 1), 4)), 3, 0)[:nt]    # exact duplicates
        else:
            t = rng.uniform(0, 1, (nt, 4)) * 0 + 0.5                             # constant matrix (scipy: identity)
        t = t.astype(np.float32)
        nt = len(t)
        npd = nt if i % 5 else int(rng.integers(0, 51))                           # usually square ('gt' mode), sometimes rectangular
        if i % 3 == 0 and nt:
            p_ = t[rng.permutation(nt)][:npd] if npd <= nt else np.concatenate([t, rng.uniform(0, 1, (npd - nt, 4)).astype(np.float32)])
        else:
            p_ = rng.uniform(0, 1, (npd, 4)).astype(np.float32)
            if kind == 1:
                p_ = (np.round(p_ * 4) / 4).astype(np.float32)
        npd = len(p_)
        pred[i, :npd] = p_
        tgts.append(t); n_pred.append(npd); off.append(off[-1] + nt)
    tb = np.concatenate(tgts + [np.zeros((1, 4), np.float32)])
    col = torch.full((N, Q), -7, dtype=torch.int32, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.lsa_boxes(dev(torch.from_numpy(pred)), dev(torch.from_numpy(tb)), dev(torch.tensor(off, dtype=torch.int32)),
                  dev(torch.tensor(n_pred, dtype=torch.int32)), col, status, 1.0)
    assert int(status.item()) == 0
    col = col.cpu().numpy()
    for i in range(N):
        c = np.float32(1.0) * np.abs(pred[i, : n_pred[i], None, :] - tgts[i][None, :, :]).sum(-1)
        exp = np.full(Q, -1, np.int64)
        if c.size:
            r, cc = linear_sum_assignment(c)
            exp[r] = cc
        assert np.array_equal(col[i], exp), (i, n_pred[i], len(tgts[i]), col[i][:12], exp[:12])


@pytest.mark.parametrize("M", [5000 + 8, 4096, 16 * 417 + 5, 53 * 256])
@pytest.mark.parametrize("K", [512, 1024, 1280])
def test_gemm_pipe_long_k(ops, M, K):
    """Plain tall-skinny products C (fp32) (+)= A (bf16) W (bf16) with N = 256, K >= 512 -- the encoder's K = 1024 / 1280 input
    gradients and the FFN's second Linear -- run on the library's own deep-pipeline kernel (gemm_pipe.hip: persistent, LDS-DMA
    rings, transposing LDS reads for the [K][N] weight) and nothing else: against a float64 product, for both weight layouts,
    written (+ bias) and accumulated in place, on row counts that are ragged (partial last 16-row unit), smaller than one unit
    per CU, and that give every workgroup tiles of unequal height; results must be bit-identical between calls (no atomics, no
    arrival order) -- and the two-image split product (B_lo) must equal the fp32 weight to 2^-16."""
    import poet_amd._lib as L
    lib = L.load()
    N = 256
    a = _rand(M, K, seed=300 + K).to(torch.bfloat16)
    w_kn = _rand(K, N, seed=301 + K, scale=1 / math.sqrt(K)).to(torch.bfloat16)      # W[K][N]: the dX form (weight (n_out, k_in))
    acc0 = _rand(M, N, seed=302 + K)
    bias = _rand(N, seed=304 + K)
    ref = a.double() @ w_kn.double()
    sc = ref.abs().max().item()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    ops.linear_dx(dev(a), dev(w_kn), out, rows=M)
    assert lib.poet_gemm_last_path() == 5, lib.poet_gemm_last_path()                 # POET_GEMM_PATH_PIPE
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-5 * sc
    out_b = torch.full_like(out, float("nan"))
    ops.linear_dx(dev(a), dev(w_kn), out_b, rows=M)
    assert torch.equal(out, out_b)                                                    # deterministic
    acc = dev(acc0.clone())
    ops.linear_dx(dev(a), dev(w_kn), acc, rows=M, add_src=acc)
    assert lib.poet_gemm_last_path() == 5
    assert (acc.double().cpu() - (ref + acc0.double())).abs().max().item() <= 2e-5 * (sc + acc0.abs().max().item())
    # [N][K] weight (the forward form) + bias
    out2 = torch.full_like(out, float("nan"))
    ops.linear_fwd(dev(a), dev(w_kn.t().contiguous()), dev(bias), out2)
    assert lib.poet_gemm_last_path() == 5
    assert (out2.double().cpu() - (ref + bias.double())).abs().max().item() <= 2e-5 * sc
    # row-strided operands: A and C as column blocks of wider buffers (the packed [d(offsets|logits) | d(value) rows] buffer)
    wide = torch.zeros(M, K + 64, dtype=torch.bfloat16, device="cuda")
    wide[:, 32:32 + K] = dev(a)
    cw = torch.zeros(M, N + 8, dtype=torch.float32, device="cuda")
    ops.gemm(wide[:, 32:], dev(w_kn), cw[:, 4:], M, N, K, lda=K + 64, ldb=N, ldc=N + 8, b_kmajor=True)
    assert lib.poet_gemm_last_path() == 5
    assert (cw[:, 4:4 + N].double().cpu() - ref).abs().max().item() <= 2e-5 * sc and (cw[:, :4] == 0).all() and (cw[:, 4 + N:] == 0).all()
    # the split-weight product on two bf16 images (hi, lo = bf16(W - hi)): ONE pass; == the fp32 weight to 2^-16, and == the
    # split-weight kernel that takes the fp32 master
    w32 = _rand(N, K, seed=303 + K, scale=1 / math.sqrt(K))
    hi = w32.to(torch.bfloat16)
    lo = (w32 - hi.float()).to(torch.bfloat16)
    two = torch.full_like(out, float("nan"))
    ops.linear_fwd(dev(a), dev(hi), dev(bias), two, W_lo=dev(lo))
    assert lib.poet_gemm_last_path() == 5
    exact = a.double() @ w32.double().t() + bias.double()
    sc = exact.abs().max().item()
    assert (two.double().cpu() - exact).abs().max().item() <= 2e-5 * sc
    one = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.linear_fwd(dev(a), dev(w32), dev(bias), one, split=True)
    assert (two.cpu() - one.cpu()).abs().max().item() <= 2e-5 * sc
    single = torch.empty_like(one)
    ops.linear_fwd(dev(a), dev(hi), dev(bias), single)
    assert (single.double().cpu() - exact).abs().max().item() > 20 * (two.double().cpu() - exact).abs().max().item()
    # (round 6) the same products stored as IEEE fp16 (PoetGemmDesc.c_f16: the operand the encoder's LayerNorm reads): THE fp16
    # rounding of the fp32 result almost everywhere (ties of the accumulation order aside), on the pipe kernel, deterministic
    for kw, want in ((dict(W_lo=dev(lo)), two), (dict(), single)):
        h = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        ops.linear_fwd(dev(a), dev(hi), dev(bias), h, **kw)
        assert lib.poet_gemm_last_path() == 5
        assert torch.isfinite(h.float()).all()
        assert (h.cpu() == want.cpu().to(torch.float16)).float().mean().item() > 0.999
        assert (h.double().cpu() - want.double().cpu()).abs().max().item() <= 2.0 ** -11 * sc * 1.01
        h2 = torch.full_like(h, float("nan"))
        ops.linear_fwd(dev(a), dev(hi), dev(bias), h2, **kw)
        assert torch.equal(h, h2)


def test_gemm_pipe_grid_independent(ops):
    """The persistent kernel deals 16-row units to workgroups and cuts them into tiles; the result may not depend on how:
    bit-identical outputs for 256 and 7 workgroups (POET_PIPE_GRID is latched per process: run the second in a fresh one)."""
    import subprocess, sys, os
    code = ("import torch, math, sys, hashlib; sys.path.insert(0, %r); from poet_amd import ops; from tests.test_kernels_gpu import _rand;"
            "M,K,N=6000,1024,256; a=_rand(M,K,seed=1).to(torch.bfloat16).cuda(); w=_rand(K,N,seed=2,scale=1/32).to(torch.bfloat16).cuda();"
            "c=_rand(M,N,seed=3).cuda(); ops.linear_dx(a,w,c,rows=M,add_src=c); print('SUM', hashlib.sha1(c.cpu().numpy().tobytes()).hexdigest())"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for grid in ("0", "7", "100"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, POET_PIPE_GRID=grid), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("SUM")][0])
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("HW", [300, 1217])
def test_groupnorm_whole_row_kernels(ops, xdt, ydt, HW, monkeypatch):
    """C = 256, G = 32, >= 256 tokens: the whole-row GroupNorm kernels (row blocks of 64, partial sums through the caller's
    scratch) -- against torch, and against the one-workgroup-per-group kernels they replace (POET_GN_NO_ROWS is read once per
    process, so that comparison is through torch's result).  Odd token counts: a partial last row block."""
    N, C, G, S, off = 3, 256, 32, HW + 70, 50
    x = (_rand(N, HW, C, seed=330) * 1.5 + 0.3).to(xdt)
    gamma = 1 + 0.1 * _rand(C, seed=331)
    beta = 0.1 * _rand(C, seed=332)
    y = torch.zeros(N, S, C, dtype=ydt, device="cuda")
    stats = torch.empty(N, G, 2, device="cuda")
    x16 = torch.empty(N, HW, C, dtype=torch.bfloat16, device="cuda") if xdt == torch.float32 else None
    ops.groupnorm_fwd(dev(x), dev(gamma), dev(beta), y, stats, N, HW, C, G, 0, HW, off, S, x16=x16)
    if x16 is not None:
        assert torch.equal(x16.cpu(), x.to(torch.bfloat16))
    xr = x.float().permute(0, 2, 1).contiguous().requires_grad_()
    g32, b32 = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr = F.group_norm(xr, G, g32, b32, 1e-5)
    _close(y[:, off:off + HW], yr.detach().permute(0, 2, 1), ydt, msg="gn rows fwd")
    assert (y[:, :off] == 0).all() and (y[:, off + HW:] == 0).all()          # nothing outside the level's token range
    mu = x.float().view(N, HW, G, 8).mean((1, 3))
    assert torch.allclose(stats[..., 0].cpu(), mu, atol=1e-5)
    dy = torch.zeros(N, S, C)
    dy[:, off:off + HW] = _rand(N, HW, C, seed=333)
    dy = dy.to(ydt)
    yr.backward(dy[:, off:off + HW].float().permute(0, 2, 1))
    dx = torch.empty(N, HW, C, dtype=xdt, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    ops.groupnorm_bwd(dev(dy), dev(x), stats, dev(gamma), dx, dg, db, N, HW, C, G, 0, HW, off, S)
    _close(dx, xr.grad.permute(0, 2, 1), xdt, msg="gn rows dx")
    _close(dg, g32.grad, ydt, scale=math.sqrt(N * HW), msg="gn rows dgamma")
    _close(db, b32.grad, ydt, scale=math.sqrt(N * HW), msg="gn rows dbeta")


@pytest.mark.parametrize("n_img,shapes,n_out,k_in", [(16, [(60, 80), (30, 40), (15, 20), (8, 10)], 768, 256),      # the encoder's d(offsets | logits) at 640x480, bs 16
                                                      (3, [(60, 80), (30, 40), (15, 20), (8, 10)], 768, 256),       # row ranges that straddle images and levels
                                                      (8, [(30, 40), (15, 20), (8, 10), (4, 5)], 1024, 256),        # LM-O levels (the smallest has 20 rows: many segments inside one 64-row stage)
                                                      (2, [(120, 160), (60, 80), (30, 40), (15, 20)], 256, 256),    # a shape the DMA ring declines: separate column-sum launch
                                                      (1, [(60, 80), (30, 40), (15, 20), (8, 10)], 768, 256),       # 6380 rows (bs 1): below the ring's row count
                                                      (1, [(12, 16), (6, 8), (3, 4)], 192, 64)])                    # 264 rows: the small-GEMM path
def test_gemm_dw_segment_sums(ops, n_img, shapes, n_out, k_in):
    """PoetGemmDesc.seg_sums (ABI v4): per-LEVEL column sums of dY out of the weight-gradient kernel's own pass over dY (the DMA-ring
    kernel multiplies its dY fragments with a level-indicator fragment: deformable_transformer.py:139-141 makes d(level_embed) the sum
    over a level's rows) == the column-sum kernel; the weight gradient itself is unchanged."""
    geom = ops.LevelGeom(shapes)
    S, L = geom.S, len(shapes)
    rows = n_img * S
    # dY is a column block of a wider buffer (the encoder writes d(offsets | logits) next to d(value) rows: row stride 3 M L P + d)
    wide = _rand(rows, n_out + 256, seed=610).to(torch.bfloat16)
    dy = wide[:, :n_out]
    x = _rand(rows, k_in, seed=611).to(torch.bfloat16)
    dyd, xd = dev(wide)[:, :n_out], dev(x)
    ldy = n_out + 256
    dw_a, dw_b = torch.zeros(n_out, k_in, device="cuda"), torch.zeros(n_out, k_in, device="cuda")
    seg_a = torch.full((L, n_out), 0.5, device="cuda")                          # (accumulates: starts from a non-zero value)
    seg_b = torch.full((L, n_out), 0.5, device="cuda")
    ops.linear_dw(dyd, xd, dw_a, rows=rows, ldy=ldy, seg=(seg_a, geom.c_segs, S))
    ops.linear_dw(dyd, xd, dw_b, rows=rows, ldy=ldy)
    ops.colsum(dyd, ldy, seg_b, n_img, S, n_out, geom.c_segs, L)
    torch.cuda.synchronize()
    assert torch.equal(dw_a, dw_b)
    starts = [int(v) for v in geom.c_segs]
    ref = torch.stack([dy.float().view(n_img, S, n_out)[:, starts[l]:starts[l + 1]].double().sum((0, 1)) for l in range(L)]) + 0.5
    scale = ref.abs().max().item()
    assert (seg_a.double().cpu() - ref).abs().max().item() < 2e-5 * scale + 1e-3
    assert (seg_a - seg_b).abs().max().item() < 2e-5 * scale + 1e-3


@pytest.mark.parametrize("rows,n2,gate", [(102080, 1024, True), (102080, 256, False), (8192 + 13, 1024, True), (204000, 1024, True), (9000, 128, False),
                                          (51200, 256, True)])
def test_linear_bwd_fused_equals_dw_plus_dx(ops, rows, n2, gate):
    """poet_linear_bwd (gemm_dwx_kernel: the weight-gradient ring with the Linear's input gradient computed from the SAME staged
    panels) against the two poet_gemm calls it replaces -- dW (+ db) by the weight-gradient kernels, dX by the streaming input-gradient
    kernel with its ReLU / dropout gate -- and against fp64 on the same bf16 operands.  FFN linear2 (n2 = 1024, gated by the hidden
    activation: deformable_transformer.py:193-197) and output_proj (n2 = 256, plain: :202-203); ragged row counts."""
    dy = (_rand(rows, 256, seed=700) * 0.5).to(torch.bfloat16)
    x = torch.relu(_rand(rows, n2, seed=701))
    x = (x * (torch.from_numpy(np.random.default_rng(5).random((rows, n2)).astype(np.float32)) > 0.1)).to(torch.bfloat16)   # ReLU + dropout zeros
    w = (_rand(256, n2, seed=702) / 16).to(torch.bfloat16)
    dyd, xd, wd = dev(dy), dev(x), dev(w)
    assert ops.linear_bwd_ok(dyd, xd, wd, torch.zeros(256, n2, device="cuda"), rows)
    scale = 1.0 / 0.9
    dw_f, db_f = torch.full((256, n2), 0.25, device="cuda"), torch.full((256,), 0.25, device="cuda")          # accumulate on top of a non-zero value
    dx_f = torch.full((rows + 2, n2), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.linear_bwd(dyd, xd, wd, dw_f, db_f, dx_f[:rows], rows=rows, gate=gate, gate_scale=scale if gate else 1.0)
    dw_t, db_t = torch.full((256, n2), 0.25, device="cuda"), torch.full((256,), 0.25, device="cuda")
    dx_t = torch.empty(rows, n2, dtype=torch.bfloat16, device="cuda")
    ops.linear_dw(dyd, xd, dw_t, rows=rows, db=db_t)
    ops.linear_dx(dyd, wd, dx_t, rows=rows, gate_ref=xd if gate else None, gate_scale=scale if gate else 1.0)
    torch.cuda.synchronize()
    assert bool((dx_f[rows:] == 7.0).all())
    # fp64 yardsticks
    dwr = dy.double().t() @ x.double() + 0.25
    dbr = dy.double().sum(0) + 0.25
    sw, sb = dwr.abs().max().item(), dbr.abs().max().item()
    assert (dw_f.double().cpu() - dwr).abs().max().item() < 2e-5 * sw + 1e-3 and (dw_t.double().cpu() - dwr).abs().max().item() < 2e-5 * sw + 1e-3
    assert (db_f.double().cpu() - dbr).abs().max().item() < 2e-5 * sb + 1e-3
    idx = torch.linspace(0, rows - 1, 509).long()
    dxr = dy[idx].double() @ w.double()
    if gate:
        dxr = torch.where(x[idx].double() > 0, dxr * scale, torch.zeros_like(dxr))
    sx = dxr.abs().max().item()
    assert (dx_f[:rows].cpu()[idx].double() - dxr).abs().max().item() <= 2.0 ** -8 * sx + 1e-6          # one bf16 rounding of the result
    assert (dx_f[:rows].float() - dx_t.float()).abs().max().item() <= 2.0 ** -7 * sx                     # two kernels, two summation orders
    if gate:
        assert bool(((dx_f[:rows] == 0) | (xd > 0)).all())                                               # the gate's zeros exactly


@pytest.mark.parametrize("rows,with_seg,m_alt", [(102080, True, 768), (6380 * 3, True, 768), (6380, True, 768), (9000, False, 768), (6380 * 2, True, 96)])
def test_gemm_dw_two_inputs(ops, rows, with_seg, m_alt):
    """PoetGemmDesc.B_alt (ABI v4): the rows m >= m_alt of the weight gradient pair with a second input -- the encoder's stacked
    [sampling_offsets ; attention_weights ; value_proj] gradient in ONE launch (gradient rows = column blocks of one buffer; inputs src + pos
    and src) -- against two separate products and fp64; with the per-level column sums over all columns riding along.  6380-row case: the
    DMA ring declines (row count), poet_gemm falls back to a column-sum launch + two plain products."""
    shapes = [(60, 80), (30, 40), (15, 20), (8, 10)]
    geom = ops.LevelGeom(shapes)
    S, L = geom.S, 4
    if not with_seg:
        S = rows
    G2 = (_rand(rows, 1024, seed=800) * 0.5).to(torch.bfloat16)
    q, src = _rand(rows, 256, seed=801).to(torch.bfloat16), _rand(rows, 256, seed=802).to(torch.bfloat16)
    g2, qd, sd = dev(G2), dev(q), dev(src)
    gw = torch.full((1024, 256), 0.25, device="cuda")
    seg = torch.full((L, 1024), 0.5, device="cuda") if with_seg else None
    ops.linear_dw(g2, qd, gw, rows=rows, ldy=1024, seg=(seg, geom.c_segs, S) if with_seg else None, x_alt=(sd, m_alt))      # (m_alt = 96: 4 heads x 2 levels, no ring tile boundary)
    ref_w = torch.cat([G2[:, :m_alt].double().t() @ q.double(), G2[:, m_alt:].double().t() @ src.double()]) + 0.25
    sc = ref_w.abs().max().item()
    assert (gw.double().cpu() - ref_w).abs().max().item() < 2e-5 * sc + 1e-3
    if with_seg:
        starts = [int(v) for v in geom.c_segs]
        ref_s = torch.stack([G2.float().view(rows // S, S, 1024)[:, starts[l]:starts[l + 1]].double().sum((0, 1)) for l in range(L)]) + 0.5
        assert (seg.double().cpu() - ref_s).abs().max().item() < 2e-5 * ref_s.abs().max().item() + 1e-3
