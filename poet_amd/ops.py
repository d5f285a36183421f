"""Tensor-level wrappers over the C ABI (no autograd here).  torch is plumbing only: it owns the
device buffers and the stream; every computation below is a libpoet_hip.so kernel."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, F32, GemmDesc

import struct as _struct

F16 = 2                                        # POET_F16: IEEE half STORAGE of the offsets | logits buffer (include/poet_hip.h)
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
# PoetGemmDesc as one packed record (natural C layout of include/poet_hip.h; checked against ctypes below)
_GEMM_PACK = _struct.Struct("@8Q 3i 4q 2i 4i i 4q 3i 3f I 4i Q 2i Q q Q Q q 2i 10i Q q 2i")     # native alignment inserts the same padding as the C compiler
assert _GEMM_PACK.size == C.sizeof(GemmDesc) and GemmDesc.seed_dev.offset + 40 + 64 + 24 == _GEMM_PACK.size, "PoetGemmDesc layout drift"

# Device word mixed into every dropout seed (and the Adam step) at run time.  None in eager mode; engine.GraphedTrainer
# sets it so captured hipGraphs draw fresh masks on each replay.
SEED_DEV = [None]


def _seed_dev():
    t = SEED_DEV[0]
    return None if t is None else t.data_ptr()
_GEMM_DESC = GemmDesc()
_GEMM_BUF = (C.c_char * C.sizeof(GemmDesc)).from_buffer(_GEMM_DESC)


# Caller-owned scratch handed to the weight-gradient GEMMs and the whole-row GroupNorm kernels (PoetGemmDesc.workspace,
# `scratch`): one buffer per (device, launch stream), allocated on first use (before any graph capture: the eager warm-up
# steps run every shape).  Users of one buffer are ordered on its stream; the side stream of POET_SIDE_STREAM=1 (weight
# gradients forked off the main stream) gets its own, so a dW reduction never shares scratch with a kernel of the main stream.
_WORKSPACE = {}
_WORKSPACE_BYTES = 48 << 20


def _workspace(device):
    key = (device, _STREAM_OVERRIDE[0] if SIDE.stream is not None and _STREAM_OVERRIDE[0] == SIDE.stream.cuda_stream else 0)
    ws = _WORKSPACE.get(key)
    if ws is None:
        ws = _WORKSPACE[key] = torch.empty(_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    return ws


def dcode(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"poet_amd: unsupported dtype {t.dtype}") from None


_STREAM_OVERRIDE = [None]


def _stream() -> int:
    """Raw hipStream_t of torch's current stream.  `with ops.pinned_stream():` resolves it once for a whole
    forward/backward program (torch.cuda.current_stream() costs ~9 us per call, x ~700 launches per step)."""
    s = _STREAM_OVERRIDE[0]
    return s if s is not None else torch.cuda.current_stream().cuda_stream


class pinned_stream:
    def __enter__(self):
        self.prev = _STREAM_OVERRIDE[0]
        _STREAM_OVERRIDE[0] = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        _STREAM_OVERRIDE[0] = self.prev
        return False


class _SideStream:
    """Weight / bias gradients are consumed only by the optimiser, so their kernels do not belong on the critical path of
    backward (the dX chain).  `SIDE.run(fn, *tensors)` forks a second HIP stream off the current one (event), runs fn there
    and keeps `tensors` alive until `join()`, which every backward node calls before it announces its bucket.  Inside a
    HIP-graph capture the fork/join become parallel branches of the graph.  OPT-IN (POET_SIDE_STREAM=1): on ROCm 7.2 the
    branches of a replayed graph did not overlap (21.8 vs 21.7 ms/step measured), so the default keeps one stream.  Off
    while the per-kernel profiler is on (timings need one stream)."""

    def __init__(self):
        self.enabled = os.environ.get("POET_SIDE_STREAM", "0") not in ("", "0")
        self.stream = None
        self.keep = []
        self.dirty = False

    def run(self, fn, *tensors):
        if not self.enabled or PROFILE.on:
            return fn()
        main = torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)
        self.keep.extend(tensors)
        prev = _STREAM_OVERRIDE[0]
        with torch.cuda.stream(self.stream):
            _STREAM_OVERRIDE[0] = self.stream.cuda_stream
            try:
                fn()
            finally:
                _STREAM_OVERRIDE[0] = prev
        self.dirty = True

    def join(self):
        if self.dirty:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.keep.clear()
            self.dirty = False


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.PoetHipError(f"poet_amd: {name} must live on the GPU (no CPU path exists)")
    return t


def _i64arr(vals: Sequence[int]):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


class _Profile:
    """Optional per-launch timing with HIP events on the launch stream (torch.cuda.Event records on torch's current
    stream, which is the stream every kernel here is enqueued on).  Off by default; bench.py switches it on for a few
    extra steps to measure the roofline numbers.  Each record carries the launch's ALGORITHMIC flops and bytes."""

    def __init__(self):
        self.on = False
        self.rec = []

    def start(self):
        self.on, self.rec = True, []

    def stop(self):
        self.on = False
        torch.cuda.synchronize()
        agg = {}
        for tag, e0, e1, fl, by in self.rec:
            a = agg.setdefault(tag, dict(total_ms=0.0, launches=0, flops=0.0, bytes=0.0))
            a["total_ms"] += e0.elapsed_time(e1)
            a["launches"] += 1
            a["flops"] += fl
            a["bytes"] += by
        self.rec = []
        return agg

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, tag, e0, flops, nbytes):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append((tag, e0, e1, float(flops), float(nbytes)))

    @staticmethod
    def roofline(agg, steps, hbm_peak_gbs, mfma_peak_tfs, tag=None):
        """Roofline line for one kernel family = one kernel symbol on one shape (default: the one that takes the most time)."""
        if tag is None:
            tag = max(agg, key=lambda k: agg[k]["total_ms"])
        a = agg[tag]
        sec = a["total_ms"] / 1e3
        # the binding roof is the one this family's own algorithmic flops / bytes take longer on
        mfma = tag.startswith("gemm") and a["flops"] / (mfma_peak_tfs * 1e12) > a["bytes"] / (hbm_peak_gbs * 1e9)
        if mfma:
            ach, peak, unit = a["flops"] / sec / 1e12, mfma_peak_tfs, "TFLOP/s"
        else:
            ach, peak, unit = a["bytes"] / sec / 1e9, hbm_peak_gbs, "GB/s"
        return {"kernel": tag, "bound": "mfma" if mfma else "hbm", "achieved": round(ach, 1), "peak": peak, "unit": unit,
                "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": a["launches"] / steps,
                "ms_per_step": round(a["total_ms"] / steps, 3),
                "avg_launch_us": round(1e3 * a["total_ms"] / a["launches"], 2),
                "algorithmic_bytes_per_launch": round(a["bytes"] / a["launches"]), "algorithmic_flops_per_launch": round(a["flops"] / a["launches"]),
                "achieved_GBs_algorithmic": round(a["bytes"] / sec / 1e9, 1)}


PROFILE = _Profile()
SIDE = _SideStream()


def _esz(t):
    return t.element_size()


_GEMM_PATHS = {0: "none", 1: "tiled", 2: "stream", 3: "dw", 4: "small", 5: "pipe"}


class LevelGeom:
    """Host-side multi-scale geometry: spatial shapes (H,W) per level and start offsets."""

    def __init__(self, shapes: Sequence[Sequence[int]]):
        self.shapes = [(int(h), int(w)) for h, w in shapes]
        self.L = len(self.shapes)
        starts, acc = [], 0
        for h, w in self.shapes:
            starts.append(acc)
            acc += h * w
        self.starts = starts
        self.S = acc
        self.c_shapes = _i64arr([v for hw in self.shapes for v in hw])
        self.c_starts = _i64arr(starts)
        self.c_segs = _i64arr(starts + [acc])

    def key(self):
        return tuple(self.shapes)


# ---- GEMM ------------------------------------------------------------------------------------------
def gemm(A: torch.Tensor, B: torch.Tensor, Cout: torch.Tensor, M: int, N: int, K: int, *, lda: int, ldb: int, ldc: int,
         a_kmajor=False, b_kmajor=False, bias=None, act=0, add_src=None, ld_add=0, gate_ref=None, gate_scale=1.0,
         row_mask=None, drop_p=0.0, seed=0, compute=None, batch=1, strideA=0, strideB=0, strideC=0, stride_bias=0,
         splitk=1, atomic=False, alpha=1.0, head_major=None, b_split=False, B_lo=None, seg=None, b_alt=None):
    """seg = (seg_sums fp32 [n_seg, >= M], starts (n_seg + 1 ints), period): per-segment column sums of A out of the weight-gradient
    form's own pass (PoetGemmDesc.seg_sums).  b_alt = (B_alt, ldb_alt, m_alt): rows m >= m_alt of C pair with B_alt (PoetGemmDesc.B_alt)."""
    lib = _lib.load()
    if not (A.is_cuda and B.is_cuda and Cout.is_cuda):
        raise _lib.PoetHipError("poet_amd: GEMM operands must live on the GPU (no CPU path exists)")
    ad, bd, cd = _DT[A.dtype], _DT[B.dtype], _DT[Cout.dtype]
    c_f16 = 0
    if cd == F16:           # an fp16 OUTPUT: 2-byte slots of a bf16-typed C written as IEEE half (PoetGemmDesc.c_f16)
        cd, c_f16 = BF16, 1
    if ad == F16 or bd == F16:
        raise TypeError("poet_amd: fp16 is a storage format of GEMM outputs only (offsets | logits), never an operand")
    if compute is None:     # bf16 MFMA as soon as any operand is stored in bf16; pure-fp32 calls use the f32 MFMA
        compute = BF16 if (ad | bd | cd) else F32
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be fp32")
    hm = head_major or (0, 0, 0)
    # caller-owned scratch: partial tiles of the weight-gradient form
    ws = _workspace(Cout.device).data_ptr() if (atomic and a_kmajor and b_kmajor and batch == 1 and K >= 4096) else 0
    d = _GEMM_DESC
    _GEMM_PACK.pack_into(_GEMM_BUF, 0, A.data_ptr(), 0, B.data_ptr(), Cout.data_ptr(),
                         0 if bias is None else bias.data_ptr(), 0 if add_src is None else add_src.data_ptr(),
                         0 if gate_ref is None else gate_ref.data_ptr(), 0 if row_mask is None else row_mask.data_ptr(),
                         M, N, K, lda, ldb, ldc, ld_add, int(a_kmajor), int(b_kmajor), ad, bd, cd, compute, batch,
                         strideA, strideB, strideC, stride_bias, splitk, int(atomic), act, alpha, gate_scale, drop_p,
                         seed & 0xFFFFFFFF, 1 if head_major is not None else 0, hm[0], hm[1], hm[2],
                         (_seed_dev() or 0) if drop_p > 0 else 0, int(b_split), c_f16, ws, _WORKSPACE_BYTES if ws else 0,
                         0 if B_lo is None else B_lo.data_ptr(),
                         *((0, 0, 0, 0) + (0,) * 10 if seg is None else
                           (seg[0].data_ptr(), seg[0].stride(0), len(seg[1]) - 1, int(seg[2])) + tuple(int(v) for v in seg[1]) + (0,) * (10 - len(seg[1]))),
                         *((0, 0, 0, 0) if b_alt is None else (b_alt[0].data_ptr(), int(b_alt[1]), int(b_alt[2]), 0)))
    if PROFILE.on:
        e0 = PROFILE.begin()
        _lib.check(lib.poet_gemm(C.byref(d), _stream()), "poet_gemm")
        # tag = the kernel family that ran (the symbol a rocprofv3 trace shows) x the GEMM's role
        path = _GEMM_PATHS[lib.poet_gemm_last_path()]
        role = "dW" if (a_kmajor and atomic) else ("dX" if b_kmajor else "fwd")
        nb = batch * (M * K * _esz(A) + M * N * _esz(Cout)) + N * K * _esz(B) * (batch if strideB else 1)
        for extra in (add_src, gate_ref):                     # epilogue operands the kernel also streams
            if extra is not None:
                nb += batch * M * N * _esz(extra)
        # one tag per kernel family, role AND shape: "the dominant kernel" of the roofline line is one kernel on one shape,
        # not 26 launches of five different GEMMs that happen to share a symbol
        # (the weight-gradient form has M = n_out, N = k_in, K = token rows)
        shape = f"_{M}x{N}" if (role == "dW" and K >= 4096) else (f"_{N}x{K}" if (role != "dW" and M >= 4096) else "")
        PROFILE.end(f"gemm_{path}_{role}{shape}", e0, 2.0 * M * N * K * batch, nb)
        return Cout
    _lib.check(lib.poet_gemm(C.byref(d), _stream()), "poet_gemm")
    return Cout


_W16 = os.environ.get("POET_W16", "0") not in ("", "0")


def linear_fwd(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], out: torch.Tensor, *, act=0,
               drop_p=0.0, seed=0, row_mask=None, add_src=None, head_major=None, ldc=None, ldx=None, split=False, W_lo=None):
    """out[rows, N] = act(x[rows, K] @ W[N, K]^T + b).  x / out may be column slices (ldx / ldc).
    split: W is the fp32 master, used as bf16 hi + bf16 lo (PoetGemmDesc.b_split); or, with W_lo, W and W_lo are the two
    bf16 images the optimiser maintains (PoetGemmDesc.B_lo: the long-K kernel, one pass over x)."""
    rows = x.numel() // x.shape[-1] if ldx is None else x.shape[0]
    N, K = W.shape
    bs = split or W_lo is not None
    if (_W16 and split and W_lo is None and K == 256 and rows >= 4096 and N % 128 == 0 and x.dtype == torch.bfloat16 and add_src is None
            and out.dtype in (torch.bfloat16, torch.float16)):
        bs = 2          # (experimental, POET_W16=1) one IEEE fp16 image of the master, one f16 MFMA per fragment pair: gemm_ws.hip WM = 2
    return gemm(x, W, out, rows, N, K, lda=ldx or K, ldb=K, ldc=ldc or N, bias=b, act=act, drop_p=drop_p, seed=seed,
                row_mask=row_mask, add_src=add_src, ld_add=(ldc or N), head_major=head_major,
                b_split=bs, B_lo=W_lo, compute=BF16 if (split or W_lo is not None) else None)


def linear_dx(dy: torch.Tensor, W: torch.Tensor, out: torch.Tensor, *, rows: int, ldy=None, add_src=None,
              gate_ref=None, gate_scale=1.0, ldc=None):
    """out[rows, K_in] = dy[rows, N_out] @ W[N_out, K_in]  (+add_src, gated)."""
    n_out, k_in = W.shape
    return gemm(dy, W, out, rows, k_in, n_out, lda=ldy or n_out, ldb=k_in, ldc=ldc or k_in, b_kmajor=True,
                add_src=add_src, ld_add=(ldc or k_in), gate_ref=gate_ref, gate_scale=gate_scale)


def linear_bwd_ok(dy, x, W16, dW, rows) -> bool:
    """Shapes poet_linear_bwd takes: a 256-wide output, bf16 operands, >= 8192 rows, n2 % 128 == 0 (POET_NO_LINEAR_BWD=1: off, A/B aid)."""
    if _NO_LINEAR_BWD:
        return False
    n2 = x.shape[1]
    return (dy.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and W16 is not None and W16.dtype == torch.bfloat16
            and dy.shape[1] == 256 and tuple(W16.shape) == (256, n2) and tuple(dW.shape) == (256, n2) and dW.dtype == torch.float32
            and rows >= 8192 and n2 % 128 == 0 and n2 <= 4096 and dy.stride(-1) == 1 and x.stride(-1) == 1 and W16.stride(-1) == 1 and dW.stride(-1) == 1
            and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and W16.stride(0) % 8 == 0
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and W16.data_ptr() % 16 == 0 and defer_small_dw.active is None)


_NO_LINEAR_BWD = os.environ.get("POET_NO_LINEAR_BWD", "0") not in ("", "0")


def linear_bwd(dy, x, W16, dW, db, dx_out, *, rows, gate=False, gate_scale=1.0):
    """dW += dy^T x, db += colsum(dy), dx_out = (dy W16) [gated by x > 0] in one launch pair (poet_linear_bwd)."""
    lib = _lib.load()
    n2 = x.shape[1]
    ws = _workspace(dW.device)
    if PROFILE.on:
        e0 = PROFILE.begin()
    _lib.check(lib.poet_linear_bwd(_req(dy, "dy").data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), W16.data_ptr(), W16.stride(0),
                                   dW.data_ptr(), dW.stride(0), _ptr(db), dx_out.data_ptr(), dx_out.stride(0), int(bool(gate)), float(gate_scale),
                                   rows, n2, ws.data_ptr(), _WORKSPACE_BYTES, _stream()), "poet_linear_bwd")
    if PROFILE.on:
        PROFILE.end(f"gemm_dwx_dW_dX_256x{n2}", e0, 4.0 * rows * 256 * n2, rows * (256 * 2 + n2 * 2 + n2 * 2))
    return dx_out


def _splitk_for(M, N, K):
    big = M >= 256 and N >= 128 and K >= 4096
    t = 128 if big else 64
    tiles = -(-M // t) * -(-N // t)
    return max(1, min(-(-1024 // tiles), -(-K // (512 if big else 128))))


_DW_MULTI = os.environ.get("POET_NO_DW_MULTI", "0") in ("", "0")       # (A/B aid, read at import)


class defer_small_dw:
    """`with ops.defer_small_dw() as D:` around the backward of a STACK of identical layers (the decoder): the weight / bias
    gradients of <= 1024-row fp32 Linears are recorded instead of launched (`D.next_layer()` before each layer), and
    `flush()` issues ONE launch per Linear for all layers (poet_gemm_dw_list) -- nothing reads a weight gradient before the
    optimiser, and the launches of these kernels cost more than their work.  Falls back to one launch per record when the
    layers did not record the same sequence of shapes.  Off while the per-kernel profiler runs."""
    active = None

    def __enter__(self):
        self.layers = []
        if not PROFILE.on and os.environ.get("POET_NO_DEFERRED_DW", "0") in ("", "0"):
            defer_small_dw.active = self
        return self

    def next_layer(self):
        self.layers.append([])

    def record(self, dy, x, dW, db, rows, ldy, ldx):
        self.layers[-1].append((dy, x, dW, db, rows, ldy, ldx))

    def flush(self):
        layers, self.layers = self.layers, []
        if not layers:
            return
        sig = lambda r: (tuple(r[2].shape), r[3] is None, r[4], r[5], r[6])
        uniform = len(layers) <= 8 and all(len(l) == len(layers[0]) and [sig(r) for r in l] == [sig(r) for r in layers[0]] for l in layers)
        if not uniform:
            for l in layers:
                for dy, x, dW, db, rows, ldy, ldx in l:
                    _linear_dw_now(dy, x, dW, rows, ldy, ldx, db)
            return
        lib = _lib.load()
        n = len(layers)
        nl = len(layers[0])
        # every Linear of every layer in ONE launch (poet_gemm_dw_multi: the lists are block ranges of one grid and overlap, where one
        # launch per Linear ran 7 latency chains of ~16 us one after the other): same rows everywhere, k_in % 4 == 0, 16-byte aligned x
        r0 = layers[0]
        if (_DW_MULTI and 1 < nl <= 8 and all(r[4] == r0[0][4] for r in r0) and all(r[2].shape[1] % 4 == 0 and (r[6] or r[2].shape[1]) % 4 == 0 for r in r0)
                and all(rec[1].data_ptr() % 16 == 0 for l in layers for rec in l)):
            flat = [l[j] for j in range(nl) for l in layers]              # list-major: entry j * n + i
            big = C.c_void_p * (nl * n)
            has_db = any(r[3] is not None for r in flat)
            rc = lib.poet_gemm_dw_multi(big(*[r[0].data_ptr() for r in flat]), big(*[r[1].data_ptr() for r in flat]), big(*[r[2].data_ptr() for r in flat]),
                                        big(*[(r[3].data_ptr() if r[3] is not None else None) for r in flat]) if has_db else None, nl, n,
                                        (C.c_int * nl)(*[r[2].shape[0] for r in r0]), (C.c_int * nl)(*[r[2].shape[1] for r in r0]), r0[0][4],
                                        (C.c_int64 * nl)(*[(r[5] or r[2].shape[0]) for r in r0]), (C.c_int64 * nl)(*[(r[6] or r[2].shape[1]) for r in r0]),
                                        _stream())
            if rc == 0:
                return
        arr = C.c_void_p * n
        for j in range(len(layers[0])):
            recs = [l[j] for l in layers]
            dy0, x0, dW0, db0, rows, ldy, ldx = recs[0]
            n_out, k_in = dW0.shape
            _lib.check(lib.poet_gemm_dw_list(arr(*[r[0].data_ptr() for r in recs]), arr(*[r[1].data_ptr() for r in recs]),
                                             arr(*[r[2].data_ptr() for r in recs]),
                                             None if db0 is None else arr(*[r[3].data_ptr() for r in recs]), n, n_out, k_in, rows,
                                             ldy or n_out, ldx or k_in, k_in, _stream()), "poet_gemm_dw_list")

    def __exit__(self, et, ev, tb):
        defer_small_dw.active = None
        if et is None:
            self.flush()
        return False


def _linear_dw_now(dy, x, dW, rows, ldy, ldx, db):
    n_out, k_in = dW.shape
    SIDE.run(lambda: gemm(dy, x, dW, n_out, k_in, rows, lda=ldy or n_out, ldb=ldx or k_in, ldc=k_in, a_kmajor=True,
                          b_kmajor=True, splitk=_splitk_for(n_out, k_in, rows), atomic=True, bias=db), dy, x)


def linear_dw(dy: torch.Tensor, x: torch.Tensor, dW: torch.Tensor, *, rows: int, ldy=None, ldx=None, db=None, seg=None, x_alt=None):
    """dW[N_out, K_in] += dy[rows, N_out]^T @ x[rows, K_in]   (fp32 atomics, split-K over rows);
    db[N_out] += column sums of dy (optional: the bias gradient, fused into the same pass where possible).
    seg = (seg_sums [n_seg, >= N_out] fp32, starts, period): seg_sums[s] += the column sums of dy over the rows of segment s of every
    period (the encoder's per-LEVEL sums of d(offsets | logits)), formed in the same pass over dy instead of by a colsum launch.
    x_alt = (x2, m_alt): the output rows >= m_alt of dW pair with x2 instead of x (two stacked Linears with different inputs whose
    gradient rows share one buffer: PoetGemmDesc.B_alt)."""
    D = defer_small_dw.active
    if (D is not None and D.layers and rows <= 1024 and dy.dtype == torch.float32 and x.dtype == torch.float32
            and dW.dtype == torch.float32 and dW.is_contiguous()):
        D.record(dy, x, dW, db, rows, ldy, ldx)
        return dW
    n_out, k_in = dW.shape
    if seg is not None and len(seg[1]) - 1 > 8:
        raise ValueError("linear_dw: seg_sums carries at most 8 segments (PoetGemmDesc.seg_start[10])")
    launch = lambda: gemm(dy, x, dW, n_out, k_in, rows, lda=ldy or n_out, ldb=ldx or k_in, ldc=k_in, a_kmajor=True,
                          b_kmajor=True, splitk=_splitk_for(n_out, k_in, rows), atomic=True, bias=db, seg=seg,
                          b_alt=None if x_alt is None else (x_alt[0], x_alt[0].stride(0), x_alt[1]))
    if seg is not None:
        # the per-segment sums are READ by the caller's next launches (colsum -> bias gradients, level_embed products): this launch
        # belongs on the caller's stream even when weight gradients otherwise fork (POET_SIDE_STREAM=1) -- ADVICE r5
        launch()
    else:
        SIDE.run(launch, dy, x, *(() if x_alt is None else (x_alt[0],)))
    return dW


# ---- MSDA ------------------------------------------------------------------------------------------
def msda_fwd(value, geom: LevelGeom, loc, attn, out):
    lib = _lib.load()
    N, S, M, D = value.shape
    Lq, P = loc.shape[1], loc.shape[4]
    _lib.check(lib.poet_msda_fwd(_req(value, "value").data_ptr(), geom.c_shapes, geom.c_starts, loc.data_ptr(),
                                 attn.data_ptr(), out.data_ptr(), N, S, M, D, geom.L, P, Lq, dcode(value), _stream()),
               "poet_msda_fwd")
    return out


def msda_bwd(value, geom: LevelGeom, loc, attn, grad_out, grad_value, grad_loc, grad_attn):
    lib = _lib.load()
    N, S, M, D = value.shape
    Lq, P = loc.shape[1], loc.shape[4]
    _lib.check(lib.poet_msda_bwd(_req(value, "value").data_ptr(), geom.c_shapes, geom.c_starts, loc.data_ptr(),
                                 attn.data_ptr(), grad_out.data_ptr(), grad_value.data_ptr(), grad_loc.data_ptr(),
                                 grad_attn.data_ptr(), N, S, M, D, geom.L, P, Lq, dcode(value), _stream()),
               "poet_msda_bwd")


def msda_fused_fwd(value, vstrides, geom: LevelGeom, offattn, ldq, logit_col, ref, ref_bs, out, N, M, D, P, Lq, grid_queries=False):
    lib = _lib.load()
    _lib.check(lib.poet_msda_fused_fwd(_req(value, "value").data_ptr(), *vstrides, geom.c_shapes, geom.c_starts,
                                       offattn.data_ptr(), ldq, logit_col, ref.data_ptr(), ref_bs, out.data_ptr(),
                                       N, geom.S, M, D, geom.L, P, Lq, dcode(value), dcode(offattn), int(grid_queries), _stream()),
               "poet_msda_fused_fwd")
    return out


def msda_fused_bwd(value, vstrides, geom: LevelGeom, offattn, ldq, logit_col, ref, ref_bs, grad_out, grad_value,
                   grad_offattn, N, M, D, P, Lq, grid_queries=False, parts=3, ld_grad=0, gv_strides=None):
    """gv_strides: (n, s, m) element strides of grad_value when it is not laid out like value."""
    lib = _lib.load()
    if PROFILE.on and not getattr(msda_fused_bwd, "_timing", False):       # time the two kernels of the backward separately
        a = (value, vstrides, geom, offattn, ldq, logit_col, ref, ref_bs, grad_out, grad_value, grad_offattn, N, M, D, P, Lq)
        nb_q = offattn.numel() * offattn.element_size() * 2 + grad_out.numel() * grad_out.element_size() + value.numel() * value.element_size()
        msda_fused_bwd._timing = True
        try:
            if parts in (2, 3, 0):
                e0 = PROFILE.begin(); msda_fused_bwd(*a, grid_queries=grid_queries, parts=2, ld_grad=ld_grad, gv_strides=gv_strides)
                PROFILE.end("msda_bwd_dvalue_scatter_tiled" if (grid_queries and offattn.dtype in (torch.bfloat16, torch.float16)) else "msda_bwd_dvalue_scatter", e0, 0.0, offattn.numel() * offattn.element_size() + grad_out.numel() * grad_out.element_size() + grad_value.numel() * grad_value.element_size())
            if parts in (1, 3, 0):
                e0 = PROFILE.begin(); msda_fused_bwd(*a, grid_queries=grid_queries, parts=1, ld_grad=ld_grad, gv_strides=gv_strides)
                PROFILE.end("msda_bwd_dq" + ("_small" if N * Lq < 4096 else ""), e0, 0.0, nb_q)
        finally:
            msda_fused_bwd._timing = False
        return
    _lib.check(lib.poet_msda_fused_bwd(_req(value, "value").data_ptr(), *vstrides, geom.c_shapes, geom.c_starts,
                                       offattn.data_ptr(), ldq, logit_col, ref.data_ptr(), ref_bs, grad_out.data_ptr(),
                                       grad_value.data_ptr(), grad_offattn.data_ptr(), ld_grad, N, geom.S, M, D, geom.L, P, Lq,
                                       dcode(value), dcode(offattn), dcode(grad_value), int(grid_queries), parts,
                                       None if gv_strides is None else (C.c_int64 * 3)(*gv_strides), _stream()), "poet_msda_fused_bwd")


# ---- norms -----------------------------------------------------------------------------------------
def ln_fwd(x, res, gamma, beta, y, z, mean, rstd, rows, d, eps=1e-5, drop_p=0.0, seed=0, y16=None, pos16=None, q16=None, res16=None):
    """res16: the bf16 head of a split input stream (res = its fp16 remainder); an fp16 y is the remainder against y16."""
    lib = _lib.load()
    _lib.check(lib.poet_ln_fwd(_req(x, "x").data_ptr(), _ptr(res), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(z),
                               _ptr(mean), _ptr(rstd), rows, d, eps, drop_p, seed & 0xFFFFFFFF, dcode(x), dcode(y) if res is None else dcode(res), dcode(y),
                               -1 if z is None else dcode(z), _ptr(y16), _ptr(pos16), _ptr(q16), _ptr(res16),
                               _seed_dev() if drop_p > 0 else None, _stream()), "poet_ln_fwd")
    return y


def ln_bwd(dy, z, mean, rstd, gamma, dz, dx, dgamma, dbeta, rows, d, drop_p=0.0, seed=0):
    lib = _lib.load()
    _lib.check(lib.poet_ln_bwd(_req(dy, "dy").data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                               dz.data_ptr(), _ptr(dx), dgamma.data_ptr(), dbeta.data_ptr(), rows, d, drop_p,
                               seed & 0xFFFFFFFF, dcode(z), dcode(dy), dcode(dz), _seed_dev() if drop_p > 0 else None, _stream()), "poet_ln_bwd")


def groupnorm_fwd(x, gamma, beta, y, stats, N, HW, Cc, G, x_off, x_stride, y_off, y_stride, eps=1e-5, x16=None):
    """x16: optional bf16 copy of an fp32 x, same layout, written in the same pass."""
    lib = _lib.load()
    _lib.check(lib.poet_groupnorm_fwd(_req(x, "x").data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                      N, HW, Cc, G, x_off, x_stride, y_off, y_stride, eps, dcode(x), dcode(y),
                                      _workspace(x.device).data_ptr(), _WORKSPACE_BYTES // 4, _ptr(x16), _stream()),
               "poet_groupnorm_fwd")


def groupnorm_bwd(dy, x, stats, gamma, dx, dgamma, dbeta, N, HW, Cc, G, x_off, x_stride, y_off, y_stride):
    lib = _lib.load()
    _lib.check(lib.poet_groupnorm_bwd(_req(dy, "dy").data_ptr(), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                                      dgamma.data_ptr(), dbeta.data_ptr(), N, HW, Cc, G, x_off, x_stride, y_off, y_stride,
                                      dcode(x), dcode(dy), _workspace(x.device).data_ptr(), _WORKSPACE_BYTES // 4, _stream()),
               "poet_groupnorm_bwd")


# ---- attention -------------------------------------------------------------------------------------
def mha_fwd(q, k, v, ld, out, ld_out, N, Q, M, hd, drop_p=0.0, seed=0):
    lib = _lib.load()
    _lib.check(lib.poet_mha_fwd(_req(q, "q").data_ptr(), k.data_ptr(), v.data_ptr(), ld, out.data_ptr(), ld_out, N, Q, M, hd,
                                drop_p, seed & 0xFFFFFFFF, _seed_dev() if drop_p > 0 else None, _stream()), "poet_mha_fwd")


def mha_bwd(q, k, v, ld, dout, ld_out, dq, dk, dv, ld_d, N, Q, M, hd, drop_p=0.0, seed=0):
    lib = _lib.load()
    _lib.check(lib.poet_mha_bwd(_req(q, "q").data_ptr(), k.data_ptr(), v.data_ptr(), ld, dout.data_ptr(), ld_out, dq.data_ptr(),
                                dk.data_ptr(), dv.data_ptr(), ld_d, N, Q, M, hd, drop_p, seed & 0xFFFFFFFF,
                                _seed_dev() if drop_p > 0 else None, _stream()), "poet_mha_bwd")


# ---- encodings / geometry --------------------------------------------------------------------------
_DIM_T = {}


def sine_dim_t(F: int, device, temperature: float = 10000.0) -> torch.Tensor:
    """Constant table temperature^(2*(i//2)/F), computed once on the host in fp32 exactly as
    models/position_encoding.py:52-53 does, then kept on the device."""
    key = (F, str(device), temperature)
    if key not in _DIM_T:
        i = torch.arange(F, dtype=torch.float32)
        _DIM_T[key] = (temperature ** (2 * (i // 2) / F)).to(device)
    return _DIM_T[key]


def pos_sine(mask_u8, out, level_embed, N, H, W, F, tok_off, tok_stride):
    lib = _lib.load()
    dim_t = sine_dim_t(F, out.device)
    _lib.check(lib.poet_pos_sine(_req(mask_u8, "mask").data_ptr(), out.data_ptr(), _ptr(level_embed), dim_t.data_ptr(), N, H, W,
                                 F, tok_off, tok_stride, dcode(out), _stream()), "poet_pos_sine")


def bbox_sine(boxes, out, n, F, valid=None, fill=-10.0):
    lib = _lib.load()
    _lib.check(lib.poet_bbox_sine(_req(boxes, "boxes").data_ptr(), _ptr(valid), out.data_ptr(), n, F, fill, _stream()),
               "poet_bbox_sine")


def dec_ref_points(ref, valid_ratios, out, N, Q, L):
    lib = _lib.load()
    _lib.check(lib.poet_dec_ref_points(_req(ref, "ref").data_ptr(), valid_ratios.data_ptr(), out.data_ptr(), N, Q, L, _stream()),
               "poet_dec_ref_points")


def valid_ratio(mask_u8, out, out_stride, N, H, W):
    lib = _lib.load()
    _lib.check(lib.poet_valid_ratio(_req(mask_u8, "mask").data_ptr(), out.data_ptr(), out_stride, N, H, W, _stream()),
               "poet_valid_ratio")


def mask_nearest(src, dst, N, H, W, Ho, Wo):
    lib = _lib.load()
    _lib.check(lib.poet_mask_nearest(_req(src, "mask").data_ptr(), dst.data_ptr(), N, H, W, Ho, Wo, _stream()), "poet_mask_nearest")


def add_rowvec(x, vec, batch, batch_stride_rows, row0, rows, cols):
    lib = _lib.load()
    _lib.check(lib.poet_add_rowvec(_req(x, "x").data_ptr(), vec.data_ptr(), batch, batch_stride_rows, row0, rows, cols, dcode(x),
                                   _stream()), "poet_add_rowvec")


def enc_ref_points(valid_ratios, geom: LevelGeom, ref, N):
    lib = _lib.load()
    _lib.check(lib.poet_enc_ref_points(_req(valid_ratios, "valid_ratios").data_ptr(), geom.c_shapes, ref.data_ptr(), N, geom.L,
                                       geom.S, _stream()), "poet_enc_ref_points")


# ---- elementwise / layout --------------------------------------------------------------------------
def add(a, b, out):
    lib = _lib.load()
    _lib.check(lib.poet_add(_req(a, "a").data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), dcode(a), dcode(b), dcode(out), _stream()),
               "poet_add")
    return out


def gelu_fwd(x, y, drop_p=0.0, seed=0):
    """y = dropout(gelu(x)) (erf form); x in y's dtype or fp32."""
    lib = _lib.load()
    _lib.check(lib.poet_gelu_fwd(_req(x, "x").data_ptr(), y.data_ptr(), x.numel(), dcode(x), dcode(y), drop_p, seed & 0xFFFFFFFF,
                                 _seed_dev() if drop_p > 0 else None, _stream()), "poet_gelu_fwd")
    return y


def gelu_bwd(dy, x, dx, drop_p=0.0, seed=0):
    """dx = dy * mask / (1 - p) * gelu'(x): the mask is redrawn from (seed, element index)."""
    lib = _lib.load()
    _lib.check(lib.poet_gelu_bwd(_req(dy, "dy").data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), dcode(x), dcode(dy), drop_p, seed & 0xFFFFFFFF,
                                 _seed_dev() if drop_p > 0 else None, _stream()), "poet_gelu_bwd")
    return dx


def add_cast(a, b16, sum16, a16):
    """a16 = bf16(a), sum16 = bf16(a + b16) in one pass (a fp32; the rest bf16, contiguous)."""
    lib = _lib.load()
    _lib.check(lib.poet_add_cast(_req(a, "a").data_ptr(), b16.data_ptr(), sum16.data_ptr(), a16.data_ptr(), a.numel(), _stream()), "poet_add_cast")
    return sum16, a16


def cast(src, dst):
    lib = _lib.load()
    _lib.check(lib.poet_cast(_req(src, "src").data_ptr(), dst.data_ptr(), src.numel(), dcode(src), dcode(dst), _stream()), "poet_cast")
    return dst


def zero_(t: torch.Tensor):
    """t[...] = 0 through the library's own fill (poet_zero): contiguous tensors whose byte size is a multiple of 4."""
    nb = t.numel() * t.element_size()
    if not t.is_cuda:
        raise _lib.PoetHipError("poet_amd: zero_ needs a GPU tensor (no CPU path exists)")
    if not t.is_contiguous() or nb % 4 or t.data_ptr() % 4:
        return t.zero_()
    if nb:
        _lib.check(_lib.load().poet_zero(t.data_ptr(), nb, _stream()), "poet_zero")
    return t


def zeros(shape, dtype, device):
    """torch.zeros through poet_zero (the step's accumulators: value-gradient maps, per-level sums)."""
    return zero_(torch.empty(shape, dtype=dtype, device=device))


def colsum(x, ld, out, batch, rows_per_batch, cols, segs=None, nseg=1):
    lib = _lib.load()
    _lib.check(lib.poet_colsum(_req(x, "x").data_ptr(), ld, out.data_ptr(), batch, rows_per_batch, cols, segs, nseg, dcode(x),
                               _stream()), "poet_colsum")


def pipe_split() -> bool:
    """long-K split-weight products on the deep-pipeline kernel with the arena's two bf16 weight images (POET_GEMM_NO_PIPE=1: the
    K-chunked weight-stationary kernel on the fp32 master)."""
    return os.environ.get("POET_GEMM_NO_PIPE", "0") in ("", "0") and os.environ.get("POET_NO_PIPE_SPLIT", "0") in ("", "0")


def oa_f16(M, D, L, P, grid_queries) -> bool:
    """fp16 storage of the encoder's sampling offsets | attention logits (8x the resolution of bf16 at the same 2 bytes: the
    offsets are pixel distances of a few units, bf16 resolves them to 1/32 px): where the kernels that read them as fp16 run --
    the shared-geometry gathers and the LDS-tiled scatter at the encoder shape -- and POET_OA_BF16=1 does not ask for bf16."""
    env = os.environ.get
    return bool(grid_queries and M == 16 and D == 16 and L == 4 and P == 4 and env("POET_OA_BF16", "0") in ("", "0")
                and env("POET_MSDA_NO_SHARED", "0") in ("", "0") and env("POET_NO_TILED_SCATTER", "0") in ("", "0")
                and env("POET_WIN_GATHER", "0") in ("", "0"))


def v_f16(M, D, L, P, grid_queries) -> bool:
    """fp16 storage of the encoder's value maps (round 6): wherever the offsets | logits are fp16 (oa_f16: the shared-geometry
    gathers are then the only readers of the maps) and POET_V_BF16=1 does not ask for bf16 (A/B aid)."""
    return oa_f16(M, D, L, P, grid_queries) and os.environ.get("POET_V_BF16", "0") in ("", "0")


def ln_f16(rows, d) -> bool:
    """fp16 storage of the projection output that a LayerNorm launch consumes (the encoder's two per layer): POET_LN_F32=1 keeps the
    fp32 accumulators (A/B aid)."""
    return rows >= 4096 and d == 256 and os.environ.get("POET_LN_F32", "0") in ("", "0")


def tiled_scatter_bf16() -> bool:
    """bf16 value-gradient maps out of the encoder's LDS-tiled scatter (POET_DV_FP32=1 or POET_NO_TILED_SCATTER=1: fp32)."""
    return os.environ.get("POET_DV_FP32", "0") in ("", "0") and os.environ.get("POET_NO_TILED_SCATTER", "0") in ("", "0")


def vgrad_to_rows(gv, vstrides, row_mask, out, N, S, M, D, ld_out=0):
    lib = _lib.load()
    _lib.check(lib.poet_vgrad_to_rows(_req(gv, "gv").data_ptr(), *vstrides, _ptr(row_mask), out.data_ptr(), ld_out, N, S, M, D, dcode(gv), dcode(out),
                                      _stream()), "poet_vgrad_to_rows")


def zero_masked_rows(x, row_mask, rows, cols):
    """x (rows, cols) with row stride x.stride(0): rows whose mask byte is set become zero."""
    lib = _lib.load()
    _lib.check(lib.poet_zero_masked_rows(_req(x, "x").data_ptr(), x.stride(0), row_mask.data_ptr(), rows, cols, dcode(x), _stream()),
               "poet_zero_masked_rows")


def nchw_to_tokens(src, dst, N, Cc, HW, tok_off, tok_stride):
    lib = _lib.load()
    _lib.check(lib.poet_nchw_to_tokens(_req(src, "src").data_ptr(), dst.data_ptr(), N, Cc, HW, tok_off, tok_stride, dcode(src),
                                       dcode(dst), _stream()), "poet_nchw_to_tokens")


def nchw_to_tokens_split(src, dst, N, Cc, HW):
    """src (N, C, HW) fp32 -> dst (N*HW, 2C) bf16 rows [hi | lo] (hi = bf16(x), lo = bf16(x - hi))."""
    lib = _lib.load()
    if src.dtype != torch.float32 or dst.dtype != torch.bfloat16:
        raise TypeError("nchw_to_tokens_split: fp32 map -> bf16 [hi | lo] rows")
    _lib.check(lib.poet_nchw_to_tokens_split(_req(src, "src").data_ptr(), dst.data_ptr(), N, Cc, HW, _stream()), "poet_nchw_to_tokens_split")


def split_rows(src, dst):
    """src (rows, K) fp32 -> dst (rows, 2K) bf16 rows [hi | lo]."""
    lib = _lib.load()
    if src.dtype != torch.float32 or dst.dtype != torch.bfloat16 or dst.shape[-1] != 2 * src.shape[-1]:
        raise TypeError("split_rows: fp32 (rows, K) -> bf16 (rows, 2K)")
    _lib.check(lib.poet_split_rows(_req(src, "src").data_ptr(), dst.data_ptr(), src.numel() // src.shape[-1], src.shape[-1], _stream()), "poet_split_rows")


def inproj_exact() -> bool:
    """bf16 policy: the input projection reads the backbone's fp32 maps as bf16 hi + lo (16 significant bits) and keeps its
    output in fp32 up to the GroupNorm (POET_INPROJ_BF16=1: single bf16 operand + bf16 output, the round-3 form)."""
    return os.environ.get("POET_INPROJ_BF16", "0") in ("", "0")


def tokens_to_nchw(src, dst, N, Cc, HW, tok_off, tok_stride):
    lib = _lib.load()
    _lib.check(lib.poet_tokens_to_nchw(_req(src, "src").data_ptr(), dst.data_ptr(), N, Cc, HW, tok_off, tok_stride, dcode(src),
                                       dcode(dst), _stream()), "poet_tokens_to_nchw")


def im2col3x3s2(src, dst, N, Cc, H, W, Ho, Wo):
    lib = _lib.load()
    _lib.check(lib.poet_im2col3x3s2(_req(src, "src").data_ptr(), dst.data_ptr(), N, Cc, H, W, Ho, Wo, dcode(src), dcode(dst),
                                    _stream()), "poet_im2col3x3s2")


def col2im3x3s2_add(dcol, dst, N, Cc, H, W, Ho, Wo, tok_off, tok_stride):
    """dst[n, tok_off + y W + x, c] += the entries of dcol (N Ho Wo, 9 C) that im2col3x3s2 read from input pixel (n, c, y, x)."""
    lib = _lib.load()
    _lib.check(lib.poet_col2im3x3s2_add(_req(dcol, "dcol").data_ptr(), _req(dst, "dst").data_ptr(), N, Cc, H, W, Ho, Wo, tok_off, tok_stride,
                                        dcode(dcol), dcode(dst), _stream()), "poet_col2im3x3s2_add")


def pose_finish_fwd(rot_all, trans_all, cls, rot, trans, R, ncls):
    lib = _lib.load()
    _lib.check(lib.poet_pose_finish_fwd(_req(rot_all, "rot_all").data_ptr(), trans_all.data_ptr(), cls.data_ptr(), rot.data_ptr(),
                                        trans.data_ptr(), R, ncls, _stream()), "poet_pose_finish_fwd")


def pose_loss(trans, rot, qi, tt, tr, n_obj, losses, gt, gr, n_obj_dev=None, weights=None, total=None):
    """trans (L,NQ,3), rot (L,NQ,3,3) fp32 contiguous; qi int64 (n_obj), tt (n_obj,3), tr (n_obj,3,3) fp32.  n_obj_dev (int32
    device word): the count is read on the device (graph capture; qi / tt / tr then have capacity NQ).  weights (L,2) fp32: the
    gradients leave multiplied by their loss weights; total (1,) fp32: the weighted sum of the losses."""
    if weights is not None and (weights.dtype != torch.float32 or not weights.is_contiguous() or tuple(weights.shape) != (trans.shape[0], 2)):
        raise ValueError("pose_loss: weights must be a contiguous fp32 (L, 2) tensor")
    lib = _lib.load()
    L, NQ = trans.shape[0], trans.numel() // (trans.shape[0] * 3)
    _lib.check(lib.poet_pose_loss(_req(trans, "trans").data_ptr(), rot.data_ptr(), _ptr(qi) or 0, _ptr(tt) or 0, _ptr(tr) or 0, n_obj, L, NQ,
                                  losses.data_ptr(), gt.data_ptr(), gr.data_ptr(), _ptr(n_obj_dev), _ptr(weights), _ptr(total), _stream()), "poet_pose_loss")
    return losses


def lsa_boxes(pred_boxes, tgt_boxes, tgt_off, n_pred, col_out, status, cost_bbox=1.0):
    """pred_boxes (N,Q,4) f32, tgt_boxes (T,4) f32, tgt_off (N+1) i32, n_pred (N) i32 -> col_out (N,Q) i32 (device tensors)."""
    lib = _lib.load()
    N, Q = pred_boxes.shape[:2]
    _lib.check(lib.poet_lsa_boxes(_req(pred_boxes, "pred_boxes").data_ptr(), _req(tgt_boxes, "tgt_boxes").data_ptr(), tgt_off.data_ptr(),
                                  n_pred.data_ptr(), float(cost_bbox), N, Q, col_out.data_ptr(), status.data_ptr(), _stream()), "poet_lsa_boxes")
    return col_out


def match_gather(col, tgt_off, tgt_pos, tgt_rot, qi, tt, tr, n_out=None):
    lib = _lib.load()
    N, Q = col.shape
    _lib.check(lib.poet_match_gather(_req(col, "col").data_ptr(), tgt_off.data_ptr(), tgt_pos.data_ptr(), tgt_rot.data_ptr(), N, Q,
                                     qi.data_ptr(), tt.data_ptr(), tr.data_ptr(), _ptr(n_out), _stream()), "poet_match_gather")


def pose_finish_bwd(rot_all, cls, drot, dtrans, drot_all, dtrans_all, R, ncls):
    lib = _lib.load()
    _lib.check(lib.poet_pose_finish_bwd(_req(rot_all, "rot_all").data_ptr(), cls.data_ptr(), drot.data_ptr(), dtrans.data_ptr(),
                                        drot_all.data_ptr(), dtrans_all.data_ptr(), R, ncls, _stream()), "poet_pose_finish_bwd")


def sqnorm(g, out):
    """out[0] += sum(g^2); out needs 1 + 1024 floats (poet_sqnorm keeps its per-workgroup partials in out[1:])."""
    if out.numel() < 1025:
        raise ValueError("sqnorm: `out` must hold 1 + 1024 floats")
    lib = _lib.load()
    _lib.check(lib.poet_sqnorm(_req(g, "g").data_ptr(), g.numel(), out.data_ptr(), _stream()), "poet_sqnorm")


def adamw(p, g, m, v, n, lr, beta1, beta2, eps, wd, step, sqnorm_buf=None, max_norm=0.0, grad_scale=1.0, p_bf16=None, step_dev=None,
          lr_scale=None, p_bf16_lo=None):
    lib = _lib.load()
    _lib.check(lib.poet_adamw(_req(p, "p").data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_bf16), _ptr(p_bf16_lo), n, lr, beta1, beta2,
                              eps, wd, step, _ptr(sqnorm_buf), max_norm, grad_scale, _ptr(step_dev), _ptr(lr_scale), _stream()), "poet_adamw")


def counter_add(word, delta=1):
    lib = _lib.load()
    _lib.check(lib.poet_counter_add(_req(word, "word").data_ptr(), delta, _stream()), "poet_counter_add")


# ---- optional timing of the non-GEMM kernels (algorithmic bytes = every tensor argument touched once) -------------
def _instrument(name, tag):
    fn = globals()[name]

    def wrapped(*a, **k):
        if not PROFILE.on:
            return fn(*a, **k)
        e0 = PROFILE.begin()
        r = fn(*a, **k)
        nb = sum(t.numel() * t.element_size() for t in list(a) + list(k.values()) if torch.is_tensor(t))
        # the 320-row decoder launches of a kernel are a different regime (launch latency) from its 102k-row encoder launches:
        # separate families, so that an average launch time means something
        small = (a[9] * a[13] < 4096) if name == "msda_fused_fwd" else nb < (1 << 22)
        PROFILE.end(tag + ("_small" if small else ""), e0, 0.0, nb)
        return r

    wrapped.__name__ = name
    globals()[name] = wrapped


for _n, _t in [("msda_fused_fwd", "msda_fused_fwd"), ("ln_fwd", "ln_fwd"), ("ln_bwd", "ln_bwd"),
               ("colsum", "colsum"), ("vgrad_to_rows", "vgrad_to_rows"), ("add", "add"), ("cast", "cast"),
               ("groupnorm_fwd", "groupnorm"), ("groupnorm_bwd", "groupnorm"), ("mha_fwd", "mha"), ("mha_bwd", "mha"),
               ("pos_sine", "pos_sine"), ("adamw", "adamw"), ("sqnorm", "sqnorm")]:
    _instrument(_n, _t)
