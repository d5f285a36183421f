"""ctypes binding of libpoet_hip.so (include/poet_hip.h).  Fails loudly when the library is
missing -- there is no CPU or eager-PyTorch fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpoet_hip.so")
ABI_VERSION = 6                                # POET_ABI_VERSION of include/poet_hip.h this binding was written against

F32, BF16 = 0, 1
vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
pi64 = C.POINTER(C.c_int64)


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("A2", vp), ("B", vp), ("C", vp), ("bias", vp), ("add_src", vp), ("gate_ref", vp), ("row_mask", vp),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", i64), ("ldb", i64), ("ldc", i64), ("ld_add", i64),
        ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32),
        ("a_dtype", C.c_int32), ("b_dtype", C.c_int32), ("c_dtype", C.c_int32), ("compute", C.c_int32),
        ("batch", C.c_int32),
        ("strideA", i64), ("strideB", i64), ("strideC", i64), ("stride_bias", i64),
        ("splitk", C.c_int32), ("atomic", C.c_int32), ("act", C.c_int32),
        ("alpha", f32), ("gate_scale", f32), ("drop_p", f32), ("seed", u32),
        ("out_mode", C.c_int32), ("hm_M", C.c_int32), ("hm_S", C.c_int32), ("hm_D", C.c_int32),
        ("seed_dev", vp), ("b_split", C.c_int32), ("c_f16", C.c_int32), ("workspace", vp), ("workspace_bytes", i64),
        ("B_lo", vp),
        ("seg_sums", vp), ("ld_seg", i64), ("seg_n", C.c_int32), ("seg_period", C.c_int32), ("seg_start", C.c_int32 * 10),
        ("B_alt", vp), ("ldb_alt", i64), ("m_alt", C.c_int32), ("reserved_alt", C.c_int32),
    ]


_PROTOS = {
    "poet_hip_version": ([], i32),
    "poet_hip_last_error": ([], C.c_char_p),
    "poet_gemm": ([C.POINTER(GemmDesc), vp], i32),
    "poet_gemm_dw_list": ([C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, i64, i64, i64, vp], i32),
    "poet_gemm_dw_multi": ([C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, C.POINTER(i32), C.POINTER(i32), i32,
                            C.POINTER(i64), C.POINTER(i64), vp], i32),
    "poet_gemm_last_path": ([], i32),
    "poet_linear_bwd": ([vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i64, i32, f32, i64, i32, vp, i64, vp], i32),
    "poet_msda_fwd": ([vp, pi64, pi64, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "poet_msda_bwd": ([vp, pi64, pi64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "poet_msda_fused_fwd": ([vp, i64, i64, i64, pi64, pi64, vp, i64, i32, vp, i64, vp,
                             i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "poet_msda_fused_bwd": ([vp, i64, i64, i64, pi64, pi64, vp, i64, i32, vp, i64, vp, vp, vp, i64,
                             i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, pi64, vp], i32),
    "poet_ln_fwd": ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, u32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp], i32),
    "poet_ln_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, u32, i32, i32, i32, vp, vp], i32),
    "poet_mha_fwd": ([vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, f32, u32, vp, vp], i32),
    "poet_mha_bwd": ([vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, f32, u32, vp, vp], i32),
    "poet_pos_sine": ([vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, vp], i32),
    "poet_bbox_sine": ([vp, vp, vp, i32, i32, f32, vp], i32),
    "poet_dec_ref_points": ([vp, vp, vp, i32, i32, i32, vp], i32),
    "poet_valid_ratio": ([vp, vp, i64, i32, i32, i32, vp], i32),
    "poet_mask_nearest": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "poet_add_rowvec": ([vp, vp, i32, i64, i64, i64, i32, i32, vp], i32),
    "poet_enc_ref_points": ([vp, pi64, vp, i32, i32, i32, vp], i32),
    "poet_add": ([vp, vp, vp, i64, i32, i32, i32, vp], i32),
    "poet_cast": ([vp, vp, i64, i32, i32, vp], i32),
    "poet_add_cast": ([vp, vp, vp, vp, i64, vp], i32),
    "poet_gelu_fwd": ([vp, vp, i64, i32, i32, f32, u32, vp, vp], i32),
    "poet_gelu_bwd": ([vp, vp, vp, i64, i32, i32, f32, u32, vp, vp], i32),
    "poet_zero": ([vp, i64, vp], i32),
    "poet_colsum": ([vp, i64, vp, i32, i64, i32, pi64, i32, i32, vp], i32),
    "poet_vgrad_to_rows": ([vp, i64, i64, i64, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp], i32),
    "poet_zero_masked_rows": ([vp, i64, vp, i64, i32, i32, vp], i32),
    "poet_nchw_to_tokens": ([vp, vp, i32, i32, i32, i64, i64, i32, i32, vp], i32),
    "poet_nchw_to_tokens_split": ([vp, vp, i32, i32, i32, vp], i32),
    "poet_split_rows": ([vp, vp, i64, i32, vp], i32),
    "poet_tokens_to_nchw": ([vp, vp, i32, i32, i32, i64, i64, i32, i32, vp], i32),
    "poet_im2col3x3s2": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "poet_col2im3x3s2_add": ([vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i32, i32, vp], i32),
    "poet_groupnorm_fwd": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, f32, i32, i32, vp, i64, vp, vp], i32),
    "poet_groupnorm_bwd": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, i32, i32, vp, i64, vp], i32),
    "poet_pose_finish_fwd": ([vp, vp, vp, vp, vp, i32, i32, vp], i32),
    "poet_pose_finish_bwd": ([vp, vp, vp, vp, vp, vp, i32, i32, vp], i32),
    "poet_pose_loss": ([vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp], i32),
    "poet_lsa_boxes": ([vp, vp, vp, vp, f32, i32, i32, vp, vp, vp], i32),
    "poet_match_gather": ([vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp], i32),
    "poet_sqnorm": ([vp, i64, vp, vp], i32),
    "poet_adamw": ([vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, f32, f32, vp, vp, vp], i32),
    "poet_counter_add": ([vp, u32, vp], i32),
}

EXPORTS = tuple(_PROTOS)
_lib = None


class PoetHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PoetHipError(
            f"{LIB_PATH} not found: build it with `python -m poet_amd.build` (hipcc, gfx950). "
            "poet_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _PROTOS.items():
        fn = getattr(lib, name)            # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = res
    if lib.poet_hip_version() != ABI_VERSION:
        raise PoetHipError("libpoet_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().poet_hip_last_error()
        raise PoetHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
