// Shared device helpers for the PoET gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/poet_hip.h"

namespace poet {

typedef uint16_t bf16_t;                                   // raw bfloat16 bits
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- error plumbing (host) ---------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define POET_CHECK(cond, code, ...)      \
    do {                                 \
        if (!(cond)) {                   \
            poet::set_error(__VA_ARGS__); \
            return (code);               \
        }                                \
    } while (0)
#define POET_LAUNCH_CHECK()                                                              \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            poet::set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return POET_ERR_LAUNCH;                                                      \
        }                                                                                \
    } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch) ---------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one VALU op per PAIR instead of ~5 per value
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// fp32 pair -> packed IEEE fp16 (v_cvt_pk_f16_f32, round-to-nearest-even) and back
typedef _Float16 f16x2h_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2h_t));
}
__device__ __forceinline__ float h_lo(uint32_t v) { return (float)__builtin_bit_cast(f16x2h_t, v)[0]; }
__device__ __forceinline__ float h_hi(uint32_t v) { return (float)__builtin_bit_cast(f16x2h_t, v)[1]; }
// 8 consecutive 2-byte outputs as fp16 (PoetGemmDesc.c_f16) -- the bf16 form is vec<bf16_t, 8>::st
__device__ __forceinline__ void st8_f16(uint16_t* p, const float* o) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_h2(o[0], o[1]), pack_h2(o[2], o[3]), pack_h2(o[4], o[5]), pack_h2(o[6], o[7]));
}

// IEEE fp16 STORAGE type of GEMM outputs (PoetGemmDesc.c_f16): a distinct type, so kernels specialised on their output
// type carry either conversion and no run-time switch
struct f16_t { uint16_t v; };

template <typename T> struct io;
template <> struct io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// load/store N (4 or 8) consecutive elements as floats; p must be aligned to N*sizeof(T)
template <typename T, int N> struct vec;
template <> struct vec<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float* o) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float* o) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct vec<float, 8> {
    static __device__ __forceinline__ void ld(const float* p, float* o) {
        vec<float, 4>::ld(p, o);
        vec<float, 4>::ld(p + 4, o + 4);
    }
    static __device__ __forceinline__ void st(float* p, const float* o) {
        vec<float, 4>::st(p, o);
        vec<float, 4>::st(p + 4, o + 4);
    }
};
template <> struct vec<bf16_t, 4> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* o) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
    }
};
template <> struct vec<bf16_t, 8> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* o) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]),
                                                  pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    }
};

template <> struct io<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return h_lo((uint32_t)p->v); }
    static __device__ __forceinline__ void st(f16_t* p, float v) { p->v = (uint16_t)(pack_h2(v, 0.f) & 0xffffu); }
};
template <> struct vec<f16_t, 4> {
    static __device__ __forceinline__ void ld(const f16_t* p, float* o) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = h_lo(v.x); o[1] = h_hi(v.x); o[2] = h_lo(v.y); o[3] = h_hi(v.y);
    }
    static __device__ __forceinline__ void st(f16_t* p, const float* o) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_h2(o[0], o[1]), pack_h2(o[2], o[3])); }
};
template <> struct vec<f16_t, 8> {
    static __device__ __forceinline__ void ld(const f16_t* p, float* o) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = h_lo(v.x); o[1] = h_hi(v.x); o[2] = h_lo(v.y); o[3] = h_hi(v.y);
        o[4] = h_lo(v.z); o[5] = h_hi(v.z); o[6] = h_lo(v.w); o[7] = h_hi(v.w);
    }
    static __device__ __forceinline__ void st(f16_t* p, const float* o) { st8_f16(reinterpret_cast<uint16_t*>(p), o); }
};

// max(x, 0) as ONE v_max_f32 (through fmaxf hipcc first canonicalises its operand: two instructions per element in epilogues that
// are issue-bound); a NaN input gives 0 like fmaxf
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// ---- counter-based dropout RNG -------------------------------------------------------------------
// keep(seed, idx): one lowbias32 hash of (idx>>1 ^ seed-mix) serves the element PAIR (idx & ~1, idx | 1), 16 bits each
// (p is quantised to 2^-16); regenerated in backward from the same (seed, idx), so no mask is ever stored.  Vectorised
// epilogues call drop_pair() once per two consecutive elements.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_pair(uint32_t seed, uint32_t pair) { return hash32(pair ^ (seed * 0x9E3779B9u)); }
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t idx, uint32_t thresh16) {
    return ((drop_pair(seed, idx >> 1) >> ((idx & 1u) * 16u)) & 0xffffu) >= thresh16;
}
__host__ __device__ __forceinline__ uint32_t drop_thresh(float p) { return (uint32_t)(p * 65536.0f); }

// ---- wave-level all-reduce (64 lanes) without LDS traffic ---------------------------------------
// 4 cyclic DPP row rotations reduce inside each 16-lane row, then the four row totals are read with v_readlane and
// combined: ~12 VALU ops and no ds_bpermute (6 dependent LDS round trips per reduction with __shfl_xor).
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_row_ror<1>(v); v += dpp_row_ror<2>(v); v += dpp_row_ror<4>(v); v += dpp_row_ror<8>(v);
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_row_ror<1>(v)); v = fmaxf(v, dpp_row_ror<2>(v)); v = fmaxf(v, dpp_row_ror<4>(v)); v = fmaxf(v, dpp_row_ror<8>(v));
    const int i = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): set it once per device a call site launches on --
// `done` is the call site's own bitmask of device ordinals (a process-wide "already set" flag left the second GPU of a
// single-process multi-GPU run without the attribute; idempotent, so a race between host threads is benign: ADVICE r5)
static inline void lds_attr_once(const void* kern, int bytes, unsigned long long& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done |= bit;
}

}  // namespace poet
