// Multi-scale deformable attention core for gfx950 (the op behind the reference's external
// `deformable_attention.MSDeformAttn`, models/deformable_transformer.py:24,201,283; semantics:
// SURVEY.md Appendix A / oracle/msda_explicit.py).
//
// Work decomposition: one thread = 8 channels of one (image n, query q, head m).  Consecutive
// threads walk the channel groups of one query row, so the (N,Lq,M*D) output / grad_out rows are
// read and written fully coalesced (16 B bf16 or 32 B f32 per lane), and the D/8 lanes of a head
// sit next to each other for the shuffle reductions of the backward pass.  Every bilinear corner
// is ONE 16-byte (bf16) load; with head-major value maps (N,M,S,D) the two x-neighbours of a
// sample are adjacent in memory and neighbouring queries share cache lines, so the 209 MB/img/layer
// of gather traffic (YCB-V) is served by L1/L2 while HBM only sees the algorithmic bytes.
//
// FUSED variants take the raw sampling_offsets / attention_weights Linear outputs and fold in the
// softmax over L*P, loc = ref + off/(W,H) and (backward) the softmax Jacobian, so neither the fp32
// locations nor the weights are ever materialised in HBM.
#include <stdlib.h>

#include "common.cuh"

namespace poet {

constexpr int MAXL = 8;           // levels a launch may carry (the specialised kernels are instantiated for 1..4, the generic ones take any)

struct MsdaP {
    const void* value;
    int64_t vs_n, vs_s, vs_m;
    int64_t gs_n, gs_s, gs_m;   // element strides of grad_value (backward; = the value strides unless the caller lays the gradient out differently)
    const void* q1;      // fused: offattn; plain: sampling_loc
    const void* q2;      // plain: attn_weight
    int64_t ldq, ldg;    // row strides (elements) of q1 and of g1 (fused: the gradient rows may sit in a wider buffer)
    int logit_col;
    const float* ref;
    int64_t ref_bs;
    void* out;
    const void* grad_out;
    float* grad_value;
    void* g1;            // fused: grad_offattn; plain: grad_loc
    void* g2;            // plain: grad_attn
    int N, S, M, D, Lq;
    int H[MAXL], W[MAXL], start[MAXL];
    int64_t total;       // N*Lq*M*(D/8) threads
    int groups, tpg;     // channel groups per row, per head
    int grid_queries;    // queries are the pixels of the flattened levels, in order (encoder self-attention)
    int parts;           // backward: bit0 = d(offsets|logits)/d(loc,attn) kernel, bit1 = d(value) scatter kernel
    int gv_bf16;         // grad_value is bf16 (LDS-tiled scatter only): windows and far corners leave through packed bf16x2 atomics
    int q_f16;           // fused, encoder shape: the offsets | logits buffer q1 holds IEEE fp16 (POET_F16), its gradient g1 stays bf16
    int v_f16;           // fused, encoder shape: the value maps hold IEEE fp16 (POET_F16; shared-geometry gathers only)
};

// a packed pair of 2-byte storage values -> two floats: bf16 (shift / mask) or fp16 (v_cvt_f32_f16); QH is a kernel template
// parameter, so each kernel carries one form
template <bool QH> __device__ __forceinline__ float q_lo(uint32_t v) { if constexpr (QH) return h_lo(v); else return __uint_as_float(v << 16); }
template <bool QH> __device__ __forceinline__ float q_hi(uint32_t v) { if constexpr (QH) return h_hi(v); else return __uint_as_float(v & 0xffff0000u); }

template <typename TQ, int P>
__device__ __forceinline__ void load_p(const TQ* p, float* o, int n) {
    if constexpr (P == 4) {
        if (n == 8) vec<TQ, 8>::ld(p, o);
        else vec<TQ, 4>::ld(p, o);
    } else {
        for (int i = 0; i < n; ++i) o[i] = io<TQ>::ld(p + i);
    }
}
template <typename TQ, int P>
__device__ __forceinline__ void store_p(TQ* p, const float* o, int n) {
    if constexpr (P == 4) {
        if (n == 8) vec<TQ, 8>::st(p, o);
        else vec<TQ, 4>::st(p, o);
    } else {
        for (int i = 0; i < n; ++i) io<TQ>::st(p + i, o[i]);
    }
}

struct Corner {
    int64_t o00, o01, o10, o11;   // element offsets of the 4 corner pixels (clamped in range)
    float w00, w01, w10, w11;     // bilinear weights, zero where the corner is outside the map
    float fx, fy;
    float m00, m01, m10, m11;     // validity (0/1)
};

__device__ __forceinline__ Corner make_corner(float px, float py, int H, int W, int start, int64_t vs_s) {
    Corner c;
    px = fminf(fmaxf(px, -2.f), (float)W + 1.f);
    py = fminf(fmaxf(py, -2.f), (float)H + 1.f);
    const float x0f = floorf(px), y0f = floorf(py);
    c.fx = px - x0f;
    c.fy = py - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float vx0 = (x0 >= 0 && x0 < W) ? 1.f : 0.f, vx1 = (x1 >= 0 && x1 < W) ? 1.f : 0.f;
    const float vy0 = (y0 >= 0 && y0 < H) ? 1.f : 0.f, vy1 = (y1 >= 0 && y1 < H) ? 1.f : 0.f;
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y1, 0), H - 1);
    c.m00 = vy0 * vx0; c.m01 = vy0 * vx1; c.m10 = vy1 * vx0; c.m11 = vy1 * vx1;
    c.w00 = (1.f - c.fy) * (1.f - c.fx) * c.m00;
    c.w01 = (1.f - c.fy) * c.fx * c.m01;
    c.w10 = c.fy * (1.f - c.fx) * c.m10;
    c.w11 = c.fy * c.fx * c.m11;
    c.o00 = (int64_t)(start + yc0 * W + xc0) * vs_s;
    c.o01 = (int64_t)(start + yc0 * W + xc1) * vs_s;
    c.o10 = (int64_t)(start + yc1 * W + xc0) * vs_s;
    c.o11 = (int64_t)(start + yc1 * W + xc1) * vs_s;
    return c;
}

// Corner geometry for the gather kernels.  Offsets are 32-bit BYTE offsets from MsdaP::value (the address forms as
// SGPR base + VGPR offset, no 64-bit VALU; the host rejects value maps of 4 GiB and more), built with 24-bit multiplies
// (v_mad_u32_u24 is full rate, v_mul_lo_u32 / v_mad_u64_u32 quarter rate: the 64-bit form of this address arithmetic was
// a third of the forward kernel's VALU time).  v_cvt_i32_f32 saturates and every use of x0/y0 is an unsigned range test
// or a clamp, so wild sampling locations need no float clamping.  Invalid corners get the address of a clamped pixel.
// a * b + c with 24-bit a, b (full-rate v_mad_u32_u24; the compiler falls back to the quarter-rate v_mul_lo_u32 when it
// cannot prove the operand ranges) and clamp(x, 0, hi) for hi >= 0 (v_med3_i32 instead of v_max + v_min)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int clamp0(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
    return r;
}

struct Geo {
    uint32_t o00, o01, o10, o11;
    float fx, fy;
    bool vx0, vx1, vy0, vy1;
};
__device__ __forceinline__ Geo corner_geo(float px, float py, int H, int W, uint32_t base, uint32_t pix_bytes) {
    Geo g;
    const float x0f = floorf(px), y0f = floorf(py);
    g.fx = px - x0f;
    g.fy = py - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = (int)((unsigned)x0 + 1u), y1 = (int)((unsigned)y0 + 1u);
    g.vx0 = (unsigned)x0 < (unsigned)W; g.vx1 = (unsigned)x1 < (unsigned)W;
    g.vy0 = (unsigned)y0 < (unsigned)H; g.vy1 = (unsigned)y1 < (unsigned)H;
    const int xc0 = clamp0(x0, W - 1), xc1 = clamp0(x1, W - 1), yc0 = clamp0(y0, H - 1), yc1 = clamp0(y1, H - 1);
    g.o00 = mad24(mad24(yc0, W, xc0), pix_bytes, base); g.o01 = mad24(mad24(yc0, W, xc1), pix_bytes, base);
    g.o10 = mad24(mad24(yc1, W, xc0), pix_bytes, base); g.o11 = mad24(mad24(yc1, W, xc1), pix_bytes, base);
    return g;
}

template <typename TV>
__device__ __forceinline__ void gather8(const void* value, uint32_t byte_off, float* o) {
    vec<TV, 8>::ld(reinterpret_cast<const TV*>(reinterpret_cast<const char*>(value) + byte_off), o);
}

template <typename TQ, int L, int P, bool FUSED>
__device__ __forceinline__ void load_weights(const MsdaP& p, int64_t row, int m, float* a) {
    constexpr int LP = L * P;
    if constexpr (FUSED) {
        const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq + p.logit_col + m * LP;
#pragma unroll
        for (int l = 0; l < L; ++l) load_p<TQ, P>(lp + l * P, a + l * P, P);
        float mx = a[0];
#pragma unroll
        for (int i = 1; i < LP; ++i) mx = fmaxf(mx, a[i]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LP; ++i) { a[i] = __expf(a[i] - mx); s += a[i]; }
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < LP; ++i) a[i] *= inv;
    } else {
        const TQ* wp = reinterpret_cast<const TQ*>(p.q2) + (row * p.M + m) * LP;
#pragma unroll
        for (int l = 0; l < L; ++l) load_p<TQ, P>(wp + l * P, a + l * P, P);
    }
}

// pixel coordinates (px,py) of the P points of level l for (row, m)
template <typename TQ, int L, int P, bool FUSED>
__device__ __forceinline__ void load_points(const MsdaP& p, int64_t row, int n, int q, int m, int l, int Wl, int Hl, float* xy) {
    if constexpr (FUSED) {
        const TQ* op = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq + (m * L + l) * P * 2;
        load_p<TQ, P>(op, xy, 2 * P);
        const float* rp = p.ref + (int64_t)n * p.ref_bs + ((int64_t)q * L + l) * 2;
        const float rx = rp[0] * (float)Wl - 0.5f, ry = rp[1] * (float)Hl - 0.5f;
#pragma unroll
        for (int i = 0; i < P; ++i) { xy[2 * i] += rx; xy[2 * i + 1] += ry; }
    } else {
        const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + ((row * p.M + m) * L + l) * P * 2;
        load_p<TQ, P>(lp, xy, 2 * P);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            xy[2 * i] = xy[2 * i] * (float)Wl - 0.5f;
            xy[2 * i + 1] = xy[2 * i + 1] * (float)Hl - 0.5f;
        }
    }
}

// Workgroup b runs on XCD b%8 (observed dispatch order).  Hand each XCD a contiguous range of query rows so that its
// private 4 MB L2 only ever holds the value maps of the image(s) it is working on (3.3 MB per image at 640x480) instead
// of all N images interleaved (measured: 370 -> 284 us per encoder launch).  Speed heuristic only.
__device__ __forceinline__ int64_t xcd_contiguous_block(int64_t b, int64_t nb) {
    const int64_t q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// One level at a time in a ROLLED loop: 4*P corner loads (16 B each) are in flight per thread, and other waves cover the
// gather latency.  (Fully unrolled, the allocator tries to keep all 16*P loads resident: 256 VGPRs at best, and a spilling
// variant -- 3x slower -- whenever unrelated code shifts.)  The kernel is VALU bound (PMC: ~80% VALU busy), so the body is
// written for instruction count: 32-bit gather offsets, masked 1-D weights, FMA chains straight into the accumulators.
template <typename TV, typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256, 4) void msda_fwd_kernel(const MsdaP p) {
    // lanes of a (query, head): [x corner 0/1][channel group]: the two x-neighbours of a sample are adjacent pixels of a
    // head-major map, so the 2*tpg lanes read 2 * 16*tpg contiguous bytes per (sample, row) -- one L1 access, not two
    const int64_t t = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= p.total * 2) return;
    const int g2 = p.groups * 2;
    const int64_t row = t / g2;
    const int c8 = (int)(t - row * g2);
    const int m = c8 / (2 * p.tpg), rem = c8 - m * 2 * p.tpg;
    const bool xc = rem >= p.tpg;
    const int dsub = xc ? rem - p.tpg : rem;
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);

    float a[L * P];
    load_weights<TQ, L, P, FUSED>(p, row, m, a);
    const uint32_t lane_base = (uint32_t)(((int64_t)n * p.vs_n + (int64_t)m * p.vs_m + dsub * 8) * (int64_t)sizeof(TV));
    const uint32_t pix_bytes = (uint32_t)p.vs_s * (uint32_t)sizeof(TV);

    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        int Hl = p.H[0], Wl = p.W[0], Sl = p.start[0];      // select chain: a run-time index into the by-value kernel
#pragma unroll                                               // argument would put the arrays in scratch memory
        for (int k = 1; k < L; ++k)
            if (l == k) { Hl = p.H[k]; Wl = p.W[k]; Sl = p.start[k]; }
        const uint32_t lvl_base = lane_base + (uint32_t)Sl * pix_bytes;
        float xy[2 * P];
        load_points<TQ, L, P, FUSED>(p, row, n, q, m, l, Wl, Hl, xy);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            float aw = a[i];
#pragma unroll
            for (int k = 1; k < L; ++k) aw = (l == k) ? a[k * P + i] : aw;          // a[] stays in registers (l is a run-time index)
            const float px = xy[2 * i], py = xy[2 * i + 1];
            const float x0f = floorf(px), y0f = floorf(py), fx = px - x0f, fy = py - y0f;
            const int x = (int)((unsigned)(int)x0f + (xc ? 1u : 0u)), y0 = (int)y0f, y1 = (int)((unsigned)y0 + 1u);
            const float wx = (unsigned)x < (unsigned)Wl ? (xc ? fx : 1.f - fx) : 0.f;
            const float w0 = (unsigned)y0 < (unsigned)Hl ? (1.f - fy) * aw * wx : 0.f;
            const float w1 = (unsigned)y1 < (unsigned)Hl ? fy * aw * wx : 0.f;
            const int xcl = clamp0(x, Wl - 1);
            const uint32_t o0 = mad24(mad24(clamp0(y0, Hl - 1), Wl, xcl), pix_bytes, lvl_base);
            const uint32_t o1 = mad24(mad24(clamp0(y1, Hl - 1), Wl, xcl), pix_bytes, lvl_base);
            float v0[8], v1[8];
            gather8<TV>(p.value, o0, v0);
            gather8<TV>(p.value, o1, v1);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) acc[ch] = fmaf(w0, v0[ch], acc[ch]);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) acc[ch] = fmaf(w1, v1[ch], acc[ch]);
        }
    }
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) acc[ch] += __shfl_xor(acc[ch], p.tpg, 64);
    if (xc) return;
    TQ* op = reinterpret_cast<TQ*>(p.out) + row * ((int64_t)p.M * p.D) + m * p.D + dsub * 8;
    vec<TQ, 8>::st(op, acc);
}

// sum over the n (power of two) consecutive lanes of a group: DPP quad permutes for the first two steps (VALU only)
__device__ __forceinline__ float group_sum(float v, int n) {
    if (n >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    if (n >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    for (int o = 4; o < n; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// 8-channel slices as loaded (16 B for bf16, 32 B for fp32) and their dot product.  (v_dot2c_f32_bf16 would take the bf16
// pairs directly, but measured here it was both slower than shift/and + FMA and wrong in this kernel: not used.)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    u32x4_t v;
    __device__ __forceinline__ void load(const void* base, uint32_t byte_off) {
        v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(base) + byte_off);
    }
    __device__ __forceinline__ float f(int i) const { return (i & 1) ? __uint_as_float(v[i >> 1] & 0xffff0000u) : __uint_as_float(v[i >> 1] << 16); }
};
template <> struct Raw8<float> {
    float x[8];
    __device__ __forceinline__ void load(const void* base, uint32_t byte_off) {
        vec<float, 8>::ld(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off), x);
    }
    __device__ __forceinline__ float f(int i) const { return x[i]; }
};
template <typename TA, typename TB>
__device__ __forceinline__ float dot8(const float (&a)[8], const Raw8<TB>& b) {
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i += 2) { d0 = fmaf(a[i], b.f(i), d0); d1 = fmaf(a[i + 1], b.f(i + 1), d1); }
    return d0 + d1;
}

// d(offsets | logits) (fused) or d(loc), d(attn) (plain).  Same lane layout as the forward: [x corner][channel group] per
// (query, head), so a lane loads the two rows of ONE x corner; with T = (1-fy) d0 + fy d1 of that corner
//   d/d(attn) = sum_x wx T,   d/d(px) = aw (T[x1] - T[x0]),   d/d(py) = aw sum_x wx (d1 - d0)
// are all plain sums over the group's 2*tpg lanes.
// (bid, nb: the block index / count of THIS part -- the kernel's own, or its share of a combined launch, see msda_bwd_both_kernel)
template <typename TV, typename TQ, int L, int P, bool FUSED>
__device__ __forceinline__ void msda_bwd_body(const MsdaP& p, int64_t bid, int64_t nb) {
    const int64_t t = xcd_contiguous_block(bid, nb) * 256 + threadIdx.x;      // same L2 argument as the forward
    if (t >= p.total * 2) return;
    const int g2 = p.groups * 2, gl = 2 * p.tpg;
    const int64_t row = t / g2;
    const int c8 = (int)(t - row * g2);
    const int m = c8 / gl, rem = c8 - m * gl;
    const bool xc = rem >= p.tpg;
    const int dsub = xc ? rem - p.tpg : rem;
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);

    float a[L * P];
    load_weights<TQ, L, P, FUSED>(p, row, m, a);
    const uint32_t lane_base = (uint32_t)(((int64_t)n * p.vs_n + (int64_t)m * p.vs_m + dsub * 8) * (int64_t)sizeof(TV));
    const uint32_t pix_bytes = (uint32_t)p.vs_s * (uint32_t)sizeof(TV);

    float g[8];
    vec<TQ, 8>::ld(reinterpret_cast<const TQ*>(p.grad_out) + row * ((int64_t)p.M * p.D) + m * p.D + dsub * 8, g);

    float da[L * P];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int Wl = p.W[l], Hl = p.H[l];
        const uint32_t lvl_base = lane_base + (uint32_t)p.start[l] * pix_bytes;
        float xy[2 * P], dxy[2 * P];
        load_points<TQ, L, P, FUSED>(p, row, n, q, m, l, Wl, Hl, xy);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const float px = xy[2 * i], py = xy[2 * i + 1];
            const float x0f = floorf(px), y0f = floorf(py), fx = px - x0f, fy = py - y0f;
            const int x = (int)((unsigned)(int)x0f + (xc ? 1u : 0u)), y0 = (int)y0f, y1 = (int)((unsigned)y0 + 1u);
            const bool vx = (unsigned)x < (unsigned)Wl;
            const int xcl = clamp0(x, Wl - 1);
            Raw8<TV> v0, v1;
            v0.load(p.value, mad24(mad24(clamp0(y0, Hl - 1), Wl, xcl), pix_bytes, lvl_base));
            v1.load(p.value, mad24(mad24(clamp0(y1, Hl - 1), Wl, xcl), pix_bytes, lvl_base));
            const float d0 = (vx && (unsigned)y0 < (unsigned)Hl) ? dot8<TQ, TV>(g, v0) : 0.f;
            const float d1 = (vx && (unsigned)y1 < (unsigned)Hl) ? dot8<TQ, TV>(g, v1) : 0.f;
            const float T = fmaf(fy, d1 - d0, d0);                       // (1-fy) d0 + fy d1
            const float wx = xc ? fx : 1.f - fx, aw = a[l * P + i];
            da[l * P + i] = group_sum(wx * T, gl);
            dxy[2 * i] = group_sum(aw * (xc ? T : -T), gl);
            dxy[2 * i + 1] = group_sum(aw * wx * (d1 - d0), gl);
        }
        if (rem == 0) {
            if constexpr (FUSED) {
                TQ* gp = reinterpret_cast<TQ*>(p.g1) + row * p.ldg + (m * L + l) * P * 2;
                store_p<TQ, P>(gp, dxy, 2 * P);          // d/d(offset) = (dpx, dpy): the W,H factors cancel
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) { dxy[2 * i] *= (float)Wl; dxy[2 * i + 1] *= (float)Hl; }
                TQ* gp = reinterpret_cast<TQ*>(p.g1) + ((row * p.M + m) * L + l) * P * 2;
                store_p<TQ, P>(gp, dxy, 2 * P);
            }
        }
    }
    if (rem == 0) {
        constexpr int LP = L * P;
        if constexpr (FUSED) {
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < LP; ++i) dot += a[i] * da[i];
#pragma unroll
            for (int i = 0; i < LP; ++i) da[i] = a[i] * (da[i] - dot);
            TQ* gp = reinterpret_cast<TQ*>(p.g1) + row * p.ldg + p.logit_col + m * LP;
#pragma unroll
            for (int l = 0; l < L; ++l) store_p<TQ, P>(gp + l * P, da + l * P, P);
        } else {
            TQ* gp = reinterpret_cast<TQ*>(p.g2) + (row * p.M + m) * LP;
#pragma unroll
            for (int l = 0; l < L; ++l) store_p<TQ, P>(gp + l * P, da + l * P, P);
        }
    }
}
template <typename TV, typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const MsdaP p) { msda_bwd_body<TV, TQ, L, P, FUSED>(p, blockIdx.x, gridDim.x); }

// ---- shared-geometry gathers: the encoder's shape (M = 16 heads x D = 16, L = 4 levels x P = 4 points, bf16) --------------
// The general gather kernels above are VALU bound, and half of their VALU time is sample GEOMETRY computed redundantly: the 4
// lanes [x corner][channel half] of a (query, head) each redo floor / range tests / clamps / 24-bit address multiplies / weights
// of all 16 points (~130 issue cycles per point: on gfx950 v_cvt, v_floor, v_cmp, v_cndmask, v_med3, v_mad_u32_u24, v_lshlrev
// and every DPP / SGPR-operand form issue at 4 cycles per wave-instruction against 2 for v_fma / v_and / v_add -- measured,
// profiles/probes/valu_probe.hip).  Here one wave = one query row (16 heads x 4 lanes) and the 4 lanes of a head split the geometry by
// LEVEL: lane o prepares the 4 points of level o for BOTH x corners (softmax over the quad with DPP) and leaves one 16-byte record
//     { byte offset of the upper corner pixel, of the lower one, weight of the upper, of the lower }
// per (x corner, point, head) in LDS; the gather loop then reads its corner's record with one ds_read_b128 (LDS pipe, no
// VALU) and spends VALU only on bf16 unpack + FMA.  The backward's per-point reductions go the other way through the same
// slots: each gather lane leaves its two partial dot products (v_dot2c_f32_bf16 on the packed operands: no unpack at all) in
// the slot it just consumed, and the level's owner lane adds the 4 partials of a point and forms d(offsets) / d(logits) for
// its level -- 8 + 4 contiguous bf16 per lane -- instead of three 4-lane shuffle reductions per point on every lane.
// What is left is the L1 path: 32 gather instructions of 1 KB per wave, each touching 16 half-used 128-byte lines.
// (Tried: preparing level by level with 2.3 KB of records per wave and 7-8 waves per SIMD instead of 4 -- slower, 208 / 296 us
// against 178 / 247: eight narrow loads per lane instead of three, four serialized prepare -> LDS -> gather chains.)
constexpr int SH_PLANE = 16 * 16 * 16 + 128;     // bytes per x-corner plane of a wave: [point 16][head 16] x 16 B (+ pad: the two
                                                 // planes sit on opposite bank halves)
constexpr int SH_WAVE = 2 * SH_PLANE;
constexpr int SH_WAVES = 4;                       // waves (= query rows) per workgroup

// 16-byte record slot of (pair, level) inside a 256-byte row of records: rotated by two slots per level.  The owner lanes write
// their records with ds_write_b128, which the LDS services in groups of 8 lanes = 2 pairs x 4 LEVELS: with slot = pair all four
// levels of a pair land on the same bank quad (measured: SQ_LDS_BANK_CONFLICT = 60 % of SQ_LDS_IDX_ACTIVE in these kernels, all
// of it record traffic); with the rotation a group covers 8 different quads, and the readers' groups (4 pairs of ONE level) stay
// conflict-free as before.
__device__ __forceinline__ int sh_slot(int pair, int level) { return ((pair + 2 * level) & 15) * 16; }

struct ShGeo {                                    // what the owner lane keeps for the backward (its level's 4 points)
    float aw[4], fx[4], fy[4];
    uint32_t valid[4];                            // bit0 vx0, bit1 vx1, bit2 vy0, bit3 vy1
};

__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)); }

// sum over 8 channels of a[ch] * b[ch], both operands packed bf16 pairs
__device__ __forceinline__ float dot8_packed(const u32x4_t& a, const u32x4_t& b) {
    float d = 0.f;
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(a[0]), "v"(b[0]));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(a[1]), "v"(b[1]));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(a[2]), "v"(b[2]));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(a[3]), "v"(b[3]));
    return d;
}

// ---- fp16 value maps (MsdaP::v_f16, round 6) ----------------------------------------------------------------------------------
// acc += w * (fp16 half of v): v_fma_mix_f32 converts the selected half on the way into the FMA -- no unpack instruction at all
// (a bf16 pair costs v_and + v_lshl before its two FMAs: 4 VALU per pair against 2, and the encoder gathers are VALU-issue bound)
__device__ __forceinline__ void fmix_lo(float& acc, uint32_t v, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(v), "v"(w));
}
__device__ __forceinline__ void fmix_hi(float& acc, uint32_t v, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(v), "v"(w));
}
// sum over 8 channels of a[ch] * b[ch], both operands packed fp16 pairs (exact products, fp32 sum)
__device__ __forceinline__ float dot8_packed_h(const u32x4_t& a, const u32x4_t& b) {
    float d = 0.f;
    asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(d) : "v"(a[0]), "v"(b[0]));
    asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(d) : "v"(a[1]), "v"(b[1]));
    asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(d) : "v"(a[2]), "v"(b[2]));
    asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(d) : "v"(a[3]), "v"(b[3]));
    return d;
}
// The backward's <grad_out, value> products want both operands in ONE 2-byte format.  grad_out arrives as bf16 with the full fp32
// exponent range (gradients of 1e-7 are routine), so it cannot simply be re-typed: the 16 channels of a (query, head) -- the 8 of
// this lane and the 8 of its channel-half partner, lane ^ 1 -- are rescaled by a common power of two that puts their largest
// exponent at 2^14, in the integer domain on the packed pairs (|x| = exponent:8 | mantissa:7 -> subtract K << 7 with unsigned
// saturation, shift left by 3 into exponent:5 | mantissa:10; what falls below 2^-28 of the largest channel becomes zero), and the
// owner lane multiplies the inverse scale back into d(offsets) / d(logits).  bf16 -> fp16 is exact for everything kept (7 -> 10
// mantissa bits); ~30 VALU per pass.  Returns the inverse scale.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ float g_to_f16_scaled(u32x4_t& g) {
    const uint32_t a0 = g[0] & 0x7fff7fffu, a1 = g[1] & 0x7fff7fffu, a2 = g[2] & 0x7fff7fffu, a3 = g[3] & 0x7fff7fffu;
    uint32_t mx = pk_max_u16(pk_max_u16(a0, a1), pk_max_u16(a2, a3));
    mx = max(mx & 0xffffu, mx >> 16);
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0xB1, 0xf, 0xf, true));       // the other channel half of the head
    const int kexp = max((int)(mx >> 7) - 29, 0);                                                   // (mx < 2^15: exponent = bits 14..7)
    const uint32_t k1 = (uint32_t)kexp << 7, kpk = k1 | (k1 << 16);
    const uint32_t a[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const us2_t d = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2_t, a[j]), __builtin_bit_cast(us2_t, kpk));
        const us2_t sh = d << (unsigned short)3;
        g[j] = (g[j] & 0x80008000u) | __builtin_bit_cast(uint32_t, sh);
    }
    return __uint_as_float((uint32_t)(kexp + 15) << 23);            // fp16 value = bf16 value * 2^(112 - kexp)
}

// LDS traffic between lanes of ONE wave needs no s_barrier (the LDS ops of a wave execute in order); this only stops the
// compiler from moving accesses across
__device__ __forceinline__ void sh_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// phase A: lane (pair = lane >> 2: its (query row, head m); level o = lane & 3) -> records of its 4 points at slot index `pair`;
// returns what the backward needs.
struct ShRaw { uint2 lg; uint4 off; };                         // packed logits (4) and offsets (4 x (dx, dy)) of (row, head m, level o)
__device__ __forceinline__ ShRaw sh_load(const MsdaP& p, int row, int m, int o) {
    const bf16_t* qrow = reinterpret_cast<const bf16_t*>(p.q1) + (int64_t)row * p.ldq;
    ShRaw r;
    r.lg = *reinterpret_cast<const uint2*>(qrow + p.logit_col + m * 16 + o * 4);
    r.off = *reinterpret_cast<const uint4*>(qrow + (m * 4 + o) * 8);
    return r;
}
template <bool KEEP, bool QH>
__device__ __forceinline__ void sh_prepare(const MsdaP& p, const ShRaw& raw, float2 rf, int n, int m, int lane, char* wlds, ShGeo& gk) {
    const int pair = lane >> 2, o = lane & 3;
    float lg[4], off[8];
    lg[0] = q_lo<QH>(raw.lg.x); lg[1] = q_hi<QH>(raw.lg.x);
    lg[2] = q_lo<QH>(raw.lg.y); lg[3] = q_hi<QH>(raw.lg.y);
    off[0] = q_lo<QH>(raw.off.x); off[1] = q_hi<QH>(raw.off.x);
    off[2] = q_lo<QH>(raw.off.y); off[3] = q_hi<QH>(raw.off.y);
    off[4] = q_lo<QH>(raw.off.z); off[5] = q_hi<QH>(raw.off.z);
    off[6] = q_lo<QH>(raw.off.w); off[7] = q_hi<QH>(raw.off.w);
    // softmax over the 16 logits of (query, head): 4 per lane, quad reductions
    float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    mx = fmaxf(mx, quad_xor1(mx)); mx = fmaxf(mx, quad_xor2(mx));
    float e[4], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[i] = __expf(lg[i] - mx); sum += e[i]; }
    sum += quad_xor1(sum); sum += quad_xor2(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);                  // 1 ulp
    int Wl = p.W[0], Hl = p.H[0], Sl = p.start[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (o == k) { Wl = p.W[k]; Hl = p.H[k]; Sl = p.start[k]; }
    const float rx = rf.x * (float)Wl - 0.5f, ry = rf.y * (float)Hl - 0.5f;
    const uint32_t pix_bytes = (uint32_t)p.vs_s * 2u;
    const uint32_t base = (uint32_t)(((int64_t)n * p.vs_n + (int64_t)m * p.vs_m) * 2) + (uint32_t)Sl * pix_bytes;
    char* rec = wlds + (o * 4) * 256 + sh_slot(pair, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float aw = e[i] * inv;
        const float px = off[2 * i] + rx, py = off[2 * i + 1] + ry;
        const float x0f = floorf(px), y0f = floorf(py), fx = px - x0f, fy = py - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;                                   // v_cvt saturates; every use is a range test or a clamp
        const int x1 = (int)((unsigned)x0 + 1u), y1 = (int)((unsigned)y0 + 1u);
        const bool vx0 = (unsigned)x0 < (unsigned)Wl, vx1 = (unsigned)x1 < (unsigned)Wl;
        const bool vy0 = (unsigned)y0 < (unsigned)Hl, vy1 = (unsigned)y1 < (unsigned)Hl;
        const uint32_t r0 = mad24(clamp0(y0, Hl - 1), Wl, 0u), r1 = mad24(clamp0(y1, Hl - 1), Wl, 0u);
        const uint32_t c0 = (uint32_t)clamp0(x0, Wl - 1), c1 = (uint32_t)clamp0(x1, Wl - 1);
        const float wy0 = vy0 ? (1.f - fy) * aw : 0.f, wy1 = vy1 ? fy * aw : 0.f;
        const float wx0 = vx0 ? 1.f - fx : 0.f, wx1 = vx1 ? fx : 0.f;
        uint4 a, b;
        a.x = mad24(r0 + c0, pix_bytes, base); a.y = mad24(r1 + c0, pix_bytes, base);
        a.z = __float_as_uint(wy0 * wx0);      a.w = __float_as_uint(wy1 * wx0);
        b.x = mad24(r0 + c1, pix_bytes, base); b.y = mad24(r1 + c1, pix_bytes, base);
        b.z = __float_as_uint(wy0 * wx1);      b.w = __float_as_uint(wy1 * wx1);
        *reinterpret_cast<uint4*>(rec + i * 256) = a;
        *reinterpret_cast<uint4*>(rec + i * 256 + SH_PLANE) = b;
        if constexpr (KEEP) {
            gk.aw[i] = aw; gk.fx[i] = fx; gk.fy[i] = fy;
            gk.valid[i] = (vx0 ? 1u : 0u) | (vx1 ? 2u : 0u) | (vy0 ? 4u : 0u) | (vy1 ? 8u : 0u);
        }
    }
    sh_wave_sync();              // the records are read by other lanes of THIS wave only
}

// Which (query row, head) a lane group of 4 works on.  QG = 1: the 16 groups of a wave are the 16 heads of ONE query.  QG = 4:
// they are 4 CONSECUTIVE queries x 4 heads, and the wave makes 4 passes over the head quarters: neighbouring queries of the
// encoder are neighbouring pixels whose samples fall into the same 128-byte lines of a head's value map, so a gather
// instruction touches ~6-8 distinct lines instead of 16 (the L1 works through a wave's request line by line).
template <int QG>
struct ShMap {
    int row, n, q, hh;
    bool live;
    __device__ __forceinline__ ShMap(const MsdaP& p, int64_t unit, int lane, int64_t rows) {
        const int pair = lane >> 2;
        const int qq = QG == 1 ? 0 : pair / (16 / QG);
        hh = QG == 1 ? pair : pair % (16 / QG);
        const int64_t r64 = unit * QG + qq;
        live = r64 < rows;
        row = (int)(live ? r64 : rows - 1);
        n = row / p.Lq;
        q = row - n * p.Lq;
    }
};

template <int QG, bool QH, bool VH>      // VH: fp16 value maps (v_fma_mix_f32 on the packed halves)
__global__ __launch_bounds__(SH_WAVES * 64) void msda_fwd_shared_kernel(const MsdaP p) {
    __shared__ __attribute__((aligned(16))) char lds[SH_WAVES * SH_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)p.N * p.Lq;
    const int64_t unit = xcd_contiguous_block(blockIdx.x, gridDim.x) * SH_WAVES + wave;      // QG consecutive query rows
    if (unit * QG >= rows) return;                                  // wave-uniform
    const ShMap<QG> mp(p, unit, lane, rows);
    char* wlds = lds + wave * SH_WAVE;
    // gather lanes: [pair][x corner xc][channel half]
    const int pair = lane >> 2, xc = (lane >> 1) & 1, dsub = lane & 1;
    const char* rd0 = wlds + xc * SH_PLANE;
    const uint32_t lane_off = (uint32_t)dsub * 16u;
    // the reference point of (row, level o) serves every pass.  (Requesting the NEXT pass's packed logits / offsets before this
    // pass's gathers was slower, 164 / 218 us against 156 / 200: memory returns in order, and an HBM-latency stream load ahead of
    // the L2-served gathers delays all of them.)
    const float2 rf = *reinterpret_cast<const float2*>(p.ref + (int64_t)mp.n * p.ref_bs + ((int64_t)mp.q * 4 + (lane & 3)) * 2);
#pragma unroll 1
    for (int pass = 0; pass < QG; ++pass) {
        const int m = pass * (16 / QG) + mp.hh;
        ShGeo unused;
        const ShRaw raw = sh_load(p, mp.row, m, lane & 3);
        sh_prepare<false, QH>(p, raw, rf, mp.n, m, lane, wlds, unused);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int l = 0; l < 4; ++l) {
            const char* rd = rd0 + sh_slot(pair, l);
            uint4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint4*>(rd + (l * 4 + i) * 256);
            Raw8<bf16_t> v0[4], v1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v0[i].load(p.value, r[i].x + lane_off); v1[i].load(p.value, r[i].y + lane_off); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float w0 = __uint_as_float(r[i].z), w1 = __uint_as_float(r[i].w);
                if constexpr (VH) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { fmix_lo(acc[2 * j], v0[i].v[j], w0); fmix_hi(acc[2 * j + 1], v0[i].v[j], w0); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { fmix_lo(acc[2 * j], v1[i].v[j], w1); fmix_hi(acc[2 * j + 1], v1[i].v[j], w1); }
                } else {
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) acc[ch] = fmaf(w0, v0[i].f(ch), acc[ch]);
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) acc[ch] = fmaf(w1, v1[i].f(ch), acc[ch]);
                }
            }
        }
        sh_wave_sync();                                             // (the next pass overwrites the records)
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) acc[ch] += quad_xor2(acc[ch]);   // the two x corners
        if (!xc && mp.live) {
            bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (int64_t)mp.row * 256 + m * 16 + dsub * 8;
            vec<bf16_t, 8>::st(op, acc);
        }
    }
}

template <int QG, bool QH, bool VH>
__global__ __launch_bounds__(SH_WAVES * 64) void msda_bwd_shared_kernel(const MsdaP p) {
    __shared__ __attribute__((aligned(16))) char lds[SH_WAVES * SH_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)p.N * p.Lq;
    const int64_t unit = xcd_contiguous_block(blockIdx.x, gridDim.x) * SH_WAVES + wave;
    if (unit * QG >= rows) return;
    const ShMap<QG> mp(p, unit, lane, rows);
    char* wlds = lds + wave * SH_WAVE;
    const int pair = lane >> 2, xc = (lane >> 1) & 1, dsub = lane & 1, o = lane & 3;
    char* slot0 = wlds + xc * SH_PLANE;
    const char* own = wlds + (o * 4) * 256 + sh_slot(pair, o);
    const uint32_t lane_off = (uint32_t)dsub * 16u;
    bf16_t* grow = reinterpret_cast<bf16_t*>(p.g1) + (int64_t)mp.row * p.ldg;
    const float2 rf = *reinterpret_cast<const float2*>(p.ref + (int64_t)mp.n * p.ref_bs + ((int64_t)mp.q * 4 + o) * 2);
#pragma unroll 1
    for (int pass = 0; pass < QG; ++pass) {
        const int m = pass * (16 / QG) + mp.hh;
        ShGeo gk;
        const ShRaw raw = sh_load(p, mp.row, m, o);
        sh_prepare<true, QH>(p, raw, rf, mp.n, m, lane, wlds, gk);
        // grad_out stays packed: <grad_out, value> over a lane's 8 channels is 4 v_dot2c_f32_bf16 (exact bf16 products, fp32 sum)
        u32x4_t g = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(p.grad_out) + (int64_t)mp.row * 256 + m * 16 + dsub * 8);
        float ginv = 1.f;                                          // fp16 value maps: grad_out re-typed to fp16 under a per-head power-of-two scale
        if constexpr (VH) ginv = g_to_f16_scaled(g);
#pragma unroll 1
        for (int l = 0; l < 4; ++l) {
            char* slot = slot0 + sh_slot(pair, l);
            uint2 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint2*>(slot + (l * 4 + i) * 256);
            Raw8<bf16_t> v0[4], v1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v0[i].load(p.value, r[i].x + lane_off); v1[i].load(p.value, r[i].y + lane_off); }
#pragma unroll
            for (int i = 0; i < 4; ++i)     // this lane's partial <grad_out, value> of the upper / lower corner, into its half of the slot it consumed
                *reinterpret_cast<float2*>(slot + (l * 4 + i) * 256 + dsub * 8) =
                    VH ? make_float2(dot8_packed_h(g, v0[i].v), dot8_packed_h(g, v1[i].v)) : make_float2(dot8_packed(g, v0[i].v), dot8_packed(g, v1[i].v));
        }
        sh_wave_sync();
        // owner lane (pair, level o): d(offsets) and d(attention) of its 4 points
        float dxy[8], da[4];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(own + i * 256);                 // x corner 0: (d0, d1) of the two channel halves
            const float4 b = *reinterpret_cast<const float4*>(own + i * 256 + SH_PLANE);      // x corner 1
            const uint32_t v = gk.valid[i];
            const float D00 = ((v & 5u) == 5u) ? a.x + a.z : 0.f, D10 = ((v & 9u) == 9u) ? a.y + a.w : 0.f;
            const float D01 = ((v & 6u) == 6u) ? b.x + b.z : 0.f, D11 = ((v & 10u) == 10u) ? b.y + b.w : 0.f;
            const float fx = gk.fx[i], fy = gk.fy[i], aw = gk.aw[i];
            const float e0 = D10 - D00, e1 = D11 - D01;
            const float T0 = fmaf(fy, e0, D00), T1 = fmaf(fy, e1, D01);
            da[i] = fmaf(fx, T1 - T0, T0);                                // (1-fx) T0 + fx T1
            if constexpr (VH) da[i] *= ginv;                              // (the dot products carry grad_out's scale: the quad shares it)
            const float aws = VH ? aw * ginv : aw;
            dxy[2 * i] = aws * (T1 - T0);                                 // d/d(offset) = (dpx, dpy): the W,H factors cancel
            dxy[2 * i + 1] = aws * fmaf(fx, e1 - e0, e0);                 // aw ((1-fx) e0 + fx e1)
            dot = fmaf(aw, da[i], dot);
        }
        sh_wave_sync();                                                   // (the next pass overwrites the slots)
        dot += quad_xor1(dot); dot += quad_xor2(dot);                     // softmax Jacobian over the 16 points of (query, head)
#pragma unroll
        for (int i = 0; i < 4; ++i) da[i] = gk.aw[i] * (da[i] - dot);
        if (mp.live) {
            vec<bf16_t, 8>::st(grow + (m * 4 + o) * 8, dxy);
            vec<bf16_t, 4>::st(grow + p.logit_col + m * 16 + o * 4, da);
        }
    }
}


#ifdef POET_PROBE_KERNELS
#include "../../profiles/probes/kernels/msda_fwd_hyb.inc"
#endif

// packed bf16x2 memory-side atomic add (global_atomic_pk_add_bf16): `p2` = the even channel of a pair (4-byte aligned)
typedef short gv_short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gv16_add2(bf16_t* p2, float lo, float hi) {
    const uint32_t w = pack_bf2(lo, hi);
    const gv_short2_t v = {(short)(w & 0xffffu), (short)(w >> 16)};
    (void)__builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) gv_short2_t*)(p2), v);
}
// ... of ONE channel c (its pair partner receives +0)
__device__ __forceinline__ void gv16_add(bf16_t* pc, int c, float v) {
    if (c & 1) gv16_add2(pc - 1, 0.f, v);
    else gv16_add2(pc, v, 0.f);
}

// d(value): pure scatter, no value loads.  One lane per channel: the D lanes of a (query, head) add D consecutive
// floats, so every atomic wave-instruction covers whole contiguous 4*D-byte segments (coalesced into a few L2 atomic
// requests) instead of 64 scattered dwords.  The softmax / corner arithmetic is recomputed per lane (it is tiny).
template <typename TQ, int L, int P, bool FUSED>
__device__ __forceinline__ void msda_bwd_dv_body(const MsdaP& p, int64_t bid) {
    const int64_t t = bid * 256 + threadIdx.x;
    if (t >= p.total * 8) return;
    const int64_t rm = t / p.D;
    const int c = (int)(t - rm * p.D);
    const int64_t row = rm / p.M;
    const int m = (int)(rm - row * p.M);
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);
    float a[L * P];
    load_weights<TQ, L, P, FUSED>(p, row, m, a);
    const float g = io<TQ>::ld(reinterpret_cast<const TQ*>(p.grad_out) + row * ((int64_t)p.M * p.D) + m * p.D + c);
    float* gv = p.grad_value + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + c;
    bf16_t* gv16 = reinterpret_cast<bf16_t*>(p.grad_value) + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + c;
    const bool b16 = p.gv_bf16 != 0;
    // bf16 maps: the even lane of a channel pair adds both channels with ONE packed bf16x2 atomic (the corner weights are the
    // same for all D lanes of a (query, head), so the pair takes every branch together)
    auto add = [&](int64_t off, float v) __attribute__((always_inline)) {
        if (b16) {
            const float o = __shfl_xor(v, 1);
            if (!(c & 1)) gv16_add2(gv16 + off, v, o);
        } else {
            atomicAdd(gv + off, v);
        }
    };
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float xy[2 * P];
        load_points<TQ, L, P, FUSED>(p, row, n, q, m, l, p.W[l], p.H[l], xy);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const Corner cn = make_corner(xy[2 * i], xy[2 * i + 1], p.H[l], p.W[l], p.start[l], p.gs_s);
            const float ag = a[l * P + i] * g;
            if (cn.w00 != 0.f) add(cn.o00, ag * cn.w00);
            if (cn.w01 != 0.f) add(cn.o01, ag * cn.w01);
            if (cn.w10 != 0.f) add(cn.o10, ag * cn.w10);
            if (cn.w11 != 0.f) add(cn.o11, ag * cn.w11);
        }
    }
}
template <typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_bwd_dv_kernel(const MsdaP p) { msda_bwd_dv_body<TQ, L, P, FUSED>(p, blockIdx.x); }

// The decoder's backward (a few hundred queries: both kernels are latency chains of ~18 us on a handful of workgroups) as ONE launch: the
// first `nv` workgroups run the value-gradient scatter, the rest d(offsets | logits) -- the two parts are independent, and branches of a
// replayed graph do not overlap on this stack, so sharing a launch is the only way they run side by side.  Same code, same
// results.
template <typename TV, typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_bwd_both_kernel(const MsdaP p, int nv) {
    if ((int)blockIdx.x < nv) msda_bwd_dv_body<TQ, L, P, FUSED>(p, blockIdx.x);
    else msda_bwd_body<TV, TQ, L, P, FUSED>(p, (int64_t)blockIdx.x - nv, (int64_t)gridDim.x - nv);
}

// d(value) for GRID queries (encoder self-attention: query q IS pixel q of the flattened levels): LDS-privatised
// scatter.  One workgroup = (spatial tile, head, image).  It owns every query whose pixel centre falls in the tile (at
// all levels), accumulates their contributions into fp32 LDS windows (tile + halo of each level) with ds_add_f32, and
// flushes each window once with coalesced global atomics; a sample whose corner falls outside the window (large
// learned offset, padded image) goes straight to a global atomic, so the result never depends on the halo.
// Global atomics drop from (16 points x 4 corners x D) per (query, head) to ~(window pixels x D) per workgroup (~20x).
struct TileP { int TX, TY, HALO, skip, order; };     // skip: timing-breakdown aid (POET_DV_SKIP bits: 1 max pass, 2 accumulate, 4 flush)
                                                      // order: 1 = the 16 heads of one (tile, image) run together on ONE XCD (below)
// 512 threads: twice the waves on the same LDS windows (the accumulate loop is VALU/LDS-issue bound at 2 waves/SIMD)
constexpr int TILED_NT = 1024;

// i / d for 0 <= i < 2^20, d >= 1, with inv = 1.f / d: float multiply + one correction instead of the ~30-instruction
// integer division sequence (the quotient estimate is off by at most 1)
__device__ __forceinline__ int idiv_small(int i, int d, float inv) {
    int q = (int)((float)i * inv);
    const int r = i - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

__device__ __forceinline__ int cdiv_i(int a, int b) { return (a >= 0) ? (a + b - 1) / b : -((-a) / b); }

// round to nearest (ties toward +inf) in ONE instruction; __float2int_rn is v_rndne + v_cvt
__device__ __forceinline__ int cvt_rpi(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// all-reduce over the caller's 16-lane DPP row with cyclic row rotations (VALU only, no LDS traffic)
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
// (v_max_f32 with the rotation folded in as its DPP operand: written through fmaxf, every step costs a v_mov_dpp, the max
// and two canonicalising v_max v, v, v -- 12 instructions for the 4 below.  s_nop 1: a DPP operand written by the previous
// VALU instruction needs two wait states, and the hazard recogniser does not look inside inline assembly.)
__device__ __forceinline__ float row16_max(float v) {
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += row_ror<1>(v); v += row_ror<2>(v); v += row_ror<4>(v); return v + row_ror<8>(v);
}

// value of lane J of the caller's 16-lane DPP row, for every lane of the row (gfx90a+ row_newbcast)
template <int J>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, false); }

// sample J of the row's query: f(byte address of the left pixel of corner row k, weight of its left corner, of its right corner)
template <int LP, int J = 0, typename F>
__device__ __forceinline__ void row_points(const int (&arow)[2], const float (&cw)[4], F&& f) {
    if constexpr (J < LP) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            f(row_bcast<J>(arow[k]), __int_as_float(row_bcast<J>(__float_as_int(cw[2 * k]))), __int_as_float(row_bcast<J>(__float_as_int(cw[2 * k + 1]))));
        row_points<LP, J + 1>(arow, cw, f);
    }
}

// global load at a 32-bit BYTE offset from a uniform base: the address forms as SGPR base + VGPR offset, no 64-bit VALU
template <typename T>
__device__ __forceinline__ T ldg32(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// D == 16, P == 4 (one DPP row of 16 lanes = the 16 channels of a head = the <= 16 sample points of a query).
template <typename TQ, int L, bool QH>
__global__ __launch_bounds__(TILED_NT) void msda_bwd_dv_tiled_kernel(const MsdaP p, const TileP tp) {
    // LDS windows are INT32 FIXED POINT: ds_add_u32 sustains ~6.8 T lane-ops/s on MI355X, ds_add_f32 only ~0.2 T/s
    // (measured), i.e. float LDS atomics are no faster than L2 atomics.  Scale = 2^k per workgroup with
    // max|grad_out| * 2^k ~ 2^18: resolution 4e-6 of the tile's largest gradient, 2^13 contributions of headroom.
    // Used for bf16 storage only (the fp32 parity path keeps the exact float scatter).
    static_assert(sizeof(TQ) == 2, "bf16 storage only");
    extern __shared__ __attribute__((aligned(16))) int win[];
    __shared__ float s_red[TILED_NT / 64];
    constexpr int P = 4, D = 16, LP = L * P, NSLOT = TILED_NT / D;
    const int tid = threadIdx.x;
    // Work item -> (head, tile, image).  The operand rows are shared between heads: head m reads 32 B (grad_out), 64 B (offsets) and
    // 32 B (logits) of rows whose 64-byte sectors it shares with head m ^ 1, and the 16 heads together read every byte of the
    // rows exactly once.  A (tiles, M, N) grid dealt round-robin over the XCDs puts heads m and m + 1 of one (tile, image) on
    // DIFFERENT XCDs (private L2s): every shared sector is fetched twice (counters, round 4: 1.84x / 1.87x the algorithmic bytes
    // at LM-O / 1280x960).  order = 1: head fastest, and each XCD takes a CONTIGUOUS range of work items, so the 16 heads of a
    // (tile, image) are resident on one XCD at the same time and share the rows through its L2.
    int tx, ty, m, n;
    {
        const int txy = tp.TX * tp.TY;
        if (tp.order) {
            const int w = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
            m = w % p.M;
            const int t = (w / p.M) % txy;
            n = w / (p.M * txy);
            tx = t % tp.TX; ty = t / tp.TX;
        } else {
            const int t = blockIdx.x % txy;
            m = (blockIdx.x / txy) % p.M;
            n = blockIdx.x / (txy * p.M);
            tx = t % tp.TX; ty = t / tp.TX;
        }
    }
    int wx0[L], wy0[L], ww[L], wh[L], wp[L], loff[L + 1];    // loff in pixels; wp = LDS row pitch of the window
    loff[0] = 1;                                             // one pad pixel in front (see `arow` below)
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int W = p.W[l], H = p.H[l];
        const int ax = max((tx * W) / tp.TX - tp.HALO, 0), bx = min(cdiv_i((tx + 1) * W, tp.TX) + tp.HALO, W);
        const int ay = max((ty * H) / tp.TY - tp.HALO, 0), by = min(cdiv_i((ty + 1) * H, tp.TY) + tp.HALO, H);
        wx0[l] = ax; wy0[l] = ay; ww[l] = bx - ax; wh[l] = by - ay; wp[l] = ww[l];
        loff[l + 1] = loff[l] + wp[l] * wh[l];
    }
    {
        int4* w4 = reinterpret_cast<int4*>(win);
        for (int i = tid; i < loff[L] * (D / 4); i += TILED_NT) w4[i] = make_int4(0, 0, 0, 0);
    }

    // lane roles (loop invariant): channel c of slot's query; and owner of sample point j = c (level l = j / P)
    const int c = tid & 15, slot = tid >> 4, r = slot & 3;
    const bool has = c < LP;
    const int l = has ? (c >> 2) : 0;
    int Wl = p.W[0], Hl = p.H[0], startl = p.start[0], lwx0 = wx0[0], lwy0 = wy0[0], lww = ww[0], lwh = wh[0], lwp = wp[0], lo = loff[0];
#pragma unroll
    for (int t = 1; t < L; ++t)
        if (l == t) { Wl = p.W[t]; Hl = p.H[t]; startl = p.start[t]; lwx0 = wx0[t]; lwy0 = wy0[t]; lww = ww[t]; lwh = wh[t]; lwp = wp[t]; lo = loff[t]; }
    const float WlF = (float)Wl, HlF = (float)Hl;
    const uint32_t g_lane = (uint32_t)(m * D + c) * 2u, g_row = (uint32_t)(p.M * D) * 2u;             // byte offsets into grad_out
    const uint32_t q_row = (uint32_t)p.ldq * 2u;
    const uint32_t lg_lane = (uint32_t)(p.logit_col + m * LP + (has ? c : 0)) * 2u;
    const uint32_t of_lane = (uint32_t)((m * LP + (has ? c : 0)) * 2) * 2u;
    const uint32_t rf_lane = (uint32_t)(n * p.ref_bs + l * 2) * 4u, rf_q = (uint32_t)(L * 2) * 4u;
    const int row0 = n * p.Lq;

    // pass 1: max |grad_out| over this tile's queries (those whose pixel centre falls in the tile, at every level)
    // -> power-of-two scale
    float gm = (tp.skip & 1) ? 4.f : 0.f;
#pragma unroll 1
    for (int lq = 0; lq < ((tp.skip & 1) ? 0 : L); ++lq) {
        const int W = p.W[lq], H = p.H[lq];
        const int qx0 = max(cdiv_i(2 * W * tx - tp.TX, 2 * tp.TX), 0), qx1 = min(max(cdiv_i(2 * W * (tx + 1) - tp.TX, 2 * tp.TX), 0), W);
        const int qy0 = max(cdiv_i(2 * H * ty - tp.TY, 2 * tp.TY), 0), qy1 = min(max(cdiv_i(2 * H * (ty + 1) - tp.TY, 2 * tp.TY), 0), H);
        const int qw = qx1 - qx0, nq = qw * (qy1 - qy0);
        const float inv_qw = 1.f / (float)max(qw, 1);
        const int qbase = row0 + p.start[lq] + qy0 * W + qx0;
        auto g_of = [&](int i) __attribute__((always_inline)) {
            const int iy = idiv_small(i, qw, inv_qw);
            return fabsf(bf2f(ldg32<TQ>(p.grad_out, (uint32_t)(qbase + iy * W + (i - iy * qw)) * g_row + g_lane)));
        };
        int i = slot;
        for (; i + 3 * NSLOT < nq; i += 4 * NSLOT) {         // 4 independent loads in flight
            const float g0 = g_of(i), g1 = g_of(i + NSLOT), g2 = g_of(i + 2 * NSLOT), g3 = g_of(i + 3 * NSLOT);
            gm = fmaxf(fmaxf(gm, fmaxf(g0, g1)), fmaxf(g2, g3));
        }
        for (; i < nq; i += NSLOT) gm = fmaxf(gm, g_of(i));
    }
    gm = wave_max(gm);
    if ((tid & 63) == 0) s_red[tid >> 6] = gm;
    __syncthreads();                                         // also orders the window zero-fill
    gm = s_red[0];
#pragma unroll
    for (int w = 1; w < TILED_NT / 64; ++w) gm = fmaxf(gm, s_red[w]);
    if (!(gm > 0.f) || !(gm < 3.0e38f)) return;               // nothing to scatter (uniform across the workgroup)
    int ex;
    (void)frexpf(gm, &ex);                                    // gm in [2^(ex-1), 2^ex)
    const float scale = ldexpf(1.f, 18 - ex), inv = ldexpf(1.f, ex - 18);

    float* gvb = p.grad_value + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + c;
    bf16_t* gvb16 = reinterpret_cast<bf16_t*>(p.grad_value) + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + c;       // (gv_bf16)
    const int lane_off = c * 4;
    // Two dummy pixels per DPP row of the wave, behind the windows.  (The LDS atomic unit works through a ds_add 16 lanes
    // = one pixel = 16 consecutive banks at a time, so the 4 rows of a wave never conflict with each other: measured, a
    // bank-quarter-aware corner order changes nothing.)
    const int dummy = (loff[L] + 2 * r) * 64;

    // ONE loop over the tile's queries of ALL levels (a flat index, the query's own level found by three compares): the
    // per-level loops cost a partly idle last round EACH -- at 640x480 a tile holds 300 + 75 + 20 + 6 queries, i.e. 5 + 2 + 1 + 1
    // = 9 rounds of 64 query slots against 7 for the flat range.
    // (plain SSA scalars and a macro, no lambdas with by-reference captures here: hipcc folds `cond ? *p1 : *p2` over captured
    // variables into a load through a SELECTED ADDRESS, which pins them -- and everything captured alongside -- in scratch memory)
    struct QGeo { int qw, qb, cnt; float iw; };
    auto qgeo = [=](int W, int H, int start) __attribute__((always_inline)) {
        const int qx0 = max(cdiv_i(2 * W * tx - tp.TX, 2 * tp.TX), 0), qx1 = min(max(cdiv_i(2 * W * (tx + 1) - tp.TX, 2 * tp.TX), 0), W);
        const int qy0 = max(cdiv_i(2 * H * ty - tp.TY, 2 * tp.TY), 0), qy1 = min(max(cdiv_i(2 * H * (ty + 1) - tp.TY, 2 * tp.TY), 0), H);
        QGeo r;
        r.qw = qx1 - qx0;
        r.iw = 1.f / (float)max(r.qw, 1);
        r.qb = start + qy0 * W + qx0;
        r.cnt = r.qw * (qy1 - qy0);
        return r;
    };
    const QGeo g0 = qgeo(p.W[0], p.H[0], p.start[0]);
    const QGeo g1 = L > 1 ? qgeo(p.W[L > 1 ? 1 : 0], p.H[L > 1 ? 1 : 0], p.start[L > 1 ? 1 : 0]) : QGeo{0, 0, 0, 1.f};
    const QGeo g2 = L > 2 ? qgeo(p.W[L > 2 ? 2 : 0], p.H[L > 2 ? 2 : 0], p.start[L > 2 ? 2 : 0]) : QGeo{0, 0, 0, 1.f};
    const QGeo g3 = L > 3 ? qgeo(p.W[L > 3 ? 3 : 0], p.H[L > 3 ? 3 : 0], p.start[L > 3 ? 3 : 0]) : QGeo{0, 0, 0, 1.f};
    const int c1 = g0.cnt, c2 = c1 + g1.cnt, c3 = c2 + g2.cnt, c4 = c3 + g3.cnt;
    const int W0 = p.W[0], W1 = p.W[L > 1 ? 1 : 0], W2 = p.W[L > 2 ? 2 : 0], W3 = p.W[L > 3 ? 3 : 0];
    {
        const int nq = (tp.skip & 2) ? 0 : c4;
        // The workgroup's 64 query slots sweep the flat order together (slot s takes queries s, s + 64, ...: 64 consecutive
        // rows of grad_out / offsets / logits per iteration; per-wave or per-slot contiguous runs measured SLOWER at 1280x960).
        // A lane keeps the constants of its current level (tile-row width and its reciprocal, first query, image row pitch) in
        // registers and only re-derives them -- three compares and five 4-way selects, ~30 VALU instructions -- when it steps
        // into the next level.
        // software prefetch: the operands of query i + NSLOT are in flight while query i is scattered
        float n_g = 0.f, n_lg = -3.0e38f, n_ox = 0.f, n_oy = 0.f, n_rx = 0.f, n_ry = 0.f;
        int cbase = 0, cend = 0, cqw = 1, cqb = 0, cW = 0;      // cursor: flat range [cbase, cend) of the current level, ...
        float ciw = 1.f;
#define DV_LEVEL(I)                                                                                                   \
        do {                                                                                                          \
            const int fi_ = (I);                                                                                      \
            const bool s1_ = L > 1 && fi_ >= c1, s2_ = L > 2 && fi_ >= c2, s3_ = L > 3 && fi_ >= c3;                    \
            cbase = s3_ ? c3 : s2_ ? c2 : s1_ ? c1 : 0;                                                               \
            cend = s3_ ? c4 : s2_ ? c3 : s1_ ? c2 : c1;                                                               \
            cqw = s3_ ? g3.qw : s2_ ? g2.qw : s1_ ? g1.qw : g0.qw;                                                    \
            cqb = s3_ ? g3.qb : s2_ ? g2.qb : s1_ ? g1.qb : g0.qb;                                                    \
            cW = s3_ ? W3 : s2_ ? W2 : s1_ ? W1 : W0;                                                                 \
            ciw = s3_ ? g3.iw : s2_ ? g2.iw : s1_ ? g1.iw : g0.iw;                                                    \
        } while (0)
#define DV_FETCH(I)                                                                                                   \
        do {                                                                                                          \
            const int li_ = (I) - cbase;                                                                              \
            const int iy_ = idiv_small(li_, cqw, ciw);                                                                \
            const int q_ = cqb + iy_ * cW + (li_ - iy_ * cqw);                                                        \
            const uint32_t row_ = (uint32_t)(row0 + q_);                                                              \
            n_g = bf2f(ldg32<TQ>(p.grad_out, row_ * g_row + g_lane));                                                 \
            const uint32_t oxy_ = ldg32<uint32_t>(p.q1, row_ * q_row + of_lane);   /* (off_x, off_y) bf16 pair */      \
            n_lg = has ? q_lo<QH>((uint32_t)ldg32<TQ>(p.q1, row_ * q_row + lg_lane)) : -3.0e38f;                      \
            n_ox = q_lo<QH>(oxy_);                                                                                    \
            n_oy = q_hi<QH>(oxy_);                                                                                    \
            const float2 rf_ = ldg32<float2>(p.ref, (uint32_t)q_ * rf_q + rf_lane);                                   \
            n_rx = rf_.x; n_ry = rf_.y;                                                                               \
        } while (0)
        if (slot < nq) { DV_LEVEL(slot); DV_FETCH(slot); }
#pragma unroll 1
        for (int i = slot; i < nq; i += NSLOT) {
            const float gs = n_g * scale, lg = n_lg, ox = n_ox, oy = n_oy, rx = n_rx, ry = n_ry;
            if (i + NSLOT < nq) {
                if (i + NSLOT >= cend) DV_LEVEL(i + NSLOT);
                DV_FETCH(i + NSLOT);
            }
            // ---- lane j (< L*P) of the slot prepares sample point j: softmax over the DPP row, then its 4 corners ----
            const float mx = row16_max(lg);
            const float e = __expf(lg - mx);                             // lanes without a point carry -3e38: e = 0
            const float aj = e * __builtin_amdgcn_rcpf(row16_sum(e));    // 1-ulp reciprocal: the weights feed a 2^-18 fixed point
            const float px = fmaf(rx, WlF, ox - 0.5f), py = fmaf(ry, HlF, oy - 0.5f);
            const float x0f = floorf(px), y0f = floorf(py), fx = px - x0f, fy = py - y0f;
            const int x0 = (int)x0f, y0 = (int)y0f;                     // v_cvt saturates; every use below is an unsigned range test
            const int lx0 = x0 - lwx0, ly0 = y0 - lwy0;
            const bool wx_in[2] = {(unsigned)lx0 < (unsigned)lww, (unsigned)(lx0 + 1) < (unsigned)lww};
            const bool wy_in[2] = {(unsigned)ly0 < (unsigned)lwh, (unsigned)(ly0 + 1) < (unsigned)lwh};
            const bool ix_in[2] = {(unsigned)x0 < (unsigned)Wl, (unsigned)(x0 + 1) < (unsigned)Wl};
            const bool iy_in[2] = {(unsigned)y0 < (unsigned)Hl, (unsigned)(y0 + 1) < (unsigned)Hl};
            const float wxv[2] = {1.f - fx, fx}, wyv[2] = {(1.f - fy) * aj, fy * aj};
            const int base = lo + ly0 * lwp + lx0;           // window slot corner 0 has (or would have)
            // Per sample: the LDS byte address of the LEFT pixel of its upper and of its lower corner row, and corner weights
            // that are 0 when the corner has nothing for the window (outside it, or outside the image -- at the coarse levels
            // a +-4 px offset leaves an 8x10 map all the time).  The right corner is the next pixel in memory (+64 B, the
            // ds_add's immediate offset: one address per corner ROW).  A zero weight adds nothing WHEREVER it lands, so the
            // address only has to stay inside the allocation: a row with one corner in the window keeps its address (the
            // other corner falls on the neighbouring pixel in memory -- the pad pixel in front of the windows when that is
            // pixel -1), a row with none goes to the DPP row's two dummy pixels behind the windows.  The accumulate loop is
            // branch-free; the rare in-image-but-outside-the-window corners are added with global atomics in a wave-voted
            // second pass, and the result never depends on the halo.
            float cw[4], wgt[4];
            bool far = false;                                // (mask arithmetic only; the far weights are formed in the rare branch)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wgt[k] = wyv[k >> 1] * wxv[k & 1];
                const bool inw = wx_in[k & 1] && wy_in[k >> 1] && has;
                cw[k] = inw ? wgt[k] : 0.f;
                far |= ix_in[k & 1] && iy_in[k >> 1] && has && !inw;      // (the windows are clipped to the image: inw implies inimg)
            }
            const bool any_x = (wx_in[0] || wx_in[1]) && has;
            const int arow[2] = {(wy_in[0] && any_x) ? base * 64 : dummy, (wy_in[1] && any_x) ? (base + lwp) * 64 : dummy};
            // ---- broadcast sample by sample to the channel lanes; int32 LDS accumulate ----
            // 5 VALU + 2 LDS ops per corner row: v_add_u32_dpp (the owner lane ships BYTE addresses, so the broadcast -- DPP
            // row_newbcast, no ds_bpermute on the pipe the atomics use -- folds into the address add), and per corner
            // v_mul_f32_dpp (weight broadcast folded), v_cvt_rpi, ds_add_u32
            row_points<LP>(arow, cw, [&](int off, float w0, float w1) {
                int* a = reinterpret_cast<int*>(reinterpret_cast<char*>(win) + (off + lane_off));
                atomicAdd(a, cvt_rpi(w0 * gs));
                atomicAdd(a + 16, cvt_rpi(w1 * gs));
            });
            // In-image corners outside the window: global atomics, one far POINT at a time (its owner lane found by ballot, its
            // corner data read with v_readlane, applied by the 16 channel lanes of its row).  Cost is proportional to the number
            // of far points: 1 % of them (learned offsets drifting past the halo) used to send half of all waves through a
            // 64-step loop.
            unsigned long long fmask = __ballot(far);
            if (fmask) {
                const int gp0 = startl + y0 * Wl + x0;
                const float gsi = gs * inv;
                float cfar[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    cfar[k] = (ix_in[k & 1] && iy_in[k >> 1] && has && !(wx_in[k & 1] && wy_in[k >> 1])) ? wgt[k] : 0.f;
                while (fmask) {
                    const int src = __ffsll((long long)fmask) - 1;
                    fmask &= fmask - 1;
                    const bool mine = ((tid & 63) >> 4) == (src >> 4);
                    const int gpb = __builtin_amdgcn_readlane(gp0, src), wlr = __builtin_amdgcn_readlane(Wl, src);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cfar[k]), src));
                        if (mine && wv != 0.f) {
                            const int64_t pe = (int64_t)(gpb + (k & 1) + (k >> 1) * wlr) * p.gs_s;
                            if (p.gv_bf16) gv16_add(gvb16 + pe, c, wv * gsi);
                            else atomicAdd(gvb + pe, wv * gsi);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tp.skip & 4) return;
    // flush.  The window geometry is RECOMPUTED here from an opaque copy of the tile index (a handful of scalar ops): kept
    // live across the accumulate loop, its ~25 uniform values push the kernel past the SGPR file and into scratch memory.
    int txf = tx, tyf = ty;
    asm volatile("" : "+s"(txf), "+s"(tyf));
    int lofff = 1;                                            // (behind the pad pixel)
    // (explicit calls with constant level indices: see qgeo above)
    auto flush_level = [&, txf, tyf](int W, int H, int start) __attribute__((always_inline)) {
        const int ax = max((txf * W) / tp.TX - tp.HALO, 0), bx = min(cdiv_i((txf + 1) * W, tp.TX) + tp.HALO, W);
        const int ay = max((tyf * H) / tp.TY - tp.HALO, 0), by = min(cdiv_i((tyf + 1) * H, tp.TY) + tp.HALO, H);
        const int wpf = bx - ax, cnt = wpf * (by - ay) * D;
        const float inv_wp = 1.f / (float)wpf;
        if (p.gv_bf16) {                                      // channel PAIRS: one packed bf16x2 atomic per nonzero pair
            bf16_t* gl16 = reinterpret_cast<bf16_t*>(p.grad_value) + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + (int64_t)start * p.gs_s;
            const int2* w2 = reinterpret_cast<const int2*>(win + lofff * D);
            for (int i = tid; i < (cnt >> 1); i += TILED_NT) {
                const int2 v = w2[i];
                if (v.x | v.y) {
                    const int cc = (i & 7) * 2, pix = i >> 3;
                    const int py = idiv_small(pix, wpf, inv_wp);
                    const int xx = ax + pix - py * wpf, yy = ay + py;
                    gv16_add2(gl16 + (int64_t)(yy * W + xx) * p.gs_s + cc, (float)v.x * inv, (float)v.y * inv);
                }
            }
        } else {
        float* gl = p.grad_value + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + (int64_t)start * p.gs_s;
        for (int i = tid; i < cnt; i += TILED_NT) {
            const int v = win[lofff * D + i];
            if (v != 0) {
                const int cc = i & 15, pix = i >> 4;
                const int py = idiv_small(pix, wpf, inv_wp);
                const int xx = ax + pix - py * wpf, yy = ay + py;
                atomicAdd(gl + (int64_t)(yy * W + xx) * p.gs_s + cc, (float)v * inv);
            }
        }
        }
        lofff += wpf * (by - ay);
    };
    flush_level(p.W[0], p.H[0], p.start[0]);
    if constexpr (L > 1) flush_level(p.W[1], p.H[1], p.start[1]);
    if constexpr (L > 2) flush_level(p.W[2], p.H[2], p.start[2]);
    if constexpr (L > 3) flush_level(p.W[3], p.H[3], p.start[3]);
}

#ifdef POET_PROBE_KERNELS
#include "../../profiles/probes/kernels/msda_win.inc"
#endif

#ifdef POET_PROBE_KERNELS
#include "../../profiles/probes/kernels/msda_dv_mfma.inc"
#endif

// host: pick the tile grid / halo so the windows fit the LDS budget; returns window PIXELS of the largest tile (0 = no plan).
// px_budget: pixels that fit; extra_px: pixels reserved behind the windows.
// int32 windows of one workgroup (dynamic LDS; + pad / dummy pixels and a few static words: under the CU's 160 KB)
constexpr int POET_DV_LDS_BYTES = 157 * 1024;
static size_t tile_footprint(const MsdaP& p, int L, int TX, int TY, int halo, size_t* max_q) {
    size_t worst = 0, worst_q = 0;
    for (int ty = 0; ty < TY; ++ty)
        for (int tx = 0; tx < TX; ++tx) {
            size_t px = 0, nq = 0;
            for (int l = 0; l < L; ++l) {
                const int W = p.W[l], H = p.H[l];
                const int ax = max((tx * W) / TX - halo, 0), bx = min(cdiv((int64_t)(tx + 1) * W, TX) + halo, W);
                const int ay = max((ty * H) / TY - halo, 0), by = min(cdiv((int64_t)(ty + 1) * H, TY) + halo, H);
                px += (size_t)(bx - ax) * (by - ay);
                nq += (size_t)(cdiv((int64_t)W, TX) + 1) * (cdiv((int64_t)H, TY) + 1);      // queries whose centre falls in the tile (upper bound)
            }
            worst = max(worst, px);
            worst_q = max(worst_q, nq);
        }
    if (max_q) *max_q = worst_q;
    return worst;
}
// The FEWEST tiles whose windows (tile + halo of every level) fit the pixel budget -- TX and TY chosen independently: larger
// tiles mean fewer halo pixels to zero and flush and fewer samples that leave the window (measured at 1280x960: 8x8 tiles
// 1152 us, 4x8 970 us, 6x5 975 us; at 640x480 2x2 458 us against 606 us for 4x4).  Ties go to the smaller footprint.
// Halo first, tile count second: a sample whose corner leaves the window takes the slow path (wave-voted global atomics), and with
// n_points = 4 the initial sampling pattern reaches 4 px (+ 1 for the right / lower bilinear corner), so a halo of 5 keeps it inside --
// measured per launch: 640x480 2 x 2 tiles 383 us at halo 4, 366 at 5, 380 at 6; 1280x960 the fewest tiles at halo 4 (8 x 3) 974 us, the
// same 5 x 6 tiling 1110 us at halo 4 and 833 at halo 5 (8 x 4: 828, 6 x 6: 867).  `halos`: the values tried, in order.
static size_t plan_tiles_px(const MsdaP& p, int L, TileP& tp, size_t px_budget, size_t extra_px, const int* halos, int n_halos) {
    for (int hi = 0; hi < n_halos; ++hi) {
        const int halo = halos[hi];
        int best_tx = 0, best_ty = 0;
        size_t best_px = 0;
        const int mx = min(16, max(p.W[0] / 4, 1)), my = min(16, max(p.H[0] / 4, 1));
        for (int TY = 1; TY <= my; ++TY)
            for (int TX = 1; TX <= mx; ++TX) {
                if (best_tx && TX * TY > best_tx * best_ty) continue;
                size_t q = 0;
                const size_t px = tile_footprint(p, L, TX, TY, halo, &q);
                // int32 fixed-point headroom of the scatter (2^18 per unit-weight contribution of the tile's largest gradient, the
                // weights of one (query, head) sum to <= 1): queries per tile x (2^18 + rounding) must stay below 2^31
                if (px + extra_px > px_budget || q > 8000) continue;
                if (!best_tx || TX * TY < best_tx * best_ty || px < best_px) { best_tx = TX; best_ty = TY; best_px = px; }
            }
        if (best_tx) { tp.TX = best_tx; tp.TY = best_ty; tp.HALO = halo; tp.skip = 0; return best_px + extra_px; }
    }
    return 0;
}
static size_t plan_tiles(const MsdaP& p, int L, TileP& tp) {
    if (p.D != 16) return 0;
#ifdef POET_PROBE_KERNELS                                        // (probe builds only: re-read per launch by profiles/probes/dv_tiles_bench.py)
    if (const char* e = getenv("POET_DV_TILES")) {               // experiment: "TX,TY,HALO" (rejected when it does not fit)
        int tx = 0, ty = 0, halo = 0;
        if (sscanf(e, "%d,%d,%d", &tx, &ty, &halo) == 3 && tx > 0 && ty > 0 && halo >= 0) {
            size_t worst = 0;
            for (int y = 0; y < ty; ++y)
                for (int x = 0; x < tx; ++x) {
                    size_t px = 0;
                    for (int l = 0; l < L; ++l) {
                        const int W = p.W[l], H = p.H[l];
                        const int ax = max((x * W) / tx - halo, 0), bx = min(cdiv((int64_t)(x + 1) * W, tx) + halo, W);
                        const int ay = max((y * H) / ty - halo, 0), by = min(cdiv((int64_t)(y + 1) * H, ty) + halo, H);
                        px += (size_t)(bx - ax) * (by - ay);
                    }
                    worst = max(worst, px);
                }
            if (worst + 9 <= (size_t)(POET_DV_LDS_BYTES / 64)) { tp.TX = tx; tp.TY = ty; tp.HALO = halo; tp.skip = 0; return (worst + 9) * 64; }
        }
    }
#endif
    // 150 KB of int32 windows (one 1024-thread workgroup per CU) + 1 pad pixel + 8 dummy pixels (see the kernel)
    static const int halos[3] = {5, 4, 2};
    return plan_tiles_px(p, L, tp, POET_DV_LDS_BYTES / (16 * 4), 9, halos, 3) * 16 * 4;
}

template <typename TQ, int L>
static bool launch_dv_tiled(const MsdaP& p, int P, hipStream_t st) {
    if constexpr (sizeof(TQ) != 2) return false;
    else {
    if (P != 4 || L * P > 16 || p.D != 16) return false;
    // the kernel addresses grad_out / the offset|logit rows / the reference points with 32-bit byte offsets
    const int64_t rows = (int64_t)p.N * p.Lq;
    if (rows * p.M * p.D * 2 >= (1ll << 32) || rows * p.ldq * 2 >= (1ll << 32) || (p.ldq & 1) ||
        ((int64_t)(p.N - 1) * p.ref_bs + (int64_t)p.Lq * L * 2) * 4 >= (1ll << 32) || (p.ref_bs & 1)) return false;
    TileP tp{};
    const size_t lds = plan_tiles(p, L, tp);
    static const int no_tiled = [] { const char* e = getenv("POET_NO_TILED_SCATTER"); return e && atoi(e) ? 1 : 0; }();     // (A/B aid, read once)
    if (no_tiled) return false;
    tp.skip = 0;
#ifdef POET_PROBE_KERNELS                                        // timing-breakdown aid of the probe scripts (re-read per launch there)
    { const char* e = getenv("POET_DV_SKIP"); tp.skip = e ? atoi(e) : 0; }
#endif
    if (!lds) return false;
    const dim3 grid(tp.TX * tp.TY * p.M * p.N), blk(TILED_NT);
    static const int order = [] { const char* e = getenv("POET_DV_ORDER"); return e ? atoi(e) : 1; }();         // (A/B aid, read once: 0 = the (tiles, M, N) grid order)
    tp.order = order;
#define POET_DV_LAUNCH(QH_) do { auto kern = msda_bwd_dv_tiled_kernel<TQ, L, QH_>;                                                          \
        static unsigned long long attr_done = 0;                             /* per instantiation and device */                            \
        lds_attr_once(reinterpret_cast<const void*>(kern), POET_DV_LDS_BYTES + 2048, attr_done);                                            \
        hipLaunchKernelGGL(kern, grid, blk, lds, st, p, tp); } while (0)
    if (p.q_f16) POET_DV_LAUNCH(true); else POET_DV_LAUNCH(false);
#undef POET_DV_LAUNCH
    return true;
    }
}

#ifdef POET_PROBE_KERNELS
#include "../../profiles/probes/kernels/msda_probe_launch.inc"
#endif

// ---- generic kernels: any n_levels <= MAXL, any n_points (L x P <= 64) -------------------------------------------------------
// `--num_feature_levels`, `--enc_n_points`, `--dec_n_points` are free parameters of the reference (main.py:71,100-101); the
// specialised kernels above are instantiated for L <= 4 and P in {1, 2, 4}.  Everything else runs here: L and P are run-time
// values, one thread = 8 channels of one (image, query, head) as in msda_fwd_kernel, scalar operand loads (no alignment
// requirement on the offset | logit rows), the softmax of the fused form in two passes over the logits instead of a register
// array, and -- backward -- d(attention) of every point parked in LDS until the softmax Jacobian's dot product is known.
// Correctness path, not a tuned one.
struct GenLevel { int H, W, start; };
__device__ __forceinline__ GenLevel gen_level(const MsdaP& p, int l) {
    GenLevel g{p.H[0], p.W[0], p.start[0]};                 // select chain: a run-time index into the by-value argument goes to scratch
#pragma unroll
    for (int k = 1; k < MAXL; ++k)
        if (l == k) { g.H = p.H[k]; g.W = p.W[k]; g.start = p.start[k]; }
    return g;
}
struct GenCorner {
    int pix[4];                                              // pixel index (start + y W + x) of the 4 corners, clamped into the map
    float w[4];                                              // bilinear weight, 0 where the corner lies outside the map
    float fx, fy;
    float m[4];                                              // validity 0 / 1
};
__device__ __forceinline__ GenCorner gen_corner(float px, float py, const GenLevel& g) {
    GenCorner c;
    px = fminf(fmaxf(px, -2.f), (float)g.W + 1.f);           // (every corner of a sample beyond these bounds is outside the map anyway)
    py = fminf(fmaxf(py, -2.f), (float)g.H + 1.f);
    const float x0f = floorf(px), y0f = floorf(py);
    c.fx = px - x0f; c.fy = py - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float vx0 = (x0 >= 0 && x0 < g.W) ? 1.f : 0.f, vx1 = (x1 >= 0 && x1 < g.W) ? 1.f : 0.f;
    const float vy0 = (y0 >= 0 && y0 < g.H) ? 1.f : 0.f, vy1 = (y1 >= 0 && y1 < g.H) ? 1.f : 0.f;
    const int xc0 = min(max(x0, 0), g.W - 1), xc1 = min(max(x1, 0), g.W - 1), yc0 = min(max(y0, 0), g.H - 1), yc1 = min(max(y1, 0), g.H - 1);
    c.m[0] = vy0 * vx0; c.m[1] = vy0 * vx1; c.m[2] = vy1 * vx0; c.m[3] = vy1 * vx1;
    c.w[0] = (1.f - c.fy) * (1.f - c.fx) * c.m[0]; c.w[1] = (1.f - c.fy) * c.fx * c.m[1];
    c.w[2] = c.fy * (1.f - c.fx) * c.m[2];         c.w[3] = c.fy * c.fx * c.m[3];
    c.pix[0] = g.start + yc0 * g.W + xc0; c.pix[1] = g.start + yc0 * g.W + xc1;
    c.pix[2] = g.start + yc1 * g.W + xc0; c.pix[3] = g.start + yc1 * g.W + xc1;
    return c;
}
// sample position (pixel units) and attention weight of point (l, i) of (row, head m); mx / inv: the fused form's softmax statistics
template <typename TQ, bool FUSED>
__device__ __forceinline__ void gen_point(const MsdaP& p, int L, int P, int64_t row, int n, int q, int m, int l, int i, const GenLevel& g,
                                          float mx, float inv, float& px, float& py, float& aw) {
    const int LP = L * P;
    if constexpr (FUSED) {
        const TQ* qrow = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq;
        const TQ* op = qrow + ((int64_t)(m * L + l) * P + i) * 2;
        const float* rp = p.ref + (int64_t)n * p.ref_bs + ((int64_t)q * L + l) * 2;
        px = io<TQ>::ld(op) + (rp[0] * (float)g.W - 0.5f);
        py = io<TQ>::ld(op + 1) + (rp[1] * (float)g.H - 0.5f);
        aw = __expf(io<TQ>::ld(qrow + p.logit_col + m * LP + l * P + i) - mx) * inv;
    } else {
        const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + (((row * p.M + m) * L + l) * P + i) * 2;
        px = io<TQ>::ld(lp) * (float)g.W - 0.5f;
        py = io<TQ>::ld(lp + 1) * (float)g.H - 0.5f;
        aw = io<TQ>::ld(reinterpret_cast<const TQ*>(p.q2) + (row * p.M + m) * LP + l * P + i);
    }
}
template <typename TQ, bool FUSED>
__device__ __forceinline__ void gen_softmax_stats(const MsdaP& p, int LP, int64_t row, int m, float& mx, float& inv) {
    mx = 0.f; inv = 1.f;
    if constexpr (FUSED) {
        const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq + p.logit_col + m * LP;
        mx = io<TQ>::ld(lp);
        for (int i = 1; i < LP; ++i) mx = fmaxf(mx, io<TQ>::ld(lp + i));
        float s = 0.f;
        for (int i = 0; i < LP; ++i) s += __expf(io<TQ>::ld(lp + i) - mx);
        inv = 1.f / s;
    }
}

template <typename TV, typename TQ, bool FUSED>
__global__ __launch_bounds__(256) void msda_gen_fwd_kernel(const MsdaP p, int L, int P) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.total) return;
    const int64_t row = t / p.groups;
    const int c8 = (int)(t - row * p.groups);
    const int m = c8 / p.tpg, dsub = c8 - m * p.tpg;
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);
    float mx, inv;
    gen_softmax_stats<TQ, FUSED>(p, L * P, row, m, mx, inv);
    const TV* vb = reinterpret_cast<const TV*>(p.value) + (int64_t)n * p.vs_n + (int64_t)m * p.vs_m + dsub * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const GenLevel g = gen_level(p, l);
#pragma unroll 1
        for (int i = 0; i < P; ++i) {
            float px, py, aw;
            gen_point<TQ, FUSED>(p, L, P, row, n, q, m, l, i, g, mx, inv, px, py, aw);
            const GenCorner c = gen_corner(px, py, g);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
                vec<TV, 8>::ld(vb + (int64_t)c.pix[k] * p.vs_s, v);
                const float w = aw * c.w[k];
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) acc[ch] = fmaf(w, v[ch], acc[ch]);
            }
        }
    }
    vec<TQ, 8>::st(reinterpret_cast<TQ*>(p.out) + row * ((int64_t)p.M * p.D) + m * p.D + dsub * 8, acc);
}

// backward of the same: p.parts bit 0 = d(offsets | logits) / d(loc), d(attn); bit 1 = the d(value) scatter (fp32 memory-side
// atomics, or packed bf16x2 ones when grad_value is bf16).  Dynamic LDS: L x P floats per thread (fused form).
template <typename TV, typename TQ, bool FUSED>
__global__ __launch_bounds__(256) void msda_gen_bwd_kernel(const MsdaP p, int L, int P) {
    extern __shared__ float gen_da[];                        // [L P][256]
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t0 < p.total;
    const int64_t t = live ? t0 : p.total - 1;               // (dead lanes shadow the last thread: the group reductions below are wave-wide)
    const int64_t row = t / p.groups;
    const int c8 = (int)(t - row * p.groups);
    const int m = c8 / p.tpg, dsub = c8 - m * p.tpg;
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);
    const int LP = L * P;
    float mx, inv;
    gen_softmax_stats<TQ, FUSED>(p, LP, row, m, mx, inv);
    const TV* vb = reinterpret_cast<const TV*>(p.value) + (int64_t)n * p.vs_n + (int64_t)m * p.vs_m + dsub * 8;
    float* gvb = p.grad_value + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + dsub * 8;
    bf16_t* gvb16 = reinterpret_cast<bf16_t*>(p.grad_value) + (int64_t)n * p.gs_n + (int64_t)m * p.gs_m + dsub * 8;
    float g[8];
    vec<TQ, 8>::ld(reinterpret_cast<const TQ*>(p.grad_out) + row * ((int64_t)p.M * p.D) + m * p.D + dsub * 8, g);
    const bool owner = live && dsub == 0 && (p.parts & 1);
    float dot = 0.f;
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const GenLevel gl = gen_level(p, l);
#pragma unroll 1
        for (int i = 0; i < P; ++i) {
            float px, py, aw;
            gen_point<TQ, FUSED>(p, L, P, row, n, q, m, l, i, gl, mx, inv, px, py, aw);
            const GenCorner c = gen_corner(px, py, gl);
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
                vec<TV, 8>::ld(vb + (int64_t)c.pix[k] * p.vs_s, v);
                float s = 0.f;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) s = fmaf(g[ch], v[ch], s);
                d[k] = s * c.m[k];
                if ((p.parts & 2) && live && c.w[k] != 0.f) {
                    const float w = aw * c.w[k];
                    const int64_t off = (int64_t)c.pix[k] * p.gs_s;
                    if (p.gv_bf16) {
#pragma unroll
                        for (int ch = 0; ch < 8; ch += 2) gv16_add2(gvb16 + off + ch, w * g[ch], w * g[ch + 1]);
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 8; ++ch) atomicAdd(gvb + off + ch, w * g[ch]);
                    }
                }
            }
            if (p.parts & 1) {
                const float fx = c.fx, fy = c.fy;
                float da = (1.f - fy) * ((1.f - fx) * d[0] + fx * d[1]) + fy * ((1.f - fx) * d[2] + fx * d[3]);
                float dx = aw * ((1.f - fy) * (d[1] - d[0]) + fy * (d[3] - d[2]));
                float dy = aw * ((1.f - fx) * (d[2] - d[0]) + fx * (d[3] - d[1]));
                da = group_sum(da, p.tpg); dx = group_sum(dx, p.tpg); dy = group_sum(dy, p.tpg);
                if (owner) {
                    if constexpr (FUSED) {
                        TQ* gp = reinterpret_cast<TQ*>(p.g1) + row * p.ldg + ((int64_t)(m * L + l) * P + i) * 2;
                        io<TQ>::st(gp, dx); io<TQ>::st(gp + 1, dy);          // d/d(offset) = (dpx, dpy): the W, H factors cancel
                        gen_da[(l * P + i) * 256 + threadIdx.x] = da;
                        dot = fmaf(aw, da, dot);
                    } else {
                        TQ* gp = reinterpret_cast<TQ*>(p.g1) + (((row * p.M + m) * L + l) * P + i) * 2;
                        io<TQ>::st(gp, dx * (float)gl.W); io<TQ>::st(gp + 1, dy * (float)gl.H);
                        io<TQ>::st(reinterpret_cast<TQ*>(p.g2) + (row * p.M + m) * LP + l * P + i, da);
                    }
                }
            }
        }
    }
    if constexpr (FUSED) {
        if (owner) {                                         // softmax Jacobian: d logit_j = a_j (d a_j - sum_k a_k d a_k)
            const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq + p.logit_col + m * LP;
            TQ* gp = reinterpret_cast<TQ*>(p.g1) + row * p.ldg + p.logit_col + m * LP;
            for (int j = 0; j < LP; ++j) {
                const float aj = __expf(io<TQ>::ld(lp + j) - mx) * inv;
                io<TQ>::st(gp + j, aj * (gen_da[j * 256 + threadIdx.x] - dot));
            }
        }
    }
}

template <typename TV, typename TQ, bool FUSED, bool BWD>
static void launch_gen(const MsdaP& p, int L, int P, hipStream_t st) {
    const dim3 grid(cdiv(p.total, 256)), block(256);
    if constexpr (BWD) {
        const size_t lds = FUSED ? (size_t)L * P * 256 * sizeof(float) : 0;
        auto kern = msda_gen_bwd_kernel<TV, TQ, FUSED>;
        static unsigned long long attr_done = 0;
        lds_attr_once(reinterpret_cast<const void*>(kern), 64 * 256 * 4, attr_done);
        hipLaunchKernelGGL(kern, grid, block, lds, st, p, L, P);
    } else {
        hipLaunchKernelGGL((msda_gen_fwd_kernel<TV, TQ, FUSED>), grid, block, 0, st, p, L, P);
    }
}

// fp16 offsets | logits (MsdaP::q_f16) are read by the shared-geometry gathers and the LDS-tiled scatters only: a launch that would
// fall through to a general kernel (which reads q1 as bf16) is refused instead
static thread_local bool g_f16_refused = false;

template <typename TV, typename TQ, int L, bool FUSED, bool BWD>
static void launch_p(const MsdaP& p, int P, hipStream_t st) {
    dim3 grid(cdiv(p.total, 256)), block(256);
    dim3 gridv(cdiv(p.total * 8, 256)), gridf(cdiv(p.total * 2, 256));
    bool dv_done = !(p.parts & 2);
    if constexpr (BWD && FUSED) {
#ifdef POET_PROBE_KERNELS
        if (!dv_done && p.grid_queries && p.Lq == p.S) dv_done = launch_dv_mfma<TQ, L>(p, P, st);
#endif
        if (!dv_done && p.grid_queries && p.Lq == p.S) dv_done = launch_dv_tiled<TQ, L>(p, P, st);
    }
    if (p.q_f16 && BWD && !dv_done) { g_f16_refused = true; return; }
    if constexpr (BWD && sizeof(TQ) == 4) {     // fp32 offsets | logits = the decoder's form: the general kernels run both parts -- as one launch
        static const bool no_both = [] { const char* e = getenv("POET_MSDA_NO_BOTH"); return e && atoi(e); }();       // (A/B aid, read once)
        if (!dv_done && (p.parts & 1) && P == 4 && !no_both && !p.v_f16 && (int64_t)gridv.x + gridf.x < (1ll << 20)) {
            hipLaunchKernelGGL((msda_bwd_both_kernel<TV, TQ, L, 4, FUSED>), dim3(gridv.x + gridf.x), block, 0, st, p, (int)gridv.x);
            return;
        }
    }
    if (BWD && !dv_done) {
        if (P == 4) hipLaunchKernelGGL((msda_bwd_dv_kernel<TQ, L, 4, FUSED>), gridv, block, 0, st, p);
        else if (P == 2) hipLaunchKernelGGL((msda_bwd_dv_kernel<TQ, L, 2, FUSED>), gridv, block, 0, st, p);
        else hipLaunchKernelGGL((msda_bwd_dv_kernel<TQ, L, 1, FUSED>), gridv, block, 0, st, p);
    }
    if (BWD && !(p.parts & 1)) return;
#ifdef POET_PROBE_KERNELS
    if constexpr (FUSED) {
        if (launch_win<TV, TQ, L, BWD>(p, P, st)) return;
    }
#endif
    if constexpr (FUSED && L == 4 && sizeof(TV) == 2 && sizeof(TQ) == 2) {
        // the encoder's shape: shared-geometry gathers (POET_MSDA_NO_SHARED=1: the general kernels below)
        static const bool no_shared = [] { const char* e = getenv("POET_MSDA_NO_SHARED"); return e && atoi(e); }();       // (A/B aids, read once)
        const int64_t rows = (int64_t)p.N * p.Lq;
        if (!no_shared && P == 4 && p.M == 16 && p.D == 16 && rows < (1ll << 31) && (p.ldq % 8) == 0 && (p.logit_col % 4) == 0 &&
            (p.ref_bs % 2) == 0) {
            // POET_SH_QGROUP=1: one query x 16 heads per wave; default 4 consecutive queries x 4 heads, 4 passes
#ifdef POET_PROBE_KERNELS
            // POET_MSDA_HYBRID=1 (forward, round 4 experiment): coarse levels staged in LDS per (image, head quarter)
            if constexpr (!BWD) {
                const char* hy_ = getenv("POET_MSDA_HYBRID");
                const int lc = (hy_ && atoi(hy_)) ? hyb_first_level(p) : -1;
                if (lc > 0 && p.grid_queries) {
                    HybP hp{lc, p.start[lc], 16};
                    { const char* e = getenv("POET_HYB_CHUNKS"); if (e && atoi(e) > 0) hp.chunks = atoi(e); }
                    const size_t lds = (size_t)(p.S - hp.px_lc) * 128 + (size_t)HYB_WAVES * SH_WAVE;
                    static bool attr_set = false;
                    if (!attr_set) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msda_fwd_hyb_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msda_fwd_hyb_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        attr_set = true;
                    }
                    const dim3 hgrid((unsigned)(p.N * 4 * hp.chunks)), hblk(HYB_WAVES * 64);
                    if (p.q_f16) hipLaunchKernelGGL((msda_fwd_hyb_kernel<true>), hgrid, hblk, lds, st, p, hp);
                    else hipLaunchKernelGGL((msda_fwd_hyb_kernel<false>), hgrid, hblk, lds, st, p, hp);
                    return;
                }
            }
#endif
            // 4 consecutive queries x 4 heads per wave, 4 passes (1, 2, 8 and 16 queries per group measured no better: DESIGN section 9)
            int qg = 4;
            size_t dyn = 0;
#ifdef POET_PROBE_KERNELS
            { const char* e = getenv("POET_SH_QGROUP"); if (e && atoi(e) == 1) qg = 1; }
            { const char* e = getenv("POET_SH_PADLDS"); if (e) dyn = (size_t)atoi(e); }     // experiment: extra LDS per workgroup (lowers the occupancy)
#endif
            const dim3 grid((unsigned)((rows + SH_WAVES * qg - 1) / (SH_WAVES * qg))), blk(SH_WAVES * 64);
#define POET_SH_LAUNCH(Q, H, V) do { if (BWD) hipLaunchKernelGGL((msda_bwd_shared_kernel<Q, H, V>), grid, blk, dyn, st, p); \
                                     else hipLaunchKernelGGL((msda_fwd_shared_kernel<Q, H, V>), grid, blk, dyn, st, p); } while (0)
            if (p.v_f16) { if (qg == 1) POET_SH_LAUNCH(1, true, true); else POET_SH_LAUNCH(4, true, true); }      // (fp16 maps come with fp16 offsets | logits)
            else if (p.q_f16) { if (qg == 1) POET_SH_LAUNCH(1, true, false); else POET_SH_LAUNCH(4, true, false); }
            else if (qg == 1) POET_SH_LAUNCH(1, false, false);
            else POET_SH_LAUNCH(4, false, false);
#undef POET_SH_LAUNCH
            return;
        }
    }
    if (p.q_f16 || p.v_f16) { g_f16_refused = true; return; }
    if (P == 4) {
        if (BWD) hipLaunchKernelGGL((msda_bwd_kernel<TV, TQ, L, 4, FUSED>), gridf, block, 0, st, p);
        else hipLaunchKernelGGL((msda_fwd_kernel<TV, TQ, L, 4, FUSED>), gridf, block, 0, st, p);
    } else if (P == 2) {
        if (BWD) hipLaunchKernelGGL((msda_bwd_kernel<TV, TQ, L, 2, FUSED>), gridf, block, 0, st, p);
        else hipLaunchKernelGGL((msda_fwd_kernel<TV, TQ, L, 2, FUSED>), gridf, block, 0, st, p);
    } else {
        if (BWD) hipLaunchKernelGGL((msda_bwd_kernel<TV, TQ, L, 1, FUSED>), gridf, block, 0, st, p);
        else hipLaunchKernelGGL((msda_fwd_kernel<TV, TQ, L, 1, FUSED>), gridf, block, 0, st, p);
    }
}

template <typename TV, typename TQ, bool FUSED, bool BWD>
static void launch_l(const MsdaP& p, int L, int P, hipStream_t st) {
    if (L > 4 || !(P == 1 || P == 2 || P == 4)) {             // outside the instantiated kernels: the generic ones
        if (p.q_f16 || p.v_f16) { g_f16_refused = true; return; }
        launch_gen<TV, TQ, FUSED, BWD>(p, L, P, st);
        return;
    }
    switch (L) {
        case 1: launch_p<TV, TQ, 1, FUSED, BWD>(p, P, st); break;
        case 2: launch_p<TV, TQ, 2, FUSED, BWD>(p, P, st); break;
        case 3: launch_p<TV, TQ, 3, FUSED, BWD>(p, P, st); break;
        default: launch_p<TV, TQ, 4, FUSED, BWD>(p, P, st); break;
    }
}

template <bool FUSED, bool BWD>
static int dispatch(const MsdaP& p, int L, int P, int v_dtype, int q_dtype, hipStream_t st) {
    {   // the gather kernels address the value maps with 32-bit byte offsets and 24-bit pixel arithmetic
        const int64_t esz = (v_dtype == POET_BF16 || v_dtype == POET_F16) ? 2 : 4;
        const int64_t span = ((int64_t)(p.N - 1) * p.vs_n + (int64_t)(p.M - 1) * p.vs_m + (int64_t)(p.S - 1) * p.vs_s + p.D) * esz;
        POET_CHECK(span < (1ll << 32) && p.vs_s * esz < (1 << 24) && p.S < (1 << 24) && p.vs_n >= 0 && p.vs_m >= 0 && p.vs_s > 0,
                   POET_ERR_UNSUPPORTED, "msda: value maps of %lld bytes exceed the 4 GiB (32-bit offset) limit of the gather kernels",
                   (long long)span);
    }
    g_f16_refused = false;
    if ((v_dtype == POET_BF16 || v_dtype == POET_F16) && q_dtype == POET_F16 && FUSED) {
        MsdaP ph = p;
        ph.q_f16 = 1;
        ph.v_f16 = v_dtype == POET_F16;                       // (round 6) fp16 value maps: the shared-geometry gathers' own format
        launch_l<bf16_t, bf16_t, FUSED, BWD>(ph, L, P, st);
        POET_CHECK(!g_f16_refused, POET_ERR_UNSUPPORTED,
                   "msda_fused: fp16 offsets | logits (q_dtype POET_F16) need the encoder shape (bf16 value maps, 16 heads x 16 channels, 4 levels x 4 points, "
                   "grid queries) served by the shared-geometry gathers and the LDS-tiled scatter");
    }
    else if (v_dtype == POET_BF16 && q_dtype == POET_BF16) launch_l<bf16_t, bf16_t, FUSED, BWD>(p, L, P, st);
    else if (v_dtype == POET_F32 && q_dtype == POET_F32) launch_l<float, float, FUSED, BWD>(p, L, P, st);
    else if (v_dtype == POET_BF16 && q_dtype == POET_F32) launch_l<bf16_t, float, FUSED, BWD>(p, L, P, st);
    else { set_error("msda: unsupported dtype pair v=%d q=%d", v_dtype, q_dtype); return POET_ERR_UNSUPPORTED; }
    return POET_OK;
}

static int fill_common(MsdaP& p, const int64_t* shapes, const int64_t* starts, int N, int S, int M, int D, int L, int P, int Lq) {
    POET_CHECK(N > 0 && S > 0 && M > 0 && Lq > 0, POET_ERR_ARG, "msda: bad dims");
    POET_CHECK(D % 8 == 0 && D >= 8, POET_ERR_UNSUPPORTED, "msda: head dim %d must be a multiple of 8", D);
    POET_CHECK((D / 8) <= 64 && (((D / 8) & ((D / 8) - 1)) == 0), POET_ERR_UNSUPPORTED, "msda: head dim %d/8 must be a power of two", D);
    POET_CHECK(L >= 1 && L <= MAXL, POET_ERR_UNSUPPORTED, "msda: n_levels %d not in 1..%d", L, MAXL);
    POET_CHECK(P >= 1 && L * P <= 64, POET_ERR_UNSUPPORTED, "msda: n_levels x n_points = %d x %d exceeds 64 samples per (query, head)", L, P);
    POET_CHECK(shapes && starts, POET_ERR_ARG, "msda: null shapes");
    int64_t tot = 0;
    for (int l = 0; l < L; ++l) {
        p.H[l] = (int)shapes[2 * l];
        p.W[l] = (int)shapes[2 * l + 1];
        p.start[l] = (int)starts[l];
        POET_CHECK(p.H[l] > 0 && p.W[l] > 0, POET_ERR_ARG, "msda: bad level shape");
        tot += (int64_t)p.H[l] * p.W[l];
    }
    POET_CHECK(tot == S, POET_ERR_ARG, "msda: sum(H*W)=%lld != S=%d", (long long)tot, S);
    p.N = N; p.S = S; p.M = M; p.D = D; p.Lq = Lq;
    p.tpg = D / 8;
    p.groups = M * p.tpg;
    p.total = (int64_t)N * Lq * p.groups;
    return POET_OK;
}

}  // namespace poet

using namespace poet;

extern "C" int poet_msda_fwd(const void* value, const int64_t* shapes, const int64_t* starts, const void* loc,
                             const void* attn, void* out, int N, int S, int M, int D, int L, int P, int Lq,
                             int dtype, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    POET_CHECK(value && loc && attn && out, POET_ERR_ARG, "msda_fwd: null pointer");
    p.value = value; p.vs_n = (int64_t)S * M * D; p.vs_s = (int64_t)M * D; p.vs_m = D;
    p.q1 = loc; p.q2 = attn; p.out = out;
    rc = dispatch<false, false>(p, L, P, dtype, dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_msda_bwd(const void* value, const int64_t* shapes, const int64_t* starts, const void* loc,
                             const void* attn, const void* grad_out, float* grad_value, void* grad_loc, void* grad_attn,
                             int N, int S, int M, int D, int L, int P, int Lq, int dtype, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    POET_CHECK(value && loc && attn && grad_out && grad_value && grad_loc && grad_attn, POET_ERR_ARG, "msda_bwd: null pointer");
    p.value = value; p.vs_n = (int64_t)S * M * D; p.vs_s = (int64_t)M * D; p.vs_m = D;
    p.gs_n = p.vs_n; p.gs_s = p.vs_s; p.gs_m = p.vs_m;
    p.q1 = loc; p.q2 = attn; p.grad_out = grad_out; p.grad_value = grad_value; p.g1 = grad_loc; p.g2 = grad_attn;
    p.parts = 3;
    hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)N * S * M * D, (hipStream_t)stream);
    POET_CHECK(e == hipSuccess, POET_ERR_LAUNCH, "msda_bwd: memset failed: %s", hipGetErrorString(e));
    rc = dispatch<false, true>(p, L, P, dtype, dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}

static int fused_args(MsdaP& p, const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m, const void* offattn,
                      int64_t ldq, int logit_col, const float* ref, int64_t ref_bs, int M, int L, int P) {
    POET_CHECK(value && offattn && ref, POET_ERR_ARG, "msda_fused: null pointer");
    const bool gen = L > 4 || !(P == 1 || P == 2 || P == 4);     // served by the generic kernels: scalar operand loads, no alignment rule
    POET_CHECK((gen || (ldq % 8 == 0 && logit_col % 8 == 0)) && logit_col >= M * L * P * 2 && ldq >= logit_col + M * L * P, POET_ERR_ARG,
               "msda_fused: ldq/logit_col must be multiples of 8 and logits must follow the offsets");
    POET_CHECK(vs_s % 8 == 0 && vs_m % 8 == 0 && vs_n % 8 == 0, POET_ERR_ARG, "msda_fused: value strides must be multiples of 8");
    p.value = value; p.vs_n = vs_n; p.vs_s = vs_s; p.vs_m = vs_m;
    p.q1 = offattn; p.ldq = ldq; p.logit_col = logit_col; p.ref = ref; p.ref_bs = ref_bs;
    return POET_OK;
}

extern "C" int poet_msda_fused_fwd(const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m, const int64_t* shapes,
                                   const int64_t* starts, const void* offattn, int64_t ldq, int logit_col,
                                   const float* ref, int64_t ref_bs, void* out, int N, int S, int M, int D, int L,
                                   int P, int Lq, int v_dtype, int q_dtype, int grid_queries, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    rc = fused_args(p, value, vs_n, vs_s, vs_m, offattn, ldq, logit_col, ref, ref_bs, M, L, P);
    if (rc) return rc;
    POET_CHECK(out, POET_ERR_ARG, "msda_fused_fwd: null out");
    p.out = out;
    p.grid_queries = grid_queries;
    rc = dispatch<true, false>(p, L, P, v_dtype, q_dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_msda_fused_bwd(const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m, const int64_t* shapes,
                                   const int64_t* starts, const void* offattn, int64_t ldq, int logit_col,
                                   const float* ref, int64_t ref_bs, const void* grad_out, void* grad_value,
                                   void* grad_offattn, int64_t ld_grad, int N, int S, int M, int D, int L, int P, int Lq,
                                   int v_dtype, int q_dtype, int gv_dtype, int grid_queries, int parts, const int64_t* gv_strides, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    rc = fused_args(p, value, vs_n, vs_s, vs_m, offattn, ldq, logit_col, ref, ref_bs, M, L, P);
    if (rc) return rc;
    POET_CHECK(grad_out && grad_value && grad_offattn, POET_ERR_ARG, "msda_fused_bwd: null pointer");
    POET_CHECK(gv_dtype == POET_F32 || gv_dtype == POET_BF16, POET_ERR_ARG, "msda_fused_bwd: gv_dtype");
    p.ldg = ld_grad > 0 ? ld_grad : ldq;
    POET_CHECK(p.ldg >= (int64_t)3 * M * L * P && ((p.ldg % 8) == 0 || L > 4 || !(P == 1 || P == 2 || P == 4)), POET_ERR_ARG, "msda_fused_bwd: ld_grad %lld", (long long)p.ldg);
    p.grad_out = grad_out; p.grad_value = reinterpret_cast<float*>(grad_value); p.g1 = grad_offattn;
    p.gv_bf16 = gv_dtype == POET_BF16;
    p.grid_queries = grid_queries;
    p.parts = (parts & 3) ? (parts & 3) : 3;
    // grad_value may be laid out differently from value ((n, s, m) element strides; NULL = the value strides): the decoder scatters
    // straight into the token-major rows its value-projection backward consumes
    p.gs_n = gv_strides ? gv_strides[0] : vs_n; p.gs_s = gv_strides ? gv_strides[1] : vs_s; p.gs_m = gv_strides ? gv_strides[2] : vs_m;
    POET_CHECK(p.gs_s > 0 && p.gs_n >= 0 && p.gs_m >= 0, POET_ERR_ARG, "msda_fused_bwd: grad_value strides");
    if (p.gv_bf16)                                            // packed bf16x2 atomics: channel pairs must be 4-byte aligned
        POET_CHECK((p.gs_n % 2 == 0) && (p.gs_m % 2 == 0) && (p.gs_s % 2 == 0) && D % 2 == 0 && (reinterpret_cast<uintptr_t>(grad_value) & 3) == 0,
                   POET_ERR_UNSUPPORTED, "msda_fused_bwd: a bf16 grad_value needs even strides and a 4-byte aligned base");
    rc = dispatch<true, true>(p, L, P, v_dtype, q_dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}
