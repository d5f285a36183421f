// Elementwise / layout / encoding / optimizer kernels of the PoET hot path (gfx950).
// All HBM-bound byte movers: vectorised 16-B lane accesses, grid-stride where the size warrants it.
#include "common.cuh"

namespace poet {

// ---- add / cast ---------------------------------------------------------------------------------
template <typename TA, typename TB, typename TO>
__global__ __launch_bounds__(256) void add_kernel(const TA* __restrict__ a, const TB* __restrict__ b, TO* __restrict__ o, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float x[8], y[8];
        vec<TA, 8>::ld(a + i, x);
        vec<TB, 8>::ld(b + i, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        vec<TO, 8>::st(o + i, x);
    } else {
        for (int64_t j = i; j < n; ++j) io<TO>::st(o + j, io<TA>::ld(a + j) + io<TB>::ld(b + j));
    }
}

// ---- GELU (+ dropout) of the FFN: `activation="gelu"` (deformable_transformer.py:347-355; F.gelu = the erf form) --------------------
// The ReLU FFN has its activation and dropout in the Linear's epilogue and its backward gate in the next product's loader (the sign of
// the stored output decides both).  GELU's derivative needs the PRE-activation, so this form keeps it: y = dropout(gelu(x)) and
// d(x) = d(y) * mask / (1 - p) * gelu'(x) are their own element-wise passes.  One dropout counter per element index, redrawn in backward.
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * __expf(-0.5f * x * x) * 0.39894228040143268f;
}

// TXI: storage of the kept pre-activation (fp32 where the bf16 policy hands the Linear's accumulators over unrounded), T: of y / dy / dx
template <typename TXI, typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const TXI* __restrict__ x, const T* __restrict__ dy, T* __restrict__ o, int64_t n,
                                                   uint32_t thresh, float dscale, uint32_t seed, const uint32_t* __restrict__ seed_dev) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const uint32_t sd = seed ^ (seed_dev ? *seed_dev * 0x9E3779B1u : 0u);
    if (i + 8 <= n) {
        float a[8], g[8];
        vec<TXI, 8>::ld(x + i, a);
        if constexpr (BWD) vec<T, 8>::ld(dy + i, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float keep = thresh ? (drop_keep(sd, (uint32_t)(i + e), thresh) ? dscale : 0.f) : 1.f;
            a[e] = BWD ? g[e] * keep * gelu_df(a[e]) : gelu_f(a[e]) * keep;
        }
        vec<T, 8>::st(o + i, a);
    } else {
        for (int64_t j = i; j < n; ++j) {
            const float keep = thresh ? (drop_keep(sd, (uint32_t)j, thresh) ? dscale : 0.f) : 1.f;
            const float a = io<TXI>::ld(x + j);
            io<T>::st(o + j, BWD ? io<T>::ld(dy + j) * keep * gelu_df(a) : gelu_f(a) * keep);
        }
    }
}

// the head of the encoder in the bf16 policy: the stream's operand copy x16 = bf16(x) and the first layer's query operand
// q = bf16(x + pos) in ONE pass over the fp32 stream (was a cast and an add: the 104 MB stream read twice); same arithmetic, bit for bit
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ sum,
                                                       bf16_t* __restrict__ a16, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float x[8], y[8];
        vec<float, 8>::ld(a + i, x);
        vec<bf16_t, 8>::ld(b + i, y);
        vec<bf16_t, 8>::st(a16 + i, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        vec<bf16_t, 8>::st(sum + i, x);
    } else {
        for (int64_t j = i; j < n; ++j) {
            const float v = a[j];
            io<bf16_t>::st(a16 + j, v);
            io<bf16_t>::st(sum + j, v + io<bf16_t>::ld(b + j));
        }
    }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float x[8];
        vec<S, 8>::ld(s + i, x);
        vec<D, 8>::st(d + i, x);
    } else {
        for (int64_t j = i; j < n; ++j) io<D>::st(d + j, io<S>::ld(s + j));
    }
}

// zero-fill of `bytes` bytes at a 4-byte aligned address: a workgroup owns 16 KB; 16-byte stores, consecutive lanes on consecutive
// 16-byte pieces (every store instruction of a wave covers 1 KB of whole lines -- with 64 contiguous bytes PER LANE each instruction
// touched a quarter of 64 different 64-byte segments: 261 MB took 75 us = 3.5 TB/s); 4-byte pieces for an unaligned base / the tail
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, int64_t words) {
    const int64_t b = (int64_t)blockIdx.x * 4096;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && b + 4096 <= words) {
        uint4* q = reinterpret_cast<uint4*>(p + b) + threadIdx.x;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        q[0] = z; q[256] = z; q[512] = z; q[768] = z;
    } else {
        for (int64_t j = b + threadIdx.x; j < words && j < b + 4096; j += 256) p[j] = 0u;
    }
}

// ---- segmented column sums (bias gradients; per-level sums for level_embed) -----------------------
constexpr int CS_MAXSEG = 8;
struct ColsumP {
    const void* x;
    int64_t ld;
    float* out;
    int64_t rows_per_batch;
    int cols;
    int64_t seg[CS_MAXSEG + 1];
    int nseg;
    int rows_per_block;
    int aligned;
};

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const ColsumP p) {
    // block = rows_per_block rows x 256 columns; each wave streams its share of the rows with 8 independent 4-column
    // loads in flight per lane, the 4 waves are combined through LDS, then ONE atomic per column per block (and segment).
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * p.rows_per_block;
    const int64_t r1 = min(p.rows_per_batch, r0 + p.rows_per_block);
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)blockIdx.z * p.rows_per_batch * p.ld;
    const bool incol = c0 < p.cols;
    const bool full = (c0 + 4 <= p.cols) && p.aligned;
    // segments covered by this block's row range
    int s_lo = 0;
    while (s_lo + 1 < p.nseg && r0 >= p.seg[s_lo + 1]) ++s_lo;
    int s_hi = s_lo;
    while (s_hi + 1 < p.nseg && r1 - 1 >= p.seg[s_hi + 1]) ++s_hi;
    for (int sg = s_lo; sg <= s_hi; ++sg) {
        const int64_t a = max(r0, p.seg[sg]), b = (sg + 1 < p.nseg || p.nseg > 1) ? min(r1, p.seg[sg + 1]) : r1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = 8;
        int64_t r = a + wid;
        if (incol) {
            for (; full && r + 4 * (U - 1) < b; r += 4 * U) {
                float v[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u) vec<T, 4>::ld(xb + (r + 4 * u) * p.ld + c0, v[u]);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
            }
            for (; r < b; r += 4)
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < p.cols) acc[e] += io<T>::ld(xb + r * p.ld + c0 + e);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wid][lane * 4 + e] = acc[e];
        __syncthreads();
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < p.cols) {
            const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (t != 0.f) atomicAdd(p.out + (int64_t)sg * p.cols + c, t);
        }
    }
}

// ---- fp32 value-gradient maps -> row-major activations -------------------------------------------
template <typename T, typename TG = float>
__global__ __launch_bounds__(256) void vgrad_rows_kernel(const TG* __restrict__ gv, int64_t vs_n, int64_t vs_s, int64_t vs_m,
                                                         const uint8_t* __restrict__ mask, T* __restrict__ out, int64_t ld_out,
                                                         int S, int M, int D, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int groups = M * D / 8;
    const int64_t row = t / groups;
    const int c8 = (int)(t - row * groups);
    const int ch = c8 * 8, m = ch / D, d = ch - m * D;
    const int n = (int)(row / S), s = (int)(row - (int64_t)n * S);
    float v[8];
    vec<TG, 8>::ld(gv + (int64_t)n * vs_n + (int64_t)s * vs_s + (int64_t)m * vs_m + d, v);
    if (mask && mask[row]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    vec<T, 8>::st(out + row * ld_out + ch, v);
}

// ---- NCHW <-> token-major (LDS tile transpose) ---------------------------------------------------
template <typename S, typename D, bool TO_TOKENS>
__global__ __launch_bounds__(256) void transpose_kernel(const S* __restrict__ src, D* __restrict__ dst, int C, int HW,
                                                        int64_t tok_off, int64_t tok_stride) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (TO_TOKENS) {
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, hw = hw0 + tx;
            tile[i][tx] = (c < C && hw < HW) ? io<S>::ld(src + ((int64_t)n * C + c) * HW + hw) : 0.f;
        }
        __syncthreads();
        for (int i = ty; i < 32; i += 8) {
            const int hw = hw0 + i, c = c0 + tx;
            if (c < C && hw < HW) io<D>::st(dst + ((int64_t)n * tok_stride + tok_off + hw) * C + c, tile[tx][i]);
        }
    } else {
        for (int i = ty; i < 32; i += 8) {
            const int hw = hw0 + i, c = c0 + tx;
            tile[i][tx] = (c < C && hw < HW) ? io<S>::ld(src + ((int64_t)n * tok_stride + tok_off + hw) * C + c) : 0.f;
        }
        __syncthreads();
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, hw = hw0 + tx;
            if (c < C && hw < HW) io<D>::st(dst + ((int64_t)n * C + c) * HW + hw, tile[tx][i]);
        }
    }
}

// NCHW fp32 map -> token-major rows [hi | lo] of 2 C bf16 each: hi = bf16(x), lo = bf16(x - hi).  The row is the activation
// operand of the input projection taken with 16 significant bits: [hi | lo] [W | W]^T = (hi + lo) W^T in ONE product with K = 2 C.
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, hw = hw0 + tx;
        tile[i][tx] = (c < C && hw < HW) ? src[((int64_t)n * C + c) * HW + hw] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int hw = hw0 + i, c = c0 + tx;
        if (c < C && hw < HW) {
            const float x = tile[tx][i];
            const bf16_t hi = f2bf(x);
            bf16_t* row = dst + ((int64_t)n * HW + hw) * (2 * C);
            row[c] = hi;
            row[C + c] = f2bf(x - bf2f(hi));
        }
    }
}

// the same split for a row-major fp32 matrix (rows, K) -> (rows, 2 K) bf16 [hi | lo] (the im2col matrix of the extra level)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int K, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int64_t row = t / K;
    const int k = (int)(t - row * K);
    const float x = src[t];
    const bf16_t hi = f2bf(x);
    dst[row * 2 * K + k] = hi;
    dst[row * 2 * K + K + k] = f2bf(x - bf2f(hi));
}

// ---- im2col for the 3x3 stride-2 pad-1 extra-level conv -----------------------------------------
template <typename S, typename D>
__global__ __launch_bounds__(256) void im2col_kernel(const S* __restrict__ src, D* __restrict__ dst, int C, int H, int W,
                                                     int Ho, int Wo, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int K = C * 9;
    const int64_t row = t / K;
    const int k = (int)(t - row * K);
    const int c = k / 9, r9 = k - c * 9, ky = r9 / 3, kx = r9 - ky * 3;
    const int n = (int)(row / (Ho * Wo)), o = (int)(row - (int64_t)n * Ho * Wo);
    const int oy = o / Wo, ox = o - oy * Wo;
    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = io<S>::ld(src + (((int64_t)n * C + c) * H + iy) * W + ix);
    io<D>::st(dst + t, v);
}

// adjoint of the above for a SECOND (third, ...) extra level, whose input is the previous projected level (reference:
// pose_estimation_transformer.py:327-330, `self.input_proj[lvl](srcs[-1])`): d(input pixel) = sum of the <= 4 im2col entries
// that read it, ADDED to the token rows [tok_off, tok_off + H W) of the (N, tok_stride, C) stream gradient -- a gather per
// input element, no atomics.
template <typename S, typename D>
__global__ __launch_bounds__(256) void col2im_add_kernel(const S* __restrict__ dcol, D* __restrict__ dst, int C, int H, int W,
                                                         int Ho, int Wo, int64_t tok_off, int64_t tok_stride, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % C);
    const int64_t pix = t / C;
    const int n = (int)(pix / ((int64_t)H * W)), o = (int)(pix - (int64_t)n * H * W);
    const int iy = o / W, ix = o - iy * W;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;                          // = 2 oy
        if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + 1 - kx;
            if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
            acc += io<S>::ld(dcol + ((int64_t)n * Ho * Wo + (int64_t)(ty >> 1) * Wo + (tx >> 1)) * ((int64_t)C * 9) + c * 9 + ky * 3 + kx);
        }
    }
    D* d = dst + ((int64_t)n * tok_stride + tok_off + o) * C + c;
    io<D>::st(d, io<D>::ld(d) + acc);
}

// ---- position encodings --------------------------------------------------------------------------
__device__ __forceinline__ void pair_st(float* o, float a, float b) { *reinterpret_cast<float2*>(o) = make_float2(a, b); }
__device__ __forceinline__ void pair_st(bf16_t* o, float a, float b) { *reinterpret_cast<uint32_t*>(o) = pack_bf2(a, b); }
template <typename T>
__global__ __launch_bounds__(256) void pos_sine_kernel(const uint8_t* __restrict__ mask, T* __restrict__ out,
                                                       const float* __restrict__ level_embed,
                                                       const float* __restrict__ dim_t, int H, int W, int F,
                                                       int64_t tok_off, int64_t tok_stride) {
    extern __shared__ float sm[];          // ey[W], ex[W]
    float* ey = sm;
    float* ex = sm + W;
    const int n = blockIdx.x / H, y = blockIdx.x % H;
    const uint8_t* mb = mask + (int64_t)n * H * W;
    const float two_pi = 6.283185307179586f;
    for (int x = threadIdx.x; x < W; x += 256) {
        int cy = 0, ty = 0, cx = 0, tx = 0;
#pragma unroll 8
        for (int r = 0; r < H; ++r) { const int u = mb[r * W + x] ? 0 : 1; ty += u; cy += (r <= y) ? u : 0; }
#pragma unroll 8
        for (int c = 0; c < W; ++c) { const int u = mb[y * W + c] ? 0 : 1; tx += u; cx += (c <= x) ? u : 0; }
        ey[x] = ((float)cy - 0.5f) / ((float)ty + 1e-6f) * two_pi;
        ex[x] = ((float)cx - 0.5f) / ((float)tx + 1e-6f) * two_pi;
    }
    __syncthreads();
    // one thread per (sin, cos) PAIR: channels 2k and 2k+1 share dim_t (position_encoding.py:52-57), so the angle, its
    // division and its range reduction are done once.  Normalised positions are <= 2 pi: the hardware sine / cosine
    // (v_sin_f32 on revolutions, abs error ~1e-6) serves them; the huge arguments of fully masked rows / columns
    // ((c - 0.5) / 1e-6) keep the exact full-range path, where 1 ulp of the argument decides the value.
    const int C2 = 2 * F, HP = F / 2;                          // pairs per half (y features | x features)
    auto emit = [&](int x, int pr, float dt, float le0, float le1) __attribute__((always_inline)) {
        const bool ypart = pr < HP;
        const int k = ypart ? pr : pr - HP;                    // pair inside the half: channels 2k, 2k+1 of it
        const float e = ypart ? ey[x] : ex[x];
        const float arg = e / dt;
        float sv, cv;
        if (fabsf(arg) <= 8.f) { sv = __sinf(arg); cv = __cosf(arg); }
        else { sv = sinf(arg); cv = cosf(arg); }
        const int c = (ypart ? 0 : F) + 2 * k;
        T* o = out + ((int64_t)n * tok_stride + tok_off + (int64_t)y * W + x) * C2 + c;
        pair_st(o, sv + le0, cv + le1);
    };
    if (256 % F == 0) {
        // a thread keeps ONE pair index and walks the row: its divisor and level-embedding pair are loaded once, no index
        // division, and the F threads of a pixel write its C2 channels as one contiguous run
        const int pr = threadIdx.x % F, k = pr < HP ? pr : pr - HP, c = (pr < HP ? 0 : F) + 2 * k;
        const float dt = dim_t[2 * k];
        const float le0 = level_embed ? level_embed[c] : 0.f, le1 = level_embed ? level_embed[c + 1] : 0.f;
        for (int x = threadIdx.x / F; x < W; x += 256 / F) emit(x, pr, dt, le0, le1);
    } else {
        for (int i = threadIdx.x; i < W * F; i += 256) {
            const int x = i / F, pr = i - x * F, k = pr < HP ? pr : pr - HP, c = (pr < HP ? 0 : F) + 2 * k;
            emit(x, pr, dim_t[2 * k], level_embed ? level_embed[c] : 0.f, level_embed ? level_embed[c + 1] : 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void bbox_sine_kernel(const float* __restrict__ boxes, const uint8_t* __restrict__ valid,
                                                        float* __restrict__ out, int n, int F, float fill) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * 4 * F) return;
    const int b = t / (4 * F), r = t - b * 4 * F, c = r / F, k = r - c * F;
    if (valid && !valid[b]) {
        out[(int64_t)b * 8 * F + c * 2 * F + k] = fill;
        out[(int64_t)b * 8 * F + c * 2 * F + F + k] = fill;
        return;
    }
    const float arg = boxes[b * 4 + c] * exp2f((float)k);          // exact power-of-two scaling
    out[(int64_t)b * 8 * F + c * 2 * F + k] = sinf(arg);            // full-range reduction: args reach 2^31
    out[(int64_t)b * 8 * F + c * 2 * F + F + k] = cosf(arg);
}

__global__ __launch_bounds__(256) void dec_ref_kernel(const float* __restrict__ ref, const float* __restrict__ vr,
                                                      float* __restrict__ out, int N, int Q, int L) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * Q * L) return;
    const int l = t % L, nq = t / L, n = nq / Q;
    out[t * 2 + 0] = ref[nq * 2 + 0] * vr[(n * L + l) * 2 + 0];
    out[t * 2 + 1] = ref[nq * 2 + 1] * vr[(n * L + l) * 2 + 1];
}

__global__ __launch_bounds__(64) void valid_ratio_kernel(const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                         int64_t out_stride, int H, int W) {
    const int n = blockIdx.x;
    const uint8_t* mb = mask + (int64_t)n * H * W;
    float vh = 0.f, vw = 0.f;
    for (int y = threadIdx.x; y < H; y += 64) vh += mb[y * W] ? 0.f : 1.f;      // ~mask[:, :, 0]
    for (int x = threadIdx.x; x < W; x += 64) vw += mb[x] ? 0.f : 1.f;          // ~mask[:, 0, :]
    vh = wave_sum(vh);
    vw = wave_sum(vw);
    if (threadIdx.x == 0) { out[n * out_stride + 0] = vw / (float)W; out[n * out_stride + 1] = vh / (float)H; }
}

__global__ __launch_bounds__(256) void mask_nearest_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           int N, int H, int W, int Ho, int Wo) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * Ho * Wo) return;
    const int n = t / (Ho * Wo), o = t - n * Ho * Wo, oy = o / Wo, ox = o - oy * Wo;
    // torch 'nearest': src = floor(dst * scale) with scale = in/out computed in fp32
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const int iy = min((int)floorf((float)oy * sy), H - 1), ix = min((int)floorf((float)ox * sx), W - 1);
    dst[t] = src[((int64_t)n * H + iy) * W + ix];
}

template <typename T>
__global__ __launch_bounds__(256) void add_rowvec_kernel(T* __restrict__ x, const float* __restrict__ vec, int64_t batch_stride_rows,
                                                         int64_t row0, int64_t rows, int cols) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * cols) return;
    const int64_t r = t / cols;
    const int c = (int)(t - r * cols);
    T* p = x + ((int64_t)blockIdx.y * batch_stride_rows + row0 + r) * cols + c;
    io<T>::st(p, io<T>::ld(p) + vec[c]);
}

struct RefP {
    const float* vr;
    float* ref;
    int N, L, S;
    int H[8], W[8], start[8];
};
__global__ __launch_bounds__(256) void enc_ref_kernel(const RefP p) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)p.N * p.S) return;
    const int n = (int)(t / p.S), s = (int)(t - (int64_t)n * p.S);
    int lq = 0;
    while (lq + 1 < p.L && s >= p.start[lq + 1]) ++lq;
    const int o = s - p.start[lq];
    const int y = o / p.W[lq], x = o - y * p.W[lq];
    const float* vr = p.vr + (int64_t)n * p.L * 2;
    const float rx = ((float)x + 0.5f) / (vr[lq * 2 + 0] * (float)p.W[lq]);
    const float ry = ((float)y + 0.5f) / (vr[lq * 2 + 1] * (float)p.H[lq]);
    for (int l = 0; l < p.L; ++l) {
        p.ref[(t * p.L + l) * 2 + 0] = rx * vr[l * 2 + 0];
        p.ref[(t * p.L + l) * 2 + 1] = ry * vr[l * 2 + 1];
    }
}

// ---- pose heads tail: class-slot gather + 6D -> rotation matrix -----------------------------------
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(256) void pose_fwd_kernel(const float* __restrict__ rot_all, const float* __restrict__ trans_all,
                                                       const int32_t* __restrict__ cls, float* __restrict__ rot,
                                                       float* __restrict__ trans, int R, int ncls) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int c = cls[r] > 0 ? cls[r] : 0;
    const float* p6 = rot_all + (int64_t)r * ncls * 6 + c * 6;
    const float m1[3] = {p6[0], p6[1], p6[2]}, m2[3] = {p6[3], p6[4], p6[5]};
    const float nx = fmaxf(sqrtf(m1[0] * m1[0] + m1[1] * m1[1] + m1[2] * m1[2]), 1e-12f);
    const float x[3] = {m1[0] / nx, m1[1] / nx, m1[2] / nx};
    float zp[3];
    cross3(x, m2, zp);
    const float nz = fmaxf(sqrtf(zp[0] * zp[0] + zp[1] * zp[1] + zp[2] * zp[2]), 1e-12f);
    const float z[3] = {zp[0] / nz, zp[1] / nz, zp[2] / nz};
    float y[3];
    cross3(z, x, y);
    float* o = rot + (int64_t)r * 9;
    for (int i = 0; i < 3; ++i) { o[i * 3 + 0] = x[i]; o[i * 3 + 1] = y[i]; o[i * 3 + 2] = z[i]; }
    for (int i = 0; i < 3; ++i) trans[(int64_t)r * 3 + i] = trans_all[(int64_t)r * ncls * 3 + c * 3 + i];
}


// ---- pose losses (pose_estimation_transformer.py:635-674) for all decoder layers in one launch ----
// One workgroup per layer: L2 translation distance and geodesic rotation angle of the matched (query, target) pairs,
// averaged over the pairs, PLUS the gradient of each of the two losses w.r.t. that layer's predictions (zero rows for
// unmatched queries).  Replaces ~70 stack / index / elementwise launches of the PyTorch formulation and their backward.
__global__ __launch_bounds__(256) void pose_loss_kernel(const float* __restrict__ trans, const float* __restrict__ rot,
                                                        const int64_t* __restrict__ qi, const float* __restrict__ tt,
                                                        const float* __restrict__ tr, int n_obj, int NQ,
                                                        float* __restrict__ losses, float* __restrict__ gt, float* __restrict__ gr,
                                                        const int32_t* __restrict__ n_obj_dev, int L, const float* __restrict__ weights,
                                                        float* __restrict__ total) {
    // (ABI v5) weights (L, 2): the gradients leave already multiplied by their loss weight and *total = sum_l w . losses[l] -- then ONE
    // workgroup walks the layers (the weighted sum needs them all; the work is a few hundred pairs per layer), so that the captured
    // step holds no framework elementwise / reduction kernels behind the loss
    const int tid = threadIdx.x;
    if (n_obj_dev) n_obj = min(max(*n_obj_dev, 0), NQ);
    __shared__ float red[2][4];
    float tot = 0.f;
    for (int l = blockIdx.x; l < L; l += gridDim.x) {
    const float w0 = weights ? weights[l * 2] : 1.f, w1 = weights ? weights[l * 2 + 1] : 1.f;
    const float* T = trans + (int64_t)l * NQ * 3;
    const float* Rm = rot + (int64_t)l * NQ * 9;
    float* GT = gt + (int64_t)l * NQ * 3;
    float* GR = gr + (int64_t)l * NQ * 9;
    for (int i = tid; i < NQ * 3; i += 256) GT[i] = 0.f;
    for (int i = tid; i < NQ * 9; i += 256) GR[i] = 0.f;
    __syncthreads();
    const float inv_n = 1.f / (float)max(n_obj, 1);
    float lt = 0.f, lr = 0.f;
    for (int i = tid; i < n_obj; i += 256) {
        const int64_t q = qi[i];
        float d[3], s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { d[k] = T[q * 3 + k] - tt[i * 3 + k]; s2 += d[k] * d[k]; }
        const float nrm = sqrtf(s2);
        lt += nrm;
        const float sc = nrm > 0.f ? w0 * inv_n / nrm : 0.f;       // d||d||/dd = d/||d|| (0 at d = 0, where torch gives NaN)
#pragma unroll
        for (int k = 0; k < 3; ++k) GT[q * 3 + k] = d[k] * sc;
        float tg[9], trc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { tg[k] = tr[i * 9 + k]; trc += Rm[q * 9 + k] * tg[k]; }   // trace(R_pred R_gt^T)
        const float x = 0.5f * (trc - 1.f);
        const float xc = fminf(fmaxf(x, -1.f + 1e-6f), 1.f - 1e-6f);
        lr += acosf(xc);
        const float dac = (x == xc) ? -0.5f * w1 * inv_n * rsqrtf(1.f - xc * xc) : 0.f;   // clamp passes the gradient only inside
#pragma unroll
        for (int k = 0; k < 9; ++k) GR[q * 9 + k] = dac * tg[k];
    }
    lt = wave_sum(lt); lr = wave_sum(lr);
    if ((tid & 63) == 0) { red[0][tid >> 6] = lt; red[1][tid >> 6] = lr; }
    __syncthreads();
    if (tid == 0) {
        const float a = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv_n, b = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv_n;
        losses[l * 2 + 0] = a;
        losses[l * 2 + 1] = b;
        tot += w0 * a + w1 * b;
    }
    __syncthreads();                                              // (red is reused by the next layer)
    }
    if (total && tid == 0) *total = tot;
}

__global__ __launch_bounds__(256) void pose_bwd_kernel(const float* __restrict__ rot_all, const int32_t* __restrict__ cls,
                                                       const float* __restrict__ drot, const float* __restrict__ dtrans,
                                                       float* __restrict__ drot_all, float* __restrict__ dtrans_all,
                                                       int R, int ncls) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int c = cls[r] > 0 ? cls[r] : 0;
    const float* p6 = rot_all + (int64_t)r * ncls * 6 + c * 6;
    const float m1[3] = {p6[0], p6[1], p6[2]}, m2[3] = {p6[3], p6[4], p6[5]};
    const float n1 = sqrtf(m1[0] * m1[0] + m1[1] * m1[1] + m1[2] * m1[2]);
    const float nx = fmaxf(n1, 1e-12f);
    const float x[3] = {m1[0] / nx, m1[1] / nx, m1[2] / nx};
    float zp[3];
    cross3(x, m2, zp);
    const float n2 = sqrtf(zp[0] * zp[0] + zp[1] * zp[1] + zp[2] * zp[2]);
    const float nz = fmaxf(n2, 1e-12f);
    const float z[3] = {zp[0] / nz, zp[1] / nz, zp[2] / nz};
    const float* g = drot + (int64_t)r * 9;
    float dx[3] = {g[0], g[3], g[6]}, dy[3] = {g[1], g[4], g[7]}, dz[3] = {g[2], g[5], g[8]};
    float t[3];
    cross3(x, dy, t);                       // y = z x x :  dz += x x dy ; dx += dy x z
    for (int i = 0; i < 3; ++i) dz[i] += t[i];
    cross3(dy, z, t);
    for (int i = 0; i < 3; ++i) dx[i] += t[i];
    float dzp[3];
    if (n2 > 1e-12f) {
        const float dot = z[0] * dz[0] + z[1] * dz[1] + z[2] * dz[2];
        for (int i = 0; i < 3; ++i) dzp[i] = (dz[i] - z[i] * dot) / nz;
    } else {
        for (int i = 0; i < 3; ++i) dzp[i] = dz[i] / nz;
    }
    cross3(m2, dzp, t);                     // z' = x x m2 : dx += m2 x dz' ; dm2 = dz' x x
    for (int i = 0; i < 3; ++i) dx[i] += t[i];
    float dm2[3], dm1[3];
    cross3(dzp, x, dm2);
    if (n1 > 1e-12f) {
        const float dot = x[0] * dx[0] + x[1] * dx[1] + x[2] * dx[2];
        for (int i = 0; i < 3; ++i) dm1[i] = (dx[i] - x[i] * dot) / nx;
    } else {
        for (int i = 0; i < 3; ++i) dm1[i] = dx[i] / nx;
    }
    float* ro = drot_all + (int64_t)r * ncls * 6;
    float* to = dtrans_all + (int64_t)r * ncls * 3;
    for (int i = 0; i < ncls * 6; ++i) ro[i] = 0.f;
    for (int i = 0; i < ncls * 3; ++i) to[i] = 0.f;
    for (int i = 0; i < 3; ++i) { ro[c * 6 + i] = dm1[i]; ro[c * 6 + 3 + i] = dm2[i]; to[c * 3 + i] = dtrans[(int64_t)r * 3 + i]; }
}

// ---- flat-arena optimizer ------------------------------------------------------------------------
// Two launches, no atomics: every replica of a data-parallel job must get the SAME bits for the clip factor out of the
// same reduced gradients (float atomics would make ranks drift apart by an ulp per step).
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (int64_t j = i; j < n; ++j) s += g[j] * g[j];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];              // fixed order per thread
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// Four consecutive elements per thread (16-byte loads / stores of p, g, m, v, 8-byte stores of the bf16 images; the per-64-element
// learning-rate multiplier and the device-side bias corrections -- two powf -- once per thread instead of once per element): the
// one-element form moved its 392 MB in 115 us.  Element arithmetic unchanged (bit-identical parameters).
__device__ __forceinline__ void adamw_one(float& pi, float gi_raw, float& mi, float& vi, float clip, float lr, float b1, float b2, float eps,
                                          float wd, float bc1, float bc2s) {
    const float gi = gi_raw * clip;
    pi = pi * (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= (lr / bc1) * (mi / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ pb, uint16_t* __restrict__ pb_lo, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                    const float* __restrict__ sqnorm, float max_norm, float gscale,
                                                    const uint32_t* __restrict__ step_dev, const float* __restrict__ lr_scale, int vec4) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (lr_scale) lr *= lr_scale[i >> 6];                    // per-64-element learning-rate multiplier (LR groups inside one range)
    if (step_dev) {                                          // graph replay: bias corrections from the device-side step
        const float t = (float)(*step_dev);
        bc1 = 1.f - powf(b1, t);
        bc2s = sqrtf(1.f - powf(b2, t));
    }
    float clip = gscale;
    if (sqnorm) {
        const float total = sqrtf(*sqnorm) * gscale;
        const float coef = fminf(max_norm / (total + 1e-6f), 1.f);
        clip *= coef;
    }
    if (vec4 && i + 4 <= n) {
        float4 p4 = *reinterpret_cast<const float4*>(p + i), m4 = *reinterpret_cast<const float4*>(m + i), v4 = *reinterpret_cast<const float4*>(v + i);
        const float4 g4 = *reinterpret_cast<const float4*>(g + i);
        adamw_one(p4.x, g4.x, m4.x, v4.x, clip, lr, b1, b2, eps, wd, bc1, bc2s);
        adamw_one(p4.y, g4.y, m4.y, v4.y, clip, lr, b1, b2, eps, wd, bc1, bc2s);
        adamw_one(p4.z, g4.z, m4.z, v4.z, clip, lr, b1, b2, eps, wd, bc1, bc2s);
        adamw_one(p4.w, g4.w, m4.w, v4.w, clip, lr, b1, b2, eps, wd, bc1, bc2s);
        *reinterpret_cast<float4*>(p + i) = p4; *reinterpret_cast<float4*>(m + i) = m4; *reinterpret_cast<float4*>(v + i) = v4;
        if (pb) {
            const bf16_t h0 = f2bf(p4.x), h1 = f2bf(p4.y), h2 = f2bf(p4.z), h3 = f2bf(p4.w);
            *reinterpret_cast<uint2*>(pb + i) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
            if (pb_lo) {                                     // the split-bf16 residual: W = hi + lo to 16 mantissa bits
                const bf16_t l0 = f2bf(p4.x - bf2f(h0)), l1 = f2bf(p4.y - bf2f(h1)), l2 = f2bf(p4.z - bf2f(h2)), l3 = f2bf(p4.w - bf2f(h3));
                *reinterpret_cast<uint2*>(pb_lo + i) = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));
            }
        }
        return;
    }
    for (int64_t k = i; k < min(i + 4, n); ++k) {            // unaligned buffers / the last elements (i % 4 == 0: same learning-rate group)
        float pi = p[k], mi = m[k], vi = v[k];
        adamw_one(pi, g[k], mi, vi, clip, lr, b1, b2, eps, wd, bc1, bc2s);
        p[k] = pi; m[k] = mi; v[k] = vi;
        if (pb) {
            const bf16_t hi = f2bf(pi);
            pb[k] = hi;
            if (pb_lo) pb_lo[k] = f2bf(pi - bf2f(hi));
        }
    }
}


// ---- batched assignment on the device: models/matcher.py:158-229 ('gt' mode: L1 box cost + linear_sum_assignment) ----
// One wave per image, at most 64 queries / targets.  The solver is SciPy's rectangular LSAP (Crouse's shortest augmenting
// path, scipy/optimize/rectangular_lsap) restated step for step in double precision, so the assignment -- ties included --
// is the one scipy.optimize.linear_sum_assignment returns for the same fp32 cost matrix: lane `it` owns entry `it` of the
// "remaining columns" list, and the sequential scan
//     for it: if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
// becomes a wave minimum plus its tie rule (equal minima: the LAST one whose column is still unassigned, else the FIRST).
// cost[r][c] = cost_bbox * (((|d0| + |d1|) + |d2|) + |d3|) in fp32, numpy's summation order.
constexpr int LSA_MAX = 64;

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(64) void lsa_boxes_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       const int* __restrict__ tgt_off, const int* __restrict__ n_pred, float cost_bbox,
                                                       int Q, int* __restrict__ col_out, int* __restrict__ status) {
    __shared__ double cost[LSA_MAX * LSA_MAX];
    __shared__ double u[LSA_MAX], v[LSA_MAX], spc[LSA_MAX];
    __shared__ int path[LSA_MAX], col4row[LSA_MAX], row4col[LSA_MAX], remaining[LSA_MAX];
    __shared__ unsigned char SR[LSA_MAX], SC[LSA_MAX];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int nr0 = n_pred[img], t0 = tgt_off[img], nc0 = tgt_off[img + 1] - t0;
    int* out = col_out + (int64_t)img * Q;
    for (int k = lane; k < Q; k += 64) out[k] = -1;
    if (nr0 <= 0 || nc0 <= 0) return;
    if (nr0 > LSA_MAX || nc0 > LSA_MAX) { if (lane == 0) atomicExch(status, 1); return; }
    // scipy transposes a tall matrix (more rows than columns) and solves that; rows = preds, columns = targets otherwise
    const bool tr = nc0 < nr0;
    const int nr = tr ? nc0 : nr0, nc = tr ? nr0 : nc0;
    for (int e = lane; e < nr0 * nc0; e += 64) {
        const int r = e / nc0, c = e - r * nc0;
        const float* a = pred + ((int64_t)img * Q + r) * 4;
        const float* b = tgt + (int64_t)(t0 + c) * 4;
        float sum = fabsf(a[0] - b[0]);
        sum = __fadd_rn(sum, fabsf(a[1] - b[1]));
        sum = __fadd_rn(sum, fabsf(a[2] - b[2]));
        sum = __fadd_rn(sum, fabsf(a[3] - b[3]));
        const double cv = (double)__fmul_rn(cost_bbox, sum);
        if (tr) cost[c * nc + r] = cv; else cost[r * nc + c] = cv;
    }
    if (lane < nr) { u[lane] = 0.0; col4row[lane] = -1; }
    if (lane < nc) { v[lane] = 0.0; path[lane] = -1; row4col[lane] = -1; }
    __syncthreads();

    for (int cur = 0; cur < nr; ++cur) {
        // ---- augmenting_path(cur) ----
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        if (lane < nc) { remaining[lane] = nc - lane - 1; SC[lane] = 0; spc[lane] = INFINITY; }
        if (lane < nr) SR[lane] = 0;
        __syncthreads();
        bool infeasible = false;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const bool act = lane < num_remaining;
            const int j = act ? remaining[lane] : 0;
            double val = INFINITY;
            bool unassigned = false;
            if (act) {
                const double r = minVal + cost[i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                val = spc[j];
                unassigned = row4col[j] == -1;
            }
            const double lowest = wave_min_f64(val);
            if (!(lowest < INFINITY)) { infeasible = true; break; }
            const unsigned long long cand = __ballot(act && val == lowest), free_c = __ballot(act && val == lowest && unassigned);
            const int index = free_c ? 63 - __clzll((long long)free_c) : __ffsll((long long)cand) - 1;
            minVal = lowest;
            __syncthreads();
            const int js = remaining[index];
            const int owner = row4col[js];
            if (owner == -1) sink = js; else i = owner;
            __syncthreads();
            if (lane == 0) { SC[js] = 1; remaining[index] = remaining[num_remaining - 1]; }
            --num_remaining;
            __syncthreads();
        }
        if (infeasible) { if (lane == 0) atomicExch(status, 2); return; }
        // ---- dual update ----
        if (lane == 0) u[cur] += minVal;
        if (lane < nr && SR[lane] && lane != cur) u[lane] += minVal - spc[col4row[lane]];
        if (lane < nc && SC[lane]) v[lane] -= minVal - spc[lane];
        __syncthreads();
        // ---- augment ----
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int ii = path[j];
                row4col[j] = ii;
                const int t = col4row[ii]; col4row[ii] = j; j = t;
                if (ii == cur) break;
            }
        }
        __syncthreads();
    }
    // result as (row -> column) over the ORIGINAL orientation: preds are rows
    if (!tr) { if (lane < nr) out[lane] = col4row[lane]; }
    else { if (lane < nr) out[col4row[lane]] = lane; }          // solved transposed: row `lane` is target `lane`, its column a pred
}

// matched pairs -> the flat arrays poet_pose_loss takes: for image b, pred row s with col[b][s] >= 0, in (b, s) order
__global__ void match_gather_kernel(const int* __restrict__ col, const int* __restrict__ tgt_off, const float* __restrict__ tpos,
                                    const float* __restrict__ trot, int N, int Q, int64_t* __restrict__ qi, float* __restrict__ tt,
                                    float* __restrict__ trm, int* __restrict__ n_out) {
    // one workgroup; output slot of pair (b, s) = matched pairs of the images before b + matched rows before s in image b
    // (the (b, s) order of the host matcher), every step parallel: the assignment sits in LDS, one thread per pair
    __shared__ int base[257];
    extern __shared__ int scol[];                                // [N * Q]
    const int tid = threadIdx.x, NQ = N * Q;
    for (int i = tid; i < NQ; i += blockDim.x) scol[i] = col[i];
    __syncthreads();
    for (int b = tid; b <= N; b += blockDim.x) {                 // base[b] = pairs of images 0 .. b-1
        int acc = 0;
        for (int i = 0; i < b * Q; ++i) acc += scol[i] >= 0;
        base[b] = acc;
        if (b == N && n_out) *n_out = acc;
    }
    __syncthreads();
    for (int i = tid; i < NQ; i += blockDim.x) {
        const int c = scol[i];
        if (c < 0) continue;
        const int b = i / Q;
        int k = base[b];
        for (int j = b * Q; j < i; ++j) k += scol[j] >= 0;
        const int g = tgt_off[b] + c;
        qi[k] = (int64_t)i;
#pragma unroll
        for (int e = 0; e < 3; ++e) tt[k * 3 + e] = tpos[g * 3 + e];
#pragma unroll
        for (int e = 0; e < 9; ++e) trm[k * 9 + e] = trot[g * 9 + e];
    }
}

}  // namespace poet

using namespace poet;
#define ST ((hipStream_t)stream)

extern "C" int poet_add(const void* a, const void* b, void* out, int64_t n, int da, int db, int dout, void* stream) {
    POET_CHECK(a && b && out && n > 0, POET_ERR_ARG, "add: bad args");
    dim3 grid(cdiv(n, 2048)), block(256);
    const int key = da * 4 + db * 2 + dout;
    if (key == 7) hipLaunchKernelGGL((add_kernel<bf16_t, bf16_t, bf16_t>), grid, block, 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else if (key == 0) hipLaunchKernelGGL((add_kernel<float, float, float>), grid, block, 0, ST, (const float*)a, (const float*)b, (float*)out, n);
    else if (key == 3) hipLaunchKernelGGL((add_kernel<float, bf16_t, bf16_t>), grid, block, 0, ST, (const float*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else if (key == 1) hipLaunchKernelGGL((add_kernel<float, float, bf16_t>), grid, block, 0, ST, (const float*)a, (const float*)b, (bf16_t*)out, n);
    else { set_error("add: unsupported dtype triple %d %d %d", da, db, dout); return POET_ERR_UNSUPPORTED; }
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_gelu_fwd(const void* x, void* y, int64_t n, int dtype_x, int dtype, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                             void* stream) {
    POET_CHECK(x && y && n > 0, POET_ERR_ARG, "gelu_fwd: bad args");
    POET_CHECK(drop_p >= 0.f && drop_p < 1.f, POET_ERR_ARG, "gelu_fwd: drop_p");
    POET_CHECK((dtype == POET_F32 || dtype == POET_BF16) && (dtype_x == dtype || dtype_x == POET_F32), POET_ERR_UNSUPPORTED,
               "gelu_fwd: dtypes x %d, y %d", dtype_x, dtype);
    const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    dim3 grid(cdiv(n, 2048)), block(256);
    if (dtype == POET_F32) hipLaunchKernelGGL((gelu_kernel<float, float, false>), grid, block, 0, ST, (const float*)x, (const float*)nullptr, (float*)y, n, th, ds, seed, seed_dev);
    else if (dtype_x == POET_F32) hipLaunchKernelGGL((gelu_kernel<float, bf16_t, false>), grid, block, 0, ST, (const float*)x, (const bf16_t*)nullptr, (bf16_t*)y, n, th, ds, seed, seed_dev);
    else hipLaunchKernelGGL((gelu_kernel<bf16_t, bf16_t, false>), grid, block, 0, ST, (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)y, n, th, ds, seed, seed_dev);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype_x, int dtype, float drop_p, uint32_t seed,
                             const uint32_t* seed_dev, void* stream) {
    POET_CHECK(dy && x && dx && n > 0, POET_ERR_ARG, "gelu_bwd: bad args");
    POET_CHECK(drop_p >= 0.f && drop_p < 1.f, POET_ERR_ARG, "gelu_bwd: drop_p");
    POET_CHECK((dtype == POET_F32 || dtype == POET_BF16) && (dtype_x == dtype || dtype_x == POET_F32), POET_ERR_UNSUPPORTED,
               "gelu_bwd: dtypes x %d, dy %d", dtype_x, dtype);
    const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    dim3 grid(cdiv(n, 2048)), block(256);
    if (dtype == POET_F32) hipLaunchKernelGGL((gelu_kernel<float, float, true>), grid, block, 0, ST, (const float*)x, (const float*)dy, (float*)dx, n, th, ds, seed, seed_dev);
    else if (dtype_x == POET_F32) hipLaunchKernelGGL((gelu_kernel<float, bf16_t, true>), grid, block, 0, ST, (const float*)x, (const bf16_t*)dy, (bf16_t*)dx, n, th, ds, seed, seed_dev);
    else hipLaunchKernelGGL((gelu_kernel<bf16_t, bf16_t, true>), grid, block, 0, ST, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, th, ds, seed, seed_dev);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_add_cast(const float* a, const void* b_bf16, void* sum_bf16, void* a_bf16, int64_t n, void* stream) {
    POET_CHECK(a && b_bf16 && sum_bf16 && a_bf16 && n > 0, POET_ERR_ARG, "add_cast: bad args");
    POET_CHECK(((reinterpret_cast<uintptr_t>(a) & 31) | (reinterpret_cast<uintptr_t>(b_bf16) & 15) | (reinterpret_cast<uintptr_t>(sum_bf16) & 15) |
                (reinterpret_cast<uintptr_t>(a_bf16) & 15)) == 0, POET_ERR_ARG, "add_cast: operands must be 16-byte (fp32: 32-byte) aligned");
    hipLaunchKernelGGL(add_cast_kernel, dim3(cdiv(n, 2048)), dim3(256), 0, ST, a, (const bf16_t*)b_bf16, (bf16_t*)sum_bf16, (bf16_t*)a_bf16, n);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_cast(const void* src, void* dst, int64_t n, int sd, int dd, void* stream) {
    POET_CHECK(src && dst && n > 0, POET_ERR_ARG, "cast: bad args");
    dim3 grid(cdiv(n, 2048)), block(256);
    if (sd == POET_F32 && dd == POET_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, ST, (const float*)src, (bf16_t*)dst, n);
    else if (sd == POET_BF16 && dd == POET_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, ST, (const bf16_t*)src, (float*)dst, n);
    else if (sd == POET_F32 && dd == POET_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, ST, (const float*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, block, 0, ST, (const bf16_t*)src, (bf16_t*)dst, n);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_zero(void* p, int64_t bytes, void* stream) {
    POET_CHECK(p && bytes >= 0, POET_ERR_ARG, "zero: bad args");
    POET_CHECK((reinterpret_cast<uintptr_t>(p) & 3) == 0 && (bytes & 3) == 0, POET_ERR_ARG, "zero: address and size must be multiples of 4 bytes");
    if (bytes == 0) return POET_OK;
    const int64_t words = bytes / 4;
    hipLaunchKernelGGL(zero_kernel, dim3(cdiv(words, 4096)), dim3(256), 0, ST, reinterpret_cast<uint32_t*>(p), words);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_colsum(const void* x, int64_t ld, float* out, int batch, int64_t rows_per_batch, int cols,
                           const int64_t* seg_start_host, int nseg, int dtype, void* stream) {
    POET_CHECK(x && out && batch > 0 && rows_per_batch > 0 && cols > 0, POET_ERR_ARG, "colsum: bad args");
    POET_CHECK(nseg >= 1 && nseg <= CS_MAXSEG, POET_ERR_UNSUPPORTED, "colsum: nseg %d not in 1..%d", nseg, CS_MAXSEG);
    ColsumP p{};
    p.x = x; p.ld = ld; p.out = out; p.rows_per_batch = rows_per_batch; p.cols = cols; p.nseg = nseg;
    if (seg_start_host) for (int i = 0; i <= nseg; ++i) p.seg[i] = seg_start_host[i];
    else { p.seg[0] = 0; p.seg[1] = rows_per_batch; }
    p.rows_per_block = rows_per_batch >= 4096 ? 128 : (rows_per_batch >= 512 ? 32 : 8);
    p.aligned = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    dim3 grid(cdiv(cols, 256), cdiv(rows_per_batch, p.rows_per_block), batch), block(256);
    if (dtype == POET_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, ST, p);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, ST, p);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_vgrad_to_rows(const void* gv, int64_t vs_n, int64_t vs_s, int64_t vs_m, const uint8_t* row_mask,
                                  void* out, int64_t ld_out, int N, int S, int M, int D, int gv_dtype, int dtype, void* stream) {
    if (ld_out <= 0) ld_out = (int64_t)M * D;
    POET_CHECK(gv && out && D % 8 == 0 && ld_out >= (int64_t)M * D && ld_out % 8 == 0, POET_ERR_ARG, "vgrad_to_rows: bad args");
    POET_CHECK(gv_dtype == POET_F32 || (gv_dtype == POET_BF16 && dtype == POET_BF16), POET_ERR_UNSUPPORTED, "vgrad_to_rows: bf16 maps go to bf16 rows");
    const int64_t total = (int64_t)N * S * M * D / 8;
    dim3 grid(cdiv(total, 256)), block(256);
    if (gv_dtype == POET_BF16) hipLaunchKernelGGL((vgrad_rows_kernel<bf16_t, bf16_t>), grid, block, 0, ST, (const bf16_t*)gv, vs_n, vs_s, vs_m, row_mask, (bf16_t*)out, ld_out, S, M, D, total);
    else if (dtype == POET_BF16) hipLaunchKernelGGL((vgrad_rows_kernel<bf16_t, float>), grid, block, 0, ST, (const float*)gv, vs_n, vs_s, vs_m, row_mask, (bf16_t*)out, ld_out, S, M, D, total);
    else hipLaunchKernelGGL((vgrad_rows_kernel<float, float>), grid, block, 0, ST, (const float*)gv, vs_n, vs_s, vs_m, row_mask, (float*)out, ld_out, S, M, D, total);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

// one wave per row: rows whose mask byte is set are overwritten with zeros (16-byte stores)
__global__ __launch_bounds__(256) void zero_masked_rows_kernel(char* x, int64_t ld_bytes, const uint8_t* __restrict__ mask, int64_t rows, int row_bytes) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows || !mask[row]) return;
    char* r = x + row * ld_bytes;
    for (int c = (threadIdx.x & 63) * 16; c < row_bytes; c += 64 * 16) *reinterpret_cast<uint4*>(r + c) = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int poet_zero_masked_rows(void* x, int64_t ld, const uint8_t* row_mask, int64_t rows, int cols, int dtype, void* stream) {
    const int esz = dtype == POET_BF16 ? 2 : 4;
    POET_CHECK(x && row_mask && rows > 0 && cols > 0 && ld >= cols, POET_ERR_ARG, "zero_masked_rows: bad args");
    POET_CHECK((dtype == POET_BF16 || dtype == POET_F32) && ((int64_t)cols * esz) % 16 == 0 && (ld * esz) % 16 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
               POET_ERR_UNSUPPORTED, "zero_masked_rows: rows must be 16-byte multiples, 16-byte aligned");
    hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, ST, reinterpret_cast<char*>(x), ld * esz, row_mask, rows, cols * esz);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

template <bool TO_TOKENS>
static int transpose_dispatch(const void* src, void* dst, int N, int C, int HW, int64_t tok_off, int64_t tok_stride,
                              int sd, int dd, void* stream) {
    POET_CHECK(src && dst && N > 0 && C > 0 && HW > 0, POET_ERR_ARG, "transpose: bad args");
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N), block(256);
    if (sd == POET_F32 && dd == POET_F32) hipLaunchKernelGGL((transpose_kernel<float, float, TO_TOKENS>), grid, block, 0, ST, (const float*)src, (float*)dst, C, HW, tok_off, tok_stride);
    else if (sd == POET_F32 && dd == POET_BF16) hipLaunchKernelGGL((transpose_kernel<float, bf16_t, TO_TOKENS>), grid, block, 0, ST, (const float*)src, (bf16_t*)dst, C, HW, tok_off, tok_stride);
    else if (sd == POET_BF16 && dd == POET_F32) hipLaunchKernelGGL((transpose_kernel<bf16_t, float, TO_TOKENS>), grid, block, 0, ST, (const bf16_t*)src, (float*)dst, C, HW, tok_off, tok_stride);
    else hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t, TO_TOKENS>), grid, block, 0, ST, (const bf16_t*)src, (bf16_t*)dst, C, HW, tok_off, tok_stride);
    POET_LAUNCH_CHECK();
    return POET_OK;
}
extern "C" int poet_nchw_to_tokens(const void* src, void* dst, int N, int C, int HW, int64_t tok_off, int64_t tok_stride, int sd, int dd, void* stream) {
    return transpose_dispatch<true>(src, dst, N, C, HW, tok_off, tok_stride, sd, dd, stream);
}
extern "C" int poet_nchw_to_tokens_split(const float* src, void* dst, int N, int C, int HW, void* stream) {
    POET_CHECK(src && dst && N > 0 && C > 0 && HW > 0, POET_ERR_ARG, "nchw_to_tokens_split: bad args");
    hipLaunchKernelGGL(transpose_split_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), N), dim3(256), 0, ST, src, (bf16_t*)dst, C, HW);
    POET_LAUNCH_CHECK();
    return POET_OK;
}
extern "C" int poet_split_rows(const float* src, void* dst, int64_t rows, int K, void* stream) {
    POET_CHECK(src && dst && rows > 0 && K > 0, POET_ERR_ARG, "split_rows: bad args");
    const int64_t total = rows * K;
    hipLaunchKernelGGL(split_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ST, src, (bf16_t*)dst, K, total);
    POET_LAUNCH_CHECK();
    return POET_OK;
}
extern "C" int poet_tokens_to_nchw(const void* src, void* dst, int N, int C, int HW, int64_t tok_off, int64_t tok_stride, int sd, int dd, void* stream) {
    return transpose_dispatch<false>(src, dst, N, C, HW, tok_off, tok_stride, sd, dd, stream);
}

extern "C" int poet_col2im3x3s2_add(const void* dcol, void* dst, int N, int C, int H, int W, int Ho, int Wo, int64_t tok_off, int64_t tok_stride,
                                    int sd, int dd, void* stream) {
    POET_CHECK(dcol && dst && N > 0 && C > 0 && H > 0 && W > 0, POET_ERR_ARG, "col2im: bad args");
    POET_CHECK(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, POET_ERR_ARG, "col2im: output size mismatch");
    POET_CHECK((sd == POET_F32 || sd == POET_BF16) && (dd == POET_F32 || dd == POET_BF16), POET_ERR_UNSUPPORTED, "col2im: dtypes");
    const int64_t total = (int64_t)N * H * W * C;
    dim3 grid(cdiv(total, 256)), block(256);
    if (sd == POET_F32 && dd == POET_F32) hipLaunchKernelGGL((col2im_add_kernel<float, float>), grid, block, 0, ST, (const float*)dcol, (float*)dst, C, H, W, Ho, Wo, tok_off, tok_stride, total);
    else if (sd == POET_BF16 && dd == POET_F32) hipLaunchKernelGGL((col2im_add_kernel<bf16_t, float>), grid, block, 0, ST, (const bf16_t*)dcol, (float*)dst, C, H, W, Ho, Wo, tok_off, tok_stride, total);
    else if (sd == POET_BF16 && dd == POET_BF16) hipLaunchKernelGGL((col2im_add_kernel<bf16_t, bf16_t>), grid, block, 0, ST, (const bf16_t*)dcol, (bf16_t*)dst, C, H, W, Ho, Wo, tok_off, tok_stride, total);
    else hipLaunchKernelGGL((col2im_add_kernel<float, bf16_t>), grid, block, 0, ST, (const float*)dcol, (bf16_t*)dst, C, H, W, Ho, Wo, tok_off, tok_stride, total);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_im2col3x3s2(const void* src, void* dst, int N, int C, int H, int W, int Ho, int Wo, int sd, int dd, void* stream) {
    POET_CHECK(src && dst && N > 0 && C > 0, POET_ERR_ARG, "im2col: bad args");
    POET_CHECK(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, POET_ERR_ARG, "im2col: output size mismatch");
    const int64_t total = (int64_t)N * Ho * Wo * C * 9;
    dim3 grid(cdiv(total, 256)), block(256);
    if (sd == POET_F32 && dd == POET_F32) hipLaunchKernelGGL((im2col_kernel<float, float>), grid, block, 0, ST, (const float*)src, (float*)dst, C, H, W, Ho, Wo, total);
    else if (sd == POET_F32 && dd == POET_BF16) hipLaunchKernelGGL((im2col_kernel<float, bf16_t>), grid, block, 0, ST, (const float*)src, (bf16_t*)dst, C, H, W, Ho, Wo, total);
    else if (sd == POET_BF16 && dd == POET_BF16) hipLaunchKernelGGL((im2col_kernel<bf16_t, bf16_t>), grid, block, 0, ST, (const bf16_t*)src, (bf16_t*)dst, C, H, W, Ho, Wo, total);
    else hipLaunchKernelGGL((im2col_kernel<bf16_t, float>), grid, block, 0, ST, (const bf16_t*)src, (float*)dst, C, H, W, Ho, Wo, total);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_pos_sine(const uint8_t* mask, void* out, const float* level_embed, const float* dim_t, int N, int H, int W,
                             int F, int64_t tok_off, int64_t tok_stride, int dtype, void* stream) {
    POET_CHECK(mask && out && dim_t && N > 0 && H > 0 && W > 0 && F > 0, POET_ERR_ARG, "pos_sine: bad args");
    dim3 grid(N * H), block(256);
    const size_t lds = sizeof(float) * 2 * W;
    if (dtype == POET_BF16) hipLaunchKernelGGL(pos_sine_kernel<bf16_t>, grid, block, lds, ST, mask, (bf16_t*)out, level_embed, dim_t, H, W, F, tok_off, tok_stride);
    else hipLaunchKernelGGL(pos_sine_kernel<float>, grid, block, lds, ST, mask, (float*)out, level_embed, dim_t, H, W, F, tok_off, tok_stride);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_bbox_sine(const float* boxes, const uint8_t* valid, float* out, int n, int F, float fill, void* stream) {
    POET_CHECK(boxes && out && n > 0 && F > 0 && F <= 64, POET_ERR_ARG, "bbox_sine: bad args");
    dim3 grid(cdiv((int64_t)n * 4 * F, 256)), block(256);
    hipLaunchKernelGGL(bbox_sine_kernel, grid, block, 0, ST, boxes, valid, out, n, F, fill);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_dec_ref_points(const float* ref, const float* valid_ratios, float* out, int N, int Q, int L, void* stream) {
    POET_CHECK(ref && valid_ratios && out && N > 0 && Q > 0 && L > 0, POET_ERR_ARG, "dec_ref_points: bad args");
    hipLaunchKernelGGL(dec_ref_kernel, dim3(cdiv((int64_t)N * Q * L, 256)), dim3(256), 0, ST, ref, valid_ratios, out, N, Q, L);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_valid_ratio(const uint8_t* mask, float* out, int64_t out_stride, int N, int H, int W, void* stream) {
    POET_CHECK(mask && out && N > 0 && H > 0 && W > 0, POET_ERR_ARG, "valid_ratio: bad args");
    hipLaunchKernelGGL(valid_ratio_kernel, dim3(N), dim3(64), 0, ST, mask, out, out_stride, H, W);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_mask_nearest(const uint8_t* src, uint8_t* dst, int N, int H, int W, int Ho, int Wo, void* stream) {
    POET_CHECK(src && dst && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, POET_ERR_ARG, "mask_nearest: bad args");
    hipLaunchKernelGGL(mask_nearest_kernel, dim3(cdiv((int64_t)N * Ho * Wo, 256)), dim3(256), 0, ST, src, dst, N, H, W, Ho, Wo);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_add_rowvec(void* x, const float* vec, int batch, int64_t batch_stride_rows, int64_t row0, int64_t rows,
                               int cols, int dtype, void* stream) {
    POET_CHECK(x && vec && batch > 0 && rows > 0 && cols > 0, POET_ERR_ARG, "add_rowvec: bad args");
    dim3 grid(cdiv(rows * cols, 256), batch), block(256);
    if (dtype == POET_BF16) hipLaunchKernelGGL(add_rowvec_kernel<bf16_t>, grid, block, 0, ST, (bf16_t*)x, vec, batch_stride_rows, row0, rows, cols);
    else hipLaunchKernelGGL(add_rowvec_kernel<float>, grid, block, 0, ST, (float*)x, vec, batch_stride_rows, row0, rows, cols);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_enc_ref_points(const float* valid_ratios, const int64_t* shapes, float* ref, int N, int L, int S, void* stream) {
    POET_CHECK(valid_ratios && shapes && ref && L >= 1 && L <= 8, POET_ERR_ARG, "enc_ref_points: bad args (1 <= n_levels <= 8)");
    RefP p{};
    p.vr = valid_ratios; p.ref = ref; p.N = N; p.L = L; p.S = S;
    int64_t acc = 0;
    for (int l = 0; l < L; ++l) { p.H[l] = (int)shapes[2 * l]; p.W[l] = (int)shapes[2 * l + 1]; p.start[l] = (int)acc; acc += (int64_t)p.H[l] * p.W[l]; }
    POET_CHECK(acc == S, POET_ERR_ARG, "enc_ref_points: sum(H*W) != S");
    dim3 grid(cdiv((int64_t)N * S, 256)), block(256);
    hipLaunchKernelGGL(enc_ref_kernel, grid, block, 0, ST, p);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_pose_finish_fwd(const float* rot_all, const float* trans_all, const int32_t* cls, float* rot, float* trans, int R, int ncls, void* stream) {
    POET_CHECK(rot_all && trans_all && cls && rot && trans && R > 0 && ncls > 0, POET_ERR_ARG, "pose_finish_fwd: bad args");
    hipLaunchKernelGGL(pose_fwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, ST, rot_all, trans_all, cls, rot, trans, R, ncls);
    POET_LAUNCH_CHECK();
    return POET_OK;
}
extern "C" int poet_pose_finish_bwd(const float* rot_all, const int32_t* cls, const float* drot, const float* dtrans, float* drot_all, float* dtrans_all, int R, int ncls, void* stream) {
    POET_CHECK(rot_all && cls && drot && dtrans && drot_all && dtrans_all && R > 0 && ncls > 0, POET_ERR_ARG, "pose_finish_bwd: bad args");
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, ST, rot_all, cls, drot, dtrans, drot_all, dtrans_all, R, ncls);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_pose_loss(const float* trans, const float* rot, const int64_t* query_idx, const float* tgt_trans,
                              const float* tgt_rot, int n_obj, int L, int NQ, float* losses, float* grad_trans, float* grad_rot,
                              const int32_t* n_obj_dev, const float* weights, float* total, void* stream) {
    POET_CHECK(trans && rot && losses && grad_trans && grad_rot && L > 0 && NQ > 0 && n_obj >= 0, POET_ERR_ARG, "pose_loss: bad args");
    POET_CHECK((n_obj == 0 && !n_obj_dev) || (query_idx && tgt_trans && tgt_rot), POET_ERR_ARG, "pose_loss: null match arrays");
    POET_CHECK(!total || weights, POET_ERR_ARG, "pose_loss: total needs weights");
    hipLaunchKernelGGL(pose_loss_kernel, dim3(total ? 1 : L), dim3(256), 0, ST, trans, rot, query_idx, tgt_trans, tgt_rot, n_obj, NQ, losses,
                       grad_trans, grad_rot, n_obj_dev, L, weights, total);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_lsa_boxes(const float* pred_boxes, const float* tgt_boxes, const int* tgt_off, const int* n_pred, float cost_bbox,
                              int N, int Q, int* col_out, int* status, void* stream) {
    POET_CHECK(pred_boxes && tgt_boxes && tgt_off && n_pred && col_out && status && N > 0 && Q > 0, POET_ERR_ARG, "lsa_boxes: bad args");
    hipLaunchKernelGGL(lsa_boxes_kernel, dim3(N), dim3(64), 0, ST, pred_boxes, tgt_boxes, tgt_off, n_pred, cost_bbox, Q, col_out, status);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_match_gather(const int* col, const int* tgt_off, const float* tgt_pos, const float* tgt_rot, int N, int Q,
                                 int64_t* query_idx, float* tgt_trans_out, float* tgt_rot_out, int* n_out, void* stream) {
    POET_CHECK(col && tgt_off && tgt_pos && tgt_rot && query_idx && tgt_trans_out && tgt_rot_out && N > 0 && N <= 256 && Q > 0, POET_ERR_ARG,
               "match_gather: bad args (N <= 256)");
    hipLaunchKernelGGL(match_gather_kernel, dim3(1), dim3(256), (size_t)N * Q * sizeof(int), ST, col, tgt_off, tgt_pos, tgt_rot, N, Q, query_idx, tgt_trans_out,
                       tgt_rot_out, n_out);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_sqnorm(const float* g, int64_t n, float* out, void* stream) {
    POET_CHECK(g && out && n > 0, POET_ERR_ARG, "sqnorm: bad args");
    int nb = cdiv(n, 1024);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, ST, g, n, out + 1);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, ST, out + 1, nb, out);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_adamw(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, uint16_t* p_bf16_lo, int64_t n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int step, const float* sqnorm, float max_norm,
                          float grad_scale, const uint32_t* step_dev, const float* lr_scale, void* stream) {
    POET_CHECK(p && g && m && v && n > 0 && (step >= 1 || step_dev), POET_ERR_ARG, "adamw: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    POET_CHECK(!p_bf16_lo || p_bf16, POET_ERR_ARG, "adamw: the lo shadow needs the hi shadow");
    const uintptr_t al = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v);
    const uintptr_t al16 = (reinterpret_cast<uintptr_t>(p_bf16) | reinterpret_cast<uintptr_t>(p_bf16_lo));
    const int vec4 = ((al & 15) == 0 && (al16 & 7) == 0) ? 1 : 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, ST, p, g, m, v, p_bf16, p_bf16_lo, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, sqnorm, max_norm, grad_scale, step_dev, lr_scale, vec4);
    POET_LAUNCH_CHECK();
    return POET_OK;
}

__global__ void counter_add_kernel(uint32_t* w, uint32_t delta) { *w += delta; }

extern "C" int poet_counter_add(uint32_t* word, uint32_t delta, void* stream) {
    POET_CHECK(word, POET_ERR_ARG, "counter_add: null");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, ST, word, delta);
    POET_LAUNCH_CHECK();
    return POET_OK;
}
