// LayerNorm (+residual +dropout) and token-major GroupNorm for gfx950, forward and backward.
// Reference call sites: deformable_transformer.py:202-203,195-196 (encoder), :279-287,271-272
// (decoder) for `x = norm(x + dropout(y))`; pose_estimation_transformer.py:106-122 for
// Conv+GroupNorm(32).  HBM-bound: one wave per row, 16-B (f32) / 8-B (bf16) lane accesses,
// statistics in fp32 registers, wave-level shuffles for the reductions.
#include "common.cuh"

namespace poet {

constexpr int LN_MAXIT = 4;   // d <= 1024

// The SPLIT residual stream (round 6): between the encoder's LayerNorms the stream y is not stored as fp32 but as its bf16 head --
// the operand copy y16 every LayerNorm writes anyway for the next GEMM -- plus the remainder y - float(y16) as IEEE fp16: the
// remainder is <= 2^-9 |y|, so its 11 bits put the pair at 2^-20 relative (fp32 keeps 2^-24), and a LayerNorm writes 52 MB less
// per launch at 102 080 rows.  (A plain fp16 stream saves twice that, measured -0.17 ms per step, but re-rolls every rounding
// downstream: its 2^-12 is within a factor 3 of the branch's own bf16 noise once the branch is smaller than the stream, and the
// all-layer maximum of the 1280x960 golden went 4.3e-3 -> 1.17e-2 in the arena pass.)
template <typename T> struct is_f16 { static constexpr bool value = false; };
template <> struct is_f16<f16_t> { static constexpr bool value = true; };
// res[e] of 4 consecutive elements: plain load, or bf16 head + fp16 remainder
template <typename TR>
__device__ __forceinline__ void ld_stream4(const TR* __restrict__ res, const bf16_t* __restrict__ hi, int64_t at, float* r) {
    vec<TR, 4>::ld(res + at, r);
    if constexpr (is_f16<TR>::value) {
        float h[4];
        vec<bf16_t, 4>::ld(hi + at, h);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] += h[e];
    }
}
// y[e]: plain store, or the fp16 remainder against the bf16 head that vec<bf16_t, 4>::st writes for the same values
template <typename TRO>
__device__ __forceinline__ void st_stream4(TRO* __restrict__ y, int64_t at, const float* o) {
    if constexpr (is_f16<TRO>::value) {
        const uint32_t p0 = pack_bf2(o[0], o[1]), p1 = pack_bf2(o[2], o[3]);
        const float rem[4] = {o[0] - __uint_as_float(p0 << 16), o[1] - __uint_as_float(p0 & 0xffff0000u),
                              o[2] - __uint_as_float(p1 << 16), o[3] - __uint_as_float(p1 & 0xffff0000u)};
        vec<f16_t, 4>::st(y + at, rem);
    } else vec<TRO, 4>::st(y + at, o);
}

// TRO: storage of y, the residual stream (= TR, the type of `res`, unless the encoder's fp16 stream starts or ends at this launch)
template <typename TX, typename TR, typename TZ = TX, typename TRO = TR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const TR* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     TRO* __restrict__ y, TZ* __restrict__ z, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int d, float eps,
                                                     uint32_t thresh, float dscale, uint32_t seed, bf16_t* __restrict__ y16,
                                                     const uint32_t* __restrict__ seed_dev,
                                                     const bf16_t* __restrict__ pos = nullptr, bf16_t* __restrict__ q16 = nullptr,
                                                     const bf16_t* __restrict__ res_hi = nullptr) {
    if (seed_dev) seed ^= *seed_dev * 0x9E3779B1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    if (row >= rows) return;
    float v[LN_MAXIT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < d) {
            float a[4];
            vec<TX, 4>::ld(x + row * d + c, a);
            if (thresh) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    a[e] = drop_keep(seed, (uint32_t)row * (uint32_t)d + (uint32_t)(c + e), thresh) ? a[e] * dscale : 0.f;
            }
            if (res) {
                float r[4];
                ld_stream4<TR>(res, res_hi, row * d + c, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] = a[e]; s += a[e]; }
        }
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = v[it][e] - mu; q += t * t; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < d) {
            float g[4], b[4], o[4];
            vec<float, 4>::ld(gamma + c, g);
            vec<float, 4>::ld(beta + c, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mu) * rs * g[e] + b[e];
            st_stream4<TRO>(y, row * d + c, o);
            if (y16) vec<bf16_t, 4>::st(y16 + row * d + c, o);
            if (z) vec<TZ, 4>::st(z + row * d + c, v[it]);
            if (q16) {                                    // the NEXT layer's query operand y + pos (deformable_transformer.py:201)
                float pp[4];
                vec<bf16_t, 4>::ld(pos + row * d + c, pp);
#pragma unroll
                for (int e = 0; e < 4; ++e) pp[e] += o[e];
                vec<bf16_t, 4>::st(q16 + row * d + c, pp);
            }
        }
    }
}

// d = 256, R rows per wave iteration, grid-stride: the loads of R rows (branch, stream, position) are in flight before the first
// reduction.  The one-row kernel above relies on occupancy alone; once the stream shrank to fp16 (260-366 MB per launch instead of
// 364-470) it stopped following the bytes (-4.5 us per launch for -104 MB).  Same arithmetic per row, bit for bit.
template <typename TX, typename TR, typename TZ, typename TRO, int R>
__global__ __launch_bounds__(256) void ln_fwd256_kernel(const TX* __restrict__ x, const TR* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        TRO* __restrict__ y, TZ* __restrict__ z, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int64_t rows, float eps,
                                                        uint32_t thresh, float dscale, uint32_t seed, bf16_t* __restrict__ y16,
                                                        const uint32_t* __restrict__ seed_dev,
                                                        const bf16_t* __restrict__ pos, bf16_t* __restrict__ q16,
                                                        const bf16_t* __restrict__ res_hi) {
    constexpr int d = 256;
    if (seed_dev) seed ^= *seed_dev * 0x9E3779B1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane * 4;
    float g[4], b[4];
    vec<float, 4>::ld(gamma + c, g);
    vec<float, 4>::ld(beta + c, b);
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * R; row0 < rows; row0 += (int64_t)gridDim.x * 4 * R) {
        float v[R][4], r[R][4], pp[R][4];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t row = min(row0 + k, rows - 1);
            vec<TX, 4>::ld(x + row * d + c, v[k]);
            if (res) ld_stream4<TR>(res, res_hi, row * d + c, r[k]);
            if (q16) vec<bf16_t, 4>::ld(pos + row * d + c, pp[k]);
        }
        float s[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t row = min(row0 + k, rows - 1);
            s[k] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = v[k][e];
                if (thresh) a = drop_keep(seed, (uint32_t)row * (uint32_t)d + (uint32_t)(c + e), thresh) ? a * dscale : 0.f;
                if (res) a += r[k][e];
                v[k][e] = a;
                s[k] += a;
            }
        }
        float mu[R], q[R], rs[R];
#pragma unroll
        for (int k = 0; k < R; ++k) mu[k] = wave_sum(s[k]) / (float)d;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            q[k] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = v[k][e] - mu[k]; q[k] += t * t; }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) rs[k] = rsqrtf(wave_sum(q[k]) / (float)d + eps);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (row0 + k < rows) {
                const int64_t row = row0 + k;
                if (lane == 0) {
                    if (mean) mean[row] = mu[k];
                    if (rstd) rstd[row] = rs[k];
                }
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mu[k]) * rs[k] * g[e] + b[e];
                st_stream4<TRO>(y, row * d + c, o);
                if (y16) vec<bf16_t, 4>::st(y16 + row * d + c, o);
                if (z) vec<TZ, 4>::st(z + row * d + c, v[k]);
                if (q16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pp[k][e] += o[e];
                    vec<bf16_t, 4>::st(q16 + row * d + c, pp[k]);
                }
            }
        }
    }
}

// TRO: storage type of the stream gradient dz (= TR unless the encoder's bf16 gradient stream starts here: fp32 in, bf16 out)
template <typename TX, typename TR, typename TRO = TR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TR* __restrict__ dy, const TX* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, TRO* __restrict__ dz,
                                                     TX* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int d,
                                                     uint32_t thresh, float dscale, uint32_t seed,
                                                     const uint32_t* __restrict__ seed_dev) {
    __shared__ float red[2][4][LN_MAXIT * 256];
    if (seed_dev) seed ^= *seed_dev * 0x9E3779B1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float ag[LN_MAXIT][4], ab[LN_MAXIT][4], gm[LN_MAXIT][4];
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
        const int c = (lane + 64 * it) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[it][e] = 0.f; ab[it][e] = 0.f; gm[it][e] = 0.f; }
        if (c < d) vec<float, 4>::ld(gamma + c, gm[it]);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float g[LN_MAXIT][4], xh[LN_MAXIT][4];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int it = 0; it < LN_MAXIT; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < d) {
                float a[4], zz[4];
                vec<TR, 4>::ld(dy + row * d + c, a);
                vec<TX, 4>::ld(z + row * d + c, zz);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[it][e] = (zz[e] - mu) * rs;
                    ag[it][e] += a[e] * xh[it][e];
                    ab[it][e] += a[e];
                    g[it][e] = a[e] * gm[it][e];
                    c1 += g[it][e];
                    c2 += g[it][e] * xh[it][e];
                }
            }
        }
        c1 = wave_sum(c1) / (float)d;
        c2 = wave_sum(c2) / (float)d;
#pragma unroll
        for (int it = 0; it < LN_MAXIT; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < d) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - c1 - xh[it][e] * c2);
                vec<TRO, 4>::st(dz + row * d + c, o);
                if (dx && ((const void*)dx != (const void*)dz || thresh)) {
                    if (thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = drop_keep(seed, (uint32_t)row * (uint32_t)d + (uint32_t)(c + e), thresh) ? o[e] * dscale : 0.f;
                    }
                    vec<TX, 4>::st(dx + row * d + c, o);
                }
            }
        }
    }
    // block reduction of the parameter gradients, then one atomic per column per block
#pragma unroll
    for (int it = 0; it < LN_MAXIT; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wid][(lane + 64 * it) * 4 + e] = ag[it][e];
            red[1][wid][(lane + 64 * it) * 4 + e] = ab[it][e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        const float sg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        atomicAdd(dgamma + c, sg);
        atomicAdd(dbeta + c, sb);
    }
}

// d = 256 (one 4-element vector per lane), R rows per wave iteration.  The kernel above keeps ONE row per wave in flight and lets
// occupancy hide the latency: its time per row does not depend on the bytes of the row, so the bf16 gradient stream (208 instead of
// 312 MB per launch) bought nothing there (66.8 -> 71.8 us).  Here the loads of R rows are issued before the first reduction.
template <typename TX, typename TR, typename TRO, int R>
__global__ __launch_bounds__(256) void ln_bwd256_kernel(const TR* __restrict__ dy, const TX* __restrict__ z,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, TRO* __restrict__ dz,
                                                        TX* __restrict__ dx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int64_t rows,
                                                        uint32_t thresh, float dscale, uint32_t seed,
                                                        const uint32_t* __restrict__ seed_dev) {
    constexpr int d = 256;
    __shared__ float red[2][4][d];
    if (seed_dev) seed ^= *seed_dev * 0x9E3779B1u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = lane * 4;
    float gm[4], ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    vec<float, 4>::ld(gamma + c, gm);
    const bool sep = dx && ((const void*)dx != (const void*)dz || thresh);
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * R; row0 < rows; row0 += (int64_t)gridDim.x * 4 * R) {
        float a[R][4], zz[R][4], mu[R], rs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = min(row0 + r, rows - 1);
            vec<TR, 4>::ld(dy + row * d + c, a[r]);
            vec<TX, 4>::ld(z + row * d + c, zz[r]);
            mu[r] = mean[row]; rs[r] = rstd[row];
        }
        float c1[R], c2[R], g[R][4], xh[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = row0 + r < rows;
            c1[r] = 0.f; c2[r] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float av = ok ? a[r][e] : 0.f;
                xh[r][e] = (zz[r][e] - mu[r]) * rs[r];
                ag[e] += av * xh[r][e];
                ab[e] += av;
                g[r][e] = av * gm[e];
                c1[r] += g[r][e];
                c2[r] += g[r][e] * xh[r][e];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { c1[r] = wave_sum(c1[r]) * (1.f / d); c2[r] = wave_sum(c2[r]) * (1.f / d); }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row0 + r < rows) {
                const int64_t row = row0 + r;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs[r] * (g[r][e] - c1[r] - xh[r][e] * c2[r]);
                vec<TRO, 4>::st(dz + row * d + c, o);
                if (sep) {
                    if (thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = drop_keep(seed, (uint32_t)row * (uint32_t)d + (uint32_t)(c + e), thresh) ? o[e] * dscale : 0.f;
                    }
                    vec<TX, 4>::st(dx + row * d + c, o);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][wid][c + e] = ag[e]; red[1][wid][c + e] = ab[e]; }
    __syncthreads();
    {
        const int cc = threadIdx.x;
        atomicAdd(dgamma + cc, red[0][0][cc] + red[0][1][cc] + red[0][2][cc] + red[0][3][cc]);
        atomicAdd(dbeta + cc, red[1][0][cc] + red[1][1][cc] + red[1][2][cc] + red[1][3][cc]);
    }
}

// ---- GroupNorm on token-major maps ------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// channel-group row access: cpg == 8 (d_model 256 / 32 groups) is one 16-B (bf16) or 32-B (f32) vector per token
template <typename T>
__device__ __forceinline__ void gn_ld(const T* p, float* v, int cpg, bool vec8) {
    if (vec8) vec<T, 8>::ld(p, v);
    else for (int c = 0; c < cpg; ++c) v[c] = io<T>::ld(p + c);
}
template <typename T>
__device__ __forceinline__ void gn_st(T* p, const float* v, int cpg, bool vec8) {
    if (vec8) vec<T, 8>::st(p, v);
    else for (int c = 0; c < cpg; ++c) io<T>::st(p + c, v[c]);
}
constexpr int GN_MAXCPG = 8;

// Workgroup -> (image n, channel group g).  A workgroup reads ONE group's 16-32 bytes out of every token row, so the 8
// groups that share a 128-byte line must run on the same XCD (workgroup b -> XCD b % 8) or every XCD's L2 fetches the
// whole tensor from HBM (measured: 394 MB fetched per launch for a 39 MB input).  Sets of 8 consecutive groups are dealt
// to XCDs round-robin; falls back to the plain order when the counts do not divide.
__device__ __forceinline__ void gn_block_to_ng(int b, int N, int G, int& n, int& g) {
    if (G % 8 == 0 && ((N * (G / 8)) % 8) == 0) {
        const int xcd = b & 7, j = b >> 3, set = (j >> 3) * 8 + xcd, spi = G / 8;   // set = n * spi + (g >> 3)
        n = set / spi;
        g = (set % spi) * 8 + (j & 7);
    } else {
        n = b / G;
        g = b % G;
    }
}

template <typename T, typename TY>
__global__ __launch_bounds__(256) void gn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TY* __restrict__ y,
                                                     float* __restrict__ stats, int HW, int C, int G,
                                                     int64_t x_off, int64_t x_stride, int64_t y_off, int64_t y_stride,
                                                     float eps, bf16_t* __restrict__ xc16) {
    __shared__ float sm[4];
    int n, g;
    gn_block_to_ng(blockIdx.x, gridDim.x / G, G, n, g);
    const int cpg = C / G;
    const bool vec8 = (cpg == 8);
    const T* xb = x + ((int64_t)n * x_stride + x_off) * C + g * cpg;
    TY* yb = y + ((int64_t)n * y_stride + y_off) * C + g * cpg;
    bf16_t* cb = xc16 ? xc16 + ((int64_t)n * x_stride + x_off) * C + g * cpg : nullptr;      // bf16 copy of the INPUT, same layout as x
    float s = 0.f, q = 0.f;
    for (int hw = threadIdx.x; hw < HW; hw += 256) {
        float v[GN_MAXCPG];
        gn_ld<T>(xb + (int64_t)hw * C, v, cpg, vec8);
        for (int c = 0; c < cpg; ++c) { s += v[c]; q += v[c] * v[c]; }
    }
    const float cnt = (float)HW * (float)cpg;
    const float mu = block_sum(s, sm) / cnt;
    const float var = fmaxf(block_sum(q, sm) / cnt - mu * mu, 0.f);
    const float rs = rsqrtf(var + eps);
    if (threadIdx.x == 0) { stats[(n * G + g) * 2] = mu; stats[(n * G + g) * 2 + 1] = rs; }
    float gm[GN_MAXCPG], bt[GN_MAXCPG];
    for (int c = 0; c < cpg; ++c) { gm[c] = gamma[g * cpg + c]; bt[c] = beta[g * cpg + c]; }
    for (int hw = threadIdx.x; hw < HW; hw += 256) {
        float v[GN_MAXCPG];
        gn_ld<T>(xb + (int64_t)hw * C, v, cpg, vec8);
        if (cb) gn_st<bf16_t>(cb + (int64_t)hw * C, v, cpg, vec8);
        for (int c = 0; c < cpg; ++c) v[c] = (v[c] - mu) * rs * gm[c] + bt[c];
        gn_st<TY>(yb + (int64_t)hw * C, v, cpg, vec8);
    }
}

template <typename T, typename TY>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const TY* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int HW, int C, int G,
                                                     int64_t x_off, int64_t x_stride, int64_t y_off, int64_t y_stride) {
    __shared__ float sm[4];
    int n, g;
    gn_block_to_ng(blockIdx.x, gridDim.x / G, G, n, g);
    const int cpg = C / G;
    const bool vec8 = (cpg == 8);
    const int64_t base = ((int64_t)n * x_stride + x_off) * C + g * cpg;
    const int64_t ybase = ((int64_t)n * y_stride + y_off) * C + g * cpg;
    const float mu = stats[(n * G + g) * 2], rs = stats[(n * G + g) * 2 + 1];
    float pg[GN_MAXCPG], pb[GN_MAXCPG], gm[GN_MAXCPG];
    for (int c = 0; c < cpg; ++c) { pg[c] = 0.f; pb[c] = 0.f; gm[c] = gamma[g * cpg + c]; }
    for (int hw = threadIdx.x; hw < HW; hw += 256) {
        float d[GN_MAXCPG], v[GN_MAXCPG];
        gn_ld<TY>(dy + ybase + (int64_t)hw * C, d, cpg, vec8);
        gn_ld<T>(x + base + (int64_t)hw * C, v, cpg, vec8);
        for (int c = 0; c < cpg; ++c) { pg[c] += d[c] * (v[c] - mu) * rs; pb[c] += d[c]; }
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < cpg; ++c) {
        const float tg = block_sum(pg[c], sm), tb = block_sum(pb[c], sm);
        if (threadIdx.x == 0) { atomicAdd(dgamma + g * cpg + c, tg); atomicAdd(dbeta + g * cpg + c, tb); }
        s1 += gm[c] * tb;
        s2 += gm[c] * tg;
    }
    const float cnt = (float)HW * (float)cpg;
    const float m1 = s1 / cnt, m2 = s2 / cnt;
    for (int hw = threadIdx.x; hw < HW; hw += 256) {
        float d[GN_MAXCPG], v[GN_MAXCPG];
        gn_ld<TY>(dy + ybase + (int64_t)hw * C, d, cpg, vec8);
        gn_ld<T>(x + base + (int64_t)hw * C, v, cpg, vec8);
        for (int c = 0; c < cpg; ++c) v[c] = rs * (d[c] * gm[c] - m1 - (v[c] - mu) * rs * m2);
        gn_st<T>(dx + base + (int64_t)hw * C, v, cpg, vec8);
    }
}

// ---- GroupNorm with whole token rows per wave (C = 256, 8 channels per group: the shape of every input_proj level) ----------
// The kernels above give a workgroup ONE group's 16-32 bytes out of every row: every lane of a load touches its own cache line
// (2.6 TB/s for the 60 x 80 level).  Here a workgroup owns GN_RB consecutive rows of one image with all 256 channels -- thread =
// (group = its 8 channels, row lane), a wave reads two whole 512-byte rows per instruction -- and the reductions that span the
// image go through a small caller-owned scratch in a fixed order (deterministic, like the one-workgroup-per-group kernels):
//   forward : gn_rows_stats  -> partial (sum, sum of squares) per (image, row block, group);
//             gn_rows_apply  -> every workgroup adds the partials of its image, normalises its rows (block 0 also stores mean / rstd);
//   backward: gn_rows_bwd_partial -> partial (sum dy xhat, sum dy) per (image, row block, channel);
//             gn_rows_bwd_apply   -> adds them (block 0 of each image also feeds d gamma / d beta), forms the two group means, writes dx.
constexpr int GN_RB = 64;                  // rows per workgroup of the apply kernels
constexpr int GN_NSB = 32;                 // row blocks per image of the partial-sum kernels (an apply workgroup adds that many partials)

template <typename T>
__global__ __launch_bounds__(256) void gn_rows_stats_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int nblk, int rb,
                                                            int64_t x_off, int64_t x_stride) {
    __shared__ float red[2][8][32];
    const int n = blockIdx.x / nblk, blk = blockIdx.x % nblk, g = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const T* xb = x + ((int64_t)n * x_stride + x_off) * 256 + g * 8;
    float s = 0.f, q = 0.f;
    const int rend = min(HW, (blk + 1) * rb);
    int r = blk * rb + rl;
    for (; r + 24 < rend; r += 32) {                                // 4 rows in flight per thread
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) vec<T, 8>::ld(xb + (int64_t)(r + 8 * u) * 256, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) { s += v[u][c]; q += v[u][c] * v[u][c]; }
    }
    for (; r < rend; r += 8) {
        float v[8];
        vec<T, 8>::ld(xb + (int64_t)r * 256, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) { s += v[c]; q += v[c] * v[c]; }
    }
    red[0][rl][g] = s; red[1][rl][g] = q;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int k = threadIdx.x >> 5;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[k][i][g];
        part[((int64_t)blockIdx.x * 32 + g) * 2 + k] = t;
    }
}

template <typename T, typename TY>
__global__ __launch_bounds__(256) void gn_rows_apply_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            TY* __restrict__ y, float* __restrict__ stats, const float* __restrict__ part,
                                                            int HW, int nblk, int nsb, int64_t x_off, int64_t x_stride, int64_t y_off, int64_t y_stride, float eps,
                                                            bf16_t* __restrict__ xc16) {
    __shared__ float smu[32], srs[32];
    const int n = blockIdx.x / nblk, blk = blockIdx.x % nblk, g = threadIdx.x & 31, rl = threadIdx.x >> 5;
    if (threadIdx.x < 32) {
        float s = 0.f, q = 0.f;
        const float2* pp = reinterpret_cast<const float2*>(part) + (int64_t)n * nsb * 32 + g;
        float2 pv[GN_NSB];                                          // all partials requested before the first is used (one round trip)
#pragma unroll
        for (int b = 0; b < GN_NSB; ++b) pv[b] = b < nsb ? pp[b * 32] : make_float2(0.f, 0.f);
#pragma unroll
        for (int b = 0; b < GN_NSB; ++b) { s += pv[b].x; q += pv[b].y; }
        const float cnt = (float)HW * 8.f;
        const float mu = s / cnt;
        const float rs = rsqrtf(fmaxf(q / cnt - mu * mu, 0.f) + eps);
        smu[g] = mu; srs[g] = rs;
        if (blk == 0) { stats[(n * 32 + g) * 2] = mu; stats[(n * 32 + g) * 2 + 1] = rs; }
    }
    __syncthreads();
    const float mu = smu[g], rs = srs[g];
    float gm[8], bt[8];
    vec<float, 8>::ld(gamma + g * 8, gm);
    vec<float, 8>::ld(beta + g * 8, bt);
    const T* xb = x + ((int64_t)n * x_stride + x_off) * 256 + g * 8;
    TY* yb = y + ((int64_t)n * y_stride + y_off) * 256 + g * 8;
    bf16_t* cb = xc16 ? xc16 + ((int64_t)n * x_stride + x_off) * 256 + g * 8 : nullptr;        // bf16 copy of the INPUT (what backward reads), same layout as x
    const int rend = min(HW, (blk + 1) * GN_RB);
    int r = blk * GN_RB + rl;
    for (; r + 24 < rend; r += 32) {                                // 4 rows in flight per thread
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) vec<T, 8>::ld(xb + (int64_t)(r + 8 * u) * 256, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cb) vec<bf16_t, 8>::st(cb + (int64_t)(r + 8 * u) * 256, v[u]);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[u][c] = (v[u][c] - mu) * rs * gm[c] + bt[c];
            vec<TY, 8>::st(yb + (int64_t)(r + 8 * u) * 256, v[u]);
        }
    }
    for (; r < rend; r += 8) {
        float v[8];
        vec<T, 8>::ld(xb + (int64_t)r * 256, v);
        if (cb) vec<bf16_t, 8>::st(cb + (int64_t)r * 256, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (v[c] - mu) * rs * gm[c] + bt[c];
        vec<TY, 8>::st(yb + (int64_t)r * 256, v);
    }
}

template <typename T, typename TY>
__global__ __launch_bounds__(256) void gn_rows_bwd_partial_kernel(const TY* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                                  float* __restrict__ part, int HW, int nblk, int rb, int64_t x_off, int64_t x_stride,
                                                                  int64_t y_off, int64_t y_stride) {
    __shared__ float red[2][8][256];
    const int n = blockIdx.x / nblk, blk = blockIdx.x % nblk, g = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const float mu = stats[(n * 32 + g) * 2], rs = stats[(n * 32 + g) * 2 + 1];
    const T* xb = x + ((int64_t)n * x_stride + x_off) * 256 + g * 8;
    const TY* db = dy + ((int64_t)n * y_stride + y_off) * 256 + g * 8;
    float pg[8], pb[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { pg[c] = 0.f; pb[c] = 0.f; }
#pragma unroll 2
    for (int r = blk * rb + rl; r < min(HW, (blk + 1) * rb); r += 8) {
        float d[8], v[8];
        vec<TY, 8>::ld(db + (int64_t)r * 256, d);
        vec<T, 8>::ld(xb + (int64_t)r * 256, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) { pg[c] += d[c] * (v[c] - mu) * rs; pb[c] += d[c]; }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { red[0][rl][g * 8 + c] = pg[c]; red[1][rl][g * 8 + c] = pb[c]; }
    __syncthreads();
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { tg += red[0][i][threadIdx.x]; tb += red[1][i][threadIdx.x]; }
    part[((int64_t)blockIdx.x * 2) * 256 + threadIdx.x] = tg;
    part[((int64_t)blockIdx.x * 2 + 1) * 256 + threadIdx.x] = tb;
}

template <typename T, typename TY>
__global__ __launch_bounds__(256) void gn_rows_bwd_apply_kernel(const TY* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, T* __restrict__ dx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, const float* __restrict__ part, int HW, int nblk, int nsb,
                                                                int64_t x_off, int64_t x_stride, int64_t y_off, int64_t y_stride) {
    __shared__ float s1[256], s2[256], sm1[32], sm2[32];
    const int n = blockIdx.x / nblk, blk = blockIdx.x % nblk, g = threadIdx.x & 31, rl = threadIdx.x >> 5;
    {
        float tg = 0.f, tb = 0.f;
        const float* pp = part + (int64_t)n * nsb * 512 + threadIdx.x;
        float pgv[GN_NSB], pbv[GN_NSB];
#pragma unroll
        for (int b = 0; b < GN_NSB; ++b) { pgv[b] = b < nsb ? pp[b * 512] : 0.f; pbv[b] = b < nsb ? pp[b * 512 + 256] : 0.f; }
#pragma unroll
        for (int b = 0; b < GN_NSB; ++b) { tg += pgv[b]; tb += pbv[b]; }
        if (blk == 0) { atomicAdd(dgamma + threadIdx.x, tg); atomicAdd(dbeta + threadIdx.x, tb); }
        const float gmc = gamma[threadIdx.x];
        s1[threadIdx.x] = gmc * tb; s2[threadIdx.x] = gmc * tg;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { a += s1[threadIdx.x * 8 + c]; b += s2[threadIdx.x * 8 + c]; }
        const float cnt = (float)HW * 8.f;
        sm1[threadIdx.x] = a / cnt; sm2[threadIdx.x] = b / cnt;
    }
    __syncthreads();
    const float m1 = sm1[g], m2 = sm2[g];
    const float mu = stats[(n * 32 + g) * 2], rs = stats[(n * 32 + g) * 2 + 1];
    float gm[8];
    vec<float, 8>::ld(gamma + g * 8, gm);
    const T* xb = x + ((int64_t)n * x_stride + x_off) * 256 + g * 8;
    const TY* db = dy + ((int64_t)n * y_stride + y_off) * 256 + g * 8;
    T* ob = dx + ((int64_t)n * x_stride + x_off) * 256 + g * 8;
#pragma unroll 2
    for (int r = blk * GN_RB + rl; r < min(HW, (blk + 1) * GN_RB); r += 8) {
        float d[8], v[8];
        vec<TY, 8>::ld(db + (int64_t)r * 256, d);
        vec<T, 8>::ld(xb + (int64_t)r * 256, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = rs * (d[c] * gm[c] - m1 - (v[c] - mu) * rs * m2);
        vec<T, 8>::st(ob + (int64_t)r * 256, v);
    }
}

}  // namespace poet

using namespace poet;

// dispatch over the supported (branch dtype, stream dtype) pairs
#define POET_DT2(dx, dr, LAUNCH)                                                                     \
    do {                                                                                             \
        if (dx == POET_BF16 && dr == POET_BF16) LAUNCH(bf16_t, bf16_t);                              \
        else if (dx == POET_F32 && dr == POET_F32) LAUNCH(float, float);                             \
        else if (dx == POET_BF16 && dr == POET_F32) LAUNCH(bf16_t, float);                           \
        else { set_error("norm: unsupported dtype pair x=%d r=%d", dx, dr); return POET_ERR_UNSUPPORTED; } \
    } while (0)

extern "C" int poet_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* z_out,
                           float* mean, float* rstd, int64_t rows, int d, float eps, float drop_p, uint32_t seed,
                           int dtype_x, int dtype_r, int dtype_y, int dtype_z, void* y_bf16, const void* pos_bf16, void* q_bf16,
                           const void* res_bf16, const uint32_t* seed_dev, void* stream) {
    POET_CHECK(x && gamma && beta && y, POET_ERR_ARG, "ln_fwd: null pointer");
    POET_CHECK((pos_bf16 == nullptr) == (q_bf16 == nullptr), POET_ERR_ARG, "ln_fwd: pos_bf16 and q_bf16 come together");
    if (dtype_z < 0) dtype_z = dtype_x;
    if (dtype_y < 0) dtype_y = dtype_r;
    POET_CHECK(rows > 0 && d > 0 && d % 4 == 0 && d <= LN_MAXIT * 256, POET_ERR_UNSUPPORTED, "ln_fwd: d=%d unsupported", d);
    POET_CHECK(drop_p >= 0.f && drop_p < 1.f, POET_ERR_ARG, "ln_fwd: drop_p");
    const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
    const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    dim3 grid(cdiv(rows, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LN_FWD(TX, TR) ln_fwd_kernel<TX, TR><<<grid, block, 0, st>>>((const TX*)x, (const TR*)res, gamma, beta, (TR*)y, (TX*)z_out, mean, rstd, rows, d, eps, th, sc, seed, (bf16_t*)y_bf16, seed_dev, (const bf16_t*)pos_bf16, (bf16_t*)q_bf16)
    if (dtype_x == POET_F16) {      // (round 6) the branch stored as IEEE fp16 by the projection (PoetGemmDesc.c_f16): fp32 or fp16 stream, bf16 saved sum
        POET_CHECK((dtype_r == POET_F32 || dtype_r == POET_F16) && (dtype_y == POET_F32 || dtype_y == POET_F16) && dtype_z == POET_BF16, POET_ERR_UNSUPPORTED,
                   "ln_fwd: an fp16 branch comes with an fp32 / split stream and a bf16 saved sum (x %d, res %d, y %d, z %d)", dtype_x, dtype_r, dtype_y, dtype_z);
        POET_CHECK(dtype_r != POET_F16 || (res && res_bf16), POET_ERR_ARG, "ln_fwd: a split stream (dtype_r = POET_F16) is the pair (res_bf16, res)");
        POET_CHECK(dtype_y != POET_F16 || y_bf16, POET_ERR_ARG, "ln_fwd: a split stream output (dtype_y = POET_F16) is the pair (y_bf16, y)");
        // d = 256 at >= 4096 rows: R rows per wave iteration, 8 workgroups per CU (POET_LN_FWD_R: 1 = the one-row kernel; A/B aid, read once)
        static const int rper = [] { const char* e = getenv("POET_LN_FWD_R"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? v : 2; }();
        static const int capf = [] { const char* e = getenv("POET_LN_FWD_NB"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();      // (102 080 rows, fp16 stream: 512: 59 / 98 us, 1024: 50 / 68, 2048: 50 / 75, 4096: 46 / 67, all: 44 / 70)
        const bool multi = d == 256 && rows >= 4096 && rper > 1;
        int nbf = cdiv(rows, 4 * rper);
        if (nbf > capf) nbf = capf;
        dim3 gridm(nbf);
#define LN_FWD_H(TR, TRO) do {                                                                                                          \
        if (multi && rper == 2) ln_fwd256_kernel<f16_t, TR, bf16_t, TRO, 2><<<gridm, block, 0, st>>>((const f16_t*)x, (const TR*)res, gamma, beta, (TRO*)y, (bf16_t*)z_out, mean, rstd, rows, eps, th, sc, seed, (bf16_t*)y_bf16, seed_dev, (const bf16_t*)pos_bf16, (bf16_t*)q_bf16, (const bf16_t*)res_bf16); \
        else if (multi) ln_fwd256_kernel<f16_t, TR, bf16_t, TRO, 4><<<gridm, block, 0, st>>>((const f16_t*)x, (const TR*)res, gamma, beta, (TRO*)y, (bf16_t*)z_out, mean, rstd, rows, eps, th, sc, seed, (bf16_t*)y_bf16, seed_dev, (const bf16_t*)pos_bf16, (bf16_t*)q_bf16, (const bf16_t*)res_bf16); \
        else ln_fwd_kernel<f16_t, TR, bf16_t, TRO><<<grid, block, 0, st>>>((const f16_t*)x, (const TR*)res, gamma, beta, (TRO*)y, (bf16_t*)z_out, mean, rstd, rows, d, eps, th, sc, seed, (bf16_t*)y_bf16, seed_dev, (const bf16_t*)pos_bf16, (bf16_t*)q_bf16, (const bf16_t*)res_bf16); \
    } while (0)
        if (dtype_r == POET_F32 && dtype_y == POET_F32) LN_FWD_H(float, float);
        else if (dtype_r == POET_F32) LN_FWD_H(float, f16_t);
        else if (dtype_y == POET_F32) LN_FWD_H(f16_t, float);
        else LN_FWD_H(f16_t, f16_t);
#undef LN_FWD_H
    } else if (dtype_y != dtype_r) {
        set_error("ln_fwd: the stream changes its storage type only behind an fp16 branch (x %d, res %d, y %d)", dtype_x, dtype_r, dtype_y);
        return POET_ERR_UNSUPPORTED;
    } else if (dtype_z == dtype_x) {
        POET_DT2(dtype_x, dtype_r, LN_FWD);
    } else {        // fp32 branch input (the GEMM's accumulators, never rounded to bf16) with the pre-norm sum saved in bf16 for backward
        POET_CHECK(dtype_x == POET_F32 && dtype_r == POET_F32 && dtype_z == POET_BF16, POET_ERR_UNSUPPORTED, "ln_fwd: dtype triple (%d,%d,%d)", dtype_x, dtype_r, dtype_z);
        ln_fwd_kernel<float, float, bf16_t><<<grid, block, 0, st>>>((const float*)x, (const float*)res, gamma, beta, (float*)y, (bf16_t*)z_out, mean, rstd, rows, d, eps, th, sc, seed, (bf16_t*)y_bf16, seed_dev, (const bf16_t*)pos_bf16, (bf16_t*)q_bf16);
    }
#undef LN_FWD
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                           void* dz_out, void* dx_out, float* dgamma, float* dbeta, int64_t rows, int d, float drop_p,
                           uint32_t seed, int dtype_x, int dtype_r, int dtype_dz, const uint32_t* seed_dev, void* stream) {
    POET_CHECK(dy && z && mean && rstd && gamma && dz_out && dgamma && dbeta, POET_ERR_ARG, "ln_bwd: null pointer");
    POET_CHECK(rows > 0 && d > 0 && d % 4 == 0 && d <= LN_MAXIT * 256, POET_ERR_UNSUPPORTED, "ln_bwd: d=%d unsupported", d);
    POET_CHECK(!(drop_p > 0.f && dx_out == dz_out), POET_ERR_ARG, "ln_bwd: dx_out must not alias dz_out when drop_p>0");
    POET_CHECK(!(dtype_x != dtype_dz && (dx_out == nullptr || dx_out == dz_out)), POET_ERR_ARG, "ln_bwd: mixed dtypes need a separate dx_out");
    POET_CHECK(dtype_dz == dtype_r || (dtype_dz == POET_BF16 && dtype_x == POET_BF16) ||
               (dtype_dz == POET_F32 && dtype_r == POET_BF16 && dtype_x == POET_BF16 && d == 256 && rows >= 4096), POET_ERR_UNSUPPORTED,
               "ln_bwd: dz dtype %d with (x, dy) dtypes (%d, %d)", dtype_dz, dtype_x, dtype_r);
    const uint32_t th = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
    const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    int nb = cdiv(rows, 4);
    // One row per wave iteration, no prefetch: occupancy is what hides the load latency, and 4 workgroups per CU is where it stops paying
    // (102 080 x 256, us per launch by grid: 256: 114, 512: 77, 768: 68, 1024: 67, 1280: 80, 2048: 78, 4096: 114 -- not the per-column atomics
    // of the epilogue: with the blocks' column sums stored to a workspace and added by a second launch the larger grids measured 81 / 76 / 90).
    // POET_LN_BWD_NB: the cap (A/B aid, read once).
    static const int cap = [] { const char* e = getenv("POET_LN_BWD_NB"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    hipStream_t st = (hipStream_t)stream;
    // d = 256 with 2-byte saved sums: R rows per wave iteration (POET_LN_BWD_R: 1 = the one-row kernel, A/B aid, read once)
    static const int rper = [] { const char* e = getenv("POET_LN_BWD_R"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 || v == 8 ? v : 4; }();
    if (d == 256 && (rper > 1 || (dtype_dz == POET_F32 && dtype_r == POET_BF16)) && rows >= 4096 && dtype_x == POET_BF16) {
        // ~200 rows per workgroup: every workgroup ends with a 2 x 256-column LDS reduction and 512 atomics, so FEWER, longer-running
        // workgroups win once several rows are in flight per wave (102 080 rows, bf16 stream, us: 256: 44, 384: 43, 512: 40, 768: 43,
        // 1024: 45, 2048: 62; 51 200 rows: 512: 27, 1024: 35; 204 000 rows: 512: 87, 1024: 81)
        static const int forced = [] { const char* e = getenv("POET_LN_BWD_NB"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
        int nb2 = forced ? forced : (int)((rows / 200 + 128) / 256) * 256;
        if (nb2 < 256) nb2 = 256;
        if (nb2 > 1024 && !forced) nb2 = 1024;
        if (nb2 > cdiv(rows, 4 * rper)) nb2 = cdiv(rows, 4 * rper);
        dim3 grid2(nb2), block2(256);
#define LN_BWD256(TR, TRO, R) ln_bwd256_kernel<bf16_t, TR, TRO, R><<<grid2, block2, 0, st>>>((const TR*)dy, (const bf16_t*)z, mean, rstd, gamma, (TRO*)dz_out, (bf16_t*)dx_out, dgamma, dbeta, rows, th, sc, seed, seed_dev)
#define LN_BWD256_R(TR, TRO) do { if (rper == 2) LN_BWD256(TR, TRO, 2); else if (rper == 8) LN_BWD256(TR, TRO, 8); else LN_BWD256(TR, TRO, 4); } while (0)
        if (dtype_r == POET_F32 && dtype_dz == POET_F32) LN_BWD256_R(float, float);
        else if (dtype_r == POET_F32) LN_BWD256_R(float, bf16_t);
        else if (dtype_dz == POET_F32) LN_BWD256_R(bf16_t, float);      // (where the bf16 stream ends: layer 0 hands fp32 to the input projection)
        else LN_BWD256_R(bf16_t, bf16_t);
#undef LN_BWD256_R
#undef LN_BWD256
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    if (nb > cap) nb = cap;
    dim3 grid(nb), block(256);
#define LN_BWD(TX, TR) ln_bwd_kernel<TX, TR><<<grid, block, 0, st>>>((const TR*)dy, (const TX*)z, mean, rstd, gamma, (TR*)dz_out, (TX*)dx_out, dgamma, dbeta, rows, d, th, sc, seed, seed_dev)
    if (dtype_dz != dtype_r)        // the bf16 gradient stream starts here: fp32 d(y) in, bf16 d(z) out (bf16 saved sum)
        ln_bwd_kernel<bf16_t, float, bf16_t><<<grid, block, 0, st>>>((const float*)dy, (const bf16_t*)z, mean, rstd, gamma, (bf16_t*)dz_out, (bf16_t*)dx_out, dgamma, dbeta, rows, d, th, sc, seed, seed_dev);
    else POET_DT2(dtype_x, dtype_r, LN_BWD);
#undef LN_BWD
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int N,
                                  int HW, int C, int G, int64_t x_off, int64_t x_stride, int64_t y_off, int64_t y_stride,
                                  float eps, int dtype_x, int dtype_y, float* scratch, int64_t scratch_floats, void* x_bf16_copy, void* stream) {
    POET_CHECK(x && gamma && beta && y && stats, POET_ERR_ARG, "groupnorm_fwd: null pointer");
    POET_CHECK(!x_bf16_copy || (dtype_x == POET_F32 && (reinterpret_cast<uintptr_t>(x_bf16_copy) & 15) == 0), POET_ERR_ARG,
               "groupnorm_fwd: x_bf16_copy goes with an fp32 x and a 16-byte aligned buffer");
    POET_CHECK(N > 0 && HW > 0 && G > 0 && C % G == 0 && C / G <= GN_MAXCPG, POET_ERR_ARG, "groupnorm_fwd: bad dims (C/G must be <= 8)");
    hipStream_t st = (hipStream_t)stream;
    {
        const int nblk = cdiv(HW, GN_RB), nsb = min(GN_NSB, nblk), rb = cdiv(cdiv(HW, nsb), 8) * 8;
        static const int no_rows = [] { const char* e = getenv("POET_GN_NO_ROWS"); return e && atoi(e) ? 1 : 0; }();
        if (!no_rows && C == 256 && G == 32 && scratch && scratch_floats >= (int64_t)N * nsb * 64 && HW >= 256 &&
            ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31) == 0) {
#define GN_ROWS_FWD(TX, TR)                                                                                                              \
            do {                                                                                                                         \
                gn_rows_stats_kernel<TX><<<dim3(N * nsb), dim3(256), 0, st>>>((const TX*)x, scratch, HW, nsb, rb, x_off, x_stride);     \
                gn_rows_apply_kernel<TX, TR><<<dim3(N * nblk), dim3(256), 0, st>>>((const TX*)x, gamma, beta, (TR*)y, stats, scratch, HW, nblk, nsb, x_off, x_stride, y_off, y_stride, eps, (bf16_t*)x_bf16_copy); \
            } while (0)
            POET_DT2(dtype_x, dtype_y, GN_ROWS_FWD);
#undef GN_ROWS_FWD
            POET_LAUNCH_CHECK();
            return POET_OK;
        }
    }
    dim3 grid(N * G), block(256);
#define GN_FWD(TX, TR) gn_fwd_kernel<TX, TR><<<grid, block, 0, st>>>((const TX*)x, gamma, beta, (TR*)y, stats, HW, C, G, x_off, x_stride, y_off, y_stride, eps, (bf16_t*)x_bf16_copy)
    POET_DT2(dtype_x, dtype_y, GN_FWD);
#undef GN_FWD
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_groupnorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                                  float* dgamma, float* dbeta, int N, int HW, int C, int G, int64_t x_off,
                                  int64_t x_stride, int64_t y_off, int64_t y_stride, int dtype_x, int dtype_y, float* scratch,
                                  int64_t scratch_floats, void* stream) {
    POET_CHECK(dy && x && stats && gamma && dx && dgamma && dbeta, POET_ERR_ARG, "groupnorm_bwd: null pointer");
    POET_CHECK(N > 0 && HW > 0 && G > 0 && C % G == 0 && C / G <= GN_MAXCPG, POET_ERR_ARG, "groupnorm_bwd: bad dims (C/G must be <= 8)");
    hipStream_t st = (hipStream_t)stream;
    {
        const int nblk = cdiv(HW, GN_RB), nsb = min(GN_NSB, nblk), rb = cdiv(cdiv(HW, nsb), 8) * 8;
        static const int no_rows = [] { const char* e = getenv("POET_GN_NO_ROWS"); return e && atoi(e) ? 1 : 0; }();
        if (!no_rows && C == 256 && G == 32 && scratch && scratch_floats >= (int64_t)N * nsb * 512 && HW >= 256 &&
            ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 31) == 0) {
#define GN_ROWS_BWD(TX, TR)                                                                                                              \
            do {                                                                                                                         \
                gn_rows_bwd_partial_kernel<TX, TR><<<dim3(N * nsb), dim3(256), 0, st>>>((const TR*)dy, (const TX*)x, stats, scratch, HW, nsb, rb, x_off, x_stride, y_off, y_stride); \
                gn_rows_bwd_apply_kernel<TX, TR><<<dim3(N * nblk), dim3(256), 0, st>>>((const TR*)dy, (const TX*)x, stats, gamma, (TX*)dx, dgamma, dbeta, scratch, HW, nblk, nsb, x_off, x_stride, y_off, y_stride); \
            } while (0)
            POET_DT2(dtype_x, dtype_y, GN_ROWS_BWD);
#undef GN_ROWS_BWD
            POET_LAUNCH_CHECK();
            return POET_OK;
        }
    }
    dim3 grid(N * G), block(256);
#define GN_BWD(TX, TR) gn_bwd_kernel<TX, TR><<<grid, block, 0, st>>>((const TR*)dy, (const TX*)x, stats, gamma, (TX*)dx, dgamma, dbeta, HW, C, G, x_off, x_stride, y_off, y_stride)
    POET_DT2(dtype_x, dtype_y, GN_BWD);
#undef GN_BWD
    POET_LAUNCH_CHECK();
    return POET_OK;
}
