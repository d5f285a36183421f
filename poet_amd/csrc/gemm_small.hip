// Latency-oriented fp32 GEMM for the 320-row decoder / pose-head Linears (reference: nn.MultiheadAttention in/out
// projections, decoder FFN and MSDeformAttn query-side projections, models/deformable_transformer.py:253-292; MLP heads,
// models/pose_estimation_transformer.py:106-122,684-688).  At M = N_img * Q = 320 rows these are ~40 MFLOP each: nothing
// is bandwidth- or MFMA-bound, the cost is the dependent chain launch -> load -> LDS -> barrier -> MFMA -> LDS -> store of
// the tiled kernel.  Here one wave owns one 16x16 output tile and loads BOTH operands straight from global memory in
// MFMA operand layout -- no LDS, no barrier, one memory round trip:
//   v_mfma_f32_16x16x4_f32 takes A[i = lane&15][k = lane>>4]; the contraction is invariant under a permutation of k that
//   is applied to both operands, so lane group g = lane>>4 is given the 16-byte pieces g, g+4, g+8, ... of a row: one load
//   instruction of the wave then covers 64 contiguous bytes of each of its 16 rows -- 16 cache-line requests.  (Giving
//   each group a contiguous quarter of K instead makes every load touch 64 different lines: measured 6.8 us for the
//   320x256x256 Linear with everything in L2, the vector-memory unit's line rate, not latency, being the bound.)
// Four accumulators (element x/y/z/w of each float4) break the 64-deep dependent MFMA chain.
#include "gemm.cuh"

namespace poet {

namespace {

__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// fold the 4 waves' partial 16x16 tiles (lane layout preserved) through LDS; afterwards wave 0 holds the sum
__device__ __forceinline__ f32x4_t fold_waves(f32x4_t v, float* red, int lane, int wid) {
    if (wid) *reinterpret_cast<f32x4_t*>(red + ((wid - 1) * 64 + lane) * 4) = v;
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4_t*>(red + (w * 64 + lane) * 4);
    }
    return v;
}

// ---- C[m][n] = epi(sum_k A[m][k] * W[n][k])   (forward; W rows K-contiguous) ----
// ---- C[m][n] = epi(sum_k A[m][k] * W[k][n])   (input gradient; BKM: W stored [K][N]) ----
// SPLIT = false: one wave per tile, 4 tiles per workgroup, whole K per wave (K <= 256: a single memory round trip).
// SPLIT = true : one tile per workgroup, wave w reduces K range [w K/4, (w+1) K/4), partial tiles folded through LDS.
template <bool BKM, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_small_kernel(const GemmK p) {
    __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4];
    const PoetGemmDesc& d = p.d;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tn = (d.N + 15) >> 4, tile = SPLIT ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wid;
    const int m0 = (tile / tn) << 4, n0 = (tile % tn) << 4;
    if (m0 >= d.M) return;                                              // SPLIT: workgroup-uniform
    const int r = lane & 15, g = lane >> 4;
    const int kw = SPLIT ? d.K >> 2 : d.K, kbase = SPLIT ? wid * kw : 0; // this wave's K range
    const int kq = kw >> 2;                                             // floats per lane group, multiple of 4: float4 j of group g = floats [16 j + 4 g, +4) of the range
    const int ncol = min(n0 + r, d.N - 1);                              // ragged N: clamp the operand, mask the store
    const int64_t zb = blockIdx.y;                                      // batch entry (independent problems of one shape)
    const float* ap = reinterpret_cast<const float*>(d.A) + zb * d.strideA + (int64_t)min(m0 + r, d.M - 1) * d.lda + kbase + g * 4;
    const float* wb = reinterpret_cast<const float*>(d.B) + zb * d.strideB;
    const float* wp = BKM ? wb + (int64_t)(kbase + g * 4) * d.ldb + ncol : wb + (int64_t)ncol * d.ldb + kbase + g * 4;
    f32x4_t acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k0 = 0; k0 < kq; k0 += 64) {                               // 64 floats per operand per trip
        float4 a[16], w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = k0 + i * 4 < kq;
            a[i] = ok ? *reinterpret_cast<const float4*>(ap + (k0 + i * 4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!BKM) {
                w[i] = ok ? *reinterpret_cast<const float4*>(wp + (k0 + i * 4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* q = wp + (int64_t)((k0 + i * 4) * 4) * d.ldb;
                w[i] = ok ? make_float4(q[0], q[d.ldb], q[2 * d.ldb], q[3 * d.ldb]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[0] = mfma4(a[i].x, w[i].x, acc[0]);
            acc[1] = mfma4(a[i].y, w[i].y, acc[1]);
            acc[2] = mfma4(a[i].z, w[i].z, acc[2]);
            acc[3] = mfma4(a[i].w, w[i].w, acc[3]);
        }
    }
    f32x4_t c4 = acc[0] + acc[1] + acc[2] + acc[3];
    if constexpr (SPLIT) {
        c4 = fold_waves(c4, red, lane, wid);
        if (wid) return;
    }
    // lane holds C[m0 + 4g + t][n0 + r], t = 0..3: 16 lanes of a row are 64 contiguous bytes
    const int col = n0 + r;
    if (col >= d.N) return;
    const float b = d.bias ? d.bias[zb * d.stride_bias + col] : 0.f;
    const uint32_t sd = d.seed ^ (d.seed_dev ? *d.seed_dev * 0x9E3779B1u : 0u);
    // (gate_ref and add_src are laid out like C: they take C's batch stride)
    const float* addp = d.add_src ? reinterpret_cast<const float*>(d.add_src) + zb * d.strideC : nullptr;
    const float* gate = d.gate_ref ? reinterpret_cast<const float*>(d.gate_ref) + zb * d.strideC : nullptr;
    float* C = reinterpret_cast<float*>(d.C) + zb * d.strideC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = m0 + g * 4 + t;
        if (row >= d.M) continue;
        float v = c4[t] * d.alpha + b;
        if (d.act == 1) v = fmaxf(v, 0.f);
        if (gate) v = gate[(int64_t)row * d.ldc + col] > 0.f ? v * d.gate_scale : 0.f;
        if (p.drop_thresh) v = drop_keep(sd, ((uint32_t)zb * (uint32_t)d.M + (uint32_t)row) * (uint32_t)d.N + (uint32_t)col, p.drop_thresh) ? v * p.drop_scale : 0.f;
        if (addp) v += addp[(int64_t)row * d.ld_add + col];
        C[(int64_t)row * d.ldc + col] = v;
    }
}

// ---- C[m][n] = epi(sum_k A[m][k] * W[k][n]),  W stored [K][N] with N % 4 == 0: the input gradient of a Linear ----
// The weight rows run along N here, so a lane that owned ONE column (the layout above) would read W with 4-byte loads, 64
// of them per 16 k.  Instead a lane loads 16 bytes = 4 CONSECUTIVE columns of one k row and feeds them to 4 MFMAs: tile j
// of the wave's 16 x 64 output holds the columns n0 + 4 r + j (a permutation of the tile's columns that the store undoes),
// all four sharing the A fragment.  Per 16 k: one 16-byte load of A and four of W (256 contiguous bytes per k row) for 16
// MFMAs.  The reduction is split over the NW = K / 64 (4..16) waves of the workgroup -- 64 MFMAs and one memory round trip per
// wave -- and the partial tiles meet in LDS laid out [wave][lane][t][j], so that thread (row, column quad) of the first four
// waves finds its 4 output columns as ONE 16-byte read per partial and writes them with one 16-byte store.
// Measured: 320 x 256 <- 1024 12.5 -> 9.5 us with everything in L2; 14.44 -> 14.26 ms per training step (44 such launches, operands cold).
constexpr int BKM_TN = 64;                  // output columns per workgroup
__global__ __launch_bounds__(1024) void gemm_small_bkm_kernel(const GemmK p, int kw) {
    extern __shared__ __attribute__((aligned(16))) float part[];       // [NW][64 lanes][4 t][4 j]
    const PoetGemmDesc& d = p.d;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int tn = (d.N + BKM_TN - 1) / BKM_TN;
    const int m0 = (blockIdx.x / tn) << 4, n0 = (blockIdx.x % tn) * BKM_TN;
    const int r = lane & 15, g = lane >> 4;
    const int64_t zb = blockIdx.y;
    const int kb = wid * kw, kend = min(kb + kw, d.K);                  // this wave's K range (kw: a multiple of 16)
    // uniform (SGPR) row base + one 32-bit per-lane byte offset per operand: 20 loads, 2 address registers
    const char* ab = reinterpret_cast<const char*>(reinterpret_cast<const float*>(d.A) + zb * d.strideA + kb);
    const char* wb = reinterpret_cast<const char*>(reinterpret_cast<const float*>(d.B) + zb * d.strideB + (int64_t)kb * d.ldb);
    const int ncol = min(n0 + 4 * r, d.N - 4);                          // ragged N: clamp the operand, mask the store
    const uint32_t a_lane = (uint32_t)(min(m0 + r, d.M - 1) * (int)d.lda + g * 4) * 4u;
    const uint32_t w_lane = (uint32_t)(g * 4 * (int)d.ldb + ncol) * 4u;
    const int64_t wrow = d.ldb * 4;                                     // bytes per k row of W
    f32x4_t acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k0 = 0; kb + k0 < kend; k0 += 64) {                        // 64 k per trip: 4 + 16 loads of 16 bytes in flight
        float4 a[4], w[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = kb + k0 + i * 16 < kend;                    // (K % 16 == 0: the 16 k of a piece are in or out together)
            a[i] = ok ? *reinterpret_cast<const float4*>(ab + (k0 + i * 16) * 4 + a_lane) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                w[i][c] = ok ? *reinterpret_cast<const float4*>(wb + (k0 + i * 16 + c) * wrow + w_lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ac[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[0] = mfma4(ac[c], w[i][c].x, acc[0]);
                acc[1] = mfma4(ac[c], w[i][c].y, acc[1]);
                acc[2] = mfma4(ac[c], w[i][c].z, acc[2]);
                acc[3] = mfma4(ac[c], w[i][c].w, acc[3]);
            }
        }
    }
    // lane (r, g) holds C[m0 + 4 g + t][n0 + 4 r + j] in acc[j][t]
    float* mine = part + ((wid * 64 + lane) << 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(mine + t * 4) = make_float4(acc[0][t], acc[1][t], acc[2][t], acc[3][t]);
    __syncthreads();
    if (threadIdx.x >= 256) return;
    const int row = threadIdx.x >> 4, r4 = threadIdx.x & 15;            // output row of the tile, column quad
    const float* src = part + ((((row >> 2) << 4) + r4) << 4) + (row & 3) * 4;
    float4 sum = *reinterpret_cast<const float4*>(src);
    for (int w2 = 1; w2 < nw; ++w2) {
        const float4 v = *reinterpret_cast<const float4*>(src + (w2 << 10));
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const int grow = m0 + row, col = n0 + 4 * r4;
    if (grow >= d.M || col >= d.N) return;                              // (N % 4 == 0: a quad is in or out as a whole)
    float v[4] = {sum.x * d.alpha, sum.y * d.alpha, sum.z * d.alpha, sum.w * d.alpha};
    const uint32_t sd = d.seed ^ (d.seed_dev ? *d.seed_dev * 0x9E3779B1u : 0u);
    const float* addp = d.add_src ? reinterpret_cast<const float*>(d.add_src) + zb * d.strideC : nullptr;
    const float* gate = d.gate_ref ? reinterpret_cast<const float*>(d.gate_ref) + zb * d.strideC : nullptr;
    float* C = reinterpret_cast<float*>(d.C) + zb * d.strideC;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (d.bias) v[e] += d.bias[zb * d.stride_bias + col + e];
        if (d.act == 1) v[e] = fmaxf(v[e], 0.f);
        if (gate) v[e] = gate[(int64_t)grow * d.ldc + col + e] > 0.f ? v[e] * d.gate_scale : 0.f;
        if (p.drop_thresh) v[e] = drop_keep(sd, ((uint32_t)zb * (uint32_t)d.M + (uint32_t)grow) * (uint32_t)d.N + (uint32_t)(col + e), p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
        if (addp) v[e] += addp[(int64_t)grow * d.ld_add + col + e];
    }
    float* cp = C + (int64_t)grow * d.ldc + col;
    if (p.c_vec) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
    else { cp[0] = v[0]; cp[1] = v[1]; cp[2] = v[2]; cp[3] = v[3]; }
}

// ---- dW[n1][n2] += sum_r Y[r][n1] * X[r][n2];  db[n1] += sum_r Y[r][n1]  (rows r = the reduction, both operands [rows][.]) ----
// One 16x16 tile per workgroup; wave w reduces rows [w K/4, (w+1) K/4), lane group g a quarter of those: at 320 rows a lane
// issues 2 x 20 independent 4-byte loads (one memory round trip), 20 MFMAs, and the partial tiles are folded through
// LDS.  The tile has one owner, so the accumulation onto the existing gradient is a plain read-modify-write.
// P: anything with the PoetGemmDesc field names the body reads (`p.d`): the launch record itself, or the list kernel's
// per-problem stand-in below
template <typename P>
__device__ __forceinline__ void small_dw_body(const P& p, float* red, float (*reds)[16]) {
    const auto& d = p.d;                                                // M = n1, N = n2, K = rows
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tn = (d.N + 15) >> 4, tile = blockIdx.x;
    const int m0 = (tile / tn) << 4, n0 = (tile % tn) << 4;
    const int r = lane & 15, g = lane >> 4;
    const int kq = (d.K + 15) >> 4;                                     // rows per (wave, lane group)
    const int rbase = (wid * 4 + g) * kq;
    const int ycol = min(m0 + r, d.M - 1), xcol = min(n0 + r, d.N - 1);
    const int64_t zb = blockIdx.y;                                      // batch entry
    const float* yp = reinterpret_cast<const float*>(d.A) + zb * d.strideA + ycol;
    const float* xp = reinterpret_cast<const float*>(d.B) + zb * d.strideB + xcol;
    f32x4_t acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float ysum = 0.f;
    for (int k0 = 0; k0 < kq; k0 += 32) {
        float y[32], x[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int row = rbase + k0 + i;
            const bool ok = k0 + i < kq && row < d.K;
            y[i] = ok ? yp[(int64_t)row * d.lda] : 0.f;
            x[i] = ok ? xp[(int64_t)row * d.ldb] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i & 3] = mfma4(y[i], x[i], acc[i & 3]);
            ysum += y[i];
        }
    }
    const bool do_sum = d.bias && n0 == 0;                              // workgroup-uniform
    if (do_sum) {
        ysum += __shfl_xor(ysum, 16, 64);
        ysum += __shfl_xor(ysum, 32, 64);
        if (g == 0) reds[wid][r] = ysum;
    }
    f32x4_t c4 = fold_waves(acc[0] + acc[1] + acc[2] + acc[3], red, lane, wid);     // (barrier inside)
    if (wid) return;
    float* C = reinterpret_cast<float*>(d.C) + zb * d.strideC;
    const int col = n0 + r;
    if (col < d.N) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = m0 + g * 4 + t;
            if (row < d.M) C[(int64_t)row * d.ldc + col] += c4[t];
        }
    }
    if (do_sum && g == 0 && m0 + r < d.M) const_cast<float*>(d.bias)[zb * d.stride_bias + m0 + r] += reds[0][r] + reds[1][r] + reds[2][r] + reds[3][r];
}

// ---- the same weight gradient with 16 x 64 tiles (n2 % 4 == 0): the structure of gemm_small_bkm_kernel with A = Y^T ----
// 16 x 16 tiles re-read every 16-column strip of Y n2/16 times and of X n1/16 times (205 MB through the L1s for the 8 MB of the
// decoder's 1024 x 256 FFN gradient).  Here lane (r, g) of a wave reads Y[k][m0 + r] (4 bytes; 16 lanes = 64 contiguous bytes of
// a row) and X[k][n0 + 4 r .. + 3] (16 bytes) and feeds four column-permuted MFMA tiles; wave w reduces rows [64 w, 64 w + 64);
// the partial tiles meet in LDS as in the input-gradient kernel and thread (row, column quad) adds 4 consecutive columns onto
// the gradient with one 16-byte read-modify-write.  The bias gradient (column sums of Y) comes out of the same loads.
template <typename P>
__device__ __forceinline__ void small_dw64_body(const P& p, float* part, float* bsum) {
    const auto& d = p.d;                                                // M = n1, N = n2, K = rows
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int tn = (d.N + BKM_TN - 1) / BKM_TN;
    const int m0 = (blockIdx.x / tn) << 4, n0 = (blockIdx.x % tn) * BKM_TN;
    const int r = lane & 15, g = lane >> 4;
    const int64_t zb = blockIdx.y;
    const int kb = wid * 64, kend = min(kb + 64, d.K);                  // this wave's rows
    const int ycol = min(m0 + r, d.M - 1), xcol = min(n0 + 4 * r, d.N - 4);
    const float* yp = reinterpret_cast<const float*>(d.A) + zb * d.strideA + ycol;
    const float* xp = reinterpret_cast<const float*>(d.B) + zb * d.strideB + xcol;
    f32x4_t acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float y[16];
    float4 x[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                   // MFMA (i, c): lane group g brings row kb + 16 i + 4 g + c
            const int row = kb + i * 16 + g * 4 + c;
            const bool ok = row < kend;
            const int rr = ok ? row : 0;
            y[i * 4 + c] = ok ? yp[(int64_t)rr * d.lda] : 0.f;
            x[i * 4 + c] = ok ? *reinterpret_cast<const float4*>(xp + (int64_t)rr * d.ldb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float ysum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc[0] = mfma4(y[e], x[e].x, acc[0]);
        acc[1] = mfma4(y[e], x[e].y, acc[1]);
        acc[2] = mfma4(y[e], x[e].z, acc[2]);
        acc[3] = mfma4(y[e], x[e].w, acc[3]);
        ysum += y[e];
    }
    const bool do_sum = d.bias && n0 == 0;                              // workgroup-uniform
    if (do_sum) {
        ysum += __shfl_xor(ysum, 16, 64);
        ysum += __shfl_xor(ysum, 32, 64);
        if (g == 0) bsum[wid * 16 + r] = ysum;
    }
    float* mine = part + ((wid * 64 + lane) << 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(mine + t * 4) = make_float4(acc[0][t], acc[1][t], acc[2][t], acc[3][t]);
    __syncthreads();
    if (threadIdx.x >= 256) return;
    const int row = threadIdx.x >> 4, r4 = threadIdx.x & 15;
    const float* src = part + ((((row >> 2) << 4) + r4) << 4) + (row & 3) * 4;
    float4 sum = *reinterpret_cast<const float4*>(src);
    for (int w2 = 1; w2 < nw; ++w2) {
        const float4 v = *reinterpret_cast<const float4*>(src + (w2 << 10));
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const int grow = m0 + row, col = n0 + 4 * r4;
    if (grow < d.M && col < d.N) {                                      // (n2 % 4 == 0: a quad is in or out as a whole)
        float* cp = reinterpret_cast<float*>(d.C) + zb * d.strideC + (int64_t)grow * d.ldc + col;
        cp[0] += sum.x; cp[1] += sum.y; cp[2] += sum.z; cp[3] += sum.w;
    }
    if (do_sum && threadIdx.x < 16 && m0 + (int)threadIdx.x < d.M) {
        float t = 0.f;
        for (int w2 = 0; w2 < nw; ++w2) t += bsum[w2 * 16 + threadIdx.x];
        const_cast<float*>(d.bias)[zb * d.stride_bias + m0 + threadIdx.x] += t;
    }
}

__global__ __launch_bounds__(1024) void gemm_small_dw64_kernel(const GemmK p) {
    extern __shared__ __attribute__((aligned(16))) float part[];
    __shared__ float bsum[16 * 16];
    small_dw64_body(p, part, bsum);
}

__global__ __launch_bounds__(256) void gemm_small_dw_kernel(const GemmK p) {
    __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4];
    __shared__ float reds[4][16];
    small_dw_body(p, red, reds);
}

// ---- the same weight gradient for up to 8 problems of ONE shape whose operands live at unrelated addresses ----
// (the dW + db of one Linear of every decoder layer: dY_l and X_l are separate allocations, so there is no batch
// stride).  grid.y = problem; its pointers come out of the kernel-argument segment through a constant-address-space
// pointer at a uniform offset (scalar loads; a run-time subscript into the by-value struct would go through scratch).
constexpr int DW_LIST_MAX = 8;
struct DwList {
    const float* Y[DW_LIST_MAX];
    const float* X[DW_LIST_MAX];
    float* C[DW_LIST_MAX];
    float* ysum[DW_LIST_MAX];
    int M, N, K;
    int64_t lda, ldb, ldc;
};
struct DwOne {                                                          // the fields small_dw_body reads
    struct {
        const void *A, *B;
        void* C;
        const float* bias;
        int M, N, K;
        int64_t lda, ldb, ldc, strideA, strideB, strideC, stride_bias;
    } d;
};

__global__ __launch_bounds__(256) void gemm_small_dw_list_kernel(const DwList l) {
    __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4];
    __shared__ float reds[4][16];
    typedef const __attribute__((address_space(4))) DwList* kernarg_t;
    kernarg_t ka = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int z = blockIdx.y;
    DwOne one;
    one.d.A = ka->Y[z]; one.d.B = ka->X[z]; one.d.C = ka->C[z]; one.d.bias = ka->ysum[z];
    one.d.M = ka->M; one.d.N = ka->N; one.d.K = ka->K;
    one.d.lda = ka->lda; one.d.ldb = ka->ldb; one.d.ldc = ka->ldc;
    one.d.strideA = one.d.strideB = one.d.strideC = one.d.stride_bias = 0;      // (the body offsets by blockIdx.y * stride)
    (void)l;
    small_dw_body(one, red, reds);
}

}  // namespace

__global__ __launch_bounds__(1024) void gemm_small_dw64_list_kernel(const DwList l) {
    extern __shared__ __attribute__((aligned(16))) float part[];
    __shared__ float bsum[16 * 16];
    typedef const __attribute__((address_space(4))) DwList* kernarg_t;
    kernarg_t ka = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int z = blockIdx.y;
    DwOne one;
    one.d.A = ka->Y[z]; one.d.B = ka->X[z]; one.d.C = ka->C[z]; one.d.bias = ka->ysum[z];
    one.d.M = ka->M; one.d.N = ka->N; one.d.K = ka->K;
    one.d.lda = ka->lda; one.d.ldb = ka->ldb; one.d.ldc = ka->ldc;
    one.d.strideA = one.d.strideB = one.d.strideC = one.d.stride_bias = 0;
    (void)l;
    small_dw64_body(one, part, bsum);
}

// ---- ... and for up to DW_MULTI_MAX such lists of DIFFERENT shapes in one launch: every deferred weight gradient of the decoder ----
// (7 Linears x 5 layers: as 7 launches of ~16 us each they were latency chains on a few dozen workgroups, one after the other; as block
// ranges of ONE launch they overlap.  grid.y = list * n + problem, grid.x = the largest list's tile count: the others' surplus exits.)
constexpr int DW_MULTI_MAX = 8;
struct DwMulti {
    DwList l[DW_MULTI_MAX];
    int n;                                                              // problems per list
};
__global__ __launch_bounds__(1024) void gemm_small_dw64_multi_kernel(const DwMulti mm) {
    extern __shared__ __attribute__((aligned(16))) float part[];
    __shared__ float bsum[16 * 16];
    typedef const __attribute__((address_space(4))) DwMulti* kernarg_t;
    kernarg_t ka = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int j = blockIdx.y / ka->n, z = blockIdx.y - j * ka->n;
    DwOne one;
    one.d.M = ka->l[j].M; one.d.N = ka->l[j].N; one.d.K = ka->l[j].K;
    if ((int)blockIdx.x >= ((one.d.M + 15) >> 4) * ((one.d.N + BKM_TN - 1) / BKM_TN)) return;
    one.d.A = ka->l[j].Y[z]; one.d.B = ka->l[j].X[z]; one.d.C = ka->l[j].C[z]; one.d.bias = ka->l[j].ysum[z];
    one.d.lda = ka->l[j].lda; one.d.ldb = ka->l[j].ldb; one.d.ldc = ka->l[j].ldc;
    // (small_dw64_body offsets by blockIdx.y * stride: zero strides, the pointers above are the problem's own)
    one.d.strideA = one.d.strideB = one.d.strideC = one.d.stride_bias = 0;
    (void)mm;
    small_dw64_body(one, part, bsum);
}

// n2 a multiple of 4, X rows 16-byte aligned, <= 1024 rows (16 waves): the 16 x 64 tiling
static bool dw64_ok(const void* X, int64_t ldx, int n2, int rows) {
    static const int off = [] { const char* e = getenv("POET_SMALL_NO_DW64"); return e && atoi(e) ? 1 : 0; }();
    return !off && n2 % 4 == 0 && n2 >= 4 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && rows <= 1024;
}
static void dw64_attr() {
    static unsigned long long attr_done[2] = {0, 0};
    lds_attr_once(reinterpret_cast<const void*>(gemm_small_dw64_kernel), 64 * 1024, attr_done[0]);
    lds_attr_once(reinterpret_cast<const void*>(gemm_small_dw64_list_kernel), 64 * 1024, attr_done[1]);
}

bool gemm_small_dw_list(const float* const* Y, const float* const* X, float* const* C, float* const* ysum, int n, int n_out, int k_in,
                        int rows, int64_t ldy, int64_t ldx, int64_t ldc, hipStream_t st) {
    if (n < 1 || n > DW_LIST_MAX || rows > 1024) return false;
    DwList l{};
    for (int i = 0; i < n; ++i) { l.Y[i] = Y[i]; l.X[i] = X[i]; l.C[i] = C[i]; l.ysum[i] = ysum ? ysum[i] : nullptr; }
    l.M = n_out; l.N = k_in; l.K = rows; l.lda = ldy; l.ldb = ldx; l.ldc = ldc;
    bool wide = dw64_ok(X[0], ldx, k_in, rows);
    for (int i = 1; i < n && wide; ++i) wide = (reinterpret_cast<uintptr_t>(X[i]) & 15) == 0;
    if (wide) {
        const int nw = max(4, (rows + 63) / 64);       // (the fold and the store use 256 threads)
        dw64_attr();
        hipLaunchKernelGGL(gemm_small_dw64_list_kernel, dim3(((n_out + 15) >> 4) * ((k_in + BKM_TN - 1) / BKM_TN), n), dim3(64 * nw), (size_t)nw * 4096, st, l);
        return true;
    }
    const int tiles = ((n_out + 15) >> 4) * ((k_in + 15) >> 4);
    hipLaunchKernelGGL(gemm_small_dw_list_kernel, dim3(tiles, n), dim3(256), 0, st, l);
    return true;
}

bool gemm_small_dw_multi(const float* const* Y, const float* const* X, float* const* C, float* const* ysum, int nl, int n, const int* n_out,
                         const int* k_in, int rows, const int64_t* ldy, const int64_t* ldx, hipStream_t st) {
    if (nl < 1 || nl > DW_MULTI_MAX || n < 1 || n > DW_LIST_MAX || rows > 1024) return false;
    DwMulti m{};
    m.n = n;
    int tiles = 0;
    for (int j = 0; j < nl; ++j) {
        DwList& l = m.l[j];
        for (int i = 0; i < n; ++i) {
            const int k = j * n + i;
            if (!Y[k] || !X[k] || !C[k]) return false;
            if (!dw64_ok(X[k], ldx[j], k_in[j], rows)) return false;
            l.Y[i] = Y[k]; l.X[i] = X[k]; l.C[i] = C[k]; l.ysum[i] = ysum ? ysum[k] : nullptr;
        }
        l.M = n_out[j]; l.N = k_in[j]; l.K = rows; l.lda = ldy[j]; l.ldb = ldx[j]; l.ldc = k_in[j];
        tiles = max(tiles, ((n_out[j] + 15) >> 4) * ((k_in[j] + BKM_TN - 1) / BKM_TN));
    }
    const int nw = max(4, (rows + 63) / 64);
    static unsigned long long attr_done = 0;
    lds_attr_once(reinterpret_cast<const void*>(gemm_small_dw64_multi_kernel), 64 * 1024, attr_done);
    hipLaunchKernelGGL(gemm_small_dw64_multi_kernel, dim3(tiles, nl * n), dim3(64 * nw), (size_t)nw * 4096, st, m);
    return true;
}

bool gemm_small_try(const GemmK& p, hipStream_t st) {
    const PoetGemmDesc& d = p.d;
    static const int disabled = [] { const char* e = getenv("POET_GEMM_NO_SMALL"); return e && atoi(e) ? 1 : 0; }();
    if (disabled) return false;
    if (d.compute != POET_F32 || d.a_dtype != POET_F32 || d.b_dtype != POET_F32 || d.c_dtype != POET_F32) return false;
    if (d.batch < 1 || d.batch > 65535 || d.A2 || d.row_mask || d.out_mode) return false;
    const bool dw_form = d.a_kmajor && d.b_kmajor && (d.atomic || d.splitk > 1);
    if (dw_form) {                                                      // M = n_out, N = k_in, K = rows
        if (d.K > 1024 || d.alpha != 1.f) return false;
        if (dw64_ok(d.B, d.ldb, d.N, d.K) && d.strideB % 4 == 0) {
            const int nw = max(4, (d.K + 63) / 64);     // (the fold and the store use 256 threads)
            dw64_attr();
            hipLaunchKernelGGL(gemm_small_dw64_kernel, dim3(((d.M + 15) >> 4) * ((d.N + BKM_TN - 1) / BKM_TN), d.batch), dim3(64 * nw), (size_t)nw * 4096, st, p);
            return true;
        }
        const int tiles = ((d.M + 15) >> 4) * ((d.N + 15) >> 4);
        hipLaunchKernelGGL(gemm_small_dw_kernel, dim3(tiles, d.batch), dim3(256), 0, st, p);
        return true;
    }
    if (d.a_kmajor || d.splitk != 1 || d.atomic) return false;
    const bool bkm_form = d.b_kmajor && d.N % 4 == 0 && d.N >= 4 && p.b_vec;      // (2048 rows there: the pose heads of all decoder layers at once)
    if (d.M > (bkm_form ? 2048 : 1024) || d.K % 16 != 0 || d.K > 4096 || !p.a_vec) return false;   // A rows: 16-byte loads
    if (!d.b_kmajor && !p.b_vec) return false;
    if (d.b_kmajor && d.N % 4 == 0 && d.N >= 4 && p.b_vec && (int64_t)d.M * d.lda < (1 << 29) && (int64_t)d.K * d.ldb < (1 << 29)) {            // weight rows along N: 16 x 64 tiles, K over 4..16 waves
        static const int no_bkm = [] { const char* e = getenv("POET_SMALL_NO_BKM"); return e && atoi(e) ? 1 : 0; }();
        if (!no_bkm) {
            int nw = (d.K + 63) / 64;
            nw = nw < 4 ? 4 : (nw > 16 ? 16 : nw);
            const int kw = (((d.K + nw - 1) / nw) + 15) / 16 * 16;
            const int tiles64 = ((d.M + 15) >> 4) * ((d.N + BKM_TN - 1) / BKM_TN);
            static unsigned long long attr_done = 0;
            lds_attr_once(reinterpret_cast<const void*>(gemm_small_bkm_kernel), 64 * 1024, attr_done);
            hipLaunchKernelGGL(gemm_small_bkm_kernel, dim3(tiles64, d.batch), dim3(64 * nw), (size_t)nw * 4096, st, p, kw);
            return true;
        }
    }
    const int tiles = ((d.M + 15) >> 4) * ((d.N + 15) >> 4);
    const bool split = d.K >= 512 && d.K % 64 == 0;                      // long reductions: 4 waves share one tile
    // whole-K tiles: one wave each; single-wave workgroups until every CU has a few (320 tiles = 80 4-wave workgroups would
    // leave two thirds of the chip -- and of its L1s -- idle)
    const int wpb = (split || (int64_t)tiles * d.batch > 2048) ? 4 : 1;
    const dim3 grid(split ? tiles : (tiles + wpb - 1) / wpb, d.batch), block(64 * wpb);
    if (d.b_kmajor) {
        if (split) hipLaunchKernelGGL((gemm_small_kernel<true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_small_kernel<true, false>), grid, block, 0, st, p);
    } else {
        if (split) hipLaunchKernelGGL((gemm_small_kernel<false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_small_kernel<false, false>), grid, block, 0, st, p);
    }
    return true;
}

}  // namespace poet
