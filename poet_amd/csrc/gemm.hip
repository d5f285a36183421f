// MFMA GEMM for gfx950 with fused epilogue -- every nn.Linear / 1x1 conv on the PoET hot path and
// their backward contractions (reference call sites listed in include/poet_hip.h).
//
// Shapes on this path are "tall and thin": M = N_img*S tokens (1e5), K in {256, 1024}, N in
// {256..1280}; arithmetic intensity is fixed by K=256 (~128-200 flop/B), i.e. close to the
// machine balance, so the kernel is built around (a) wide coalesced staging with enough bytes in
// flight per CU to cover HBM latency, (b) fp32->bf16 conversion and layout changes (NCHW, K-major
// operands of the backward contractions) done while staging instead of in separate HBM passes,
// (c) an epilogue that applies everything the consumer needs (bias, ReLU, gate, dropout, residual
// add, row mask, head-major value layout, split-K atomics) so no elementwise kernel re-reads C.
//
// Tile: BMxBN per 256-thread workgroup (4 waves as 2x2), each wave (BM/2)x(BN/2) in 16x16 MFMA
// fragments.  One K stage = BKB bytes of K per row in the compute type (128 B for 128x128 tiles,
// 256 B for 64x64 tiles).  LDS image per stage: [row][k] with a 16-B pad per row, double buffered;
// every fragment is one ds_read_b128 per lane (row = lane&15, 16-B chunk = lane>>4) per 64 B of K:
//   bf16: one v_mfma_f32_16x16x32_bf16 per fragment pair per 64 B.
//   f32 : four v_mfma_f32_16x16x4_f32 (element t of both 16-B chunks feeds MFMA t; A and B use the
//         same k permutation, so the contraction is exact).
// Pipeline: global data is held RAW in registers as 16-B chunks from the load of stage t+2 until
// its LDS write one iteration later (into the buffer that was read two stages ago), so a stage's
// loads have a full compute stage plus the other resident workgroup to land, with ONE barrier per
// stage.  K-major operands (stored [K][rows]: dW = dY^T X, dX = dY W, NCHW features) are transposed
// at the LDS write: a thread holds 4 consecutive k-rows x 8 rows and writes 8 x (4 k-values).
#include "common.cuh"

#include "gemm.cuh"

#include <stdlib.h>

namespace poet {

template <typename Src, typename CT, int N> struct cvt_pack;           // N source elements -> packed compute type
template <> struct cvt_pack<bf16_t, bf16_t, 8> { static __device__ __forceinline__ uint4 run(uint4 v) { return v; } };
template <> struct cvt_pack<float, float, 4> { static __device__ __forceinline__ uint4 run(uint4 v) { return v; } };
template <> struct cvt_pack<float, bf16_t, 4> {
    static __device__ __forceinline__ uint2 run(uint4 v) {
        return make_uint2(pack_bf2(__uint_as_float(v.x), __uint_as_float(v.y)), pack_bf2(__uint_as_float(v.z), __uint_as_float(v.w)));
    }
};

// 16 bytes of `base[row*ld + k ...]`, zero-filled outside [0,rtot) x [.,kend); scalar path when unaligned
template <typename Src>
__device__ __forceinline__ uint4 load_chunk(const Src* __restrict__ base, int64_t ld, int row, int rtot, int k, int kend, int vec_ok) {
    constexpr int ES = 16 / sizeof(Src);
    if (row < rtot && k + ES <= kend && vec_ok) return *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + k);
    Src tmp[ES];
#pragma unroll
    for (int e = 0; e < ES; ++e) tmp[e] = (row < rtot && k + e < kend) ? base[(int64_t)row * ld + k + e] : Src(0);
    uint4 r;
    __builtin_memcpy(&r, tmp, 16);
    return r;
}

template <typename Src> __device__ __forceinline__ float elem_of(const uint4& v, int e);
template <> __device__ __forceinline__ float elem_of<float>(const uint4& v, int e) {
    return __uint_as_float(e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w);
}
template <> __device__ __forceinline__ float elem_of<bf16_t>(const uint4& v, int e) {
    const uint32_t w = (e >> 1) == 0 ? v.x : (e >> 1) == 1 ? v.y : (e >> 1) == 2 ? v.z : v.w;
    return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
}

// ---- K-contiguous operand: src[row*ld + k] --------------------------------------------------------
template <typename Src, typename CT, int R, int BKB>
struct LoaderKC {
    static constexpr int ES = 16 / sizeof(Src);                 // source elements per 16-B chunk
    static constexpr int BK = BKB / sizeof(CT);                 // K elements per stage
    static constexpr int CPR = BK / ES;                         // chunks per row
    static constexpr int NCH = R * CPR;
    static constexpr int NI = (NCH + 255) / 256;
    static constexpr int PITCH = BKB + 16;
    static_assert(sizeof(Src) >= sizeof(CT), "bf16 storage with f32 compute is not used");
    uint4 v[NI];

    __device__ __forceinline__ void load(const Src* __restrict__ src, int64_t ld, int r0, int rtot, int k0, int kend, int vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256;
            if (NCH % 256 == 0 || c < NCH) {
                const int row = c / CPR, kc = c % CPR;
                v[i] = load_chunk<Src>(src, ld, r0 + row, rtot, k0 + kc * ES, kend, vec_ok);
            }
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256;
            if (NCH % 256 == 0 || c < NCH) {
                const int row = c / CPR, kc = c % CPR;
                auto w = cvt_pack<Src, CT, ES>::run(v[i]);
                *reinterpret_cast<decltype(w)*>(lds + row * PITCH + kc * (int)sizeof(w)) = w;
            }
        }
    }
    // fp32 source used as bf16 hi + bf16 lo (PoetGemmDesc.b_split): two LDS images, `lo_off` bytes apart
    __device__ __forceinline__ void store_split(char* __restrict__ lds, int lo_off, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256;
            if (NCH % 256 == 0 || c < NCH) {
                const int row = c / CPR, kc = c % CPR;
                const float f0 = __uint_as_float(v[i].x), f1 = __uint_as_float(v[i].y), f2 = __uint_as_float(v[i].z), f3 = __uint_as_float(v[i].w);
                const uint2 hi = make_uint2(pack_bf2(f0, f1), pack_bf2(f2, f3));
                const uint2 lo = make_uint2(pack_bf2(f0 - __uint_as_float(hi.x << 16), f1 - __uint_as_float(hi.x & 0xffff0000u)),
                                            pack_bf2(f2 - __uint_as_float(hi.y << 16), f3 - __uint_as_float(hi.y & 0xffff0000u)));
                *reinterpret_cast<uint2*>(lds + row * PITCH + kc * 8) = hi;
                *reinterpret_cast<uint2*>(lds + lo_off + row * PITCH + kc * 8) = lo;
            }
        }
    }
};

// ---- K-major operand: src[k*ld + row]; an item = 4 consecutive k-rows x 8 rows, transposed at the LDS write ----
// (items are numbered [row group / 4][k quad][row group % 4]: 4 consecutive threads read 4 x 32 = 128 contiguous bytes of a
// memory row -- with the k quad fastest every lane of a load touched its own cache line: 24 us for a 7 MB operand)
template <typename Src, typename CT, int R, int BKB>
struct LoaderKM {
    static constexpr int ES = 16 / sizeof(Src);
    static constexpr int BK = BKB / sizeof(CT);
    static constexpr int NKQ = BK / 4;
    static constexpr int ITEMS = NKQ * (R / 8);
    static_assert((R / 8) % 4 == 0, "item numbering: row groups in fours");
    static constexpr int NI = (ITEMS + 255) / 256;
    static constexpr int CPI = 8 / ES;                          // 16-B chunks per k-row of an item (1 bf16, 2 f32)
    static constexpr int PITCH = BKB + 16;
    uint4 v[NI][4][CPI];

    __device__ __forceinline__ void load(const Src* __restrict__ src, int64_t ld, int r0, int rtot, int k0, int kend, int vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int item = tid + i * 256;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                const int kq = (item >> 2) % NKQ, rg = (item / (4 * NKQ)) * 4 + (item & 3);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int cc = 0; cc < CPI; ++cc)      // the memory row of a K-major source is k; its columns are tile rows
                        v[i][kk][cc] = load_chunk<Src>(src, ld, k0 + kq * 4 + kk, kend, r0 + rg * 8 + cc * ES, rtot, vec_ok);
            }
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int item = tid + i * 256;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                const int kq = (item >> 2) % NKQ, rg = (item / (4 * NKQ)) * 4 + (item & 3);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f0 = elem_of<Src>(v[i][0][j / ES], j % ES), f1 = elem_of<Src>(v[i][1][j / ES], j % ES);
                    const float f2 = elem_of<Src>(v[i][2][j / ES], j % ES), f3 = elem_of<Src>(v[i][3][j / ES], j % ES);
                    char* dst = lds + (rg * 8 + j) * PITCH;
                    if constexpr (sizeof(CT) == 2) *reinterpret_cast<uint2*>(dst + kq * 8) = make_uint2(pack_bf2(f0, f1), pack_bf2(f2, f3));
                    else *reinterpret_cast<float4*>(dst + kq * 16) = make_float4(f0, f1, f2, f3);
                }
            }
        }
    }
};

template <typename Src, typename CT, int R, int BKB, bool KM> struct LoaderSel;
template <typename Src, typename CT, int R, int BKB> struct LoaderSel<Src, CT, R, BKB, false> { using type = LoaderKC<Src, CT, R, BKB>; };
template <typename Src, typename CT, int R, int BKB> struct LoaderSel<Src, CT, R, BKB, true> { using type = LoaderKM<Src, CT, R, BKB>; };

template <typename TA, typename TB, typename TC, typename CT, int BM, int BN, int BKB, bool AKM, bool BKM, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmK p) {
    constexpr int BK = BKB / sizeof(CT);
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int PITCH = BKB + 16;
    constexpr int KSTEPS = BKB / 64;
    constexpr int STAGE = (BM + (SPLIT ? 2 : 1) * BN) * PITCH;     // SPLIT: B as two bf16 images (hi, lo) of the fp32 weight
    static_assert(!SPLIT || (!BKM && sizeof(TB) == 4 && sizeof(CT) == 2), "b_split: fp32 [N,K] weight, bf16 MFMA");
#define POET_STORE_B(dst)                                         \
    do {                                                          \
        if constexpr (SPLIT) lb.store_split((dst), BN * PITCH, tid); \
        else lb.store((dst), tid);                                \
    } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const PoetGemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware tile order: workgroup L runs on XCD L%8 (observed dispatch), so give each XCD a CONTIGUOUS range of
    // tiles in (m-panel major, n minor) order -- the N tiles of one A panel then share one L2 instead of 8.  Pure
    // speed heuristic: any placement computes the same result.
    const int gx = gridDim.x, ntile = gx * gridDim.y;
    int L = blockIdx.y * gx + blockIdx.x;
    {
        const int q = ntile >> 3, r = ntile & 7, xcd = L & 7, idx = L >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (L % gx) * BN, m0 = (L / gx) * BM;
    const int zb = blockIdx.z / d.splitk, sk = blockIdx.z % d.splitk;
    const int kbeg = sk * p.kchunk;
    const int kend = min(d.K, kbeg + p.kchunk);
    if (kbeg >= kend) return;

    const TA* A = reinterpret_cast<const TA*>(d.A) + (int64_t)zb * d.strideA;
    const TB* B = reinterpret_cast<const TB*>(d.B) + (int64_t)zb * d.strideB;

    typename LoaderSel<TA, CT, BM, BKB, AKM>::type la;
    typename LoaderSel<TB, CT, BN, BKB, BKM>::type lb;

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: stage 0 -> LDS buffer 0, stage 1 -> registers
    la.load(A, d.lda, m0, d.M, kbeg, kend, p.a_vec, tid);
    lb.load(B, d.ldb, n0, d.N, kbeg, kend, p.b_vec, tid);
    la.store(smem, tid);
    POET_STORE_B(smem + BM * PITCH);
    if (kbeg + BK < kend) {
        la.load(A, d.lda, m0, d.M, kbeg + BK, kend, p.a_vec, tid);
        lb.load(B, d.ldb, n0, d.N, kbeg + BK, kend, p.b_vec, tid);
    }
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // registers hold stage k0+BK (issued one iteration ago): park it in the other LDS buffer (its last readers
        // passed the barrier that ended the previous iteration), then put stage k0+2BK in flight.
        if (k0 + BK < kend) {
            char* nxt = smem + (buf ^ 1) * STAGE;
            la.store(nxt, tid);
            POET_STORE_B(nxt + BM * PITCH);
            if (k0 + 2 * BK < kend) {
                la.load(A, d.lda, m0, d.M, k0 + 2 * BK, kend, p.a_vec, tid);
                lb.load(B, d.ldb, n0, d.N, k0 + 2 * BK, kend, p.b_vec, tid);
            }
        }
        const char* As = smem + buf * STAGE;
        const char* Bs = As + BM * PITCH;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            uint4 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                af[i] = *reinterpret_cast<const uint4*>(As + (wm * WM + i * 16 + frow) * PITCH + ks * 64 + fchunk * 16);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bfr[j] = *reinterpret_cast<const uint4*>(Bs + (wn * WN + j * 16 + frow) * PITCH + ks * 64 + fchunk * 16);
            uint4 blo[SPLIT ? FN : 1];
            if constexpr (SPLIT) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    blo[j] = *reinterpret_cast<const uint4*>(Bs + (BN + wn * WN + j * 16 + frow) * PITCH + ks * 64 + fchunk * 16);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (sizeof(CT) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                        if constexpr (SPLIT)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, blo[j]), acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].x), __uint_as_float(bfr[j].x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].y), __uint_as_float(bfr[j].y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].z), __uint_as_float(bfr[j].z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].w), __uint_as_float(bfr[j].w), acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: accumulators -> LDS [row][col] (f32) -> 8 consecutive columns per lane --------------------------
    // so bias / gate / residual are 16-32 B vector loads and C leaves as 16-B (bf16x8) or 2x16-B (f32x8) stores that
    // are contiguous along the row across lanes (the MFMA layout itself would give 2-byte stores 4 rows apart).
    constexpr int CP = BN + 4;                               // fp32 pitch of the staged C tile
    // (launch_one sizes the dynamic LDS as max(2 * STAGE, BM * CP * 4): the C tile re-uses the staging buffers)
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                Cs[(wm * WM + i * 16 + fchunk * 4 + t) * CP + wn * WN + j * 16 + frow] = acc[i][j][t] * d.alpha;
    __syncthreads();

    TC* C = reinterpret_cast<TC*>(d.C) + (int64_t)zb * d.strideC;
    const bool use_atomic = (d.atomic != 0) || (d.splitk > 1);
    if (use_atomic) {
        if constexpr (sizeof(TC) == 4) {
            float* Cf = reinterpret_cast<float*>(C);
            for (int idx = tid; idx < BM * BN; idx += 256) {
                const int row = idx / BN, col = idx - row * BN;
                if (m0 + row < d.M && n0 + col < d.N) atomicAdd(Cf + (int64_t)(m0 + row) * d.ldc + n0 + col, Cs[row * CP + col]);
            }
        }
        return;
    }
    const float* bias = d.bias ? d.bias + (int64_t)zb * d.stride_bias : nullptr;
    // (gate_ref and add_src are laid out like C: they take C's batch stride)
    const TC* addp = d.add_src ? reinterpret_cast<const TC*>(d.add_src) + (int64_t)zb * d.strideC : nullptr;
    const TC* gate = d.gate_ref ? reinterpret_cast<const TC*>(d.gate_ref) + (int64_t)zb * d.strideC : nullptr;
    constexpr int C8 = BN / 8;
    for (int idx = tid; idx < BM * C8; idx += 256) {
        const int row = idx / C8, c8 = idx - row * C8;
        const int grow = m0 + row, gcol = n0 + c8 * 8;
        if (grow >= d.M || gcol >= d.N) continue;
        float v[8];
        vec<float, 8>::ld(Cs + row * CP + c8 * 8, v);
        const bool full = (gcol + 8 <= d.N) && p.c_vec;
        int64_t off;
        if (d.out_mode == 1) {
            const int hn = grow / d.hm_S, hs = grow - hn * d.hm_S, hm = gcol / d.hm_D, hd = gcol - hm * d.hm_D;
            off = (((int64_t)hn * d.hm_M + hm) * d.hm_S + hs) * d.hm_D + hd;
        } else {
            off = (int64_t)grow * d.ldc + gcol;
        }
        if (bias) {
            if (full) { float b[8]; vec<float, 8>::ld(bias + gcol, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += b[e]; }
            else { for (int e = 0; e < 8; ++e) if (gcol + e < d.N) v[e] += bias[gcol + e]; }
        }
        if (d.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (gate) {
            float g[8];
            if (full) vec<TC, 8>::ld(gate + (int64_t)grow * d.ldc + gcol, g);
            else for (int e = 0; e < 8; ++e) g[e] = (gcol + e < d.N) ? io<TC>::ld(gate + (int64_t)grow * d.ldc + gcol + e) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = g[e] > 0.f ? v[e] * d.gate_scale : 0.f;
        }
        if (p.drop_thresh) {
            const uint32_t base = (uint32_t)grow * (uint32_t)d.N + (uint32_t)gcol;
            const uint32_t sd = d.seed ^ (d.seed_dev ? *d.seed_dev * 0x9E3779B1u : 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = drop_keep(sd, base + e, p.drop_thresh) ? v[e] * p.drop_scale : 0.f;
        }
        if (addp) {
            float a8[8];
            if (full) vec<TC, 8>::ld(addp + (int64_t)grow * d.ld_add + gcol, a8);
            else for (int e = 0; e < 8; ++e) a8[e] = (gcol + e < d.N) ? io<TC>::ld(addp + (int64_t)grow * d.ld_add + gcol + e) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a8[e];
        }
        if (d.row_mask && d.row_mask[grow]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if constexpr (sizeof(TC) == 2) {
            if (d.c_f16) {                                      // 2-byte outputs as IEEE fp16 (PoetGemmDesc.c_f16)
                uint16_t* C16 = reinterpret_cast<uint16_t*>(C);
                if (full) st8_f16(C16 + off, v);
                else if (d.out_mode == 1) {
                    for (int e = 0; e < 8; ++e) if (gcol + e < d.N) {
                        const int cc = gcol + e, hn = grow / d.hm_S, hs = grow - hn * d.hm_S, hm = cc / d.hm_D, hd = cc - hm * d.hm_D;
                        C16[(((int64_t)hn * d.hm_M + hm) * d.hm_S + hs) * d.hm_D + hd] = (uint16_t)(pack_h2(v[e], 0.f) & 0xffffu);
                    }
                } else {
                    for (int e = 0; e < 8; ++e) if (gcol + e < d.N) C16[off + e] = (uint16_t)(pack_h2(v[e], 0.f) & 0xffffu);
                }
                continue;
            }
        }
        if (full) {
            vec<TC, 8>::st(C + off, v);
        } else if (d.out_mode == 1) {
            for (int e = 0; e < 8; ++e) if (gcol + e < d.N) {
                const int cc = gcol + e, hn = grow / d.hm_S, hs = grow - hn * d.hm_S, hm = cc / d.hm_D, hd = cc - hm * d.hm_D;
                io<TC>::st(C + (((int64_t)hn * d.hm_M + hm) * d.hm_S + hs) * d.hm_D + hd, v[e]);
            }
        } else {
            for (int e = 0; e < 8; ++e) if (gcol + e < d.N) io<TC>::st(C + off + e, v[e]);
        }
    }
}

template <typename TA, typename TB, typename TC, typename CT, int BM, int BN, int BKB, bool AKM, bool BKM, bool SPLIT = false>
static void launch_one(const GemmK& p, hipStream_t st) {
    constexpr int LDS_ST = 2 * (BM + (SPLIT ? 2 : 1) * BN) * (BKB + 16), LDS_C = BM * (BN + 4) * 4;
    constexpr int LDS = LDS_ST > LDS_C ? LDS_ST : LDS_C;
    auto kern = gemm_kernel<TA, TB, TC, CT, BM, BN, BKB, AKM, BKM, SPLIT>;
    static unsigned long long attr_done = 0;          // per instantiation and device
    lds_attr_once(reinterpret_cast<const void*>(kern), LDS, attr_done);
    const PoetGemmDesc& d = p.d;
    dim3 grid(cdiv(d.N, BN), cdiv(d.M, BM), d.batch * d.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, p);
}

template <typename TA, typename TB, typename TC, typename CT, int BM, int BN, int BKB>
static void launch_layout(const GemmK& p, hipStream_t st) {
    const int key = p.d.a_kmajor * 2 + p.d.b_kmajor;
    switch (key) {
        case 0: launch_one<TA, TB, TC, CT, BM, BN, BKB, false, false>(p, st); break;
        case 1: launch_one<TA, TB, TC, CT, BM, BN, BKB, false, true>(p, st); break;
        case 2: launch_one<TA, TB, TC, CT, BM, BN, BKB, true, false>(p, st); break;
        default: launch_one<TA, TB, TC, CT, BM, BN, BKB, true, true>(p, st); break;
    }
}

// tile choice: 0 = 128x128 (BKB 128), 1 = 64x64 (BKB 256), 2 = 32x32 (BKB 256; fp32-MFMA problems with few rows, so the
// 1/16-rate f32 matrix pipe of more than a handful of CUs is used)
static int tile_choice(const PoetGemmDesc& d) {
    const int64_t big_blocks = (int64_t)cdiv(d.M, 128) * cdiv(d.N, 128) * d.batch * d.splitk;
    // (fewer than 48 big tiles leave most of the 256 CUs idle: the 3x3 stride-2 conv of the extra level is 1280 x 256 x 2304)
    if (d.M >= 256 && d.N >= 128 && !(d.compute == POET_F32 && big_blocks < 128) && !(d.compute != POET_F32 && big_blocks < 48)) return 0;
    if (d.compute == POET_F32 && (int64_t)cdiv(d.M, 64) * cdiv(d.N, 64) * d.batch * d.splitk < 256)
        return (d.splitk == 1) ? 3 : 2;       // 3: whole-K stages (K=256 fp32 in one), the latency-bound 320-row decoder GEMMs
    return 1;
}

// b_split (fp32 [N,K] weight as bf16 hi + lo): the two B images make a 128-byte K stage of the 128x128 tile 110 KB of LDS,
// i.e. ONE 4-wave workgroup per CU and nothing to overlap its barriers with (measured: 4.4x slower than the plain kernel);
// 64-byte stages keep two workgroups per CU (61 KB).  POET_SPLIT_BKB=128 selects the wide stage (A/B aid).
template <typename TA, typename TB, typename TC, typename CT, int BM, int BN, int BKB>
static void launch_split_layout(const GemmK& p, hipStream_t st) {
    if (p.d.a_kmajor) launch_one<TA, TB, TC, CT, BM, BN, BKB, true, false, true>(p, st);
    else launch_one<TA, TB, TC, CT, BM, BN, BKB, false, false, true>(p, st);
}
static int split_bkb() {
    static const int v = [] { const char* e = getenv("POET_SPLIT_BKB"); return e ? atoi(e) : 64; }();
    return v;
}

template <typename TA, typename TB, typename TC, typename CT>
static void launch_tile(const GemmK& p, hipStream_t st) {
    const int tc = tile_choice(p.d);
    if constexpr (sizeof(TB) == 4 && sizeof(CT) == 2) {
        if (p.d.b_split) {                          // (validated by poet_gemm: b_kmajor == 0)
            if (tc == 0 && split_bkb() == 128) launch_split_layout<TA, TB, TC, CT, 128, 128, 128>(p, st);
            else if (tc == 0) launch_split_layout<TA, TB, TC, CT, 128, 128, 64>(p, st);
            else launch_split_layout<TA, TB, TC, CT, 64, 64, 128>(p, st);
            return;
        }
    }
    if (tc == 0) launch_layout<TA, TB, TC, CT, 128, 128, 128>(p, st);
    else if (tc == 1) launch_layout<TA, TB, TC, CT, 64, 64, 256>(p, st);
    else if constexpr (sizeof(CT) == 4) {
        if (tc == 3) launch_layout<TA, TB, TC, CT, 32, 32, 1024>(p, st);
        else launch_layout<TA, TB, TC, CT, 32, 32, 256>(p, st);
    }
}

static bool vec_ok(const void* ptr, int64_t ld, int64_t stride) {
    return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && (ld % 8 == 0) && (stride % 8 == 0);
}

}  // namespace poet

static thread_local int g_last_path = POET_GEMM_PATH_NONE;
extern "C" int poet_gemm_last_path(void) { return g_last_path; }

extern "C" int poet_gemm(const PoetGemmDesc* desc, void* stream) {
    using namespace poet;
    POET_CHECK(desc != nullptr, POET_ERR_ARG, "poet_gemm: null descriptor");
    GemmK p;
    p.d = *desc;
    PoetGemmDesc& d = p.d;
    POET_CHECK(d.A && d.B && d.C, POET_ERR_ARG, "poet_gemm: null A/B/C");
    POET_CHECK(d.M > 0 && d.N > 0 && d.K > 0, POET_ERR_ARG, "poet_gemm: bad dims %d %d %d", d.M, d.N, d.K);
    POET_CHECK(d.A2 == nullptr, POET_ERR_UNSUPPORTED, "poet_gemm: A2 prologue not implemented");
    if (d.batch < 1) d.batch = 1;
    if (d.splitk < 1) d.splitk = 1;
    const bool atomic = d.atomic || d.splitk > 1;
    // weight-gradient form (A = dY and B = X both stored [rows][.], fp32 accumulate): `bias` is then an OUTPUT, the fp32
    // [M] vector that receives += the column sums of A, i.e. the bias gradient that always accompanies a weight gradient
    const bool dw_form = atomic && d.a_kmajor && d.b_kmajor;
    float* ysum = dw_form ? const_cast<float*>(d.bias) : nullptr;
    if (atomic) {
        POET_CHECK(d.c_dtype == POET_F32, POET_ERR_ARG, "poet_gemm: atomic/split-K needs fp32 C");
        POET_CHECK((!d.bias || dw_form) && !d.act && !d.gate_ref && !d.add_src && d.drop_p == 0.f && !d.row_mask, POET_ERR_ARG,
                   "poet_gemm: atomic/split-K allows no epilogue");
    }
    if (d.seg_sums) {                                      // per-segment column sums of A out of the weight-gradient pass (ABI v4)
        POET_CHECK(dw_form && d.batch == 1 && !ysum, POET_ERR_ARG, "poet_gemm: seg_sums belongs to the weight-gradient form (a_kmajor, b_kmajor, atomic), batch 1, bias NULL");
        POET_CHECK(d.seg_n >= 1 && d.seg_n <= 8 && d.seg_period >= 64 && d.K % d.seg_period == 0 && d.ld_seg >= d.M && d.seg_start[0] == 0 &&
                   d.seg_start[d.seg_n] == d.seg_period, POET_ERR_ARG, "poet_gemm: seg_sums: 1..8 segments covering a period >= 64 that divides the row count");
        for (int i = 0; i < d.seg_n; ++i) POET_CHECK(d.seg_start[i] <= d.seg_start[i + 1], POET_ERR_ARG, "poet_gemm: seg_start must ascend");
    }
    if (d.B_alt)                                           // rows m >= m_alt of C pair with B_alt (ABI v4)
        POET_CHECK(dw_form && d.batch == 1 && d.m_alt > 0 && d.m_alt < d.M && d.ldb_alt >= d.N, POET_ERR_ARG,
                   "poet_gemm: B_alt belongs to the weight-gradient form, batch 1, 0 < m_alt < M");       // (one launch needs m_alt % 256 == 0; else two products)
    POET_CHECK(d.drop_p >= 0.f && d.drop_p < 1.f, POET_ERR_ARG, "poet_gemm: drop_p");
    if (d.c_f16)
        POET_CHECK(d.c_dtype == POET_BF16 && !d.add_src && !d.gate_ref && !atomic, POET_ERR_ARG,
                   "poet_gemm: c_f16 needs a 2-byte C (c_dtype POET_BF16) and no add_src / gate_ref / split-K");
    if (d.out_mode == 1) POET_CHECK(d.hm_M > 0 && d.hm_S > 0 && d.hm_D > 0 && d.hm_M * d.hm_D == d.N, POET_ERR_ARG, "poet_gemm: head-major dims");
    if (d.b_split && d.B_lo)
        POET_CHECK(d.b_dtype == POET_BF16 && d.compute == POET_BF16 && !atomic, POET_ERR_ARG,
                   "poet_gemm: b_split with B_lo needs two bf16 weight images, bf16 compute, no split-K");
    else if (d.b_split)
        POET_CHECK(d.b_dtype == POET_F32 && d.compute == POET_BF16 && !d.b_kmajor && !atomic, POET_ERR_ARG,
                   "poet_gemm: b_split needs an fp32 [N,K] weight, bf16 compute, no split-K");
    else
        POET_CHECK(d.B_lo == nullptr, POET_ERR_ARG, "poet_gemm: B_lo without b_split");
    const int tcs = tile_choice(d);
    const int BKB = d.b_split ? (tcs == 0 ? split_bkb() : 128) : (tcs == 0 ? 128 : (tcs == 3 ? 1024 : 256));
    const int BK = BKB / (d.compute == POET_BF16 ? 2 : 4);
    p.kchunk = cdiv(cdiv(d.K, d.splitk), BK) * BK;
    p.a_vec = vec_ok(d.A, d.lda, d.strideA);
    p.b_vec = vec_ok(d.B, d.ldb, d.strideB);
    p.c_vec = vec_ok(d.C, d.ldc, d.strideC) && (!d.add_src || vec_ok(d.add_src, d.ld_add, 0)) && (!d.gate_ref || vec_ok(d.gate_ref, d.ldc, 0)) &&
              (!d.bias || vec_ok(d.bias, 8, d.stride_bias)) && (d.out_mode != 1 || d.hm_D % 8 == 0);
    p.drop_thresh = d.drop_p > 0.f ? drop_thresh(d.drop_p) : 0u;
    p.drop_scale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
    if (d.gate_scale == 0.f) d.gate_scale = 1.f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    if (d.B_alt && !gemm_dwr_try(p, st)) {
        // no kernel but the DMA ring pairs two inputs in one launch: (per-segment sums over ALL of A first, then) two plain products
        if (d.seg_sums) {
            POET_CHECK(d.ld_seg == d.M, POET_ERR_UNSUPPORTED, "poet_gemm: seg_sums with ld_seg != M needs the DMA-ring weight-gradient kernel");
            int64_t segs[10];
            for (int i = 0; i <= d.seg_n; ++i) segs[i] = d.seg_start[i];
            const int rc = poet_colsum(d.A, d.lda, d.seg_sums, d.K / d.seg_period, d.seg_period, d.M, segs, d.seg_n, d.a_dtype, stream);
            if (rc) return rc;
        }
        const int esz = d.a_dtype == POET_F32 ? 4 : 2;
        PoetGemmDesc lo = *desc, hi = *desc;
        lo.seg_sums = nullptr; hi.seg_sums = nullptr; lo.B_alt = nullptr; hi.B_alt = nullptr;
        lo.M = d.m_alt;
        hi.M = d.M - d.m_alt;
        hi.A = reinterpret_cast<const char*>(d.A) + (int64_t)d.m_alt * esz;
        hi.B = d.B_alt; hi.ldb = d.ldb_alt;
        hi.C = reinterpret_cast<char*>(d.C) + (int64_t)d.m_alt * d.ldc * 4;
        if (d.bias) hi.bias = d.bias + d.m_alt;
        int rc = poet_gemm(&lo, stream);
        if (rc) return rc;
        return poet_gemm(&hi, stream);
    }
    if (d.B_alt) {
        g_last_path = POET_GEMM_PATH_DW;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    if (d.seg_sums && gemm_dwr_try(p, st)) {                // the DMA-ring dW forms the per-segment sums in its own pass over A
        g_last_path = POET_GEMM_PATH_DW;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    if (d.seg_sums) {                                       // every other kernel: a column-sum launch of its own over A, then the plain product
        POET_CHECK(d.ld_seg == d.M, POET_ERR_UNSUPPORTED, "poet_gemm: seg_sums with ld_seg != M needs the DMA-ring weight-gradient kernel");
        int64_t segs[10];
        for (int i = 0; i <= d.seg_n; ++i) segs[i] = d.seg_start[i];
        const int rc = poet_colsum(d.A, d.lda, d.seg_sums, d.K / d.seg_period, d.seg_period, d.M, segs, d.seg_n, d.a_dtype, stream);
        if (rc) return rc;
        d.seg_sums = nullptr;
    }
    if (gemm_dwr_try(p, st) || gemm_dw_try(p, st)) {        // streaming dW (gemm_dwr.hip: DMA ring, wide shapes; gemm_dw.hip); both fuse the bias gradient of the dW form
        g_last_path = POET_GEMM_PATH_DW;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    if (gemm_small_try(p, st)) {                            // latency-oriented 320-row kernels (gemm_small.hip); same fusion
        g_last_path = POET_GEMM_PATH_SMALL;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    if (ysum) {                            // generic path: the column sums are a separate launch
        POET_CHECK(d.batch == 1, POET_ERR_UNSUPPORTED, "poet_gemm: batched bias-gradient output only in the <= 1024-row kernels");
        const int rc = poet_colsum(d.A, d.lda, ysum, 1, d.K, d.M, nullptr, 1, d.a_dtype, stream);
        if (rc) return rc;
        d.bias = nullptr;
    }
    if (gemm_pipe_try(p, st)) {                             // plain N = 256, K >= 512 products: the deep-pipeline kernel (gemm_pipe.hip)
        g_last_path = POET_GEMM_PATH_PIPE;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    POET_CHECK(d.B_lo == nullptr, POET_ERR_UNSUPPORTED,
               "poet_gemm: two-image split weights (B_lo) only in the long-K kernel (N = 256, K >= 512, K %% 64 == 0, M >= 4096)");
#ifdef POET_PROBE_KERNELS
    if (gemm_wr_try(p, st)) {                               // (probe build only: profiles/probes/kernels/gemm_wr.hip, POET_GEMM_WR=1)
        g_last_path = POET_GEMM_PATH_STREAM;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
#endif
    if (gemm_ws_try(p, st)) {
        g_last_path = POET_GEMM_PATH_STREAM;
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
    g_last_path = POET_GEMM_PATH_TILED;
    const int key = (d.a_dtype << 3) | (d.b_dtype << 2) | (d.c_dtype << 1) | d.compute;
    switch (key) {
        case (POET_BF16 << 3) | (POET_F32 << 2) | (POET_BF16 << 1) | POET_BF16:
            launch_tile<bf16_t, float, bf16_t, bf16_t>(p, st); break;
        case (POET_BF16 << 3) | (POET_F32 << 2) | (POET_F32 << 1) | POET_BF16:
            launch_tile<bf16_t, float, float, bf16_t>(p, st); break;
        case (POET_BF16 << 3) | (POET_BF16 << 2) | (POET_F32 << 1) | POET_BF16:
            launch_tile<bf16_t, bf16_t, float, bf16_t>(p, st); break;
        case (POET_BF16 << 3) | (POET_BF16 << 2) | (POET_BF16 << 1) | POET_BF16:
            launch_tile<bf16_t, bf16_t, bf16_t, bf16_t>(p, st); break;
        case (POET_F32 << 3) | (POET_F32 << 2) | (POET_F32 << 1) | POET_F32:
            launch_tile<float, float, float, float>(p, st); break;
        case (POET_F32 << 3) | (POET_F32 << 2) | (POET_BF16 << 1) | POET_BF16:
            launch_tile<float, float, bf16_t, bf16_t>(p, st); break;
        default:
            POET_CHECK(false, POET_ERR_UNSUPPORTED, "poet_gemm: unsupported dtype combo a=%d b=%d c=%d compute=%d",
                       d.a_dtype, d.b_dtype, d.c_dtype, d.compute);
    }
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_gemm_dw_multi(const float* const* dy, const float* const* x, float* const* dw, float* const* db, int n_lists, int n,
                                  const int* n_out, const int* k_in, int rows, const int64_t* ldy, const int64_t* ldx, void* stream) {
    using namespace poet;
    POET_CHECK(dy && x && dw && n_out && k_in && ldy && ldx && n_lists >= 1 && n_lists <= 8 && n >= 1 && n <= 8, POET_ERR_ARG,
               "poet_gemm_dw_multi: 1..8 lists of 1..8 problems");
    POET_CHECK(rows > 0 && rows <= 1024, POET_ERR_UNSUPPORTED, "poet_gemm_dw_multi: rows %d not in 1..1024", rows);
    POET_CHECK(gemm_small_dw_multi(dy, x, dw, db, n_lists, n, n_out, k_in, rows, ldy, ldx, reinterpret_cast<hipStream_t>(stream)),
               POET_ERR_UNSUPPORTED, "poet_gemm_dw_multi: unsupported problem (null operand, unaligned x, k_in not a multiple of 4)");
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_gemm_dw_list(const float* const* dy, const float* const* x, float* const* dw, float* const* db, int n,
                                 int n_out, int k_in, int rows, int64_t ldy, int64_t ldx, int64_t ldw, void* stream) {
    using namespace poet;
    POET_CHECK(dy && x && dw && n >= 1 && n <= 8, POET_ERR_ARG, "poet_gemm_dw_list: 1..8 problems");
    POET_CHECK(n_out > 0 && k_in > 0 && rows > 0 && rows <= 1024, POET_ERR_UNSUPPORTED, "poet_gemm_dw_list: rows %d not in 1..1024", rows);
    for (int i = 0; i < n; ++i) POET_CHECK(dy[i] && x[i] && dw[i], POET_ERR_ARG, "poet_gemm_dw_list: null operand %d", i);
    POET_CHECK(gemm_small_dw_list(dy, x, dw, db, n, n_out, k_in, rows, ldy, ldx, ldw, reinterpret_cast<hipStream_t>(stream)),
               POET_ERR_UNSUPPORTED, "poet_gemm_dw_list: unsupported problem");
    POET_LAUNCH_CHECK();
    return POET_OK;
}
