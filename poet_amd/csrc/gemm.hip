// MFMA GEMM for gfx950 with fused epilogue -- every nn.Linear / 1x1 conv on the PoET hot path and
// their backward contractions (reference call sites listed in include/poet_hip.h).
//
// Shapes on this path are "tall and thin": M = N_img*S tokens (1e5), K in {256, 1024}, N in
// {256..1280}; arithmetic intensity is fixed by K=256 (~128-200 flop/B), i.e. close to the
// machine balance, so the kernel is built around (a) wide coalesced staging, (b) fp32->bf16
// conversion and layout changes (NCHW, K-major operands for the backward contractions) done
// while staging instead of in separate HBM passes, (c) an epilogue that applies everything the
// consumer needs (bias, ReLU, gate, dropout, residual add, row mask, head-major value layout,
// split-K atomics) so no elementwise kernel ever re-reads the output.
//
// Tile: BMxBN per 256-thread workgroup (4 waves as 2x2), each wave (BM/2)x(BN/2) in 16x16 MFMA
// fragments.  One K stage = 64 bytes of K per row in the compute type (32 bf16 / 16 f32): an LDS
// row is 64 B payload + 16 B pad (80 B pitch) for both operands, stored [row][k], so every
// fragment is one ds_read_b128 per lane (row = lane&15, 16-B chunk = lane>>4).
//   bf16: one v_mfma_f32_16x16x32_bf16 per fragment pair per stage.
//   f32 : four v_mfma_f32_16x16x4_f32 (element t of both 16-B chunks feeds MFMA t; A and B use the
//         same k permutation, so the contraction is exact).
// K-major operands (stored [K][rows], the dW = dY^T X and dX = dY W contractions, NCHW features)
// are transposed in registers while staging: a thread loads 4 consecutive k-rows x 8 rows and
// writes 8 x (4 k-values) so LDS keeps the same [row][k] image and the MFMA loop is unchanged.
#include "common.cuh"

namespace poet {

struct GemmK {
    PoetGemmDesc d;
    int kchunk;
    int a_vec, b_vec;
    uint32_t drop_thresh;
    float drop_scale;
};

template <typename CT> struct ct_traits;
template <> struct ct_traits<bf16_t> { static constexpr int E = 8; };
template <> struct ct_traits<float> { static constexpr int E = 4; };

constexpr int LDS_PITCH16 = 5;   // uint4 per LDS row (64 B payload + 16 B pad)

// ---- K-contiguous operand: src[row*ld + k] ----------------------------------------------------
template <typename Src, typename CT, int R>
struct LoaderKC {
    static constexpr int E = ct_traits<CT>::E;
    static constexpr int NI = (R * 4) / 256;
    static_assert(NI >= 1, "tile too small");
    float v[NI][E];

    __device__ __forceinline__ void load(const Src* __restrict__ src, int64_t ld, int r0, int rtot,
                                         int k0, int kend, int vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 2, kc = c & 3;
            const int gr = r0 + row, gk = k0 + kc * E;
            if (gr < rtot && gk + E <= kend && vec_ok) {
                vec<Src, E>::ld(src + (int64_t)gr * ld + gk, v[i]);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    v[i][e] = (gr < rtot && gk + e < kend) ? io<Src>::ld(src + (int64_t)gr * ld + gk + e) : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(uint4* __restrict__ lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 2, kc = c & 3;
            uint4 w;
            if constexpr (E == 8) {
                w = make_uint4(pack_bf2(v[i][0], v[i][1]), pack_bf2(v[i][2], v[i][3]),
                               pack_bf2(v[i][4], v[i][5]), pack_bf2(v[i][6], v[i][7]));
            } else {
                w = make_uint4(__float_as_uint(v[i][0]), __float_as_uint(v[i][1]),
                               __float_as_uint(v[i][2]), __float_as_uint(v[i][3]));
            }
            lds[row * LDS_PITCH16 + kc] = w;
        }
    }
};

// ---- K-major operand: src[k*ld + row] (register transpose while staging) -----------------------
template <typename Src, typename CT, int R, int TOFF>
struct LoaderKM {
    static constexpr int E = ct_traits<CT>::E;
    static constexpr int NKQ = E;                 // k-quads per stage (BK = 4E)
    static constexpr int ITEMS = NKQ * (R / 8);
    static_assert(ITEMS <= 256, "one item per thread");
    float v[4][8];

    __device__ __forceinline__ void load(const Src* __restrict__ src, int64_t ld, int r0, int rtot,
                                         int k0, int kend, int vec_ok, int tid) {
        const int item = tid - TOFF;
        if (item < 0 || item >= ITEMS) return;
        const int kq = item % NKQ, rg = item / NKQ;
        const int gr = r0 + rg * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gk = k0 + kq * 4 + i;
            if (gk < kend && gr + 8 <= rtot && vec_ok) {
                vec<Src, 8>::ld(src + (int64_t)gk * ld + gr, v[i]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[i][j] = (gk < kend && gr + j < rtot) ? io<Src>::ld(src + (int64_t)gk * ld + gr + j) : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(uint4* __restrict__ lds, int tid) const {
        const int item = tid - TOFF;
        if (item < 0 || item >= ITEMS) return;
        const int kq = item % NKQ, rg = item / NKQ;
        char* base = reinterpret_cast<char*>(lds);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = rg * 8 + j;
            if constexpr (E == 8) {
                *reinterpret_cast<uint2*>(base + row * (LDS_PITCH16 * 16) + kq * 8) =
                    make_uint2(pack_bf2(v[0][j], v[1][j]), pack_bf2(v[2][j], v[3][j]));
            } else {
                *reinterpret_cast<float4*>(base + row * (LDS_PITCH16 * 16) + kq * 16) =
                    make_float4(v[0][j], v[1][j], v[2][j], v[3][j]);
            }
        }
    }
};

template <typename Src, typename CT, int R, bool KM, int TOFF> struct LoaderSel;
template <typename Src, typename CT, int R, int TOFF> struct LoaderSel<Src, CT, R, false, TOFF> { using type = LoaderKC<Src, CT, R>; };
template <typename Src, typename CT, int R, int TOFF> struct LoaderSel<Src, CT, R, true, TOFF> { using type = LoaderKM<Src, CT, R, TOFF>; };

template <typename TA, typename TB, typename TC, typename CT, int BM, int BN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmK p) {
    constexpr int E = ct_traits<CT>::E;
    constexpr int BK = 4 * E;
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
    constexpr int BTOFF = (BKM && (E * (BN / 8) <= 128)) ? 128 : 0;
    __shared__ uint4 lds[(BM + BN) * LDS_PITCH16];
    uint4* As = lds;
    uint4* Bs = lds + BM * LDS_PITCH16;

    const PoetGemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int zb = blockIdx.z / d.splitk, sk = blockIdx.z % d.splitk;
    const int kbeg = sk * p.kchunk;
    const int kend = min(d.K, kbeg + p.kchunk);
    if (kbeg >= kend) return;

    const TA* A = reinterpret_cast<const TA*>(d.A) + (int64_t)zb * d.strideA;
    const TB* B = reinterpret_cast<const TB*>(d.B) + (int64_t)zb * d.strideB;

    typename LoaderSel<TA, CT, BM, AKM, 0>::type la;
    typename LoaderSel<TB, CT, BN, BKM, BTOFF>::type lb;

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    la.load(A, d.lda, m0, d.M, kbeg, kend, p.a_vec, tid);
    lb.load(B, d.ldb, n0, d.N, kbeg, kend, p.b_vec, tid);
    la.store(As, tid);
    lb.store(Bs, tid);
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) {
            la.load(A, d.lda, m0, d.M, k0 + BK, kend, p.a_vec, tid);
            lb.load(B, d.ldb, n0, d.N, k0 + BK, kend, p.b_vec, tid);
        }
        uint4 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = As[(wm * WM + i * 16 + frow) * LDS_PITCH16 + fchunk];
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = Bs[(wn * WN + j * 16 + frow) * LDS_PITCH16 + fchunk];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (E == 8) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].x), __uint_as_float(bfr[j].x), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].y), __uint_as_float(bfr[j].y), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].z), __uint_as_float(bfr[j].z), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].w), __uint_as_float(bfr[j].w), acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            la.store(As, tid);
            lb.store(Bs, tid);
            __syncthreads();
        }
    }

    // ---- epilogue -------------------------------------------------------------------------
    TC* C = reinterpret_cast<TC*>(d.C) + (int64_t)zb * d.strideC;
    const float* bias = d.bias ? d.bias + (int64_t)zb * d.stride_bias : nullptr;
    const TC* addp = reinterpret_cast<const TC*>(d.add_src);
    const TC* gate = reinterpret_cast<const TC*>(d.gate_ref);
    const bool use_atomic = (d.atomic != 0) || (d.splitk > 1);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = m0 + wm * WM + i * 16 + fchunk * 4 + t;
            if (row >= d.M) continue;
            const bool masked = d.row_mask && d.row_mask[row];
            int hm_n = 0, hm_s = 0;
            if (d.out_mode == 1) { hm_n = row / d.hm_S; hm_s = row - hm_n * d.hm_S; }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int col = n0 + wn * WN + j * 16 + frow;
                if (col >= d.N) continue;
                float val = acc[i][j][t] * d.alpha;
                int64_t off;
                if (d.out_mode == 1) {
                    const int hm_m = col / d.hm_D, hm_d = col - hm_m * d.hm_D;
                    off = (((int64_t)hm_n * d.hm_M + hm_m) * d.hm_S + hm_s) * d.hm_D + hm_d;
                } else {
                    off = (int64_t)row * d.ldc + col;
                }
                if (use_atomic) {
                    if constexpr (sizeof(TC) == 4) atomicAdd(reinterpret_cast<float*>(C) + off, val);
                    continue;
                }
                if (bias) val += bias[col];
                if (d.act == 1) val = fmaxf(val, 0.f);
                if (gate) val = (io<TC>::ld(gate + (int64_t)row * d.ldc + col) > 0.f) ? val * d.gate_scale : 0.f;
                if (p.drop_thresh) {
                    val = drop_keep(d.seed, (uint32_t)row * (uint32_t)d.N + (uint32_t)col, p.drop_thresh) ? val * p.drop_scale : 0.f;
                }
                if (addp) val += io<TC>::ld(addp + (int64_t)row * d.ld_add + col);
                if (masked) val = 0.f;
                io<TC>::st(C + off, val);
            }
        }
    }
}

template <typename TA, typename TB, typename TC, typename CT, int BM, int BN>
static int launch_layout(const GemmK& p, dim3 grid, hipStream_t st) {
    const int key = p.d.a_kmajor * 2 + p.d.b_kmajor;
    switch (key) {
        case 0: hipLaunchKernelGGL((gemm_kernel<TA, TB, TC, CT, BM, BN, false, false>), grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<TA, TB, TC, CT, BM, BN, false, true>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<TA, TB, TC, CT, BM, BN, true, false>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<TA, TB, TC, CT, BM, BN, true, true>), grid, dim3(256), 0, st, p); break;
    }
    return 0;
}

template <typename TA, typename TB, typename TC, typename CT>
static int launch_tile(const GemmK& p, hipStream_t st) {
    const PoetGemmDesc& d = p.d;
    const bool big = (d.M >= 512 && d.N >= 128);
    const int BM = big ? 128 : 64, BN = big ? 128 : 64;
    dim3 grid(cdiv(d.N, BN), cdiv(d.M, BM), d.batch * d.splitk);
    if (big) return launch_layout<TA, TB, TC, CT, 128, 128>(p, grid, st);
    return launch_layout<TA, TB, TC, CT, 64, 64>(p, grid, st);
}

static bool vec_ok(const void* ptr, int64_t ld, int64_t stride) {
    return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && (ld % 8 == 0) && (stride % 8 == 0);
}

}  // namespace poet

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
    if (atomic) {
        POET_CHECK(d.c_dtype == POET_F32, POET_ERR_ARG, "poet_gemm: atomic/split-K needs fp32 C");
        POET_CHECK(!d.bias && !d.act && !d.gate_ref && !d.add_src && d.drop_p == 0.f && !d.row_mask, POET_ERR_ARG,
                   "poet_gemm: atomic/split-K allows no epilogue");
    }
    POET_CHECK(d.drop_p >= 0.f && d.drop_p < 1.f, POET_ERR_ARG, "poet_gemm: drop_p");
    if (d.out_mode == 1) POET_CHECK(d.hm_M > 0 && d.hm_S > 0 && d.hm_D > 0 && d.hm_M * d.hm_D == d.N, POET_ERR_ARG, "poet_gemm: head-major dims");
    const int E = d.compute == POET_BF16 ? 8 : 4;
    const int BK = 4 * E;
    p.kchunk = cdiv(cdiv(d.K, d.splitk), BK) * BK;
    p.a_vec = vec_ok(d.A, d.lda, d.strideA);
    p.b_vec = vec_ok(d.B, d.ldb, d.strideB);
    p.drop_thresh = d.drop_p > 0.f ? drop_thresh(d.drop_p) : 0u;
    p.drop_scale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
    if (d.gate_scale == 0.f) d.gate_scale = 1.f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

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
