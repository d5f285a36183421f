// Weight-stationary streaming GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * W^T),  M ~ 1e5 token rows, K = 256,
// W a (slice of a) 256-column weight.  Every nn.Linear of the encoder with a 256-wide input and its input-gradient
// twin (reference call sites: models/deformable_transformer.py:215-240 encoder layer, MSDeformAttn's four
// projections) has this shape at bs*S = 102080 rows.  They are HBM-bound (K = 256 gives ~128 flop/B against a machine
// balance of ~400), so the kernel is organised around the byte stream, not the MFMA:
//
//  * persistent workgroups; a workgroup keeps ONE BN x K slice of W in LDS for its whole life (loaded once, 64 KB),
//    so the only recurring traffic is A in and C out;
//  * the activation operand never touches LDS: a wave owns 32 consecutive rows and loads its MFMA B-operand fragments
//    straight from global memory (lane = row l&15, 16-byte k-chunk l>>4), 16 x 16 B per lane for the whole K = 256.
//    The registers of chunk kk are refilled with the NEXT 32 rows' chunk kk as soon as chunk kk has been consumed
//    (rolling prefetch), so every wave always has 16 KB in flight with no barrier anywhere in the loop;
//  * the product is computed transposed, D = W_frag (16 n x 32 k) * A_frag^T (32 k x 16 m): a lane then holds 4
//    consecutive n for ONE row m.  The rows of W are laid out in LDS in a permuted order so that across the 4 fragments
//    of a 64-column group a lane owns two runs of 8 consecutive output columns of its row (32 apart): the whole epilogue (bias, ReLU, ReLU
//    gate, dropout, residual add, row mask, head-major scatter) runs on registers with 16/32-byte vector accesses and
//    every vector access of a 16-lane row group covers 64 B (bf16) / 128 B (f32) contiguous bytes per row -- no LDS staging, no barrier;
//  * 16-row units are dealt to waves as contiguous balanced ranges (6 or 7 units per wave at 102080 rows), the
//    N/BN column slices of the same rows run on the same XCD (workgroup b -> XCD b % 8) so A is fetched from HBM once
//    and re-read through that XCD's L2.
#include "gemm.cuh"

#include <type_traits>

namespace poet {

namespace {

__device__ __forceinline__ int ws_perm(int rho) {          // LDS row (fragment jn, MFMA row i) -> column inside the slice
    // fragment jn = 4*jq + jf, MFMA row i = 4*g + t  ->  column 64*jq + 32*(jf>>1) + 8*g + 4*(jf&1) + t: lane group g
    // owns columns [8g, 8g+8) and [32+8g, 32+8g+8) of every 64-column group
    const int jn = rho >> 4, i = rho & 15;
    return (jn >> 2) * 64 + ((jn >> 1) & 1) * 32 + (i >> 2) * 8 + (jn & 1) * 4 + (i & 3);
}
__device__ __forceinline__ int ws_inv_perm(int n) {
    const int jq = n >> 6, h = (n >> 5) & 1, g = (n >> 3) & 3, lo = (n >> 2) & 1, t = n & 3;
    return ((jq * 4 + h * 2 + lo) << 4) + g * 4 + t;
}

enum : int { WS_GATE = 1, WS_ADD = 2, WS_MASK = 4 };

// LDS image of a weight slice: rows of ROWB bytes, NO padding, the 16-byte chunks of a row XOR-swizzled by the row's low 4 bits.
// A fragment read (ds_read_b128: lane = MFMA row frow = lane & 15, k-chunk g = lane >> 4) is serviced in the hardware's four
// 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}
// (MI355X_MICROARCH.md, LDS table), i.e. rows {0-3, 12-15} of chunk g and rows {4-11} of chunk g + 1: with the swizzle the 16
// lanes of every group land on 16 different 16-byte bank quads.  (The padded pitch this replaces, ROWB + 16, put row 11 / chunk
// g + 1 and row 12 / chunk g on the same quad: every group took two LDS cycles, SQ_LDS_BANK_CONFLICT = 48 % of SQ_LDS_IDX_ACTIVE.)
template <int ROWB>
__device__ __forceinline__ int ws_sw(int rho, int byte) { return rho * ROWB + ((((byte >> 4) ^ (rho & 15)) << 4) | (byte & 15)); }


template <typename TC> struct run8;                        // 8 consecutive C-typed values <-> raw 16-byte registers
template <> struct run8<bf16_t> {
    static constexpr int NV = 1;
    static __device__ __forceinline__ void ld(const bf16_t* p, uint4* r) { r[0] = *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ void dec(const uint4* r, float* o) {
        o[0] = __uint_as_float(r[0].x << 16); o[1] = __uint_as_float(r[0].x & 0xffff0000u);
        o[2] = __uint_as_float(r[0].y << 16); o[3] = __uint_as_float(r[0].y & 0xffff0000u);
        o[4] = __uint_as_float(r[0].z << 16); o[5] = __uint_as_float(r[0].z & 0xffff0000u);
        o[6] = __uint_as_float(r[0].w << 16); o[7] = __uint_as_float(r[0].w & 0xffff0000u);
    }
};
template <> struct run8<f16_t> {                           // (fp16 outputs have no add_src / gate_ref: ld / dec exist for the type's sake)
    static constexpr int NV = 1;
    static __device__ __forceinline__ void ld(const f16_t* p, uint4* r) { r[0] = *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ void dec(const uint4* r, float* o) {
        o[0] = h_lo(r[0].x); o[1] = h_hi(r[0].x); o[2] = h_lo(r[0].y); o[3] = h_hi(r[0].y);
        o[4] = h_lo(r[0].z); o[5] = h_hi(r[0].z); o[6] = h_lo(r[0].w); o[7] = h_hi(r[0].w);
    }
};
template <> struct run8<float> {
    static constexpr int NV = 2;
    static __device__ __forceinline__ void ld(const float* p, uint4* r) {
        r[0] = *reinterpret_cast<const uint4*>(p);
        r[1] = *reinterpret_cast<const uint4*>(p + 4);
    }
    static __device__ __forceinline__ void dec(const uint4* r, float* o) {
        o[0] = __uint_as_float(r[0].x); o[1] = __uint_as_float(r[0].y); o[2] = __uint_as_float(r[0].z); o[3] = __uint_as_float(r[0].w);
        o[4] = __uint_as_float(r[1].x); o[5] = __uint_as_float(r[1].y); o[6] = __uint_as_float(r[1].z); o[7] = __uint_as_float(r[1].w);
    }
};

// SPLIT (PoetGemmDesc.b_split): W arrives as the fp32 master and lives in LDS as TWO bf16 images, hi = bf16(W) and
// lo = bf16(W - hi); every activation fragment meets both (two MFMAs), so the weight carries 16 mantissa bits at no extra HBM
// traffic -- the matrix pipe of these HBM-bound kernels is ~25 % busy without it.
// WM = 2 (PoetGemmDesc.b_split = 2, round 6, experimental): ONE image of the fp32 master rounded to IEEE fp16 (2^-12: the emulated policy
// loses 1 % against the split, tests/tools/prec_ablate.py W16TEST) and ONE v_mfma_f32_16x16x32_f16 per fragment pair; the bf16 activation
// fragment converts to fp16 in registers -- exactly, 8 -> 11 mantissa bits (v_cvt_pkrtz: magnitudes past 65504 saturate) -- 12 VALU
// per 16-byte fragment against the 8 matrix instructions it then feeds.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2r_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8_t bf8_to_f16x8(const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)));
    return __builtin_bit_cast(f16x8_t, make_uint4(o[0], o[1], o[2], o[3]));
}

template <typename TC, int KIND, bool WKM, int FM, int NTH, int WM = 0, int BN = 128>
__global__ __launch_bounds__(NTH, (NTH >= 1024 ? 4 : 2)) void gemm_ws_kernel(const GemmK p) {
    constexpr bool SPLIT = WM == 1, F16W = WM == 2;
    static_assert(!F16W || !WKM, "fp16 weights are a forward ([N,K] weight) feature");
    constexpr int KS = 8, NWV = NTH / 64;
    constexpr int K = KS * 32, PITCH = K * 2, FNT = BN / 16, NQ = BN / 64, NV = run8<TC>::NV;
    constexpr int LO = SPLIT ? BN * PITCH : 0;                           // byte offset of the lo image
    constexpr bool GATE = KIND & WS_GATE, ADD = KIND & WS_ADD, MASK = KIND & WS_MASK;
    static_assert(!SPLIT || !WKM, "b_split is a forward ([N,K] weight) feature");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [BN][PITCH] bf16 rows of W (permuted, chunks swizzled: ws_sw) (| lo image) | bias[BN] f32
    float* sbias = reinterpret_cast<float*>(smem + (SPLIT ? 2 : 1) * BN * PITCH);
    const PoetGemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int frow = lane & 15, g = lane >> 4;

    // ---- work assignment ----
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int NT = d.N / BN, groups = per / NT;
    if (j >= groups * NT) return;
    const int nt = j % NT, G = (j / NT) * 8 + xcd, NW = groups * 8 * NWV;
    const int n0 = nt * BN;
    const int U = (d.M + 15) >> 4, wv = G * NWV + wid;
    const int u_lo = (int)((int64_t)wv * U / NW), u_hi = (int)((int64_t)(wv + 1) * U / NW);

    // ---- A prefetch for the first 32 rows goes out before W is touched ----
    const bf16_t* A = reinterpret_cast<const bf16_t*>(d.A);
    uint4 a[FM][KS];
    auto arow = [&](int u, int fm) {                                   // row of fragment fm of the unit pair starting at u, clamped
        return max(min(min(u + fm, u_hi - 1) * 16 + frow, d.M - 1), 0);
    };
    if (u_lo < u_hi) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const bf16_t* p0 = A + (int64_t)arow(u_lo, fm) * d.lda + g * 8;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a[fm][kk] = *reinterpret_cast<const uint4*>(p0 + kk * 32);
        }
    }

    // ---- stationary W slice -> LDS (once) ----
    const bf16_t* B = reinterpret_cast<const bf16_t*>(d.B);
    // (all of a thread's requests are in flight before its first LDS store: one exposed latency, not one per chunk)
    if constexpr (SPLIT) {                                              // fp32 W[n][k] -> hi | lo images
        const float* Bf = reinterpret_cast<const float*>(d.B);
        constexpr int CPR = K / 4, NCH = BN * CPR, GRP = 8;            // 16-B chunks of 4 floats, 8 in flight per thread
        static_assert(NCH % (NTH * GRP) == 0, "split W staging");
#pragma unroll 1
        for (int base = 0; base < NCH; base += NTH * GRP) {
            float4 wv[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int idx = base + tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                wv[i] = *reinterpret_cast<const float4*>(Bf + (int64_t)(n0 + ws_perm(rho)) * d.ldb + kc * 4);
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int idx = base + tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                const uint2 hi = make_uint2(pack_bf2(wv[i].x, wv[i].y), pack_bf2(wv[i].z, wv[i].w));
                const uint2 lo = make_uint2(pack_bf2(wv[i].x - __uint_as_float(hi.x << 16), wv[i].y - __uint_as_float(hi.x & 0xffff0000u)),
                                            pack_bf2(wv[i].z - __uint_as_float(hi.y << 16), wv[i].w - __uint_as_float(hi.y & 0xffff0000u)));
                *reinterpret_cast<uint2*>(smem + ws_sw<PITCH>(rho, kc * 8)) = hi;
                *reinterpret_cast<uint2*>(smem + LO + ws_sw<PITCH>(rho, kc * 8)) = lo;
            }
        }
    } else if constexpr (F16W) {                                        // fp32 W[n][k] -> one IEEE fp16 image
        const float* Bf = reinterpret_cast<const float*>(d.B);
        constexpr int CPR = K / 4, NCH = BN * CPR, GRP = 8;
        static_assert(NCH % (NTH * GRP) == 0, "fp16 W staging");
#pragma unroll 1
        for (int base = 0; base < NCH; base += NTH * GRP) {
            float4 wv[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int idx = base + tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                wv[i] = *reinterpret_cast<const float4*>(Bf + (int64_t)(n0 + ws_perm(rho)) * d.ldb + kc * 4);
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int idx = base + tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                *reinterpret_cast<uint2*>(smem + ws_sw<PITCH>(rho, kc * 8)) = make_uint2(pack_h2(wv[i].x, wv[i].y), pack_h2(wv[i].z, wv[i].w));
            }
        }
    } else if constexpr (!WKM) {                                        // W[n][k], k contiguous
        constexpr int CPR = K / 8, NIT = BN * CPR / NTH;
        uint4 wv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
            wv[i] = *reinterpret_cast<const uint4*>(B + (int64_t)(n0 + ws_perm(rho)) * d.ldb + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
            *reinterpret_cast<uint4*>(smem + ws_sw<PITCH>(rho, kc * 16)) = wv[i];
        }
    } else {                                                            // W[k][n] (input-gradient GEMMs): transpose on the way in
        constexpr int NG = BN / 8, NIT = (K / 4) * NG / NTH;            // item: 4 consecutive k x 8 consecutive n
        uint4 wv[NIT][4];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * NTH, kq = idx / NG, nl = (idx - kq * NG) * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[i][r] = *reinterpret_cast<const uint4*>(B + (int64_t)(kq * 4 + r) * d.ldb + n0 + nl);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * NTH, kq = idx / NG, nl = (idx - kq * NG) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                uint32_t h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t w = (e >> 1) == 0 ? wv[i][r].x : (e >> 1) == 1 ? wv[i][r].y : (e >> 1) == 2 ? wv[i][r].z : wv[i][r].w;
                    h[r] = (e & 1) ? (w >> 16) : (w & 0xffffu);
                }
                *reinterpret_cast<uint2*>(smem + ws_sw<PITCH>(ws_inv_perm(nl + e), kq * 8)) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            }
        }
    }
    if (tid < BN) sbias[tid] = d.bias ? d.bias[n0 + tid] : 0.f;
    __syncthreads();

    const TC* addp = reinterpret_cast<const TC*>(d.add_src);
    const TC* gate = reinterpret_cast<const TC*>(d.gate_ref);
    TC* C = reinterpret_cast<TC*>(d.C);
    const uint32_t sd = d.seed ^ (d.seed_dev ? *d.seed_dev * 0x9E3779B1u : 0u);
    int wsw[4];                                                         // fragment read offsets of this lane: k-step kk uses wsw[kk & 3]
#pragma unroll
    for (int j = 0; j < 4; ++j) wsw[j] = ws_sw<PITCH>(frow, (j * 4 + g) * 16);

    // One iteration = 32 rows (two 16-row fragments) x BN columns.  TAIL = false: both fragments wholly valid, no
    // predication, every VMEM op of the body is unconditional, so the compiler's s_waitcnt counts are exact and nothing
    // ever drains the queue: operands of the epilogue (gate / residual / row mask) are requested FIRST, the A refills
    // follow inside the k loop, the stores come last.  TAIL = true (at most once per wave, after the loop): rows past the
    // wave's range are clamped on load and skipped on store.
    auto iteration = [&](int u, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        int grow[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) grow[fm] = TAIL ? arow(u, fm) : (u + fm) * 16 + frow;
        uint4 gq[GATE ? FM : 1][NQ][2][NV], rq[ADD ? FM : 1][NQ][2][NV];
        uint32_t mk[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            mk[fm] = 0u;
            if constexpr (MASK) mk[fm] = d.row_mask[grow[fm]];
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int gcol = n0 + jq * 64 + h * 32 + g * 8;
                    if constexpr (GATE) run8<TC>::ld(gate + (int64_t)grow[fm] * d.ldc + gcol, gq[fm][jq][h]);
                    if constexpr (ADD) run8<TC>::ld(addp + (int64_t)grow[fm] * d.ld_add + gcol, rq[fm][jq][h]);
                }
        }
        f32x4_t acc[FM][FNT];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FNT; ++jn) acc[i][jn] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(wsw[0]), "+v"(wsw[1]), "+v"(wsw[2]), "+v"(wsw[3]));      // W fragments are loop-invariant: keep them in LDS, not hoisted into 256 VGPRs
        const char* wl[4] = {smem + wsw[0], smem + wsw[1], smem + wsw[2], smem + wsw[3]};
        const int un = TAIL ? u : u + FM;       // the refill of the last iteration re-reads its own rows (cache hit, never used)
        const bf16_t* qn[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) qn[fm] = A + (int64_t)arow(un, fm) * d.lda + g * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if constexpr (F16W) {
                f16x8_t ah[FM];
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) ah[fm] = bf8_to_f16x8(a[fm][kk]);
#pragma unroll
                for (int jn = 0; jn < FNT; ++jn) {
                    const f16x8_t wf = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(wl[kk & 3] + jn * 16 * PITCH + (kk >> 2) * 256));
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) acc[fm][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, ah[fm], acc[fm][jn], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int jn = 0; jn < FNT; ++jn) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wl[kk & 3] + jn * 16 * PITCH + (kk >> 2) * 256));
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    acc[fm][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, a[fm][kk]), acc[fm][jn], 0, 0, 0);
                if constexpr (SPLIT) {
                    const bf16x8_t wlo = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wl[kk & 3] + LO + jn * 16 * PITCH + (kk >> 2) * 256));
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm)
                        acc[fm][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, __builtin_bit_cast(bf16x8_t, a[fm][kk]), acc[fm][jn], 0, 0, 0);
                }
            }
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) a[fm][kk] = *reinterpret_cast<const uint4*>(qn[fm] + kk * 32);
            __builtin_amdgcn_sched_barrier(0);                          // keep the refill HERE (the scheduler sinks it otherwise)
        }

        // ---- epilogue on registers: lane = row frow of fragment fm, columns [8g, 8g+8) and [32+8g, 32+8g+8) of each 64 ----
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const bool live = !TAIL || ((u + fm) < u_hi && (u + fm) * 16 + frow < d.M);
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                for (int h = 0; h < 2; ++h) {                           // one run of 8 consecutive columns at a time
                    const int lcol = jq * 64 + h * 32 + g * 8;          // column inside the slice
                    float v[8];
#pragma unroll
                    for (int lo = 0; lo < 2; ++lo)
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[lo * 4 + t] = acc[fm][jq * 4 + h * 2 + lo][t];
                    if (d.alpha != 1.f) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= d.alpha;
                    }
                    {
                        float b[8];
                        vec<float, 8>::ld(sbias + lcol, b);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += b[e];
                    }
                    if (d.act == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu1(v[e]);
                    }
                    if constexpr (GATE) {
                        float gt[8];
                        run8<TC>::dec(gq[fm][jq][h], gt);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = gt[e] > 0.f ? v[e] * d.gate_scale : 0.f;
                    }
                    if (p.drop_thresh) {
                        const uint32_t base = (uint32_t)grow[fm] * (uint32_t)d.N + (uint32_t)(n0 + lcol);
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {              // base is even: (e, e+1) share one hash
                            const uint32_t hsh = drop_pair(sd, (base + e) >> 1);
                            v[e] = (hsh & 0xffffu) >= p.drop_thresh ? v[e] * p.drop_scale : 0.f;
                            v[e + 1] = (hsh >> 16) >= p.drop_thresh ? v[e + 1] * p.drop_scale : 0.f;
                        }
                    }
                    if constexpr (ADD) {
                        float r[8];
                        run8<TC>::dec(rq[fm][jq][h], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if constexpr (MASK) {
                        if (mk[fm] & 0xffu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = 0.f;
                        }
                    }
                    if (live) {
                        const int gcol = n0 + lcol;
                        if (d.out_mode == 1) {
                            const int hn = grow[fm] / d.hm_S, hs = grow[fm] - hn * d.hm_S, hm = gcol / d.hm_D, hd = gcol - hm * d.hm_D;
                            vec<TC, 8>::st(C + (((int64_t)hn * d.hm_M + hm) * d.hm_S + hs) * d.hm_D + hd, v);
                        } else {
                            vec<TC, 8>::st(C + (int64_t)grow[fm] * d.ldc + gcol, v);
                        }
                    }
                }
        }
    };

    int u = u_lo;
    const int u_full = min(u_hi, d.M >> 4);                             // units below u_full are wholly inside [0, M)
    for (; u + FM <= u_full; u += FM) iteration(u, std::false_type{});
    if (u < u_hi) iteration(u, std::true_type{});
}

// ---- software-pipelined form of the wide split-weight forward products (FFN1: ReLU + dropout, bf16; offsets | logits: fp16) ----
// gemm_ws_kernel above runs a 32-row x 128-column unit as two serial phases -- 256 matrix instructions, then ~17 vector instructions
// per output for the epilogue -- and with two waves per SIMD that start in step the phases of the partner coincide: the matrix pipe
// idles while both run their epilogues (measured: MFMA, epilogue and store time ADD).  Here the unit is cut
// into its two 64-column halves and the wave's own stream is interleaved BY HAND: the 128 matrix instructions of one half are issued
// in 32 fenced chunks of four, and every second chunk carries the epilogue of one output pair of the half finished before, inside
// the same 64 accumulator registers (a `sched_group_barrier` pipeline of the same shape was not honoured by the scheduler and took
// minutes to compile).  The epilogue is on a diet for the same reason: the dropout's 1 / (1 - p) is folded into the weight images
// and the bias while they are staged (ReLU commutes with a positive scale: no multiply), ReLU is one v_max_i32 on the bit pattern,
// the high half of a hash word is compared in place (h >= T << 16: no shift), the bias of a lane's 16 columns is read once per
// phase.  Without dropout the outputs are BIT-IDENTICAL to gemm_ws_kernel's (same accumulation order, bias last), so the goldens'
// realisation of the bf16 policy does not move; with dropout they differ by the fp32 rounding of the scaled weight (before its
// hi / lo split), and the dropout MASK is identical.  WSP_AHEAD: chunks of fragment prefetch (1, 2 and 4 measured equal).
#ifndef WSP_AHEAD
#define WSP_AHEAD 2
#endif
template <typename TC, bool ACT, bool DROP>
__global__ __launch_bounds__(512, 2) void gemm_wsp_kernel(const GemmK p) {
    constexpr int KS = 8, FM = 2, BN = 128, NTH = 512, NWV = NTH / 64;
    constexpr int K = KS * 32, PITCH = K * 2, LO = BN * PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // hi image | lo image | bias[BN] f32 (x the dropout scale)
    float* sbias = reinterpret_cast<float*>(smem + 2 * BN * PITCH);
    const PoetGemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, g = lane >> 4;

    // ---- work assignment (as gemm_ws_kernel) ----
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int NT = d.N / BN, groups = per / NT;
    if (j >= groups * NT) return;
    const int nt = j % NT, G = (j / NT) * 8 + xcd, NW = groups * 8 * NWV;
    const int n0 = nt * BN;
    const int U = (d.M + 15) >> 4, wv = G * NWV + wid;
    const int u_lo = (int)((int64_t)wv * U / NW), u_hi = (int)((int64_t)(wv + 1) * U / NW);

    const bf16_t* A = reinterpret_cast<const bf16_t*>(d.A);
    uint4 a[FM][KS];
    auto arow = [&](int u, int fm) { return max(min(min(u + fm, u_hi - 1) * 16 + frow, d.M - 1), 0); };
    if (u_lo < u_hi) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const bf16_t* p0 = A + (int64_t)arow(u_lo, fm) * d.lda + g * 8;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a[fm][kk] = *reinterpret_cast<const uint4*>(p0 + kk * 32);
        }
    }

    // ---- stationary weight slice: fp32 W x (dropout scale) -> hi | lo images ----
    const float wscale = DROP ? p.drop_scale : 1.f;
    {
        const float* Bf = reinterpret_cast<const float*>(d.B);
        constexpr int CPR = K / 4, NCH = BN * CPR, GRP = 8;
        static_assert(NCH % (NTH * GRP) == 0, "split W staging");
#pragma unroll 1
        for (int base = 0; base < NCH; base += NTH * GRP) {
            float4 wv4[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int idx = base + tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                wv4[i] = *reinterpret_cast<const float4*>(Bf + (int64_t)(n0 + ws_perm(rho)) * d.ldb + kc * 4);
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int idx = base + tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                const float w0 = wv4[i].x * wscale, w1 = wv4[i].y * wscale, w2 = wv4[i].z * wscale, w3 = wv4[i].w * wscale;
                const uint2 hi = make_uint2(pack_bf2(w0, w1), pack_bf2(w2, w3));
                const uint2 lo = make_uint2(pack_bf2(w0 - __uint_as_float(hi.x << 16), w1 - __uint_as_float(hi.x & 0xffff0000u)),
                                            pack_bf2(w2 - __uint_as_float(hi.y << 16), w3 - __uint_as_float(hi.y & 0xffff0000u)));
                *reinterpret_cast<uint2*>(smem + ws_sw<PITCH>(rho, kc * 8)) = hi;
                *reinterpret_cast<uint2*>(smem + LO + ws_sw<PITCH>(rho, kc * 8)) = lo;
            }
        }
    }
    if (tid < BN) sbias[tid] = (d.bias ? d.bias[n0 + tid] : 0.f) * wscale;
    __syncthreads();

    TC* C = reinterpret_cast<TC*>(d.C);
    const uint32_t sd = (d.seed ^ (d.seed_dev ? *d.seed_dev * 0x9E3779B1u : 0u)) * 0x9E3779B9u;      // drop_pair(seed, i) = hash32(i ^ this)
    const uint32_t thr_lo = p.drop_thresh, thr_hi = p.drop_thresh << 16;
    int wsw[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) wsw[jj] = ws_sw<PITCH>(frow, (jj * 4 + g) * 16);
    const float* bl = sbias + 8 * g;                                     // this lane's 8-column runs start at 64 H + 32 r + 8 g

    f32x4_t acc[FM][8];
    // One PHASE = the 128 matrix instructions of half H (fragments 4H .. 4H + 3) of the unit pair whose rows are in a[], cut into 32
    // chunks of four (one weight fragment pair x two row fragments), each fenced from the next; EPI: every second chunk also carries
    // the epilogue of ONE output pair of the OTHER half (finished by the phase before) of the unit pair starting at ue -- 16 pairs
    // per lane and half, a 16-byte store after every fourth.  REFILL: a[][kk] is re-requested for the rows qn[] once step kk has used it.
    // The weight fragments of chunk c + 1 are read before the matrix instructions of chunk c are issued.
    auto phase = [&](auto htag, auto rtag, auto etag, auto ttag, const bf16_t* const* qn, int ue) __attribute__((always_inline)) {
        constexpr int H = decltype(htag)::value, HE = 1 - H;
        constexpr bool REFILL = decltype(rtag)::value, EPI = decltype(etag)::value, TAIL = decltype(ttag)::value;
        asm volatile("" : "+v"(wsw[0]), "+v"(wsw[1]), "+v"(wsw[2]), "+v"(wsw[3]));      // fragments stay in LDS (not hoisted into registers)
        const char* wl[4] = {smem + wsw[0], smem + wsw[1], smem + wsw[2], smem + wsw[3]};
        f32x4_t acn[FM][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acn[0][jj] = acn[1][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        f32x4_t bb[2][2];                                                // the bias of this lane's 16 columns of half HE, read once per phase
        if constexpr (EPI) {                                             // (read at its use it would drain the fragment prefetch: lgkmcnt counts in order)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) bb[r][q4] = *reinterpret_cast<const f32x4_t*>(bl + HE * 64 + r * 32 + q4 * 4);
        }
        TC* crow[FM];
        uint32_t hb[FM];
        bool live[FM];
        if constexpr (EPI) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int grow = TAIL ? arow(ue, fm) : (ue + fm) * 16 + frow;
                live[fm] = !TAIL || ((ue + fm) < u_hi && (ue + fm) * 16 + frow < d.M);
                crow[fm] = C + (int64_t)grow * d.ldc + n0 + HE * 64 + g * 8;
                hb[fm] = (((uint32_t)grow * (uint32_t)d.N + (uint32_t)(n0 + HE * 64 + g * 8)) >> 1) ^ sd;     // (bits 0, 1 and 4 of the pair index are 0 here: + (r * 16 + pp) == ^)
            }
        }
        uint32_t pk[4];
        // chunk c: k-step kk = c >> 2, image (c >> 1) & 1 (hi, hi, lo, lo), fragments jj = 2 (c & 1), + 1: the hi and the lo product of
        // one accumulator are eight matrix instructions apart.  The two fragments of chunk c + WSP_AHEAD are read while chunk c is issued.
        auto frag = [&](int c, int i) {
            const int kk = c >> 2, img = (c >> 1) & 1, jn = H * 4 + 2 * (c & 1) + i;
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wl[kk & 3] + img * LO + jn * 16 * PITCH + (kk >> 2) * 256));
        };
        constexpr int AH = WSP_AHEAD;
        bf16x8_t wq[AH][2];
#pragma unroll
        for (int c = 0; c < AH; ++c) { wq[c][0] = frag(c, 0); wq[c][1] = frag(c, 1); }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int kk = c >> 2, j0 = 2 * (c & 1);
            const bf16x8_t w0 = wq[c % AH][0], w1 = wq[c % AH][1];
            if (c + AH < 32) { wq[c % AH][0] = frag(c + AH, 0); wq[c % AH][1] = frag(c + AH, 1); }
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acn[fm][j0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, __builtin_bit_cast(bf16x8_t, a[fm][kk]), acn[fm][j0], 0, 0, 0);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acn[fm][j0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, __builtin_bit_cast(bf16x8_t, a[fm][kk]), acn[fm][j0 + 1], 0, 0, 0);
            if constexpr (EPI) {
                if (c & 1) {
                    const int q = c >> 1, fm = q >> 3, r = (q >> 2) & 1, pp = q & 3;
                    const f32x4_t av = acc[fm][HE * 4 + r * 2 + (pp >> 1)];
                    float v0 = av[(pp & 1) * 2] + bb[r][pp >> 1][(pp & 1) * 2], v1 = av[(pp & 1) * 2 + 1] + bb[r][pp >> 1][(pp & 1) * 2 + 1];      // (bias LAST, as gemm_ws_kernel: bit-identical without dropout)
                    if constexpr (ACT) {                                 // max(x, 0) on the bit pattern: one v_max_i32, nothing to canonicalise
                        v0 = __int_as_float(max(__float_as_int(v0), 0));
                        v1 = __int_as_float(max(__float_as_int(v1), 0));
                    }
                    if constexpr (DROP) {
                        const uint32_t hsh = hash32(hb[fm] ^ (uint32_t)(r * 16 + pp));
                        v0 = (hsh & 0xffffu) >= thr_lo ? v0 : 0.f;
                        v1 = hsh >= thr_hi ? v1 : 0.f;
                    }
                    pk[pp] = std::is_same<TC, f16_t>::value ? pack_h2(v0, v1) : pack_bf2(v0, v1);
                    if (pp == 3 && live[fm]) *reinterpret_cast<uint4*>(crow[fm] + r * 32) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
            }
            if constexpr (REFILL) {
                if ((c & 3) == 3) {
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) a[fm][kk] = *reinterpret_cast<const uint4*>(qn[fm] + kk * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[fm][H * 4 + jj] = acn[fm][jj];
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N_ = std::false_type;

    int u = u_lo;
    const int u_full = min(u_hi, d.M >> 4);
    const bf16_t* qn[FM] = {A, A};
    if (u < u_hi) {
        phase(H0{}, N_{}, N_{}, N_{}, qn, u);                             // prologue: half 0 of the first pair (rows clamped on load)
        while (u + FM <= u_full) {                                        // the pair at u is whole
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) qn[fm] = A + (int64_t)arow(u + FM, fm) * d.lda + g * 8;
            phase(H1{}, Y{}, Y{}, N_{}, qn, u);                           // half 1 of this pair (+ refills)  ||  epilogue of its half 0
            phase(H0{}, N_{}, Y{}, N_{}, qn, u);                          // half 0 of the next pair (whole, ragged or none)  ||  epilogue of half 1
            u += FM;
        }
        if (u < u_hi) {                                                   // ragged end of the wave's range: masked stores
            phase(H1{}, N_{}, Y{}, Y{}, qn, u);
            phase(H0{}, N_{}, Y{}, Y{}, qn, u);
        }
    }
}

#ifdef POET_PROBE_KERNELS
#include "gemm_wss.inc"      // store waves + LDS ring for the same products: measured slower (profiles/probes/kernels/)
#endif

// ---- K-chunked variant: K = KC x 128 (FFN2 and the K = 512 / 768 / 1024 input-gradient products) ----------------------
// The weight no longer fits LDS whole, so the loop nest is turned inside out: a workgroup owns ONE (256-row block, 128-column
// slice) item, its 8 waves keep the 32 x 128 accumulators of their rows in registers for the whole K loop, and the W chunk
// [128 columns][128 k] (hi | lo images when SPLIT) is DOUBLE-BUFFERED in LDS: the global loads of chunk c+1 (W through
// registers, fp32 -> hi/lo conversion included; the activation rows straight into MFMA operand registers) are issued before
// the MFMAs of chunk c and land under them, its LDS stores follow, ONE barrier per chunk.  Weight traffic per activation row is
// 1/2 of the 128 x 128 tiled kernel's, and the epilogue is the register epilogue of the kernel above.
template <typename TC, int KIND, bool WKM, bool SPLIT, int BN = 128, int NTH = 512>
__global__ __launch_bounds__(NTH, 2) void gemm_wsk_kernel(const GemmK p) {
    constexpr int KS = 4, FM = 2, RBLK = NTH / 2;                        // rows per workgroup: 32 per wave
    constexpr int K = KS * 32, PITCH = K * 2, IMG = BN * PITCH, FNT = BN / 16, NQ = BN / 64, NV = run8<TC>::NV;
    constexpr int BUF = (SPLIT ? 2 : 1) * IMG;                           // one W chunk (hi | lo)
    constexpr bool GATE = KIND & WS_GATE, ADD = KIND & WS_ADD, MASK = KIND & WS_MASK;
    static_assert(!SPLIT || !WKM, "b_split is a forward ([N,K] weight) feature");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x BUF | bias[BN] f32
    float* sbias = reinterpret_cast<float*>(smem + 2 * BUF);
    const PoetGemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int frow = lane & 15, g = lane >> 4;

    // item = (row block, column slice); the slices of one row block run on one XCD (A is re-read through that L2)
    const int NT = d.N / BN, RB = (d.M + RBLK - 1) / RBLK;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nt = j % NT, rb = (j / NT) * 8 + xcd;
    if (rb >= RB) return;
    const int n0 = nt * BN, KC = d.K >> 7;
    const int row0 = rb * RBLK + wid * 32;                               // this wave's 32 rows
    const bf16_t* A = reinterpret_cast<const bf16_t*>(d.A);
    int grow[FM];
    const bf16_t* ap[FM];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        grow[fm] = row0 + fm * 16 + frow;
        ap[fm] = A + (int64_t)min(grow[fm], d.M - 1) * d.lda + g * 8;
    }

    // W chunk staging, split in two halves around the MFMAs: loads (global -> registers), then stores (registers -> LDS)
    static_assert(BN == 128 || (BN == 64 && !WKM), "slice widths");
    static_assert(NTH == 512 || !WKM, "the transposing staging assumes 512 threads");
    constexpr int NWV = SPLIT ? BN * 32 / NTH : (WKM ? 4 : BN * 16 / NTH);      // 16-B registers per thread and chunk
    auto w_load = [&](uint4 (&wv)[NWV], int c) __attribute__((always_inline)) {
        if constexpr (SPLIT) {
            const float* Bf = reinterpret_cast<const float*>(d.B) + c * K;
            constexpr int CPR = K / 4;                                   // float4 chunks per row
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int idx = tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                wv[i] = *reinterpret_cast<const uint4*>(Bf + (int64_t)(n0 + ws_perm(rho)) * d.ldb + kc * 4);
            }
        } else if constexpr (!WKM) {
            const bf16_t* B = reinterpret_cast<const bf16_t*>(d.B) + c * K;
            constexpr int CPR = K / 8;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int idx = tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                wv[i] = *reinterpret_cast<const uint4*>(B + (int64_t)(n0 + ws_perm(rho)) * d.ldb + kc * 8);
            }
        } else {                                                         // W[k][n]: item = 4 consecutive k x 8 consecutive n
            const bf16_t* B = reinterpret_cast<const bf16_t*>(d.B) + (int64_t)c * K * d.ldb;
            constexpr int NG = BN / 8;
            const int kq = tid / NG, nl = (tid - kq * NG) * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = *reinterpret_cast<const uint4*>(B + (int64_t)(kq * 4 + r) * d.ldb + n0 + nl);
        }
    };
    auto w_store = [&](const uint4 (&wv)[NWV], char* buf) __attribute__((always_inline)) {
        if constexpr (SPLIT) {
            constexpr int CPR = K / 4;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int idx = tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                const float f0 = __uint_as_float(wv[i].x), f1 = __uint_as_float(wv[i].y), f2 = __uint_as_float(wv[i].z), f3 = __uint_as_float(wv[i].w);
                const uint2 hi = make_uint2(pack_bf2(f0, f1), pack_bf2(f2, f3));
                const uint2 lo = make_uint2(pack_bf2(f0 - __uint_as_float(hi.x << 16), f1 - __uint_as_float(hi.x & 0xffff0000u)),
                                            pack_bf2(f2 - __uint_as_float(hi.y << 16), f3 - __uint_as_float(hi.y & 0xffff0000u)));
                *reinterpret_cast<uint2*>(buf + ws_sw<PITCH>(rho, kc * 8)) = hi;
                *reinterpret_cast<uint2*>(buf + IMG + ws_sw<PITCH>(rho, kc * 8)) = lo;
            }
        } else if constexpr (!WKM) {
            constexpr int CPR = K / 8;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int idx = tid + i * NTH, rho = idx / CPR, kc = idx - rho * CPR;
                // (element by element: copied as a whole 16-byte object, the staging array is kept in scratch memory)
                *reinterpret_cast<uint4*>(buf + ws_sw<PITCH>(rho, kc * 16)) = make_uint4(wv[i].x, wv[i].y, wv[i].z, wv[i].w);
            }
        } else {
            constexpr int NG = BN / 8;
            const int kq = tid / NG, nl = (tid - kq * NG) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                uint32_t h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t w = (e >> 1) == 0 ? wv[r].x : (e >> 1) == 1 ? wv[r].y : (e >> 1) == 2 ? wv[r].z : wv[r].w;
                    h[r] = (e & 1) ? (w >> 16) : (w & 0xffffu);
                }
                *reinterpret_cast<uint2*>(buf + ws_sw<PITCH>(ws_inv_perm(nl + e), kq * 8)) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            }
        }
    };

    uint4 a[FM][KS], an[FM][KS];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[fm][kk] = *reinterpret_cast<const uint4*>(ap[fm] + kk * 32);
    uint4 wv[NWV];
    w_load(wv, 0);
    if (tid < BN) sbias[tid] = d.bias ? d.bias[n0 + tid] : 0.f;
    f32x4_t acc[FM][FNT];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int jn = 0; jn < FNT; ++jn) acc[i][jn] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    w_store(wv, smem);
    __syncthreads();
    int wsw[4];                                                         // fragment read offsets of this lane: k-step kk uses wsw[kk & 3]
#pragma unroll
    for (int j = 0; j < 4; ++j) wsw[j] = ws_sw<PITCH>(frow, (j * 4 + g) * 16);

#pragma unroll 1
    for (int c = 0; c < KC; ++c) {
        const bool more = c + 1 < KC;
        if (more) {                                                      // chunk c+1: everything global goes in flight now
            w_load(wv, c + 1);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) an[fm][kk] = *reinterpret_cast<const uint4*>(ap[fm] + (c + 1) * K + kk * 32);
        }
        asm volatile("" : "+v"(wsw[0]), "+v"(wsw[1]), "+v"(wsw[2]), "+v"(wsw[3]));
        const char* wb = smem + (c & 1) * BUF;
        const char* wl[4] = {wb + wsw[0], wb + wsw[1], wb + wsw[2], wb + wsw[3]};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int jn = 0; jn < FNT; ++jn) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wl[kk & 3] + jn * 16 * PITCH + (kk >> 2) * 256));
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    acc[fm][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, a[fm][kk]), acc[fm][jn], 0, 0, 0);
                if constexpr (SPLIT) {
                    const bf16x8_t wlo = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wl[kk & 3] + IMG + jn * 16 * PITCH + (kk >> 2) * 256));
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm)
                        acc[fm][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, __builtin_bit_cast(bf16x8_t, a[fm][kk]), acc[fm][jn], 0, 0, 0);
                }
            }
        }
        if (more) {
            w_store(wv, smem + ((c + 1) & 1) * BUF);                         // (its last readers passed the barrier that ended chunk c-1)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) a[fm][kk] = an[fm][kk];
            __syncthreads();
        }
    }

    // ---- epilogue on registers (as gemm_ws_kernel) ----
    const TC* addp = reinterpret_cast<const TC*>(d.add_src);
    const TC* gate = reinterpret_cast<const TC*>(d.gate_ref);
    TC* C = reinterpret_cast<TC*>(d.C);
    const uint32_t sd = d.seed ^ (d.seed_dev ? *d.seed_dev * 0x9E3779B1u : 0u);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const bool live = grow[fm] < d.M;
        const int gr = min(grow[fm], d.M - 1);
        uint32_t mk = 0u;
        if constexpr (MASK) mk = d.row_mask[gr];
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int lcol = jq * 64 + h * 32 + g * 8, gcol = n0 + lcol;
                float v[8];
#pragma unroll
                for (int lo = 0; lo < 2; ++lo)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[lo * 4 + t] = acc[fm][jq * 4 + h * 2 + lo][t];
                    if (d.alpha != 1.f) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= d.alpha;
                    }
                {
                    float b[8];
                    vec<float, 8>::ld(sbias + lcol, b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += b[e];
                }
                if (d.act == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = relu1(v[e]);
                }
                if constexpr (GATE) {
                    uint4 gq[NV];
                    float gt[8];
                    run8<TC>::ld(gate + (int64_t)gr * d.ldc + gcol, gq);
                    run8<TC>::dec(gq, gt);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gt[e] > 0.f ? v[e] * d.gate_scale : 0.f;
                }
                if (p.drop_thresh) {
                    const uint32_t base = (uint32_t)gr * (uint32_t)d.N + (uint32_t)gcol;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const uint32_t hsh = drop_pair(sd, (base + e) >> 1);
                        v[e] = (hsh & 0xffffu) >= p.drop_thresh ? v[e] * p.drop_scale : 0.f;
                        v[e + 1] = (hsh >> 16) >= p.drop_thresh ? v[e + 1] * p.drop_scale : 0.f;
                    }
                }
                if constexpr (ADD) {
                    uint4 rq[NV];
                    float r[8];
                    run8<TC>::ld(addp + (int64_t)gr * d.ld_add + gcol, rq);
                    run8<TC>::dec(rq, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if constexpr (MASK) {
                    if (mk & 0xffu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 0.f;
                    }
                }
                if (live) {
                    if (d.out_mode == 1) {
                        const int hn = gr / d.hm_S, hs = gr - hn * d.hm_S, hm = gcol / d.hm_D, hd = gcol - hm * d.hm_D;
                        vec<TC, 8>::st(C + (((int64_t)hn * d.hm_M + hm) * d.hm_S + hs) * d.hm_D + hd, v);
                    } else {
                        vec<TC, 8>::st(C + (int64_t)gr * d.ldc + gcol, v);
                    }
                }
            }
    }
}

template <typename TC, int KIND, bool WKM, bool SPLIT, int BN = 128, int NTH = 512>
bool wsk_launch(const GemmK& p, hipStream_t st) {
    constexpr int LDS = 2 * (SPLIT ? 2 : 1) * BN * (4 * 64) + BN * 4;
    auto kern = gemm_wsk_kernel<TC, KIND, WKM, SPLIT, BN, NTH>;
    static unsigned long long attr_done = 0;          // per instantiation and device
    lds_attr_once(reinterpret_cast<const void*>(kern), LDS, attr_done);
    const int NT = p.d.N / BN, RB = (p.d.M + NTH / 2 - 1) / (NTH / 2);
    const int nblocks = ((RB + 7) / 8) * NT * 8;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NTH), LDS, st, p);
    return true;
}

template <typename TC>
bool wsk_kind(const GemmK& p, hipStream_t st) {
    const PoetGemmDesc& d = p.d;
    const int kind = (d.gate_ref ? WS_GATE : 0) | (d.add_src ? WS_ADD : 0) | (d.row_mask ? WS_MASK : 0);
    if (d.b_split) {
        // 64-column slices: the two weight images of a 128-wide chunk would leave ONE workgroup per CU (139 KB), whose
        // activation loads (one chunk = 1.8 us of MFMAs ahead) then sit exposed; at 70 KB two 4-wave workgroups cover each other
        static const int bn = [] { const char* e = getenv("POET_WSK_SPLIT_BN"); return e ? atoi(e) : 64; }();
        if (kind != 0) return false;
        return bn == 128 ? wsk_launch<TC, 0, false, true, 128, 512>(p, st) : wsk_launch<TC, 0, false, true, 64, 256>(p, st);
    }
    if (!d.b_kmajor) {
        if (kind == 0) return wsk_launch<TC, 0, false, false>(p, st);
        return false;
    }
    if constexpr (sizeof(TC) == 4) {
        if (kind == WS_ADD) return wsk_launch<TC, WS_ADD, true, false>(p, st);       // FFN1 dX accumulated onto the fp32 stream gradient
        if (kind == 0) return wsk_launch<TC, 0, true, false>(p, st);
        return false;
    } else {
        switch (kind) {
            case 0: return wsk_launch<TC, 0, true, false>(p, st);
            case WS_ADD: return wsk_launch<TC, WS_ADD, true, false>(p, st);
            case WS_GATE: return wsk_launch<TC, WS_GATE, true, false>(p, st);
            default: return false;
        }
    }
}

template <typename TC, int KIND, bool WKM, int FM, int NTH, int WM = 0, int BN = 128>
void ws_launch_cfg(const GemmK& p, int nblocks, hipStream_t st) {
    constexpr int LDS = (WM == 1 ? 2 : 1) * BN * (8 * 64) + BN * 4;
    auto kern = gemm_ws_kernel<TC, KIND, WKM, FM, NTH, WM, BN>;
    static unsigned long long attr_done = 0;          // per instantiation and device
    lds_attr_once(reinterpret_cast<const void*>(kern), LDS, attr_done);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NTH), LDS, st, p);
}

// Two shapes of the same kernel.  Narrow outputs (N <= 256: 1-2 column slices) are latency-bound: 16 rows per wave need
// ~120-150 VGPRs, so eight waves share one weight slice and 3-4 waves per SIMD hide the latency (and the f32-residual
// variants stop spilling).  Wide outputs (N >= 512) re-use every weight fragment for two row fragments instead, which
// halves the LDS traffic that would otherwise bound them (measured at M = 102080: N = 256 31.3 -> 29.3 us with 16 rows,
// N = 1024 90 -> 105 us).
template <typename TC, int KIND, bool WKM>
bool ws_launch(const GemmK& p, int nblocks, hipStream_t st) {
    if (p.d.N <= 256) { ws_launch_cfg<TC, KIND, WKM, 1, 512>(p, nblocks, st); return true; }
    // two row fragments per wave + prefetched epilogue operands: the f32-residual / gate+residual forms would spill
    // (scratch memory = 2-3x slower than the tiled kernel they would replace); they do not occur on the path
    if constexpr ((sizeof(TC) == 4 && (KIND & (WS_ADD | WS_GATE))) || KIND == (WS_GATE | WS_ADD)) return false;
    else { ws_launch_cfg<TC, KIND, WKM, 2, 256>(p, nblocks, st); return true; }
}

template <typename TC, bool ACT, bool DROP>
void wsp_launch(const GemmK& p, int nblocks, hipStream_t st) {
    constexpr int LDS = 2 * 128 * 512 + 128 * 4;
    auto kern = gemm_wsp_kernel<TC, ACT, DROP>;
    static unsigned long long attr_done = 0;          // per instantiation and device
    lds_attr_once(reinterpret_cast<const void*>(kern), LDS, attr_done);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), LDS, st, p);
}

// b_split shapes.  The two weight images take 135 KB of LDS for a 128-column slice: ONE workgroup per CU with twice the
// waves keeps the occupancy of the plain kernel (wide outputs: 8 waves; narrow: 16 waves need <= 128 VGPRs -- variant 0),
// or a 64-column slice keeps two workgroups per CU at twice the A re-reads through L2 (variant 1; variant 2, the default =
// one 8-wave workgroup per CU on a 128-column slice: 13.94 against 14.01 ms per step at 640x480, 31.96 / 32.07 at 1280x960,
// no difference at the LM-O shape).  POET_WS_SPLIT_NARROW selects the narrow-output variant (A/B aid).
int ws_blocks(int N, int BN, int wg_per_cu);
template <typename TC, int KIND>
bool ws_launch_split(const GemmK& p, hipStream_t st) {
    static const int narrow = [] { const char* e = getenv("POET_WS_SPLIT_NARROW"); return e ? atoi(e) : 2; }();
    if (p.d.N <= 256) {
        if (narrow == 2) ws_launch_cfg<TC, KIND, false, 1, 512, 1, 128>(p, ws_blocks(p.d.N, 128, 1), st);
        else ws_launch_cfg<TC, KIND, false, 1, 512, 1, 64>(p, ws_blocks(p.d.N, 64, 2), st);
        return true;
    }
    if constexpr (sizeof(TC) == 4 && (KIND & (WS_ADD | WS_GATE))) return false;
    else {
        if constexpr (KIND == 0 && sizeof(TC) == 2) {
            // the software-pipelined form (gemm_wsp_kernel): plain stores, no epilogue operands, alpha = 1.  POET_WS_PIPE=0: off (A/B aid)
            static const int pipe = [] { const char* e = getenv("POET_WS_PIPE"); return e ? atoi(e) : 1; }();
            const PoetGemmDesc& d = p.d;
            if (pipe && d.out_mode == 0 && d.alpha == 1.f && (d.act == 0 || d.act == 1) && (p.drop_thresh == 0 || d.act == 1)) {
                const int nb = ws_blocks(d.N, 128, 1);
#ifdef POET_PROBE_KERNELS
                if (pipe == 2 && (uint64_t)d.M * (uint64_t)d.lda * 2u < (1ull << 32)) {      // (32-bit byte offsets into A)
                    if (d.act == 1 && p.drop_thresh) wss_launch<TC, true, true>(p, nb, st);
                    else if (d.act == 1) wss_launch<TC, true, false>(p, nb, st);
                    else wss_launch<TC, false, false>(p, nb, st);
                    return true;
                }
#endif
                if (d.act == 1 && p.drop_thresh) wsp_launch<TC, true, true>(p, nb, st);
                else if (d.act == 1) wsp_launch<TC, true, false>(p, nb, st);
                else wsp_launch<TC, false, false>(p, nb, st);
                return true;
            }
        }
        ws_launch_cfg<TC, KIND, false, 2, 512, 1, 128>(p, ws_blocks(p.d.N, 128, 1), st);
        return true;
    }
}

// single fp16 weight image (b_split = 2): the LDS footprint and the shapes of the single-image bf16 form
template <typename TC, int KIND>
bool ws_launch_w16(const GemmK& p, int nblocks, hipStream_t st) {
    if (p.d.N <= 256) ws_launch_cfg<TC, KIND, false, 1, 512, 2, 128>(p, nblocks, st);
    else ws_launch_cfg<TC, KIND, false, 2, 256, 2, 128>(p, nblocks, st);
    return true;
}

// the epilogue kinds that occur on the path: forward {plain, +residual, +row mask}, input gradient {plain, ReLU gate,
// accumulate, both}; anything else goes to the generic kernel
template <typename TC>
bool ws_kind(const GemmK& p, int nblocks, hipStream_t st) {
    const PoetGemmDesc& d = p.d;
    const int kind = (d.gate_ref ? WS_GATE : 0) | (d.add_src ? WS_ADD : 0) | (d.row_mask ? WS_MASK : 0);
    if constexpr (sizeof(TC) == 2) {
        // fp16 outputs (c_f16): the forward without add_src / gate_ref, specialised on the output type; anything else goes to the
        // generic kernel
        // (round 6: + the row-masked head-major form = the encoder's fp16 value maps, split or single-image weights)
        if (d.c_f16) {
            if (d.b_split == 2) return kind == 0 ? ws_launch_w16<f16_t, 0>(p, nblocks, st) : kind == WS_MASK ? ws_launch_w16<f16_t, WS_MASK>(p, nblocks, st) : false;
            if (d.b_split) return kind == 0 ? ws_launch_split<f16_t, 0>(p, st) : kind == WS_MASK ? ws_launch_split<f16_t, WS_MASK>(p, st) : false;
            if (d.b_kmajor) return false;
            return kind == 0 ? ws_launch<f16_t, 0, false>(p, nblocks, st) : kind == WS_MASK ? ws_launch<f16_t, WS_MASK, false>(p, nblocks, st) : false;
        }
    }
    if (d.b_split == 2) {
        if constexpr (sizeof(TC) == 2) {
            switch (kind) {
                case 0: return ws_launch_w16<TC, 0>(p, nblocks, st);
                case WS_MASK: return ws_launch_w16<TC, WS_MASK>(p, nblocks, st);
                default: return false;
            }
        } else return false;
    }
    if (d.b_split) {
        switch (kind) {
            case 0: return ws_launch_split<TC, 0>(p, st);
            case WS_ADD: return ws_launch_split<TC, WS_ADD>(p, st);
            case WS_MASK: return ws_launch_split<TC, WS_MASK>(p, st);
            default: return false;
        }
    }
    if (!d.b_kmajor) {
        switch (kind) {
            case 0: return ws_launch<TC, 0, false>(p, nblocks, st);
            case WS_ADD: return ws_launch<TC, WS_ADD, false>(p, nblocks, st);
            case WS_MASK: return ws_launch<TC, WS_MASK, false>(p, nblocks, st);
            default: return false;
        }
    }
    switch (kind) {
        case 0: return ws_launch<TC, 0, true>(p, nblocks, st);
        case WS_GATE: return ws_launch<TC, WS_GATE, true>(p, nblocks, st);
        case WS_ADD: return ws_launch<TC, WS_ADD, true>(p, nblocks, st);
        case WS_GATE | WS_ADD:
            if constexpr (sizeof(TC) == 2) return ws_launch<TC, WS_GATE | WS_ADD, true>(p, nblocks, st);
            return false;                                               // f32: 128 VGPRs of prefetched epilogue operands
        default: return false;
    }
}

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

int ws_blocks(int N, int BN, int wg_per_cu) {
    const int NT = N / BN;
    int per = wg_per_cu * device_cus() / 8;                             // workgroups per XCD
    if (per < NT) per = NT;
    return per * 8;
}

}  // namespace

bool gemm_ws_try(const GemmK& p, hipStream_t st) {
    const PoetGemmDesc& d = p.d;
    static const int disabled = [] { const char* e = getenv("POET_GEMM_NO_WS"); return e && atoi(e) ? 1 : 0; }();
    if (disabled) return false;
    if (d.compute != POET_BF16 || d.a_dtype != POET_BF16 || d.b_dtype != (d.b_split ? POET_F32 : POET_BF16) || d.a_kmajor || d.batch != 1 ||
        d.splitk != 1 || d.atomic || d.A2)
        return false;
    if (d.b_split && ((reinterpret_cast<uintptr_t>(d.B) & 15) || (d.ldb & 3))) return false;
    if (d.N % 128 != 0 || d.M < 4096) return false;
    if (!p.a_vec || !p.b_vec || !p.c_vec) return false;
    if (d.out_mode == 1 && d.hm_D % 8 != 0) return false;
    if (d.K != 256 && d.b_split == 2) return false;                     // (the fp16-weight form exists at K = 256 only)
    if (d.K != 256) {                                                   // K = 512 / 768 / 1024 ...: the K-chunked kernel
        static const int no_wsk = [] { const char* e = getenv("POET_GEMM_NO_WSK"); return e && atoi(e) ? 1 : 0; }();
        // (single-weight forms are still faster on the tiled kernel: the W chunk restaging of a one-workgroup-per-CU kernel is
        // exposed; POET_GEMM_WSK_ALL=1 routes them here anyway)
        static const int all = [] { const char* e = getenv("POET_GEMM_WSK_ALL"); return e && atoi(e) ? 1 : 0; }();
        if (no_wsk || d.K % 128 != 0 || d.K > 4096 || (!d.b_split && !all) || d.c_f16) return false;
        return d.c_dtype == POET_BF16 ? wsk_kind<bf16_t>(p, st) : wsk_kind<float>(p, st);
    }
    const int NT = d.N / 128;
    int per = 2 * device_cus() / 8;                                     // two workgroups per CU, per XCD
    if (per < NT) per = NT;
    const int nblocks = per * 8;
    return d.c_dtype == POET_BF16 ? ws_kind<bf16_t>(p, nblocks, st) : ws_kind<float>(p, nblocks, st);
}

}  // namespace poet
