// Deep-pipeline tall-skinny GEMM for gfx950:  C[M, 256] (fp32) (+)= A[M, K] (bf16) * W (bf16 [, + W_lo])  (+ bias),
// M ~ 1e5 token rows, K >= 256 a multiple of 64 -- the long-K Linears of the encoder layer seen from backward
// (models/deformable_transformer.py:193-197 forward_ffn: d(src) += d(hidden) W1; :201 MSDeformAttn: d(src) +=
// [d(offsets|logits) | d(value) rows] [W_so ; W_aw ; W_v]; the decoder's d(memory) = d(values) W_v with K = 1280) and the FFN's
// second Linear forward with split-bf16 weights (:185 linear2, hi + lo in ONE pass over the hidden activation).
//
// These products are HBM-bound (A 209 MB + C 104..209 MB against 54 GFLOP at K = 1024), so the kernel is organised around
// keeping every CU's load stream full from the first to the last byte:
//
//  * persistent: one 512-thread workgroup per CU owns a CONTIGUOUS range of 16-row units (24.9 units per CU at 102 080 rows,
//    dealt 24/25 -- no tile quantisation, no stream-K fix-up, results independent of the grid) and cuts it into nearly equal
//    tiles of <= 8 units; the k-loops of its tiles form ONE flat sequence of steps;
//  * operands reach LDS by DMA (global_load_lds_dwordx4, no VGPRs, no ds_write): a ring of NST stages (A 128 x BK | W BK x 256
//    [| W_lo]), PD = NST - 1 stages in flight ACROSS tile boundaries, so the C read-modify-write of one tile runs under the
//    operand loads of the next; counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per step (the DMA and the C loads are
//    inline asm: hipcc would drain the queue with vmcnt(0) in front of every ds_read);
//  * LDS images are plain row-major copies of memory with XOR-swizzled 16-byte chunks (the swizzle is applied to the per-lane
//    SOURCE address of the DMA and to the read address): ds_read_b128 fragments are conflict-free, and a weight stored [K][N]
//    (the input-gradient form) is read with ds_read_b64_tr_b16, the CDNA4 transposing LDS read -- no transposed weight copy;
//  * all 8 waves hold all rows of the tile and 32 of the 256 columns (1 x 8 wave grid): tile heights of 1..8 units cost
//    MFMA time in proportion; the product is computed transposed (W fragment as the MFMA's A operand) so that a lane owns
//    4 consecutive columns of one row: 16-byte C loads / stores, 64 contiguous bytes per row and instruction;
//  * the C loads of a tile go out in its first step and are added to the accumulators PD steps later; its stores go out in
//    the first step of the NEXT tile, straight from the old accumulator registers -- neither latency is exposed, and no
//    second register set lives through a k-loop.
#include "gemm.cuh"

#include <stdlib.h>
#include <type_traits>

namespace poet {

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;

constexpr int PIPE_BN = 256;         // output columns (the whole N)
constexpr int PIPE_MAXF = 8;         // 16-row units per tile
constexpr int PIPE_NT = 512;         // threads

struct PipeP {
    const bf16_t* A;
    const bf16_t* W;
    const bf16_t* Wlo;
    float* C;
    const float* bias;
    int64_t lda, ldb, ldc;           // elements
    int M, K;
    unsigned long long* prof;        // DBG & 16: per workgroup 8 cycle counters (wait, barrier, issue, compute, ...)
    int dbg;                         // timing experiments only (POET_PIPE_DBG -> template DBG): 1 = no W DMA, 2 = no MFMA, 4 = no A DMA, 8 = no C stores
};

// chunk swizzles of the row-major LDS images (16-byte chunks of a row XORed with a function of the row):
// 128-byte rows (BK = 64): two rows share a 256-byte bank line; rows r and r + 8.. would meet, and the 16-lane groups of
// ds_read_b128 mix chunk c (rows 0-3, 12-15) with chunk c ^ 1 (rows 4-11)
__device__ __forceinline__ int swz128(int row) { const int j = (row >> 1) & 7; return j ^ (((j + 2) >> 2) & 1); }
// 64-byte rows (BK = 32): four rows per bank line
__device__ __forceinline__ int swz64(int row) { const int x = (row >> 2) & 3; return x ^ ((x & 1) << 1); }
template <int BK> __device__ __forceinline__ int swz_row(int row) { return BK == 64 ? swz128(row) : swz64(row); }
// [K][N] weight image (512-byte rows): 32-byte pieces inside each 256-byte half row, for the transposing read
__device__ __forceinline__ int swz_km(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// one LDS-DMA instruction: 64 lanes x 16 bytes from `base + voff[lane]` to LDS bytes [lds, lds + 1024) in lane order
__device__ __forceinline__ void dma16(uint32_t voff, const void* base, uint32_t lds) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ void gload16(f32x4_t& dst, uint32_t voff, const void* base) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BK, bool WKM, bool SPLIT, bool ACC, int NST, int STSLACK, int DBG = 0>
__global__ __launch_bounds__(PIPE_NT, 2) void gemm_pipe_kernel(const PipeP p) {
    constexpr int PD = NST - 1;
    constexpr int ROWB = BK * 2, CPR = ROWB / 16;
    constexpr int A_BYTES = 128 * ROWB, W_BYTES = BK * 512;
    constexpr int STAGE = A_BYTES + (SPLIT ? 2 : 1) * W_BYTES;
    constexpr int NA = A_BYTES / 8192, NW = W_BYTES / 8192;           // DMA instructions per thread and stage
    constexpr int NPS = NA + (SPLIT ? 2 : 1) * NW;
    constexpr int LW = NPS * (PD - 1);                                 // loads allowed in flight at the top of a step
    constexpr int NC = 2 * PIPE_MAXF;                                  // most C loads / stores per thread and tile
    constexpr int KH = BK / 32;
    static_assert(NA >= 1 && LW + NC + 16 <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kc = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- this workgroup's rows: units [u0, u0 + nu), cut into nt tiles of tb (+1) units ----
    const int NU = (p.M + 15) >> 4, G = gridDim.x, g = blockIdx.x;
    const int nu = NU / G + (g < NU % G ? 1 : 0);
    if (nu == 0) return;
    const int u0 = g * (NU / G) + min(g, NU % G);
    const int nt = (nu + PIPE_MAXF - 1) / PIPE_MAXF, tb = nu / nt, tr = nu % nt;
    const int nk = p.K / BK, S = nt * nk;
    auto tile_u0 = [&](int ti) { return u0 + ti * tb + min(ti, tr); };
    auto tile_nf = [&](int ti) { return tb + (ti < tr ? 1 : 0); };

    // ---- per-lane DMA source offsets (bytes) ----
    const int ldaB = (int)p.lda * 2, ldbB = (int)p.ldb * 2;
    int a_row[NA], a_col[NA];
    uint32_t w_off[NW];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int idx = q * PIPE_NT + tid, row = idx / CPR, pc = idx % CPR;
        a_row[q] = row;
        a_col[q] = (pc ^ swz_row<BK>(row)) * 16;
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int idx = q * PIPE_NT + tid;
        if constexpr (WKM) {
            const int krow = idx >> 5, pg = idx & 31, pp = pg >> 1;
            const int lp = (pp & 8) | ((pp ^ swz_km(krow)) & 7);
            w_off[q] = (uint32_t)(krow * ldbB + lp * 32 + (pg & 1) * 16);
        } else {
            const int n = idx / CPR, pc = idx % CPR;
            w_off[q] = (uint32_t)(n * ldbB + (pc ^ swz_row<BK>(n)) * 16);
        }
    }
    // ---- per-lane LDS read offsets ----
    int a_rd[KH];                                                       // fragment row m16, chunk h*4 + kc (swizzled)
#pragma unroll
    for (int h = 0; h < KH; ++h) a_rd[h] = m16 * ROWB + (((h * 4 + kc) & (CPR - 1)) ^ swz_row<BK>(m16)) * 16;
    int w_rd[2];                                                        // [K][N] image: transposing read of fragment j
    if constexpr (WKM) {
        const int krl = 8 * kc + (m16 >> 2), sw = (m16 >> 2) | ((kc & 1) << 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave * 2 + j;
            w_rd[j] = krl * 512 + (((piece ^ sw) & 7) | (piece & 8)) * 32 + (m16 & 3) * 8;
        }
    }

    // ---- loader state: the stage PD steps ahead of the one being computed ----
    int l_tile = 0, l_kk = 0, l_slot = 0, l_left = S;
    uint32_t a_off[NA];
    const char* l_abase;
    auto l_new_tile = [&]() {
        const int r0 = tile_u0(l_tile) * 16, rv = min(tile_nf(l_tile) * 16, p.M - r0);
#pragma unroll
        for (int q = 0; q < NA; ++q) a_off[q] = (uint32_t)(min(a_row[q], rv - 1) * ldaB + a_col[q]);
        l_abase = reinterpret_cast<const char*>(p.A) + (int64_t)r0 * ldaB;
    };
    l_new_tile();
    auto issue = [&]() {                                                // DMA of one stage into ring slot l_slot
        const uint32_t slot = lds0 + l_slot * STAGE + wave * 1024;
        const char* ab = l_abase + l_kk * ROWB;
        if constexpr (!(DBG & 4)) {
#pragma unroll
        for (int q = 0; q < NA; ++q) dma16(a_off[q], ab, slot + q * 8192);
        }
        if constexpr (!(DBG & 1)) {
        const char* wb = reinterpret_cast<const char*>(p.W) + (WKM ? (int64_t)l_kk * BK * ldbB : (int64_t)l_kk * ROWB);
#pragma unroll
        for (int q = 0; q < NW; ++q) dma16(w_off[q], wb, slot + A_BYTES + q * 8192);
        if constexpr (SPLIT) {
            const char* wl = reinterpret_cast<const char*>(p.Wlo) + (WKM ? (int64_t)l_kk * BK * ldbB : (int64_t)l_kk * ROWB);
#pragma unroll
            for (int q = 0; q < NW; ++q) dma16(w_off[q], wl, slot + A_BYTES + W_BYTES + q * 8192);
        }
        }
        --l_left;
        l_slot = (l_slot + 1 == NST) ? 0 : l_slot + 1;
        if (++l_kk == nk) {
            l_kk = 0;
            if (++l_tile < nt) l_new_tile();
        }
    };
#pragma unroll 1
    for (int i = 0; i < PD; ++i) issue();                               // (S >= nk >= 4 > PD)

    // ---- bias: this lane's 4 columns of each of its 2 fragments ----
    float bia[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) bia[j][t] = (!ACC && p.bias) ? p.bias[wave * 32 + j * 16 + kc * 4 + t] : 0.f;

    // ---- output bookkeeping ----
    const int ldcB = (int)p.ldc * 4;
    const uint32_t c_lane = (uint32_t)(m16 * ldcB + (wave * 32 + kc * 4) * 4);   // + i * 16 * ldcB + j * 64
    f32x4_t outv[PIPE_MAXF][2];
    int pend_nf = 0, pend_r0 = 0;
    auto flush = [&]() {                                                // stores of the finished tile (plain, compiler-visible)
        char* cb = reinterpret_cast<char*>(p.C) + (int64_t)pend_r0 * ldcB;
#pragma unroll
        for (int i = 0; i < PIPE_MAXF; ++i) {
            if (i < pend_nf && pend_r0 + i * 16 + m16 < p.M && !(DBG & 8)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4_t*>(cb + c_lane + i * 16 * ldcB + j * 64) = outv[i][j];
            }
        }
        pend_nf = 0;
    };

    int c_slot = 0, s = 0;                                              // compute slot, flat step index
    unsigned long long pw = 0, pb = 0, pi = 0, pc = 0, p0 = 0, pstart = 0;
    if constexpr (DBG & 16) pstart = __builtin_amdgcn_s_memtime();
    auto run_tile = [&](auto nf_tag, int ti) {
        constexpr int NF = decltype(nf_tag)::value;
        constexpr int NCL = ACC ? 2 * NF : 0;                            // C loads of this tile per thread
        const int r0 = tile_u0(ti) * 16;
        f32x4_t acc[NF][2], cin[ACC ? NF : 1][2];
        // one step: wait for stage s, barrier, [stores of the previous tile, C loads of this one], DMA of stage s + PD, MFMAs.
        // PH: 0 = first step of the tile (the previous tile's stores and this tile's C loads go out), 1 = steps 1 .. PD-1 (the C
        // loads are younger than the stage waited for), 2 = step PD (every load older than the newest PD-1 stages has landed,
        // the C loads among them: acc += C), 3 = the rest.  Steps 0 .. PD are straight-line code: the registers the asm C
        // loads write must not pass through a loop-carried copy before their wait.
        auto step = [&](auto ph_tag) {
            constexpr int PH = decltype(ph_tag)::value;
            unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if constexpr (DBG & 16) t0 = __builtin_amdgcn_s_memtime();
            // stage s has landed when at most the loads issued after it are outstanding
            const int later = min(S, s + PD) - (s + 1);                  // stages issued after stage s
            if (later < PD - 1) wait_vm<0>();
            else if (PH == 1 && STSLACK && ti > 0) {                     // (optimistic A/B variant: stores complete in issue order)
                if (tb >= 6) wait_vm<LW + NCL + 12>(); else wait_vm<LW + NCL>();
            } else if (PH == 1) wait_vm<LW + NCL>();
            else wait_vm<LW>();
            if constexpr (DBG & 16) t1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (DBG & 16) t2 = __builtin_amdgcn_s_memtime();
            if constexpr (PH == 0) {
                if (pend_nf) flush();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < NF; ++i)                             // (after the flush: the old accumulators are dead now)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = ACC ? f32x4_t{0.f, 0.f, 0.f, 0.f} : f32x4_t{bia[j][0], bia[j][1], bia[j][2], bia[j][3]};
                if constexpr (ACC) {                                     // C of this tile: lands under PD steps of MFMAs
                    const char* cb = reinterpret_cast<const char*>(p.C) + (int64_t)r0 * ldcB;
                    const int rv = min(NF * 16, p.M - r0);
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const uint32_t ro = (uint32_t)(min(i * 16 + m16, rv - 1) * ldcB) + (c_lane - (uint32_t)(m16 * ldcB));
#pragma unroll
                        for (int j = 0; j < 2; ++j) gload16(cin[i][j], ro + j * 64, cb);
                    }
                }
            }
            if constexpr (PH == 2 && ACC) {
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(cin[i][j]));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] += cin[i][j];
            }
            if (l_left > 0) issue();
            if constexpr (DBG & 16) t3 = __builtin_amdgcn_s_memtime();
            // ---- MFMAs of step s out of ring slot c_slot ----
            const char* sl = smem + c_slot * STAGE;
            c_slot = (c_slot + 1 == NST) ? 0 : c_slot + 1;
            if constexpr (!(DBG & 2)) {
#pragma unroll
            for (int h = 0; h < KH; ++h) {
                bf16x8_t wf[2], wl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (WKM) {
                        const char* wp = sl + A_BYTES + h * 32 * 512 + w_rd[j];
                        struct { v4s_t lo, hi; } u;
                        u.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(wp));
                        u.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(wp + 4 * 512));
                        wf[j] = __builtin_bit_cast(bf16x8_t, u);
                        if constexpr (SPLIT) {
                            u.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(wp + W_BYTES));
                            u.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(wp + W_BYTES + 4 * 512));
                            wl[j] = __builtin_bit_cast(bf16x8_t, u);
                        }
                    } else {
                        const char* wp = sl + A_BYTES + (wave * 32 + j * 16) * ROWB + a_rd[h];
                        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp));
                        if constexpr (SPLIT) wl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + W_BYTES));
                    }
                }
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const bf16x8_t af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sl + i * 16 * ROWB + a_rd[h]));
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[i][j], 0, 0, 0);
                        if constexpr (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], af, acc[i][j], 0, 0, 0);
                    }
                }
            }
            }
            asm volatile("" ::: "memory");
            if constexpr (DBG & 16) {
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                pw += t1 - t0; pb += t2 - t1; pi += t3 - t2; pc += t4 - t3;
                if (PH == 0) p0 += t3 - t2;
            }
            ++s;
        };
        step(std::integral_constant<int, 0>{});
#pragma unroll
        for (int kk = 1; kk < PD; ++kk) step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
#pragma unroll 1
        for (int kk = PD + 1; kk < nk; ++kk) step(std::integral_constant<int, 3>{});
        // ---- tile done: its stores go out at the top of the next step, from the accumulator registers themselves ----
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) outv[i][j] = acc[i][j];
        pend_nf = NF;
        pend_r0 = r0;
    };

#pragma unroll 1
    for (int ti = 0; ti < nt; ++ti) {
        switch (tile_nf(ti)) {
            case 1: run_tile(std::integral_constant<int, 1>{}, ti); break;
            case 2: run_tile(std::integral_constant<int, 2>{}, ti); break;
            case 3: run_tile(std::integral_constant<int, 3>{}, ti); break;
            case 4: run_tile(std::integral_constant<int, 4>{}, ti); break;
            case 5: run_tile(std::integral_constant<int, 5>{}, ti); break;
            case 6: run_tile(std::integral_constant<int, 6>{}, ti); break;
            case 7: run_tile(std::integral_constant<int, 7>{}, ti); break;
            default: run_tile(std::integral_constant<int, 8>{}, ti); break;
        }
    }
    flush();
    if constexpr (DBG & 16) {
        if (p.prof && lane == 0 && (wave == 0 || wave == 7)) {
            unsigned long long* o = p.prof + (blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
            o[0] = pw; o[1] = pb; o[2] = pi; o[3] = pc; o[4] = p0; o[5] = __builtin_amdgcn_s_memtime() - pstart; o[6] = S; o[7] = pstart;
        }
    }
}

int pipe_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <int BK, bool WKM, bool SPLIT, bool ACC, int NST, int STSLACK, int DBG = 0>
void pipe_launch(const PipeP& p, int grid, hipStream_t st) {
    constexpr int STAGE = 128 * BK * 2 + (SPLIT ? 2 : 1) * BK * 512, LDS = NST * STAGE;
    static_assert(LDS <= 163840, "LDS ring");
    auto kern = gemm_pipe_kernel<BK, WKM, SPLIT, ACC, NST, STSLACK, DBG>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PIPE_NT), LDS, st, p);
}

template <bool WKM, bool SPLIT, bool ACC>
void pipe_dispatch(const PipeP& p, int grid, int cfg, hipStream_t st) {
    // cfg (POET_PIPE_CFG, A/B aid): 0 default; 1 = the other K stage width; 2 = optimistic store slack
    if constexpr (SPLIT) {
        if constexpr (!ACC) {
            if (cfg == 1) return pipe_launch<64, WKM, true, ACC, 2, 0>(p, grid, st);
        }
        pipe_launch<32, WKM, true, ACC, 4, 0>(p, grid, st);
    } else {
        if constexpr (!WKM) {                                           // (the [K][N] form of this variant spills)
            if (cfg == 2) return pipe_launch<64, WKM, false, ACC, 3, 16>(p, grid, st);
        }
        if constexpr (WKM && ACC) {                                     // timing experiments on the main shape
            switch (p.dbg) {
                case 1: return pipe_launch<64, WKM, false, ACC, 3, 0, 1>(p, grid, st);
                case 2: return pipe_launch<64, WKM, false, ACC, 3, 0, 2>(p, grid, st);
                case 3: return pipe_launch<64, WKM, false, ACC, 3, 0, 3>(p, grid, st);
                case 4: return pipe_launch<64, WKM, false, ACC, 3, 0, 4>(p, grid, st);
                case 7: return pipe_launch<64, WKM, false, ACC, 3, 0, 7>(p, grid, st);
                case 8: return pipe_launch<64, WKM, false, ACC, 3, 0, 8>(p, grid, st);
                case 16: return pipe_launch<64, WKM, false, ACC, 3, 0, 16>(p, grid, st);
                default: break;
            }
        }
        if (cfg == 1) pipe_launch<32, WKM, false, ACC, 6, 0>(p, grid, st);
        else pipe_launch<64, WKM, false, ACC, 3, 0>(p, grid, st);
    }
}

}  // namespace

bool gemm_pipe_try(const GemmK& g, hipStream_t st) {
    const PoetGemmDesc& d = g.d;
    static const int disabled = [] { const char* e = getenv("POET_GEMM_NO_PIPE"); return e && atoi(e) ? 1 : 0; }();
    if (disabled) return false;
    // plain long-K products: bf16 operands, fp32 result written (+ bias) or accumulated in place
    if (d.a_dtype != POET_BF16 || d.b_dtype != POET_BF16 || d.c_dtype != POET_F32 || d.compute != POET_BF16) return false;
    if (d.a_kmajor || d.batch != 1 || d.splitk != 1 || d.atomic || d.A2) return false;
    if (d.act || d.gate_ref || d.row_mask || d.drop_p != 0.f || d.out_mode != 0 || d.alpha != 1.f) return false;
    if (d.bias && d.add_src) return false;
    if (d.add_src && (d.add_src != d.C || d.ld_add != d.ldc)) return false;
    if (d.b_split && !d.B_lo) return false;                             // (split here = two bf16 images, not an fp32 master)
    if (d.N != PIPE_BN || d.M < 4096 || d.K < 512 || d.K % 64 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B) | reinterpret_cast<uintptr_t>(d.C) | reinterpret_cast<uintptr_t>(d.B_lo)) & 15) return false;
    if ((d.lda & 7) || (d.ldb & 7) || (d.ldc & 3)) return false;
    if (d.lda * 2 * 128 >= (1LL << 31) || d.ldc * 4 * 128 >= (1LL << 31) || (int64_t)d.K * d.ldb * 2 >= (1LL << 31)) return false;   // 32-bit lane offsets
    PipeP p;
    p.A = reinterpret_cast<const bf16_t*>(d.A);
    p.W = reinterpret_cast<const bf16_t*>(d.B);
    p.Wlo = reinterpret_cast<const bf16_t*>(d.B_lo);
    p.C = reinterpret_cast<float*>(d.C);
    p.bias = d.bias;
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.K = d.K;
    { const char* de = getenv("POET_PIPE_DBG"); p.dbg = de ? atoi(de) : 0; }
    { const char* pe = getenv("POET_PIPE_PROF_PTR"); p.prof = pe ? reinterpret_cast<unsigned long long*>(strtoull(pe, nullptr, 0)) : nullptr; }
    const int NU = (d.M + 15) / 16;
    static const int forced = [] { const char* e = getenv("POET_PIPE_GRID"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    int grid = forced ? forced : pipe_cus();
    if (grid > NU) grid = NU;
    const char* ce = getenv("POET_PIPE_CFG");
    const int cfg = ce ? atoi(ce) : 0;
    const bool acc = d.add_src != nullptr, split = d.b_split != 0;
    const int key = (d.b_kmajor ? 4 : 0) | (split ? 2 : 0) | (acc ? 1 : 0);
    switch (key) {
        case 0: pipe_dispatch<false, false, false>(p, grid, cfg, st); break;
        case 1: pipe_dispatch<false, false, true>(p, grid, cfg, st); break;
        case 2: pipe_dispatch<false, true, false>(p, grid, cfg, st); break;
        case 3: pipe_dispatch<false, true, true>(p, grid, cfg, st); break;
        case 4: pipe_dispatch<true, false, false>(p, grid, cfg, st); break;
        case 5: pipe_dispatch<true, false, true>(p, grid, cfg, st); break;
        case 6: pipe_dispatch<true, true, false>(p, grid, cfg, st); break;
        default: pipe_dispatch<true, true, true>(p, grid, cfg, st); break;
    }
    return true;
}

}  // namespace poet
