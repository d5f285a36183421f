// Deep-pipeline tall-skinny GEMM for gfx950:  C[M, 256] (fp32) (+)= A[M, K] (bf16) * W (bf16 [, + W_lo])  (+ bias),
// M ~ 1e5 token rows, K >= 512 a multiple of 64 -- the long-K Linears of the encoder layer seen from backward
// (models/deformable_transformer.py:193-197 forward_ffn: d(src) += d(hidden) W1; :201 MSDeformAttn: d(src) +=
// [d(offsets|logits) | d(value) rows] [W_so ; W_aw ; W_v]; the decoder's d(memory) = d(values) W_v with K = 1280) and the FFN's
// second Linear forward with split-bf16 weights (:185 linear2, hi + lo in ONE pass over the hidden activation).
//
// These products are HBM-bound (A 209 MB + C 104..209 MB against 54 GFLOP at K = 1024).  What shaped the kernel, in the order
// the measurements came in (profiles/probes/pipe_probe.py, dma_probe.hip, dma_probe2.hip; MI355X):
//
//  * persistent: one 768-thread workgroup per CU owns a CONTIGUOUS range of 16-row units (24.9 units per CU at 102 080 rows,
//    dealt 24 / 25 -- no tile quantisation, no stream-K fix-up, results independent of the grid) and cuts it into nearly equal
//    tiles of <= 13 units; the k-loops of its tiles form ONE flat sequence of steps;
//  * operands reach LDS by DMA (global_load_lds_dwordx4: no VGPRs, no ds_write) into TWO rings, A stages (208 x BK) and W stages
//    (BK x 256 [| W_lo]), NSTA - 1 / NSTW - 1 stages in flight ACROSS tile boundaries, so the C read-modify-write of one tile
//    runs under the operand loads of the next; counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per step (the DMA is
//    inline asm: hipcc drains the queue with vmcnt(0) in front of every ds_read when it sees the builtin);
//  * the waves are SPECIALISED: 8 compute waves, 2 that only issue the A DMA, 2 that only issue the W DMA.  A vector-memory
//    instruction blocks its wave until the CU's memory pipeline takes it -- ~150 cycles per DMA instruction under load, measured
//    with s_memtime: in the first version (every wave loads and computes) each wave spent half of every step inside its six
//    DMA instructions while the matrix pipe idled -- and vmcnt completes in issue order per wave, so only separate waves let
//    the two streams run at different depths;
//  * the CU's vector-memory return path is IN ORDER ACROSS WAVES: an L2-hit stream issued next to an HBM-miss stream runs at
//    the HBM stream's pace (dma_probe2: 117 GB/s per CU alone, 20.7 GB/s next to a 22.7 GB/s HBM stream; LDS-DMA and plain loads
//    alike).  Every W byte re-read from L2 therefore costs as much pipeline occupancy as an A byte from HBM: the lever is
//    fewer W passes, i.e. TALLER tiles -- 13 units (2 W passes per CU) instead of 8 (4 passes): 101 -> 93 us, and the split
//    product (two weight images) 140 -> 108 us;
//  * hence 8 compute waves x 32 columns (1 x 8 wave grid, every wave holds ALL rows of the tile: 104 accumulator registers at
//    13 units, 3 waves per SIMD at <= 168 registers; tile heights of 1..13 units cost MFMA time in proportion);
//  * LDS images are plain row-major copies of memory with XOR-swizzled 16-byte chunks (the swizzle is applied to the per-lane
//    SOURCE address of the DMA and to the read address): ds_read_b128 fragments are conflict-free, and a weight stored [K][N]
//    (the input-gradient form) is read with ds_read_b64_tr_b16, the CDNA4 transposing LDS read -- no transposed weight copy;
//  * the product is computed transposed (W fragment as the MFMA's A operand) so that a lane owns 4 consecutive columns of
//    one row: 16-byte C loads / stores, 64 contiguous bytes per row and instruction (all five lane -> address maps tried in
//    ldst_probe.hip stream at 5.9-6.3 TB/s: the pattern is not what limits the epilogue); the accumulators of a tile START as
//    C (plain loads into the accumulator registers, or the bias) and leave by plain stores at its end, while the loader waves
//    keep the next tile's stages coming.
#include "gemm.cuh"

#include <stdlib.h>
#include <type_traits>

namespace poet {

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;

constexpr int PIPE_BN = 256;         // output columns (the whole N)
constexpr int PIPE_MAXF = 13;        // 16-row units per tile (208 rows: the 25 units of a CU at 102 080 rows are 13 + 12)
constexpr int PIPE_NCW = 8;          // compute waves: 32 columns each
constexpr int PIPE_NT = 768;         // threads: 8 compute waves + 2 A-loader waves + 2 W-loader waves (3 waves per SIMD)

struct PipeP {
    const bf16_t* A;
    const bf16_t* W;
    const bf16_t* Wlo;
    float* C;                        // (HOUT kernels: fp16 storage behind the same pointer)
    const float* bias;
    int64_t lda, ldb, ldc;           // elements
    int M, K;
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
    // (M0 is written behind the compiler's back: hipcc rejects "m0" on the clobber list as a reserved register -- "may lead to
    // undefined behaviour" -- so the guarantee is structural instead: nothing else in gemm_pipe_kernel uses M0 (no movrel indexing,
    // no LDS-DMA builtin, no ds_gws / sendmsg); tests/test_abi_cpu.py::test_pipe_kernel_m0_only_in_dma greps the ISA for it)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ void gload16(f32x4_t& dst, uint32_t voff, const void* base) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// OUT = 1 (round 6): C is stored as IEEE fp16 (PoetGemmDesc.c_f16; plain write only) -- the LayerNorm that reads it next takes 2 bytes per
// element instead of 4 on both sides; a lane's 4 consecutive columns leave as one 8-byte store, 32 contiguous bytes per row.
// OUT = 2: C is bfloat16, written or ACCUMULATED in place (the accumulators start as the unpacked bf16 C and leave rounded to nearest
// even): the encoder's bf16 gradient stream -- d(src) += d(hidden) W1 and d(src) += [d(offsets|logits) | d(value)] W at 2 + 2 instead
// of 4 + 4 bytes per element of the read-modify-write.
template <int BK, bool WKM, bool SPLIT, bool ACC, int NSTA, int NSTW, int OUT = 0>
__global__ __launch_bounds__(PIPE_NT, 3) void gemm_pipe_kernel(const PipeP p) {
    constexpr bool HOUT = OUT == 1, BOUT = OUT == 2;
    static_assert(!(HOUT && ACC), "fp16 output: plain write only");
    constexpr int ROWB = BK * 2, CPR = ROWB / 16;
    constexpr int NLT = 128;                                           // threads per loader role (2 waves)
    constexpr int NA = (PIPE_MAXF * 16 * CPR + NLT - 1) / NLT;         // DMA instructions per loader thread and stage
    constexpr int A_BYTES = NA * NLT * 16;                             // (rows past the tile are clamped re-reads into padding)
    constexpr int W_BYTES = BK * 512, WS_BYTES = (SPLIT ? 2 : 1) * W_BYTES, NW = W_BYTES / (NLT * 16);
    constexpr int W_RING = NSTA * A_BYTES;                             // LDS: A ring | W ring
    constexpr int KH = BK / 32, NJ = PIPE_BN / PIPE_NCW / 16;
    static_assert(NA * (NSTA - 2) <= 63 && (SPLIT ? 2 : 1) * NW * (NSTW - 2) <= 63, "vmcnt range");
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
    const int ldaB = (int)p.lda * 2, ldbB = (int)p.ldb * 2;

    if (wave >= PIPE_NCW + 2) {
        // ====== W loader waves: the weight stage of every step, by LDS-DMA, NSTW - 1 stages ahead ======
        constexpr int PD = NSTW - 1, NPS = (SPLIT ? 2 : 1) * NW, LW = NPS * (PD - 1);
        const int lt = tid - (PIPE_NCW + 2) * 64, lw = wave - (PIPE_NCW + 2);
        uint32_t w_off[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int idx = q * NLT + lt;
            if constexpr (WKM) {
                const int krow = idx >> 5, pg = idx & 31, pp = pg >> 1;
                const int lp = (pp & 8) | ((pp ^ swz_km(krow)) & 7);
                w_off[q] = (uint32_t)(krow * ldbB + lp * 32 + (pg & 1) * 16);
            } else {
                const int n = idx / CPR, pc = idx % CPR;
                w_off[q] = (uint32_t)(n * ldbB + (pc ^ swz_row<BK>(n)) * 16);
            }
        }
        int l_kk = 0, l_slot = 0, l_left = S;
        auto issue = [&]() {
            const uint32_t slot = lds0 + W_RING + l_slot * WS_BYTES + lw * 1024;
            {
                const char* wb = reinterpret_cast<const char*>(p.W) + (WKM ? (int64_t)l_kk * BK * ldbB : (int64_t)l_kk * ROWB);
#pragma unroll
                for (int q = 0; q < NW; ++q) dma16(w_off[q], wb, slot + q * (NLT * 16));
                if constexpr (SPLIT) {
                    const char* wl = reinterpret_cast<const char*>(p.Wlo) + (WKM ? (int64_t)l_kk * BK * ldbB : (int64_t)l_kk * ROWB);
#pragma unroll
                    for (int q = 0; q < NW; ++q) dma16(w_off[q], wl, slot + W_BYTES + q * (NLT * 16));
                }
            }
            --l_left;
            l_slot = (l_slot + 1 == NSTW) ? 0 : l_slot + 1;
            if (++l_kk == nk) l_kk = 0;
        };
#pragma unroll 1
        for (int i = 0; i < PD; ++i) issue();                           // (S >= nk >= 8 > PD)
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            // stage s has landed (this wave's share) when at most the loads issued after it are outstanding
            if (min(S, s + PD) - (s + 1) < PD - 1) wait_vm<0>(); else wait_vm<LW>();
            __builtin_amdgcn_s_barrier();                               // every share of stage s landed; slot of stage s - 1 is free
            if (l_left > 0) issue();
        }
        return;
    }
    if (wave >= PIPE_NCW) {
        // ====== A loader waves: the activation stage of every step, NSTA - 1 stages ahead ======
        constexpr int PD = NSTA - 1, LW = NA * (PD - 1);
        const int lt = tid - PIPE_NCW * 64, lw = wave - PIPE_NCW;
        int a_row[NA], a_col[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int idx = q * NLT + lt, row = idx / CPR, pc = idx % CPR;
            a_row[q] = row;
            a_col[q] = (pc ^ swz_row<BK>(row)) * 16;
        }
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
        auto issue = [&]() {
            const uint32_t slot = lds0 + l_slot * A_BYTES + lw * 1024;
            const char* ab = l_abase + l_kk * ROWB;
            {
#pragma unroll
                for (int q = 0; q < NA; ++q) dma16(a_off[q], ab, slot + q * (NLT * 16));
            }
            --l_left;
            l_slot = (l_slot + 1 == NSTA) ? 0 : l_slot + 1;
            if (++l_kk == nk) {
                l_kk = 0;
                if (++l_tile < nt) l_new_tile();
            }
        };
#pragma unroll 1
        for (int i = 0; i < PD; ++i) issue();
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            if (min(S, s + PD) - (s + 1) < PD - 1) wait_vm<0>(); else wait_vm<LW>();
            __builtin_amdgcn_s_barrier();
            if (l_left > 0) issue();
        }
        return;
    }

    // =========================== compute waves: 32 columns each, all rows of the tile ===========================
    // ---- per-lane LDS read offsets ----
    int a_rd[KH];                                                       // fragment row m16, chunk h*4 + kc (swizzled)
#pragma unroll
    for (int h = 0; h < KH; ++h) a_rd[h] = m16 * ROWB + (((h * 4 + kc) & (CPR - 1)) ^ swz_row<BK>(m16)) * 16;
    int w_rd[NJ];                                                       // [K][N] image: transposing read of fragment j
    if constexpr (WKM) {
        const int krl = 8 * kc + (m16 >> 2), sw = (m16 >> 2) | ((kc & 1) << 2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int piece = wave * NJ + j;
            w_rd[j] = krl * 512 + (((piece ^ sw) & 7) | (piece & 8)) * 32 + (m16 & 3) * 8;
        }
    }
    constexpr int CES = OUT ? 2 : 4;                                    // bytes per stored element
    const int ldcB = (int)p.ldc * CES;
    const uint32_t c_col = (uint32_t)((wave * NJ * 16 + kc * 4) * CES);   // + j * 16 * CES bytes

    int a_slot = 0, w_slot = 0;
    auto run_tile = [&](auto nf_tag, int ti) {
        constexpr int NF = decltype(nf_tag)::value;
        const int r0 = tile_u0(ti) * 16, rv = min(NF * 16, p.M - r0);
        char* cb = reinterpret_cast<char*>(p.C) + (int64_t)r0 * ldcB;
        // The accumulators START as C (or the bias): plain loads straight into the registers the MFMAs accumulate in -- no second
        // register set.  Their latency is paid once per tile by the compute waves only (the loader waves keep filling the rings,
        // and the C traffic itself keeps the memory system busy meanwhile); the compute waves have slack against the byte stream.
        f32x4_t acc[NF][NJ];
        // (row offsets are recomputed from an opaque copy of the lane's row at both ends of the tile: hoisted out of the tile
        // loop they would sit in 2 x 13 registers through every k-loop)
        int mo = m16;
        asm volatile("" : "+v"(mo));
        if constexpr (ACC) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const char* cr = cb + (uint32_t)(min(i * 16 + mo, rv - 1) * ldcB) + c_col;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (BOUT) {
                        const uint2 u = *reinterpret_cast<const uint2*>(cr + j * 32);
                        acc[i][j] = f32x4_t{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
                    } else acc[i][j] = *reinterpret_cast<const f32x4_t*>(cr + j * 64);
                }
            }
        } else {
            f32x4_t b4[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b4[j] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + wave * NJ * 16 + j * 16 + kc * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = b4[j];
        }
#pragma unroll 1
        for (int kk = 0; kk < nk; ++kk) {
            __builtin_amdgcn_s_barrier();                               // the stages of this step have landed
            asm volatile("" ::: "memory");
            const char* sl = smem + a_slot * A_BYTES;                   // A stage
            const char* sw = smem + W_RING + w_slot * WS_BYTES;          // W stage (hi | lo)
            a_slot = (a_slot + 1 == NSTA) ? 0 : a_slot + 1;
            w_slot = (w_slot + 1 == NSTW) ? 0 : w_slot + 1;
            {
#pragma unroll
            for (int h = 0; h < KH; ++h) {
                bf16x8_t wf[NJ], wl[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (WKM) {
                        const char* wp = sw + h * 32 * 512 + w_rd[j];
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
                        const char* wp = sw + (wave * NJ * 16 + j * 16) * ROWB + a_rd[h];
                        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp));
                        if constexpr (SPLIT) wl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + W_BYTES));
                    }
                }
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    if constexpr (SPLIT) {                                // (bounds the fragments the scheduler reads ahead: 168 registers)
                        if (i % 4 == 0 && i) __builtin_amdgcn_sched_barrier(0);
                    }
                    const bf16x8_t af = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sl + i * 16 * ROWB + a_rd[h]));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[i][j], 0, 0, 0);
                        if constexpr (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], af, acc[i][j], 0, 0, 0);
                    }
                }
            }
            }
            asm volatile("" ::: "memory");
        }
        // ---- tile done: plain stores straight from the accumulators ----
        asm volatile("" : "+v"(mo));
#pragma unroll
        for (int i = 0; i < NF; ++i)
            if (i * 16 + mo < rv) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (HOUT)
                        *reinterpret_cast<uint2*>(cb + (uint32_t)((i * 16 + mo) * ldcB) + c_col + j * 32) =
                            make_uint2(pack_h2(acc[i][j][0], acc[i][j][1]), pack_h2(acc[i][j][2], acc[i][j][3]));
                    else if constexpr (BOUT)
                        *reinterpret_cast<uint2*>(cb + (uint32_t)((i * 16 + mo) * ldcB) + c_col + j * 32) =
                            make_uint2(pack_bf2(acc[i][j][0], acc[i][j][1]), pack_bf2(acc[i][j][2], acc[i][j][3]));
                    else *reinterpret_cast<f32x4_t*>(cb + (uint32_t)((i * 16 + mo) * ldcB) + c_col + j * 64) = acc[i][j];
                }
            }
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
            case 8: run_tile(std::integral_constant<int, 8>{}, ti); break;
            case 9: run_tile(std::integral_constant<int, 9>{}, ti); break;
            case 10: run_tile(std::integral_constant<int, 10>{}, ti); break;
            case 11: run_tile(std::integral_constant<int, 11>{}, ti); break;
            case 12: run_tile(std::integral_constant<int, 12>{}, ti); break;
            default: run_tile(std::integral_constant<int, 13>{}, ti); break;
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

template <int BK, bool WKM, bool SPLIT, bool ACC, int NSTA, int NSTW, int OUT = 0>
void pipe_launch(const PipeP& p, int grid, hipStream_t st) {
    constexpr int NA = (PIPE_MAXF * 16 * (BK / 8) + 127) / 128;
    constexpr int LDS = NSTA * NA * 2048 + NSTW * (SPLIT ? 2 : 1) * BK * 512;
    static_assert(LDS <= 163840, "LDS rings");
    auto kern = gemm_pipe_kernel<BK, WKM, SPLIT, ACC, NSTA, NSTW, OUT>;
    static unsigned long long attr_done = 0;
    lds_attr_once(reinterpret_cast<const void*>(kern), LDS, attr_done);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PIPE_NT), LDS, st, p);
}

template <bool WKM, bool SPLIT, bool ACC>
void pipe_dispatch(const PipeP& p, int grid, int cfg, hipStream_t st) {
    // cfg (POET_PIPE_CFG, A/B aid): K stage width and ring depths
    if constexpr (SPLIT) {
        if (cfg == 1) pipe_launch<32, WKM, true, ACC, 4, 3>(p, grid, st);      // 56 + 96 KB
        else pipe_launch<32, WKM, true, ACC, 3, 3>(p, grid, st);               // 42 + 96 KB
    } else {
        if (cfg == 1) pipe_launch<64, WKM, false, ACC, 3, 2>(p, grid, st);     // 78 + 64 KB
        else if (cfg == 2) pipe_launch<32, WKM, false, ACC, 6, 4>(p, grid, st);  // 84 + 64 KB
        else if (cfg == 3) pipe_launch<32, WKM, false, ACC, 4, 6>(p, grid, st);  // 56 + 96 KB
        else pipe_launch<32, WKM, false, ACC, 5, 5>(p, grid, st);              // 70 + 80 KB
    }
}

}  // namespace

bool gemm_pipe_try(const GemmK& g, hipStream_t st) {
    const PoetGemmDesc& d = g.d;
    static const int disabled = [] { const char* e = getenv("POET_GEMM_NO_PIPE"); return e && atoi(e) ? 1 : 0; }();
    if (disabled) return false;
    // plain long-K products: bf16 operands, fp32 result written (+ bias) or accumulated in place
    // (round 6: or an fp16 result -- c_dtype POET_BF16 with c_f16 -- of the forward form, plain write)
    const bool hout = d.c_dtype == POET_BF16 && d.c_f16 && !d.b_kmajor && !d.add_src;
    // (or a bf16 result of the input-gradient form, written or accumulated in place: the encoder's bf16 gradient stream)
    const bool bout = d.c_dtype == POET_BF16 && !d.c_f16 && d.b_kmajor && !d.b_split && !d.bias;
    if (d.a_dtype != POET_BF16 || d.b_dtype != POET_BF16 || (d.c_dtype != POET_F32 && !hout && !bout) || d.compute != POET_BF16) return false;
    if (d.a_kmajor || d.batch != 1 || d.splitk != 1 || d.atomic || d.A2) return false;
    if (d.act || d.gate_ref || d.row_mask || d.drop_p != 0.f || d.out_mode != 0 || d.alpha != 1.f) return false;
    if (d.bias && d.add_src) return false;
    if (d.add_src && (d.add_src != d.C || d.ld_add != d.ldc)) return false;
    if (d.b_split && !d.B_lo) return false;                             // (split here = two bf16 images, not an fp32 master)
    if (d.N != PIPE_BN || d.M < 4096 || d.K < 512 || d.K % 64 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B) | reinterpret_cast<uintptr_t>(d.C) | reinterpret_cast<uintptr_t>(d.B_lo)) & 15) return false;
    if ((d.lda & 7) || (d.ldb & 7) || (d.ldc & 3)) return false;
    if (bout && (reinterpret_cast<uintptr_t>(d.C) & 7)) return false;
    if (d.lda * 2 * 128 >= (1LL << 31) || d.ldc * 4 * 128 >= (1LL << 31) || (int64_t)d.K * d.ldb * 2 >= (1LL << 31)) return false;   // 32-bit lane offsets
    PipeP p;
    p.A = reinterpret_cast<const bf16_t*>(d.A);
    p.W = reinterpret_cast<const bf16_t*>(d.B);
    p.Wlo = reinterpret_cast<const bf16_t*>(d.B_lo);
    p.C = reinterpret_cast<float*>(d.C);
    p.bias = d.bias;
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.K = d.K;
    const int NU = (d.M + 15) / 16;
    static const int forced = [] { const char* e = getenv("POET_PIPE_GRID"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    int grid = forced ? forced : pipe_cus();
    if (grid > NU) grid = NU;
    static const int cfg = [] { const char* e = getenv("POET_PIPE_CFG"); return e ? atoi(e) : 0; }();
    const bool acc = d.add_src != nullptr, split = d.b_split != 0;
    const int key = (d.b_kmajor ? 4 : 0) | (split ? 2 : 0) | (acc ? 1 : 0);
    if (hout) {                                                         // (the default ring depths; POET_PIPE_CFG applies to the fp32 forms)
        if (split) pipe_launch<32, false, true, false, 3, 3, 1>(p, grid, st);
        else pipe_launch<32, false, false, false, 5, 5, 1>(p, grid, st);
        return true;
    }
    if (bout) {                                                         // (POET_PIPE_CFG 1-3: the ring shapes of pipe_dispatch, A/B aid)
        if (acc) {
            if (cfg == 1) pipe_launch<64, true, false, true, 3, 2, 2>(p, grid, st);
            else if (cfg == 2) pipe_launch<32, true, false, true, 6, 4, 2>(p, grid, st);
            else if (cfg == 3) pipe_launch<32, true, false, true, 4, 6, 2>(p, grid, st);
            else pipe_launch<32, true, false, true, 5, 5, 2>(p, grid, st);
        } else pipe_launch<32, true, false, false, 5, 5, 2>(p, grid, st);
        return true;
    }
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
