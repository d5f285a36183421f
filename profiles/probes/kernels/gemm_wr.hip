// Register-stationary streaming GEMM for gfx950:  C[M, N] (bf16 | fp16) = epilogue(A[M, 256] (bf16) * W[N, 256]^T (fp32 master,
// used as bf16 hi + bf16 lo)),  M ~ 1e5 token rows, N a multiple of 256: the WIDE forward Linears of the encoder layer with a
// 256-wide input -- the FFN's first Linear (256 -> 1024, ReLU + dropout: models/deformable_transformer.py:184 linear1) and
// MSDeformAttn's stacked sampling_offsets | attention_weights projection (256 -> 768: :201).
//
// Why another kernel.  The weight-stationary kernel of gemm_ws.hip keeps a 128-column slice of W in LDS (two images: 128 KB of
// the 160) and every wave loads its activation rows itself, computes, and stores: measured its
// three phases ADD UP -- 51 us for the bare load / store pattern, 66 us without the stores, 100-125 us complete -- because a
// wave's loads queue behind its stores in the CU's in-order vector-memory path and nothing else runs meanwhile; half the
// matrix instructions (single weight image) buy only 18-22 %.  Here the roles of the two memories are swapped:
//
//  * the WEIGHT lives in REGISTERS: 8 compute waves x 32 columns = a 256-column slice per workgroup; a wave keeps its 32 x 256
//    weights as MFMA A-operand fragments, hi and lo image, 2 x 64 = 128 VGPRs, loaded and split ONCE (fp32 master -> bf16 pair);
//    no weight byte is read again, from LDS or anywhere, for the workgroup's whole life;
//  * the ACTIVATION streams through LDS: a ring of WR_R 16-row units (8 KB each, 128 KB: ~4 us of stream in flight per CU) filled
//    by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), every wave issuing ONE 1 KB piece per unit, counted
//    s_waitcnt vmcnt(N) + one raw s_barrier per unit; the image is a plain copy of memory with the 16-byte chunks of a row
//    XOR-swizzled by the row (on the DMA's per-lane SOURCE address and on the read address): conflict-free ds_read_b128 under
//    the hardware's lane groups; every unit is read by all 8 waves (64 KB of LDS reads per unit and CU: 12 % of the port);
//  * per unit a wave issues 8 ds_read_b128, 32 MFMAs (2 column fragments x 8 k-steps x 2 images, same accumulation order as
//    gemm_ws: results are bit-identical) and ONE 16-byte store per lane: the product is computed transposed and the weight rows
//    are permuted across the two fragments so that a lane owns 8 consecutive columns of one row;
//  * the column slices of the same rows run on the same XCD (workgroup b -> XCD b % 8): A comes from HBM once and from that
//    XCD's L2 for the other slices;
//  * one barrier per STEP of two units; the two waves of a SIMD run half a step apart (see the loop).
//
// MEASURED (round 4, profiles/probes/ws_split_bench.py, 102 080 rows): 256 -> 1024 with ReLU + dropout 136 us, 256 -> 768 90 us --
// against 136 / 85 us for gemm_ws.  Three forms of this kernel (one barrier per unit, the two waves of a SIMD in phase; half a unit
// apart; two-unit steps) and gemm_ws all land within 5 % of each other, and with the activation DMA, the MFMAs AND the stores
// compiled out the loop still takes 71 us (95 us with the dropout hash): what is left then -- ~100 instructions per unit and wave at
// one issue per ~4-5 cycles per wave, the LDS round trip behind the barrier, the barrier itself -- is the floor of a two-waves-per-
// SIMD kernel whose per-output epilogue is 12-16 instructions.  SQ counters (profiles/probes/wr_pmc.sh): waves active 22 %, issue-
// stalled 41 %, parked 36 %, matrix pipe 51 % busy.  Bit-identical to gemm_ws (same accumulation order); kept OPT-IN as the
// measurement.
#include "gemm.cuh"

#include <stdlib.h>

namespace poet {

namespace {

constexpr int WR_NW = 8, WR_NT = WR_NW * 64;     // compute waves (= loader waves: one DMA piece each per unit)
constexpr int WR_BN = 256;                        // columns per workgroup: 32 per wave
constexpr int WR_R = 16;                          // ring depth in 16-row units
constexpr int WR_SU = 2, WR_RS = WR_R / WR_SU;    // units per step (one barrier per step), ring depth in steps
constexpr int WR_UNIT = 16 * 512;                 // bytes of one unit: 16 rows x 256 bf16
constexpr int WR_LDS = WR_R * WR_UNIT;

struct WrP {
    const bf16_t* A;
    const float* W;
    void* C;
    const float* bias;
    int64_t lda, ldb, ldc;                        // elements
    int M, N;
    int act;
    uint32_t drop_thresh;
    float drop_scale;
    uint32_t seed;
    const uint32_t* seed_dev;
};

// one LDS-DMA instruction: 64 lanes x 16 bytes from `base + voff[lane]` to LDS bytes [lds, lds + 1024) in lane order
// (M0 is written behind the compiler's back, as in gemm_pipe.hip: nothing else in this kernel uses M0)
__device__ __forceinline__ void wr_dma16(uint32_t voff, const void* base, uint32_t lds) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename TC>
__global__ __launch_bounds__(WR_NT, 2) void gemm_wr_kernel(const WrP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kc = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- work assignment: (column slice, row chunk); the slices of a chunk share an XCD (b % 8) ----
    const int NS = p.N / WR_BN, b = blockIdx.x;
    const int grp = b / (8 * NS), inb = b - grp * (8 * NS);
    const int slice = inb >> 3, chunk = grp * 8 + (inb & 7);
    const int nchunks = (gridDim.x / (8 * NS)) * 8;
    if (chunk >= nchunks) return;                                       // (workgroups beyond the last whole group of 8 x NS)
    const int NU = (p.M + 15) >> 4;
    const int u0 = (int)((int64_t)chunk * NU / nchunks), u1 = (int)((int64_t)(chunk + 1) * NU / nchunks);
    const int nu = u1 - u0;
    if (nu <= 0) return;

    // ---- activation DMA: a STEP = WR_SU consecutive units; this wave's pieces = rows 2 wave, 2 wave + 1 of each of them
    // (lane -> row, swizzled 16-byte chunk; 32-bit byte offsets from A, rows clamped to the matrix) ----
    const int ldaB = (int)p.lda * 2;
    const int drow = 2 * wave + (lane >> 5), dchunk = (lane & 31) ^ drow;             // (drow < 16: the swizzle key is the row itself)
    const int nsteps = (nu + WR_SU - 1) / WR_SU;
    auto issue = [&](int st) {                                          // step st -> ring slots (st % WR_RS) * WR_SU ...
#pragma unroll
        for (int k = 0; k < WR_SU; ++k) {
            const int row = min((u0 + st * WR_SU + k) * 16 + drow, p.M - 1);
            wr_dma16((uint32_t)(row * ldaB + dchunk * 16), p.A, lds0 + (uint32_t)(((st % WR_RS) * WR_SU + k) * WR_UNIT + wave * 1024));
        }
    };
    const int npre = min(WR_RS - 1, nsteps);
#pragma unroll 1
    for (int st = 0; st < npre; ++st) issue(st);

    // ---- this wave's 32 weight rows -> registers, split into hi | lo bf16 images (once) ----
    // fragment f, MFMA row i = 4 g + t  <->  column 8 g + 4 f + t of the wave's 32: lane group kc then owns columns [8 kc, 8 kc + 8)
    const int col_w = slice * WR_BN + wave * 32;
    bf16x8_t whi[2][8], wlo[2][8];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float* wr = p.W + (int64_t)(col_w + 8 * (m16 >> 2) + 4 * f + (m16 & 3)) * p.ldb + kc * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 a = *reinterpret_cast<const float4*>(wr + kk * 32), c = *reinterpret_cast<const float4*>(wr + kk * 32 + 4);
            const uint4 hi = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(c.x, c.y), pack_bf2(c.z, c.w));
            const uint4 lo = make_uint4(pack_bf2(a.x - __uint_as_float(hi.x << 16), a.y - __uint_as_float(hi.x & 0xffff0000u)),
                                        pack_bf2(a.z - __uint_as_float(hi.y << 16), a.w - __uint_as_float(hi.y & 0xffff0000u)),
                                        pack_bf2(c.x - __uint_as_float(hi.z << 16), c.y - __uint_as_float(hi.z & 0xffff0000u)),
                                        pack_bf2(c.z - __uint_as_float(hi.w << 16), c.w - __uint_as_float(hi.w & 0xffff0000u)));
            whi[f][kk] = __builtin_bit_cast(bf16x8_t, hi);
            wlo[f][kk] = __builtin_bit_cast(bf16x8_t, lo);
        }
    }
    const int col0 = col_w + 8 * kc;                                    // this lane's 8 output columns
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = p.bias ? p.bias[col0 + e] : 0.f;
    const uint32_t sd = p.seed ^ (p.seed_dev ? *p.seed_dev * 0x9E3779B1u : 0u);
    // (the weight / bias loads above are ordinary loads: drained here, so that the counted waits below see DMA pieces and stores only)
    wr_wait_vm<0>();
    // re-issue accounting: the prologue's pieces are complete now (vmcnt(0) waited for them as well)

    // fragment read offsets: row m16, chunk kk * 4 + kc, swizzled by the row
    int a_rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_rd[j] = m16 * 512 + (((j * 4 + kc) ^ m16) & 15) * 16;
    TC* Cb = reinterpret_cast<TC*>(p.C);

    // epilogue on registers: row m16 of unit u, columns [col0, col0 + 8)
    auto epilogue = [&](const f32x4_t& a0, const f32x4_t& a1, int u) __attribute__((always_inline)) {
        const int grow = (u0 + u) * 16 + m16;
        float v[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[t] = a0[t] + bias8[t]; v[4 + t] = a1[t] + bias8[4 + t]; }
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu1(v[e]);
        }
        if (p.drop_thresh) {
            const uint32_t base = (uint32_t)grow * (uint32_t)p.N + (uint32_t)col0;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {                            // base is even: (e, e + 1) share one hash (gemm_ws.hip's mask)
                const uint32_t hsh = drop_pair(sd, (base + e) >> 1);
                v[e] = (hsh & 0xffffu) >= p.drop_thresh ? v[e] * p.drop_scale : 0.f;
                v[e + 1] = (hsh >> 16) >= p.drop_thresh ? v[e + 1] * p.drop_scale : 0.f;
            }
        }
        if (u < nu && grow < p.M) vec<TC, 8>::st(Cb + (int64_t)grow * p.ldc + col0, v);       // (u >= nu: the odd tail of the last step)
    };

    // The two waves of a SIMD (w and w + 4: a workgroup's waves are dealt to the SIMDs cyclically) run HALF A STEP APART: the barrier
    // that publishes a step releases all eight together, and if they all ran "MFMAs, then epilogue" both waves of a SIMD would want
    // the matrix pipe at the same time and the VALU at the same time.  So the late half (waves 4-7) runs the epilogue of step s - 1
    // BEFORE the MFMAs of step s: while one wave of a SIMD feeds the matrix pipe, its partner issues the epilogue's VALU work and stores.
    const bool late = wave >= 4;
    f32x4_t pv[WR_SU][2];
#pragma unroll 1
    for (int st = 0; st < nsteps; ++st) {
        // step st has landed (this wave's pieces) when at most 4 (WR_RS - 5) younger operations are outstanding: after its pieces a wave
        // issued WR_SU stores, then WR_SU pieces + WR_SU stores per step -- the wait below is stricter than that count in every phase of
        // the loop (head, steady state, tail) and leaves 3 steps of stream in flight
        wr_wait_vm<2 * WR_SU * (WR_RS - 5)>();
        __builtin_amdgcn_s_barrier();                                   // every piece of step st landed; step st - 1 is consumed by all waves
        asm volatile("" ::: "memory");
        if (st + WR_RS - 1 < nsteps) issue(st + WR_RS - 1);             // into the slots of step st - 1
        if (late && st > 0) {
#pragma unroll
            for (int k = 0; k < WR_SU; ++k) epilogue(pv[k][0], pv[k][1], (st - 1) * WR_SU + k);
        }
        const char* sl = smem + (st % WR_RS) * (WR_SU * WR_UNIT);
        bf16x8_t af[WR_SU][8];
#pragma unroll
        for (int k = 0; k < WR_SU; ++k)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                af[k][kk] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sl + k * WR_UNIT + a_rd[kk & 3] + (kk >> 2) * 256));
        f32x4_t acc[WR_SU][2];
#pragma unroll
        for (int k = 0; k < WR_SU; ++k) {
            acc[k][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[k][1] = acc[k][0];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    acc[k][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi[f][kk], af[k][kk], acc[k][f], 0, 0, 0);
                    acc[k][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[f][kk], af[k][kk], acc[k][f], 0, 0, 0);
                }
        }
        asm volatile("" ::: "memory");
        if (late) {
#pragma unroll
            for (int k = 0; k < WR_SU; ++k) { pv[k][0] = acc[k][0]; pv[k][1] = acc[k][1]; }
        } else {
#pragma unroll
            for (int k = 0; k < WR_SU; ++k) epilogue(acc[k][0], acc[k][1], st * WR_SU + k);
        }
    }
    if (late) {
#pragma unroll
        for (int k = 0; k < WR_SU; ++k) epilogue(pv[k][0], pv[k][1], (nsteps - 1) * WR_SU + k);
    }
}

int wr_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <typename TC>
void wr_launch(const WrP& p, hipStream_t st) {
    auto kern = gemm_wr_kernel<TC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS);
        attr_set = true;
    }
    const int NS = p.N / WR_BN;
    int grid = (wr_cus() / (8 * NS)) * (8 * NS);                         // whole groups of 8 x NS workgroups: one per CU
    if (grid <= 0) grid = 8 * NS;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WR_NT), WR_LDS, st, p);
}

}  // namespace

// the wide split-weight forward products with a 256-wide input; false = not this kernel's problem (the caller goes on to gemm_ws)
bool gemm_wr_try(const GemmK& g, hipStream_t st) {
    const PoetGemmDesc& d = g.d;
    // OPT-IN (POET_GEMM_WR=1): measured equal to the LDS-stationary kernel of gemm_ws.hip at both shapes, so
    // the default stays gemm_ws; read per call so that a test can switch it
    { const char* e = getenv("POET_GEMM_WR"); if (!(e && atoi(e))) return false; }
    if (!d.b_split || d.B_lo || d.K != 256 || d.N % WR_BN != 0 || d.N < 512 || d.M < 4096) return false;
    if (d.compute != POET_BF16 || d.a_dtype != POET_BF16 || d.b_dtype != POET_F32 || d.c_dtype != POET_BF16) return false;
    if (d.a_kmajor || d.b_kmajor || d.batch != 1 || d.splitk != 1 || d.atomic || d.A2) return false;
    if (d.add_src || d.gate_ref || d.row_mask || d.out_mode != 0 || d.alpha != 1.f) return false;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B) | reinterpret_cast<uintptr_t>(d.C)) & 15) return false;
    if ((d.lda & 7) || (d.ldb & 3) || (d.ldc & 7)) return false;
    if ((int64_t)d.M * d.lda * 2 >= (1LL << 31)) return false;           // 32-bit byte offsets into A
    WrP p;
    p.A = reinterpret_cast<const bf16_t*>(d.A);
    p.W = reinterpret_cast<const float*>(d.B);
    p.C = d.C;
    p.bias = d.bias;
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
    p.M = d.M; p.N = d.N;
    p.act = d.act;
    p.drop_thresh = g.drop_thresh; p.drop_scale = g.drop_scale;
    p.seed = d.seed; p.seed_dev = d.seed_dev;
    if (d.c_f16) wr_launch<f16_t>(p, st);
    else wr_launch<bf16_t>(p, st);
    return true;
}

}  // namespace poet
