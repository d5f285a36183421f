// Weight-gradient GEMM for gfx950:  dW[n1, n2] += sum_r dY[r, n1] * X[r, n2],  r over ~1e5 token rows, n1/n2 in
// {256 .. 1024} -- the backward of every encoder nn.Linear (reference: autograd of the Linear layers in
// models/deformable_transformer.py:215-240 and of MSDeformAttn's projections).  Both operands are stored row-major with
// the REDUCTION index r as the slow dimension, the worst case for an MFMA (a lane needs 8 consecutive r of one column).
//
//  * a workgroup owns one 128x128 tile of dW and one contiguous range of rows; it streams 64-row stages of the two
//    operand panels (64 x 128 bf16 each) through a double-buffered LDS image that is a plain row-major copy of memory
//    (coalesced 16-byte global loads, 16-byte LDS stores) with the 32-byte pieces of a row XOR-swizzled by the row index;
//  * fragments come out of LDS with ds_read_b64_tr_b16 (CDNA4 transpose read): 16 lanes pointing at a 4 (r) x 16
//    (column) patch get, per lane, 4 consecutive r of ONE column -- two of them are an MFMA operand, no register shuffles,
//    and the swizzle makes every 16/32-lane phase of the read hit distinct banks;
//  * few row ranges (16..64, chosen so ~1.5 workgroups per CU exist): the fp32 atomics that merge the partial tiles are
//    the expensive part (they execute memory-side), so their count S*n1*n2 is kept small instead of the usual
//    "1024 small split-K workgroups";
//  * the tiles of one row range run on the same XCD (workgroup b -> XCD b % 8), so each panel is fetched from HBM once
//    and re-read by the other tiles of that range through the XCD's L2.
#include "gemm.cuh"

namespace poet {

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;

constexpr int DW_T = 128;            // tile edge (both n1 and n2)
constexpr int DW_RS = 64;            // rows per stage
constexpr int DW_ROWB = DW_T * 2;    // bytes per LDS row (no padding: swizzled)
constexpr int DW_PANEL = DW_RS * DW_ROWB;          // 16 KB
constexpr int DW_STAGE = 2 * DW_PANEL;             // Y panel | X panel

__device__ __forceinline__ int dw_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

struct DwP {
    const bf16_t* Y;
    const bf16_t* X;
    float* C;
    float* ysum;          // optional [n1]: += column sums of Y (the bias gradient), accumulated by the n2-tile-0 workgroups
    int64_t ldy, ldx, ldc;
    int rows, n1, n2, ntiles, tiles_n2, splits;
    float* ws;            // optional workspace [splits][ntiles][128 x 128]: partial tiles leave by plain stores, dw_reduce_kernel merges
};

// LEAN (default): the stage loop over WHOLE 64-row stages without a single predicate (a ragged last stage runs the generic code once) --
// see the block comment in front of the lean loop below.
template <bool LEAN>
__global__ __launch_bounds__(256, 2) void gemm_dw_kernel(const DwP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- work assignment: (tile, row range); the tiles of a range share an XCD ----
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % p.ntiles, split = (j / p.ntiles) * 8 + xcd;
    if (split >= p.splits) return;
    const int n1_0 = (tile / p.tiles_n2) * DW_T, n2_0 = (tile % p.tiles_n2) * DW_T;
    const int nst = (p.rows + DW_RS - 1) / DW_RS;                        // stages in the whole problem
    const int s_lo = (int)((int64_t)split * nst / p.splits), s_hi = (int)((int64_t)(split + 1) * nst / p.splits);
    if (s_lo >= s_hi) return;

    // ---- global -> register staging: 4 x 16 B of each panel per thread per stage ----
    const bf16_t* Yp = p.Y + n1_0;
    const bf16_t* Xp = p.X + n2_0;
    // One stage lives in registers on top of the two in LDS (a second register stage measured no faster: with two
    // workgroups per CU ~64 KB are in flight per CU, above the ~47 KB Little's law asks for at full HBM rate).
    uint4 ry[1][4], rx[1][4];
    // Per-thread constants hoisted out of the stage loop (the staging code used to cost more issue slots than the MFMAs):
    // chunk c = tid & 15 of row (tid >> 4) + 16 i; 32-bit element offsets (rows * ld < 2^31 is checked on the host).
    const int c16 = tid & 15, rt = tid >> 4;
    int yo[4], xo[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rt + 16 * i;
        yo[i] = row * (int)p.ldy + c16 * 8;
        xo[i] = row * (int)p.ldx + c16 * 8;
        lo[i] = row * DW_ROWB + ((((c16 >> 1) ^ dw_swz(row)) << 5) | ((c16 & 1) << 4));
    }
    const int ystep = DW_RS * (int)p.ldy, xstep = DW_RS * (int)p.ldx;
    const int ytail = (p.rows - 1) * (int)p.ldy + c16 * 8, xtail = (p.rows - 1) * (int)p.ldx + c16 * 8;
    // loads are UNCONDITIONAL (row clamped, zero selected at the LDS store): with a branch around them the compiler cannot
    // count outstanding loads and every LDS store would drain the queue, collapsing the prefetch depth
    auto load = [&](int s, uint4 (&y)[4], uint4 (&x)[4]) {
        const int r0 = s * DW_RS + rt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = r0 + 16 * i < p.rows;
            y[i] = *reinterpret_cast<const uint4*>(Yp + (in ? s * ystep + yo[i] : ytail));
            x[i] = *reinterpret_cast<const uint4*>(Xp + (in ? s * xstep + xo[i] : xtail));
        }
    };
    const bool do_sum = p.ysum != nullptr && n2_0 == 0;                 // workgroup-uniform
    float ys[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};            // this thread's 8 columns (chunk tid & 15) of Y
    auto store = [&](char* buf, int s, const uint4 (&y)[4], const uint4 (&x)[4]) {
        const bool whole = s < s_hi && (s + 1) * DW_RS <= p.rows;       // workgroup-uniform: no masking on interior stages
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 ym = y[i], xm = x[i];
            if (!whole) {                                               // value selects, never pointer selects
                const uint32_t m = (s < s_hi && s * DW_RS + rt + 16 * i < p.rows) ? 0xffffffffu : 0u;
                ym = make_uint4(ym.x & m, ym.y & m, ym.z & m, ym.w & m);
                xm = make_uint4(xm.x & m, xm.y & m, xm.z & m, xm.w & m);
            }
            if (do_sum) {
                ys[0] += __uint_as_float(ym.x << 16); ys[1] += __uint_as_float(ym.x & 0xffff0000u);
                ys[2] += __uint_as_float(ym.y << 16); ys[3] += __uint_as_float(ym.y & 0xffff0000u);
                ys[4] += __uint_as_float(ym.z << 16); ys[5] += __uint_as_float(ym.z & 0xffff0000u);
                ys[6] += __uint_as_float(ym.w << 16); ys[7] += __uint_as_float(ym.w & 0xffff0000u);
            }
            *reinterpret_cast<uint4*>(buf + lo[i]) = ym;
            *reinterpret_cast<uint4*>(buf + DW_PANEL + lo[i]) = xm;
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // per-lane constants of the transpose read: source lane r of a 16-lane block points at row 8b + (r>>2) (+4 for the
    // second half), 8-byte chunk r&3 of the fragment's 32-byte piece
    const int r16 = lane & 15, b4 = lane >> 4;
    const int trow = 8 * b4 + (r16 >> 2), tswz = (r16 >> 2) | ((b4 & 1) << 2), tcol = (r16 & 3) * 8;

    auto compute = [&](const char* yb) {
        const char* xb = yb + DW_PANEL;
#pragma unroll
        for (int ks = 0; ks < DW_RS / 32; ++ks) {
            bf16x8_t fy[4], fx[4];
            const int row0 = ks * 32 + trow;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int py = (((wm * 4 + f) ^ tswz) << 5) + tcol, px = (((wn * 4 + f) ^ tswz) << 5) + tcol;
                struct { v4s_t lo, hi; } u, v;
                u.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(yb + row0 * DW_ROWB + py));
                u.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(yb + (row0 + 4) * DW_ROWB + py));
                v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(xb + row0 * DW_ROWB + px));
                v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(xb + (row0 + 4) * DW_ROWB + px));
                fy[f] = __builtin_bit_cast(bf16x8_t, u);
                fx[f] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[jj], acc[i][jj], 0, 0, 0);
        }
    };

    if constexpr (LEAN) {
        // The generic loop below issues ~280 instructions per stage and wave around its 32 MFMAs (round-4 disassembly: 118 VALU -- row
        // clamps and value selects of the ragged-edge handling, 64-bit address arithmetic, LDS addresses recomputed per read -- 45
        // s_waitcnt, 15 branches), and with two 4-wave workgroups per CU a wave is ONE instruction stream that issues every ~4-5
        // cycles: the kernel sat on its issue rate, not on memory (matrix pipe 28 % busy).  With rows % 64 == 0 no stage is ragged, so:
        // loads are SGPR base (advanced per stage) + a hoisted 32-bit lane offset; LDS stores and the 32 transposing reads are a
        // hoisted lane address + an immediate (the loop is unrolled by two so that the buffer index is a compile-time constant); the
        // prefetch past the last stage re-reads the last stage (uniform clamp) instead of being predicated.
        const char* Yb = reinterpret_cast<const char*>(Yp);
        const char* Xb = reinterpret_cast<const char*>(Xp);
        const int64_t ystB = (int64_t)DW_RS * p.ldy * 2, xstB = (int64_t)DW_RS * p.ldx * 2;
        uint32_t yoB[4], xoB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { yoB[i] = (uint32_t)yo[i] * 2u; xoB[i] = (uint32_t)xo[i] * 2u; }
        auto loadL = [&](int s, uint4 (&y)[4], uint4 (&x)[4]) __attribute__((always_inline)) {
            const int sc = min(s, p.rows / DW_RS - 1);                   // (uniform: the last WHOLE stage)
            const char* yb = Yb + sc * ystB;
            const char* xb = Xb + sc * xstB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[i] = *reinterpret_cast<const uint4*>(yb + yoB[i]);
                x[i] = *reinterpret_cast<const uint4*>(xb + xoB[i]);
            }
        };
        auto storeL = [&](char* buf, int st, const uint4 (&y)[4], const uint4 (&x)[4]) __attribute__((always_inline)) {
            const bool sum = do_sum && st < min(s_hi, p.rows / DW_RS);    // (uniform; a stage past this workgroup's whole stages is staged but never used)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (sum) {
                    ys[0] += __uint_as_float(y[i].x << 16); ys[1] += __uint_as_float(y[i].x & 0xffff0000u);
                    ys[2] += __uint_as_float(y[i].y << 16); ys[3] += __uint_as_float(y[i].y & 0xffff0000u);
                    ys[4] += __uint_as_float(y[i].z << 16); ys[5] += __uint_as_float(y[i].z & 0xffff0000u);
                    ys[6] += __uint_as_float(y[i].w << 16); ys[7] += __uint_as_float(y[i].w & 0xffff0000u);
                }
                *reinterpret_cast<uint4*>(buf + lo[i]) = make_uint4(y[i].x, y[i].y, y[i].z, y[i].w);
                *reinterpret_cast<uint4*>(buf + DW_PANEL + lo[i]) = make_uint4(x[i].x, x[i].y, x[i].z, x[i].w);
            }
        };
        // transposing-read lane addresses: [fragment] for the Y panel (this wave's 64 rows of the tile) and the X panel
        int ty[4], tx[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            ty[f] = trow * DW_ROWB + ((((wm * 4 + f) ^ tswz) << 5) + tcol);
            tx[f] = trow * DW_ROWB + ((((wn * 4 + f) ^ tswz) << 5) + tcol) + DW_PANEL;
        }
        auto computeL = [&](const char* sb) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < DW_RS / 32; ++ks) {
                bf16x8_t fy[4], fx[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    struct { v4s_t lo, hi; } u, v;
                    u.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + ty[f] + ks * 32 * DW_ROWB));
                    u.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + ty[f] + (ks * 32 + 4) * DW_ROWB));
                    v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + tx[f] + ks * 32 * DW_ROWB));
                    v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + tx[f] + (ks * 32 + 4) * DW_ROWB));
                    fy[f] = __builtin_bit_cast(bf16x8_t, u);
                    fx[f] = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[jj], acc[i][jj], 0, 0, 0);
            }
        };
        // A ragged LAST stage of the whole problem (rows % 64 != 0: 1280 x 960 has 204 000 token rows) belongs to exactly one row range:
        // that workgroup runs the lean loop over its whole stages and the generic, masked stage code once at the end.
        const int nfull = p.rows / DW_RS;                                 // whole stages of the problem
        const int e_hi = min(s_hi, nfull);                               // this workgroup's whole stages: [s_lo, e_hi)
        if (s_lo < e_hi) {
            loadL(s_lo, ry[0], rx[0]);
            storeL(smem, s_lo, ry[0], rx[0]);
            loadL(s_lo + 1, ry[0], rx[0]);
            __syncthreads();
            int s = s_lo;
#pragma unroll 1
            for (; s + 1 < e_hi; s += 2) {                               // stage s in buffer 0, stage s + 1 in the registers
                storeL(smem + DW_STAGE, s + 1, ry[0], rx[0]);
                loadL(s + 2, ry[0], rx[0]);
                __builtin_amdgcn_sched_barrier(0);
                computeL(smem);
                __syncthreads();
                storeL(smem, s + 2, ry[0], rx[0]);                        // (stage s + 2: computed by the next trip, or never read)
                loadL(s + 3, ry[0], rx[0]);
                __builtin_amdgcn_sched_barrier(0);
                computeL(smem + DW_STAGE);
                __syncthreads();
            }
            if (s < e_hi) {                                              // odd number of stages: the last one sits in buffer 0
                computeL(smem);
                __syncthreads();
            }
        }
        if (e_hi < s_hi) {                                               // the ragged stage (workgroup-uniform; its readers of smem are past a barrier)
            load(e_hi, ry[0], rx[0]);
            store(smem, e_hi, ry[0], rx[0]);
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    } else {
    // prologue: stage s_lo -> LDS buffer 0; stage s_lo+1 -> registers
    load(s_lo, ry[0], rx[0]);
    store(smem, s_lo, ry[0], rx[0]);
    load(s_lo + 1, ry[0], rx[0]);
    __syncthreads();

    // iteration s: LDS buffer `buf` holds stage s, the registers hold s+1.  Park them in the other LDS buffer (its
    // readers passed the barrier that ended iteration s-1), refill with s+2, compute s, barrier.
    int buf = 0;
    for (int s = s_lo; s < s_hi; ++s) {
        store(smem + (buf ^ 1) * DW_STAGE, s + 1, ry[0], rx[0]);
        load(s + 2, ry[0], rx[0]);
        __builtin_amdgcn_sched_barrier(0);                  // the refill goes out BEFORE the MFMA block (the scheduler sinks it)
        compute(smem + buf * DW_STAGE);
        __syncthreads();
        buf ^= 1;
    }
    }

    // ---- bias gradient: 16 threads (tid >> 4) hold partial sums of the same 8 columns; fold them through LDS ----
    if (do_sum) {                                                        // (the last barrier of the loop freed smem)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 4) * DW_T + (tid & 15) * 8 + e] = ys[e];
        __syncthreads();
        if (tid < DW_T) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * DW_T + tid];
            atomicAdd(p.ysum + n1_0 + tid, t);
        }
    }

    // ---- merge the partial tile: lane holds rows 4*b4 + t, column r16 of each 16x16 fragment ----
    if (p.ws) {
        // plain, fully coalesced stores of the register image (wave, fragment, t, lane); dw_reduce_kernel sums the row ranges.
        // (fp32 atomics execute memory-side at ~0.4 T/s: 4-8 M of them cost more than the MFMA loop of the 256-wide shapes)
        float* wp = p.ws + ((int64_t)split * p.ntiles + tile) * (DW_T * DW_T) + wid * 4096 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int t = 0; t < 4; ++t) wp[((i * 4 + jj) * 4 + t) * 64] = acc[i][jj][t];
        return;
    }
    float* Cp = p.C + (int64_t)(n1_0 + wm * 64) * p.ldc + n2_0 + wn * 64 + r16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                atomicAdd(Cp + (int64_t)(i * 16 + b4 * 4 + t) * p.ldc + jj * 16, acc[i][jj][t]);
}

// dW[tile] += sum over row ranges of the partial tiles (single owner per element: a plain read-modify-write)
__global__ __launch_bounds__(256) void dw_reduce_kernel(const DwP p, int nsplit_used) {
    const int tile = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;               // e: index inside the 128 x 128 register image
    const float* wp = p.ws + (int64_t)tile * (DW_T * DW_T) + e;
    const int64_t stride = (int64_t)p.ntiles * (DW_T * DW_T);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nsplit_used; k += 4) {
        s0 += wp[(int64_t)k * stride]; s1 += wp[(int64_t)(k + 1) * stride];
        s2 += wp[(int64_t)(k + 2) * stride]; s3 += wp[(int64_t)(k + 3) * stride];
    }
    for (; k < nsplit_used; ++k) s0 += wp[(int64_t)k * stride];
    const int lane = e & 63, t = (e >> 6) & 3, fr = (e >> 8) & 15, wid = e >> 12;
    const int i = fr >> 2, jj = fr & 3, wm = wid >> 1, wn = wid & 1, r16 = lane & 15, b4 = lane >> 4;
    const int n1_0 = (tile / p.tiles_n2) * DW_T, n2_0 = (tile % p.tiles_n2) * DW_T;
    float* c = p.C + (int64_t)(n1_0 + wm * 64 + i * 16 + b4 * 4 + t) * p.ldc + n2_0 + wn * 64 + jj * 16 + r16;
    *c += (s0 + s1) + (s2 + s3);
}

}  // namespace

bool gemm_dw_try(const GemmK& g, hipStream_t st) {
    const PoetGemmDesc& d = g.d;
    static const int disabled = [] { const char* e = getenv("POET_GEMM_NO_DW"); return e && atoi(e) ? 1 : 0; }();
    if (disabled) return false;
    // the dW form of poet_gemm: A = dY stored [K = rows][M = n1], B = X stored [K = rows][N = n2], fp32 atomic output
    if (!d.a_kmajor || !d.b_kmajor || !(d.atomic || d.splitk > 1) || d.batch != 1 || d.A2) return false;
    if (d.a_dtype != POET_BF16 || d.b_dtype != POET_BF16 || d.c_dtype != POET_F32 || d.compute != POET_BF16) return false;
    if (d.M % DW_T != 0 || d.N % DW_T != 0 || d.K < 4096 || d.alpha != 1.f) return false;
    if (!g.a_vec || !g.b_vec) return false;
    if ((int64_t)d.K * d.lda >= (1LL << 31) || (int64_t)d.K * d.ldb >= (1LL << 31)) return false;   // 32-bit element offsets in the kernel
    DwP p;
    p.Y = reinterpret_cast<const bf16_t*>(d.A);
    p.X = reinterpret_cast<const bf16_t*>(d.B);
    p.C = reinterpret_cast<float*>(d.C);
    p.ysum = const_cast<float*>(d.bias);
    p.ldy = d.lda; p.ldx = d.ldb; p.ldc = d.ldc;
    p.rows = d.K; p.n1 = d.M; p.n2 = d.N;
    p.tiles_n2 = d.N / DW_T;
    p.ntiles = (d.M / DW_T) * p.tiles_n2;
    // ~2 workgroups per CU for the wide problems; the 4-tile (256x256) ones run faster with one per CU: half the row
    // ranges means half the memory-side atomics (measured 46.9 -> 40.6 us), which outweighs the lost latency hiding
    static const int forced = [] { const char* e = getenv("POET_DW_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const int target = forced ? forced : (p.ntiles <= 4 ? 256 : 512);
    int per = target / 8 / p.ntiles;                                   // row ranges per XCD
    if (per < 1) per = 1;
    p.splits = per * 8;
    const int nst = (p.rows + DW_RS - 1) / DW_RS;
    if (p.splits > nst) p.splits = nst;
    const int nblocks = ((p.splits + 7) / 8) * p.ntiles * 8;
    static unsigned long long attr_done[2] = {0, 0};
    lds_attr_once(reinterpret_cast<const void*>(gemm_dw_kernel<false>), 2 * DW_STAGE, attr_done[0]);
    lds_attr_once(reinterpret_cast<const void*>(gemm_dw_kernel<true>), 2 * DW_STAGE, attr_done[1]);
    // partial tiles through the caller's workspace when it is large enough (PoetGemmDesc.workspace), else fp32 atomics
    const int64_t need = (int64_t)p.splits * p.ntiles * DW_T * DW_T * 4;
    static const int no_ws = [] { const char* e = getenv("POET_DW_NO_WORKSPACE"); return e && atoi(e) ? 1 : 0; }();
    p.ws = (!no_ws && d.workspace && d.workspace_bytes >= need && (reinterpret_cast<uintptr_t>(d.workspace) & 15) == 0) ? reinterpret_cast<float*>(d.workspace) : nullptr;
    static const int no_lean = [] { const char* e = getenv("POET_DW_NO_LEAN"); return e && atoi(e) ? 1 : 0; }();      // (A/B aid)
    if (p.rows >= DW_RS && !no_lean && (int64_t)p.rows * p.ldy * 2 < (1LL << 32) && (int64_t)p.rows * p.ldx * 2 < (1LL << 32))
        hipLaunchKernelGGL(gemm_dw_kernel<true>, dim3(nblocks), dim3(256), 2 * DW_STAGE, st, p);
    else hipLaunchKernelGGL(gemm_dw_kernel<false>, dim3(nblocks), dim3(256), 2 * DW_STAGE, st, p);
    if (p.ws) hipLaunchKernelGGL(dw_reduce_kernel, dim3(DW_T * DW_T / 256, p.ntiles), dim3(256), 0, st, p, p.splits);
    return true;
}

}  // namespace poet
