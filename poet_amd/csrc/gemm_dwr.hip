// Weight-gradient GEMM, DMA-ring form, for gfx950:  dW[n1, n2] += sum_r dY[r, n1] * X[r, n2]  (+ db[n1] += sum_r dY[r, n1]),
// r over ~1e5 token rows -- the backward of every encoder nn.Linear (reference: autograd of the Linear layers in
// models/deformable_transformer.py:193-208 and of MSDeformAttn's projections).  Same product as gemm_dw.hip, organised the way
// the counters of round 4 asked for:
//
//  * one persistent 768-thread workgroup per CU owns a 256 (n1) x 128 (n2) tile of dW and one contiguous range of 64-row stages
//    (1024 x 256: 8 tiles x 32 row ranges): a quarter less panel traffic out of the L2s than 128 x 128 tiles (627 instead of
//    836 MB at 1024 x 256) at HALF the partial-tile bytes of a 256 x 256 tile;
//  * the two operand panels of a stage (64 x 256 of dY, 64 x 128 of X: 48 KB) reach LDS by DMA (global_load_lds_dwordx4: no
//    VGPRs, no ds_write, no staging instructions in the compute waves' stream) into a THREE-stage ring, two stages = 96 KB in
//    flight per CU, issued by 4 loader waves that do nothing else; counted s_waitcnt vmcnt(N) + one raw s_barrier per stage
//    (the skeleton of gemm_pipe.hip);
//  * the LDS images are plain row-major copies of memory (the reduction index r is the slow dimension of BOTH operands) with the
//    32-byte pieces of a row XOR-swizzled by the row index -- applied to the per-lane SOURCE address of the DMA -- and the 8
//    compute waves (4 x 2, 64 x 64 of the tile each) take their MFMA fragments out with ds_read_b64_tr_b16, the CDNA4 transposing
//    LDS read, exactly as gemm_dw.hip does;
//  * the bias gradient (column sums of dY) comes off the matrix pipe: the n2-tile-0 workgroups multiply the dY fragments they
//    already hold with a fragment of ones (4 extra MFMAs per 16 in four of their eight waves);
//  * partial tiles (and the bias partials) leave by plain stores into the caller's workspace and ONE reduction kernel adds the
//    row ranges into the gradient in a fixed order: no atomics at all, deterministic.
#include "gemm.cuh"

#include <stdlib.h>

namespace poet {

namespace {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;

constexpr int R_TY = 256, R_TX = 128;            // tile: n1 x n2
constexpr int R_RS = 64;                         // rows per stage
constexpr int R_NST = 3;                         // ring depth (R_NST - 1 stages in flight)
constexpr int R_YROWB = R_TY * 2, R_XROWB = R_TX * 2;
constexpr int R_YPANEL = R_RS * R_YROWB;         // 32 KB
constexpr int R_XPANEL = R_RS * R_XROWB;         // 16 KB
constexpr int R_STAGE = R_YPANEL + R_XPANEL;     // 48 KB
constexpr int R_NCW = 8, R_NLW = 4, R_NT = (R_NCW + R_NLW) * 64;
constexpr int R_YQ = R_YPANEL / 1024 / R_NLW;    // DMA instructions per loader wave and stage: 8 (dY) + 4 (X)
constexpr int R_XQ = R_XPANEL / 1024 / R_NLW;
constexpr int R_TILE_F = R_TY * R_TX;            // floats per partial tile

struct DwrP {
    const bf16_t* Y;
    const bf16_t* X;
    float* C;
    float* ysum;
    float* ws;
    float* wsb;           // bias-gradient partials [splits][nsum][n1] behind the partial tiles (nullptr: no column sums)
    // per-segment column sums of dY (PoetGemmDesc.seg_sums): nsum = seg_n > 0 ? seg_n : 1 sums per column
    float* seg_out;
    int64_t ld_seg;
    int seg_n, seg_period;
    int seg_start[9];
    const bf16_t* X_alt;  // tiles with n1_0 >= m_alt pair with X_alt (PoetGemmDesc.B_alt); nullptr: off
    int64_t ldx_alt;
    int m_alt;
    // fused input gradient (gemm_dwx_kernel): dX[rows][n2] = (dY W) (x gate), W bf16 [256][n2]
    const bf16_t* Wt;
    bf16_t* dx;
    int64_t ldw, lddx;
    int gate;
    float gate_scale;
    int64_t ldy, ldx, ldc;
    int rows, n1, n2, ntiles, tiles_n2, splits;
};

// piece swizzle of a row (gemm_dw.hip's): 32-byte pieces XORed with (row & 3) | bit 3 of the row << 2
__device__ __forceinline__ int r_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ void r_dma16(uint32_t voff, const void* base, uint32_t lds) {
    // (M0 is written behind the compiler's back -- hipcc rejects "m0" on a clobber list; nothing else in this translation unit uses M0:
    // tests/test_abi_cpu.py::test_pipe_kernel_m0_only_in_dma greps the ISA)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void r_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(R_NT, 3) void gemm_dwr_kernel(const DwrP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- work assignment: (tile, row range); the tiles of a range share an XCD (workgroup b -> XCD b % 8) ----
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % p.ntiles, split = (j / p.ntiles) * 8 + xcd;
    if (split >= p.splits) return;
    const int n1_0 = (tile / p.tiles_n2) * R_TY, n2_0 = (tile % p.tiles_n2) * R_TX;
    const int nst = (p.rows + R_RS - 1) / R_RS;
    const int s_lo = (int)((int64_t)split * nst / p.splits), s_hi = (int)((int64_t)(split + 1) * nst / p.splits);
    const int S = s_hi - s_lo;
    if (S <= 0) return;
    // a ragged LAST stage of the whole problem: its rows past p.rows are clamped re-reads that the compute waves zero in LDS
    const int ragged = (s_hi == nst && (p.rows % R_RS) != 0) ? (p.rows - (nst - 1) * R_RS) : 0;      // valid rows of that stage, 0 = none

    if (wave >= R_NCW) {
        // ====== loader waves: the two panels of every stage by LDS-DMA, R_NST - 1 stages ahead ======
        constexpr int PD = R_NST - 1, NPL = R_YQ + R_XQ, LW = NPL * (PD - 1);
        const int lw = wave - R_NCW;
        const bool alt = p.X_alt != nullptr && n1_0 >= p.m_alt;          // (uniform) this tile's rows of dW pair with the second input
        const int ldyB = (int)p.ldy * 2, ldxB = (int)(alt ? p.ldx_alt : p.ldx) * 2;
        // instruction q of this wave covers 1 KB of a panel image: dY 2 rows x 512 B, X 4 rows x 256 B; lane -> (row, 16-byte chunk)
        // of the IMAGE, whose content is the source chunk with its 32-byte piece un-swizzled
        int y_row[R_YQ], y_col[R_YQ], x_row[R_XQ], x_col[R_XQ];
#pragma unroll
        for (int q = 0; q < R_YQ; ++q) {
            const int pos = ((lw * R_YQ + q) * 64 + lane) * 16, row = pos / R_YROWB, c = (pos % R_YROWB) >> 4, pc = c >> 1;
            y_row[q] = row;
            y_col[q] = ((((pc & 8) | ((pc ^ r_swz(row)) & 7)) << 1) | (c & 1)) * 16;
        }
#pragma unroll
        for (int q = 0; q < R_XQ; ++q) {
            const int pos = ((lw * R_XQ + q) * 64 + lane) * 16, row = pos / R_XROWB, c = (pos % R_XROWB) >> 4, pc = c >> 1;
            x_row[q] = row;
            x_col[q] = ((((pc ^ r_swz(row)) & 7) << 1) | (c & 1)) * 16;
        }
        const char* Yb = reinterpret_cast<const char*>(p.Y + n1_0);
        const char* Xb = reinterpret_cast<const char*>((alt ? p.X_alt : p.X) + n2_0);
        int l_s = s_lo, l_slot = 0;
        auto issue = [&]() {
            const int r0 = l_s * R_RS;
            const int rv = min(R_RS, p.rows - r0);                      // (uniform) valid rows of this stage
            const char* yb = Yb + (int64_t)r0 * ldyB;
            const char* xb = Xb + (int64_t)r0 * ldxB;
            const uint32_t slot = lds0 + l_slot * R_STAGE;
#pragma unroll
            for (int q = 0; q < R_YQ; ++q)
                r_dma16((uint32_t)(min(y_row[q], rv - 1) * ldyB + y_col[q]), yb, slot + (lw * R_YQ + q) * 1024);
#pragma unroll
            for (int q = 0; q < R_XQ; ++q)
                r_dma16((uint32_t)(min(x_row[q], rv - 1) * ldxB + x_col[q]), xb, slot + R_YPANEL + (lw * R_XQ + q) * 1024);
            ++l_s;
            l_slot = (l_slot + 1 == R_NST) ? 0 : l_slot + 1;
        };
#pragma unroll 1
        for (int i = 0; i < min(PD, S); ++i) issue();
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            // stage s has landed (this wave's share) when at most the loads issued after it are outstanding
            if (min(S, s + PD) - (s + 1) < PD - 1) r_wait_vm<0>(); else r_wait_vm<LW>();
            __builtin_amdgcn_s_barrier();                               // every share of stage s landed; the slot of stage s - 1 is free
            if (ragged && s == S - 1) __builtin_amdgcn_s_barrier();     // (the compute waves zero the clamped rows in between)
            if (l_s < s_hi) issue();
        }
        return;
    }

    // =========================== compute waves: 64 x 64 of the tile each (4 x 2) ===========================
    const int wm = wave >> 1, wn = wave & 1;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // column sums of dY (bias gradient / per-segment sums): in the n2-tile-0 workgroups, the two waves that hold the same dY fragments
    // (wn = 0, 1) take two of the four fragments each, so every SIMD carries the same 2 extra MFMAs per 16
    const bool do_sum = p.wsb != nullptr && n2_0 == 0;                 // workgroup-uniform
    f32x4_t accs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) accs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
    {
        const uint4 o = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        ones = __builtin_bit_cast(bf16x8_t, o);
    }

    // per-lane constants of the transposing read: source lane r of a 16-lane block points at row 8 b + (r >> 2) (+ 4 for the
    // second half), 8-byte chunk r & 3 of the fragment's 32-byte piece
    const int r16 = lane & 15, b4 = lane >> 4;
    const int trow = 8 * b4 + (r16 >> 2), tswz = (r16 >> 2) | ((b4 & 1) << 2), tcol = (r16 & 3) * 8;
    int ty[4], tx[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int py = wm * 4 + f, px = wn * 4 + f;
        ty[f] = trow * R_YROWB + (((py & 8) | ((py ^ tswz) & 7)) << 5) + tcol;
        tx[f] = R_YPANEL + trow * R_XROWB + (((px ^ tswz) & 7) << 5) + tcol;
    }

    // segment-indicator fragments of a stage's two 32-row blocks (pos = position of the stage's first row inside its period, uniform)
    bf16x8_t indk[R_RS / 32] = {ones, ones};
    // the segment table in SGPRs for the whole kernel: read through `p` inside the loop it is re-fetched from the kernel-argument
    // segment every stage (s_load + s_waitcnt lgkmcnt(0): ~15 % on the tile-column-0 workgroups)
    int sst[8], s_per = p.seg_period, s_n = p.seg_n;
#pragma unroll
    for (int t = 0; t < 8; ++t) { sst[t] = (t >= 1 && t < p.seg_n) ? p.seg_start[t] : 0x7fffffff; asm volatile("" : "+s"(sst[t])); }
    asm volatile("" : "+s"(s_per), "+s"(s_n));
    // scalar cursor: the segment `cur` the stage's first row lies in and the position `nxt` where it ends.  A stage that ends before
    // `nxt` (almost all of them: levels are hundreds of rows) has both fragments = ones in column `cur`; only a stage that straddles a
    // segment or image boundary takes the block-by-block / row-by-row forms below.  (The first version re-derived the segment of every
    // 32-row block with a 14-compare chain per block: 15 us of 58 on the tile-column-0 workgroups.)
    int cur = 0, nxt = 0x7fffffff;
    auto locate = [&](int pos) __attribute__((always_inline)) {
        int c = 0, n = s_per;
#pragma unroll
        for (int t = 1; t < 8; ++t) {
            c += (pos >= sst[t]) ? 1 : 0;
            n = (sst[t] > pos && sst[t] < n) ? sst[t] : n;
        }
        cur = __builtin_amdgcn_readfirstlane(c);
        nxt = __builtin_amdgcn_readfirstlane(n);
    };
    auto make_ind = [&](int pos0) __attribute__((always_inline)) {
        if (!(do_sum && s_n > 0)) return;
        if (__builtin_amdgcn_readfirstlane(pos0 + R_RS - 1 < nxt ? 1 : 0)) {
            const uint32_t w1 = (cur == r16) ? 0x3f803f80u : 0u;
            const uint4 o = make_uint4(w1, w1, w1, w1);
#pragma unroll
            for (int ks = 0; ks < R_RS / 32; ++ks) indk[ks] = __builtin_bit_cast(bf16x8_t, o);
            return;
        }
        {
#pragma unroll
            for (int ks = 0; ks < R_RS / 32; ++ks) {
                bf16x8_t ind;
                {
                    // column n of the B fragment = indicator "row k belongs to segment n": the product then IS the per-segment column sum.
                    // Almost every 32-row block lies inside ONE segment (levels are hundreds of rows): decided on the scalar unit, the
                    // fragment is then all-ones in the column of that segment.  A block that straddles a segment or image boundary
                    // takes the per-row form: this lane holds k = 8 b4 .. 8 b4 + 7 of the block for column n = r16.
                    const int qa = __builtin_amdgcn_readfirstlane(pos0) + ks * 32;       // (uniform, and SAID so: a branch hipcc takes for divergent runs both sides under exec masks) position of the block's first row inside its period
                    int sa = 0, sb = 0;
#pragma unroll
                    for (int t = 1; t < 8; ++t) {
                        sa += (qa >= sst[t]) ? 1 : 0;
                        sb += (qa + 31 >= sst[t]) ? 1 : 0;
                    }
                    if (__builtin_amdgcn_readfirstlane((qa + 31 < s_per && sa == sb) ? 1 : 0)) {
                        const uint32_t w1 = (sa == r16) ? 0x3f803f80u : 0u;
                        const uint4 o = make_uint4(w1, w1, w1, w1);
                        ind = __builtin_bit_cast(bf16x8_t, o);
                    } else {
                        int q0 = qa + 8 * b4;
                        if (q0 >= s_per) q0 -= s_per;
                        uint32_t w[4];
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            uint32_t pk = 0u;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                int q = q0 + e2 * 2 + h;
                                if (q >= s_per) q -= s_per;
                                int sgm = 0;
#pragma unroll
                                for (int t = 1; t < 8; ++t) sgm += (q >= sst[t]) ? 1 : 0;
                                pk |= (sgm == r16 ? 0x3f80u : 0u) << (16 * h);
                            }
                            w[e2] = pk;
                        }
                        const uint4 o = make_uint4(w[0], w[1], w[2], w[3]);
                        ind = __builtin_bit_cast(bf16x8_t, o);
                    }
                }
                indk[ks] = ind;
            }
        }
    };
    int slot = 0;
    int pos0 = p.seg_n > 0 ? (int)(((int64_t)s_lo * R_RS) % p.seg_period) : 0;      // position of the stage's first row inside its period (uniform)
    if (do_sum && s_n > 0) locate(pos0);
    make_ind(pos0);
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        __builtin_amdgcn_s_barrier();                                   // stage s has landed
        asm volatile("" ::: "memory");
        char* sb = smem + slot * R_STAGE;
        slot = (slot + 1 == R_NST) ? 0 : slot + 1;
        if (ragged && s == S - 1) {                                     // zero the rows past the end of the problem (uniform branch)
            const int ct = tid;                                         // 512 compute threads
            for (int i = ct; i < (R_RS - ragged) * (R_YROWB / 16); i += R_NCW * 64)
                *reinterpret_cast<uint4*>(sb + ragged * R_YROWB + i * 16) = make_uint4(0, 0, 0, 0);
            for (int i = ct; i < (R_RS - ragged) * (R_XROWB / 16); i += R_NCW * 64)
                *reinterpret_cast<uint4*>(sb + R_YPANEL + ragged * R_XROWB + i * 16) = make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): the zeros are in LDS
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int ks = 0; ks < R_RS / 32; ++ks) {
            bf16x8_t fy[4], fx[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                struct { v4s_t lo, hi; } u, v;
                u.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + ty[f] + ks * 32 * R_YROWB));
                u.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + ty[f] + (ks * 32 + 4) * R_YROWB));
                v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + tx[f] + ks * 32 * R_XROWB));
                v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + tx[f] + (ks * 32 + 4) * R_XROWB));
                fy[f] = __builtin_bit_cast(bf16x8_t, u);
                fx[f] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[jj], acc[i][jj], 0, 0, 0);
            if (do_sum) {
                const bf16x8_t ind = indk[ks];
                if (wn == 0) {
                    accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[0], ind, accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[1], ind, accs[1], 0, 0, 0);
                } else {
                    accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[2], ind, accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[3], ind, accs[1], 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        pos0 += R_RS;
        if (do_sum && s_n > 0) {
            const bool wrap = pos0 >= s_per;
            if (wrap) pos0 -= s_per;
            if (wrap || pos0 >= nxt) locate(pos0);
        }
        make_ind(pos0);                 // the NEXT stage's fragments, formed behind this stage's MFMAs (issued, still executing)
    }

    // ---- bias gradient: every column of accs[i] is the row sum; lanes r16 == 0 hold rows 4 b4 + t of fragment i.  Partials go to
    // the workspace like the tiles (128 row ranges adding into 256 addresses with memory-side atomics cost 16 us at 256 x 256) ----
    // (per-segment sums: column r16 < seg_n of accs[i] is segment r16's sum)
    const int nsum = p.seg_n > 0 ? p.seg_n : 1;
    if (do_sum && r16 < nsum) {
        float* bp = p.wsb + ((int64_t)split * nsum + r16) * p.n1 + n1_0 + wm * 64 + wn * 32 + b4 * 4;       // fragments 2 wn, 2 wn + 1
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4_t*>(bp + i * 16) = accs[i];
    }
    // ---- partial tile: plain, fully coalesced stores of the register image (wave, fragment, t, lane) ----
    float* wp = p.ws + ((int64_t)split * p.ntiles + tile) * R_TILE_F + wave * 4096 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t) wp[((i * 4 + jj) * 4 + t) * 64] = acc[i][jj][t];
}

// ---- the same weight gradient WITH the Linear's input gradient in the same pass --------------------------------------------------
// Backward of y = x W^T + b for the encoder's Linears with a 256-wide output (FFN linear2: deformable_transformer.py:193-197, and
// MSDeformAttn's output_proj :202-203): autograd runs  dW += dY^T X,  db += colsum(dY)  and  dX = dY W (x ReLU / dropout gate of the
// hidden activation)  as two products that BOTH stream dY (rows x 256) and, for the gated form, X (rows x n2: 209 MB for the FFN's
// hidden activation -- once as the dW operand, once as the gate).  Here the workgroup that owns the 256 x 128 tile of dW and a row
// range holds, per 32-row stage, exactly the operands of the matching dX block as well: the dY panel (32 x 256 = the whole reduction
// of dX) and the X panel (32 x 128 = the gate of its 128 output columns).  It keeps its 256 x 128 slab of W in LDS for its whole life
// (64 KB, columns permuted so that a lane of the transposed product owns 8 CONSECUTIVE output columns: 16-byte stores, 64 contiguous
// bytes per row and instruction) and writes the dX block straight from registers: the hidden activation is read ONCE, the separate
// input-gradient launch (130 us at 102 080 x 1024) disappears, and the matrix pipe of this memory-bound kernel goes from 40 % to 80 %.
constexpr int X_RS = 32;                          // rows per stage
constexpr int X_NST = 4;                          // ring depth: 3 stages = 72 KB in flight
constexpr int X_YPANEL = X_RS * R_YROWB;          // 16 KB
constexpr int X_XPANEL = X_RS * R_XROWB;          // 8 KB
constexpr int X_STAGE = X_YPANEL + X_XPANEL;      // 24 KB
constexpr int X_WSLAB = R_TY * R_XROWB;           // 64 KB: W[o][128 columns of the slab]
constexpr int X_LDS = X_NST * X_STAGE + X_WSLAB;  // 160 KB
constexpr int X_YQ = X_YPANEL / 1024 / R_NLW;     // DMA instructions per loader wave and stage: 4 + 2
constexpr int X_XQ = X_XPANEL / 1024 / R_NLW;

__global__ __launch_bounds__(R_NT, 3) void gemm_dwx_kernel(const DwrP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    char* wslab = smem + X_NST * X_STAGE;

    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % p.ntiles, split = (j / p.ntiles) * 8 + xcd;          // n1 = 256: one tile row, tile = the 128-column slab of n2
    if (split >= p.splits) return;
    const int n2_0 = tile * R_TX;
    const int nst = (p.rows + X_RS - 1) / X_RS;
    const int s_lo = (int)((int64_t)split * nst / p.splits), s_hi = (int)((int64_t)(split + 1) * nst / p.splits);
    const int S = s_hi - s_lo;
    if (S <= 0) return;
    const int ragged = (s_hi == nst && (p.rows % X_RS) != 0) ? (p.rows - (nst - 1) * X_RS) : 0;

    if (wave >= R_NCW) {
        // ====== loader waves ======
        constexpr int PD = X_NST - 1, NPL = X_YQ + X_XQ, LW = NPL * (PD - 1);
        const int lw = wave - R_NCW;
        const int ldyB = (int)p.ldy * 2, ldxB = (int)p.ldx * 2;
        int y_row[X_YQ], y_col[X_YQ], x_row[X_XQ], x_col[X_XQ];
#pragma unroll
        for (int q = 0; q < X_YQ; ++q) {
            const int pos = ((lw * X_YQ + q) * 64 + lane) * 16, row = pos / R_YROWB, c = (pos % R_YROWB) >> 4, pc = c >> 1;
            y_row[q] = row;
            y_col[q] = ((((pc & 8) | ((pc ^ r_swz(row)) & 7)) << 1) | (c & 1)) * 16;
        }
#pragma unroll
        for (int q = 0; q < X_XQ; ++q) {
            const int pos = ((lw * X_XQ + q) * 64 + lane) * 16, row = pos / R_XROWB, c = (pos % R_XROWB) >> 4, pc = c >> 1;
            x_row[q] = row;
            x_col[q] = ((((pc ^ r_swz(row)) & 7) << 1) | (c & 1)) * 16;
        }
        const char* Yb = reinterpret_cast<const char*>(p.Y);
        const char* Xb = reinterpret_cast<const char*>(p.X + n2_0);
        int l_s = s_lo, l_slot = 0;
        auto issue = [&]() {
            const int r0 = l_s * X_RS;
            const int rv = min(X_RS, p.rows - r0);
            const char* yb = Yb + (int64_t)r0 * ldyB;
            const char* xb = Xb + (int64_t)r0 * ldxB;
            const uint32_t slot = lds0 + l_slot * X_STAGE;
#pragma unroll
            for (int q = 0; q < X_YQ; ++q)
                r_dma16((uint32_t)(min(y_row[q], rv - 1) * ldyB + y_col[q]), yb, slot + (lw * X_YQ + q) * 1024);
#pragma unroll
            for (int q = 0; q < X_XQ; ++q)
                r_dma16((uint32_t)(min(x_row[q], rv - 1) * ldxB + x_col[q]), xb, slot + X_YPANEL + (lw * X_XQ + q) * 1024);
            ++l_s;
            l_slot = (l_slot + 1 == X_NST) ? 0 : l_slot + 1;
        };
#pragma unroll 1
        for (int i = 0; i < min(PD, S); ++i) issue();
        __builtin_amdgcn_s_barrier();                                   // (the compute waves have staged the W slab)
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            if (min(S, s + PD) - (s + 1) < PD - 1) r_wait_vm<0>(); else r_wait_vm<LW>();
            __builtin_amdgcn_s_barrier();
            if (ragged && s == S - 1) __builtin_amdgcn_s_barrier();
            if (l_s < s_hi) issue();
        }
        return;
    }

    // =========================== compute waves ===========================
    // ---- the W slab: W[o][n2_0 + h] -> image [o][ic], ic = column permutation inside every 32-column group: h = 8 g + 4 tl + t
    // (g < 4, tl < 2, t < 4) sits at ic = 16 tl + 4 g + t, so that the two 16-column fragments tl = 0, 1 of a group give a lane
    // (MFMA rows 4 g + t) the 8 consecutive output columns 8 g .. 8 g + 7; 32-byte pieces XOR-swizzled by the row like the panels ----
    {
        const bf16_t* Wg = p.Wt + n2_0;
        for (int it = tid; it < R_TY * (R_TX / 8); it += R_NCW * 64) {     // item: 8 consecutive h of one row o
            const int o = it >> 4, h0 = (it & 15) * 8, g32 = h0 >> 5, g = (h0 >> 3) & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(Wg + (int64_t)o * p.ldw + h0);
            const int sw = r_swz(o);
            const int ic0 = g32 * 32 + 4 * g, ic1 = ic0 + 16;                 // tl = 0: h0 .. h0 + 3;  tl = 1: h0 + 4 .. h0 + 7
            *reinterpret_cast<uint2*>(wslab + o * R_XROWB + ((((ic0 >> 4) ^ sw) & 7) << 5) + (ic0 & 15) * 2) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(wslab + o * R_XROWB + ((((ic1 >> 4) ^ sw) & 7) << 5) + (ic1 & 15) * 2) = make_uint2(v.z, v.w);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int wm = wave >> 1, wn = wave & 1;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_sum = p.wsb != nullptr && n2_0 == 0;
    f32x4_t accs[2];
    accs[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accs[1] = accs[0];
    bf16x8_t ones;
    {
        const uint4 o = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        ones = __builtin_bit_cast(bf16x8_t, o);
    }
    const int r16 = lane & 15, b4 = lane >> 4;
    const int trow = 8 * b4 + (r16 >> 2), tswz = (r16 >> 2) | ((b4 & 1) << 2), tcol = (r16 & 3) * 8;
    int ty[4], tx[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int py = wm * 4 + f, px = wn * 4 + f;
        ty[f] = trow * R_YROWB + (((py & 8) | ((py ^ tswz) & 7)) << 5) + tcol;
        tx[f] = X_YPANEL + trow * R_XROWB + (((px ^ tswz) & 7) << 5) + tcol;
    }
    // dX part: this wave computes rows ru * 16 .. + 15 of the stage x the 32-column group cg of the slab (two 16-column fragments)
    const int ru = wave & 1, cg = wave >> 1;
    const int drow = ru * 16 + r16, dsw = r_swz(drow);                  // this lane's dY row as the B operand (row r16 of the unit)
    int wx[2];                                                          // W fragment tl: image piece 2 cg + tl
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) wx[tl] = trow * R_XROWB + ((((cg * 2 + tl) ^ tswz) & 7) << 5) + tcol;
    // gate: hidden[row][8 b4 .. 8 b4 + 7 of group cg]: 16-byte chunk 4 cg + b4 of the X panel row
    const int gate_off = X_YPANEL + drow * R_XROWB + ((((cg * 2 + (b4 >> 1)) ^ dsw) & 7) << 5) + (b4 & 1) * 16;
    const int out_col = n2_0 + cg * 32 + b4 * 8;

    int slot = 0;
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        __builtin_amdgcn_s_barrier();                                   // stage s has landed
        asm volatile("" ::: "memory");
        char* sb = smem + slot * X_STAGE;
        slot = (slot + 1 == X_NST) ? 0 : slot + 1;
        if (ragged && s == S - 1) {
            for (int i = tid; i < (X_RS - ragged) * (R_YROWB / 16); i += R_NCW * 64)
                *reinterpret_cast<uint4*>(sb + ragged * R_YROWB + i * 16) = make_uint4(0, 0, 0, 0);
            for (int i = tid; i < (X_RS - ragged) * (R_XROWB / 16); i += R_NCW * 64)
                *reinterpret_cast<uint4*>(sb + X_YPANEL + ragged * R_XROWB + i * 16) = make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // ---- weight gradient: one 32-row block ----
        {
            bf16x8_t fy[4], fx[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                struct { v4s_t lo, hi; } u, v;
                u.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + ty[f]));
                u.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + ty[f] + 4 * R_YROWB));
                v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + tx[f]));
                v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(sb + tx[f] + 4 * R_XROWB));
                fy[f] = __builtin_bit_cast(bf16x8_t, u);
                fx[f] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[jj], acc[i][jj], 0, 0, 0);
            if (do_sum) {
                if (wn == 0) {
                    accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[0], ones, accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[1], ones, accs[1], 0, 0, 0);
                } else {
                    accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[2], ones, accs[0], 0, 0, 0);
                    accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[3], ones, accs[1], 0, 0, 0);
                }
            }
        }
        // ---- input gradient block: D'[h][r] = sum_o W[o][h] dY[r][o], transposed product, 8 k-steps of 32 ----
        f32x4_t dacc[2];
        dacc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dacc[1] = dacc[0];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) {
            const int pc = kq * 2 + (b4 >> 1);                          // 32-byte piece of the dY row holding o = 32 kq + 8 b4 .. + 7
            const bf16x8_t dyf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(
                sb + drow * R_YROWB + (((pc & 8) | ((pc ^ dsw) & 7)) << 5) + (b4 & 1) * 16));
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                struct { v4s_t lo, hi; } w;
                w.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(wslab + wx[tl] + kq * 32 * R_XROWB));
                w.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(wslab + wx[tl] + (kq * 32 + 4) * R_XROWB));
                dacc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), dyf, dacc[tl], 0, 0, 0);
            }
        }
        {
            float v[8] = {dacc[0][0], dacc[0][1], dacc[0][2], dacc[0][3], dacc[1][0], dacc[1][1], dacc[1][2], dacc[1][3]};
            if (p.gate) {
                const uint4 hq = *reinterpret_cast<const uint4*>(sb + gate_off);
                const uint32_t hw[4] = {hq.x, hq.y, hq.z, hq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = __uint_as_float(hw[e] << 16) > 0.f ? v[2 * e] * p.gate_scale : 0.f;
                    v[2 * e + 1] = __uint_as_float(hw[e] & 0xffff0000u) > 0.f ? v[2 * e + 1] * p.gate_scale : 0.f;
                }
            }
            const int64_t grow = (int64_t)(s_lo + s) * X_RS + drow;
            if (grow < p.rows)
                *reinterpret_cast<uint4*>(p.dx + grow * p.lddx + out_col) =
                    make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
        }
        asm volatile("" ::: "memory");
    }

    if (do_sum && r16 == 0) {
        float* bp = p.wsb + (int64_t)split * p.n1 + wm * 64 + wn * 32 + b4 * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4_t*>(bp + i * 16) = accs[i];
    }
    float* wp = p.ws + ((int64_t)split * p.ntiles + tile) * R_TILE_F + wave * 4096 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t) wp[((i * 4 + jj) * 4 + t) * 64] = acc[i][jj][t];
}

// dW[tile] += sum over row ranges of the partial tiles (single owner per element: a plain read-modify-write, fixed order)
__global__ __launch_bounds__(256) void dwr_reduce_kernel(const DwrP p) {
    const int tile = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;               // e: index inside the 256 x 128 register image
    const float* wp = p.ws + (int64_t)tile * R_TILE_F + e;
    const int64_t stride = (int64_t)p.ntiles * R_TILE_F;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < p.splits; k += 4) {
        s0 += wp[(int64_t)k * stride]; s1 += wp[(int64_t)(k + 1) * stride];
        s2 += wp[(int64_t)(k + 2) * stride]; s3 += wp[(int64_t)(k + 3) * stride];
    }
    for (; k < p.splits; ++k) s0 += wp[(int64_t)k * stride];
    const int lane = e & 63, t = (e >> 6) & 3, fr = (e >> 8) & 15, wid = e >> 12;
    const int i = fr >> 2, jj = fr & 3, wm = wid >> 1, wn = wid & 1, r16 = lane & 15, b4 = lane >> 4;
    const int n1_0 = (tile / p.tiles_n2) * R_TY, n2_0 = (tile % p.tiles_n2) * R_TX;
    float* c = p.C + (int64_t)(n1_0 + wm * 64 + i * 16 + b4 * 4 + t) * p.ldc + n2_0 + wn * 64 + jj * 16 + r16;
    *c += (s0 + s1) + (s2 + s3);
    const int nsum = p.seg_n > 0 ? p.seg_n : 1;
    if (p.wsb && (int)blockIdx.x < 8 * nsum && n2_0 == 0) {             // column sums of dY over the row ranges (fixed order):
        // per segment 8 blocks x 32 entries, 8 lanes per entry take every 8th row range (independent loads), folded with three shuffle steps
        const int sg = blockIdx.x >> 3, ent = (blockIdx.x & 7) * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
        const float* bp = p.wsb + (int64_t)sg * p.n1 + n1_0 + ent;
        const int64_t st = (int64_t)nsum * p.n1;
        float b0 = 0.f, b1 = 0.f;
        int q = part;
        for (; q + 8 < p.splits; q += 16) { b0 += bp[(int64_t)q * st]; b1 += bp[(int64_t)(q + 8) * st]; }
        if (q < p.splits) b0 += bp[(int64_t)q * st];
        float b = b0 + b1;
        b += __shfl_xor(b, 1); b += __shfl_xor(b, 2); b += __shfl_xor(b, 4);
        if (part == 0) {
            if (p.seg_n > 0) p.seg_out[(int64_t)sg * p.ld_seg + n1_0 + ent] += b;
            else p.ysum[n1_0 + ent] += b;
        }
    }
}

int dwr_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

}  // namespace

bool gemm_dwr_try(const GemmK& g, hipStream_t st) {
    const PoetGemmDesc& d = g.d;
    // which shapes take the ring (POET_DW_RING: 0 = none, 1 = all eligible, default = the ones it measured faster on: see gemm_dwr_shapes)
    static const int mode = [] { const char* e = getenv("POET_DW_RING"); return e ? atoi(e) : 2; }();
    if (mode == 0) return false;
    // the dW form of poet_gemm: A = dY stored [K = rows][M = n1], B = X stored [K = rows][N = n2], fp32 accumulate into C
    if (!d.a_kmajor || !d.b_kmajor || !(d.atomic || d.splitk > 1) || d.batch != 1 || d.A2) return false;
    if (d.a_dtype != POET_BF16 || d.b_dtype != POET_BF16 || d.c_dtype != POET_F32 || d.compute != POET_BF16) return false;
    if (d.M % R_TY != 0 || d.N % R_TX != 0 || d.K < 8192 || d.alpha != 1.f) return false;
    if (!g.a_vec || !g.b_vec) return false;
    if ((int64_t)R_RS * d.lda * 2 >= (1LL << 31) || (int64_t)R_RS * d.ldb * 2 >= (1LL << 31)) return false;   // 32-bit lane offsets inside a stage
    if (d.B_alt && ((int64_t)R_RS * d.ldb_alt * 2 >= (1LL << 31) || (reinterpret_cast<uintptr_t>(d.B_alt) & 15) || (d.ldb_alt & 7) || d.m_alt % R_TY != 0)) return false;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B)) & 15) return false;
    if ((d.lda & 7) || (d.ldb & 7)) return false;
    DwrP p;
    p.Y = reinterpret_cast<const bf16_t*>(d.A);
    p.X = reinterpret_cast<const bf16_t*>(d.B);
    p.C = reinterpret_cast<float*>(d.C);
    p.ysum = const_cast<float*>(d.bias);
    p.ldy = d.lda; p.ldx = d.ldb; p.ldc = d.ldc;
    p.rows = d.K; p.n1 = d.M; p.n2 = d.N;
    p.tiles_n2 = d.N / R_TX;
    p.ntiles = (d.M / R_TY) * p.tiles_n2;
    if (mode == 2 && p.ntiles < 6) return false;                         // 256 x 256 (2 tiles, 128 row ranges of 12 stages): the register-staged kernel
    const int cus = dwr_cus();
    int per = cus / 8 / p.ntiles;                                      // row ranges per XCD: one workgroup per CU
    if (per < 1) per = 1;
    p.splits = per * 8;
    const int nst = (p.rows + R_RS - 1) / R_RS;
    if (p.splits > nst) p.splits = nst;
    p.X_alt = reinterpret_cast<const bf16_t*>(d.B_alt); p.ldx_alt = d.ldb_alt; p.m_alt = d.m_alt;
    p.seg_out = d.seg_sums; p.ld_seg = d.ld_seg; p.seg_n = d.seg_sums ? d.seg_n : 0; p.seg_period = d.seg_sums ? d.seg_period : 1;
    for (int i = 0; i < 9; ++i) p.seg_start[i] = d.seg_sums ? d.seg_start[i] : 0;
    const int nsum = p.seg_n > 0 ? p.seg_n : 1;
    const bool sums = p.ysum != nullptr || p.seg_n > 0;
    const int64_t tile_bytes = (int64_t)p.splits * p.ntiles * R_TILE_F * 4;
    const int64_t need = tile_bytes + (sums ? (int64_t)p.splits * nsum * p.n1 * 4 : 0);
    if (!d.workspace || d.workspace_bytes < need || (reinterpret_cast<uintptr_t>(d.workspace) & 15)) return false;
    p.ws = reinterpret_cast<float*>(d.workspace);
    p.wsb = sums ? p.ws + tile_bytes / 4 : nullptr;
    const int nblocks = ((p.splits + 7) / 8) * p.ntiles * 8;
    static unsigned long long attr_done = 0;
    lds_attr_once(reinterpret_cast<const void*>(gemm_dwr_kernel), R_NST * R_STAGE, attr_done);
    hipLaunchKernelGGL(gemm_dwr_kernel, dim3(nblocks), dim3(R_NT), R_NST * R_STAGE, st, p);
    hipLaunchKernelGGL(dwr_reduce_kernel, dim3(R_TILE_F / 256, p.ntiles), dim3(256), 0, st, p);
    return true;
}

}  // namespace poet

using namespace poet;

// dW += dY^T X,  db += colsum(dY),  dX = (dY W) [x (X > 0 ? gate_scale : 0)]  in ONE pass (gemm_dwx_kernel above): the backward of the
// encoder's Linears with a 256-wide output -- FFN linear2 (deformable_transformer.py:193-197; gate = 1: X is the hidden activation after
// ReLU and dropout, its zeros are the gate) and MSDeformAttn's output_proj (:202-203; gate = 0).  dy bf16 [rows][256], x bf16
// [rows][n2], w bf16 [256][n2] (the Linear's weight, [out][in]), dw fp32 [256][n2] (accumulated), db fp32 [256] or NULL (accumulated),
// dx bf16 [rows][n2] (written).  rows >= 8192, n2 a multiple of 128; the workspace takes the partial tiles.  POET_ERR_UNSUPPORTED
// outside that: callers then issue the two poet_gemm calls.
extern "C" int poet_linear_bwd(const void* dy, int64_t ldy, const void* x, int64_t ldx, const void* w, int64_t ldw, float* dw, int64_t lddw,
                               float* db, void* dx, int64_t lddx, int gate, float gate_scale, int64_t rows, int n2, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    POET_CHECK(dy && x && w && dw && dx, POET_ERR_ARG, "linear_bwd: null pointer");
    POET_CHECK(rows >= 8192 && rows < (1ll << 24) && n2 >= 128 && n2 % R_TX == 0 && n2 <= 4096, POET_ERR_UNSUPPORTED,
               "linear_bwd: rows=%lld n2=%d outside the fused kernel's range (rows >= 8192, n2 a multiple of 128)", (long long)rows, n2);
    POET_CHECK(((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx) |
                 reinterpret_cast<uintptr_t>(workspace)) & 15) == 0 && (ldy & 7) == 0 && (ldx & 7) == 0 && (ldw & 7) == 0 && (lddx & 7) == 0 &&
               ldy >= R_TY && ldx >= n2 && ldw >= n2 && lddx >= n2 && lddw >= n2, POET_ERR_ARG, "linear_bwd: 16-byte aligned operands, row strides multiples of 8");
    POET_CHECK((int64_t)X_RS * ldy * 2 < (1ll << 31) && (int64_t)X_RS * ldx * 2 < (1ll << 31), POET_ERR_UNSUPPORTED, "linear_bwd: 32-bit lane offsets");
    DwrP p{};
    p.Y = reinterpret_cast<const bf16_t*>(dy);
    p.X = reinterpret_cast<const bf16_t*>(x);
    p.Wt = reinterpret_cast<const bf16_t*>(w);
    p.dx = reinterpret_cast<bf16_t*>(dx);
    p.C = dw; p.ysum = db;
    p.ldy = ldy; p.ldx = ldx; p.ldc = lddw; p.ldw = ldw; p.lddx = lddx;
    p.rows = (int)rows; p.n1 = R_TY; p.n2 = n2;
    p.tiles_n2 = n2 / R_TX; p.ntiles = p.tiles_n2;
    p.gate = gate ? 1 : 0; p.gate_scale = gate_scale;
    p.seg_n = 0; p.seg_period = 1;
    const int cus = dwr_cus();
    int per = cus / 8 / p.ntiles;
    if (per < 1) per = 1;
    p.splits = per * 8;
    const int nst = (p.rows + X_RS - 1) / X_RS;
    if (p.splits > nst) p.splits = nst;
    const int64_t tile_bytes = (int64_t)p.splits * p.ntiles * R_TILE_F * 4;
    const int64_t need = tile_bytes + (db ? (int64_t)p.splits * p.n1 * 4 : 0);
    POET_CHECK(workspace && workspace_bytes >= need, POET_ERR_UNSUPPORTED, "linear_bwd: workspace of %lld bytes needed", (long long)need);
    p.ws = reinterpret_cast<float*>(workspace);
    p.wsb = db ? p.ws + tile_bytes / 4 : nullptr;
    const int nblocks = ((p.splits + 7) / 8) * p.ntiles * 8;
    static unsigned long long attr_done = 0;
    lds_attr_once(reinterpret_cast<const void*>(gemm_dwx_kernel), X_LDS, attr_done);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_dwx_kernel, dim3(nblocks), dim3(R_NT), X_LDS, st, p);
    hipLaunchKernelGGL(dwr_reduce_kernel, dim3(R_TILE_F / 256, p.ntiles), dim3(256), 0, st, p);
    POET_LAUNCH_CHECK();
    return POET_OK;
}
