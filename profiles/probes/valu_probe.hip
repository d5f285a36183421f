// VALU issue-rate probe: wave64 instructions per cycle per SIMD for a few op classes (gfx950), inline asm so the compiler
// cannot repack them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND> __global__ __launch_bounds__(256) void probe(float* out, int iters, float s) {
    float a[16];
    double d[8];
    uint32_t u[16];
    uint64_t mask = __ballot(threadIdx.x & 1);
    uint32_t sc = 0;
    uint32_t addr = ((threadIdx.x & 56) + 3) * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 7 + i; }
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = a[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (KIND == 1 && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[i]));
                if (KIND == 2) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                if (KIND == 3) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
                if (KIND == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(it));
                if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(it) : "vcc");
                if (KIND == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (KIND == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 8) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(u[i]));
                if (KIND == 9) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 10) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(it) : "vcc");
                if (KIND == 11) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
                if (KIND == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 13) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "v"(s));
                if (KIND == 14) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "v"(s));
                if (KIND == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(it), "s"(mask));
                if (KIND == 16) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (KIND == 17) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 18) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (KIND == 19) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(it), "v"(iters));
                if (KIND == 20) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(it));
                if (KIND == 21) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(it), "v"(iters));
                if (KIND == 22) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
                if (KIND == 23) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(u[i]) : "v"(addr));
                if (KIND == 24) asm volatile("v_cvt_f32_bf16 %0, %0" : "+v"(u[i]));
                if (KIND == 25) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == 26) asm volatile("v_mad_u32_u24 %0, %0, 5, %1" : "+v"(u[i]) : "v"(it));
                if (KIND == 27) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(it), "v"(iters));
                if (KIND == 28) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (KIND == 29) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(u[i]));
                if (KIND == 30) asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
                if (KIND == 31) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(u[i]));
                if (KIND == 32) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 33) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[i & 7]));
                if (KIND == 34) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(mask) : "v"(u[i]), "v"(it));
                if (KIND == 35) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sc) : "v"(u[i]));
                if (KIND == 36) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(a[(i + 1) & 15]));
                if (KIND == 38) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 39) asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %2, 0xffff0000, %1" : "=v"(u[i]), "=v"(u[(i + 3) & 15]) : "v"(u[(i + 1) & 15]));
                if (KIND == 40) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "s"(0xffff0000u), "v"(u[(i + 1) & 15]));
                if (KIND == 41) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
                if (KIND == 42) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
                if (KIND == 37) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i] + (float)u[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + (float)mask + sc;
}
template <int KIND> void run(const char* name, int per_iter, int wpb) {
    float* out; hipMalloc(&out, 1 << 24);
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, wpb * 64>>>(out, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<blocks, wpb * 64>>>(out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * (double)wpb, inst = waves * iters * per_iter;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %d waves/blk %.3f ms  clk per wave-instruction per SIMD %.2f\n", name, wpb, ms, 1024 * cyc / inst);
    hipFree(out);
}
int main() {
    for (int wpb = 4; wpb >= 4; wpb /= 4) {
    run<38>("v_mov_b32_sdwa W0->W1", 64, wpb);
    run<39>("lshl16 + and pair (2 ops)", 128, wpb);
    run<40>("v_and_b32 sgpr mask", 64, wpb);
    run<41>("v_pk_fma_f32 3 regs", 64, wpb);
    run<42>("v_fma_f32 3 regs", 64, wpb);
    run<13>("v_fma_mix_f32 lo", 64, wpb);
    run<14>("v_fma_mix_f32 hi", 64, wpb);
    run<15>("v_cndmask_b32_e64 sgpr", 64, wpb);
    run<16>("v_add_u32", 64, wpb);
    run<17>("v_mul_f32", 64, wpb);
    run<25>("v_sub_f32", 64, wpb);
    run<36>("v_fmac_f32", 64, wpb);
    run<18>("v_max_i32", 64, wpb);
    run<19>("v_med3_i32", 64, wpb);
    run<20>("v_lshl_add_u32", 64, wpb);
    run<21>("v_perm_b32", 64, wpb);
    run<22>("v_bfe_u32", 64, wpb);
    run<23>("ds_bpermute_b32+wait", 64, wpb);
    run<24>("v_cvt_f32_bf16", 64, wpb);
    run<26>("v_mad_u32_u24 inline const", 64, wpb);
    run<27>("v_and_or_b32", 64, wpb);
    run<28>("v_xor_b32", 64, wpb);
    run<29>("v_ashrrev_i32", 64, wpb);
    run<37>("v_lshlrev_b32 (no self dep)", 64, wpb);
    run<30>("v_mov_dpp row_newbcast", 64, wpb);
    run<31>("v_cvt_f32_i32", 64, wpb);
    run<32>("v_fract_f32", 64, wpb);
    run<33>("v_pk_mul_f32", 64, wpb);
    run<34>("v_cmp_lt_u32_e64 -> sgpr", 64, wpb);
    run<35>("v_readlane_b32", 64, wpb);
    run<0>("v_fma_f32", 64, wpb);
    run<12>("v_add_f32", 64, wpb);
    run<1>("v_pk_fma_f32", 32, wpb);
    run<2>("v_lshlrev_b32", 64, wpb);
    run<3>("v_and_b32 (literal)", 64, wpb);
    run<4>("v_mad_u32_u24", 64, wpb);
    run<5>("v_cndmask_b32", 64, wpb);
    run<6>("v_mul_lo_u32", 64, wpb);
    run<7>("v_exp_f32", 64, wpb);
    run<8>("v_cvt_i32_f32", 64, wpb);
    run<9>("v_floor_f32", 64, wpb);
    run<10>("v_cmp_lt_u32", 64, wpb);
    run<11>("v_mov_b32_dpp quad_perm", 64, wpb);
    }
    return 0;
}
