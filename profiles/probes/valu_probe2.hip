// Round 6: issue rate of the packed-fp16 / mixed-precision VALU forms the fp16 value maps could use (gfx950), ns per wave-instruction
// per SIMD at 4 waves per SIMD of independent instructions (same harness as valu_probe.hip, reported in ns: no assumed clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND> __global__ __launch_bounds__(256) void probe(float* out, int iters, float s) {
    float a[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = 0x3c003c00u + threadIdx.x * 7 + i; }
    const uint32_t hs = 0x3c003c01u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (KIND == 1) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(hs));
                if (KIND == 2) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(hs));
                if (KIND == 3) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(hs));
                if (KIND == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(hs));
                if (KIND == 5) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(hs));
                if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "v"(s));
                if (KIND == 7) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "v"(s));
                if (KIND == 8) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(hs));
                if (KIND == 9) asm volatile("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(hs));
                if (KIND == 10) asm volatile("v_pk_lshlrev_b16 %0, 3, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 11) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                if (KIND == 12) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 15]));
                if (KIND == 13) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 15]));
                if (KIND == 14) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]), "v"(hs));
                if (KIND == 15) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(hs));
                if (KIND == 16) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(s), "v"(a[i]));
                if (KIND == 17) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*reinterpret_cast<double*>(&a[(i & 7) * 2])) : "v"(*reinterpret_cast<double*>(&u[(i & 7) * 2])), "v"(*reinterpret_cast<double*>(&u[((i + 1) & 7) * 2])));
                if (KIND == 18) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
                if (KIND == 19) asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(hs));   // (broadcast low half of src1)
                if (KIND == 20) asm volatile("v_and_b32 %0, 0x7fff7fff, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (KIND == 21) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(hs), "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND> void run(const char* name, int wpb = 4) {
    float* out; hipMalloc(&out, 1 << 24);
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, wpb * 64>>>(out, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<blocks, wpb * 64>>>(out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * (double)wpb, inst = waves * iters * 64;
    printf("%-36s %.3f ms   ns per wave-instruction per SIMD %.2f\n", name, ms, 1024 * (ms * 1e6) / inst);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f16"); run<19>("v_pk_fma_f16 op_sel bcast"); run<2>("v_pk_mul_f16"); run<3>("v_pk_add_f16");
    run<4>("v_dot2c_f32_f16"); run<5>("v_dot2_f32_f16 (vop3p)"); run<15>("v_dot2c_f32_bf16"); run<6>("v_fma_mix_f32 lo"); run<7>("v_fma_mix_f32 hi");
    run<16>("v_fma_mixlo_f16"); run<17>("v_pk_fma_f32"); run<8>("v_pk_max_u16"); run<9>("v_pk_sub_u16 clamp"); run<10>("v_pk_lshlrev_b16");
    run<11>("v_cvt_f32_f16"); run<12>("v_cvt_pk_f16_f32"); run<13>("v_cvt_pkrtz_f16_f32"); run<14>("v_perm_b32"); run<18>("v_alignbit_b32");
    run<20>("v_and_b32 literal"); run<21>("v_bfi_b32");
    return 0;
}
