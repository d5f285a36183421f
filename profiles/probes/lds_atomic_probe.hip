// rates of LDS atomic flavours on gfx950 (wave-instructions per clock per CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += 1024) reinterpret_cast<int*>(lds)[i] = 0;
    __syncthreads();
    // 16 lanes = one "pixel" of 16 channels; rows of a wave hit different pixels; address pattern pseudo-random per iter
    const int c = tid & 15, row = tid >> 4;
    unsigned h = row * 2654435761u;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const int pix = (h >> 8) & 1023;
        if (MODE == 0) atomicAdd(reinterpret_cast<int*>(lds) + pix * 16 + c, it);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + pix * 16 + c, (unsigned long long)it);
        else if (MODE == 2) atomicAdd(reinterpret_cast<float*>(lds) + pix * 16 + c, 1.0f);
        else if (MODE == 3) { if (c < 8) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + pix * 8 + c, (unsigned long long)it); }
        else if (MODE == 4) { reinterpret_cast<int*>(lds)[pix * 16 + c] += it; }     // plain RMW (racy; rate reference)
        else if (MODE == 5) { asm volatile("ds_pk_add_bf16 %0, %1" :: "v"((pix * 16 + c) * 4), "v"(0x3f803f80) : "memory"); }
        else if (MODE == 6) { asm volatile("ds_add_f64 %0, %1" :: "v"((pix * 16 + c) * 8), "v"(1.0) : "memory"); }
    }
    __syncthreads();
    long long t1 = clock64();
    if (tid == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(t1 - t0);
    if (tid == 1 && blockIdx.x == 0) out[1] = reinterpret_cast<int*>(lds)[5];
}
template <int MODE> void run(const char* name, unsigned long long* d, int lds) {
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<MODE><<<256, 1024, lds>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 1024, lds>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    // per CU: 16 waves x iters wave-instructions
    double winst = 16.0 * iters;
    printf("%-28s %8.1f us   %6.2f clk(shader clock64 ticks)/wave-instr per CU   %6.2f ns/winstr/CU  (%.3g T lane-ops/s chip)\n", name, ms * 1e3, (double)h[0] / winst,
           ms * 1e6 / winst, 256.0 * winst * 64 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    run<0>("ds_add_u32 (16 rows/wg x64)", d, 131072);
    run<1>("ds_add_u64", d, 131072 * 1 + 0);   // 1024 pix*16*8 = 128 KB
    run<2>("ds_add_f32", d, 131072);
    run<3>("ds_add_u64 half lanes", d, 131072);
    run<4>("plain rmw b32", d, 131072);
    run<5>("ds_pk_add_bf16", d, 131072);
    run<6>("ds_add_f64", d, 131072);
    return 0;
}
