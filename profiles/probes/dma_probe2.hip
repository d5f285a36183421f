// Does an L2-hit stream wait behind an HBM-miss stream of OTHER waves of the same CU?  Waves 0-3 stream a region read once
// (HBM), waves 4-7 re-read a shared 512 KB buffer (L2), both by LDS-DMA with NI x 2 instructions in flight per wave, no
// barrier between the roles; each role's own finish time is taken with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void dma16(uint32_t voff, const void* base, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
template <int NI, int ROLES /* 1: only HBM waves run, 2: only L2 waves, 3: both */, bool PLAIN>
__global__ __launch_bounds__(512) void k(const char* __restrict__ shared_buf, const char* __restrict__ stream_buf, unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const bool hbm = wave < 4;
    if ((hbm && !(ROLES & 1)) || (!hbm && !(ROLES & 2))) return;
    const size_t per_wave = (size_t)iters * NI * 1024;
    const char* sb = stream_buf + ((size_t)blockIdx.x * 4 + (wave & 3)) * per_wave;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float4 a0 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const char* base = hbm ? sb + (size_t)(it * NI + q) * 1024 : shared_buf + (size_t)((((it * NI + q) * 4 + (wave & 3)) * 1024) & (512 * 1024 - 1));
            if (PLAIN) { const float4 v = *reinterpret_cast<const float4*>(base + lane * 16); a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
            else dma16((uint32_t)(lane * 16), base, lds0 + wave * 16384 + ((it & 1) * NI + q) * 1024);
        }
        if (!PLAIN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    }
    if (!PLAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0 + (a0.x + a0.y + a0.z + a0.w == 12345.678f ? 1 : 0);
}
template <int NI, int ROLES, bool PLAIN> void run(const char* name, const char* sh, const char* st, unsigned long long* out, int iters) {
    auto kern = k<NI, ROLES, PLAIN>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256, 512, 8 * 16384>>>(sh, st, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemset(out, 0, 256 * 8 * 8);
    (void)hipEventRecord(e0);
    kern<<<256, 512, 8 * 16384>>>(sh, st, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double th = 0, tl = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? th : tl) += (double)h[b * 8 + w] / 1024.0;
    const double bytes_role = 256.0 * 4 * iters * NI * 1024;
    printf("%-34s NI=%d kernel %7.1f us | HBM waves: %8.0f cyc avg -> %6.1f GB/s/CU @2.1GHz | L2 waves: %8.0f cyc avg -> %6.1f GB/s/CU\n", name, NI, ms * 1e3,
           th, th > 0 ? bytes_role / 256 / (th / 2.1e9) / 1e9 : 0.0, tl, tl > 0 ? bytes_role / 256 / (tl / 2.1e9) / 1e9 : 0.0);
}
int main() {
    char *sh, *st; unsigned long long* out;
    const int iters = 96;
    (void)hipMalloc(&sh, 512 * 1024); (void)hipMalloc(&st, (size_t)256 * 4 * iters * 8 * 1024 + (1 << 20)); (void)hipMalloc(&out, 256 * 8 * 8);
    (void)hipMemset(sh, 1, 512 * 1024); (void)hipMemset(st, 1, (size_t)256 * 4 * iters * 8 * 1024);
    run<6, 1, false>("DMA: HBM waves only", sh, st, out, iters);
    run<6, 2, false>("DMA: L2 waves only", sh, st, out, iters);
    run<6, 3, false>("DMA: both roles", sh, st, out, iters);
    run<8, 3, false>("DMA: both roles", sh, st, out, iters);
    run<6, 1, true>("plain loads: HBM waves only", sh, st, out, iters);
    run<6, 2, true>("plain loads: L2 waves only", sh, st, out, iters);
    run<6, 3, true>("plain loads: both roles", sh, st, out, iters);
    return 0;
}
