// Per-CU ingest ceilings on gfx950: LDS-DMA (global_load_lds_dwordx4) and plain dwordx4 loads, from an L2-resident 512 KB
// buffer every workgroup re-reads (a GEMM's weight) and from a streamed region read once (its activation).
// 256 workgroups x 512 threads; every wave keeps 2 batches of NI 1-KB instructions in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void dma16(uint32_t voff, const void* base, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
// MODE 0: DMA from the shared 512 KB buffer; 1: DMA from a streamed region; 2: half / half; 3: plain loads from the shared buffer;
// 4: plain loads streamed
template <int MODE, int NI>
__global__ __launch_bounds__(512) void k(const char* __restrict__ shared_buf, const char* __restrict__ stream_buf, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const size_t per_wg = (size_t)iters * NI * 8192;                    // bytes a workgroup streams
    const char* sb = stream_buf + (size_t)blockIdx.x * per_wg;
    float acc = 0.f;
    if (MODE <= 2) {
        int slot = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const bool sh = MODE == 0 || (MODE == 2 && (q & 1));
                const char* base = sh ? shared_buf + (size_t)(((it * NI + q) * 8192) & (512 * 1024 - 1)) : sb + (size_t)(it * NI + q) * 8192;
                dma16((uint32_t)(tid * 16), base, lds0 + slot * (NI * 8192) + q * 8192 + wave * 1024);
            }
            slot = slot == 2 ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<float*>(smem + tid * 4);
    } else {
        float4 a0 = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const char* base = MODE == 3 ? shared_buf + (size_t)(((it * NI + q) * 8192) & (512 * 1024 - 1)) : sb + (size_t)(it * NI + q) * 8192;
                const float4 v = *reinterpret_cast<const float4*>(base + tid * 16);
                a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
            }
        }
        acc = a0.x + a0.y + a0.z + a0.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE, int NI> void run(const char* name, const char* sh, const char* st, float* out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto kern = k<MODE, NI>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NI * 8192);
    for (int i = 0; i < 2; ++i) kern<<<256, 512, 3 * NI * 8192>>>(sh, st, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) kern<<<256, 512, 3 * NI * 8192>>>(sh, st, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * NI * 8192;
    printf("%-44s NI=%d %8.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", name, NI, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12, bytes / 256 / (ms / 10 * 1e-3) / 1e9);
}
int main() {
    char *sh, *st; float* out;
    const int iters = 32;
    (void)hipMalloc(&sh, 512 * 1024); (void)hipMalloc(&st, (size_t)256 * iters * 6 * 8192 + (1 << 20)); (void)hipMalloc(&out, 4);
    (void)hipMemset(sh, 1, 512 * 1024); (void)hipMemset(st, 1, (size_t)256 * iters * 6 * 8192);
    run<0, 6>("DMA, L2-resident 512 KB", sh, st, out, iters);
    run<1, 6>("DMA, streamed (HBM)", sh, st, out, iters);
    run<2, 6>("DMA, half resident / half streamed", sh, st, out, iters);
    run<0, 2>("DMA, L2-resident 512 KB", sh, st, out, iters * 3);
    run<1, 2>("DMA, streamed (HBM)", sh, st, out, iters * 3);
    run<3, 6>("plain loads, L2-resident 512 KB", sh, st, out, iters);
    run<4, 6>("plain loads, streamed (HBM)", sh, st, out, iters);
    return 0;
}
