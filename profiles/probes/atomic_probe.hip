// memory-side atomic rates: global fp32 add vs packed bf16x2 add, flush-like pattern (each lane its own 4-byte word, 64-B runs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short2_t __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void k(float* f, int n_words, int rounds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        const int i = (int)(((int64_t)t + (int64_t)r * 9973 * 64) % n_words);
        if (KIND == 0) atomicAdd(f + i, 1.0f);
        if (KIND == 1) {
            short2_t v = {0x3f80, 0x3f80};
            __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) short2_t*)(f + i), v);
        }
    }
}
template <int KIND> void run(const char* name, float* f, int n_words) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 4096, rounds = 16;
    k<KIND><<<blocks, 256>>>(f, n_words, 2); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(f, n_words, rounds); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * rounds;
    printf("%-28s %8.1f us for %.1f M lane-atomics: %.2f T/s\n", name, ms * 1e3, ops / 1e6, ops / (ms * 1e-3) / 1e12);
}
int main() {
    const int n_words = 26 * 1024 * 1024;         // 104 MB
    float* f; hipMalloc(&f, (size_t)n_words * 4); hipMemset(f, 0, (size_t)n_words * 4);
    run<0>("global_atomic_add_f32", f, n_words);
    run<1>("global_atomic_pk_add_bf16", f, n_words);
    uint32_t h[4]; hipMemcpy(h, f, 16, hipMemcpyDeviceToHost);
    printf("word0 = %08x\n", h[0]);
    return 0;
}
