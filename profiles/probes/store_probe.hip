// How fast can a wave-per-32-rows kernel stream A (M x 256 bf16) in and C (M x 1024 bf16) out, with the store pattern of the
// weight-stationary GEMM epilogue (lane = row l&15, 16-B piece (l>>4) of a 64-B run) against fully coalesced rows?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int PATTERN, int NCOL>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ A, uint4* __restrict__ C, int M) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int frow = lane & 15, g = lane >> 4;
    const int waves = gridDim.x * 8, wv = blockIdx.x * 8 + wid;
    const int U = M / 16;
    for (int u = wv * 2; u + 1 < U; u += waves * 2) {
        uint4 acc = make_uint4(0, 0, 0, 0);
        for (int fm = 0; fm < 2; ++fm) {
            const uint4* p = A + (int64_t)((u + fm) * 16 + frow) * 32 + g;      // row of 256 bf16 = 32 uint4; lane takes chunk g + 4 kk
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) { uint4 v = p[kk * 4]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
        }
        constexpr int ROWV = NCOL / 8;                                            // uint4 per output row
        for (int fm = 0; fm < 2; ++fm) {
            if (PATTERN == 0) {          // ws epilogue: per 64-column group two runs of 8 columns per lane: [8g, 8g+8) and [32+8g, ...)
                uint4* q = C + (int64_t)((u + fm) * 16 + frow) * ROWV;
#pragma unroll
                for (int jq = 0; jq < NCOL / 64; ++jq)
#pragma unroll
                    for (int h = 0; h < 2; ++h) q[jq * 8 + h * 4 + g] = acc;
            } else {                     // coalesced: a wave instruction writes 1 KB contiguous
#pragma unroll
                for (int i = 0; i < 16 * ROWV / 64; ++i) {
                    C[(int64_t)(u + fm) * 16 * ROWV + i * 64 + lane] = acc;
                }
            }
        }
    }
}
template <int PATTERN, int NCOL> void run(const char* name, const uint4* A, uint4* C, int M) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<PATTERN, NCOL><<<256, 512>>>(A, C, M);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<PATTERN, NCOL><<<256, 512>>>(A, C, M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)M * 512 + (double)M * NCOL * 2;
    printf("%-44s %7.1f us   %.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const int M = 16 * 6380;
    uint4 *A, *C;
    hipMalloc(&A, (size_t)M * 512); hipMalloc(&C, (size_t)M * 2048);
    hipMemset(A, 1, (size_t)M * 512);
    run<0, 1024>("N=1024 ws-epilogue store pattern", A, C, M);
    run<1, 1024>("N=1024 coalesced rows", A, C, M);
    run<0, 256>("N=256 ws-epilogue store pattern", A, C, M);
    run<1, 256>("N=256 coalesced rows", A, C, M);
    return 0;
}
