// Store / read-modify-write patterns for an fp32 [M][256] result held in MFMA accumulator layouts (gemm_pipe.hip epilogue):
// which lane -> address map lets a CU drain its tile fastest?  256 workgroups x 8 waves, each workgroup owns 400 rows.
//   S0  transposed-product layout as is: lane (m = l&15, c = l>>4) holds row m, 4 floats at col w*32 + j*16 + c*4: per instruction
//       16 rows x 64 B
//   S1  S0 after a DPP row_ror:8 exchange between the two column fragments: per instruction 8 rows x 128 B (whole lines), the 8
//       lanes of a line are NOT consecutive lane ids
//   S2  8 rows x 128 B with consecutive lanes contiguous (lane l: row l>>3, 16-B chunk l&7) -- needs an LDS transpose in a kernel
//   S3  one whole 1 KB row per instruction (lane l: col 4l) -- needs an LDS transpose and a different wave->column ownership
//   S4  non-transposed product layout: lane (n = l&15, g = l>>4) holds col w*32 + j*16 + n of rows g*4 + t: dword stores, 16
//       consecutive lanes = 64 B of one row, 4 rows per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PAT, bool RMW>
__global__ __launch_bounds__(512) void k(float* __restrict__ C, int M) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m16 = lane & 15, c = lane >> 4;
    const int U = M / 16, G = gridDim.x, g = blockIdx.x;
    const int nu = U / G + (g < U % G), u0 = g * (U / G) + min(g, U % G);
    for (int u = u0; u < u0 + nu; ++u) {                 // one 16-row unit x 256 columns per iteration: 16 KB
        float* base = C + (int64_t)u * 16 * 256;
        if (PAT == 0 || PAT == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* p = (PAT == 0) ? base + m16 * 256 + w * 32 + j * 16 + c * 4
                                      : base + ((m16 & 7) + 8 * j) * 256 + w * 32 + ((m16 >> 3) * 4 + c) * 4;
                f4 v = {1.f, 2.f, 3.f, (float)lane};
                if (RMW) v += *reinterpret_cast<const f4*>(p);
                *reinterpret_cast<f4*>(p) = v;
            }
        } else if (PAT == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* p = base + ((lane >> 3) + 8 * j) * 256 + w * 32 + (lane & 7) * 4;
                f4 v = {1.f, 2.f, 3.f, (float)lane};
                if (RMW) v += *reinterpret_cast<const f4*>(p);
                *reinterpret_cast<f4*>(p) = v;
            }
        } else if (PAT == 3) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* p = base + (w * 2 + j) * 256 + lane * 4;
                f4 v = {1.f, 2.f, 3.f, (float)lane};
                if (RMW) v += *reinterpret_cast<const f4*>(p);
                *reinterpret_cast<f4*>(p) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float* p = base + (c * 4 + t) * 256 + w * 32 + j * 16 + m16;
                    float v = (float)lane;
                    if (RMW) v += *p;
                    *p = v;
                }
        }
    }
}
template <int PAT, bool RMW> void run(const char* name, float* C, int M) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<PAT, RMW><<<256, 512>>>(C, M);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<PAT, RMW><<<256, 512>>>(C, M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)M * 1024 * (RMW ? 2 : 1);
    printf("%-28s %-6s %7.1f us   %.2f TB/s\n", name, RMW ? "rmw" : "store", ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const int M = 16 * 6380;
    float* C;
    hipMalloc(&C, (size_t)M * 1024);
    hipMemset(C, 0, (size_t)M * 1024);
    run<0, false>("S0 16 rows x 64 B", C, M);   run<0, true>("S0 16 rows x 64 B", C, M);
    run<1, false>("S1 8 x 128 B (dpp order)", C, M); run<1, true>("S1 8 x 128 B (dpp order)", C, M);
    run<2, false>("S2 8 x 128 B (linear)", C, M);    run<2, true>("S2 8 x 128 B (linear)", C, M);
    run<3, false>("S3 1 KB rows", C, M);             run<3, true>("S3 1 KB rows", C, M);
    run<4, false>("S4 dword, 4 rows x 64 B", C, M);  run<4, true>("S4 dword, 4 rows x 64 B", C, M);
    return 0;
}
