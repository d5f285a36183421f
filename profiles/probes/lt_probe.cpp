// hipBLASLt on the tall-skinny input-gradient shapes: C[M,N] (fp32, row-major) = A[M,K] (bf16) * W[K,N] (bf16) + beta * C
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("fail %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(1); } } while (0)
int main() {
    const int64_t M = 16 * 6380;
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    void* ws; size_t wsz = 64 << 20; CK(hipMalloc(&ws, wsz));
    struct S { int K, N; float beta; int cf32; } shapes[] = {{1024, 256, 1.f, 1}, {1024, 256, 0.f, 1}, {1024, 256, 0.f, 0}, {768, 256, 1.f, 1}, {1280, 256, 0.f, 1}, {256, 256, 1.f, 1}, {256, 1024, 0.f, 0}};
    for (auto s : shapes) {
        const int64_t K = s.K, N = s.N;
        void *A, *W, *C;
        CK(hipMalloc(&A, M * K * 2)); CK(hipMalloc(&W, K * N * 2)); CK(hipMalloc(&C, M * N * 4));
        CK(hipMemset(A, 0, M * K * 2)); CK(hipMemset(W, 0, K * N * 2)); CK(hipMemset(C, 0, M * N * 4));
        // column-major view: C_cm (N x M, ld N) = W_cm (N x K, ld N) * A_cm (K x M, ld K)
        hipblasLtMatmulDesc_t md; CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opn = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
        CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
        hipblasLtMatrixLayout_t la, lb, lc;
        const hipDataType ct = s.cf32 ? HIP_R_32F : HIP_R_16BF;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, N, K, N));
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));
        CK(hipblasLtMatrixLayoutCreate(&lc, ct, N, M, N));
        hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
        hipblasLtMatmulHeuristicResult_t res[8]; int nres = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, md, la, lb, lc, lc, pref, 8, res, &nres));
        float alpha = 1.f, beta = s.beta;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9; int bi = -1;
        for (int i = 0; i < nres; ++i) {
            for (int r = 0; r < 3; ++r) CK(hipblasLtMatmul(h, md, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 20; ++r) CK(hipblasLtMatmul(h, md, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0));
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms / 20 < best) { best = ms / 20; bi = i; }
            if (i == 0) printf("K=%4d N=%4d beta=%.0f C=%s: heuristic #0 %.1f us", s.K, s.N, s.beta, s.cf32 ? "f32" : "bf16", ms / 20 * 1e3);
        }
        printf("   best of %d: %.1f us (#%d)\n", nres, best * 1e3, bi);
        hipFree(A); hipFree(W); hipFree(C);
    }
    return 0;
}
