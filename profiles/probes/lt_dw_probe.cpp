// hipBLASLt on the weight-gradient shapes: C[n1,n2] (fp32 row-major) += Y[R,n1]^T * X[R,n2], R = 102080
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("fail %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(1); } } while (0)
int main() {
    const int64_t R = 16 * 6380;
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    void* ws; size_t wsz = (size_t)(getenv("WSZ_MB") ? atoi(getenv("WSZ_MB")) : 256) << 20; CK(hipMalloc(&ws, wsz));
    struct S { int n1, n2; } shapes[] = {{256, 256}, {768, 256}, {1024, 256}, {256, 1024}, {1280, 256}};
    for (auto s : shapes) {
        const int64_t n1 = s.n1, n2 = s.n2;
        void *Y, *X, *C;
        CK(hipMalloc(&Y, R * n1 * 2)); CK(hipMalloc(&X, R * n2 * 2)); CK(hipMalloc(&C, n1 * n2 * 4));
        CK(hipMemset(Y, 0, R * n1 * 2)); CK(hipMemset(X, 0, R * n2 * 2)); CK(hipMemset(C, 0, n1 * n2 * 4));
        hipblasLtMatmulDesc_t md; CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opn = HIPBLAS_OP_N, opt = HIPBLAS_OP_T;
        CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
        CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &opt, sizeof(opt)));
        float* bg; CK(hipMalloc(&bg, 4096 * 4));
        if (getenv("BGRAD")) {
            hipblasLtEpilogue_t ep = atoi(getenv("BGRAD")) == 1 ? HIPBLASLT_EPILOGUE_BGRADB : HIPBLASLT_EPILOGUE_BGRADA;
            CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
            CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bg, sizeof(bg)));
            int32_t bt = HIP_R_32F;
            CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
        }
        hipblasLtMatrixLayout_t la, lb, lc;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, n2, R, n2));      // X' (n2 x R)
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, n1, R, n1));      // Y' (n1 x R), transposed
        CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, n2, n1, n2));
        hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
        hipblasLtMatmulHeuristicResult_t res[16]; int nres = 0;
        { auto st_ = hipblasLtMatmulAlgoGetHeuristic(h, md, la, lb, lc, lc, pref, 16, res, &nres); if (st_ != 0) { printf("n1=%d n2=%d: heuristic status %d\n", s.n1, s.n2, (int)st_); continue; } }
        float alpha = 1.f, beta = 1.f;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9; int bi = -1;
        printf("n1=%4d n2=%4d:", s.n1, s.n2);
        for (int i = 0; i < nres; ++i) {
            for (int r = 0; r < 3; ++r) CK(hipblasLtMatmul(h, md, &alpha, X, la, Y, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 20; ++r) CK(hipblasLtMatmul(h, md, &alpha, X, la, Y, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, 0));
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms / 20 < best) { best = ms / 20; bi = i; }
            printf(" %.1f(ws %zu MB)", ms / 20 * 1e3, res[i].workspaceSize >> 20);
        }
        printf("   best of %d: %.1f us (#%d)  floor@6TB/s %.1f us\n", nres, best * 1e3, bi, (R * (n1 + n2) * 2) / 6e6);
        hipFree(Y); hipFree(X); hipFree(C);
    }
    return 0;
}
