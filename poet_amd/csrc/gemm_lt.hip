// Plain tall-skinny GEMMs through hipBLASLt.
//
// The hand-written kernels of this library exist for the FUSED products of the path (bias / ReLU / dropout / gate / row mask /
// head-major scatter epilogues, split-bf16 weights, the transposed-operand weight gradient).  A handful of products of the
// encoder's backward are plain GEMMs,
//     C[M, N] (fp32) (+)= A[M, K] (bf16) * W[K, N] (bf16),   M ~ 1e5 rows, N = 256, K = 768 / 1024 / 1280
// (d(src) += d(hidden) W1, d(src) += [d(offsets|logits) | d(value) rows] [W_so ; W_aw ; W_v], d(memory) = d(values) W_v:
// models/deformable_transformer.py:193-208's Linears seen from backward), and for those the vendor library is simply
// faster than the tiled kernel of gemm.hip: 75 us against 148 us at K = 1024 with beta = 1 (measured on MI355X,
// profiles/probes/lt_probe.cpp / profiles/probes/dx_bench.py).  They go to hipblasLtMatmul with an algorithm picked once per shape by the library's
// own heuristic; everything else stays on the kernels of this directory.  POET_GEMM_NO_LT=1 disables the route (A/B).
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "gemm.cuh"

namespace poet {

namespace {

struct LtPlan {
    hipblasLtMatmulDesc_t md = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};

struct LtState {
    hipblasLtHandle_t handle = nullptr;
    bool failed = false;
    std::mutex mu;
    // (M, N, K, lda, ldb, ldc, b_kmajor, ws, bias pointer): a descriptor per bias vector -- re-pointing a cached descriptor at
    // another bias after its algorithm was chosen did not take effect (measured: every layer got the first layer's bias)
    std::map<std::tuple<int, int, int, int64_t, int64_t, int64_t, int, size_t, uintptr_t>, LtPlan> plans;
};

LtState& lt_state() {
    static LtState s;
    return s;
}

// Row-major C[M,N] = A[M,K] * B  is the column-major  C^T (N x M, ld ldc) = op(B') (N x K) * A^T (K x M, ld lda), where the
// weight is either W[K][N] row-major (column-major N x K, ld ldb: no transpose) or W[N][K] row-major (column-major K x N:
// transposed).
LtPlan make_plan(LtState& s, const PoetGemmDesc& d, size_t ws_bytes) {
    LtPlan pl;
    if (hipblasLtMatmulDescCreate(&pl.md, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return pl;
    const hipblasOperation_t opa = d.b_kmajor ? HIPBLAS_OP_N : HIPBLAS_OP_T, opb = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa));
    hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb));
    if (d.bias) {                                               // fp32 [N] = one value per ROW of the column-major result (N x M)
        const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
        const hipDataType bt = HIP_R_32F;
        if (hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)) != HIPBLAS_STATUS_SUCCESS) return pl;
        if (hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS) return pl;
        const void* bp = d.bias;
        if (hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS) return pl;
    }
    hipblasStatus_t st;
    if (d.b_kmajor) st = hipblasLtMatrixLayoutCreate(&pl.la, HIP_R_16BF, d.N, d.K, d.ldb);
    else st = hipblasLtMatrixLayoutCreate(&pl.la, HIP_R_16BF, d.K, d.N, d.ldb);
    if (st != HIPBLAS_STATUS_SUCCESS) return pl;
    if (hipblasLtMatrixLayoutCreate(&pl.lb, HIP_R_16BF, d.K, d.M, d.lda) != HIPBLAS_STATUS_SUCCESS) return pl;
    if (hipblasLtMatrixLayoutCreate(&pl.lc, HIP_R_32F, d.N, d.M, d.ldc) != HIPBLAS_STATUS_SUCCESS) return pl;
    hipblasLtMatmulPreference_t pref = nullptr;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return pl;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes));
    hipblasLtMatmulHeuristicResult_t res[1];
    int n = 0;
    st = hipblasLtMatmulAlgoGetHeuristic(s.handle, pl.md, pl.la, pl.lb, pl.lc, pl.lc, pref, 1, res, &n);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || n < 1 || res[0].state != HIPBLAS_STATUS_SUCCESS || res[0].workspaceSize > ws_bytes) return pl;
    pl.algo = res[0].algo;
    pl.ws = res[0].workspaceSize;
    pl.ok = true;
    return pl;
}

}  // namespace

bool gemm_lt_try(const GemmK& p, hipStream_t st) {
    const PoetGemmDesc& d = p.d;
    static const int disabled = [] { const char* e = getenv("POET_GEMM_NO_LT"); return e && atoi(e) ? 1 : 0; }();
    if (disabled) return false;
    // plain products only: bf16 operands, fp32 result written or accumulated in place, nothing fused
    if (d.a_dtype != POET_BF16 || d.b_dtype != POET_BF16 || d.c_dtype != POET_F32 || d.compute != POET_BF16) return false;
    if (d.a_kmajor || d.batch != 1 || d.splitk != 1 || d.atomic || d.A2 || d.b_split) return false;
    if (d.act || d.gate_ref || d.row_mask || d.drop_p != 0.f || d.out_mode != 0 || d.alpha != 1.f) return false;
    if (d.bias && d.add_src) return false;                              // (bias only on the writing form)
    if (d.add_src && (d.add_src != d.C || d.ld_add != d.ldc)) return false;
    if (d.M < 4096 || d.K < 512 || d.N % 16 != 0 || d.K % 16 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B) | reinterpret_cast<uintptr_t>(d.C)) & 15) return false;
    if ((d.lda & 7) || (d.ldb & 7) || (d.ldc & 3)) return false;

    LtState& s = lt_state();
    LtPlan pl;
    {
        std::lock_guard<std::mutex> lock(s.mu);
        if (s.failed) return false;
        if (!s.handle && hipblasLtCreate(&s.handle) != HIPBLAS_STATUS_SUCCESS) { s.failed = true; return false; }
        const size_t ws_bytes = d.workspace && (reinterpret_cast<uintptr_t>(d.workspace) & 255) == 0 ? (size_t)d.workspace_bytes : 0;
        const auto key = std::make_tuple(d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.b_kmajor, ws_bytes, reinterpret_cast<uintptr_t>(d.bias));
        auto it = s.plans.find(key);
        if (it == s.plans.end()) it = s.plans.emplace(key, make_plan(s, d, ws_bytes)).first;
        pl = it->second;
    }
    if (!pl.ok) return false;
    const float alpha = 1.f, beta = d.add_src ? 1.f : 0.f;
    const hipblasStatus_t rc = hipblasLtMatmul(s.handle, pl.md, &alpha, d.B, pl.la, d.A, pl.lb, &beta, d.C, pl.lc, d.C, pl.lc, &pl.algo,
                                               pl.ws ? d.workspace : nullptr, pl.ws, st);
    return rc == HIPBLAS_STATUS_SUCCESS;
}

}  // namespace poet
