// Shared between gemm.hip (generic tiled kernel + dispatch) and gemm_ws.hip (weight-stationary streaming kernel).
#pragma once
#include "common.cuh"

namespace poet {

struct GemmK {
    PoetGemmDesc d;
    int kchunk;
    int a_vec, b_vec, c_vec;
    uint32_t drop_thresh;
    float drop_scale;
};

// launches the streaming kernel and returns true when the problem is one it handles; false = use the generic kernel
bool gemm_ws_try(const GemmK& p, hipStream_t st);
#ifdef POET_PROBE_KERNELS
// register-stationary kernel for the wide (N % 256 == 0, N >= 512) split-weight forward products with K = 256: measured equal to
// gemm_ws, lives in profiles/probes/kernels/gemm_wr.hip and is compiled only into probe builds (POET_BUILD_PROBES=1)
bool gemm_wr_try(const GemmK& p, hipStream_t st);
#endif
// same contract for the weight-gradient kernels: the DMA-ring form (gemm_dwr.hip: 256 x 128 tiles, wide shapes) is asked first, then
// the register-staged one (gemm_dw.hip)
bool gemm_dwr_try(const GemmK& p, hipStream_t st);
bool gemm_dw_try(const GemmK& p, hipStream_t st);
// plain tall-skinny bf16 x bf16 -> fp32 (+=) products with N = 256, K >= 512: the deep-pipeline kernel (gemm_pipe.hip)
bool gemm_pipe_try(const GemmK& p, hipStream_t st);
// latency-oriented fp32 kernel for the <= 1024-row decoder / head Linears (gemm_small.hip)
bool gemm_small_try(const GemmK& p, hipStream_t st);
// dW[i] += Y[i]^T X[i] (+ column sums of Y[i]) for n <= 8 fp32 problems of one shape, rows <= 1024, in one launch
bool gemm_small_dw_multi(const float* const* Y, const float* const* X, float* const* C, float* const* ysum, int nl, int n, const int* n_out,
                         const int* k_in, int rows, const int64_t* ldy, const int64_t* ldx, hipStream_t st);
bool gemm_small_dw_list(const float* const* Y, const float* const* X, float* const* C, float* const* ysum, int n, int n_out, int k_in,
                        int rows, int64_t ldy, int64_t ldx, int64_t ldc, hipStream_t st);

}  // namespace poet
