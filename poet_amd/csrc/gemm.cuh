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
// same contract for the weight-gradient kernel (gemm_dw.hip)
bool gemm_dw_try(const GemmK& p, hipStream_t st);
// latency-oriented fp32 kernel for the <= 1024-row decoder / head Linears (gemm_small.hip)
bool gemm_small_try(const GemmK& p, hipStream_t st);

}  // namespace poet
