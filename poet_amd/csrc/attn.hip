// Small-Q multi-head self-attention core (decoder self-attention over <= 128 object queries: `--num_queries`, main.py:98;
// reference: nn.MultiheadAttention at models/deformable_transformer.py:253,277-278, called with
// q = k = tgt + query_pos, v = tgt, NO key-padding mask: dummy queries attend and are attended).
// One workgroup per (image, head) -- one wave up to 64 queries (every BASELINE.json configuration), two waves up to 128: Q x Q
// scores live in LDS, one lane per query row, fp32 throughout.
// The in/out projections are poet_gemm calls; this kernel is only softmax(q k^T / sqrt(hd)) v with
// dropout on the probabilities (same counter RNG in forward and backward, nothing stored).
#include "common.cuh"

namespace poet {

struct MhaP {
    const float *q, *k, *v, *dout;
    float *out, *dq, *dk, *dv;
    int64_t ld, ld_out, ld_d;
    int N, Q, M;
    float scale;
    uint32_t thresh, seed;
    float dscale;
    const uint32_t* seed_dev;
};

template <int HD, int NT>
__global__ __launch_bounds__(NT) void mha_fwd_kernel(const MhaP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = p.Q;
    constexpr int PH = HD;            // rows are read by all lanes at once (broadcast): no padding, 16-byte LDS reads
    float* sk = smem;                 // [Q][HD]
    float* sv = sk + Q * PH;          // [Q][HD]
    float* sp = sv + Q * PH;          // [Q][Q+1]
    const int n = blockIdx.x / p.M, m = blockIdx.x % p.M;
    const int lane = threadIdx.x;
    const int64_t rbase = (int64_t)n * Q;
    for (int i = lane; i < Q * HD; i += NT) {
        const int r = i / HD, c = i % HD;
        sk[r * PH + c] = p.k[(rbase + r) * p.ld + m * HD + c];
        sv[r * PH + c] = p.v[(rbase + r) * p.ld + m * HD + c];
    }
    __syncthreads();
    if (lane >= Q) return;
    float qr[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) qr[c] = p.q[(rbase + lane) * p.ld + m * HD + c] * p.scale;
    float mx = -3.0e38f;
#pragma unroll 4
    for (int j = 0; j < Q; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += qr[c] * sk[j * PH + c];
        sp[lane * (Q + 1) + j] = s;
        mx = fmaxf(mx, s);
    }
    float sum = 0.f;
#pragma unroll 4
    for (int j = 0; j < Q; ++j) {
        const float e = __expf(sp[lane * (Q + 1) + j] - mx);
        sp[lane * (Q + 1) + j] = e;
        sum += e;
    }
    const float inv = 1.f / sum;
    float acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    const uint32_t ibase = ((uint32_t)blockIdx.x * Q + lane) * Q;
#pragma unroll 4
    for (int j = 0; j < Q; ++j) {
        float pj = sp[lane * (Q + 1) + j] * inv;
        if (p.thresh) pj = drop_keep(p.seed ^ (p.seed_dev ? *p.seed_dev * 0x9E3779B1u : 0u), ibase + j, p.thresh) ? pj * p.dscale : 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] += pj * sv[j * PH + c];
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) p.out[(rbase + lane) * p.ld_out + m * HD + c] = acc[c];
}

// OVERLAY (two waves, Q > 64): the row-phase operands (k, v) and the column-phase operands (scaled q, d(out)) SHARE their LDS --
// 2 Q hd + 2 Q (Q + 1) floats = 148 KB at Q = 128, hd = 16 instead of 165 KB; a lane reads its own q / d(out) row from memory
// SWZ (Q a multiple of 64 whose padded score matrices would not fit: Q = 128 at head dim 32 needs 164 864 B with the Q + 1 pitch,
// 163 840 = all of the CU's LDS without): pitch Q and the column index XORed with the row's low 6 bits instead of the pad -- a lane's
// row walk (lane * Q + (j ^ lane)) and the column phase's walk down a column (i * Q + (lane ^ i)) both stay on 64 different banks.
template <int HD, int NT, bool OVERLAY, bool SWZ = false>
__global__ __launch_bounds__(NT) void mha_bwd_kernel(const MhaP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = p.Q, PQ = SWZ ? Q : Q + 1;
    auto at = [PQ](int r, int c) __attribute__((always_inline)) { return r * PQ + (SWZ ? (c ^ (r & 63)) : c); };
    constexpr int PH = HD;            // rows are read by all lanes at once (broadcast): no padding, 16-byte LDS reads
    float* sq = smem;                 // scaled q
    float* sk = OVERLAY ? smem : sq + Q * PH;
    float* sv = sk + Q * PH;
    float* sdo = OVERLAY ? sv : sv + Q * PH;
    float* spd = sdo + Q * PH;        // dropped probabilities  [Q][Q+1]
    float* sds = spd + Q * PQ;        // dS                      [Q][Q+1]
    const int n = blockIdx.x / p.M, m = blockIdx.x % p.M;
    const int lane = threadIdx.x;
    const int64_t rbase = (int64_t)n * Q;
    for (int i = lane; i < Q * HD; i += NT) {
        const int r = i / HD, c = i % HD;
        if constexpr (!OVERLAY) {
            sq[r * PH + c] = p.q[(rbase + r) * p.ld + m * HD + c] * p.scale;
            sdo[r * PH + c] = p.dout[(rbase + r) * p.ld_out + m * HD + c];
        }
        sk[r * PH + c] = p.k[(rbase + r) * p.ld + m * HD + c];
        sv[r * PH + c] = p.v[(rbase + r) * p.ld + m * HD + c];
    }
    __syncthreads();
    if (lane < Q) {
        float qr[HD], dor[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            if constexpr (OVERLAY) {
                qr[c] = p.q[(rbase + lane) * p.ld + m * HD + c] * p.scale;
                dor[c] = p.dout[(rbase + lane) * p.ld_out + m * HD + c];
            } else {
                qr[c] = sq[lane * PH + c]; dor[c] = sdo[lane * PH + c];
            }
        }
        float mx = -3.0e38f;
    #pragma unroll 4
    for (int j = 0; j < Q; ++j) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) s += qr[c] * sk[j * PH + c];
            spd[at(lane, j)] = s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
    #pragma unroll 4
    for (int j = 0; j < Q; ++j) {
            const float e = __expf(spd[at(lane, j)] - mx);
            spd[at(lane, j)] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        const uint32_t ibase = ((uint32_t)blockIdx.x * Q + lane) * Q;
        float dot = 0.f;
    #pragma unroll 4
    for (int j = 0; j < Q; ++j) {
            const float pj = spd[at(lane, j)] * inv;
            float keep = 1.f;
            if (p.thresh) keep = drop_keep(p.seed ^ (p.seed_dev ? *p.seed_dev * 0x9E3779B1u : 0u), ibase + j, p.thresh) ? p.dscale : 0.f;
            float dpd = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) dpd += dor[c] * sv[j * PH + c];
            const float dp = dpd * keep;
            spd[at(lane, j)] = pj * keep;
            sds[at(lane, j)] = dp;            // temporarily dP
            dot += pj * dp;
        }
        float dqr[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dqr[c] = 0.f;
    #pragma unroll 4
    for (int j = 0; j < Q; ++j) {
            // recover p_j from the dropped value is not possible when keep == 0, so recompute it
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) s += qr[c] * sk[j * PH + c];
            const float pj = __expf(s - mx) * inv;
            const float ds = pj * (sds[at(lane, j)] - dot);
            sds[at(lane, j)] = ds;
#pragma unroll
            for (int c = 0; c < HD; ++c) dqr[c] += ds * sk[j * PH + c];
        }
#pragma unroll
        for (int c = 0; c < HD; ++c) p.dq[(rbase + lane) * p.ld_d + m * HD + c] = dqr[c] * p.scale;
    }
    __syncthreads();
    if constexpr (OVERLAY) {          // every reader of k / v is past the barrier: their LDS now takes scaled q / d(out)
        for (int i = lane; i < Q * HD; i += NT) {
            const int r = i / HD, c = i % HD;
            sq[r * PH + c] = p.q[(rbase + r) * p.ld + m * HD + c] * p.scale;
            sdo[r * PH + c] = p.dout[(rbase + r) * p.ld_out + m * HD + c];
        }
        __syncthreads();
    }
    if (lane < Q) {
        float dkr[HD], dvr[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) { dkr[c] = 0.f; dvr[c] = 0.f; }
#pragma unroll 4
        for (int i = 0; i < Q; ++i) {
            const float ds = sds[at(i, lane)], pd = spd[at(i, lane)];
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                dkr[c] += ds * sq[i * PH + c];      // sq already carries the 1/sqrt(hd) scale
                dvr[c] += pd * sdo[i * PH + c];
            }
        }
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            p.dk[(rbase + lane) * p.ld_d + m * HD + c] = dkr[c];
            p.dv[(rbase + lane) * p.ld_d + m * HD + c] = dvr[c];
        }
    }
}

constexpr size_t MHA_LDS_MAX = 160 * 1024;
// dynamic LDS beyond the 64 KiB default needs the function attribute (set per kernel whenever a launch asks for more)
static void mha_lds_attr(const void* fn, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MHA_LDS_MAX);
}

static int mha_common(MhaP& p, int N, int Q, int M, int hd, float drop_p, uint32_t seed, const uint32_t* seed_dev) {
    POET_CHECK(N > 0 && M > 0 && Q > 0 && Q <= 128, POET_ERR_UNSUPPORTED, "mha: Q=%d must be in 1..128", Q);
    POET_CHECK(hd == 16 || hd == 32 || hd == 64, POET_ERR_UNSUPPORTED, "mha: head dim %d not in {16,32,64}", hd);
    POET_CHECK(drop_p >= 0.f && drop_p < 1.f, POET_ERR_ARG, "mha: drop_p");
    p.N = N; p.Q = Q; p.M = M;
    p.scale = 1.f / sqrtf((float)hd);
    p.thresh = drop_p > 0.f ? drop_thresh(drop_p) : 0u;
    p.dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    p.seed = seed;
    p.seed_dev = seed_dev;
    return POET_OK;
}

}  // namespace poet

using namespace poet;

extern "C" int poet_mha_fwd(const float* q, const float* k, const float* v, int64_t ld, float* out, int64_t ld_out,
                            int N, int Q, int M, int hd, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    MhaP p{};
    int rc = mha_common(p, N, Q, M, hd, drop_p, seed, seed_dev);
    if (rc) return rc;
    POET_CHECK(q && k && v && out, POET_ERR_ARG, "mha_fwd: null pointer");
    p.q = q; p.k = k; p.v = v; p.out = out; p.ld = ld; p.ld_out = ld_out;
    const size_t lds = sizeof(float) * (2 * Q * hd + Q * (Q + 1));
    POET_CHECK(lds <= MHA_LDS_MAX, POET_ERR_UNSUPPORTED, "mha_fwd: Q=%d at head dim %d needs %zu bytes of LDS (> 160 KiB)", Q, hd, lds);
    dim3 grid(N * M);
    hipStream_t st = (hipStream_t)stream;
#define POET_MHA_FWD(HD, NT) do { mha_lds_attr(reinterpret_cast<const void*>(mha_fwd_kernel<HD, NT>), lds);        \
                                  hipLaunchKernelGGL((mha_fwd_kernel<HD, NT>), grid, dim3(NT), lds, st, p); } while (0)
    if (Q <= 64) { if (hd == 16) POET_MHA_FWD(16, 64); else if (hd == 32) POET_MHA_FWD(32, 64); else POET_MHA_FWD(64, 64); }
    else { if (hd == 16) POET_MHA_FWD(16, 128); else if (hd == 32) POET_MHA_FWD(32, 128); else POET_MHA_FWD(64, 128); }
#undef POET_MHA_FWD
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_mha_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* dout, int64_t ld_out,
                            float* dq, float* dk, float* dv, int64_t ld_d, int N, int Q, int M, int hd, float drop_p,
                            uint32_t seed, const uint32_t* seed_dev, void* stream) {
    MhaP p{};
    int rc = mha_common(p, N, Q, M, hd, drop_p, seed, seed_dev);
    if (rc) return rc;
    POET_CHECK(q && k && v && dout && dq && dk && dv, POET_ERR_ARG, "mha_bwd: null pointer");
    p.q = q; p.k = k; p.v = v; p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv; p.ld = ld; p.ld_out = ld_out; p.ld_d = ld_d;
    const bool two = Q > 64;                                            // two waves: k / v and q / d(out) share their LDS (OVERLAY)
    size_t lds = sizeof(float) * ((two ? 2 : 4) * Q * hd + 2 * Q * (Q + 1));
    // swizzled score matrices (no pad column) where the padded ones do not fit: Q = 128 at head dim 32
    const bool swz = two && lds > MHA_LDS_MAX && hd == 32 && Q % 64 == 0 && sizeof(float) * (2 * Q * hd + 2 * Q * Q) <= MHA_LDS_MAX;
    if (swz) lds = sizeof(float) * (2 * Q * hd + 2 * Q * Q);
    POET_CHECK(lds <= MHA_LDS_MAX, POET_ERR_UNSUPPORTED,
               "mha_bwd: Q=%d at head dim %d needs %zu bytes of LDS (> 160 KiB; limits: Q <= 128 at head dim 16 / 32, Q <= 114 at head dim 64)", Q, hd, lds);
    dim3 grid(N * M);
    hipStream_t st = (hipStream_t)stream;
#define POET_MHA_BWD(HD, NT, OV) do { mha_lds_attr(reinterpret_cast<const void*>(mha_bwd_kernel<HD, NT, OV>), lds);   \
                                      hipLaunchKernelGGL((mha_bwd_kernel<HD, NT, OV>), grid, dim3(NT), lds, st, p); } while (0)
    if (!two) { if (hd == 16) POET_MHA_BWD(16, 64, false); else if (hd == 32) POET_MHA_BWD(32, 64, false); else POET_MHA_BWD(64, 64, false); }
    else if (swz) {
        mha_lds_attr(reinterpret_cast<const void*>(mha_bwd_kernel<32, 128, true, true>), lds);
        hipLaunchKernelGGL((mha_bwd_kernel<32, 128, true, true>), grid, dim3(128), lds, st, p);
    }
    else { if (hd == 16) POET_MHA_BWD(16, 128, true); else if (hd == 32) POET_MHA_BWD(32, 128, true); else POET_MHA_BWD(64, 128, true); }
#undef POET_MHA_BWD
    POET_LAUNCH_CHECK();
    return POET_OK;
}
