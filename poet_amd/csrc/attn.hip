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


// ---- generic form: any query count up to MHA_GEN_QMAX, any head dim ---------------------------------------------------------------
// `--num_queries` and `--hidden_dim` / `--nheads` are free parameters of the reference (main.py:94-98) and nn.MultiheadAttention has no
// limit on either; the kernels above hold Q x Q scores in LDS with one lane per query row (Q <= 128, head dim 16 / 32 / 64).  Everything
// else runs here: one workgroup per (image, head), one WAVE per query row (forward, backward row phase) or per key column (backward
// column phase), lanes over the other index, probabilities recomputed in the column phase from the row statistics (max, 1 / sum,
// <p, dP>) kept in LDS -- no Q x Q matrix anywhere, no scratch in memory.  Same dropout counter as above.  Correctness path, not a tuned one.
constexpr int MHA_GEN_QMAX = 2048;
constexpr int MHA_GEN_NT = 256;

__device__ __forceinline__ float gen_dot(const float* __restrict__ a, const float* __restrict__ b, int hd) {
    float s = 0.f;
    for (int c = 0; c < hd; ++c) s += a[c] * b[c];
    return s;
}

__global__ __launch_bounds__(MHA_GEN_NT) void mha_gen_fwd_kernel(const MhaP p, int hd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = p.Q, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sp = smem + wave * Q;                                        // this wave's probability row
    const int n = blockIdx.x / p.M, m = blockIdx.x % p.M;
    const int64_t rbase = (int64_t)n * Q;
    const uint32_t seed = p.seed ^ (p.seed_dev ? *p.seed_dev * 0x9E3779B1u : 0u);
    for (int i = wave; i < Q; i += MHA_GEN_NT / 64) {
        const float* qi = p.q + (rbase + i) * p.ld + m * hd;
        float mx = -3.0e38f;
        for (int j = lane; j < Q; j += 64) {
            const float s = gen_dot(qi, p.k + (rbase + j) * p.ld + m * hd, hd) * p.scale;
            sp[j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Q; j += 64) { const float e = __expf(sp[j] - mx); sp[j] = e; sum += e; }
        const float inv = 1.f / wave_sum(sum);
        const uint32_t ibase = ((uint32_t)blockIdx.x * Q + i) * Q;
        for (int j = lane; j < Q; j += 64) {
            float pj = sp[j] * inv;
            if (p.thresh) pj = drop_keep(seed, ibase + j, p.thresh) ? pj * p.dscale : 0.f;
            sp[j] = pj;
        }
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < hd; c += 64) {
            const float* vc = p.v + rbase * p.ld + m * hd + c;
            float acc = 0.f;
            for (int j = 0; j < Q; ++j) acc += sp[j] * vc[(int64_t)j * p.ld];
            p.out[(rbase + i) * p.ld_out + m * hd + c] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(MHA_GEN_NT) void mha_gen_bwd_kernel(const MhaP p, int hd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Q = p.Q, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* st = smem;                                                   // [Q][3]: row max, 1 / row sum, <p, dP>
    float* sa = st + 3 * Q + wave * 2 * Q;                              // two Q-vectors per wave
    float* sb = sa + Q;
    const int n = blockIdx.x / p.M, m = blockIdx.x % p.M;
    const int64_t rbase = (int64_t)n * Q;
    const uint32_t seed = p.seed ^ (p.seed_dev ? *p.seed_dev * 0x9E3779B1u : 0u);
    // row phase: statistics and d(q)
    for (int i = wave; i < Q; i += MHA_GEN_NT / 64) {
        const float* qi = p.q + (rbase + i) * p.ld + m * hd;
        const float* doi = p.dout + (rbase + i) * p.ld_out + m * hd;
        float mx = -3.0e38f;
        for (int j = lane; j < Q; j += 64) {
            const float s = gen_dot(qi, p.k + (rbase + j) * p.ld + m * hd, hd) * p.scale;
            sa[j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Q; j += 64) { const float e = __expf(sa[j] - mx); sa[j] = e; sum += e; }
        const float inv = 1.f / wave_sum(sum);
        const uint32_t ibase = ((uint32_t)blockIdx.x * Q + i) * Q;
        float dot = 0.f;
        for (int j = lane; j < Q; j += 64) {
            const float pj = sa[j] * inv;
            float keep = 1.f;
            if (p.thresh) keep = drop_keep(seed, ibase + j, p.thresh) ? p.dscale : 0.f;
            const float dp = gen_dot(doi, p.v + (rbase + j) * p.ld + m * hd, hd) * keep;
            sa[j] = pj;
            sb[j] = dp;
            dot += pj * dp;
        }
        dot = wave_sum(dot);
        for (int j = lane; j < Q; j += 64) sb[j] = sa[j] * (sb[j] - dot);          // dS
        if (lane == 0) { st[3 * i] = mx; st[3 * i + 1] = inv; st[3 * i + 2] = dot; }
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < hd; c += 64) {
            const float* kc = p.k + rbase * p.ld + m * hd + c;
            float acc = 0.f;
            for (int j = 0; j < Q; ++j) acc += sb[j] * kc[(int64_t)j * p.ld];
            p.dq[(rbase + i) * p.ld_d + m * hd + c] = acc * p.scale;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // column phase: d(k), d(v) of key j from the recomputed column of probabilities
    for (int j = wave; j < Q; j += MHA_GEN_NT / 64) {
        const float* kj = p.k + (rbase + j) * p.ld + m * hd;
        const float* vj = p.v + (rbase + j) * p.ld + m * hd;
        for (int i = lane; i < Q; i += 64) {
            const float s = gen_dot(p.q + (rbase + i) * p.ld + m * hd, kj, hd) * p.scale;
            const float pj = __expf(s - st[3 * i]) * st[3 * i + 1];
            float keep = 1.f;
            if (p.thresh) keep = drop_keep(seed, ((uint32_t)blockIdx.x * Q + i) * Q + j, p.thresh) ? p.dscale : 0.f;
            const float dp = gen_dot(p.dout + (rbase + i) * p.ld_out + m * hd, vj, hd) * keep;
            sa[i] = pj * (dp - st[3 * i + 2]) * p.scale;                              // dS (with the 1 / sqrt(hd) of the scaled q)
            sb[i] = pj * keep;                                                        // dropped probability
        }
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < hd; c += 64) {
            const float* qc = p.q + rbase * p.ld + m * hd + c;
            const float* dc = p.dout + rbase * p.ld_out + m * hd + c;
            float ak = 0.f, av = 0.f;
            for (int i = 0; i < Q; ++i) { ak += sa[i] * qc[(int64_t)i * p.ld]; av += sb[i] * dc[(int64_t)i * p.ld_out]; }
            p.dk[(rbase + j) * p.ld_d + m * hd + c] = ak;
            p.dv[(rbase + j) * p.ld_d + m * hd + c] = av;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr size_t MHA_LDS_MAX = 160 * 1024;
// dynamic LDS beyond the 64 KiB default needs the function attribute (set per kernel whenever a launch asks for more)
static void mha_lds_attr(const void* fn, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MHA_LDS_MAX);
}

static inline bool mha_small(int Q, int hd) { return Q <= 128 && (hd == 16 || hd == 32 || hd == 64); }

static int mha_common(MhaP& p, int N, int Q, int M, int hd, float drop_p, uint32_t seed, const uint32_t* seed_dev) {
    POET_CHECK(N > 0 && M > 0 && Q > 0 && hd > 0, POET_ERR_ARG, "mha: N=%d M=%d Q=%d hd=%d", N, M, Q, hd);
    POET_CHECK(Q <= MHA_GEN_QMAX, POET_ERR_UNSUPPORTED, "mha: Q=%d must be in 1..%d", Q, MHA_GEN_QMAX);
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
    dim3 grid(N * M);
    hipStream_t st = (hipStream_t)stream;
    if (!mha_small(Q, hd) || lds > MHA_LDS_MAX) {                       // generic form (any Q <= 2048, any head dim)
        const size_t lg = sizeof(float) * (MHA_GEN_NT / 64) * Q;
        hipLaunchKernelGGL(mha_gen_fwd_kernel, grid, dim3(MHA_GEN_NT), lg, st, p, hd);
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
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
    dim3 grid(N * M);
    hipStream_t st = (hipStream_t)stream;
    if (!mha_small(Q, hd) || lds > MHA_LDS_MAX) {                       // generic form (e.g. Q > 128, head dim 48, Q > 114 at head dim 64)
        const size_t lg = sizeof(float) * (3 * Q + (MHA_GEN_NT / 64) * 2 * Q);
        mha_lds_attr(reinterpret_cast<const void*>(mha_gen_bwd_kernel), lg);
        hipLaunchKernelGGL(mha_gen_bwd_kernel, grid, dim3(MHA_GEN_NT), lg, st, p, hd);
        POET_LAUNCH_CHECK();
        return POET_OK;
    }
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
