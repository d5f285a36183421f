// Multi-scale deformable attention core for gfx950 (the op behind the reference's external
// `deformable_attention.MSDeformAttn`, models/deformable_transformer.py:24,201,283; semantics:
// SURVEY.md Appendix A / oracle/msda_explicit.py).
//
// Work decomposition: one thread = 8 channels of one (image n, query q, head m).  Consecutive
// threads walk the channel groups of one query row, so the (N,Lq,M*D) output / grad_out rows are
// read and written fully coalesced (16 B bf16 or 32 B f32 per lane), and the D/8 lanes of a head
// sit next to each other for the shuffle reductions of the backward pass.  Every bilinear corner
// is ONE 16-byte (bf16) load; with head-major value maps (N,M,S,D) the two x-neighbours of a
// sample are adjacent in memory and neighbouring queries share cache lines, so the 209 MB/img/layer
// of gather traffic (YCB-V) is served by L1/L2 while HBM only sees the algorithmic bytes.
//
// FUSED variants take the raw sampling_offsets / attention_weights Linear outputs and fold in the
// softmax over L*P, loc = ref + off/(W,H) and (backward) the softmax Jacobian, so neither the fp32
// locations nor the weights are ever materialised in HBM.
#include "common.cuh"

namespace poet {

constexpr int MAXL = 4;

struct MsdaP {
    const void* value;
    int64_t vs_n, vs_s, vs_m;
    const void* q1;      // fused: offattn; plain: sampling_loc
    const void* q2;      // plain: attn_weight
    int64_t ldq;
    int logit_col;
    const float* ref;
    int64_t ref_bs;
    void* out;
    const void* grad_out;
    float* grad_value;
    void* g1;            // fused: grad_offattn; plain: grad_loc
    void* g2;            // plain: grad_attn
    int N, S, M, D, Lq;
    int H[MAXL], W[MAXL], start[MAXL];
    int64_t total;       // N*Lq*M*(D/8) threads
    int groups, tpg;     // channel groups per row, per head
};

template <typename TQ, int P>
__device__ __forceinline__ void load_p(const TQ* p, float* o, int n) {
    if constexpr (P == 4) {
        if (n == 8) vec<TQ, 8>::ld(p, o);
        else vec<TQ, 4>::ld(p, o);
    } else {
        for (int i = 0; i < n; ++i) o[i] = io<TQ>::ld(p + i);
    }
}
template <typename TQ, int P>
__device__ __forceinline__ void store_p(TQ* p, const float* o, int n) {
    if constexpr (P == 4) {
        if (n == 8) vec<TQ, 8>::st(p, o);
        else vec<TQ, 4>::st(p, o);
    } else {
        for (int i = 0; i < n; ++i) io<TQ>::st(p + i, o[i]);
    }
}

struct Corner {
    int64_t o00, o01, o10, o11;   // element offsets of the 4 corner pixels (clamped in range)
    float w00, w01, w10, w11;     // bilinear weights, zero where the corner is outside the map
    float fx, fy;
    float m00, m01, m10, m11;     // validity (0/1)
};

__device__ __forceinline__ Corner make_corner(float px, float py, int H, int W, int start, int64_t vs_s) {
    Corner c;
    px = fminf(fmaxf(px, -2.f), (float)W + 1.f);
    py = fminf(fmaxf(py, -2.f), (float)H + 1.f);
    const float x0f = floorf(px), y0f = floorf(py);
    c.fx = px - x0f;
    c.fy = py - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float vx0 = (x0 >= 0 && x0 < W) ? 1.f : 0.f, vx1 = (x1 >= 0 && x1 < W) ? 1.f : 0.f;
    const float vy0 = (y0 >= 0 && y0 < H) ? 1.f : 0.f, vy1 = (y1 >= 0 && y1 < H) ? 1.f : 0.f;
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y1, 0), H - 1);
    c.m00 = vy0 * vx0; c.m01 = vy0 * vx1; c.m10 = vy1 * vx0; c.m11 = vy1 * vx1;
    c.w00 = (1.f - c.fy) * (1.f - c.fx) * c.m00;
    c.w01 = (1.f - c.fy) * c.fx * c.m01;
    c.w10 = c.fy * (1.f - c.fx) * c.m10;
    c.w11 = c.fy * c.fx * c.m11;
    c.o00 = (int64_t)(start + yc0 * W + xc0) * vs_s;
    c.o01 = (int64_t)(start + yc0 * W + xc1) * vs_s;
    c.o10 = (int64_t)(start + yc1 * W + xc0) * vs_s;
    c.o11 = (int64_t)(start + yc1 * W + xc1) * vs_s;
    return c;
}

template <typename TQ, int L, int P, bool FUSED>
__device__ __forceinline__ void load_weights(const MsdaP& p, int64_t row, int m, float* a) {
    constexpr int LP = L * P;
    if constexpr (FUSED) {
        const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq + p.logit_col + m * LP;
#pragma unroll
        for (int l = 0; l < L; ++l) load_p<TQ, P>(lp + l * P, a + l * P, P);
        float mx = a[0];
#pragma unroll
        for (int i = 1; i < LP; ++i) mx = fmaxf(mx, a[i]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LP; ++i) { a[i] = __expf(a[i] - mx); s += a[i]; }
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < LP; ++i) a[i] *= inv;
    } else {
        const TQ* wp = reinterpret_cast<const TQ*>(p.q2) + (row * p.M + m) * LP;
#pragma unroll
        for (int l = 0; l < L; ++l) load_p<TQ, P>(wp + l * P, a + l * P, P);
    }
}

// pixel coordinates (px,py) of the P points of level l for (row, m)
template <typename TQ, int L, int P, bool FUSED>
__device__ __forceinline__ void load_points(const MsdaP& p, int64_t row, int n, int q, int m, int l, float* xy) {
    if constexpr (FUSED) {
        const TQ* op = reinterpret_cast<const TQ*>(p.q1) + row * p.ldq + (m * L + l) * P * 2;
        load_p<TQ, P>(op, xy, 2 * P);
        const float* rp = p.ref + (int64_t)n * p.ref_bs + ((int64_t)q * L + l) * 2;
        const float rx = rp[0] * (float)p.W[l] - 0.5f, ry = rp[1] * (float)p.H[l] - 0.5f;
#pragma unroll
        for (int i = 0; i < P; ++i) { xy[2 * i] += rx; xy[2 * i + 1] += ry; }
    } else {
        const TQ* lp = reinterpret_cast<const TQ*>(p.q1) + ((row * p.M + m) * L + l) * P * 2;
        load_p<TQ, P>(lp, xy, 2 * P);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            xy[2 * i] = xy[2 * i] * (float)p.W[l] - 0.5f;
            xy[2 * i + 1] = xy[2 * i + 1] * (float)p.H[l] - 0.5f;
        }
    }
}

template <typename TV, typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const MsdaP p) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.total) return;
    const int64_t row = t / p.groups;
    const int c8 = (int)(t - row * p.groups);
    const int m = c8 / p.tpg, dsub = c8 - m * p.tpg;
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);

    float a[L * P];
    load_weights<TQ, L, P, FUSED>(p, row, m, a);
    const TV* vbase = reinterpret_cast<const TV*>(p.value) + (int64_t)n * p.vs_n + (int64_t)m * p.vs_m + dsub * 8;

    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float xy[2 * P];
        load_points<TQ, L, P, FUSED>(p, row, n, q, m, l, xy);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const Corner c = make_corner(xy[2 * i], xy[2 * i + 1], p.H[l], p.W[l], p.start[l], p.vs_s);
            const float aw = a[l * P + i];
            float v00[8], v01[8], v10[8], v11[8];
            vec<TV, 8>::ld(vbase + c.o00, v00);
            vec<TV, 8>::ld(vbase + c.o01, v01);
            vec<TV, 8>::ld(vbase + c.o10, v10);
            vec<TV, 8>::ld(vbase + c.o11, v11);
            const float w00 = aw * c.w00, w01 = aw * c.w01, w10 = aw * c.w10, w11 = aw * c.w11;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch)
                acc[ch] += w00 * v00[ch] + w01 * v01[ch] + w10 * v10[ch] + w11 * v11[ch];
        }
    }
    TQ* op = reinterpret_cast<TQ*>(p.out) + row * ((int64_t)p.M * p.D) + m * p.D + dsub * 8;
    vec<TQ, 8>::st(op, acc);
}

__device__ __forceinline__ float group_sum(float v, int tpg) {
    for (int o = 1; o < tpg; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename TV, typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const MsdaP p) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.total) return;
    const int64_t row = t / p.groups;
    const int c8 = (int)(t - row * p.groups);
    const int m = c8 / p.tpg, dsub = c8 - m * p.tpg;
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);

    float a[L * P];
    load_weights<TQ, L, P, FUSED>(p, row, m, a);
    const int64_t voff = (int64_t)n * p.vs_n + (int64_t)m * p.vs_m + dsub * 8;
    const TV* vbase = reinterpret_cast<const TV*>(p.value) + voff;

    float g[8];
    vec<TQ, 8>::ld(reinterpret_cast<const TQ*>(p.grad_out) + row * ((int64_t)p.M * p.D) + m * p.D + dsub * 8, g);

    float da[L * P];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float xy[2 * P], dxy[2 * P];
        load_points<TQ, L, P, FUSED>(p, row, n, q, m, l, xy);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const Corner c = make_corner(xy[2 * i], xy[2 * i + 1], p.H[l], p.W[l], p.start[l], p.vs_s);
            const float aw = a[l * P + i];
            float v00[8], v01[8], v10[8], v11[8];
            vec<TV, 8>::ld(vbase + c.o00, v00);
            vec<TV, 8>::ld(vbase + c.o01, v01);
            vec<TV, 8>::ld(vbase + c.o10, v10);
            vec<TV, 8>::ld(vbase + c.o11, v11);
            float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                d00 += g[ch] * v00[ch]; d01 += g[ch] * v01[ch];
                d10 += g[ch] * v10[ch]; d11 += g[ch] * v11[ch];
            }
            d00 *= c.m00; d01 *= c.m01; d10 *= c.m10; d11 *= c.m11;
            float s_da = c.w00 * d00 + c.w01 * d01 + c.w10 * d10 + c.w11 * d11;   // weights already carry validity;
            // (d.. were masked too, which is harmless: m*m = m)
            float s_dx = aw * ((1.f - c.fy) * (d01 - d00) + c.fy * (d11 - d10));
            float s_dy = aw * ((1.f - c.fx) * (d10 - d00) + c.fx * (d11 - d01));
            da[l * P + i] = group_sum(s_da, p.tpg);
            dxy[2 * i] = group_sum(s_dx, p.tpg);
            dxy[2 * i + 1] = group_sum(s_dy, p.tpg);
        }
        if (dsub == 0) {
            if constexpr (FUSED) {
                TQ* gp = reinterpret_cast<TQ*>(p.g1) + row * p.ldq + (m * L + l) * P * 2;
                store_p<TQ, P>(gp, dxy, 2 * P);          // d/d(offset) = (dpx, dpy): the W,H factors cancel
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) { dxy[2 * i] *= (float)p.W[l]; dxy[2 * i + 1] *= (float)p.H[l]; }
                TQ* gp = reinterpret_cast<TQ*>(p.g1) + ((row * p.M + m) * L + l) * P * 2;
                store_p<TQ, P>(gp, dxy, 2 * P);
            }
        }
    }
    if (dsub == 0) {
        constexpr int LP = L * P;
        if constexpr (FUSED) {
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < LP; ++i) dot += a[i] * da[i];
#pragma unroll
            for (int i = 0; i < LP; ++i) da[i] = a[i] * (da[i] - dot);
            TQ* gp = reinterpret_cast<TQ*>(p.g1) + row * p.ldq + p.logit_col + m * LP;
#pragma unroll
            for (int l = 0; l < L; ++l) store_p<TQ, P>(gp + l * P, da + l * P, P);
        } else {
            TQ* gp = reinterpret_cast<TQ*>(p.g2) + (row * p.M + m) * LP;
#pragma unroll
            for (int l = 0; l < L; ++l) store_p<TQ, P>(gp + l * P, da + l * P, P);
        }
    }
}

// d(value): pure scatter, no value loads.  One lane per channel: the D lanes of a (query, head) add D consecutive
// floats, so every atomic wave-instruction covers whole contiguous 4*D-byte segments (coalesced into a few L2 atomic
// requests) instead of 64 scattered dwords.  The softmax / corner arithmetic is recomputed per lane (it is tiny).
template <typename TQ, int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_bwd_dv_kernel(const MsdaP p) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.total * 8) return;
    const int64_t rm = t / p.D;
    const int c = (int)(t - rm * p.D);
    const int64_t row = rm / p.M;
    const int m = (int)(rm - row * p.M);
    const int n = (int)(row / p.Lq), q = (int)(row - (int64_t)n * p.Lq);
    float a[L * P];
    load_weights<TQ, L, P, FUSED>(p, row, m, a);
    const float g = io<TQ>::ld(reinterpret_cast<const TQ*>(p.grad_out) + row * ((int64_t)p.M * p.D) + m * p.D + c);
    float* gv = p.grad_value + (int64_t)n * p.vs_n + (int64_t)m * p.vs_m + c;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float xy[2 * P];
        load_points<TQ, L, P, FUSED>(p, row, n, q, m, l, xy);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const Corner cn = make_corner(xy[2 * i], xy[2 * i + 1], p.H[l], p.W[l], p.start[l], p.vs_s);
            const float ag = a[l * P + i] * g;
            if (cn.w00 != 0.f) atomicAdd(gv + cn.o00, ag * cn.w00);
            if (cn.w01 != 0.f) atomicAdd(gv + cn.o01, ag * cn.w01);
            if (cn.w10 != 0.f) atomicAdd(gv + cn.o10, ag * cn.w10);
            if (cn.w11 != 0.f) atomicAdd(gv + cn.o11, ag * cn.w11);
        }
    }
}

template <typename TV, typename TQ, int L, bool FUSED, bool BWD>
static void launch_p(const MsdaP& p, int P, hipStream_t st) {
    dim3 grid(cdiv(p.total, 256)), block(256);
    dim3 gridv(cdiv(p.total * 8, 256));
    if (BWD) {
        if (P == 4) hipLaunchKernelGGL((msda_bwd_dv_kernel<TQ, L, 4, FUSED>), gridv, block, 0, st, p);
        else if (P == 2) hipLaunchKernelGGL((msda_bwd_dv_kernel<TQ, L, 2, FUSED>), gridv, block, 0, st, p);
        else hipLaunchKernelGGL((msda_bwd_dv_kernel<TQ, L, 1, FUSED>), gridv, block, 0, st, p);
    }
    if (P == 4) {
        if (BWD) hipLaunchKernelGGL((msda_bwd_kernel<TV, TQ, L, 4, FUSED>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((msda_fwd_kernel<TV, TQ, L, 4, FUSED>), grid, block, 0, st, p);
    } else if (P == 2) {
        if (BWD) hipLaunchKernelGGL((msda_bwd_kernel<TV, TQ, L, 2, FUSED>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((msda_fwd_kernel<TV, TQ, L, 2, FUSED>), grid, block, 0, st, p);
    } else {
        if (BWD) hipLaunchKernelGGL((msda_bwd_kernel<TV, TQ, L, 1, FUSED>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((msda_fwd_kernel<TV, TQ, L, 1, FUSED>), grid, block, 0, st, p);
    }
}

template <typename TV, typename TQ, bool FUSED, bool BWD>
static void launch_l(const MsdaP& p, int L, int P, hipStream_t st) {
    switch (L) {
        case 1: launch_p<TV, TQ, 1, FUSED, BWD>(p, P, st); break;
        case 2: launch_p<TV, TQ, 2, FUSED, BWD>(p, P, st); break;
        case 3: launch_p<TV, TQ, 3, FUSED, BWD>(p, P, st); break;
        default: launch_p<TV, TQ, 4, FUSED, BWD>(p, P, st); break;
    }
}

template <bool FUSED, bool BWD>
static int dispatch(const MsdaP& p, int L, int P, int v_dtype, int q_dtype, hipStream_t st) {
    if (v_dtype == POET_BF16 && q_dtype == POET_BF16) launch_l<bf16_t, bf16_t, FUSED, BWD>(p, L, P, st);
    else if (v_dtype == POET_F32 && q_dtype == POET_F32) launch_l<float, float, FUSED, BWD>(p, L, P, st);
    else if (v_dtype == POET_BF16 && q_dtype == POET_F32) launch_l<bf16_t, float, FUSED, BWD>(p, L, P, st);
    else { set_error("msda: unsupported dtype pair v=%d q=%d", v_dtype, q_dtype); return POET_ERR_UNSUPPORTED; }
    return POET_OK;
}

static int fill_common(MsdaP& p, const int64_t* shapes, const int64_t* starts, int N, int S, int M, int D, int L, int P, int Lq) {
    POET_CHECK(N > 0 && S > 0 && M > 0 && Lq > 0, POET_ERR_ARG, "msda: bad dims");
    POET_CHECK(D % 8 == 0 && D >= 8, POET_ERR_UNSUPPORTED, "msda: head dim %d must be a multiple of 8", D);
    POET_CHECK((D / 8) <= 64 && (((D / 8) & ((D / 8) - 1)) == 0), POET_ERR_UNSUPPORTED, "msda: head dim %d/8 must be a power of two", D);
    POET_CHECK(L >= 1 && L <= MAXL, POET_ERR_UNSUPPORTED, "msda: n_levels %d not in 1..%d", L, MAXL);
    POET_CHECK(P == 4 || P == 2 || P == 1, POET_ERR_UNSUPPORTED, "msda: n_points %d not in {1,2,4}", P);
    POET_CHECK(shapes && starts, POET_ERR_ARG, "msda: null shapes");
    int64_t tot = 0;
    for (int l = 0; l < L; ++l) {
        p.H[l] = (int)shapes[2 * l];
        p.W[l] = (int)shapes[2 * l + 1];
        p.start[l] = (int)starts[l];
        POET_CHECK(p.H[l] > 0 && p.W[l] > 0, POET_ERR_ARG, "msda: bad level shape");
        tot += (int64_t)p.H[l] * p.W[l];
    }
    POET_CHECK(tot == S, POET_ERR_ARG, "msda: sum(H*W)=%lld != S=%d", (long long)tot, S);
    p.N = N; p.S = S; p.M = M; p.D = D; p.Lq = Lq;
    p.tpg = D / 8;
    p.groups = M * p.tpg;
    p.total = (int64_t)N * Lq * p.groups;
    return POET_OK;
}

}  // namespace poet

using namespace poet;

extern "C" int poet_msda_fwd(const void* value, const int64_t* shapes, const int64_t* starts, const void* loc,
                             const void* attn, void* out, int N, int S, int M, int D, int L, int P, int Lq,
                             int dtype, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    POET_CHECK(value && loc && attn && out, POET_ERR_ARG, "msda_fwd: null pointer");
    p.value = value; p.vs_n = (int64_t)S * M * D; p.vs_s = (int64_t)M * D; p.vs_m = D;
    p.q1 = loc; p.q2 = attn; p.out = out;
    rc = dispatch<false, false>(p, L, P, dtype, dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_msda_bwd(const void* value, const int64_t* shapes, const int64_t* starts, const void* loc,
                             const void* attn, const void* grad_out, float* grad_value, void* grad_loc, void* grad_attn,
                             int N, int S, int M, int D, int L, int P, int Lq, int dtype, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    POET_CHECK(value && loc && attn && grad_out && grad_value && grad_loc && grad_attn, POET_ERR_ARG, "msda_bwd: null pointer");
    p.value = value; p.vs_n = (int64_t)S * M * D; p.vs_s = (int64_t)M * D; p.vs_m = D;
    p.q1 = loc; p.q2 = attn; p.grad_out = grad_out; p.grad_value = grad_value; p.g1 = grad_loc; p.g2 = grad_attn;
    hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)N * S * M * D, (hipStream_t)stream);
    POET_CHECK(e == hipSuccess, POET_ERR_LAUNCH, "msda_bwd: memset failed: %s", hipGetErrorString(e));
    rc = dispatch<false, true>(p, L, P, dtype, dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}

static int fused_args(MsdaP& p, const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m, const void* offattn,
                      int64_t ldq, int logit_col, const float* ref, int64_t ref_bs, int M, int L, int P) {
    POET_CHECK(value && offattn && ref, POET_ERR_ARG, "msda_fused: null pointer");
    POET_CHECK(ldq % 8 == 0 && logit_col % 8 == 0 && logit_col >= M * L * P * 2, POET_ERR_ARG,
               "msda_fused: ldq/logit_col must be multiples of 8 and logits must follow the offsets");
    POET_CHECK(vs_s % 8 == 0 && vs_m % 8 == 0 && vs_n % 8 == 0, POET_ERR_ARG, "msda_fused: value strides must be multiples of 8");
    p.value = value; p.vs_n = vs_n; p.vs_s = vs_s; p.vs_m = vs_m;
    p.q1 = offattn; p.ldq = ldq; p.logit_col = logit_col; p.ref = ref; p.ref_bs = ref_bs;
    return POET_OK;
}

extern "C" int poet_msda_fused_fwd(const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m, const int64_t* shapes,
                                   const int64_t* starts, const void* offattn, int64_t ldq, int logit_col,
                                   const float* ref, int64_t ref_bs, void* out, int N, int S, int M, int D, int L,
                                   int P, int Lq, int v_dtype, int q_dtype, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    rc = fused_args(p, value, vs_n, vs_s, vs_m, offattn, ldq, logit_col, ref, ref_bs, M, L, P);
    if (rc) return rc;
    POET_CHECK(out, POET_ERR_ARG, "msda_fused_fwd: null out");
    p.out = out;
    rc = dispatch<true, false>(p, L, P, v_dtype, q_dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}

extern "C" int poet_msda_fused_bwd(const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m, const int64_t* shapes,
                                   const int64_t* starts, const void* offattn, int64_t ldq, int logit_col,
                                   const float* ref, int64_t ref_bs, const void* grad_out, float* grad_value,
                                   void* grad_offattn, int N, int S, int M, int D, int L, int P, int Lq,
                                   int v_dtype, int q_dtype, void* stream) {
    MsdaP p{};
    int rc = fill_common(p, shapes, starts, N, S, M, D, L, P, Lq);
    if (rc) return rc;
    rc = fused_args(p, value, vs_n, vs_s, vs_m, offattn, ldq, logit_col, ref, ref_bs, M, L, P);
    if (rc) return rc;
    POET_CHECK(grad_out && grad_value && grad_offattn, POET_ERR_ARG, "msda_fused_bwd: null pointer");
    p.grad_out = grad_out; p.grad_value = grad_value; p.g1 = grad_offattn;
    rc = dispatch<true, true>(p, L, P, v_dtype, q_dtype, (hipStream_t)stream);
    if (rc) return rc;
    POET_LAUNCH_CHECK();
    return POET_OK;
}
