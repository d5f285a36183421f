// VERDICT r3 #8 (a persistent decoder-layer kernel): what does ONE synchronisation point cost on this stack, as a grid barrier
// inside a persistent kernel against a dependent kernel node of a replayed hipGraph?  Both chains run K "ops" of the decoder's
// size: every op reads a 320 x 256 fp32 activation (written by the previous op: a true dependency through memory), does a few
// FMAs per element and writes it back -- the memory round trip of a 320-row Linear without its arithmetic, spread over G
// workgroups.  (a) ONE kernel, G persistent workgroups, a sense-reversing grid barrier between ops (agent-scope release / acquire
// on one counter word, s_sleep in the spin); (b) K kernel launches of the same grid captured once and replayed.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp profiles/probes/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ROWS = 320, COLS = 256, NEL = ROWS * COLS;

__device__ __forceinline__ void op_body(const float* __restrict__ src, float* __restrict__ dst, int g, int G, int tid, int nt, float k) {
    for (int i = g * nt + tid; i < NEL; i += G * nt) dst[i] = fmaf(src[i], 0.999f, k);
}

__global__ void chain_node(const float* src, float* dst, float k) { op_body(src, dst, blockIdx.x, gridDim.x, threadIdx.x, blockDim.x, k); }

// sense-reversing barrier on {count, generation}: the last arriver resets the count and bumps the generation
__device__ __forceinline__ void grid_barrier(unsigned* bar, int G, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);             // (agent scope on AMDGPU: this workgroup's stores are visible device-wide)
        const unsigned arrived = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == (unsigned)G) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + 1, gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    ++gen;
    __syncthreads();
}

__global__ void chain_persistent(float* a, float* b, unsigned* bar, int K) {
    unsigned gen = 0;
    float* src = a;
    float* dst = b;
    for (int k = 0; k < K; ++k) {
        // (sc1 / device-coherent accesses would be needed for CORRECT hand-over of the data through the non-coherent L1s; the probe
        // only times the synchronisation, and reads with the L1 bypassed so that the traffic is the real one)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NEL; i += gridDim.x * blockDim.x)
            dst[i] = fmaf(__builtin_nontemporal_load(src + i), 0.999f, (float)k);
        grid_barrier(bar, gridDim.x, gen);
        float* t = src; src = dst; dst = t;
    }
}

int main() {
    float *a, *b;
    unsigned* bar;
    CK(hipMalloc(&a, NEL * 4)); CK(hipMalloc(&b, NEL * 4)); CK(hipMalloc(&bar, 8));
    CK(hipMemset(a, 0, NEL * 4)); CK(hipMemset(b, 0, NEL * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 200, NT = 256;
    printf("chain of %d dependent ops on a %d x %d fp32 activation; us per op\n", K, ROWS, COLS);
    printf("%8s %18s %18s\n", "grid", "graph nodes", "grid barriers");
    for (int G : {16, 32, 64, 128, 256}) {
        // (b) hipGraph of K kernel nodes
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(chain_node, dim3(G), dim3(NT), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, (float)k);
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms_g; CK(hipEventElapsedTime(&ms_g, e0, e1));
        // (a) persistent kernel with grid barriers (G <= CUs: every workgroup is resident, so the spin cannot deadlock)
        CK(hipMemsetAsync(bar, 0, 8, st));
        for (int w = 0; w < 3; ++w) { CK(hipMemsetAsync(bar, 0, 8, st)); hipLaunchKernelGGL(chain_persistent, dim3(G), dim3(NT), 0, st, a, b, bar, K); }
        CK(hipStreamSynchronize(st));
        float ms_p = 0.f;
        for (int r = 0; r < 10; ++r) {
            CK(hipMemsetAsync(bar, 0, 8, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(chain_persistent, dim3(G), dim3(NT), 0, st, a, b, bar, K);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms_p += t;
        }
        printf("%8d %18.2f %18.2f\n", G, ms_g * 1e3 / (10 * K), ms_p * 1e3 / (10 * K));
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
