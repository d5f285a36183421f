// Layout check of v_mfma_f32_16x16x16_f16 and ds_read_b64_tr_b16 as msda_bwd_dv_mfma_kernel uses them:
//   A[i][k]: lane l holds A[l & 15][4 (l >> 4) + j], B[k][n]: lane l holds B[4 (l >> 4) + j][l & 15], D[i][n]: lane l holds D[4 (l >> 4) + r][l & 15];
//   the transposing read of a row-major [16 k][16 n] fp16 matrix gives the B operand.
// Build + run:  hipcc --offload-arch=gfx950 -O2 profiles/probes/mfma16_probe.hip -o /tmp/mfma16_probe && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;

__global__ void probe(const float* A, const float* B, float* D1, float* D2) {
    __shared__ __attribute__((aligned(16))) _Float16 sa[16 * 16];   // [px i][k]   (the W tile layout)
    __shared__ __attribute__((aligned(16))) _Float16 sb[16 * 16];   // [k][n]      (the staged grad_out rows)
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) { sa[i] = (_Float16)A[i]; sb[i] = (_Float16)B[i]; }
    __syncthreads();
    const f16x4_t a = *reinterpret_cast<const f16x4_t*>(reinterpret_cast<const char*>(sa) + (lane & 15) * 32 + (lane >> 4) * 8);
    // (1) B gathered element-wise
    f16x4_t b1;
    for (int j = 0; j < 4; ++j) b1[j] = sb[(4 * (lane >> 4) + j) * 16 + (lane & 15)];
    // (2) B through the transposing read
    const int g_rd = ((lane >> 4) * 4 + ((lane & 15) >> 2)) * 32 + (lane & 15 & 3) * 8;
    const f16x4_t b2 = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(reinterpret_cast<char*>(sb) + g_rd)));
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b1, z, 0, 0, 0);
    const f32x4_t d2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b2, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        D1[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = d1[r];
        D2[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = d2[r];
    }
}

int main() {
    float hA[256], hB[256], hD1[256], hD2[256], ref[256];
    for (int i = 0; i < 256; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 16 + n]; ref[i * 16 + n] = s; }
    float *dA, *dB, *dD1, *dD2;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD1, 1024); hipMalloc(&dD2, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD1, dD2);
    hipMemcpy(hD1, dD1, 1024, hipMemcpyDeviceToHost); hipMemcpy(hD2, dD2, 1024, hipMemcpyDeviceToHost);
    float e1 = 0, e2 = 0;
    for (int i = 0; i < 256; ++i) { e1 = fmaxf(e1, fabsf(hD1[i] - ref[i])); e2 = fmaxf(e2, fabsf(hD2[i] - ref[i])); }
    printf("mfma_16x16x16f16 layout: max err element-wise B %g, transposing-read B %g (0 = as assumed)\n", e1, e2);
    return 0;
}
