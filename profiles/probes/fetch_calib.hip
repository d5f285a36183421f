// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the MSDA value-gradient scatter (VERDICT r3 #7a).
// Every kernel reads a KNOWN number of distinct bytes exactly once from a 1.5 GiB buffer (6x the 256 MiB Infinity Cache), so
// FETCH_SIZE x 1 KiB / bytes is the counter's scale for that pattern:
//   stream16   16 B per lane, fully coalesced (the guide's reference pattern: FETCH_SIZE reports 1/2)
//   seg32_s512 a 16-lane DPP row reads 32 contiguous bytes (2 B per lane) of a 512-byte row: grad_out of one head
//   seg64_s1536 a row reads 64 contiguous bytes (4 B per lane) of a 1536-byte row: the offsets of one head in the offsets | logits rows
//   seg32_s1536 32 contiguous bytes of a 1536-byte row: the logits of one head
//   lane16_s1536 16 B per lane, consecutive lanes 1536 bytes apart (lane-per-query operand loads of the matrix-core scatter)
// Build: hipcc --offload-arch=gfx950 -O2 profiles/probes/fetch_calib.hip -o profiles/probes/_bin/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- profiles/probes/_bin/fetch_calib   (collect_calib.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void stream16(const uint4* p, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// rows of `stride` bytes; each 16-lane group reads `seg` contiguous bytes at byte offset `off` of its row, seg / 16 bytes per lane
template <typename T>
__global__ void segread(const char* p, size_t rows, int stride, int off, uint32_t* sink) {
    uint32_t acc = 0;
    const int c = threadIdx.x & 15;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < rows; r += ((size_t)gridDim.x * blockDim.x) >> 4) {
        const T v = *reinterpret_cast<const T*>(p + r * stride + off + c * sizeof(T));
        acc ^= (uint32_t)v;
    }
    if (acc == 0x1234u) *sink = acc;          // (a value a 16-bit xor CAN reach: otherwise the loop is dead code)
}
__global__ void lane16(const char* p, size_t rows, int stride, int off, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + r * stride + off);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)1536 << 20;
    char* buf; uint32_t* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int G = 4096, B = 256;
    for (int rep = 0; rep < 3; ++rep) {
        stream16<<<G, B>>>((const uint4*)buf, bytes / 16, sink);
        segread<uint16_t><<<G, B>>>(buf, bytes / 512, 512, 64, sink);
        segread<uint32_t><<<G, B>>>(buf, bytes / 1536, 1536, 128, sink);
        segread<uint16_t><<<G, B>>>(buf, bytes / 1536, 1536, 1024 + 96, sink);
        lane16<<<G, B>>>(buf, bytes / 1536, 1536, 256, sink);
        hipDeviceSynchronize();
    }
    printf("bytes read per launch: stream16 %zu seg32_s512 %zu seg64_s1536 %zu seg32_s1536 %zu lane16_s1536 %zu\n", bytes, bytes / 512 * 32, bytes / 1536 * 64,
           bytes / 1536 * 32, bytes / 1536 * 16);
    return 0;
}
