// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of score_approx32_kernel: every kernel
// reads a KNOWN number of bytes exactly once from buffers far larger than L2 + Infinity Cache, so
// FETCH_SIZE * 1024 / bytes is the counter's scale for that pattern.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (a) 16 B per lane, contiguous across the wave (the residual stream)
__global__ void stream_x4(const u32x4* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// (b) 4 B per lane, contiguous (codes, inv_norm)
__global__ void stream_x1(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) out[0] = acc;
}
// (c) 64-byte rows picked by a permutation (each row once): lane (r, h) of a wave reads bytes [16h, 16h+16) and
//     [32+16h, 48+16h) of row perm[32 * step + r] -- the score-row gather
__global__ void gather64(const unsigned char* __restrict__ table, const uint32_t* __restrict__ perm, size_t n_rows,
                         uint32_t* out) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (size_t s = wave; s * 32 < n_rows; s += waves) {
        const uint32_t row = perm[s * 32 + r];
        const unsigned char* q = table + (size_t)row * 64 + 16 * h;
        const u32x4 a = *reinterpret_cast<const u32x4*>(q), b = *reinterpret_cast<const u32x4*>(q + 32);
        acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)4 << 30;                 // 4 GiB per pattern
    unsigned char* buf; uint32_t* perm; uint32_t* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    const size_t n_rows = bytes / 64;                     // 64 Mi rows
    hipMalloc(&perm, n_rows * 4);
    {   // a bijective scramble of the row ids: multiply by an odd constant modulo 2^26
        uint32_t* h = (uint32_t*)malloc(n_rows * 4);
        for (size_t i = 0; i < n_rows; ++i) h[i] = (uint32_t)((i * 2654435761ull) & (n_rows - 1));
        hipMemcpy(perm, h, n_rows * 4, hipMemcpyHostToDevice);
        free(h);
    }
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_x4, dim3(4096), dim3(256), 0, 0, (const u32x4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(stream_x1, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
    hipLaunchKernelGGL(gather64, dim3(4096), dim3(256), 0, 0, buf, perm, n_rows, out);
    hipDeviceSynchronize();
    printf("bytes per kernel: stream_x4 %zu, stream_x1 %zu, gather64 %zu (+ %zu of row ids)\n", bytes, bytes, bytes, n_rows * 4);
    return 0;
}
