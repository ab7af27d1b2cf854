// What the matrix pipe sustains chip-wide on v_mfma_f32_32x32x16_f16 (2 waves per SIMD, four independent accumulators,
// operands with fp16 bit patterns of moderate magnitude), alone and with the plane GEMM's LDS-read mix next to it
// (8 ds_read_b128 per 12 MFMAs, conflict-free addresses) -- the ceiling gemm_planes_kernel's compute side is measured against.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NLDS>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, unsigned seed) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x30003000u + ((i * 2654435761u + seed) & 0x0fff0fffu);
    __syncthreads();
    u32x4 a[2], b[2];
    for (int q = 0; q < 2; ++q)
        for (int j = 0; j < 4; ++j) {
            a[q][j] = 0x30003000u + (((lane + 1) * 2654435761u * (j + 1 + 4 * q) + seed) & 0x0fff0fffu);
            b[q][j] = 0x30003000u + (((lane + 7) * 40503u * (j + 3 + 4 * q) + seed) & 0x8fff8fffu);
        }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const unsigned char* p = lds + (lane & 31) * 64 + ((lane >> 5) << 4) + (threadIdx.x >> 6) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t & 1]), __builtin_bit_cast(f16x8, b[t >> 1]), acc[t], 0, 0, 0);
        }
        if (NLDS) {
#pragma unroll
            for (int q = 0; q < NLDS; ++q) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(p + ((q * 2048 + it * 32) & 4095 & ~31) + (q & 1) * 0);
                if (q < 2) a[q] = v; else if (q < 4) b[q - 2] = v; else { a[q & 1][0] ^= v[0] & 1u; }
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NLDS>
void run(const char* name, float* out, int grid = 512 * 4) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NLDS>, dim3(grid), dim3(256), 0, 0, out, iters, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NLDS>, dim3(grid), dim3(256), 0, 0, out, iters, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 12 * 32768.0;
    printf("%-28s %8.3f ms  %7.0f TFLOP/s  (%.2f of 2.5 PF)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
}

int main() {
    float* out; hipMalloc(&out, sizeof(float) * 512 * 4 * 256);
    run<0>("mfma only", out);
    run<0>("mfma only (again)", out);
    run<4>("mfma + 4 ds_read_b128 / 12", out);
    run<8>("mfma + 8 ds_read_b128 / 12", out);
    // ONE work-group per CU = one wave per SIMD: what a single wave can issue (the peak column counts 256 x 4 waves)
    run<0>("mfma only, 1 wave per SIMD", out, 256);
    run<8>("mfma + 8 ds_read, 1 wave/SIMD", out, 256);
    return 0;
}
