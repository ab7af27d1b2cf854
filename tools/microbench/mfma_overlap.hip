// Micro-benchmark: does a dependent chain of v_mfma_f32_32x32x16_bf16 overlap with VALU / LDS work of the same wave
// and of co-resident waves?  Prints cycles per loop iteration (s_memtime) for several mixes.
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] mfma_overlap.hip -o mfma_overlap && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NMFMA, int NVALU, int NLDS, int SHAPE, bool INTERLEAVE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = lane * 0.5f + j;
    float l = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVE) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) {
                if (SHAPE == 32) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                else acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NVALU / (NMFMA ? NMFMA : 1); ++q) v[(m + q) & 7] = v[(m + q) & 7] * 1.0001f + 0.5f;
#pragma unroll
                for (int q = 0; q < NLDS / (NMFMA ? NMFMA : 1); ++q) l += lds[(lane * 2 + m * 67 + q * 131 + it) & 4095];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) {
                if (SHAPE == 32) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                else acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NVALU; ++q) v[q & 7] = v[q & 7] * 1.0001f + 0.5f;
#pragma unroll
            for (int q = 0; q < NLDS; ++q) l += lds[(lane * 2 + q * 131 + it) & 4095];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = l + acc4[0] + acc4[3];
    for (int j = 0; j < 16; ++j) s += acc[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int NMFMA, int NVALU, int NLDS, int SHAPE, bool INTERLEAVE>
void run(const char* name, int waves_per_simd) {
    const int iters = 2000;
    const int threads = 256;                       // 4 waves = 1 per SIMD
    const int blocks = 256 * waves_per_simd;       // waves_per_simd blocks per CU
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    hipLaunchKernelGGL((k<NMFMA, NVALU, NLDS, SHAPE, INTERLEAVE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NMFMA, NVALU, NLDS, SHAPE, INTERLEAVE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= h.size();
    // s_memtime ticks at 100 MHz on some parts: report wall-derived SIMD time too
    printf("%-44s waves/SIMD %d: memtime ticks/iter %.1f | wall %.3f ms -> per-SIMD ns/iter/wave-slot %.1f\n", name, waves_per_simd,
           avg / iters, ms, ms * 1e6 / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<10, 0, 0, 32, false>("10 MFMA32 chain only", w);
        run<0, 60, 0, 32, false>("60 VALU only", w);
        run<0, 0, 20, 32, false>("20 LDS reads only", w);
        run<10, 60, 0, 32, false>("10 MFMA32 then 60 VALU", w);
        run<10, 60, 0, 32, true>("10 MFMA32 interleaved with 60 VALU", w);
        run<10, 60, 20, 32, false>("10 MFMA32 then 60 VALU + 20 LDS", w);
        run<10, 60, 20, 32, true>("10 MFMA32 interleaved 60 VALU + 20 LDS", w);
        run<20, 60, 0, 16, true>("20 MFMA16 interleaved with 60 VALU", w);
    }
    return 0;
}
