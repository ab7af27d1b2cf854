#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void tiny2(const float* __restrict__ a, float* __restrict__ p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = a[i] + 1.f; }
int main() {
    float *p, *q; hipMalloc(&p, 4 << 20); hipMalloc(&q, 4 << 20); hipMemset(p, 0, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 1024, 4096}) {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, 0, p, blocks * 256);
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, 0, p, blocks * 256);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dependent chain, %5d blocks: %.2f us per launch\n", blocks, ms / 200 * 1e3);
    }
    // graph of the same chain
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(1024), dim3(256), 0, st, p, 1024 * 256);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 200 dependent launches (1024 blocks): %.2f us per kernel\n", ms / 200 * 1e3);
    return 0;
}
