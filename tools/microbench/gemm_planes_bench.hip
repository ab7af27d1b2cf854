// Times gemm_planes_kernel (csrc/encoder_kernels.hpp) on the Linear shapes of a 32 x 32-token query batch and its two
// ablations (no DMA / no MFMA), to see which side of the kernel sets its time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../colbert.jl_amd/csrc gemm_planes_bench.hip -o gemm_planes_bench
#include "approx_kernels.hpp"
#include "encoder_kernels.hpp"
#include <cstdio>
#include <vector>
using namespace clb;

// touches every 128-byte line of [p, p + bytes): pulls it into the Infinity Cache (and the reading XCD's L2)
__global__ void prefetch_kernel(const unsigned char* __restrict__ p, size_t bytes, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t off = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 128; off < bytes; off += (size_t)gridDim.x * blockDim.x * 128)
        acc ^= *reinterpret_cast<const uint32_t*>(p + off);
    if (acc == 0x12345678u) *sink = acc;
}

// COLD: the weight operand rotates over `nrot` different buffers (12 layers of weights do not fit the 256-MB Infinity
// Cache: every Linear of an encode reads its weights from HBM); PF: a prefetch kernel on a second stream touches the NEXT
// launch's weights while this one runs
template <int WM, int WN, int ST, int NS = 3, bool F16 = false>
float run_cold(const GemmPArgs& g, int bm, int bn, int ks, int reps, std::vector<uint16_t*>& Bs, bool pf, hipStream_t s2, uint32_t* sink) {
    GemmPArgs a = g; a.ksplit = ks;
    const dim3 grid((unsigned)gemm_planes_grid(g.M, g.N, bm, bn, ks));
    const size_t lds = (size_t)ST * NS * (bm + bn) * 64;
    auto kern = gemm_planes_kernel<2, 2, WM, WN, NS, ST, 0, F16>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1, ev; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const size_t bbytes = (size_t)g.b_plane * 6;
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) {
        a.B = Bs[i % Bs.size()];
        if (pf) {
            hipEventRecord(ev, 0);
            hipStreamWaitEvent(s2, ev, 0);
            hipLaunchKernelGGL(prefetch_kernel, dim3(256), dim3(256), 0, s2, reinterpret_cast<const unsigned char*>(Bs[(i + 1) % Bs.size()]), bbytes, sink);
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

template <int WM, int WN, int ST, int ABL, int WGM = 2, int WGN = 2, int NS = 3, bool F16 = false>
float run(const GemmPArgs& g, int bm, int bn, int ks, int reps) {
    GemmPArgs a = g; a.ksplit = ks;
    const dim3 grid((unsigned)gemm_planes_grid(g.M, g.N, bm, bn, ks));
    const size_t lds = (size_t)ST * NS * (bm + bn) * 64;
    auto kern = gemm_planes_kernel<WGM, WGN, WM, WN, NS, ST, ABL, F16>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int sustained = argc > 1 ? atoi(argv[1]) : 0;
    struct Shape { const char* name; int M, N, K; } shapes[] = {{"qkv", 1024, 2304, 768}, {"attn_out", 1024, 768, 768},
                                                               {"ffn_in", 1024, 3072, 768}, {"ffn_out", 1024, 768, 3072}};
    for (auto& sh : shapes) {
        const int64_t ap = (int64_t)sh.M * sh.K, bp = (int64_t)sh.N * sh.K;
        uint16_t *A, *B; float* C;
        hipMalloc(&A, ap * 6); hipMalloc(&B, bp * 6); hipMalloc(&C, sizeof(float) * 8 * sh.M * sh.N);
        if (getenv("GPB_RANDOM")) {      // fp16 bit patterns of moderate magnitude instead of zeros (switching activity = power = clock)
            std::vector<uint16_t> ha(ap * 3), hb(bp * 3);
            uint32_t x = 12345u;
            for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000u + ((x >> 16) & 0x0fffu)) | (uint16_t)((x >> 3) & 0x8000u); }
            for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000u + ((x >> 16) & 0x0fffu)) | (uint16_t)((x >> 3) & 0x8000u); }
            hipMemcpy(A, ha.data(), ap * 6, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), bp * 6, hipMemcpyHostToDevice);
        } else { hipMemset(A, 0, ap * 6); hipMemset(B, 0, bp * 6); }
        GemmPArgs g{A, B, ap, bp, C, nullptr, nullptr, nullptr, 0, sh.M, sh.N, sh.K, sh.N, 0, 1};
        const double flop = 6.0 * 2.0 * sh.M * sh.N * sh.K;
        auto report = [&](const char* cfg, int bm, int bn, int ks, float us, float us_nodma, float us_nomfma) {
            const double bytes = (double)((sh.N + bn - 1) / bn) * ((sh.M + bm - 1) / bm) * (sh.K / 32) * 3.0 * (bm + bn) * 64;
            printf("%-9s %-14s ks=%d  %7.1f us  %6.0f TF  L2 %5.1f TB/s | no-DMA %7.1f us | no-MFMA %7.1f us\n", sh.name, cfg, ks, us,
                   flop / us / 1e6, bytes / us / 1e6, us_nodma, us_nomfma);
        };
#define CFG(NAME, BM, BN, WM, WN, ST, KS)                                                                    \
        report(NAME, BM, BN, KS, run<WM, WN, ST, 0>(g, BM, BN, KS, 20), run<WM, WN, ST, 1>(g, BM, BN, KS, 20), run<WM, WN, ST, 2>(g, BM, BN, KS, 20));
        const int ksn = sh.N <= 768 ? 4 : 1;
        if (sustained) {       // the same launch `sustained` times back to back: what clock does the chip hold under this load?
            for (int rep = 0; rep < 3; ++rep)
                printf("%-9s f16 64x64x2 ks=%d sustained x%d: %7.1f us per launch\n", sh.name, ksn, sustained,
                       run<1, 1, 2, 0, 2, 2, 2, true>(g, 64, 64, ksn, sustained));
            hipFree(A); hipFree(B); hipFree(C);
            continue;
        }
        if (sh.N <= 768) { CFG("64x64x2", 64, 64, 1, 1, 2, 8) CFG("64x64x3", 64, 64, 1, 1, 3, 8) CFG("64x128x3", 64, 128, 1, 2, 3, 8) CFG("64x64x2", 64, 64, 1, 1, 2, 2) }
#define CFGW(NAME, BM, BN, WGM, WGN, WM, WN, ST, KS)                                                         \
        report(NAME, BM, BN, KS, run<WM, WN, ST, 0, WGM, WGN>(g, BM, BN, KS, 20), run<WM, WN, ST, 1, WGM, WGN>(g, BM, BN, KS, 20), run<WM, WN, ST, 2, WGM, WGN>(g, BM, BN, KS, 20));
#define CFGH(NAME, BM, BN, WGM, WGN, WM, WN, ST, KS)                                                         \
        report(NAME, BM, BN, KS, run<WM, WN, ST, 0, WGM, WGN, 2, true>(g, BM, BN, KS, 20), run<WM, WN, ST, 1, WGM, WGN, 2, true>(g, BM, BN, KS, 20), run<WM, WN, ST, 2, WGM, WGN, 2, true>(g, BM, BN, KS, 20));
        CFGH("f16 64x64x2", 64, 64, 2, 2, 1, 1, 2, ksn)
        CFGH("f16 64x64x3", 64, 64, 2, 2, 1, 1, 3, ksn)
        CFGH("f16 64x64x4", 64, 64, 2, 2, 1, 1, 4, ksn)
        CFGH("f16 64x128x3", 64, 128, 2, 2, 1, 2, 3, ksn)
        CFGH("f16 128x64x3", 128, 64, 2, 2, 2, 1, 3, ksn)
        CFGH("f16 128x128x3", 128, 128, 2, 2, 2, 2, 3, ksn)
        CFGH("f16 128x128x4", 128, 128, 2, 2, 2, 2, 4, ksn)
        CFGH("f16 128x128x3 8w", 128, 128, 4, 2, 1, 2, 3, ksn)
        CFGH("f16 128x96x4 4w", 128, 96, 4, 1, 1, 3, 4, ksn)
        CFGW("128x96x3 4w", 128, 96, 4, 1, 1, 3, 3, ksn)
        CFGW("128x96x2 4w", 128, 96, 4, 1, 1, 3, 2, ksn)
        CFGW("128x128x2 8w", 128, 128, 4, 2, 1, 2, 2, ksn)
        CFGW("128x128x3 8w", 128, 128, 4, 2, 1, 2, 3, ksn)
        CFGW("128x192x2 8w", 128, 192, 4, 2, 1, 3, 2, ksn)
        CFG("64x64x2", 64, 64, 1, 1, 2, ksn)
        CFG("64x64x3", 64, 64, 1, 1, 3, ksn)
        CFG("64x128x2", 64, 128, 1, 2, 2, ksn)
        CFG("64x128x3", 64, 128, 1, 2, 3, ksn)
        CFG("128x128x2", 128, 128, 2, 2, 2, ksn)
        CFG("128x128x3", 128, 128, 2, 2, 3, ksn)
        {   // cold weights: 24 rotating buffers (> 256 MB in total for the wide shapes)
            std::vector<uint16_t*> Bs(24);
            for (auto& b : Bs) { hipMalloc(&b, bp * 6); hipMemset(b, 0, bp * 6); }
            hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            uint32_t* sink; hipMalloc(&sink, 4);
#define COLD(NAME, BM, BN, WM, WN, ST, KS)                                                                   \
            printf("%-9s %-14s ks=%d  COLD weights %7.1f us | with prefetch of the next launch's weights %7.1f us\n", sh.name, NAME, KS, \
                   run_cold<WM, WN, ST>(g, BM, BN, KS, 48, Bs, false, s2, sink), run_cold<WM, WN, ST>(g, BM, BN, KS, 48, Bs, true, s2, sink));
            printf("%-9s f16 64x64x3  ks=%d  COLD weights %7.1f us\n", sh.name, ksn, run_cold<1, 1, 3, 2, true>(g, 64, 64, ksn, 48, Bs, false, s2, sink));
            COLD("64x64x2", 64, 64, 1, 1, 2, ksn)
            COLD("64x64x3", 64, 64, 1, 1, 3, ksn)
            COLD("64x128x3", 64, 128, 1, 2, 3, ksn)
            COLD("128x128x3", 128, 128, 2, 2, 3, ksn)
            for (auto& b : Bs) hipFree(b);
            hipFree(sink); hipStreamDestroy(s2);
        }
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
