// gemm_planes_kernel (csrc/encoder_kernels.hpp) on the Linear shapes of a PASSAGE batch (64 x 300 tokens: M = 19 200), f16x3
// planes: which work-group tile keeps the matrix pipe fed when there is no shortage of tiles?  Each line: the full kernel,
// the MFMA / LDS-read side alone, the DMA side alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../colbert.jl_amd/csrc gemm_planes_bigm_bench.hip -o gemm_planes_bigm_bench
#include "approx_kernels.hpp"
#include "encoder_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace clb;

template <int WGM, int WGN, int WM, int WN, int ST, int ABL, int V = 1>
float run(const GemmPArgs& g, int reps) {
    constexpr int bm = 32 * WM * WGM, bn = 32 * WN * WGN;
    const dim3 grid((unsigned)gemm_planes_grid(g.M, g.N, bm, bn, 1));
    const size_t lds = (size_t)ST * 2 * (bm + bn) * 64;
    void (*kern)(GemmPArgs);
    if constexpr (V == 2) kern = gemm_planes2_kernel<WGM, WGN, WM, WN, 2, ST, ABL, true>;
    else kern = gemm_planes_kernel<WGM, WGN, WM, WN, 2, ST, ABL, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, 0, g);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), lds, 0, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return -1.f; }
    return ms / reps * 1e3f;
}

int main() {
    const int Mrows = getenv("GPB_M") ? atoi(getenv("GPB_M")) : 19200;
    struct Shape { const char* name; int M, N, K; } shapes[] = {{"qkv", Mrows, 2304, 768}, {"attn_out", Mrows, 768, 768},
                                                               {"ffn_in", Mrows, 3072, 768}, {"ffn_out", Mrows, 768, 3072}};
    for (auto& sh : shapes) {
        const int64_t ap = (int64_t)sh.M * sh.K, bp = (int64_t)sh.N * sh.K;
        uint16_t *A, *B; float* C;
        hipMalloc(&A, ap * 4); hipMalloc(&B, bp * 4); hipMalloc(&C, sizeof(float) * (size_t)sh.M * sh.N);
        {
            std::vector<uint16_t> ha(ap * 2), hb(bp * 2);
            uint32_t x = 12345u;
            for (auto& v : ha) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000u + ((x >> 16) & 0x0fffu)) | (uint16_t)((x >> 3) & 0x8000u); }
            for (auto& v : hb) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3000u + ((x >> 16) & 0x0fffu)) | (uint16_t)((x >> 3) & 0x8000u); }
            hipMemcpy(A, ha.data(), ap * 4, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), bp * 4, hipMemcpyHostToDevice);
        }
        GemmPArgs g{A, B, ap, bp, getenv("GPB_NOC") ? nullptr : C, nullptr, nullptr, nullptr, 0, sh.M, sh.N, sh.K, sh.N, 0, 1, 1.0f};
        const double flop = 3.0 * 2.0 * sh.M * sh.N * sh.K;
#define CFG(NAME, WGM, WGN, WM, WN, ST)                                                                                           \
        {                                                                                                                         \
            const float us = run<WGM, WGN, WM, WN, ST, 0>(g, 10), u1 = run<WGM, WGN, WM, WN, ST, 1>(g, 10), u2 = run<WGM, WGN, WM, WN, ST, 2>(g, 10); \
            constexpr int bm = 32 * WM * WGM, bn = 32 * WN * WGN;                                                                 \
            const double bytes = (double)((sh.N + bn - 1) / bn) * ((sh.M + bm - 1) / bm) * (sh.K / 32) * 2.0 * (bm + bn) * 64;    \
            printf("%-9s %-22s %7.1f us  %6.0f TF (%.2f of 2.5 PF)  L2->LDS %5.1f TB/s | no-DMA %7.1f us | no-MFMA %7.1f us\n", sh.name, NAME, us, \
                   flop / us / 1e6, flop / us / 1e6 / 2500.0, bytes / us / 1e6, u1, u2);                                          \
            fflush(stdout);                                                                                                       \
        }
        CFG("128x128x2 4w", 2, 2, 2, 2, 2)
#define CFG2(NAME, WGM, WGN, WM, WN, ST)                                                                                          \
        {                                                                                                                         \
            const float us = run<WGM, WGN, WM, WN, ST, 0, 2>(g, 10), u1 = run<WGM, WGN, WM, WN, ST, 1, 2>(g, 10), u2 = run<WGM, WGN, WM, WN, ST, 2, 2>(g, 10), u3 = run<WGM, WGN, WM, WN, ST, 3, 2>(g, 10); \
            constexpr int bm = 32 * WM * WGM, bn = 32 * WN * WGN;                                                                 \
            const double bytes = (double)((sh.N + bn - 1) / bn) * ((sh.M + bm - 1) / bm) * (sh.K / 32) * 2.0 * (bm + bn) * 64;    \
            printf("%-9s v2 %-19s %7.1f us  %6.0f TF (%.2f of 2.5 PF)  L2->LDS %5.1f TB/s | no-DMA %7.1f us | no-MFMA %7.1f us | no-DMA, no barrier %7.1f us\n", sh.name, NAME, us, \
                   flop / us / 1e6, flop / us / 1e6 / 2500.0, bytes / us / 1e6, u1, u2, u3);                                          \
            fflush(stdout);                                                                                                       \
        }
        if (Mrows <= 4096) {
            CFG("64x64x2 4w", 2, 2, 1, 1, 2)
            CFG2("64x64x2 4w", 2, 2, 1, 1, 2)
            CFG2("64x64x3 4w", 2, 2, 1, 1, 3)
            CFG2("128x64x2 4w", 2, 2, 2, 1, 2)
            CFG2("128x64x3 4w", 2, 2, 2, 1, 3)
            CFG2("64x128x2 4w", 2, 2, 1, 2, 2)
            CFG2("64x128x3 4w", 2, 2, 1, 2, 3)
            CFG2("128x128x4 4w", 2, 2, 2, 2, 4)
            CFG2("128x96x2 4w(4x1,1x3)", 4, 1, 1, 3, 2)
            CFG2("128x96x3 4w(4x1,1x3)", 4, 1, 1, 3, 3)
            CFG2("128x96x4 4w(4x1,1x3)", 4, 1, 1, 3, 4)
            CFG2("96x128x3 4w(1x4,3x1)", 1, 4, 3, 1, 3)
            CFG2("128x96x3 8w(4x2,1x?)", 4, 2, 1, 1, 3)
        }
        CFG2("128x128x2 4w", 2, 2, 2, 2, 2)
        CFG2("128x128x3 4w", 2, 2, 2, 2, 3)
        CFG2("128x128x4 4w", 2, 2, 2, 2, 4)
        CFG2("128x128x5 4w", 2, 2, 2, 2, 5)
        CFG2("256x128x3 8w(4x2)", 4, 2, 2, 2, 3)
        CFG2("128x256x3 8w(2x4)", 2, 4, 2, 2, 3)
        CFG2("256x128x2 8w(4x2)", 4, 2, 2, 2, 2)
        CFG2("128x256x2 8w(2x4)", 2, 4, 2, 2, 2)
        CFG2("256x256x2 8w(2x4,4x2)", 2, 4, 4, 2, 2)
        CFG2("256x256x2 8w(4x2,2x4)", 4, 2, 2, 4, 2)
        if (g.C && sh.N == 768 && sh.K == 768) {     // the two forms agree (same products; the accumulation order inside an MFMA may differ)
            std::vector<float> c1((size_t)sh.M * sh.N), c2((size_t)sh.M * sh.N);
            run<2, 2, 2, 2, 2, 0, 1>(g, 1); hipDeviceSynchronize();
            hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
            hipMemset(C, 0, c1.size() * 4);
            run<2, 2, 2, 2, 2, 0, 2>(g, 1); hipDeviceSynchronize();
            hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost);
            double md = 0, mx = 0; size_t nz = 0;
            for (size_t q = 0; q < c1.size(); ++q) { md = fmax(md, fabs((double)c1[q] - c2[q])); mx = fmax(mx, fabs((double)c1[q])); nz += c2[q] != 0.f; }
            printf("check: max |old - new| = %.3g (max |C| %.3g, nonzero %zu of %zu)\n", md, mx, nz, c1.size());
        }
        if (!getenv("GPB_SHORT")) {
        CFG("128x128x3 4w", 2, 2, 2, 2, 3)
        CFG("256x128x2 8w(4x2)", 4, 2, 2, 2, 2)
        CFG("256x128x3 8w(4x2)", 4, 2, 2, 2, 3)
        CFG("128x256x2 8w(2x4)", 2, 4, 2, 2, 2)
        CFG("256x256x2 8w(2x4,4x2)", 2, 4, 4, 2, 2)
        CFG("256x256x2 8w(4x2,2x4)", 4, 2, 2, 4, 2)
        CFG("256x256x2 16w(4x4)", 4, 4, 2, 2, 2)
        CFG("256x128x2 4w(2x2,4x2)", 2, 2, 4, 2, 2)
        CFG("256x128x3 4w(2x2,4x2)", 2, 2, 4, 2, 3)
        }
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
