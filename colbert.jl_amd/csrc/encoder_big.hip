// encoder_big.hip -- the 256 x 256 tile of gemm_planes2_kernel with FOUR waves of 128 x 128 outputs each instead of eight of
// 64 x 128 (round 5).  With three MFMA products per fp32 product on two 16-bit planes per operand, a wave's 32-deep step reads
// NS (WM + WN) x 2 fragments of 1 KB from LDS for 3 WM WN x 2 MFMAs: 0.50 reads per MFMA at 2 x 4 tiles per wave -- the eight
// waves of a work-group then ask the CU's LDS for 192 KB per 1 536 MFMA cycles, its whole bandwidth -- and 0.33 at 4 x 4.
// Sixteen accumulator tiles are 256 registers: they live in the ACCUMULATION registers (the MFMA's native AGPR form), which the
// rest of the library gives up for -amdgpu-mfma-vgpr-form (its kernels want their MFMA results in plain VGPRs for the VALU
// epilogues that follow every few MFMAs).  Hence a translation unit of its own, compiled WITHOUT that flag (Makefile).
// MEASURED (profiles/r05_experiments.md): bit-identical to the eight-wave tile and slower -- attention-output Linear 2.72 against
// 1.55 ms per 64 x 300 batch, FFN-out 4.93 against 3.45: one wave per SIMD has nothing to cover the barrier and the DMA wait of
// every 32-deep step with.  The LDS reads were not what bounds the eight-wave tile.  Off (COLBERT_ENC_WIDE_WAVES=1 selects it).
#include "codec_kernels.hpp"
#include "encoder_kernels.hpp"

namespace clb {

// LN: 0 plain, 1 consumer, 2 producer (gemm_planes2_kernel's comment).  fp16 planes, two stages: 128 KB of LDS, one
// work-group of 256 threads per CU.  grid: gemm_planes_grid(M, N, 256, 256, ksplit).
bool launch_planes2_wide(hipStream_t st, int ln_mode, const GemmPArgs& g, unsigned grid) {
    const size_t lds = (size_t)2 * 2 * (256 + 256) * 64;
#define CLB_WIDE_CASE(MODE_)                                                                                          \
    if (ln_mode == MODE_) {                                                                                           \
        auto kern = gemm_planes2_kernel<2, 2, 4, 4, 2, 2, 0, true, MODE_>;                                            \
        allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);                                             \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, g);                                                  \
        return true;                                                                                                  \
    }
    CLB_WIDE_CASE(0) CLB_WIDE_CASE(1) CLB_WIDE_CASE(2)
#undef CLB_WIDE_CASE
    return false;
}

}  // namespace clb
