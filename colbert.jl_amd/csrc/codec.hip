// codec.hip -- stand-alone codec / index-build / encoder-epilogue entry points of the C ABI
// (include/colbert_hip.h).  Host buffers in, host buffers out; all compute on the selected device.
#include <algorithm>
#include <cmath>

#include "codec_kernels.hpp"
#include "sort.hpp"

using namespace clb;

namespace {

inline int blocks_for(int64_t n, int bs = 256) { return (int)std::max<int64_t>(1, (n + bs - 1) / bs); }

struct Stream {
    hipStream_t st = nullptr;
    int init() {
        CLB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return CLB_OK;
    }
    ~Stream() {
        if (st) (void)hipStreamDestroy(st);
    }
};

// Scratch of the bf16x3 nearest-centroid path, kept across calls (k-means calls it once per iteration).
struct NearestScratch {
    DevBuf hi, lo, bias, partial, cn;    // hi: the bf16 hi plane, or the one fp16 plane of the single-product lists (lo unused then)
    DevBuf ovf_list, ovf_count, ovf_keys;   // points whose candidate lists overflowed (near / mass ties): re-scored against all centroids
    // second tier (single-product lists only): the undecided points as a dense block, decided by the three-product lists
    DevBuf hi3, lo3, cn3, xc, codes_c, partial_c, ovf2_list, ovf2_count, ovf2_keys;
};

// device-side: codes (1-based UInt32) of n embeddings against K centroids; MODE 0 argmax dot, 1 k-means.
// dim 128, K >= 32: approximate scores at the 16-bit MFMA rate with a proven bound -- ONE fp16 product per fp32 product since
// round 5 (nearest_top_f16_dma_kernel: 512 points per staged centroid tile, the measured conversion errors of points and
// centroids in the refine margin; 1 M passages: k-means 18.6 -> 7.1 s, compress 5.6 -> 2.3 s), the three-product bf16 split
// of rounds 2-4 (256 points per tile) when a centroid component is outside the fp16 range -- then an exact re-evaluation
// of the few candidates per point with the canonical arithmetic: the same codes as the all-fp32 kernel (kept as the small-K /
// other-dim path), bit for bit.
template <int MODE>
int nearest_centroids(hipStream_t st, const float* dC, const float* dc2, int dim, int K, const float* dX, int64_t n,
                      uint32_t* dOut, NearestScratch* scratch = nullptr) {
    if (n == 0) return CLB_OK;
    const bool force_fp32 = CLB_KNOB("CLB_DEBUG_NEAREST_FP32", 0) != 0;
    if (dim == kDim && K >= 32 && scratch && !force_fp32) {
        NearestScratch& w = *scratch;
        const size_t cel = (size_t)K * kDim;
        const int kpad = (K + 31) / 32 * 32 + 32;
        CLB_TRY(w.hi.ensure(sizeof(uint16_t) * cel));
        CLB_TRY(w.bias.ensure(sizeof(float) * kpad));
        CLB_TRY(w.cn.ensure(2 * sizeof(unsigned int)));
        CLB_HIP(hipMemsetAsync(w.cn.p, 0, 2 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(max_row_norm_kernel, dim3(std::max(1, std::min(1024, K / 256))), dim3(256), 0, st, dC, K,
                           w.cn.as<unsigned int>());
        // One fp16 product per fp32 product (nearest_top_f16_kernel) unless a centroid component is outside the fp16 range
        // (the measured max ||c - fp16(c)|| comes back infinite: 4 bytes and one wait per call) or a comparison run asks
        // for the three-product bf16 split, COLBERT_NEAREST_PRODUCTS=3
        const char* products = getenv("COLBERT_NEAREST_PRODUCTS");        // read per call: tests and comparison runs switch it
        const bool want_x1 = !(products && strcmp(products, "3") == 0);
        bool x1 = false;
        if (want_x1) {
            float dc = 0.f;
            hipLaunchKernelGGL(max_row_f16_err_kernel, dim3(std::max(1, std::min(1024, K / 4))), dim3(256), 0, st, dC, K,
                               w.cn.as<unsigned int>() + 1);
            CLB_HIP(hipMemcpyAsync(&dc, w.cn.as<unsigned int>() + 1, sizeof dc, hipMemcpyDeviceToHost, st));
            CLB_HIP(hipStreamSynchronize(st));
            x1 = dc > 0.f && std::isfinite(dc);
            if (!x1) CLB_HIP(hipMemsetAsync(w.cn.as<unsigned int>() + 1, 0, sizeof(unsigned int), st));
        }
        const char* staging = CLB_ENV("COLBERT_NEAREST_STAGING");         // "registers": the first form of the kernel (tuning builds)
        const bool dma = x1 && !(staging && strcmp(staging, "registers") == 0);
        if (dma) {          // the tiled table of nearest_top_f16_dma_kernel: whole tiles, the last one padded with copies of row K - 1
            const int64_t n_chunks = (int64_t)((K + 31) / 32) * 512;
            CLB_TRY(w.hi.ensure(16 * (size_t)n_chunks));
            hipLaunchKernelGGL(to_f16_tiled_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, dC, K, w.hi.as<uint16_t>(), n_chunks);
        } else if (x1) {
            hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)((cel + 255) / 256)), dim3(256), 0, st, dC, w.hi.as<uint16_t>(), (int64_t)cel);
        } else {
            CLB_TRY(w.lo.ensure(sizeof(uint16_t) * cel));
            hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((cel + 255) / 256)), dim3(256), 0, st, dC,
                               w.hi.as<uint16_t>(), w.lo.as<uint16_t>(), (int64_t)cel);
        }
        if (MODE == 1)
            hipLaunchKernelGGL(half_neg_kernel, dim3((kpad + 255) / 256), dim3(256), 0, st, dc2, K, w.bias.as<float>(), kpad);
        const int n_tiles = (K + 31) / 32;
        const int64_t chunk_max = (int64_t)1 << 22;                     // points per launch: 256 MB of group lists
        CLB_TRY(w.partial.ensure(sizeof(ValIdx) * (size_t)((std::min(n, chunk_max) + 31) / 32) * 2 * 32 * kTopPartial));
        CLB_TRY(w.ovf_list.ensure(sizeof(uint32_t) * (size_t)std::min(n, chunk_max)));
        CLB_TRY(w.ovf_count.ensure(sizeof(unsigned int)));
        CLB_TRY(w.ovf_keys.ensure(sizeof(unsigned long long) * (size_t)std::min(n, chunk_max)));
        const size_t lds = 2 * 2 * 32 * kRowBytes16;
        bool tier2_ready = false;       // the bf16 split of THIS call's centroids exists (built when the first chunk needs it)
        for (int64_t p0 = 0; p0 < n; p0 += chunk_max) {
            const int64_t m = std::min(chunk_max, n - p0);
            const int groups32 = (int)((m + 31) / 32);                  // "queries" of 32 points
            const dim3 grid(1, (unsigned)((groups32 + kMqQueries - 1) / kMqQueries));
            constexpr int kNq = 4;                                      // groups of 32 points per wave of the single-product kernel
            const dim3 grid1((unsigned)((groups32 + 4 * kNq - 1) / (4 * kNq)));
if (dma && MODE == 1)
                hipLaunchKernelGGL((nearest_top_f16_dma_kernel<true>), grid1, dim3(256), 3 * 8192, st, w.hi.as<uint16_t>(),
                                   dX + (size_t)p0 * kDim, w.partial.as<ValIdx>(), K, groups32, n_tiles, w.bias.as<float>(), m);
            else if (dma)
                hipLaunchKernelGGL((nearest_top_f16_dma_kernel<false>), grid1, dim3(256), 3 * 8192, st, w.hi.as<uint16_t>(),
                                   dX + (size_t)p0 * kDim, w.partial.as<ValIdx>(), K, groups32, n_tiles, (const float*)nullptr, m);
#ifdef CLB_ABLATIONS
            else if (x1 && MODE == 1)
                hipLaunchKernelGGL((nearest_top_f16_kernel<true, kNq>), grid1, dim3(256), lds / 2, st, w.hi.as<uint16_t>(),
                                   dX + (size_t)p0 * kDim, w.partial.as<ValIdx>(), K, groups32, n_tiles, w.bias.as<float>(), m);
            else if (x1)
                hipLaunchKernelGGL((nearest_top_f16_kernel<false, kNq>), grid1, dim3(256), lds / 2, st, w.hi.as<uint16_t>(),
                                   dX + (size_t)p0 * kDim, w.partial.as<ValIdx>(), K, groups32, n_tiles, (const float*)nullptr, m);
#endif
            else if (MODE == 1)
                hipLaunchKernelGGL((centroid_top_bf16x3_mq_kernel<false, true>), grid, dim3(256), lds, st,
                                   w.hi.as<uint16_t>(), w.lo.as<uint16_t>(), dX + (size_t)p0 * kDim,
                                   w.partial.as<ValIdx>(), (uint32_t*)nullptr, K, 32, groups32, n_tiles,
                                   w.bias.as<float>(), m);
            else
                hipLaunchKernelGGL((centroid_top_bf16x3_mq_kernel<false, false>), grid, dim3(256), lds, st,
                                   w.hi.as<uint16_t>(), w.lo.as<uint16_t>(), dX + (size_t)p0 * kDim,
                                   w.partial.as<ValIdx>(), (uint32_t*)nullptr, K, 32, groups32, n_tiles,
                                   (const float*)nullptr, m);
            CLB_HIP(hipMemsetAsync(w.ovf_count.p, 0, sizeof(unsigned int), st));
            hipLaunchKernelGGL(nearest_refine_kernel<MODE>, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, st,
                               w.partial.as<ValIdx>(), dC, dc2, dX + (size_t)p0 * kDim, m, K, w.cn.as<unsigned int>(),
                               dOut + p0, w.ovf_list.as<uint32_t>(), w.ovf_count.as<unsigned int>(), w.ovf_keys.as<unsigned long long>());
            // The undecided points.  Three-product lists: straight to the exhaustive scan (the list is normally empty: the launch
            // reads a zero and ends).  Single-product lists: a few dozen points per million on well-spread data, but 2.5 % on nearly
            // degenerate embeddings (a random-weight encoder), where the exhaustive scan of them cost as much as the lists of all
            // points -- so more than a handful go through the three-product lists first (their margin is five times tighter) and
            // only what THOSE cannot decide is scanned.  One 4-byte read-back per chunk of up to 4 M points.
            const int slices = std::max(1, std::min(n_tiles / 8, 128));
            unsigned int undecided = 0, undecided2 = 0;
            if (x1) {
                CLB_HIP(hipMemcpyAsync(&undecided, w.ovf_count.p, sizeof undecided, hipMemcpyDeviceToHost, st));
                CLB_HIP(hipStreamSynchronize(st));
            }
            if (x1 && undecided > 64) {
                const int64_t mc = undecided;
                if (!tier2_ready) {
                    CLB_TRY(w.hi3.ensure(sizeof(uint16_t) * cel));
                    CLB_TRY(w.lo3.ensure(sizeof(uint16_t) * cel));
                    CLB_TRY(w.cn3.ensure(2 * sizeof(unsigned int)));
                    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((cel + 255) / 256)), dim3(256), 0, st, dC,
                                       w.hi3.as<uint16_t>(), w.lo3.as<uint16_t>(), (int64_t)cel);
                    CLB_HIP(hipMemsetAsync(w.cn3.p, 0, 2 * sizeof(unsigned int), st));          // [1] = 0: the three-product margin
                    CLB_HIP(hipMemcpyAsync(w.cn3.p, w.cn.p, sizeof(unsigned int), hipMemcpyDeviceToDevice, st));
                    tier2_ready = true;
                }
                CLB_TRY(w.xc.ensure(sizeof(float) * kDim * (size_t)mc));
                CLB_TRY(w.codes_c.ensure(sizeof(uint32_t) * (size_t)mc));
                CLB_TRY(w.partial_c.ensure(sizeof(ValIdx) * (size_t)((mc + 31) / 32) * 2 * 32 * kTopPartial));
                CLB_TRY(w.ovf2_list.ensure(sizeof(uint32_t) * (size_t)mc));
                CLB_TRY(w.ovf2_keys.ensure(sizeof(unsigned long long) * (size_t)mc));
                CLB_TRY(w.ovf2_count.ensure(sizeof(unsigned int)));
                hipLaunchKernelGGL(gather_points_kernel, dim3(blocks_for(mc * 32)), dim3(256), 0, st, dX + (size_t)p0 * kDim,
                                   w.ovf_list.as<uint32_t>(), mc, w.xc.as<float>());
                const int groups_c = (int)((mc + 31) / 32);
                const dim3 grid_c(1, (unsigned)((groups_c + kMqQueries - 1) / kMqQueries));
                if (MODE == 1)
                    hipLaunchKernelGGL((centroid_top_bf16x3_mq_kernel<false, true>), grid_c, dim3(256), lds, st,
                                       w.hi3.as<uint16_t>(), w.lo3.as<uint16_t>(), w.xc.as<float>(), w.partial_c.as<ValIdx>(),
                                       (uint32_t*)nullptr, K, 32, groups_c, n_tiles, w.bias.as<float>(), mc);
                else
                    hipLaunchKernelGGL((centroid_top_bf16x3_mq_kernel<false, false>), grid_c, dim3(256), lds, st,
                                       w.hi3.as<uint16_t>(), w.lo3.as<uint16_t>(), w.xc.as<float>(), w.partial_c.as<ValIdx>(),
                                       (uint32_t*)nullptr, K, 32, groups_c, n_tiles, (const float*)nullptr, mc);
                CLB_HIP(hipMemsetAsync(w.ovf2_count.p, 0, sizeof(unsigned int), st));
                hipLaunchKernelGGL(nearest_refine_kernel<MODE>, dim3((unsigned)((mc + 15) / 16)), dim3(256), 0, st,
                                   w.partial_c.as<ValIdx>(), dC, dc2, w.xc.as<float>(), mc, K, w.cn3.as<unsigned int>(),
                                   w.codes_c.as<uint32_t>(), w.ovf2_list.as<uint32_t>(), w.ovf2_count.as<unsigned int>(),
                                   w.ovf2_keys.as<unsigned long long>());
                const int pairs = (int)std::max<int64_t>(1, std::min<int64_t>(2048 / slices, (mc + 63) / 64));
                hipLaunchKernelGGL(nearest_centroid_mfma_list_kernel<MODE>, dim3(pairs, slices), dim3(128),
                                   2 * 32 * kCentTileStride * sizeof(float), st, dC, dc2, K, w.xc.as<float>(),
                                   w.ovf2_list.as<uint32_t>(), w.ovf2_count.as<unsigned int>(), w.ovf2_keys.as<unsigned long long>());
                hipLaunchKernelGGL(nearest_list_finalize_kernel, dim3(64), dim3(256), 0, st, w.ovf2_list.as<uint32_t>(),
                                   w.ovf2_count.as<unsigned int>(), w.ovf2_keys.as<unsigned long long>(), w.codes_c.as<uint32_t>());
                hipLaunchKernelGGL(scatter_codes_kernel, dim3(blocks_for(mc)), dim3(256), 0, st, w.ovf_list.as<uint32_t>(), mc,
                                   w.codes_c.as<uint32_t>(), dOut + p0);
                if (CLB_ENV("COLBERT_DEBUG_NEAREST")) {
                    CLB_HIP(hipMemcpyAsync(&undecided2, w.ovf2_count.p, sizeof undecided2, hipMemcpyDeviceToHost, st));
                    CLB_HIP(hipStreamSynchronize(st));
                }
            } else {
                // on the fp32 MFMA, all K centroids per point, the centroid tiles dealt to `slices` work-groups per pair of 32-point tiles
                const int pairs = (int)std::max<int64_t>(1, std::min<int64_t>(2048 / slices, (m + 63) / 64));
                hipLaunchKernelGGL(nearest_centroid_mfma_list_kernel<MODE>, dim3(pairs, slices), dim3(128),
                                   2 * 32 * kCentTileStride * sizeof(float), st, dC, dc2, K, dX + (size_t)p0 * kDim,
                                   w.ovf_list.as<uint32_t>(), w.ovf_count.as<unsigned int>(), w.ovf_keys.as<unsigned long long>());
                hipLaunchKernelGGL(nearest_list_finalize_kernel, dim3(64), dim3(256), 0, st, w.ovf_list.as<uint32_t>(),
                                   w.ovf_count.as<unsigned int>(), w.ovf_keys.as<unsigned long long>(), dOut + p0);
            }
            if (CLB_ENV("COLBERT_DEBUG_NEAREST")) {      // how many points the lists left undecided (a wait per chunk: debugging only)
                unsigned int cnt = 0;
                CLB_HIP(hipMemcpyAsync(&cnt, w.ovf_count.p, sizeof cnt, hipMemcpyDeviceToHost, st));
                CLB_HIP(hipStreamSynchronize(st));
                fprintf(stderr, "nearest_centroids<%d>: %lld points, K = %d, %s lists, %u undecided, %u of them also by the bf16 x3 lists "
                        "(scanned against all centroids: %u)\n", MODE, (long long)m, K, x1 ? "fp16 x1" : "bf16 x3", cnt,
                        x1 && cnt > 64 ? undecided2 : 0u, x1 && cnt > 64 ? undecided2 : cnt);
            }
        }
    } else if (dim == kDim) {
        const int64_t ptiles = (n + 31) / 32;
        hipLaunchKernelGGL(nearest_centroid_mfma_kernel<MODE>, dim3((unsigned)((ptiles + 1) / 2)), dim3(128),
                           2 * 32 * kCentTileStride * sizeof(float), st, dC, dc2, K, dX, n, dOut);
    } else {
        hipLaunchKernelGGL(nearest_centroid_generic_kernel<MODE>, dim3(blocks_for(n, 64)), dim3(64), 0, st, dC, dc2,
                           dim, K, dX, n, dOut);
    }
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

// Statistics.quantile (type 7) on a sorted device array: reads the two neighbours it needs
int quantile7_device(const float* d_sorted, int64_t n, double p, float* out) {
    const double m = 1.0 + p * (1.0 - 1.0 - 1.0);
    const double aleph = (double)n * p + m;
    int64_t j = (int64_t)std::trunc(aleph);
    j = std::min<int64_t>(std::max<int64_t>(j, 1), n - 1);
    double g = aleph - (double)j;
    g = std::min(std::max(g, 0.0), 1.0);
    float ab[2];
    if (n == 1) {
        CLB_HIP(hipMemcpy(ab, d_sorted, sizeof(float), hipMemcpyDeviceToHost));
        ab[1] = ab[0];
    } else {
        CLB_HIP(hipMemcpy(ab, d_sorted + (j - 1), 2 * sizeof(float), hipMemcpyDeviceToHost));
    }
    const float diff = ab[1] - ab[0];
    *out = (float)((double)ab[0] + g * (double)diff);
    return CLB_OK;
}

}  // namespace

extern "C" {

int clb_normalize_columns(int device, float* X, int64_t dim, int64_t n) {
    if (dim < 0 || n < 0) return fail(CLB_EARGUMENT, "negative size");
    CLB_TRY(use_device(device));
    if (dim == 0 || n == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    DevBuf d;
    CLB_TRY(upload(d, X, sizeof(float) * dim * n, s.st));
    hipLaunchKernelGGL(normalize_columns_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, s.st, d.as<float>(), (int)dim, n);
    CLB_HIP(hipMemcpyAsync(X, d.p, sizeof(float) * dim * n, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

int clb_decompress(int device, int64_t dim, int nbits, const float* centroids, int64_t K,
                   const float* bucket_weights, int64_t n_weights, const uint32_t* codes, int64_t n_codes,
                   const uint8_t* residuals, int64_t res_rows, int64_t res_cols, float* out) {
    // argument contract of decompress / decompress_residuals (residual.jl:701-706, 763-768)
    if (n_codes != res_cols) return fail(CLB_EDOMAIN, "The number of codes should be equal to the number of residual embeddings!");
    for (int64_t e = 0; e < n_codes; ++e)
        if (codes[e] < 1 || (int64_t)codes[e] > K) return fail(CLB_EDOMAIN, "All the codes must be in the valid range of centroid IDs!");
    if (dim % 8 != 0) return fail(CLB_EDOMAIN, "dim should be a multiple of 8!");
    if (nbits < 1 || res_rows != dim / 8 * nbits) return fail(CLB_EDOMAIN, "The dimension each residual in binary_residuals should be (dim / 8) * nbits!");
    if (n_weights != ((int64_t)1 << nbits)) return fail(CLB_EDOMAIN, "bucket_weights should have length 2^nbits!");
    if (nbits != 1 && nbits != 2 && nbits != 4 && nbits != 8)
        return fail(CLB_EUNSUPPORTED, "the HIP codec supports nbits in {1,2,4,8} (got %d)", nbits);
    CLB_TRY(use_device(device));
    if (n_codes == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    DevBuf dC, dW, dCodes, dRes, dOut;
    CLB_TRY(upload(dC, centroids, sizeof(float) * dim * K, s.st));
    CLB_TRY(upload(dW, bucket_weights, sizeof(float) * n_weights, s.st));
    CLB_TRY(upload(dCodes, codes, sizeof(uint32_t) * n_codes, s.st));
    CLB_TRY(upload(dRes, residuals, (size_t)res_rows * res_cols, s.st));
    CLB_TRY(dOut.alloc(sizeof(float) * dim * n_codes));
    if (dim == kDim && nbits <= 4) {
        const int grid = (int)std::min<int64_t>(2048, (n_codes + 63) / 64);
        switch (nbits) {
            case 1: hipLaunchKernelGGL(decompress_dim128_kernel<1>, dim3(grid), dim3(256), 0, s.st, dC.as<float>(), dW.as<float>(), dCodes.as<uint32_t>(), dRes.as<uint8_t>(), n_codes, dOut.as<float>()); break;
            case 2: hipLaunchKernelGGL(decompress_dim128_kernel<2>, dim3(grid), dim3(256), 0, s.st, dC.as<float>(), dW.as<float>(), dCodes.as<uint32_t>(), dRes.as<uint8_t>(), n_codes, dOut.as<float>()); break;
            default: hipLaunchKernelGGL(decompress_dim128_kernel<4>, dim3(grid), dim3(256), 0, s.st, dC.as<float>(), dW.as<float>(), dCodes.as<uint32_t>(), dRes.as<uint8_t>(), n_codes, dOut.as<float>()); break;
        }
    } else {
        hipLaunchKernelGGL(decompress_generic_kernel, dim3(blocks_for(n_codes, 64)), dim3(64), 0, s.st, (int)dim, nbits,
                           dC.as<float>(), dW.as<float>(), dCodes.as<uint32_t>(), dRes.as<uint8_t>(), n_codes, dOut.as<float>());
    }
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(out, dOut.p, sizeof(float) * dim * n_codes, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

int clb_maxsim(int device, const float* Q, int64_t dim, int64_t T, const float* D, int64_t n_D,
               const int64_t* pids, int64_t n_pids, const int64_t* doclens, int64_t n_docs, float* scores) {
    std::vector<int64_t> off((size_t)n_pids + 1, 0);
    for (int64_t j = 0; j < n_pids; ++j) {
        if (pids[j] < 1 || pids[j] > n_docs) return fail(CLB_EBOUNDS, "pid %lld outside 1..%lld", (long long)pids[j], (long long)n_docs);
        off[j + 1] = off[j] + doclens[pids[j] - 1];
    }
    if (off[n_pids] != n_D)  // ranking.jl:71-74
        return fail(CLB_EDIMENSION, "The total number of embeddings for pids does not match with the dimension of D!");
    for (int64_t j = 0; j < n_pids; ++j)
        if (off[j + 1] == off[j]) return fail(CLB_EARGUMENT, "reducing over an empty collection is not allowed (passage %lld has no embeddings)", (long long)pids[j]);
    CLB_TRY(use_device(device));
    if (n_pids == 0) return CLB_OK;
    if (T > 1024) return fail(CLB_EUNSUPPORTED, "T > 1024");
    Stream s; CLB_TRY(s.init());
    DevBuf dQ, dD, dOff, dS;
    CLB_TRY(upload(dQ, Q, sizeof(float) * dim * T, s.st));
    CLB_TRY(upload(dD, D, sizeof(float) * dim * n_D, s.st));
    CLB_TRY(upload(dOff, off.data(), sizeof(int64_t) * off.size(), s.st));
    CLB_TRY(dS.alloc(sizeof(float) * n_pids));
    const int bs = (int)std::min<int64_t>(256, (T + 63) / 64 * 64);
    hipLaunchKernelGGL(maxsim_generic_kernel, dim3((unsigned)n_pids), dim3(bs), sizeof(float) * T, s.st, dQ.as<float>(),
                       (int)dim, (int)T, dD.as<float>(), dOff.as<int64_t>(), dS.as<float>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(scores, dS.p, sizeof(float) * n_pids, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

int clb_compress_into_codes(int device, uint32_t* codes, int64_t n_codes, const float* centroids, int64_t dim,
                            int64_t K, const float* embs, int64_t n) {
    if (n_codes != n) return fail(CLB_EDIMENSION, "length(codes) must be equal to the number of embeddings!");
    if (K < 1 && n > 0) return fail(CLB_EARGUMENT, "no centroids");
    CLB_TRY(use_device(device));
    if (n == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    DevBuf dC, dX, dOut;
    CLB_TRY(upload(dC, centroids, sizeof(float) * dim * K, s.st));
    CLB_TRY(upload(dX, embs, sizeof(float) * dim * n, s.st));
    CLB_TRY(dOut.alloc(sizeof(uint32_t) * n));
    NearestScratch nscratch;
    CLB_TRY(nearest_centroids<0>(s.st, dC.as<float>(), nullptr, (int)dim, (int)K, dX.as<float>(), n, dOut.as<uint32_t>(), &nscratch));
    CLB_HIP(hipMemcpyAsync(codes, dOut.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

int clb_compress(int device, const float* centroids, int64_t K, const float* bucket_cutoffs, int64_t n_cutoffs,
                 int64_t dim, int nbits, const float* embs, int64_t n, uint32_t* codes, uint8_t* residuals) {
    if (dim % 8 != 0) return fail(CLB_EDOMAIN, "dims should be a multiple of 8!");
    if (nbits < 1 || nbits > 16 || n_cutoffs != ((int64_t)1 << nbits) - 1) return fail(CLB_EDOMAIN, "length(bucket_cutoffs) should be 2^nbits - 1!");
    if (K < 1 && n > 0) return fail(CLB_EARGUMENT, "no centroids");
    CLB_TRY(use_device(device));
    if (n == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    const int64_t rows = dim / 8 * nbits;
    DevBuf dC, dX, dCut, dCodes, dRes;
    CLB_TRY(upload(dC, centroids, sizeof(float) * dim * K, s.st));
    CLB_TRY(upload(dX, embs, sizeof(float) * dim * n, s.st));
    CLB_TRY(upload(dCut, bucket_cutoffs, sizeof(float) * std::max<int64_t>(n_cutoffs, 1), s.st));
    CLB_TRY(dCodes.alloc(sizeof(uint32_t) * n));
    CLB_TRY(dRes.alloc((size_t)rows * n));
    NearestScratch nscratch;
    CLB_TRY(nearest_centroids<0>(s.st, dC.as<float>(), nullptr, (int)dim, (int)K, dX.as<float>(), n, dCodes.as<uint32_t>(), &nscratch));
    hipLaunchKernelGGL(pack_residuals_kernel, dim3(blocks_for(n * rows)), dim3(256), 0, s.st, dC.as<float>(),
                       dCut.as<float>(), (int)n_cutoffs, (int)dim, nbits, dX.as<float>(), dCodes.as<uint32_t>(), n,
                       dRes.as<uint8_t>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(codes, dCodes.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipMemcpyAsync(residuals, dRes.p, (size_t)rows * n, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

// ---- codec handle: centroids, cutoffs and the nearest-centroid scratch stay on the device across chunks -----------
struct clb_codec {
    int device = 0;
    int64_t dim = 0, K = 0, n_cutoffs = 0;
    int nbits = 0;
    DevBuf dC, dCut;
    NearestScratch nscratch;
    bool split_done = false;
};

int clb_codec_create(int device, int64_t dim, int nbits, int64_t K, const float* centroids,
                     const float* bucket_cutoffs, int64_t n_cutoffs, clb_codec** out) {
    if (!out) return fail(CLB_EARGUMENT, "out is null");
    *out = nullptr;
    if (!centroids || (!bucket_cutoffs && n_cutoffs > 0)) return fail(CLB_EARGUMENT, "null argument");
    if (dim < 8 || dim % 8 != 0) return fail(CLB_EDOMAIN, "dims should be a multiple of 8!");                // residual.jl:523-525
    if (nbits < 1 || nbits > 16 || n_cutoffs != ((int64_t)1 << nbits) - 1)
        return fail(CLB_EDOMAIN, "length(bucket_cutoffs) should be 2^nbits - 1!");
    if (K < 1) return fail(CLB_EARGUMENT, "no centroids");
    CLB_TRY(use_device(device));
    auto* c = new clb_codec();
    c->device = device; c->dim = dim; c->K = K; c->nbits = nbits; c->n_cutoffs = n_cutoffs;
    int rc;
    if ((rc = c->dC.alloc(sizeof(float) * dim * K)) || (rc = c->dCut.alloc(sizeof(float) * std::max<int64_t>(n_cutoffs, 1)))) {
        delete c;
        return rc;
    }
    // hipMemcpyDefault: host or device pointers (the centroids of a device-resident k-means never visit the host)
    if (hipMemcpy(c->dC.p, centroids, sizeof(float) * dim * K, hipMemcpyDefault) != hipSuccess ||
        (n_cutoffs > 0 && hipMemcpy(c->dCut.p, bucket_cutoffs, sizeof(float) * n_cutoffs, hipMemcpyDefault) != hipSuccess)) {
        delete c;
        return fail(CLB_EHIP, "codec upload failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = c;
    return CLB_OK;
}

int clb_codec_destroy(clb_codec* c) {
    if (!c) return CLB_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    delete c;
    return CLB_OK;
}

int clb_codec_compress_device(clb_codec* c, const float* d_embs, int64_t n, uint32_t* d_codes, uint8_t* d_residuals,
                              void* hip_stream) {
    if (!c) return fail(CLB_EARGUMENT, "codec is null");
    if (n < 0) return fail(CLB_EARGUMENT, "negative size");
    if (n == 0) return CLB_OK;
    if (!d_embs || !d_codes || !d_residuals) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(c->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t rows = c->dim / 8 * c->nbits;
    CLB_TRY(nearest_centroids<0>(st, c->dC.as<float>(), nullptr, (int)c->dim, (int)c->K, d_embs, n, d_codes, &c->nscratch));
    hipLaunchKernelGGL(pack_residuals_kernel, dim3(blocks_for(n * rows)), dim3(256), 0, st, c->dC.as<float>(),
                       c->dCut.as<float>(), (int)c->n_cutoffs, (int)c->dim, c->nbits, d_embs, d_codes, n, d_residuals);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

int clb_kmeans(int device, const float* data, int64_t dim, int64_t n, float* centroids, int64_t K,
               int64_t max_iters, float tol, int64_t point_bsize, int32_t* assignments, int64_t* iters_done) {
    if (K < 1 || dim < 1 || point_bsize < 1) return fail(CLB_EDIMENSION, "size(centroids, 2) must be k!");
    if (n >= (int64_t)0xffffffffll) return fail(CLB_EUNSUPPORTED, "n too large");
    CLB_TRY(use_device(device));
    Stream s; CLB_TRY(s.init());
    DevBuf dX, dC, dNew, dC2, dAssign, dOrder, dIota, dKeys, dCounts, dStart, dDelta, dErr, dCnt32;
    NearestScratch nscratch;
    CLB_TRY(upload(dX, data, sizeof(float) * dim * std::max<int64_t>(n, 1), s.st));
    CLB_TRY(upload(dC, centroids, sizeof(float) * dim * K, s.st));
    CLB_TRY(dNew.alloc(sizeof(float) * dim * K));
    CLB_TRY(dC2.alloc(sizeof(float) * K));
    CLB_TRY(dAssign.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dOrder.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dIota.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dKeys.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dCounts.alloc(sizeof(uint32_t) * (K + 1)));
    CLB_TRY(dStart.alloc(sizeof(uint32_t) * (K + 2)));
    CLB_TRY(dCnt32.alloc(sizeof(int) * K));
    CLB_TRY(dDelta.alloc(sizeof(unsigned int)));
    CLB_TRY(dErr.alloc(sizeof(int)));
    if (n > 0) hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n)), dim3(256), 0, s.st, dIota.as<uint32_t>(), n);
    int end_bit = 1;
    while (((int64_t)1 << end_bit) <= K) ++end_bit;
    int64_t it = 0;
    for (it = 0; it < max_iters; ++it) {
        hipLaunchKernelGGL(centroid_sumsq_kernel, dim3(blocks_for(K, 64)), dim3(64), 0, s.st, dC.as<float>(), (int)dim,
                           (int)K, dC2.as<float>());
        CLB_TRY(nearest_centroids<1>(s.st, dC.as<float>(), dC2.as<float>(), (int)dim, (int)K, dX.as<float>(), n,
                                     dAssign.as<uint32_t>(), &nscratch));
        // group the points by cluster, ascending point id inside a cluster (stable)
        CLB_HIP(hipMemsetAsync(dCounts.p, 0, sizeof(uint32_t) * (K + 1), s.st));
        CLB_HIP(hipMemsetAsync(dErr.p, 0, sizeof(int), s.st));
        if (n > 0) {
            hipLaunchKernelGGL(code_histogram_kernel, dim3(blocks_for(n)), dim3(256), 0, s.st, dAssign.as<uint32_t>(), n,
                               (uint32_t)K, dCounts.as<unsigned int>(), dErr.as<int>());
            CLB_TRY(sort_pairs_u32(dAssign.as<uint32_t>(), dKeys.as<uint32_t>(), dIota.as<uint32_t>(),
                                   dOrder.as<uint32_t>(), (size_t)n, end_bit, s.st));
        }
        CLB_TRY(exclusive_scan_u32(dCounts.as<uint32_t>(), dStart.as<uint32_t>(), (size_t)K, s.st));
        hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(blocks_for(K * dim)), dim3(256), 0, s.st, dX.as<float>(),
                           (int)dim, dOrder.as<uint32_t>(), dStart.as<uint32_t>(), (int)K, (int)point_bsize,
                           dNew.as<float>(), dCnt32.as<int>());
        CLB_HIP(hipMemsetAsync(dDelta.p, 0, sizeof(unsigned int), s.st));
        hipLaunchKernelGGL(kmeans_finalize_kernel, dim3(blocks_for(K * dim)), dim3(256), 0, s.st, dNew.as<float>(),
                           dCnt32.as<int>(), dC.as<float>(), (int)dim, (int)K, dDelta.as<unsigned int>());
        CLB_HIP(hipGetLastError());
        unsigned int bits = 0;
        CLB_HIP(hipMemcpyAsync(&bits, dDelta.p, sizeof bits, hipMemcpyDeviceToHost, s.st));
        CLB_HIP(hipStreamSynchronize(s.st));
        float delta;
        memcpy(&delta, &bits, sizeof delta);
        if (delta < tol) { ++it; break; }  // utils.jl:308-311: the previous centroids stay
        CLB_HIP(hipMemcpyAsync(dC.p, dNew.p, sizeof(float) * dim * K, hipMemcpyDeviceToDevice, s.st));
    }
    if (iters_done) *iters_done = it;
    CLB_HIP(hipMemcpyAsync(centroids, dC.p, sizeof(float) * dim * K, hipMemcpyDeviceToHost, s.st));
    if (n > 0 && max_iters > 0)
        CLB_HIP(hipMemcpyAsync(assignments, dAssign.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

// ---- sharded k-means: the points of one shard stay on the device across iterations ----------------------------
struct clb_kmeans_shard {
    int device = 0;
    int64_t dim = 0, n = 0, K = 0, point_bsize = 0;
    int end_bit = 1;
    Stream s;
    const float* X = nullptr;      // the shard's points: dX (uploaded copy) or the caller's device array (create_device)
    DevBuf dX, dC, dNew, dC2, dAssign, dOrder, dIota, dKeys, dCounts, dStart, dErr, dCnt32, dCnt64;
    DevBuf sort_tmp, scan_tmp;     // rocPRIM temporaries (kept: the per-iteration sort and scan are only enqueued)
    DevBuf dNext, dDelta;          // update_device: next centroids and max-abs delta (allocated once: a hipFree per
                                   // iteration would be a device-wide synchronisation under the exchange)
    NearestScratch nscratch;
};

// shared by create (host points, uploaded) and create_device (device points, borrowed)
static int kmeans_shard_create_impl(int device, const float* data, bool on_device, int64_t dim, int64_t n, int64_t K,
                                    int64_t point_bsize, clb_kmeans_shard** out);

int clb_kmeans_shard_create(int device, const float* data, int64_t dim, int64_t n, int64_t K, int64_t point_bsize,
                            clb_kmeans_shard** out) {
    return kmeans_shard_create_impl(device, data, false, dim, n, K, point_bsize, out);
}

int clb_kmeans_shard_create_device(int device, const float* d_data, int64_t dim, int64_t n, int64_t K,
                                   int64_t point_bsize, clb_kmeans_shard** out) {
    return kmeans_shard_create_impl(device, d_data, true, dim, n, K, point_bsize, out);
}

static int kmeans_shard_create_impl(int device, const float* data, bool on_device, int64_t dim, int64_t n, int64_t K,
                                    int64_t point_bsize, clb_kmeans_shard** out) {
    if (!out) return fail(CLB_EARGUMENT, "out is null");
    *out = nullptr;
    if (K < 1 || dim < 1 || point_bsize < 1 || n < 0) return fail(CLB_EDIMENSION, "size(centroids, 2) must be k!");
    if (n >= (int64_t)0xffffffffll) return fail(CLB_EUNSUPPORTED, "n too large");
    CLB_TRY(use_device(device));
    auto* h = new clb_kmeans_shard();
    h->device = device; h->dim = dim; h->n = n; h->K = K; h->point_bsize = point_bsize;
    auto bail = [&](int rc) { delete h; return rc; };
    int rc;
    if ((rc = h->s.init())) return bail(rc);
    const int64_t n1 = std::max<int64_t>(n, 1);
    if (n > 0) {
        if (!data) return bail(fail(CLB_EARGUMENT, "data is null"));
        if (on_device) h->X = data;
        else if ((rc = upload(h->dX, data, sizeof(float) * dim * n, h->s.st))) return bail(rc);
    } else {      // a rank whose part of the clustering sample is empty: nothing to read from the host
        if ((rc = h->dX.alloc(sizeof(float) * dim))) return bail(rc);
        if (hipMemsetAsync(h->dX.p, 0, sizeof(float) * dim, h->s.st) != hipSuccess) return bail(fail(CLB_EHIP, "memset failed"));
    }
    if (!h->X) h->X = h->dX.as<float>();
    if ((rc = h->dC.alloc(sizeof(float) * dim * K)) || (rc = h->dNew.alloc(sizeof(float) * dim * K)) ||
        (rc = h->dC2.alloc(sizeof(float) * K)) || (rc = h->dAssign.alloc(sizeof(uint32_t) * n1)) ||
        (rc = h->dOrder.alloc(sizeof(uint32_t) * n1)) || (rc = h->dIota.alloc(sizeof(uint32_t) * n1)) ||
        (rc = h->dKeys.alloc(sizeof(uint32_t) * n1)) || (rc = h->dCounts.alloc(sizeof(uint32_t) * (K + 1))) ||
        (rc = h->dStart.alloc(sizeof(uint32_t) * (K + 2))) || (rc = h->dCnt32.alloc(sizeof(int) * K)) ||
        (rc = h->dCnt64.alloc(sizeof(long long) * K)) || (rc = h->dErr.alloc(sizeof(int))) ||
        (rc = h->dNext.alloc(sizeof(float) * dim * K)) || (rc = h->dDelta.alloc(sizeof(unsigned int))))
        return bail(rc);
    if (n > 0) hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n)), dim3(256), 0, h->s.st, h->dIota.as<uint32_t>(), n);
    while (((int64_t)1 << h->end_bit) <= K) ++h->end_bit;
    if (hipStreamSynchronize(h->s.st) != hipSuccess) return bail(fail(CLB_EHIP, "shard upload failed"));
    *out = h;
    return CLB_OK;
}

int clb_kmeans_shard_destroy(clb_kmeans_shard* h) {
    if (!h) return CLB_OK;
    (void)hipSetDevice(h->device);
    if (h->s.st) (void)hipStreamSynchronize(h->s.st);
    delete h;
    return CLB_OK;
}

// one pass over the shard's points with the centroids in h->dC: sums -> h->dNew, counts -> h->dCnt64 (enqueued on st)
static int kmeans_shard_pass_core(clb_kmeans_shard* h, hipStream_t st);

int clb_kmeans_shard_pass(clb_kmeans_shard* h, const float* centroids, float* sums, int64_t* counts,
                          int32_t* assignments) {
    if (!h || !centroids || !sums || !counts) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(h->device));
    hipStream_t st = h->s.st;
    const int64_t n = h->n, K = h->K, dim = h->dim;
    CLB_HIP(hipMemcpyAsync(h->dC.p, centroids, sizeof(float) * dim * K, hipMemcpyHostToDevice, st));
    CLB_TRY(kmeans_shard_pass_core(h, st));
    CLB_HIP(hipMemcpyAsync(sums, h->dNew.p, sizeof(float) * dim * K, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipMemcpyAsync(counts, h->dCnt64.p, sizeof(int64_t) * K, hipMemcpyDeviceToHost, st));
    if (assignments && n > 0)
        CLB_HIP(hipMemcpyAsync(assignments, h->dAssign.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    return CLB_OK;
}

static int kmeans_shard_pass_core(clb_kmeans_shard* h, hipStream_t st) {
    const int64_t n = h->n, K = h->K, dim = h->dim;
    hipLaunchKernelGGL(centroid_sumsq_kernel, dim3(blocks_for(K, 64)), dim3(64), 0, st, h->dC.as<float>(), (int)dim,
                       (int)K, h->dC2.as<float>());
    CLB_TRY(nearest_centroids<1>(st, h->dC.as<float>(), h->dC2.as<float>(), (int)dim, (int)K, h->X, n,
                                 h->dAssign.as<uint32_t>(), &h->nscratch));
    CLB_HIP(hipMemsetAsync(h->dCounts.p, 0, sizeof(uint32_t) * (K + 1), st));
    CLB_HIP(hipMemsetAsync(h->dErr.p, 0, sizeof(int), st));
    if (n > 0) {
        hipLaunchKernelGGL(code_histogram_kernel, dim3(blocks_for(n)), dim3(256), 0, st, h->dAssign.as<uint32_t>(), n,
                           (uint32_t)K, h->dCounts.as<unsigned int>(), h->dErr.as<int>());
        CLB_TRY(sort_pairs_u32(h->dAssign.as<uint32_t>(), h->dKeys.as<uint32_t>(), h->dIota.as<uint32_t>(),
                               h->dOrder.as<uint32_t>(), (size_t)n, h->end_bit, st, &h->sort_tmp));
    }
    CLB_TRY(exclusive_scan_u32(h->dCounts.as<uint32_t>(), h->dStart.as<uint32_t>(), (size_t)K, st, &h->scan_tmp));
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(blocks_for(K * dim)), dim3(256), 0, st, h->X,
                       (int)dim, h->dOrder.as<uint32_t>(), h->dStart.as<uint32_t>(), (int)K, (int)h->point_bsize,
                       h->dNew.as<float>(), h->dCnt32.as<int>());
    hipLaunchKernelGGL(widen_counts_kernel, dim3(blocks_for(K)), dim3(256), 0, st, h->dCnt32.as<int>(),
                       h->dCnt64.as<long long>(), (int)K);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

// ---- the same iteration with the exchange on the device (no host round trip of the 64-MB blocks) -----------------
// offset of the counts inside a block: the sums padded to a multiple of 8 bytes
static inline size_t kmeans_block_counts_offset(const clb_kmeans_shard* h) {
    return ((size_t)sizeof(float) * h->dim * h->K + 7) / 8 * 8;
}

int64_t clb_kmeans_shard_block_bytes(const clb_kmeans_shard* h) {
    return h ? (int64_t)(kmeans_block_counts_offset(h) + sizeof(int64_t) * h->K) : 0;
}

int clb_kmeans_shard_set_centroids(clb_kmeans_shard* h, const float* centroids) {
    if (!h || !centroids) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(h->device));
    // hipMemcpyDefault: `centroids` may be a host or a device pointer (unified addressing tells them apart)
    CLB_HIP(hipMemcpyAsync(h->dC.p, centroids, sizeof(float) * h->dim * h->K, hipMemcpyDefault, h->s.st));
    CLB_HIP(hipStreamSynchronize(h->s.st));
    return CLB_OK;
}

int clb_kmeans_shard_get_centroids(clb_kmeans_shard* h, float* centroids) {
    if (!h || !centroids) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(h->device));
    CLB_HIP(hipMemcpy(centroids, h->dC.p, sizeof(float) * h->dim * h->K, hipMemcpyDefault));
    return CLB_OK;
}

int clb_kmeans_shard_get_assignments(clb_kmeans_shard* h, int32_t* assignments) {
    if (!h || (!assignments && h->n > 0)) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(h->device));
    CLB_HIP(hipStreamSynchronize(h->s.st));
    if (h->n > 0) CLB_HIP(hipMemcpy(assignments, h->dAssign.p, sizeof(int32_t) * h->n, hipMemcpyDefault));
    return CLB_OK;
}

int clb_kmeans_shard_pass_device(clb_kmeans_shard* h, void* d_block, void* hip_stream) {
    if (!h || !d_block) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    // the pass runs on the shard's own stream (its scratch lives there); the caller's stream waits for it
    hipEvent_t ev = nullptr, ev2 = nullptr;
    CLB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CLB_HIP(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    int rc = CLB_OK;
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(h->s.st, ev, 0) != hipSuccess) rc = fail(CLB_EHIP, "event failed");
    if (!rc) rc = kmeans_shard_pass_core(h, h->s.st);
    if (!rc) {
        char* blk = static_cast<char*>(d_block);
        if (hipMemcpyAsync(blk, h->dNew.p, sizeof(float) * h->dim * h->K, hipMemcpyDeviceToDevice, h->s.st) != hipSuccess ||
            hipMemcpyAsync(blk + kmeans_block_counts_offset(h), h->dCnt64.p, sizeof(int64_t) * h->K, hipMemcpyDeviceToDevice, h->s.st) != hipSuccess ||
            hipEventRecord(ev2, h->s.st) != hipSuccess || hipStreamWaitEvent(st, ev2, 0) != hipSuccess)
            rc = fail(CLB_EHIP, "block copy failed");
    }
    (void)hipEventDestroy(ev); (void)hipEventDestroy(ev2);
    return rc;
}

int clb_kmeans_shard_update_device(clb_kmeans_shard* h, const void* d_gathered, int64_t world, float tol, float* delta_out,
                                   int* converged, void* hip_stream) {
    if (!h || !d_gathered) return fail(CLB_EARGUMENT, "null argument");
    if (world < 1) return fail(CLB_EDIMENSION, "world must be >= 1");
    CLB_TRY(use_device(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t K = h->K, dim = h->dim;
    const size_t blk = (size_t)clb_kmeans_shard_block_bytes(h);
    DevBuf& dDelta = h->dDelta;
    DevBuf& dNext = h->dNext;
    CLB_HIP(hipMemsetAsync(dDelta.p, 0, sizeof(unsigned int), st));
    const char* g = static_cast<const char*>(d_gathered);
    hipLaunchKernelGGL(kmeans_reduce_update_kernel, dim3(blocks_for(K * dim)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(g), reinterpret_cast<const long long*>(g + kmeans_block_counts_offset(h)),
                       (int)world, h->dC.as<float>(), (int)dim, (int)K, dNext.as<float>(), dDelta.as<unsigned int>(),
                       blk / sizeof(float), blk / sizeof(long long));
    CLB_HIP(hipGetLastError());
    unsigned int bits = 0;
    CLB_HIP(hipMemcpyAsync(&bits, dDelta.p, sizeof bits, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    float delta;
    memcpy(&delta, &bits, sizeof delta);
    if (delta_out) *delta_out = delta;
    const int conv = delta < tol;      // utils.jl:308-311: the previous centroids stay
    if (converged) *converged = conv;
    if (!conv) {   // the stream is idle (synchronised above): the new centroids become the handle's by a pointer swap
        std::swap(h->dC.p, dNext.p);
        std::swap(h->dC.bytes, dNext.bytes);
    }
    return CLB_OK;
}

int clb_kmeans_reduce_update(int device, float* centroids, const float* gathered_sums, const int64_t* gathered_counts,
                             int64_t world, int64_t dim, int64_t K, float tol, float* delta_out, int* converged) {
    if (!centroids || !gathered_sums || !gathered_counts) return fail(CLB_EARGUMENT, "null argument");
    if (world < 1 || K < 1 || dim < 1) return fail(CLB_EDIMENSION, "world, dim and K must be >= 1");
    CLB_TRY(use_device(device));
    Stream s; CLB_TRY(s.init());
    DevBuf dS, dCt, dOld, dNew, dDelta;
    CLB_TRY(upload(dS, gathered_sums, sizeof(float) * world * dim * K, s.st));
    CLB_TRY(upload(dCt, gathered_counts, sizeof(int64_t) * world * K, s.st));
    CLB_TRY(upload(dOld, centroids, sizeof(float) * dim * K, s.st));
    CLB_TRY(dNew.alloc(sizeof(float) * dim * K));
    CLB_TRY(dDelta.alloc(sizeof(unsigned int)));
    CLB_HIP(hipMemsetAsync(dDelta.p, 0, sizeof(unsigned int), s.st));
    hipLaunchKernelGGL(kmeans_reduce_update_kernel, dim3(blocks_for(K * dim)), dim3(256), 0, s.st, dS.as<float>(),
                       dCt.as<long long>(), (int)world, dOld.as<float>(), (int)dim, (int)K, dNew.as<float>(),
                       dDelta.as<unsigned int>(), (size_t)K * dim, (size_t)K);
    CLB_HIP(hipGetLastError());
    unsigned int bits = 0;
    CLB_HIP(hipMemcpyAsync(&bits, dDelta.p, sizeof bits, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    float delta;
    memcpy(&delta, &bits, sizeof delta);
    if (delta_out) *delta_out = delta;
    const int conv = delta < tol;      // utils.jl:308-311: the previous centroids stay
    if (converged) *converged = conv;
    if (!conv) {
        CLB_HIP(hipMemcpyAsync(centroids, dNew.p, sizeof(float) * dim * K, hipMemcpyDeviceToHost, s.st));
        CLB_HIP(hipStreamSynchronize(s.st));
    }
    return CLB_OK;
}

int clb_compute_avg_residuals(int device, int nbits, const float* centroids, int64_t dim, int64_t K,
                              const float* heldout, int64_t n, uint32_t* codes, int64_t n_codes,
                              float* bucket_cutoffs, float* bucket_weights, float* avg_residual) {
    if (n_codes != n) return fail(CLB_EDIMENSION, "length(codes) must be equal to the number of embeddings in heldout!");
    if (n < 1 || dim < 1) return fail(CLB_EARGUMENT, "empty heldout set");
    if (nbits < 1 || nbits > 16) return fail(CLB_EDOMAIN, "nbits out of range");
    CLB_TRY(use_device(device));
    Stream s; CLB_TRY(s.init());
    DevBuf dC, dX, dCodes, dRes, dSorted, dAbs;
    CLB_TRY(upload(dC, centroids, sizeof(float) * dim * K, s.st));
    CLB_TRY(upload(dX, heldout, sizeof(float) * dim * n, s.st));
    CLB_TRY(dCodes.alloc(sizeof(uint32_t) * n));
    CLB_TRY(dRes.alloc(sizeof(float) * dim * n));
    CLB_TRY(dSorted.alloc(sizeof(float) * dim * n));
    CLB_TRY(dAbs.alloc(sizeof(double)));
    CLB_HIP(hipMemsetAsync(dAbs.p, 0, sizeof(double), s.st));
    NearestScratch nscratch;
    CLB_TRY(nearest_centroids<0>(s.st, dC.as<float>(), nullptr, (int)dim, (int)K, dX.as<float>(), n, dCodes.as<uint32_t>(), &nscratch));
    hipLaunchKernelGGL(heldout_residual_kernel, dim3(blocks_for(n * dim)), dim3(256), 0, s.st, dC.as<float>(), (int)dim,
                       dX.as<float>(), dCodes.as<uint32_t>(), n, dRes.as<float>(), dAbs.as<double>());
    CLB_HIP(hipGetLastError());
    CLB_TRY(sort_keys_f32(dRes.as<float>(), dSorted.as<float>(), (size_t)(dim * n), s.st));
    double abs_sum = 0;
    CLB_HIP(hipMemcpy(&abs_sum, dAbs.p, sizeof(double), hipMemcpyDeviceToHost));
    CLB_HIP(hipMemcpy(codes, dCodes.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    *avg_residual = (float)(abs_sum / (double)(dim * n));
    const int64_t nopt = (int64_t)1 << nbits;
    for (int64_t q = 1; q < nopt; ++q)
        CLB_TRY(quantile7_device(dSorted.as<float>(), dim * n, (double)q / (double)nopt, &bucket_cutoffs[q - 1]));
    for (int64_t q = 0; q < nopt; ++q)
        CLB_TRY(quantile7_device(dSorted.as<float>(), dim * n, (double)q / (double)nopt + 0.5 / (double)nopt, &bucket_weights[q]));
    return CLB_OK;
}

// _build_ivf on device arrays: stable LSD radix sort of (code, embedding id) + histogram; d_ivf / d_lens are device
// pointers (Int64, 1-based ids).  Synchronises `st` (the code range check must be read back: counts(values, K) throws).
static int build_ivf_core(hipStream_t st, const uint32_t* d_codes, int64_t n, int64_t K, int64_t* d_ivf, int64_t* d_lens) {
    DevBuf dKeys, dIota, dOrder, dCounts, dErr;
    CLB_TRY(dKeys.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dIota.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dOrder.alloc(sizeof(uint32_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dCounts.alloc(sizeof(uint32_t) * std::max<int64_t>(K, 1)));
    CLB_TRY(dErr.alloc(sizeof(int)));
    CLB_HIP(hipMemsetAsync(dCounts.p, 0, dCounts.bytes, st));
    CLB_HIP(hipMemsetAsync(dErr.p, 0, sizeof(int), st));
    if (n > 0) {
        hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n)), dim3(256), 0, st, dIota.as<uint32_t>(), n);
        hipLaunchKernelGGL(code_histogram_kernel, dim3(blocks_for(n)), dim3(256), 0, st, d_codes, n, (uint32_t)K,
                           dCounts.as<unsigned int>(), dErr.as<int>());
        int herr = 0;
        CLB_HIP(hipMemcpyAsync(&herr, dErr.p, sizeof(int), hipMemcpyDeviceToHost, st));
        CLB_HIP(hipStreamSynchronize(st));
        if (herr) return fail(CLB_EBOUNDS, "codes outside 1..num_partitions");  // counts(values, K) would throw
        int end_bit = 1;
        while (((int64_t)1 << end_bit) <= K) ++end_bit;
        CLB_TRY(sort_pairs_u32(d_codes, dKeys.as<uint32_t>(), dIota.as<uint32_t>(), dOrder.as<uint32_t>(), (size_t)n,
                               end_bit, st));
        hipLaunchKernelGGL(ivf_widen_kernel, dim3(blocks_for(n)), dim3(256), 0, st, dOrder.as<uint32_t>(), n, d_ivf);
    }
    if (K > 0)
        hipLaunchKernelGGL(widen_u32_i64_kernel, dim3(blocks_for(K)), dim3(256), 0, st, dCounts.as<uint32_t>(), K, d_lens);
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipStreamSynchronize(st));      // the scratch above is freed on return
    return CLB_OK;
}

int clb_build_ivf(int device, const uint32_t* codes, int64_t n, int64_t K, int64_t* ivf, int64_t* ivf_lengths) {
    if (K < 0 || n < 0) return fail(CLB_EARGUMENT, "negative size");
    if (n >= (int64_t)0xffffffffll) return fail(CLB_EUNSUPPORTED, "n too large");
    CLB_TRY(use_device(device));
    Stream s; CLB_TRY(s.init());
    DevBuf dCodes, dIvf, dLens;
    CLB_TRY(upload(dCodes, codes, sizeof(uint32_t) * std::max<int64_t>(n, 1), s.st));
    CLB_TRY(dIvf.alloc(sizeof(int64_t) * std::max<int64_t>(n, 1)));
    CLB_TRY(dLens.alloc(sizeof(int64_t) * std::max<int64_t>(K, 1)));
    CLB_TRY(build_ivf_core(s.st, dCodes.as<uint32_t>(), n, K, dIvf.as<int64_t>(), dLens.as<int64_t>()));
    if (n > 0) CLB_HIP(hipMemcpyAsync(ivf, dIvf.p, sizeof(int64_t) * n, hipMemcpyDeviceToHost, s.st));
    if (K > 0) CLB_HIP(hipMemcpyAsync(ivf_lengths, dLens.p, sizeof(int64_t) * K, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

// Test hooks of the hand-written sort and scan (csrc/sort.hip): host arrays in, host arrays out.  key_bits = 32: uint32 keys
// sorted on their low `end_bit` bits; 64: uint64 keys on all bits; -32: float keys (ascending, -0.0 < +0.0).  vals may be null.
int clb_debug_sort(int device, int key_bits, const void* keys, const uint32_t* vals, int64_t n, int end_bit, void* keys_out,
                   uint32_t* vals_out) {
    if (n < 0 || (n > 0 && (!keys || !keys_out)) || (vals && !vals_out)) return fail(CLB_EARGUMENT, "clb_debug_sort: null argument or negative n");
    if (key_bits != 32 && key_bits != 64 && key_bits != -32) return fail(CLB_EARGUMENT, "clb_debug_sort: key_bits must be 32, 64 or -32 (float)");
    CLB_TRY(use_device(device));
    if (n == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    const size_t kb = key_bits == 64 ? 8 : 4;
    DevBuf dk, dko, dv, dvo;
    CLB_TRY(upload(dk, keys, kb * n, s.st));
    CLB_TRY(dko.alloc(kb * n));
    if (vals) { CLB_TRY(upload(dv, vals, 4 * (size_t)n, s.st)); CLB_TRY(dvo.alloc(4 * (size_t)n)); }
    if (key_bits == -32) {
        if (vals) return fail(CLB_EUNSUPPORTED, "clb_debug_sort: float keys sort without values");
        CLB_TRY(sort_keys_f32(dk.as<float>(), dko.as<float>(), (size_t)n, s.st));
    } else if (key_bits == 32) {
        if (!vals) return fail(CLB_EUNSUPPORTED, "clb_debug_sort: uint32 keys sort as pairs");
        CLB_TRY(sort_pairs_u32(dk.as<uint32_t>(), dko.as<uint32_t>(), dv.as<uint32_t>(), dvo.as<uint32_t>(), (size_t)n, end_bit, s.st));
    } else if (vals) {
        CLB_TRY(sort_pairs_u64(dk.as<uint64_t>(), dko.as<uint64_t>(), dv.as<uint32_t>(), dvo.as<uint32_t>(), (size_t)n, s.st));
    } else {
        CLB_TRY(sort_keys_u64(dk.as<uint64_t>(), dko.as<uint64_t>(), (size_t)n, s.st));
    }
    CLB_HIP(hipMemcpyAsync(keys_out, dko.p, kb * n, hipMemcpyDeviceToHost, s.st));
    if (vals) CLB_HIP(hipMemcpyAsync(vals_out, dvo.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}
// out[0 .. n] = exclusive prefix sums of in[0 .. n) (out[n] = the total, modulo 2^32)
int clb_debug_exclusive_scan(int device, const uint32_t* in, int64_t n, uint32_t* out) {
    if (n < 0 || !out || (n > 0 && !in)) return fail(CLB_EARGUMENT, "clb_debug_exclusive_scan: null argument or negative n");
    CLB_TRY(use_device(device));
    Stream s; CLB_TRY(s.init());
    DevBuf di, dout;
    CLB_TRY(di.alloc(4 * (size_t)(n + 1)));
    CLB_HIP(hipMemsetAsync(di.p, 0, 4 * (size_t)(n + 1), s.st));
    if (n > 0) CLB_HIP(hipMemcpyAsync(di.p, in, 4 * (size_t)n, hipMemcpyHostToDevice, s.st));
    CLB_TRY(dout.alloc(4 * (size_t)(n + 1)));
    CLB_TRY(exclusive_scan_u32(di.as<uint32_t>(), dout.as<uint32_t>(), (size_t)n, s.st));
    CLB_HIP(hipMemcpyAsync(out, dout.p, 4 * (size_t)(n + 1), hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

int clb_build_ivf_device(int device, const uint32_t* d_codes, int64_t n, int64_t K, int64_t* d_ivf,
                         int64_t* d_ivf_lengths, void* hip_stream) {
    if (K < 0 || n < 0) return fail(CLB_EARGUMENT, "negative size");
    if (n >= (int64_t)0xffffffffll) return fail(CLB_EUNSUPPORTED, "n too large");
    if ((n > 0 && (!d_codes || !d_ivf)) || (K > 0 && !d_ivf_lengths)) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(device));
    return build_ivf_core((hipStream_t)hip_stream, d_codes, n, K, d_ivf, d_ivf_lengths);
}

// ---- plain device arrays for hosts without their own (include/colbert_hip.h) ------------------------------------------
int clb_device_malloc(int device, int64_t bytes, void** d_out) {
    if (!d_out || bytes < 0) return fail(CLB_EARGUMENT, "clb_device_malloc: null output or negative size");
    *d_out = nullptr;
    CLB_TRY(use_device(device));
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) return fail(CLB_ENOMEM, "hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
    *d_out = p;
    return CLB_OK;
}

int clb_device_free(int device, void* d_ptr) {
    if (!d_ptr) return CLB_OK;
    CLB_TRY(use_device(device));
    CLB_HIP(hipFree(d_ptr));
    return CLB_OK;
}

int clb_device_upload(int device, void* d_dst, const void* src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!d_dst || !src))) return fail(CLB_EARGUMENT, "clb_device_upload: null pointer or negative size");
    CLB_TRY(use_device(device));
    if (bytes) CLB_HIP(hipMemcpy(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return CLB_OK;
}

int clb_device_download(int device, void* dst, const void* d_src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !d_src))) return fail(CLB_EARGUMENT, "clb_device_download: null pointer or negative size");
    CLB_TRY(use_device(device));
    if (bytes) CLB_HIP(hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return CLB_OK;
}

int clb_device_synchronize(int device) {
    CLB_TRY(use_device(device));
    CLB_HIP(hipDeviceSynchronize());
    return CLB_OK;
}

int clb_device_memory(int device, int64_t* free_bytes, int64_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(device));
    size_t f = 0, t = 0;
    CLB_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f; *total_bytes = (int64_t)t;
    return CLB_OK;
}

int clb_gather_rows_device(int device, const void* d_src, int64_t n_src, int64_t row_bytes, const int64_t* d_rows, int64_t n,
                           void* d_dst, void* hip_stream) {
    if (n < 0 || n_src < 0 || row_bytes < 4 || row_bytes % 4) return fail(CLB_EARGUMENT, "gather: sizes must be >= 0 and row_bytes a positive multiple of 4");
    if (n == 0) return CLB_OK;
    if (!d_src || !d_rows || !d_dst) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(device));
    hipStream_t st = (hipStream_t)hip_stream;
    DevBuf dErr;
    CLB_TRY(dErr.alloc(sizeof(int)));
    CLB_HIP(hipMemsetAsync(dErr.p, 0, sizeof(int), st));
    const bool wide = row_bytes % 16 == 0 && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0);
    const int64_t per_row = wide ? row_bytes / 16 : row_bytes / 4;            // 16-byte (else 4-byte) pieces per row
    const int64_t total = n * per_row;
    if (wide)
        hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(blocks_for(total)), dim3(256), 0, st, (const uint4*)d_src, n_src, per_row,
                           d_rows, n, (uint4*)d_dst, dErr.as<int>());
    else
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(blocks_for(total)), dim3(256), 0, st, (const uint32_t*)d_src, n_src, per_row,
                           d_rows, n, (uint32_t*)d_dst, dErr.as<int>());
    CLB_HIP(hipGetLastError());
    int herr = 0;
    CLB_HIP(hipMemcpyAsync(&herr, dErr.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    if (herr) return fail(CLB_EBOUNDS, "gather: a row index lies outside 0..%lld", (long long)(n_src - 1));
    return CLB_OK;
}

int clb_doc_epilogue(int device, const float* D, int64_t dim, int64_t L, int64_t N, const int32_t* integer_ids,
                     const int64_t* skiplist, int64_t n_skip, float* out, int64_t* doclens, int64_t* n_out) {
    CLB_TRY(use_device(device));
    *n_out = 0;
    if (L * N == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    DevBuf dD, dIds, dSkip, dMask, dLens, dStart, dOut;
    CLB_TRY(upload(dD, D, sizeof(float) * dim * L * N, s.st));
    CLB_TRY(upload(dIds, integer_ids, sizeof(int32_t) * L * N, s.st));
    CLB_TRY(upload(dSkip, skiplist, sizeof(int64_t) * std::max<int64_t>(n_skip, 1), s.st));
    CLB_TRY(dMask.alloc((size_t)L * N));
    CLB_TRY(dLens.alloc(sizeof(int64_t) * N));
    hipLaunchKernelGGL(epilogue_mask_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, s.st, dIds.as<int32_t>(), (int)L,
                       (int)N, dSkip.as<int64_t>(), (int)n_skip, dMask.as<uint8_t>(), dLens.as<int64_t>());
    CLB_HIP(hipMemcpyAsync(doclens, dLens.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    std::vector<int64_t> start((size_t)N);
    int64_t run = 0;
    for (int64_t i = 0; i < N; ++i) { start[i] = run; run += doclens[i]; }
    *n_out = run;
    if (run == 0) return CLB_OK;
    CLB_TRY(upload(dStart, start.data(), sizeof(int64_t) * N, s.st));
    CLB_TRY(dOut.alloc(sizeof(float) * dim * run));
    hipLaunchKernelGGL(epilogue_normalize_kernel, dim3(blocks_for(L * N, 64)), dim3(64), 0, s.st, dD.as<float>(), (int)dim,
                       (int)L, (int)N, dMask.as<uint8_t>(), dStart.as<int64_t>(), dOut.as<float>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(out, dOut.p, sizeof(float) * dim * run, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

int clb_query_epilogue(int device, float* Q, int64_t dim, int64_t L, int64_t N, const int32_t* integer_ids,
                       const int64_t* skiplist, int64_t n_skip) {
    CLB_TRY(use_device(device));
    if (L * N == 0) return CLB_OK;
    Stream s; CLB_TRY(s.init());
    DevBuf dQ, dIds, dSkip, dMask, dLens, dOut;
    CLB_TRY(upload(dQ, Q, sizeof(float) * dim * L * N, s.st));
    CLB_TRY(upload(dIds, integer_ids, sizeof(int32_t) * L * N, s.st));
    CLB_TRY(upload(dSkip, skiplist, sizeof(int64_t) * std::max<int64_t>(n_skip, 1), s.st));
    CLB_TRY(dMask.alloc((size_t)L * N));
    CLB_TRY(dLens.alloc(sizeof(int64_t) * N));
    CLB_TRY(dOut.alloc(sizeof(float) * dim * L * N));
    hipLaunchKernelGGL(epilogue_mask_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, s.st, dIds.as<int32_t>(), (int)L,
                       (int)N, dSkip.as<int64_t>(), (int)n_skip, dMask.as<uint8_t>(), dLens.as<int64_t>());
    hipLaunchKernelGGL(epilogue_normalize_kernel, dim3(blocks_for(L * N, 64)), dim3(64), 0, s.st, dQ.as<float>(), (int)dim,
                       (int)L, (int)N, dMask.as<uint8_t>(), (const int64_t*)nullptr, dOut.as<float>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(Q, dOut.p, sizeof(float) * dim * L * N, hipMemcpyDeviceToHost, s.st));
    CLB_HIP(hipStreamSynchronize(s.st));
    return CLB_OK;
}

}  // extern "C"
