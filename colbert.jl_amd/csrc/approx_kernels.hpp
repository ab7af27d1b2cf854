// approx_kernels.hpp -- two-pass mode: bf16-MFMA approximate MaxSim with a proven error bound, used to
// select a superset of the top-k that the exact kernel then re-scores.  (placeholder: not built yet)
#pragma once
#include "common.hpp"

namespace clb {

inline bool approx_supported(int /*dim*/, int /*nbits*/) { return false; }
inline size_t approx_cells_bytes(int64_t, int64_t, int64_t) { return 16; }
inline int build_inv_norms(hipStream_t, const float*, const float*, const uint32_t*, const uint8_t*, int64_t,
                           float*) {
    return CLB_OK;
}

}  // namespace clb
