// sort.hip -- see sort.hpp
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.hpp"
#include "sort.hpp"

namespace clb {

int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                   size_t n, int end_bit, hipStream_t st, DevBuf* scratch) {
    if (n == 0) return CLB_OK;
    size_t tmp_bytes = 0;
    CLB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st));
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(tmp_bytes));
    CLB_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st));
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

int sort_keys_f32(const float* keys_in, float* keys_out, size_t n, hipStream_t st, DevBuf* scratch) {
    if (n == 0) return CLB_OK;
    size_t tmp_bytes = 0;
    CLB_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys_in, keys_out, n, 0, 32, st));
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(tmp_bytes));
    CLB_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, keys_in, keys_out, n, 0, 32, st));
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

int sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                   hipStream_t st, DevBuf* scratch) {
    if (n == 0) return CLB_OK;
    size_t tmp_bytes = 0;
    CLB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, st));
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(tmp_bytes));
    CLB_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, st));
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

int sort_keys_u64(const uint64_t* keys_in, uint64_t* keys_out, size_t n, hipStream_t st, DevBuf* scratch) {
    if (n == 0) return CLB_OK;
    size_t tmp_bytes = 0;
    CLB_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys_in, keys_out, n, 0, 64, st));
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(tmp_bytes));
    CLB_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, keys_in, keys_out, n, 0, 64, st));
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, hipStream_t st, DevBuf* scratch) {
    // scan n+1 inputs (the caller pads in[n] = 0) so that out[n] = total
    size_t tmp_bytes = 0;
    CLB_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st));
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(tmp_bytes));
    CLB_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st));
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

}  // namespace clb
