// sort.hpp -- device radix sorts used by the index build (the only library primitive in this build:
// rocPRIM's stable LSD radix sort; everything else is hand-written).  Implemented in sort.hip so that
// the rocPRIM headers are compiled once.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace clb {
struct DevBuf;
// Every function takes an optional `scratch` buffer for rocPRIM's temporary storage: with one (grown on demand, kept by
// the caller across calls) the sort is only ENQUEUED on `st` -- no allocation, no host synchronisation; without one the
// temporary is allocated and freed inside the call, which therefore waits for the stream.
// stable sort of (key, value) pairs by key, ascending: values keep their input order among equal keys
// -- exactly Julia's sortperm(codes) when values = 0..n-1 (collection_indexer.jl:350)
int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                   size_t n, int end_bit, hipStream_t st, DevBuf* scratch = nullptr);
// ascending sort of float keys (quantiles of the pooled residuals, collection_indexer.jl:147-150)
int sort_keys_f32(const float* keys_in, float* keys_out, size_t n, hipStream_t st, DevBuf* scratch = nullptr);
// stable ascending sorts on 64-bit keys (general-shape search path: top-nprobe per token for nprobe > 32, top-k for
// k above the single-work-group sort)
int sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                   hipStream_t st, DevBuf* scratch = nullptr);
int sort_keys_u64(const uint64_t* keys_in, uint64_t* keys_out, size_t n, hipStream_t st, DevBuf* scratch = nullptr);
// exclusive prefix sum of n uint32 counts (out has n+1 entries, out[n] = total)
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, hipStream_t st, DevBuf* scratch = nullptr);
}  // namespace clb
