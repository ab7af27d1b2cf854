// sort.hip -- see sort.hpp.  Hand-written for gfx950 (round 6; rocPRIM did this until round 5 -- its radix sort alone was
// 9 MB of the 14-MB library: every tuning of every architecture it knows, compiled for this one).
//
// Stable LSD radix sort, 8-bit digits, ceil(end_bit / 8) passes.  One pass:
//   radix_hist_kernel     work-group g counts the digits of ITS tile (256 threads x 32 items = 8 192 consecutive elements) into
//                         counts[digit][g] -- digit-major, so that ONE exclusive scan of the flattened array yields, for every
//                         (digit, tile), where that tile's elements of that digit start in the output: all smaller digits
//                         first, and within a digit the tiles in input order (stability across tiles);
//   scan                  exclusive_scan_u32 below, in place;
//   radix_scatter_kernel  the tile again; wave w owns the 2 048 consecutive elements w of the tile and keeps their keys in
//                         registers (lane l, item i = element 64 i + l: coalesced loads, and chunk i = the 64 elements of item i
//                         precedes chunk i + 1).  Per-wave digit counts (LDS atomics) + the tile's scanned bases give every wave
//                         its own running cursor per digit; then chunk by chunk: the lanes holding the same digit find each
//                         other with 8 ballots, rank = number of such lanes below, position = cursor[digit] + rank, and the
//                         lowest of them advances the cursor.  A wave's LDS operations execute in order, so all lanes read
//                         the cursor before that store: no barrier inside the loop, and input order is kept inside the tile.
// Keys and values ping-pong between the output arrays and a temporary pair in `scratch` such that the last pass lands in the
// output; the input is never written.  float keys sort as their order-preserving unsigned images.
#include <algorithm>
#include <cstring>

#include "common.hpp"
#include "sort.hpp"

namespace clb {

namespace {

constexpr int kSortItems = 32;                      // keys per lane
constexpr int kSortWaveTile = 64 * kSortItems;      // 2 048 consecutive elements per wave
constexpr int kSortTile = 4 * kSortWaveTile;        // 8 192 per work-group

template <class K>
__device__ __forceinline__ uint32_t digit_of(K key, int shift) { return (uint32_t)(key >> shift) & 255u; }

template <class K>
static __global__ __launch_bounds__(256) void radix_hist_kernel(const K* __restrict__ keys, size_t n, int shift,
                                                               uint32_t* __restrict__ counts, uint32_t ntiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSortTile;
#pragma unroll 8
    for (int i = 0; i < kSortTile / 256; ++i) {
        const size_t e = base + (size_t)i * 256 + threadIdx.x;
        if (e < n) atomicAdd(&h[digit_of(keys[e], shift)], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <class K, bool VALS>
static __global__ __launch_bounds__(256) void radix_scatter_kernel(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                                  const uint32_t* __restrict__ vals_in,
                                                                  uint32_t* __restrict__ vals_out, size_t n, int shift,
                                                                  const uint32_t* __restrict__ bases, uint32_t ntiles) {
    __shared__ uint32_t cur[4][256];                // per wave: count, then running cursor, of every digit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += 256) (&cur[0][0])[i] = 0u;
    __syncthreads();
    const size_t wbase = (size_t)blockIdx.x * kSortTile + (size_t)wave * kSortWaveTile;
    K key[kSortItems];
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        const size_t e = wbase + (size_t)i * 64 + lane;
        key[i] = e < n ? keys_in[e] : (K)0;
        if (e < n) atomicAdd(&cur[wave][digit_of(key[i], shift)], 1u);
    }
    __syncthreads();
    {   // digit d = threadIdx.x: the tile's base, then the waves in order
        uint32_t run = bases[(size_t)threadIdx.x * ntiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = cur[w][threadIdx.x];
            cur[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
    uint32_t* mycur = cur[wave];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll      // (fully: key[] must stay in registers)
    for (int i = 0; i < kSortItems; ++i) {
        const size_t e = wbase + (size_t)i * 64 + lane;
        const bool valid = e < n;
        const uint32_t d = digit_of(key[i], shift);
        unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below);
        const uint32_t start = mycur[d];
        __builtin_amdgcn_wave_barrier();            // every lane has read the cursor (the LDS executes a wave's operations in order)
        if (valid && rank == 0) mycur[d] = start + (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const size_t pos = (size_t)start + rank;
            keys_out[pos] = key[i];
            if (VALS) vals_out[pos] = vals_in[e];
        }
    }
}

// ---- exclusive scan of m uint32 (sums wrap modulo 2^32 like any uint32 sum) -----------------------------------------------
constexpr int kScanItems = 16;
constexpr int kScanTile = 1024 * kScanItems;        // 16 384 elements per work-group

// MODE 0: scan the tile, add offs[blockIdx.x] (nullptr: 0) -> out;  MODE 1: the tile's sum -> sums[blockIdx.x]
template <int MODE>
static __global__ __launch_bounds__(1024) void scan_tile_kernel(const uint32_t* in, uint32_t* out,   // (may alias: in place)
                                                               size_t m, const uint32_t* __restrict__ offs,
                                                               uint32_t* __restrict__ sums) {
    __shared__ uint32_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t tot = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < m ? in[base + i] : 0u;
        tot += v[i];
    }
    uint32_t x = tot;                                 // inclusive scan of the lanes' totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t s = wsum[w];
        wbase += w < wave ? s : 0u;
        all += s;
    }
    if (MODE == 1) {
        if (threadIdx.x == 0) sums[blockIdx.x] = all;
        return;
    }
    uint32_t run = wbase + x - tot + (offs ? offs[blockIdx.x] : 0u);
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < m) out[base + i] = run;
        run += v[i];
    }
}

// in-place capable (in == out): a tile reads all its inputs before it writes, and tiles are disjoint
int scan_u32_device(const uint32_t* in, uint32_t* out, size_t m, hipStream_t st, uint32_t* sums /* >= tiles + tiles/16384 + 2 */) {
    if (m == 0) return CLB_OK;
    const size_t tiles = (m + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        hipLaunchKernelGGL(scan_tile_kernel<0>, dim3(1), dim3(1024), 0, st, in, out, m, (const uint32_t*)nullptr, (uint32_t*)nullptr);
        CLB_HIP(hipGetLastError());
        return CLB_OK;
    }
    if (tiles > (size_t)kScanTile * kScanTile) return fail(CLB_EUNSUPPORTED, "scan of %zu elements: more than three levels", m);
    hipLaunchKernelGGL(scan_tile_kernel<1>, dim3((unsigned)tiles), dim3(1024), 0, st, in, (uint32_t*)nullptr, m,
                       (const uint32_t*)nullptr, sums);
    CLB_TRY(scan_u32_device(sums, sums, tiles, st, sums + tiles));          // the tile sums, scanned in place
    hipLaunchKernelGGL(scan_tile_kernel<0>, dim3((unsigned)tiles), dim3(1024), 0, st, in, out, m, (const uint32_t*)sums, (uint32_t*)nullptr);
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}
inline size_t scan_scratch_words(size_t m) {
    const size_t tiles = (m + kScanTile - 1) / kScanTile;
    return tiles <= 1 ? 2 : tiles + scan_scratch_words(tiles) + 2;
}

// float <-> order-preserving unsigned image, in place
static __global__ void f32_to_keys_kernel(const float* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f32_order_key(in[i]);
}
static __global__ void keys_to_f32_kernel(uint32_t* __restrict__ io, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) io[i] = __float_as_uint(f32_from_order_key(io[i]));
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// keys_in is only read when `first_from_in`; otherwise the data already sits in keys_out (the float path converts there)
template <class K, bool VALS>
int radix_sort(const K* keys_in, K* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit,
               hipStream_t st, DevBuf* scratch, bool first_from_in = true) {
    if (n == 0) return CLB_OK;
    if (n >= ((size_t)1 << 32)) return fail(CLB_EUNSUPPORTED, "radix sort of %zu elements: positions are 32-bit", n);
    const int passes = std::max(1, (end_bit + 7) / 8);
    const size_t ntiles = (n + kSortTile - 1) / kSortTile;
    const size_t ncount = 256 * ntiles;
    const size_t b_counts = align256(sizeof(uint32_t) * (ncount + scan_scratch_words(ncount) + 4));
    const size_t b_keys = align256(sizeof(K) * n), b_vals = VALS ? align256(sizeof(uint32_t) * n) : 0;
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(b_counts + b_keys + b_vals));
    uint32_t* counts = tmp.as<uint32_t>();
    K* tkeys = reinterpret_cast<K*>(static_cast<char*>(tmp.p) + b_counts);
    uint32_t* tvals = VALS ? reinterpret_cast<uint32_t*>(static_cast<char*>(tmp.p) + b_counts + b_keys) : nullptr;
    // the last pass writes the output arrays; the passes before it alternate backwards from there
    const K* src_k = first_from_in ? keys_in : keys_out;
    const uint32_t* src_v = vals_in;
    bool to_out = (passes % 2) == 1;
    if (!first_from_in && to_out) {       // the data sits in keys_out and the first pass would write there: start from a copy
        CLB_HIP(hipMemcpyAsync(tkeys, keys_out, sizeof(K) * n, hipMemcpyDeviceToDevice, st));
        src_k = tkeys;
    }
    K* last_k = nullptr;
    for (int p = 0; p < passes; ++p) {
        K* dst_k = to_out ? keys_out : tkeys;
        uint32_t* dst_v = to_out ? vals_out : tvals;
        const int shift = 8 * p;
        hipLaunchKernelGGL(radix_hist_kernel<K>, dim3((unsigned)ntiles), dim3(256), 0, st, src_k, n, shift, counts, (uint32_t)ntiles);
        CLB_TRY(scan_u32_device(counts, counts, ncount, st, counts + ncount + 2));
        hipLaunchKernelGGL((radix_scatter_kernel<K, VALS>), dim3((unsigned)ntiles), dim3(256), 0, st, src_k, dst_k, src_v, dst_v, n,
                           shift, (const uint32_t*)counts, (uint32_t)ntiles);
        src_k = dst_k;
        src_v = dst_v;
        last_k = dst_k;
        to_out = !to_out;
    }
    if (last_k != keys_out) return fail(CLB_EARGUMENT, "radix sort: pass parity");      // cannot happen (see above)
    CLB_HIP(hipGetLastError());
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

}  // namespace

int sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                   size_t n, int end_bit, hipStream_t st, DevBuf* scratch) {
    return radix_sort<uint32_t, true>(keys_in, keys_out, vals_in, vals_out, n, end_bit, st, scratch);
}

int sort_keys_f32(const float* keys_in, float* keys_out, size_t n, hipStream_t st, DevBuf* scratch) {
    if (n == 0) return CLB_OK;
    uint32_t* ko = reinterpret_cast<uint32_t*>(keys_out);
    hipLaunchKernelGGL(f32_to_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys_in, ko, n);
    DevBuf local;
    DevBuf* tmp = scratch ? scratch : &local;
    CLB_TRY((radix_sort<uint32_t, false>(nullptr, ko, nullptr, nullptr, n, 32, st, tmp, /*first_from_in=*/false)));
    hipLaunchKernelGGL(keys_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ko, n);
    CLB_HIP(hipGetLastError());
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));
    return CLB_OK;
}

int sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                   hipStream_t st, DevBuf* scratch) {
    return radix_sort<uint64_t, true>(keys_in, keys_out, vals_in, vals_out, n, 64, st, scratch);
}

int sort_keys_u64(const uint64_t* keys_in, uint64_t* keys_out, size_t n, hipStream_t st, DevBuf* scratch) {
    return radix_sort<uint64_t, false>(keys_in, keys_out, nullptr, nullptr, n, 64, st, scratch);
}

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, hipStream_t st, DevBuf* scratch) {
    // scan n+1 inputs (the caller pads in[n] = 0) so that out[n] = total
    DevBuf local;
    DevBuf& tmp = scratch ? *scratch : local;
    CLB_TRY(tmp.ensure(sizeof(uint32_t) * (scan_scratch_words(n + 1) + 4)));
    CLB_TRY(scan_u32_device(in, out, n + 1, st, tmp.as<uint32_t>()));
    if (!scratch) CLB_HIP(hipStreamSynchronize(st));      // `local` is freed on return
    return CLB_OK;
}

}  // namespace clb
