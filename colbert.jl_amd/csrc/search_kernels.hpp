// search_kernels.hpp -- HIP kernels (gfx950) for ColBERT.jl's search path, exact variant.
//
// Reference stages:  S1 centroid scores (ranking.jl:27)  S2 top-nprobe (utils.jl:327-332,
// ranking.jl:31-32)  S3 IVF union -> candidate pids (ranking.jl:7-21,35-43)  S4 gather (ranking.jl:46-67)
// S5 decompress (residual.jl:698-784)  S6 maxsim (ranking.jl:69-86)  S7 sortperm + first k
// (searching.jl:125-127).  S4-S6 are one fused kernel: nothing decompressed is ever written to HBM.
//
// Arithmetic follows the canonical order stated in oracle/colbert_oracle.h bit-for-bit:
//   dot products  = d-ascending fmaf chains: gfx950's f32 MFMA (v_mfma_f32_32x32x2_f32 /
//                   v_mfma_f32_16x16x4_f32) is exactly that chain when k is walked in order;
//   sum of squares= 4 interleaved partial sums (d mod 4), products rounded before the add;
//   x / (sqrtf(n2) + eps) with IEEE sqrt and divide (-fhip-fp32-correctly-rounded-divide-sqrt);
//   maxsim        = per-token max over the passage, sequential sum over tokens.
// Built with -ffp-contract=off so that no other multiply-add is fused behind our back.
#pragma once
#include <hip/hip_fp16.h>

#include "common.hpp"

namespace clb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kDim = 128;             // embedding dimension of the HIP search path
constexpr int kCentTileStride = 132;  // dwords per staged centroid row: 16-B aligned, b128-conflict-free
constexpr int kMaxTopK = 16384;       // k limit of the single-workgroup final sort (8 B per slot in LDS: 128 KB)

// -------------------------------------------------------------------------------------------------
// S1  cells[b][c][t] = dot(Q[b][:,t], C[:,c])            (ranking.jl:27  `Q' * centroids`)
// One wave owns a 32-centroid x 32-token tile: v_mfma_f32_32x32x2_f32, 64 chained steps (k = d).
// Centroid rows are staged through a wave-private LDS tile in full 512-B rows (coalesced dwordx4),
// read back conflict-free with ds_read_b128 (row stride 132 dwords).  MFMA-bound: 2*32*128 flop per
// centroid against 512 B read.
// grid = (workgroups over centroid tiles, B * TT), block = 128 (2 waves), LDS = 2 * 32 * 132 * 4.
// -------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(128) void centroid_scores_kernel(const float* __restrict__ C,
                                                               const float* __restrict__ Q,
                                                               float* __restrict__ cells, int K, int T,
                                                               int TT, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y / TT, tt = blockIdx.y % TT;
    const int Tpad = TT * 32;
    float* my = lds + wave * (32 * kCentTileStride);

    // B operand: lane (token i, half h) holds Q[t][2s+h], s = 0..63
    float qf[64];
    {
        const int t = tt * 32 + i;
        const float* qrow = Q + ((size_t)b * T + (t < T ? t : T - 1)) * kDim;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float4 v = *reinterpret_cast<const float4*>(qrow + 4 * m);
            if (t >= T) v = make_float4(0.f, 0.f, 0.f, 0.f);
            qf[2 * m] = h ? v.y : v.x;
            qf[2 * m + 1] = h ? v.w : v.z;
        }
    }
    const int waves_total = gridDim.x * 2;
    for (int tile = blockIdx.x * 2 + wave; tile < n_tiles; tile += waves_total) {
        const int c0 = tile * 32;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int row = 2 * m + h;
            int c = c0 + row;
            c = c < K ? c : K - 1;
            float4 v = *reinterpret_cast<const float4*>(C + (size_t)c * kDim + 4 * i);
            *reinterpret_cast<float4*>(my + row * kCentTileStride + 4 * i) = v;
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float4 a4 = *reinterpret_cast<const float4*>(my + i * kCentTileStride + 4 * m);
            const float a0 = h ? a4.y : a4.x;
            const float a1 = h ? a4.w : a4.z;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qf[2 * m], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qf[2 * m + 1], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int c = c0 + row;
            if (c < K) cells[((size_t)b * K + c) * Tpad + tt * 32 + i] = acc[r];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// S2  top-nprobe centroids per query token, ordered by (score desc, index asc) -- what
// mapslices(partialsortperm(v, 1:nprobe, rev=true)) returns (utils.jl:327-332).
// Stage a: each workgroup scans a slice of centroids for all tokens (coalesced rows of Tpad floats)
// and writes one partial list per token.  Stage b merges the partial lists.
// -------------------------------------------------------------------------------------------------
struct ValIdx {
    float v;
    int i;
};
__device__ __forceinline__ bool better(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

template <int NP>
__device__ __forceinline__ void topn_insert(float (&bv)[NP], int (&bi)[NP], float v, int idx) {
    if (!better(v, idx, bv[NP - 1], bi[NP - 1])) return;
    bv[NP - 1] = v;
    bi[NP - 1] = idx;
#pragma unroll
    for (int p = NP - 1; p > 0; --p) {
        if (better(bv[p], bi[p], bv[p - 1], bi[p - 1])) {
            float tv = bv[p]; bv[p] = bv[p - 1]; bv[p - 1] = tv;
            int ti = bi[p]; bi[p] = bi[p - 1]; bi[p - 1] = ti;
        }
    }
}

// grid = (NBLK, B), block = 256.  partial: [B][NBLK][Tpad][NP]
template <int NP>
static __global__ __launch_bounds__(256) void topn_partial_kernel(const float* __restrict__ cells,
                                                           ValIdx* __restrict__ partial, int K, int Tpad) {
    __shared__ ValIdx sh[256 * NP];
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int t = threadIdx.x % Tpad, sub = threadIdx.x / Tpad, nsub = 256 / Tpad;
    const int chunk = (K + nblk - 1) / nblk;
    const int c_begin = blk * chunk, c_end = min(K, c_begin + chunk);
    float bv[NP];
    int bi[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { bv[p] = kNegInf; bi[p] = 0x7fffffff; }
    const float* base = cells + (size_t)b * K * Tpad + t;
    for (int c = c_begin + sub; c < c_end; c += nsub) topn_insert<NP>(bv, bi, base[(size_t)c * Tpad], c);
#pragma unroll
    for (int p = 0; p < NP; ++p) sh[threadIdx.x * NP + p] = ValIdx{bv[p], bi[p]};
    __syncthreads();
    if (sub == 0) {
        for (int s = 1; s < nsub; ++s)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                ValIdx x = sh[(s * Tpad + t) * NP + p];
                topn_insert<NP>(bv, bi, x.v, x.i);
            }
        ValIdx* out = partial + (((size_t)b * nblk + blk) * Tpad + t) * NP;
#pragma unroll
        for (int p = 0; p < NP; ++p) out[p] = ValIdx{bv[p], bi[p]};
    }
}

// grid = B, block = Tpad.  sel: [B][Tpad][NP] centroid ids (0-based)
template <int NP>
static __global__ void topn_final_kernel(const ValIdx* __restrict__ partial, int* __restrict__ sel, int nblk,
                                  int Tpad) {
    const int b = blockIdx.x, t = threadIdx.x;
    float bv[NP];
    int bi[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { bv[p] = kNegInf; bi[p] = 0x7fffffff; }
    for (int blk = 0; blk < nblk; ++blk) {
        const ValIdx* in = partial + (((size_t)b * nblk + blk) * Tpad + t) * NP;
#pragma unroll
        for (int p = 0; p < NP; ++p) topn_insert<NP>(bv, bi, in[p].v, in[p].i);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) sel[((size_t)b * Tpad + t) * NP + p] = bi[p];
}

// -------------------------------------------------------------------------------------------------
// S1 + S2 fused (T <= 32, nprobe <= 2): the same MFMA tiling as centroid_scores_kernel, but
//   * the next tile's centroid rows are prefetched into registers while the current tile is on the MFMA;
//   * each lane keeps the running top-2 (score desc, index asc) of its token over the centroids it sees, so the
//     T x K score matrix is never re-read (the reference moves it to the host and sorts every row, ranking.jl:30-31);
//   * in two-pass mode the scores are written once, as fp16 pairs {t, t+16} (the layout pass 1 gathers), and the
//     fp32 matrix is not written at all.
// partial: [B][nslots][32][2] with nslots = waves * 2 halves.  grid = (gx, B), block = 128, LDS = 2*32*132*4.
// -------------------------------------------------------------------------------------------------
template <bool WRITE_HALF>
static __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void centroid_top2_kernel(const float* __restrict__ C,
                                                                    const float* __restrict__ Q,
                                                                    ValIdx* __restrict__ partial,
                                                                    uint32_t* __restrict__ cells16, int K, int T,
                                                                    int n_tiles, const int* __restrict__ only_flagged) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    if (only_flagged && only_flagged[b] == 0) return;
    float* my = lds + wave * (32 * kCentTileStride);
    float qf[64];
    {
        const float* qrow = Q + ((size_t)b * T + (i < T ? i : T - 1)) * kDim;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float4 v = *reinterpret_cast<const float4*>(qrow + 4 * m);
            if (i >= T) v = make_float4(0.f, 0.f, 0.f, 0.f);
            qf[2 * m] = h ? v.y : v.x;
            qf[2 * m + 1] = h ? v.w : v.z;
        }
    }
    float bv[2] = {kNegInf, kNegInf};
    int bi[2] = {0x7fffffff, 0x7fffffff};
    const int waves_total = gridDim.x * 2;
    int tile = blockIdx.x * 2 + wave;
    // prefetch registers: 16 named float4 (a loop-carried array ends up in scratch)
#define CLB_REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define CLB_PF_DECL(m) float4 pf##m;
#define CLB_PF_LOAD(m)                                                                   \
    {                                                                                    \
        int c = tl * 32 + 2 * m + h;                                                     \
        c = c < K ? c : K - 1;                                                           \
        pf##m = *reinterpret_cast<const float4*>(C + (size_t)c * kDim + 4 * i);          \
    }
#define CLB_PF_STORE(m) *reinterpret_cast<float4*>(my + (2 * m + h) * kCentTileStride + 4 * i) = pf##m;
    CLB_REP16(CLB_PF_DECL)
    {
        const int tl = tile < n_tiles ? tile : n_tiles - 1;
        CLB_REP16(CLB_PF_LOAD)
    }
    while (tile < n_tiles) {
        const int c0 = tile * 32;
        CLB_REP16(CLB_PF_STORE)
        const int next = tile + waves_total;
        {
            const int tl = next < n_tiles ? next : n_tiles - 1;   // the last prefetch is redundant, never wrong
            CLB_REP16(CLB_PF_LOAD)
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            float4 a4 = *reinterpret_cast<const float4*>(my + i * kCentTileStride + 4 * m);
            const float a0 = h ? a4.y : a4.x;
            const float a1 = h ? a4.w : a4.z;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qf[2 * m], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qf[2 * m + 1], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;   // ascending in r for a fixed half
            if (c < K) topn_insert<2>(bv, bi, acc[r], c);
            if (WRITE_HALF) {
                const float other = __shfl_xor(acc[r], 1, 64);   // token i ^ 1, same centroid row
                if ((i & 1) == 0 && c < K) {                     // row of 32 fp16 in token order: pairs {i, i+1}
                    const __half2 hv = __floats2half2_rn(acc[r], other);
                    cells16[((size_t)b * K + c) * 16 + (i >> 1)] = *reinterpret_cast<const uint32_t*>(&hv);
                }
            }
        }
        tile = next;
    }
#undef CLB_PF_DECL
#undef CLB_PF_LOAD
#undef CLB_PF_STORE
#undef CLB_REP16
    const int slot = (blockIdx.x * 2 + wave) * 2 + h;
    const int nslots = gridDim.x * 4;
    ValIdx* out = partial + (((size_t)b * nslots + slot) * 32 + i) * 2;
    out[0] = ValIdx{bv[0], bi[0]};
    out[1] = ValIdx{bv[1], bi[1]};
}

// Merge of the per-slot top-2 lists: one wave per (token, query).  grid = (32, B), block = 64.
// sel layout as topn_final_kernel<2>: [B][32][2].
static __global__ __launch_bounds__(64) void top2_merge_kernel(const ValIdx* __restrict__ partial,
                                                              int* __restrict__ sel, int nslots,
                                                              const int* __restrict__ only_flagged) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (only_flagged && only_flagged[b] == 0) return;
    float bv[2] = {kNegInf, kNegInf};
    int bi[2] = {0x7fffffff, 0x7fffffff};
    for (int sl = lane; sl < nslots; sl += 64) {
        const ValIdx* in = partial + (((size_t)b * nslots + sl) * 32 + t) * 2;
        topn_insert<2>(bv, bi, in[0].v, in[0].i);
        topn_insert<2>(bv, bi, in[1].v, in[1].i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v0 = __shfl_xor(bv[0], o, 64), v1 = __shfl_xor(bv[1], o, 64);
        const int i0 = __shfl_xor(bi[0], o, 64), i1 = __shfl_xor(bi[1], o, 64);
        topn_insert<2>(bv, bi, v0, i0);
        topn_insert<2>(bv, bi, v1, i1);
    }
    if (lane == 0) {
        sel[((size_t)b * 32 + t) * 2] = bi[0];
        sel[((size_t)b * 32 + t) * 2 + 1] = bi[1];
    }
}

// -------------------------------------------------------------------------------------------------
// S3  candidate pids = sort(unique(emb2pid[ union of the selected centroids' IVF lists ]))
// (ranking.jl:32-43).  The IVF is stored as passage ids (emb2pid applied once at load), the union is
// a bitmap over the shard's passages (atomicOr), and the ascending pid list falls out of a bitmap
// compaction -- no sort.  grid = (T*nprobe, B), block = 256.
// -------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void mark_candidates_kernel(const int* __restrict__ sel,
                                                              const uint32_t* __restrict__ ivf_off,
                                                              const uint32_t* __restrict__ ivf_pid,
                                                              uint32_t* __restrict__ bitmap, int T,
                                                              int Tpad, int NP, int nprobe, int W) {
    const int b = blockIdx.y;
    const int t = blockIdx.x / nprobe, p = blockIdx.x % nprobe;
    const int* s = sel + (size_t)b * Tpad * NP;
    const int cid = s[t * NP + p];
    // the same centroid selected by an earlier (token, rank): its list is already being marked.  One entry per thread
    // and a block-wide OR: the serial scan this replaces was a chain of up to T*nprobe dependent L2 round trips in
    // every thread -- most of the kernel's time
    bool dup = false;
    for (int base = 0; base < (int)blockIdx.x; base += 256) {
        const int e = base + (int)threadIdx.x;
        if (e < (int)blockIdx.x && s[(e / nprobe) * NP + (e % nprobe)] == cid) dup = true;
    }
    if (__syncthreads_or(dup)) return;
    (void)T;
    uint32_t* bm = bitmap + (size_t)b * W;
    const uint32_t lo = ivf_off[cid], hi = ivf_off[cid + 1];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint32_t pid = ivf_pid[i];
        atomicOr(&bm[pid >> 5], 1u << (pid & 31));
    }
}

constexpr int kScanBlock = 256;      // threads
constexpr int kWordsPerThread = 4;   // bitmap words per thread in count/emit

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* sh /*>=8 ints*/, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    int base = 0;
    int tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int s = sh[w];
        if (w < wave) base += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return base + x - v;
}

// mark + count in one kernel WITHOUT global atomics.  Device-scope atomics on the bitmap are served past the
// (per-XCD, mutually incoherent) L2s: 1.25 M of them per 32-query batch cost 45 us -- the fabric request rate, not the
// 4 us their bytes would suggest.  Here a work-group owns a SLICE of one query's bitmap (kMarkSliceBlocks count blocks
// = 4096 words = 131 072 passages) in LDS, reads ALL selected IVF lists of the query (they stay in L2: 8 slices re-read
// 156 KB each) and keeps the pids that fall into its slice, then stores the slice with plain stores and leaves the
// per-block popcounts the emit kernel wants.  kMarkLists lists are in flight per round.
// grid = (ceil(nblk / kMarkSliceBlocks), B), block = 1024.
constexpr int kMarkSliceBlocks = 4;
constexpr int kMarkSliceBlocksBig = 16;   // with slice_bounds_kernel (shards of more than 16 small slices)
constexpr int kMarkLists = 64;       // IVF lists per round: 16 threads of the 1 024 walk one list
constexpr int kMarkDepth = 16;       // entries of its list a thread keeps in flight
constexpr int kMarkSliceWords = kMarkSliceBlocks * kScanBlock * kWordsPerThread;   // 4096

// SEARCH: shards with more than 16 slices (2 M passages and up).  Re-reading every list in every slice would cost
// nslices x the list bytes, so the work-group first finds ITS part of each list -- the lists hold non-decreasing
// passage ids (checked once at load: ivf_lists_sorted_kernel; an index without that property keeps the atomic path) --
// which slice_bounds_kernel has found for all slices at once (one thread per list and slice boundary: the searches are
// chains of ~12 dependent loads, far too long to sit in front of every marking work-group -- that version was no faster
// than the atomic path).
// bounds[b][l][j] = first entry of list l of query b with passage id >= j * (passages per slice); j = 0..nslices.
// grid = (ceil(nl * (nslices + 1) / 256), B), block = 256.
static __global__ __launch_bounds__(256) void slice_bounds_kernel(const int* __restrict__ sel, const uint32_t* __restrict__ ivf_off,
                                                                  const uint32_t* __restrict__ ivf_pid, int T, int Tpad, int NP,
                                                                  int nprobe, int nslices, uint32_t slice_passages,
                                                                  uint32_t* __restrict__ bounds) {
    const int b = blockIdx.y, nl = T * nprobe;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nl * (nslices + 1)) return;
    const int l = idx / (nslices + 1), j = idx % (nslices + 1);
    const int cid = sel[(size_t)b * Tpad * NP + (l / nprobe) * NP + (l % nprobe)];
    uint32_t a = ivf_off[cid], e = ivf_off[cid + 1];
    if (j == nslices) {
        a = e;                                            // everything is below the end of the last slice
    } else {
        const uint32_t target = (uint32_t)j * slice_passages;
        while (a < e) {
            const uint32_t mid = a + ((e - a) >> 1);
            if (ivf_pid[mid] < target) a = mid + 1; else e = mid;
        }
    }
    bounds[((size_t)b * nl + l) * (nslices + 1) + j] = a;
}

// SB = count blocks (1 024 bitmap words = 32 768 passages each) per slice: 4 (16 KB of LDS) for small shards, 16 (64 KB) with
// SEARCH -- a work-group's fixed costs (zeroing, two rounds of list loads, the write-out) are ~30 us whatever the slice
// holds, and a 10 M-passage shard has 77 four-block slices per query.
template <bool SEARCH, int SB>
static __global__ __launch_bounds__(1024) void mark_count_kernel(const int* __restrict__ sel,
                                                                 const uint32_t* __restrict__ ivf_off,
                                                                 const uint32_t* __restrict__ ivf_pid,
                                                                 uint32_t* __restrict__ bitmap,
                                                                 int* __restrict__ blocksum, int T, int Tpad, int NP,
                                                                 int nprobe, int W, int nblk,
                                                                 const uint32_t* __restrict__ bounds = nullptr) {
    __shared__ uint32_t lbm[(SB * 1024)];
    __shared__ int cnt[SB];
    __shared__ uint32_t s_lo[kMarkLists], s_hi[kMarkLists];
    const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < (SB * 1024); i += 1024) lbm[i] = 0u;
    if (tid < SB) cnt[tid] = 0;
    __syncthreads();
    const int* s = sel + (size_t)b * Tpad * NP;
    const uint32_t p_lo = (uint32_t)slice * (SB * 1024) * 32u, p_n = (uint32_t)(SB * 1024) * 32u;
    const int nl = T * nprobe;
    // kMarkLists lists per round, 1024 / kMarkLists threads each.  The bounds of the round's lists are fetched ONCE (one thread per
    // list) and shared through LDS -- every thread used to load all of them itself: 1 536 broadcast loads per wave and round,
    // two thirds of this kernel's 30 us -- and a thread keeps kMarkDepth entries of its list in flight.
    constexpr int kPer = 1024 / kMarkLists;
    const int gl = tid / kPer, sub = tid % kPer;
    for (int l0 = 0; l0 < nl; l0 += kMarkLists) {
        if (l0) __syncthreads();                                   // the previous round's bounds have been read
        if (tid < kMarkLists) {
            const int l = l0 + tid;
            uint32_t a = 0u, e = 0u;
            if (l < nl) {
                if (SEARCH) {
                    const uint32_t* bq = bounds + ((size_t)b * nl + l) * (gridDim.x + 1) + slice;   // [query][list][slice boundary]
                    a = bq[0];
                    e = bq[1];
                } else {
                    const int cid = s[(l / nprobe) * NP + (l % nprobe)];
                    a = ivf_off[cid];
                    e = ivf_off[cid + 1];
                }
            }
            s_lo[tid] = a;
            s_hi[tid] = e;
        }
        __syncthreads();
        const uint32_t e = s_hi[gl];
        for (uint32_t i0 = s_lo[gl] + (uint32_t)sub; i0 < e; i0 += (uint32_t)(kPer * kMarkDepth)) {
            uint32_t pid[kMarkDepth];
#pragma unroll
            for (int u = 0; u < kMarkDepth; ++u) {
                const uint32_t i = i0 + (uint32_t)(u * kPer);
                pid[u] = i < e ? ivf_pid[i] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < kMarkDepth; ++u) {
                const uint32_t q = pid[u] - p_lo;                   // 0xffffffff - p_lo >= p_n: never in range
                if (q < p_n && pid[u] != 0xffffffffu) atomicOr(&lbm[q >> 5], 1u << (q & 31));
            }
        }
    }
    __syncthreads();
    uint32_t* bm = bitmap + (size_t)b * W;
    const int w0 = slice * (SB * 1024);
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const int wi = j * 1024 + tid;
        const uint32_t word = lbm[wi];
        if (w0 + wi < W) bm[w0 + wi] = word;
        int c = __popc(word);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0 && c) atomicAdd(&cnt[j], c);
    }
    __syncthreads();
    if (tid < SB && slice * SB + tid < nblk)
        blocksum[(size_t)b * nblk + slice * SB + tid] = cnt[tid];
}

// grid = (nblk, B)
static __global__ __launch_bounds__(kScanBlock) void bitmap_count_kernel(const uint32_t* __restrict__ bitmap,
                                                                  int* __restrict__ blocksum, int W) {
    __shared__ int sh[8];
    const int b = blockIdx.y;
    const uint32_t* bm = bitmap + (size_t)b * W;
    const int w0 = (blockIdx.x * kScanBlock + threadIdx.x) * kWordsPerThread;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kWordsPerThread; ++j)
        if (w0 + j < W) cnt += __popc(bm[w0 + j]);
    int total;
    (void)block_exclusive_scan_256(cnt, sh, total);
    if (threadIdx.x == 0) blocksum[(size_t)b * gridDim.x + blockIdx.x] = total;
}

// grid = (nblk, B): emit ascending local pids (0-based) and clear the bitmap for the next query
// The position of a block's first candidate is the sum of the counts of the blocks before it: every block adds them up
// itself (at most a few hundred values) instead of waiting for a scan kernel; block 0 also leaves the total in ncand.
static __global__ __launch_bounds__(kScanBlock) void bitmap_emit_kernel(uint32_t* __restrict__ bitmap,
                                                                 const int* __restrict__ blockcnt,
                                                                 uint32_t* __restrict__ cand,
                                                                 const uint32_t* __restrict__ doc_off,
                                                                 uint2* __restrict__ cand_hdr, int W,
                                                                 size_t cand_cap, int* __restrict__ ncand) {
    __shared__ int sh[8];
    const int b = blockIdx.y;
    int before = 0;
    {
        const int* bc = blockcnt + (size_t)b * gridDim.x;
        const int upto = blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x;      // block 0: everything, for the total
        int part = 0;
        for (int i = threadIdx.x; i < upto; i += kScanBlock) part += bc[i];
        int total_;
        (void)block_exclusive_scan_256(part, sh, total_);
        if (blockIdx.x == 0) { if (threadIdx.x == 0) ncand[b] = total_; }
        else before = total_;
    }
    uint32_t* bm = bitmap + (size_t)b * W;
    const int w0 = (blockIdx.x * kScanBlock + threadIdx.x) * kWordsPerThread;
    uint32_t w[kWordsPerThread];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kWordsPerThread; ++j) {
        w[j] = (w0 + j < W) ? bm[w0 + j] : 0u;
        cnt += __popc(w[j]);
    }
    int total;
    int pos = block_exclusive_scan_256(cnt, sh, total) + before;
    uint32_t* out = cand + (size_t)b * cand_cap;
    uint2* hdr = cand_hdr + (size_t)b * cand_cap;   // {first embedding, length} of every candidate passage
#pragma unroll
    for (int j = 0; j < kWordsPerThread; ++j) {
        uint32_t x = w[j];
        if (x) bm[w0 + j] = 0u;
        while (x) {
            const int bit = __ffs((int)x) - 1;
            x &= x - 1;
            if ((size_t)pos < cand_cap) {
                const uint32_t pid = (uint32_t)((w0 + j) * 32 + bit);
                out[pos] = pid;
                const uint32_t o = doc_off[pid];
                hdr[pos] = make_uint2(o, doc_off[pid + 1] - o);
            }
            ++pos;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// S4+S5+S6 fused, exact: for every candidate passage gather its packed codes/residuals, decompress
// (centroid + bucket weight, L2-normalise), Q'D on the f32 MFMA, per-token max over the passage,
// sequential sum over tokens.  (ranking.jl:46-86, residual.jl:698-784, utils.jl:320-325)
//
// One wave per passage, 16 embeddings per step (v_mfma_f32_16x16x4_f32: rows = embeddings,
// cols = query tokens, k = 4 consecutive dims).  Lane (r = lane&15, g = lane>>4) owns dims 4s+g of
// embedding r, s = 0..31 -- exactly one of the four interleaved partial sums of the canonical norm.
// grid = (G, B), block = 256 (4 waves).
// -------------------------------------------------------------------------------------------------
template <int NBITS>
__device__ __forceinline__ float bucket_weight(uint32_t idx, const float (&w)[4], float wlane) {
    if constexpr (NBITS == 1) {
        return idx ? w[1] : w[0];
    } else if constexpr (NBITS == 2) {
        const float lo = (idx & 1) ? w[1] : w[0];
        const float hi = (idx & 1) ? w[3] : w[2];
        return (idx & 2) ? hi : lo;
    } else {
        return __shfl(wlane, (int)idx, 64);  // table spread over lanes 0..2^NBITS-1
    }
}

// Decompresses the 32 dims {4s+g} of one embedding and L2-normalises; returns the values in x[].
// R = the embedding's packed residual as dwords (NBITS*4 of them).
template <int NBITS>
__device__ __forceinline__ void decompress_lane_dims(const float* __restrict__ cent_row /*+g*/,
                                                     const uint32_t (&R)[NBITS * 4], int g,
                                                     const float (&w)[4], float wlane, float (&x)[32]) {
    float p = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int bitpos = 4 * s * NBITS;  // + g*NBITS at run time; never crosses a dword
        const uint32_t idx = (R[bitpos >> 5] >> ((bitpos & 31) + g * NBITS)) & ((1u << NBITS) - 1u);
        const float v = cent_row[4 * s] + bucket_weight<NBITS>(idx, w, wlane);
        x[s] = v;
        const float sq = v * v;
        p = p + sq;
    }
    // (p0+p1)+(p2+p3): partners differ in g, i.e. lane^16 and lane^32
    const float a = p + __shfl_xor(p, 16, 64);
    const float n2 = a + __shfl_xor(a, 32, 64);
    const float den = sqrtf(n2) + FLT_EPSILON;
#pragma unroll
    for (int s = 0; s < 32; ++s) x[s] = x[s] / den;
}

// Correctly rounded x / den from a correctly rounded reciprocal y = RN(1/den) (Markstein): q0 = RN(x*y),
// r = x - den*q0 (exact in the fma), q = RN(q0 + r*y).  Checked against IEEE division on 4.5e8 cases including
// every 24-bit significand of den; valid away from the over/underflow ranges (the caller guards den).
__device__ __forceinline__ float div_by_reciprocal(float x, float den, float y) {
    const float q0 = x * y;
    const float r = fmaf(-den, q0, x);
    return fmaf(r, y, q0);
}

// Bucket-weight table in LDS, replicated over the 32 banks: tbl[idx][bank] = weights[idx].  Lane L reads
// tbl[idx][L & 31], so the data-dependent lookups of a wave never collide on a bank (a [field][byte] table indexed
// by the packed byte cost ~5 LDS cycles per read in conflicts).
template <int NBITS>
constexpr int weight_table_floats() { return (1 << NBITS) * 32; }
template <int NBITS>
__device__ __forceinline__ void fill_weight_table(float* tbl, const float* __restrict__ weights) {
    for (int i = threadIdx.x; i < weight_table_floats<NBITS>(); i += blockDim.x) tbl[i] = weights[i >> 5];
}

// Decompresses the 32 dims {4s+g} of one embedding (canonical order, see decompress_lane_dims) using the LDS
// weight table and the reciprocal division.  R = packed residual dwords.  The centroid values come either
// straight from global memory (cent_row, used by the stand-alone decompress kernel) or from a staged LDS tile
// (ctile != nullptr: slot(s, r) = s*16 + (r ^ (s & 15)), 16 B per slot -- see score_exact_kernel).
template <int NBITS>
__device__ __forceinline__ void decompress_lane_dims_fast(const float* __restrict__ cent_row /*+g*/,
                                                          const uint32_t (&R)[NBITS * 4], int g,
                                                          const float* __restrict__ tbl, float (&x)[32],
                                                          const float* ctile = nullptr, int r = 0) {
    float p = 0.f;
    const float* tbl_lane = tbl + (threadIdx.x & 31);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        // dim d = 4s+g starts at bit d*NBITS = 4s*NBITS + g*NBITS; a group of four dims never crosses a dword
        const int bit0 = 4 * s * NBITS;
        const uint32_t idx = __builtin_amdgcn_ubfe(R[bit0 >> 5], (uint32_t)((bit0 & 31) + g * NBITS), (uint32_t)NBITS);
        // staged tile: the four words of a 16-byte slot are rotated by two for slots >= 8 -- without it slots j and j + 8
        // share their banks and every one of these 32 reads took a second LDS cycle (SQ_LDS_BANK_CONFLICT = one per MFMA)
        const float c = ctile ? ctile[(s * 16 + (r ^ (s & 15))) * 4 + ((g + 2 * (((r >> 3) ^ ((s & 15) >> 3)) & 1)) & 3)] : cent_row[4 * s];
        const float val = c + tbl_lane[idx * 32];
        x[s] = val;
        const float sq = val * val;
        p = p + sq;
    }
    const float a = p + __shfl_xor(p, 16, 64);
    const float n2 = a + __shfl_xor(a, 32, 64);
    const float den = sqrtf(n2) + FLT_EPSILON;
    if (__builtin_expect(den > 1e-18f && den < 1e18f, 1)) {
        const float y = 1.0f / den;
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] = div_by_reciprocal(x[s], den, y);
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] = x[s] / den;
    }
}

// grid = (G, B), block = 256 (4 waves).  `list` (optional) restricts the work to the listed candidate slots
// (two-pass mode).  LDS:
//   * the bucket-weight table;
//   * the query operand of the current 32-token group, [chunk k][lane][4 floats] = {Q[t0][8k+g'], Q[t1][8k+g'],
//     Q[t0][8k+4+g'], Q[t1][8k+4+g']}: one conflict-free ds_read_b128 feeds four MFMAs, no resident registers;
//   * per wave an 8-KB tile of the step's 16 centroid rows.  The rows are fetched as whole 512-B rows (8
//     coalesced dwordx4 loads, issued one step ahead so that they fly during the MFMA phase) instead of 32
//     strided dword gathers per lane -- the gathers alone held the kernel at 0.9 ms for 3.4 M embeddings (TA
//     bound); 16-B chunk c of row r sits at slot c*16 + (r ^ (c & 15)), which makes both the b128 writes and
//     the stride-4 b32 reads of lane (r, g) at most 2-way bank conflicted.
// This kernel scores EVERY embedding of a passage (single-pass mode, nbits 1/4, T > 32); the two-pass mode uses
// score_exact_flat_kernel below, which multiplies only the rows named by the 256-bit row mask of each listed
// passage (passages longer than kMaxMaskedRows embeddings take every row there too).
constexpr int kMaxMaskedRows = 256;

template <int NBITS>
static __global__ __launch_bounds__(256, 3) void score_exact_kernel(
    const float* __restrict__ C, const float* __restrict__ weights, const uint32_t* __restrict__ codes0,
    const uint8_t* __restrict__ residuals, const uint2* __restrict__ cand_hdr, const float* __restrict__ Q,
    const int* __restrict__ ncand, float* __restrict__ scores, int T, size_t cand_cap,
    const int* __restrict__ list /*optional*/, const int* __restrict__ nlist) {
    constexpr int RD = NBITS * 4;  // residual dwords per embedding
    __shared__ float tbl[weight_table_floats<NBITS>()];
    __shared__ __attribute__((aligned(16))) float qlds[16 * 64 * 4];    // 16 KB
    __shared__ __attribute__((aligned(16))) float ctiles[4 * 512 * 4];  // 4 waves x 8 KB
    fill_weight_table<NBITS>(tbl, weights);
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int wave_global = blockIdx.x * 4 + wave;
    const int nwaves = gridDim.x * 4;
    const int n = list ? nlist[b] : ncand[b];
    const int TT = (T + 31) >> 5;
    const uint2* hdr = cand_hdr + (size_t)b * cand_cap;
    const int* lst = list ? list + (size_t)b * cand_cap : nullptr;
    float* ctile = ctiles + wave * 2048;
    // row loader: instruction m covers rows 2m and 2m+1; this lane moves chunk c4 of row 2m + (lane >> 5)
    const int c4 = lane & 31, rhalf = lane >> 5;

    // stage the query operand of token group tt: every wave uses the same lane -> (token, k) map
    auto stage_q = [&](int tt) {
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int k = i >> 6, l = i & 63;
            const int rr = l & 15, gg = l >> 4;
            const int t0 = tt * 32 + rr, t1 = t0 + 16;
            const float* qa = Q + ((size_t)b * T + (t0 < T ? t0 : T - 1)) * kDim + gg;
            const float* qb = Q + ((size_t)b * T + (t1 < T ? t1 : T - 1)) * kDim + gg;
            float4 v;
            v.x = t0 < T ? qa[8 * k] : 0.f;
            v.y = t1 < T ? qb[8 * k] : 0.f;
            v.z = t0 < T ? qa[8 * k + 4] : 0.f;
            v.w = t1 < T ? qb[8 * k + 4] : 0.f;
            *reinterpret_cast<float4*>(qlds + (size_t)i * 4) = v;
        }
    };
    const float4* qv = reinterpret_cast<const float4*>(qlds) + lane;

#define CLB_REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define CLB_ROW_DECL(m) float4 row##m;
#define CLB_ROW_LOAD(m)                                                                                        \
    {                                                                                                          \
        const uint32_t ca = __builtin_amdgcn_readlane(code_rows, 2 * m), cb = __builtin_amdgcn_readlane(code_rows, 2 * m + 1); \
        row##m = *reinterpret_cast<const float4*>(C + (size_t)(rhalf ? cb : ca) * kDim + 4 * c4);              \
    }
#define CLB_ROW_STORE(m)                                                                                       \
    {   /* two 8-byte stores: {x, y} and {z, w} swap places in the slots >= 8 (see decompress_lane_dims_fast) */   \
        float* sl_ = ctile + (c4 * 16 + ((2 * m + rhalf) ^ (c4 & 15))) * 4;                                    \
        const int ro_ = 2 * ((((2 * m + rhalf) >> 3) ^ (c4 >> 3)) & 1);                                        \
        *reinterpret_cast<float2*>(sl_ + ro_) = make_float2(row##m.x, row##m.y);                               \
        *reinterpret_cast<float2*>(sl_ + (ro_ ^ 2)) = make_float2(row##m.z, row##m.w);                         \
    }

    // with one token group the passage loop runs per wave; with several, all waves of the workgroup walk the
    // groups together (the staged operand is shared), so the loop order is group-major
    for (int tt = 0; tt < TT; ++tt) {
        __syncthreads();
        stage_q(tt);
        __syncthreads();
        // The header of the next passage and the slot of the one after are requested while the current passage is
        // being scored: the list -> header -> codes -> rows chain of dependent loads otherwise leaves the wave idle
        // for several memory round trips per passage.
        int slot_cur = wave_global < n ? (lst ? lst[wave_global] : wave_global) : 0;
        int slot_nxt = wave_global + nwaves < n ? (lst ? lst[wave_global + nwaves] : wave_global + nwaves) : 0;
        uint2 h_cur = hdr[slot_cur];
        for (int j = wave_global; j < n; j += nwaves) {
            const int slot = slot_cur;  // index into the candidate array
            const uint32_t off = __builtin_amdgcn_readfirstlane(h_cur.x);
            const int len = (int)__builtin_amdgcn_readfirstlane(h_cur.y);
            {
                const int j1 = j + nwaves, j2 = j + 2 * nwaves;
                slot_cur = slot_nxt;
                if (j1 < n) h_cur = hdr[slot_nxt];
                slot_nxt = j2 < n ? (lst ? lst[j2] : j2) : 0;
            }
            float m0 = kNegInf, m1 = kNegInf;
            // software prefetch, two steps deep for the codes (the row loads of step i+1 need code(i+1) early)
            auto row_at = [&](int el) -> uint32_t { return (uint32_t)(el < len ? el : len - 1); };
            auto code_at = [&](int base) { return codes0[off + row_at(base + r)]; };
            uint32_t Rn[RD];
            auto load_res = [&](int base) {
                const uint32_t e = off + row_at(base + r);
                const uint32_t* rp = reinterpret_cast<const uint32_t*>(residuals + (size_t)e * (RD * 4));
#pragma unroll
                for (int k4 = 0; k4 < RD; k4 += 4) {
                    uint4 v = *reinterpret_cast<const uint4*>(rp + k4);
                    Rn[k4] = v.x; Rn[k4 + 1] = v.y; Rn[k4 + 2] = v.z; Rn[k4 + 3] = v.w;
                }
            };
            uint32_t code_rows = code_at(0);      // code of embedding r in the step whose rows are being fetched
            uint32_t code_next = code_at(16);     // ... and one step further
            load_res(0);
            CLB_REP8(CLB_ROW_DECL)
            CLB_REP8(CLB_ROW_LOAD)                // rows of step 0
            for (int base = 0; base < len; base += 16) {
                uint32_t R[RD];
#pragma unroll
                for (int k4 = 0; k4 < RD; ++k4) R[k4] = Rn[k4];
                CLB_REP8(CLB_ROW_STORE)
                __builtin_amdgcn_wave_barrier();
                if (base + 16 < len) load_res(base + 16);
                float x[32];
                decompress_lane_dims_fast<NBITS>(nullptr, R, g, tbl, x, ctile, r);
                __builtin_amdgcn_wave_barrier();
                // rows of the next step fly during the MFMA phase; its codes were requested a step ago
                code_rows = code_next;
                if (base + 16 < len) { CLB_REP8(CLB_ROW_LOAD) }
                code_next = code_at(base + 32);
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                // fully unrolled (x[] stays statically indexed) with the query operand read one k-step ahead, so
                // the LDS latency hides behind the four MFMAs in flight instead of stalling every group
                float4 qnext = qv[0];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float4 q = qnext;
                    if (k < 15) qnext = qv[(k + 1) * 64];
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k], q.x, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k], q.y, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k + 1], q.z, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k + 1], q.w, a1, 0, 0, 0);
                }
                // accumulator rows: embedding base + 4*g + reg ; cols: token r.  Rows past the passage are
                // clamped copies of its last embedding: they cannot change the max.
                m0 = fmaxf(fmaxf(m0, fmaxf(a0[0], a0[1])), fmaxf(a0[2], a0[3]));
                m1 = fmaxf(fmaxf(m1, fmaxf(a1[0], a1[1])), fmaxf(a1[2], a1[3]));
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 16, 64));
            m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, 16, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
            // sequential sum over tokens (ranking.jl:83 `sum(maximum(..., dims=2))`); token groups are added in
            // order, so the running total lives in scores[] between groups
            float total = tt == 0 ? 0.f : scores[(size_t)b * cand_cap + slot];
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const float v = __shfl(t < 16 ? m0 : m1, t & 15, 64);
                if (tt * 32 + t < T) total = total + v;
            }
            if (lane == 0) scores[(size_t)b * cand_cap + slot] = total;
        }
    }
#undef CLB_REP8
#undef CLB_ROW_DECL
#undef CLB_ROW_LOAD
#undef CLB_ROW_STORE
}

// hist[bin] += 1 for every active lane.  MaxSim scores of one query share their exponent and leading mantissa bits,
// so in the first radix passes a whole wave hits one or two bins and per-lane LDS atomics serialise 64 deep; lanes
// sharing a bin are therefore merged into one atomic (up to four distinct bins per call, plain atomics after that).
// -------------------------------------------------------------------------------------------------
// Pass 2 of the two-pass mode as a flat step pipeline (nbits 2, T <= 32).  Same arithmetic per step as
// score_exact_kernel<2, true>; what changes is the control structure.  With the row subset a listed passage is
// only ~2.4 steps of 16 rows, so a per-passage loop spends most of its time in the start-up chain of dependent
// loads (list -> header -> codes -> centroid rows).  Here each wave walks a wave-uniform iterator over the steps
// of ITS passages (headers and row masks of 64 passages at a time sit in VGPRs and are extracted with v_readlane,
// as in score_approx32_kernel) and keeps three steps in flight across passage boundaries:
//   stage A (step i+2): row index of lane r = the (base + r)-th set bit of the passage's 256-bit mask (a
//                       branch-free popcount search), then its code and 32-B residual are requested;
//   stage G (step i+1): the 16 centroid rows are requested as whole 512-B rows;
//   stage C (step i)  : rows -> swizzled LDS tile, decompress, 64 fp32 MFMAs, running per-token max; at the last
//                       step of a passage reduce, sum the tokens in order, store the score.
// Steps past the end of the wave's work read a dummy address and are discarded.
// grid = (G, B), block = 256; the waves of a query's work-groups take its listed passages round-robin.
// -------------------------------------------------------------------------------------------------
struct ExactStepTag {
    int slot;   // candidate slot to store the score to, -1 = dummy step
    int slot1;  // (round 6) the passage that BEGINS inside this step at row p, see below
    int info;   // bits 0..4: p = 16, or the row at which the wave's next passage starts inside this step (rows [0, p) end passage
                // `slot`, rows [p, 16) begin passage `slot1`); bit 8: `slot` ends in this step; bit 9: `slot1` ends in it too
};
// the n-th (0-based) set bit of a 256-bit mask held in four wave-uniform 64-bit words; n < popcount(mask)
__device__ __forceinline__ uint32_t nth_set_bit_256(unsigned long long w0, unsigned long long w1,
                                                    unsigned long long w2, unsigned long long w3, int n) {
    const int c0 = __popcll(w0), c1 = c0 + __popcll(w1), c2 = c1 + __popcll(w2);
    unsigned long long w = w0;
    int add = 0, loc = n;
    if (n >= c0) { w = w1; add = 64; loc = n - c0; }
    if (n >= c1) { w = w2; add = 128; loc = n - c1; }
    if (n >= c2) { w = w3; add = 192; loc = n - c2; }
    uint32_t x = (uint32_t)w;
    const int cl = __popc(x);
    if (loc >= cl) { x = (uint32_t)(w >> 32); add += 32; loc -= cl; }
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const int t = __popc(x & ((1u << width) - 1u));
        if (loc >= t) { x >>= width; add += width; loc -= t; }
    }
    return (uint32_t)add;
}

static __global__ __launch_bounds__(256, 3) void score_exact_flat_kernel(
    const float* __restrict__ C, const float* __restrict__ weights, const uint32_t* __restrict__ codes0,
    const uint8_t* __restrict__ residuals, const uint2* __restrict__ cand_hdr, const float* __restrict__ Q,
    float* __restrict__ scores, int T, size_t cand_cap, const int* __restrict__ list,
    const int* __restrict__ nlist, const unsigned long long* __restrict__ rowmask) {
    constexpr int NBITS = 2;
    __shared__ float tbl[weight_table_floats<NBITS>()];
    __shared__ __attribute__((aligned(16))) float qlds[16 * 64 * 4];    // 16 KB
    __shared__ __attribute__((aligned(16))) float ctiles[4 * 512 * 4];  // 4 waves x 8 KB
    fill_weight_table<NBITS>(tbl, weights);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ctile = ctiles + wave * 2048;
    const int c4 = lane & 31, rhalf = lane >> 5;    // row loader: instruction m moves chunk c4 of row 2m + rhalf
    const float4* qv = reinterpret_cast<const float4*>(qlds) + lane;

#define CLB_REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define CLB_ROW_DECL(m) float4 rowA##m;
#define CLB_ROW_LOAD(P, CODE, m)                                                                               \
    {                                                                                                          \
        const uint32_t ca = __builtin_amdgcn_readlane(CODE, 2 * m), cb = __builtin_amdgcn_readlane(CODE, 2 * m + 1); \
        row##P##m = *reinterpret_cast<const float4*>(C + (size_t)(rhalf ? cb : ca) * kDim + 4 * c4);           \
    }
#define CLB_ROW_LOAD8(P, CODE)                                                                                 \
    CLB_ROW_LOAD(P, CODE, 0) CLB_ROW_LOAD(P, CODE, 1) CLB_ROW_LOAD(P, CODE, 2) CLB_ROW_LOAD(P, CODE, 3)        \
    CLB_ROW_LOAD(P, CODE, 4) CLB_ROW_LOAD(P, CODE, 5) CLB_ROW_LOAD(P, CODE, 6) CLB_ROW_LOAD(P, CODE, 7)
#define CLB_ROW_STORE(P, m)                                                                                    \
    {   /* two 8-byte stores: {x, y} and {z, w} swap places in the slots >= 8 (see decompress_lane_dims_fast) */   \
        float* sl_ = ctile + (c4 * 16 + ((2 * m + rhalf) ^ (c4 & 15))) * 4;                                    \
        const int ro_ = 2 * ((((2 * m + rhalf) >> 3) ^ (c4 >> 3)) & 1);                                        \
        *reinterpret_cast<float2*>(sl_ + ro_) = make_float2(row##P##m.x, row##P##m.y);                         \
        *reinterpret_cast<float2*>(sl_ + (ro_ ^ 2)) = make_float2(row##P##m.z, row##P##m.w);                   \
    }
#define CLB_ROW_STORE8(P)                                                                                      \
    CLB_ROW_STORE(P, 0) CLB_ROW_STORE(P, 1) CLB_ROW_STORE(P, 2) CLB_ROW_STORE(P, 3)                            \
    CLB_ROW_STORE(P, 4) CLB_ROW_STORE(P, 5) CLB_ROW_STORE(P, 6) CLB_ROW_STORE(P, 7)

    {
        const int b = blockIdx.y;
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {     // query operand, see score_exact_kernel
            const int k = i >> 6, l = i & 63;
            const int rr = l & 15, gg = l >> 4;
            const int t0 = rr, t1 = rr + 16;
            const float* qa = Q + ((size_t)b * T + (t0 < T ? t0 : T - 1)) * kDim + gg;
            const float* qb = Q + ((size_t)b * T + (t1 < T ? t1 : T - 1)) * kDim + gg;
            float4 v;
            v.x = t0 < T ? qa[8 * k] : 0.f;
            v.y = t1 < T ? qb[8 * k] : 0.f;
            v.z = t0 < T ? qa[8 * k + 4] : 0.f;
            v.w = t1 < T ? qb[8 * k + 4] : 0.f;
            *reinterpret_cast<float4*>(qlds + (size_t)i * 4) = v;
        }
        __syncthreads();
        const int n = nlist[b];
        const int* lst = list + (size_t)b * cand_cap;
        const uint2* hdr = cand_hdr + (size_t)b * cand_cap;
        const unsigned long long* mw = rowmask + (size_t)b * cand_cap * 4;
        float* out = scores + (size_t)b * cand_cap;
        const int stride = gridDim.x * 4;

        for (int j0 = blockIdx.x * 4 + wave; j0 < n; j0 += 64 * stride) {
            // lane k of the wave holds the description of the wave's k-th passage of this batch
            const int jl = j0 + lane * stride;
            const int jj = jl < n ? jl : j0;
            const int slot_l = lst[jj];
            const uint2 hv = hdr[slot_l];
            const unsigned long long mk0 = mw[(size_t)jj * 4], mk1 = mw[(size_t)jj * 4 + 1],
                                     mk2 = mw[(size_t)jj * 4 + 2], mk3 = mw[(size_t)jj * 4 + 3];
            const bool ident_l = (int)hv.y > kMaxMaskedRows;      // too long for the mask: every row
            // rows to multiply; bit 31 flags the identity mapping
            const uint32_t nj_l = ident_l ? (hv.y | 0x80000000u)
                                          : (uint32_t)(__popcll(mk0) + __popcll(mk1) + __popcll(mk2) + __popcll(mk3));
            const int nd = (n - j0 + stride - 1) / stride < 64 ? (n - j0 + stride - 1) / stride : 64;
            int it_k = 0, it_base = 0;
            uint32_t it_off, it_nj;
            int it_slot;
            unsigned long long it_m0, it_m1, it_m2, it_m3;
#define CLB_IT_LOAD(KK)                                                                                        \
    {                                                                                                          \
        it_off = __builtin_amdgcn_readlane(hv.x, KK);                                                          \
        it_nj = __builtin_amdgcn_readlane(nj_l, KK);                                                           \
        it_slot = __builtin_amdgcn_readlane(slot_l, KK);                                                       \
        it_m0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk0 >> 32), KK) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk0, KK); \
        it_m1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk1 >> 32), KK) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk1, KK); \
        it_m2 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk2 >> 32), KK) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk2, KK); \
        it_m3 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk3 >> 32), KK) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk3, KK); \
    }
            CLB_IT_LOAD(0)

            // stage A: describe the next step, request the code and residual of this lane's row.
            // Round 6: a passage's selected rows (26 on average) used to be padded to whole steps of 16 -- a fifth of all row slots held
            // copies of a last row.  When the current passage has fewer than 16 rows left and the wave has a further passage in this
            // chunk, the step's remaining row slots [p, 16) now start that passage (at most two passages per step: a next passage
            // shorter than 16 - p rows is padded with copies of ITS last row and ends in the same step).
#define CLB_XSTAGE_A(CODE, R0, R1, TAG)                                                                        \
    {                                                                                                          \
        const bool live = it_k < nd;                                                                           \
        const int nj = (int)(it_nj & 0x7fffffffu);                                                             \
        const int rem = nj - it_base;                                                                          \
        uint32_t e;                                                                                            \
        if (rem < 16 && it_k + 1 < nd) {     /* (wave-uniform) the boundary step of two passages */             \
            const int kn = it_k + 1;                                                                           \
            const uint32_t nx_off = __builtin_amdgcn_readlane(hv.x, kn);                                       \
            const uint32_t nx_nj = __builtin_amdgcn_readlane(nj_l, kn);                                        \
            const int nx_slot = __builtin_amdgcn_readlane(slot_l, kn);                                         \
            const unsigned long long nx_m0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk0 >> 32), kn) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk0, kn); \
            const unsigned long long nx_m1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk1 >> 32), kn) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk1, kn); \
            const unsigned long long nx_m2 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk2 >> 32), kn) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk2, kn); \
            const unsigned long long nx_m3 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(mk3 >> 32), kn) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)mk3, kn); \
            const int njn = (int)(nx_nj & 0x7fffffffu);                                                        \
            const int take = 16 - rem < njn ? 16 - rem : njn;                                                  \
            const bool second = r >= rem;            /* this lane's row slot belongs to the next passage */      \
            const int ia = it_base + r;                                                                        \
            const int ib = r - rem < njn ? r - rem : njn - 1;                                                  \
            const int idx = second ? ib : ia;                                                                  \
            const bool ident = ((second ? nx_nj : it_nj) & 0x80000000u) != 0u;                                 \
            const uint32_t row = ident ? (uint32_t)idx                                                         \
                                       : nth_set_bit_256(second ? nx_m0 : it_m0, second ? nx_m1 : it_m1,       \
                                                         second ? nx_m2 : it_m2, second ? nx_m3 : it_m3, idx); \
            e = (second ? nx_off : it_off) + row;                                                              \
            const int last1 = njn <= 16 - rem;                                                                 \
            TAG.slot = it_slot;                                                                                \
            TAG.slot1 = nx_slot;                                                                               \
            TAG.info = rem | 0x100 | (last1 << 9);                                                             \
            it_k += 1;                                                                                         \
            if (last1) {                                                                                       \
                it_k += 1;                                                                                     \
                const int kk = it_k < 64 ? it_k : 63;                                                          \
                CLB_IT_LOAD(kk)                                                                                \
                it_base = 0;                                                                                   \
            } else {                                                                                           \
                it_off = nx_off; it_nj = nx_nj; it_slot = nx_slot;                                             \
                it_m0 = nx_m0; it_m1 = nx_m1; it_m2 = nx_m2; it_m3 = nx_m3;                                    \
                it_base = take;                                                                                \
            }                                                                                                  \
        } else {                                                                                               \
            const int idx = it_base + r < nj ? it_base + r : nj - 1;                                           \
            const uint32_t row = (it_nj & 0x80000000u) ? (uint32_t)idx : nth_set_bit_256(it_m0, it_m1, it_m2, it_m3, idx); \
            e = live ? it_off + row : 0u;                                                                      \
            const int last = it_base + 16 >= nj;                                                               \
            TAG.slot = live ? it_slot : -1;                                                                    \
            TAG.slot1 = -1;                                                                                    \
            TAG.info = 16 | (last << 8);                                                                       \
            it_base += 16;                                                                                     \
            if (last) {                                                                                        \
                it_k += 1;                                                                                     \
                const int kk = it_k < 64 ? it_k : 63;                                                          \
                CLB_IT_LOAD(kk)                                                                                \
                it_base = 0;                                                                                   \
            }                                                                                                  \
        }                                                                                                      \
        CODE = codes0[e];                                                                                      \
        const uint4* rp = reinterpret_cast<const uint4*>(residuals + (size_t)e * 32);                          \
        R0 = rp[0];                                                                                            \
        R1 = rp[1];                                                                                            \
    }
            // stage C: rows (requested one step ago) -> LDS, decompress, request the next step's rows into the
            // registers just freed, MFMAs
#define CLB_XSTAGE_C(P, R0, R1, TAG, CODE_NEXT2)                                                               \
    {                                                                                                          \
        CLB_ROW_STORE8(P)                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                                       \
        const uint32_t R[8] = {R0.x, R0.y, R0.z, R0.w, R1.x, R1.y, R1.z, R1.w};                                \
        float x[32];                                                                                           \
        decompress_lane_dims_fast<NBITS>(nullptr, R, g, tbl, x, ctile, r);                                     \
        __builtin_amdgcn_wave_barrier();                                                                       \
        CLB_ROW_LOAD8(P, CODE_NEXT2)                                                                           \
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};                                            \
        float4 qnext = qv[0];                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                       \
            const float4 q = qnext;                                                                            \
            if (k < 15) qnext = qv[(k + 1) * 64];                                                              \
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k], q.x, a0, 0, 0, 0);                             \
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k], q.y, a1, 0, 0, 0);                             \
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k + 1], q.z, a0, 0, 0, 0);                         \
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2 * k + 1], q.w, a1, 0, 0, 0);                         \
        }                                                                                                      \
        /* rows past the selection are copies of a last row: they cannot change a max.  c0 / c1: the running maxima of the    */ \
        /* passage rows [0, p) belong to; n0 / n1: those of the passage that begins at row p (accumulator row = 4 g + reg)      */ \
        float c0, c1, n0 = kNegInf, n1 = kNegInf;                                                              \
        const int p_ = TAG.info & 31;                                                                          \
        if (p_ >= 16) {                                                                                        \
            c0 = fmaxf(fmaxf(m0, fmaxf(a0[0], a0[1])), fmaxf(a0[2], a0[3]));                                   \
            c1 = fmaxf(fmaxf(m1, fmaxf(a1[0], a1[1])), fmaxf(a1[2], a1[3]));                                   \
        } else {                                                                                               \
            c0 = m0;                                                                                           \
            c1 = m1;                                                                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                 \
                const bool first_ = 4 * g + q_ < p_;                                                           \
                c0 = fmaxf(c0, first_ ? a0[q_] : kNegInf);                                                     \
                c1 = fmaxf(c1, first_ ? a1[q_] : kNegInf);                                                     \
                n0 = fmaxf(n0, first_ ? kNegInf : a0[q_]);                                                     \
                n1 = fmaxf(n1, first_ ? kNegInf : a1[q_]);                                                     \
            }                                                                                                  \
        }                                                                                                      \
        {                                                                                                      \
            const int nfin_ = ((TAG.info >> 8) & 1) + ((TAG.info >> 9) & 1);       /* passages that end in this step */ \
            int cs_ = TAG.slot;                                                                                \
            _Pragma("unroll 1") for (int f_ = 0; f_ < nfin_; ++f_) {                                           \
                c0 = fmaxf(c0, __shfl_xor(c0, 16, 64));                                                        \
                c0 = fmaxf(c0, __shfl_xor(c0, 32, 64));                                                        \
                c1 = fmaxf(c1, __shfl_xor(c1, 16, 64));                                                        \
                c1 = fmaxf(c1, __shfl_xor(c1, 32, 64));                                                        \
                /* sequential sum over tokens (ranking.jl:83).  The maxima reach the adder through v_readlane (a scalar  */ \
                /* operand, no LDS round trip as with ds_bpermute: 32 of those per passage were a quarter of the wave's  */ \
                /* non-MFMA time); tokens past T add +0.0, which leaves a sum that started from +0.0 unchanged            */ \
                float total = 0.f;                                                                             \
                _Pragma("unroll") for (int t = 0; t < 32; ++t) {                                               \
                    const uint32_t sv = __builtin_amdgcn_readlane(__float_as_uint(t < 16 ? c0 : c1), t & 15);  \
                    total = total + __uint_as_float(t < T ? sv : 0u);                                          \
                }                                                                                              \
                if (lane == 0 && cs_ >= 0) out[cs_] = total;                                                   \
                c0 = n0;                                                                                       \
                c1 = n1;                                                                                       \
                n0 = kNegInf;                                                                                  \
                n1 = kNegInf;                                                                                  \
                cs_ = TAG.slot1;                                                                               \
            }                                                                                                  \
        }                                                                                                      \
        m0 = c0;                                                                                               \
        m1 = c1;                                                                                               \
    }

            float m0 = kNegInf, m1 = kNegInf;
            uint32_t cd0, cd1, cd2;
            uint4 ra0, rb0, ra1, rb1, ra2, rb2;
            ExactStepTag t0, t1, t2;
            CLB_REP8(CLB_ROW_DECL)
            CLB_XSTAGE_A(cd0, ra0, rb0, t0)
            CLB_XSTAGE_A(cd1, ra1, rb1, t1)
            CLB_ROW_LOAD8(A, cd0)
            while (t0.slot >= 0) {
                CLB_XSTAGE_A(cd2, ra2, rb2, t2)
                CLB_XSTAGE_C(A, ra0, rb0, t0, cd1)
                CLB_XSTAGE_A(cd0, ra0, rb0, t0)
                CLB_XSTAGE_C(A, ra1, rb1, t1, cd2)
                CLB_XSTAGE_A(cd1, ra1, rb1, t1)
                CLB_XSTAGE_C(A, ra2, rb2, t2, cd0)
            }
#undef CLB_IT_LOAD
#undef CLB_XSTAGE_A
#undef CLB_XSTAGE_C
        }
    }
#undef CLB_REP8
#undef CLB_ROW_DECL
#undef CLB_ROW_LOAD
#undef CLB_ROW_LOAD8
#undef CLB_ROW_STORE
#undef CLB_ROW_STORE8
}

// One radix-select step on a finished 256-bin histogram, by the first wave of the block: the digit d whose bin
// holds the rem-th largest key (bins above d hold fewer than rem keys, together with bin d at least rem), and the
// rank that remains inside that bin.  Lane L owns bins 4L..4L+3; a wave-wide suffix sum replaces the serial walk
// over 256 LDS words (which cost ~7 us per pass).  Requires 1 <= rem <= sum(hist).
__device__ __forceinline__ void radix_pick(const int* hist, int rem, uint32_t prefix, int shift,
                                           uint32_t* s_prefix, int* s_remaining) {
    const int lane = threadIdx.x & 63;
    const int4 h = *reinterpret_cast<const int4*>(hist + 4 * lane);
    const int own = h.x + h.y + h.z + h.w;
    int x = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_down(x, o, 64);
        if (lane + o < 64) x += y;
    }
    int above = x - own;                       // keys in bins above this lane's four
    const int hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        if (above < rem && rem <= above + hv[j]) {
            *s_prefix = prefix | ((uint32_t)(4 * lane + j) << shift);
            *s_remaining = rem - above;
        }
        above += hv[j];
    }
}

constexpr int kSelCache = 32;   // elements per thread kept in registers by the selection kernels

__device__ __forceinline__ void hist_add_aggregated(int* hist, uint32_t bin, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
#pragma unroll 1
    for (int it = 0; it < 4 && todo != 0; ++it) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        const uint32_t lb = __builtin_amdgcn_readlane(bin, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(active && bin == lb) & todo;
        if (__popcll(same) < 8) break;                 // spread-out bins: per-lane atomics are cheaper
        if (lane == leader) atomicAdd(&hist[lb], (int)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1);
}

// Radix select of the rem-th largest of the block's keys, written with the CLB_SEL_FOR_EACH macro of the calling
// kernel (which yields `valid` and `key` for every element the thread owns).  The bits on which the smallest and
// the largest key agree are skipped: MaxSim scores of one query share their exponent and leading mantissa bits, so
// a fixed MSB-first schedule spends its first passes on histograms with one occupied bin.  The first pass covers
// the (up to) eight bits below the highest differing bit -- well spread -- and later passes only touch the keys of
// the selected bin.  Leaves tau in *s_prefix and the rank inside the == tau group in *s_remaining.
// Needs: __shared__ hist[256] (16-byte aligned), s_prefix, s_remaining (= rem on entry), s_kmin, s_kmax.
// CLB_RADIX_SELECT_N(LIMIT): stop after LIMIT digit passes -- *s_prefix is then the LOWER EDGE of the bin the rem-th
// largest key lies in (its unresolved low bits zero), a lower bound of that key to 2^-8 (one pass) / 2^-16 (two passes)
// of the key range; the rank in *s_remaining is then the rank inside that bin.
#define CLB_RADIX_SELECT() CLB_RADIX_SELECT_N(99)
#define CLB_RADIX_SELECT_N(LIMIT)                                                                           \
    {                                                                                                       \
        uint32_t kmin_ = 0xffffffffu, kmax_ = 0u;                                                           \
        CLB_SEL_FOR_EACH(if (valid) { kmin_ = key < kmin_ ? key : kmin_; kmax_ = key > kmax_ ? key : kmax_; }) \
        _Pragma("unroll") for (int o_ = 32; o_ > 0; o_ >>= 1) {                                             \
            const uint32_t a_ = __shfl_xor(kmin_, o_, 64), b_ = __shfl_xor(kmax_, o_, 64);                  \
            kmin_ = a_ < kmin_ ? a_ : kmin_;                                                                \
            kmax_ = b_ > kmax_ ? b_ : kmax_;                                                                \
        }                                                                                                   \
        if (tid == 0) { s_kmin = 0xffffffffu; s_kmax = 0u; }                                                \
        __syncthreads();                                                                                    \
        if ((tid & 63) == 0) { atomicMin(&s_kmin, kmin_); atomicMax(&s_kmax, kmax_); }                      \
        __syncthreads();                                                                                    \
        const uint32_t diff_ = s_kmin ^ s_kmax;                                                             \
        if (diff_ == 0u) {                                                                                  \
            if (tid == 0) s_prefix = s_kmax;          /* all keys equal: rank unchanged */                  \
            __syncthreads();                                                                                \
        } else {                                                                                            \
            const int top_ = 31 - __clz((int)diff_);                                                        \
            int shift_ = top_ > 7 ? top_ - 7 : 0;                                                           \
            int width_ = top_ - shift_ + 1;                                                                 \
            if (tid == 0) s_prefix = top_ == 31 ? 0u : (s_kmax & (0xffffffffu << (top_ + 1)));              \
            int passes_ = 0;                                                                                \
            for (;;) {                                                                                      \
                if (tid < 256) hist[tid] = 0;                                                               \
                __syncthreads();                                                                            \
                const uint32_t prefix_ = s_prefix;                                                          \
                const uint32_t himask_ = shift_ + width_ >= 32 ? 0u : (0xffffffffu << (shift_ + width_));   \
                const uint32_t bmask_ = (1u << width_) - 1u;                                                \
                CLB_SEL_FOR_EACH(hist_add_aggregated(hist, (key >> shift_) & bmask_,                        \
                                                     valid && (key & himask_) == prefix_);)                 \
                __syncthreads();                                                                            \
                if (tid < 64) radix_pick(hist, s_remaining, prefix_, shift_, &s_prefix, &s_remaining);      \
                __syncthreads();                                                                            \
                if (shift_ == 0 || ++passes_ >= (LIMIT)) break;                                             \
                const int ns_ = shift_ > 8 ? shift_ - 8 : 0;                                                \
                width_ = shift_ - ns_;                                                                      \
                shift_ = ns_;                                                                               \
            }                                                                                               \
        }                                                                                                   \
    }

// -------------------------------------------------------------------------------------------------
// S7  indices = sortperm(scores, rev=true) ; first k     (searching.jl:125-127)
// The sort is stable, so ties keep ascending candidate order = ascending pid.  One workgroup per
// query: radix-select the k-th largest score, ordered compaction of {score > tau} plus the first
// (k - count_gt) of {score == tau}, bitonic sort of those k by (score desc, index asc).
// When `list` is given, only the listed candidate slots take part (two-pass mode).
// grid = B, block = 1024, LDS = 8 * next_pow2(k) + small.
// -------------------------------------------------------------------------------------------------
// Short lists (the two-pass mode: ~1.2 k re-scored passages per query for k = 1000) are ranked instead of sorted
// (topk_rank_kernel below); this kernel then leaves them alone.
constexpr int kRankMax = 4096;       // listed passages per query the rank kernel takes (32 KB of keys in LDS)
constexpr int kRankBlocks = 8;       // work-groups per query of the rank kernel

static __global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ scores,
                                                    const uint32_t* __restrict__ cand,
                                                    const int* __restrict__ ncand,
                                                    const int* __restrict__ list,
                                                    const int* __restrict__ nlist, int k, int kpow2,
                                                    size_t cand_cap, int64_t pid_offset,
                                                    int64_t* __restrict__ out_pids,
                                                    float* __restrict__ out_scores,
                                                    int* __restrict__ short_flag,
                                                    int64_t* __restrict__ n_cand_out /*optional*/,
                                                    int ranked_elsewhere = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];  // kpow2 entries
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ int sh_scan[32];
    __shared__ uint32_t s_prefix, s_kmin, s_kmax;
    __shared__ int s_remaining;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = list ? nlist[b] : ncand[b];
    if (ranked_elsewhere && list && n <= kRankMax) return;     // topk_rank_kernel has this query
    const float* sc = scores + (size_t)b * cand_cap;
    const int* lst = list ? list + (size_t)b * cand_cap : nullptr;
    const int keff = n < k ? n : k;
    // sort only as many slots as this query fills: a shard of a multi-GPU run often holds far fewer than k
    int kp = 1;
    while (kp < keff) kp <<= 1;
    kp = kp < kpow2 ? kp : kpow2;
    if (tid == 0) {
        short_flag[b] = n < k ? 1 : 0;
        if (n_cand_out) n_cand_out[b] = ncand[b];
    }

    // Thread t owns the contiguous elements [t*chunk, (t+1)*chunk) of the (listed) candidates; up to kSelCache of
    // them stay in registers as (order key, slot) for the four radix passes and the compaction, so the scores are
    // read from memory once.  Larger inputs (single-pass mode on a big shard) re-read them in every pass.
    const int chunk = (n + 1023) >> 10;
    const int i0 = tid * chunk;
    const bool cached = chunk <= kSelCache;   // uniform over the block
    uint32_t ckey[kSelCache];
    int cslot[kSelCache];
    if (cached) {
#pragma unroll
        for (int c = 0; c < kSelCache; ++c) {
            const int i = i0 + c;
            const bool valid = c < chunk && i < n;
            cslot[c] = valid ? (lst ? lst[i] : i) : 0;
            ckey[c] = valid ? f32_order_key(sc[cslot[c]]) : 0u;
        }
    }
#define CLB_SEL_FOR_EACH(...)                                                                   \
    if (cached) {                                                                               \
        _Pragma("unroll") for (int c = 0; c < kSelCache; ++c) {                                 \
            if (c >= chunk) break;                                                              \
            const bool valid = i0 + c < n;                                                      \
            const int slot = cslot[c];                                                          \
            const uint32_t key = ckey[c];                                                       \
            (void)slot;                                                                         \
            __VA_ARGS__                                                                         \
        }                                                                                       \
    } else {                                                                                    \
        for (int c = 0; c < chunk; ++c) {                                                       \
            const bool valid = i0 + c < n;                                                      \
            const int slot = valid ? (lst ? lst[i0 + c] : i0 + c) : 0;                          \
            const uint32_t key = valid ? f32_order_key(sc[slot]) : 0u;                          \
            (void)slot;                                                                         \
            __VA_ARGS__                                                                         \
        }                                                                                       \
    }

    // ---- radix select: tau = keff-th largest key -------------------------------------------------
    if (tid == 0) { s_prefix = 0u; s_remaining = keff; }
    __syncthreads();
    if (keff > 0) {
        if (cached) {
            CLB_RADIX_SELECT()
        } else {
            // too many keys for the registers: the radix passes do not care about order, so they read strided
            // (coalesced) instead of the contiguous chunks the ordered compaction below needs
#pragma push_macro("CLB_SEL_FOR_EACH")
#undef CLB_SEL_FOR_EACH
#define CLB_SEL_FOR_EACH(...)                                                                   \
    for (int c = 0; c < chunk; ++c) {                                                           \
        const int i_ = c * 1024 + tid;                                                          \
        const bool valid = i_ < n;                                                              \
        const uint32_t key = valid ? f32_order_key(sc[lst ? lst[i_] : i_]) : 0u;                \
        __VA_ARGS__                                                                             \
    }
            CLB_RADIX_SELECT()
#pragma pop_macro("CLB_SEL_FOR_EACH")
        }
    }
    const uint32_t tau = s_prefix;
    const int need_eq = s_remaining;  // how many of the == tau entries to take, lowest index first

    // ---- ordered compaction: one block-wide scan of the per-thread (#gt, #eq) counts ---------------
    for (int i = tid; i < kp; i += 1024) skeys[i] = 0ull;
    __syncthreads();
    if (keff > 0) {
        int ngt = 0, neq = 0;
        CLB_SEL_FOR_EACH(ngt += valid && key > tau; neq += valid && key == tau;)
        const int lane = tid & 63, wave = tid >> 6;
        int xg = ngt, xe = neq;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int yg = __shfl_up(xg, o, 64), ye = __shfl_up(xe, o, 64);
            if (lane >= o) { xg += yg; xe += ye; }
        }
        if (lane == 63) { sh_scan[wave] = xg; sh_scan[16 + wave] = xe; }
        __syncthreads();
        int gt_before = xg - ngt, eq_rank = xe - neq;
        for (int w2 = 0; w2 < wave; ++w2) { gt_before += sh_scan[w2]; eq_rank += sh_scan[16 + w2]; }
        CLB_SEL_FOR_EACH(
            if (valid) {
                const bool gt = key > tau, eq = key == tau;
                // position = (#gt before) + (#eq taken before) ; eq taken before = min(eq_rank, need_eq)
                if (gt || (eq && eq_rank < need_eq)) {
                    const int pos = gt_before + (eq_rank < need_eq ? eq_rank : need_eq);
                    // slot ascends with i, so ~slot makes lower index = larger key on equal scores
                    skeys[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (uint32_t)slot);
                }
                gt_before += gt;
                eq_rank += eq;
            })
    }
#undef CLB_SEL_FOR_EACH
    // ---- bitonic sort, descending -----------------------------------------------------------------
    // pair i of a step touches elements inside the 128-aligned tile of its wave whenever stride <= 64, so only the
    // wide strides need the block barrier (6 of the 55 steps at k = 1000)
    __syncthreads();   // the compaction above wrote skeys from arbitrary threads
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64 || (kp >> 1) > 1024) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
            for (int i = tid; i < (kp >> 1); i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = skeys[lo], c = skeys[hi];
                if ((a < c) == desc) { skeys[lo] = c; skeys[hi] = a; }
            }
        }
    }
    __syncthreads();
    const uint32_t* cnd = cand + (size_t)b * cand_cap;
    for (int i = tid; i < k; i += 1024) {
        int64_t pid = 0;
        float s = kNegInf;
        if (i < keff) {
            const unsigned long long key = skeys[i];
            const uint32_t slot = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
            pid = pid_offset + (int64_t)cnd[slot] + 1;
            s = sc[slot];
        }
        out_pids[(size_t)b * k + i] = pid;
        out_scores[(size_t)b * k + i] = s;
    }
}

// S7 for short lists: sortperm(scores, rev=true)[1:k] (searching.jl:125-127) by RANKING -- entry i goes to position
// #{ j : key_j > key_i } with key = (score order key, ~slot), so equal scores keep ascending candidate order = ascending
// pid exactly as the stable sort does, and no sorting network (55 barrier-separated steps for 1 024 keys, ~28 us on the
// one work-group a query used to get) is needed: n^2 comparisons spread over kRankBlocks work-groups per query.
// Every work-group holds all n keys in LDS; thread (e = tid & 255, quarter = tid >> 8) counts the keys of its quarter
// that beat element e of the work-group's share (a wave reads one key at a time: LDS broadcast), the four partial
// counts meet in LDS.  Queries with more than kRankMax listed passages (or no list: single-pass mode) are left to
// topk_kernel.  grid = (kRankBlocks, B), block = 1024.
static __global__ __launch_bounds__(1024) void topk_rank_kernel(const float* __restrict__ scores,
                                                                const uint32_t* __restrict__ cand,
                                                                const int* __restrict__ ncand,
                                                                const int* __restrict__ list,
                                                                const int* __restrict__ nlist, int k, size_t cand_cap,
                                                                int64_t pid_offset, int64_t* __restrict__ out_pids,
                                                                float* __restrict__ out_scores,
                                                                int* __restrict__ short_flag,
                                                                int64_t* __restrict__ n_cand_out /*optional*/) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[kRankMax];
    __shared__ int ranks[256];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int n = nlist[b];
    if (n > kRankMax) return;
    const float* sc = scores + (size_t)b * cand_cap;
    const int* lst = list + (size_t)b * cand_cap;
    const int keff = n < k ? n : k;
    if (g == 0) {
        if (tid == 0) {
            short_flag[b] = n < k ? 1 : 0;
            if (n_cand_out) n_cand_out[b] = ncand[b];
        }
        for (int i = keff + tid; i < k; i += 1024) {          // fewer than k candidates: (0, -Inf) padding
            out_pids[(size_t)b * k + i] = 0;
            out_scores[(size_t)b * k + i] = kNegInf;
        }
    }
    for (int i = tid; i < n; i += 1024) {
        const int slot = lst[i];
        keys[i] = ((unsigned long long)f32_order_key(sc[slot]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)slot);
    }
    __syncthreads();
    const uint32_t* cnd = cand + (size_t)b * cand_cap;
    const int e = tid & 255, quarter = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int q0 = (int)(((long long)n * quarter) >> 2), q1 = (int)(((long long)n * (quarter + 1)) >> 2);
    for (int base = g * 256; base < n; base += kRankBlocks * 256) {
        const int i = base + e;
        if (tid < 256) ranks[tid] = 0;
        __syncthreads();
        const unsigned long long mine = i < n ? keys[i] : ~0ull;
        int cnt = 0;
#pragma unroll 4
        for (int j = q0; j < q1; ++j) cnt += keys[j] > mine;
        if (cnt) atomicAdd(&ranks[e], cnt);
        __syncthreads();
        if (tid < 256 && i < n) {
            const int r = ranks[e];
            if (r < keff) {
                const uint32_t slot = 0xffffffffu - (uint32_t)(mine & 0xffffffffull);
                out_pids[(size_t)b * k + r] = pid_offset + (int64_t)cnd[slot] + 1;
                out_scores[(size_t)b * k + r] = sc[slot];
            }
        }
        __syncthreads();
    }
}

// ---- load-time helpers ---------------------------------------------------------------------------
// ivf (1-based embedding ids) -> local passage ids via binary search in doc_off (emb2pid, searching.jl:82-91)
static __global__ void ivf_to_pid_kernel(const int64_t* __restrict__ ivf, const uint32_t* __restrict__ doc_off,
                                  uint32_t* __restrict__ ivf_pid, int64_t n_emb, int n_docs,
                                  int* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_emb) return;
    const int64_t eid = ivf[i] - 1;
    if (eid < 0 || eid >= n_emb) { atomicOr(err, 1); ivf_pid[i] = 0; return; }
    int lo = 0, hi = n_docs;  // largest p with doc_off[p] <= eid
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)doc_off[mid] <= eid) lo = mid; else hi = mid;
    }
    ivf_pid[i] = (uint32_t)lo;
}

// *unsorted |= 4 when some IVF list does not hold non-decreasing passage ids (the reference's sortperm-built lists do).
// One wave per list.  grid = ceil(K / 4), block = 256.
static __global__ void ivf_lists_sorted_kernel(const uint32_t* __restrict__ ivf_off, const uint32_t* __restrict__ ivf_pid,
                                               int K, int* __restrict__ unsorted) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= K) return;
    const uint32_t lo = ivf_off[c], hi = ivf_off[c + 1];
    bool bad = false;
    for (uint32_t i = lo + 1 + lane; i < hi; i += 64) bad |= ivf_pid[i] < ivf_pid[i - 1];
    if (bad) atomicOr(unsorted, 4);
}

// Load-time reordering of every passage's embeddings by centroid code (MaxSim takes a maximum over a passage's
// embeddings, so their order inside the passage is free): key = local pid << 32 | code, value = embedding id.
static __global__ void passage_code_keys_kernel(const uint32_t* __restrict__ codes0, const uint32_t* __restrict__ doc_off,
                                                int64_t n_emb, int n_docs, unsigned long long* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_emb) return;
    int lo = 0, hi = n_docs;  // largest p with doc_off[p] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)doc_off[mid] <= e) lo = mid; else hi = mid;
    }
    keys[e] = ((unsigned long long)(uint32_t)lo << 32) | codes0[e];
    vals[e] = (uint32_t)e;
}
// new[e] = old[perm[e]] for the codes (one thread per embedding) and the residual rows (16-byte pieces)
static __global__ void permute_codes_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ src,
                                            uint32_t* __restrict__ dst, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = src[perm[e]];
}
static __global__ void permute_rows16_kernel(const uint32_t* __restrict__ perm, const uint4* __restrict__ src,
                                             uint4* __restrict__ dst, int64_t n, int pieces /* 16-byte pieces per row */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * pieces) return;
    const int64_t e = i / pieces;
    const int q = (int)(i % pieces);
    dst[i] = src[(size_t)perm[e] * pieces + q];
}

// codes: 1-based -> 0-based, range check (decompress's DomainError, residual.jl:766-768)
static __global__ void codes_to_zero_based_kernel(uint32_t* __restrict__ codes, int64_t n, uint32_t K,
                                           int* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = codes[i];
    if (c < 1u || c > K) { atomicOr(err, 2); codes[i] = 0; } else codes[i] = c - 1u;
}

// work counters of one batch: [0] candidate passages, [1] candidate embeddings, [2] passages in the
// exact re-score list, [3] their embeddings.  grid = (32, B), block = 256.
static __global__ void batch_stats_kernel(const uint32_t* __restrict__ cand, const int* __restrict__ ncand,
                                   const int* __restrict__ list, const int* __restrict__ nlist,
                                   const uint32_t* __restrict__ doc_off, size_t cand_cap,
                                   unsigned long long* __restrict__ stats,
                                   const unsigned long long* __restrict__ rowmask /*optional*/) {
    const int b = blockIdx.y;
    const uint32_t* cnd = cand + (size_t)b * cand_cap;
    unsigned long long embs = 0, lembs = 0;
    const int n = ncand[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = cnd[i];
        embs += doc_off[p + 1] - doc_off[p];
    }
    const int nl = list ? nlist[b] : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x) {
        const uint32_t p = cnd[list[(size_t)b * cand_cap + i]];
        const uint32_t len = doc_off[p + 1] - doc_off[p];
        if (rowmask && len <= (uint32_t)kMaxMaskedRows) {      // rows the exact kernel really multiplies
            const unsigned long long* mw = rowmask + ((size_t)b * cand_cap + i) * 4;
            lembs += __popcll(mw[0]) + __popcll(mw[1]) + __popcll(mw[2]) + __popcll(mw[3]);
        } else {
            lembs += len;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        embs += __shfl_down(embs, o, 64);
        lembs += __shfl_down(lembs, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (embs) atomicAdd(&stats[1], embs);
        if (lembs) atomicAdd(&stats[3], lembs);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&stats[0], (unsigned long long)n);
        atomicAdd(&stats[2], (unsigned long long)nl);
    }
}

// Cross-shard merge of sorted top-k lists (the final sortperm of search(), searching.jl:125-127, applied
// to the union of the shards' results).  in: [n_lists][B][k] records sorted by (score desc, pid asc),
// padded with (0, -Inf).  Every record finds its rank in the merged order by binary search in the other
// lists (pids are unique across shards, so ranks are unique).  grid = (ceil(n_lists*k/256), B).
__device__ __forceinline__ bool rec_before(float s, int64_t p, float s2, int64_t p2) {
    return s > s2 || (s == s2 && p < p2);
}
// pid_stride / score_stride: elements between the (B, k) blocks of consecutive lists (B*k when the lists are
// stacked densely; larger when every rank's pids and scores travel in one packed all-gather buffer).
static __global__ __launch_bounds__(256) void merge_topk_kernel(const int64_t* __restrict__ pids,
                                                         const float* __restrict__ scores, int k, int n_lists,
                                                         int B, size_t pid_stride, size_t score_stride,
                                                         int64_t* __restrict__ out_pids,
                                                         float* __restrict__ out_scores) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_lists * k) return;
    const int l = idx / k, i = idx % k;
    const int64_t p = pids[(size_t)l * pid_stride + (size_t)b * k + i];
    const float s = scores[(size_t)l * score_stride + (size_t)b * k + i];
    if (p <= 0) return;  // padding
    int rank = i;
    for (int l2 = 0; l2 < n_lists; ++l2) {
        if (l2 == l) continue;
        const int64_t* pl = pids + (size_t)l2 * pid_stride + (size_t)b * k;
        const float* sl = scores + (size_t)l2 * score_stride + (size_t)b * k;
        int lo = 0, hi = k;  // number of records of list l2 that come before (s, p)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int64_t p2 = pl[mid];
            const float s2 = sl[mid];
            if (p2 > 0 && rec_before(s2, p2, s, p)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (rank < k) {
        out_pids[(size_t)b * k + rank] = p;
        out_scores[(size_t)b * k + rank] = s;
    }
}
static __global__ void fill_pad_kernel(int64_t* __restrict__ pids, float* __restrict__ scores, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { pids[i] = 0; scores[i] = kNegInf; }
}

}  // namespace clb
