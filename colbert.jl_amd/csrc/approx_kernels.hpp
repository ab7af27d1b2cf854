// approx_kernels.hpp -- two-pass search, pass 1: approximate MaxSim on the 16-bit-input MFMA (fp16 operands since round 3; bf16 before) straight from the
// packed index, plus the selection of the candidates that must be re-scored exactly.
//
// Why: an exact fp32 score costs 8 192 flop per 36 packed bytes (227 flop/B; the fp32-MFMA ridge is ~25), so
// a single exact pass cannot come near the HBM roofline.  Pass 1 scores EVERY candidate passage approximately
// with a proven bound eps on |approx - canonical fp32 score|; tau = k-th largest approximate score; every
// candidate with approx >= tau - 2*eps is re-scored by score_exact_kernel.  Claim: that set contains the exact
// top-k.  Proof: k candidates have approx >= tau, hence exact >= tau - eps, so the k-th largest exact score
// theta >= tau - eps; a member of the exact top-k has exact >= theta, hence approx >= exact - eps >= tau - 2 eps.
// The final topk_kernel then runs on exact scores only, so the output (pids AND fp32 scores) is identical
// to the single-pass mode by construction; tests assert it bit-for-bit.
//
// The approximation (per query token t and embedding e, x = c + r the decompressed vector, den = ||x|| + eps32):
//     S[t][e] = Q_t . x / den  =  ( Q_t . c  +  Q_t . r ) * inv_norm[e]
//   * Q_t . c  = cells[t][code]  -- already computed exactly by centroid_scores_kernel (S1); stored for this
//                pass as fp16 rows [K][32 tokens] (64 B per centroid) to halve the gather;
//   * Q_t . r  -- r[d] = bucket_weight[idx[d]] takes only 2^nbits values: a 4-entry fp16 LUT applied with
//                v_perm_b32 to 2 dims at a time (selector built from the packed nibble with one u24 multiply),
//                fed as the A operand of v_mfma_f32_16x16x32_bf16 against fp16(Q) -- no centroid row is read,
//                nothing is normalised per element;
//   * inv_norm[e] = 1/(sqrtf(sumsq(c+r)) + eps32), precomputed once per index (fp32, canonical sumsq).
// Per embedding this pass reads 32 B residual + ONE 4-B word (code | quantised inv_norm) from HBM (streaming) and gathers one
// 64-B fp16 cells row from L2/MALL.  Algorithmic bytes (roofline accounting): 36 B per embedding.
//
// Error bound (u = 2^-24; qn = max_t ||Q_t||2; cn = max ||c||2; rn = sqrt(dim) * max|w| >= ||r||2;
// im = max inv_norm), per (t, e):
//   cells:   bf16x3 MFMA value vs canonical fp32 chain <= 1.25*7.4e-5*qn*cn       (centroid_top_bf16x3_kernel)
//            fp16 storage                            <= 2^-11 * qn*cn + 2^-25  (|cells| <= qn*cn < 65504: guarded in
//                                                       select_margin_kernel; 2^-25: values below the normal range)
//   Q.r:     fp16(Q), fp16(w)                        <= dq * rb + qn * dw_rn  (dq = max_t ||Q_t - fp16(Q_t)|| per query,
//                                                       rb = max ||fp16 residual vector||, dw_rn = sqrt(dim) * max |w - fp16(w)|
//                                                       per index: measured, not the generic 2^-9 relative bounds)
//            fp32 accumulation in the MFMA           <= 2*128*u*qn*rn
//   scaling: add, multiply, inv_norm rounding        <= 8*u*qn
//   oracle:  canonical fp32 value vs real arithmetic <= 320*u*qn               (x, sumsq, sqrt, divide, chain)
//   eps_t = im * (cells + Q.r terms) + 328*u*qn ;  eps = T * eps_t + 2*T*T*u*qn (the two token sums).
// eps is evaluated per query on the device (select_margin_kernel) and multiplied by kEpsSafety.
#pragma once
#include <hip/hip_fp16.h>

#include <algorithm>

#include "common.hpp"
#include "search_kernels.hpp"

namespace clb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr float kEpsSafety = 1.25f;

inline bool approx_supported(int dim, int nbits) { return dim == kDim && nbits == 2; }
// (+ 2 KB: the spill block the batched centroid kernel sends its out-of-range stores to)
inline size_t approx_cells_bytes(int64_t B, int64_t K, int64_t /*Tpad*/) { return (size_t)B * K * 16 * 4 + 2048; }

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return f32_to_bf16_rne(lo) | (f32_to_bf16_rne(hi) << 16);
}
// Two fp32 values -> the two operand words of the split-bf16 products: hi = RNE(x), lo = RNE(x - hi), value 0 in the low
// half of each word.  gfx950 converts a pair in one instruction (v_cvt_pk_bf16_f32, round to nearest even: the same
// results as f32_to_bf16_rne for finite inputs); the integer form above costs ~12 VALU operations per value, and every
// wave of the centroid kernels splits 256 query values before its first tile -- ~7 us of a 0.11-ms launch
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32pair{x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32pair{r0, r1}, bf16x2));
}
// fp16 (round to nearest even, subnormals kept: v_cvt_f16_f32) -- the operand format of pass 1's Q.r MFMAs since round 3:
// 11 significant bits where bf16 has 8, so the measured rounding terms of the error bound (dq, dw_rn) are 8x smaller;
// v_mfma_f32_32x32x16_f16 multiplies subnormal inputs exactly (tools/microbench/mfma_f16_denorm.hip, checked on MI355X)
__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half_rn(x)); }
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    const __half2 h = __floats2half2_rn(lo, hi);
    return *reinterpret_cast<const uint32_t*>(&h);
}

// NOTE: never feed an MFMA accumulator straight into this helper.  The wait states an MFMA result needs before a
// VALU read are inserted by the compiler only for instructions it can see through; through the inline asm a stale
// register is read (seen in group_max16: wrong centroids selected).  Arguments must come from ordinary VALU code.
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---- index-load: inv_norm[e] = 1/(sqrtf(sumsq(c + r)) + eps32), canonical sumsq; running max -------------------
static __global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ C,
                                                             const float* __restrict__ weights,
                                                             const uint32_t* __restrict__ codes0,
                                                             const uint8_t* __restrict__ residuals, int64_t n,
                                                             float* __restrict__ inv_norm,
                                                             unsigned int* __restrict__ inv_max_bits,
                                                             unsigned int* __restrict__ r2_max_bits,
                                                             unsigned int* __restrict__ inv_min_bits) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float w[4], wb2[4];      // wb2: squares of the fp16-rounded weights (the residual vector pass 1 multiplies)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w[j] = weights[j];
        const float wb = round_f16(weights[j]);
        wb2[j] = wb * wb;
    }
    const int64_t groups = (n + 15) / 16;
    float vmax = 0.f, r2max = 0.f, vmin = __builtin_inff();
    for (int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); grp < groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t el = grp * 16 + r;
        const int64_t e = el < n ? el : n - 1;
        uint32_t R[8];
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(residuals + (size_t)e * 32);
        uint4 v0 = *reinterpret_cast<const uint4*>(rp), v1 = *reinterpret_cast<const uint4*>(rp + 4);
        R[0] = v0.x; R[1] = v0.y; R[2] = v0.z; R[3] = v0.w; R[4] = v1.x; R[5] = v1.y; R[6] = v1.z; R[7] = v1.w;
        const float* cent = C + (size_t)codes0[e] * kDim + g;
        float p = 0.f, rr = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int bitpos = 8 * s;
            const uint32_t idx = (R[bitpos >> 5] >> ((bitpos & 31) + g * 2)) & 3u;
            const float lo = (idx & 1) ? w[1] : w[0];
            const float hi = (idx & 1) ? w[3] : w[2];
            const float x = cent[4 * s] + ((idx & 2) ? hi : lo);
            const float sq = x * x;
            p = p + sq;
            rr += (idx & 2) ? ((idx & 1) ? wb2[3] : wb2[2]) : ((idx & 1) ? wb2[1] : wb2[0]);
        }
        const float a = p + __shfl_xor(p, 16, 64);
        const float n2 = a + __shfl_xor(a, 32, 64);
        const float inv = 1.0f / (sqrtf(n2) + FLT_EPSILON);
        if (g == 0 && el < n) inv_norm[e] = inv;
        vmax = fmaxf(vmax, inv);
        vmin = fminf(vmin, inv);
        rr += __shfl_xor(rr, 16, 64);
        rr += __shfl_xor(rr, 32, 64);
        r2max = fmaxf(r2max, rr);
    }
    for (int o = 32; o > 0; o >>= 1) {
        vmax = fmaxf(vmax, __shfl_down(vmax, o, 64));
        vmin = fminf(vmin, __shfl_down(vmin, o, 64));
        r2max = fmaxf(r2max, __shfl_down(r2max, o, 64));
    }
    if (lane == 0) {
        atomicMax(inv_max_bits, __float_as_uint(vmax));
        atomicMin(inv_min_bits, __float_as_uint(vmin));      // positive floats order like their bit patterns
        atomicMax(r2_max_bits, __float_as_uint(r2max));
    }
}

// Pass 1 streams ONE word per embedding: the 0-based centroid code in the low `cbits` bits and inv_norm quantised to
// the remaining bits, inv' = inv_lo + q * step (|inv' - inv| <= step / 2, accounted for in the error bound).  36 B per
// embedding are then streamed -- exactly the algorithmic bytes -- instead of 40, with one load less per step.
static __global__ void pack_code_inv_kernel(const uint32_t* __restrict__ codes0, const float* __restrict__ inv_norm,
                                            int64_t n, int cbits, float inv_lo, float inv_step_rcp, uint32_t qmax,
                                            uint32_t* __restrict__ codeinv) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float qf = rintf((inv_norm[e] - inv_lo) * inv_step_rcp);
    qf = fminf(fmaxf(qf, 0.f), (float)qmax);
    codeinv[e] = codes0[e] | ((uint32_t)qf << cbits);
}

// -------------------------------------------------------------------------------------------------------------
// S1 + S2 at the bf16-MFMA rate with exact selection ("bf16x3"): every fp32 operand is split x = hi + lo + d,
// |d| <= 2^-18 |x| (hi, lo bf16), and Q.c ~= Qh.Ch + Qh.Cl + Ql.Ch on v_mfma_f32_32x32x16_bf16 (24 MFMAs of 32
// cycles per 32x32 tile instead of 64 fp32 MFMAs of 64 cycles).  Bound on |approx - canonical fp32 chain|:
//   split (3 dropped terms)  3 * 2^-18 * qn*cn ;  fp32 accumulation of 384 products  2*384*u*qn*cn ;
//   canonical chain vs real arithmetic  2*128*u*qn*cn      =>  eps_c = kEpsSafety * 7.4e-5 * qn * cn.
// Per token the 8 best approximate centroids are kept; with a2 = 2nd best approximate score every centroid whose
// approximate score is >= a2 - 2 eps_c is re-scored with the canonical fp32 fmaf chain (top_refine_kernel) and the
// exact top-nprobe by (score desc, index asc) comes out -- the same ids as the fp32 kernel, bit for bit.  If more
// than 8 centroids could qualify the query is flagged and redone by the fp32 kernel (never seen in practice).
// -------------------------------------------------------------------------------------------------------------
constexpr int kRowBytes16 = 272;   // staged bf16 row: 256 B + 16 B pad -> conflict-free ds_read_b128 per 16-lane group
constexpr int kTopPartial = 4;     // per-lane list length in the bf16x3 kernel
constexpr int kTopRefine = 4;      // groups of 16 centroids re-scored exactly per ROUND of the refine wave (one per 16 lanes)
constexpr int kTopRefineCap = 256; // qualifying groups a token may have before the wave scans all K centroids (round 5: was
                                   // kTopRefine -- on an index with near-degenerate centroids, what a random-weight encoder
                                   // gives, five qualifying groups already cost a ~2-ms scan per token)

static __global__ void split_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi,
                                         uint16_t* __restrict__ lo, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const uint32_t h = f32_to_bf16_rne(v);
    const float rem = v - __uint_as_float(h << 16);
    hi[i] = (uint16_t)h;
    lo[i] = (uint16_t)f32_to_bf16_rne(rem);
}

// Round 5, "f16x1": the score table of the batched centroid stage from ONE fp16 product per fp32 product.  The table is stored
// as fp16 anyway (relative rounding 2^-11: 4.9e-4 |score|, the largest term of e_cells), so the three-product bf16 split
// (7.4e-5 qn cn) buys the TABLE nothing; what the product must not lose is bounded by the measured conversion errors:
//   |Q.c - Q'.c'| <= ||Q - Q'|| ||c'|| + ||Q|| ||c - c'||  =  dq * cn16 + qn * dc      (Q', c' the fp16-rounded operands;
// their products are exact in fp32, the accumulation adds 2*128*u*qn*cn) -- dq per query (select_margin_kernel measures it
// for pass 1's operand already), dc = max_c ||c - fp16(c)|| per index (max_row_f16_err_kernel).  A third of the MFMAs:
// centroid_top_bf16x3_teams_kernel<true>, 0.110 -> 0.090 ms per 32 queries x 131 072 centroids even with the lo plane still staged.
// |approximate - canonical| of one centroid score, whichever centroid kernel produced it: the three-product bf16 split
// (7.4e-5 qn cn) or one fp16 product (measured conversion errors + fp32 accumulation; dc = 0 disables the term)
__device__ __forceinline__ float centroid_product_bound(float qn, float dq, float cn, float dc) {
    const float b3 = 7.4e-5f * qn * cn;
    const float b1 = dc > 0.f ? 1.001f * (dq * cn * 1.0005f + qn * dc) + 384.f * 5.9604645e-08f * qn * cn : 0.f;
    return fmaxf(b3, b1);
}
static __global__ void to_f16_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const __half h = __float2half_rn(x[i]);
    out[i] = *reinterpret_cast<const uint16_t*>(&h);
}
// max over the rows of ||c - fp16(c)|| (a component beyond the fp16 range makes it infinite: such an index is searched exactly)
static __global__ __launch_bounds__(256) void max_row_f16_err_kernel(const float* __restrict__ C, int K, unsigned int* __restrict__ out_bits) {
    const int lane = threadIdx.x & 63;
    float m = 1e-30f;                                 // never exactly 0: 0 is the callers' "single-product kernel not in use"
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < K; c += gridDim.x * 4) {
        const float a = C[(size_t)c * kDim + lane], b = C[(size_t)c * kDim + 64 + lane];
        const float da = fabsf(a) < 6.0e4f ? a - round_f16(a) : __builtin_inff(), db = fabsf(b) < 6.0e4f ? b - round_f16(b) : __builtin_inff();
        float q = da * da + db * db;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        m = fmaxf(m, sqrtf(q) * 1.0001f);
    }
    if (lane == 0) atomicMax(out_bits, __float_as_uint(m));
}

// -------------------------------------------------------------------------------------------------------------
// Round 6, "8-bit score rows": the score table pass 1 gathers from as 32-BYTE rows [K][32 tokens] of 8-bit cells instead of
// 64-byte fp16 rows -- half the bytes per gathered row, half the table per query (4 MB at K = 131 072: one XCD's L2).  On an
// index whose codes are not id-adjacent (uniform codes, any k-means-built index) the row gather is the largest term of pass 1
// (profiles/r05_pass1_ablations.jsonl: -0.61 of 3.46 ms on the built 1 M index).
// The cells of a (query, token) are LINEAR in the score over the range [lo_t, hi_t] the token's K scores REALLY span:
//     step_t = (hi_t - lo_t) / 254 ;   k_t = lo_t / step_t ;   cell = clamp(rint(score / step_t - k_t), 0, 255) ;
//     score ~ (cell + k_t) * step_t ,   off by at most half a step (+ the rounding of the operations: 0.5005 step_t in query_bound)
// A first form took the range from norms alone, +-||Q_t|| max||c|| (profiles/r06_score_rows_apriori_range_ab.jsonl): on a
// k-means-built index max||c|| is an outlier (~1 where the typical centroid has norm 0.35) and the a-priori step came out 5x
// the one below -- pass 2 re-scored 3.6x the rows and the search got slower.  The measured range is only known after the
// last centroid, so the batched centroid kernel keeps writing its fp16 table and tracks every token's running minimum and
// maximum on the way (centroid_top_bf16x3_teams_kernel: `rangep`); token_range_kernel reduces the work-groups' partial
// ranges and requantise_cells_kernel rewrites the table as 8-bit rows (268 MB read, 134 MB written per 32 queries at
// K = 131 072: ~0.08 ms, which the halved gather has to earn back -- the format is chosen per index, search.hip).
// tscale[b][t] = {step_t, 1 / step_t, k_t, A_t = max |score|}, t < 32.  One writer: the requantisation, pass 1, the row
// sweep and the bound all READ these words, so every stage scales by the same bits.  The fp16 format uses A_t alone: its
// storage error is 2^-11 A_t, where the bound used to assume |score| <= ||Q_t|| max||c||.
// -------------------------------------------------------------------------------------------------------------
constexpr float kCell8Steps = 254.0f;               // cells 0 .. 254 span [lo_t, hi_t]; 255 absorbs the rounding of k_t
// grid = B, block = 1024 (32 threads per token).  rangep: [b][t][nparts] {min, max} as left by the centroid kernel.
static __global__ __launch_bounds__(1024) void token_range_kernel(const float2* __restrict__ rangep, int nparts, int T,
                                                                 float4* __restrict__ tscale) {
    const int b = blockIdx.x, t = threadIdx.x >> 5, part = threadIdx.x & 31;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    const float2* rp = rangep + ((size_t)b * 32 + t) * nparts;
    for (int i = part; i < nparts; i += 32) {
        const float2 v = rp[i];
        lo = fminf(lo, v.x);
        hi = fmaxf(hi, v.y);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if (part == 0) {
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        const float amax = fmaxf(fabsf(lo), fabsf(hi));
        // a range that is not a pair of finite numbers (a NaN / infinite query): zeros -- query_bound declares the query unsafe
        if (t < T && hi >= lo && amax < 1.0e30f) {
            // never a step below 1e-6 of the magnitude: k_t stays a number whose sum with a cell is exact to 1/16 cell
            const float step = fmaxf(fmaxf((hi - lo) * (1.0f / kCell8Steps), amax * 1.0e-6f), 1.0e-30f);
            const float rstep = 1.0f / step;
            out = make_float4(step, rstep, lo * rstep, amax);
        }
        tscale[(size_t)b * 32 + t] = out;
    }
}

// fp16 rows [K][32 tokens] (64 B) -> 8-bit rows [K][32 tokens] (32 B), per query.  Thread = 16 tokens of one centroid: two
// 16-byte loads, one 16-byte store.  grid = (blocks, B), block = 256.
static __global__ __launch_bounds__(256) void requantise_cells_kernel(const uint32_t* __restrict__ cells16,
                                                                     const float4* __restrict__ tscale,
                                                                     uint32_t* __restrict__ cells8, int K) {
    __shared__ float2 sc[32];                        // {1 / step_t, k_t}
    const int b = blockIdx.y;
    if (threadIdx.x < 32) {
        const float4 ts = tscale[(size_t)b * 32 + threadIdx.x];
        sc[threadIdx.x] = make_float2(ts.y, ts.z);
    }
    __syncthreads();
    const int hsel = threadIdx.x & 1;                // tokens 16 hsel .. 16 hsel + 15
    const u32x4* src = reinterpret_cast<const u32x4*>(cells16 + (size_t)b * K * 16);
    u32x4* dst = reinterpret_cast<u32x4*>(cells8 + (size_t)b * K * 8);
    const int64_t total = (int64_t)K * 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const u32x4 a = __builtin_nontemporal_load(src + 2 * i), c = __builtin_nontemporal_load(src + 2 * i + 1);
        const uint32_t w[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t hbits = (w[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            const __half hv = *reinterpret_cast<const __half*>(&hbits);
            const float2 s2 = sc[16 * hsel + j];
            o[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(rintf(fmaf(__half2float(hv), s2.x, -s2.y)), (uint32_t)(j & 3), o[j >> 2]);
        }
        dst[i] = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// Insert into a lane's descending list of approximate scores.  Ordering is by value only: which of several EQUAL
// approximate scores survives at the end of a list is irrelevant, because top_refine_kernel re-scores everything above
// its cut and treats a list whose last entry reaches the cut as overflowed.  Branch-free shift (3 compares, 14
// selects) behind a wave-level reject.
template <int NP>
__device__ __forceinline__ void topn_insert_lazy(float (&bv)[NP], int (&bi)[NP], float v, int idx) {
    if (__builtin_amdgcn_ballot_w64(v > bv[NP - 1]) == 0) return;
    const bool in = v > bv[NP - 1];
#pragma unroll
    for (int p = NP - 1; p > 0; --p) {
        const bool above = v > bv[p - 1];              // v belongs above slot p: slot p takes its upper neighbour
        const float nv = above ? bv[p - 1] : v;
        const int ni = above ? bi[p - 1] : idx;
        const bool touch = in && v > bv[p];            // slots below v's position are unchanged
        bv[p] = touch ? nv : bv[p];
        bi[p] = touch ? ni : bi[p];
    }
    if (in && v > bv[0]) { bv[0] = v; bi[0] = idx; }
}

// Largest of the 16 scores a lane holds of one 32-centroid tile (its "group": token i, centroids
// c0 + (r & 3) + 8 (r >> 2) + 4 h).  The per-lane lists keep the best GROUPS by this maximum -- one list operation
// per tile instead of sixteen; top_refine_kernel re-scores all 16 centroids of every group that can matter.
__device__ __forceinline__ float group_max16(const f32x16& acc, int c0, int h, int K) {
    if (__builtin_expect(c0 + 32 > K, 0)) {         // last, partial tile: rows past K are copies of row K-1
        float m = kNegInf;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, c0 + (r & 3) + 8 * (r >> 2) + 4 * h < K ? acc[r] : kNegInf);
        return m;
    }
    // plain fmaxf (the compiler fuses them to v_max3_f32): the accumulator comes straight from the MFMA, and the
    // wait states that hazard needs are only inserted for instructions the compiler can see through -- an inline
    // asm v_max3 here read stale registers
    float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
    m = fmaxf(fmaxf(m, acc[3]), acc[4]);
    m = fmaxf(fmaxf(m, acc[5]), acc[6]);
    m = fmaxf(fmaxf(m, acc[7]), acc[8]);
    m = fmaxf(fmaxf(m, acc[9]), acc[10]);
    m = fmaxf(fmaxf(m, acc[11]), acc[12]);
    m = fmaxf(fmaxf(m, acc[13]), acc[14]);
    return fmaxf(m, acc[15]);
}

// smallest of the 16 scores a lane holds of one tile (plain fminf: see group_max16 about the accumulator hazard)
__device__ __forceinline__ float group_min16(const f32x16& acc) {
    float m = fminf(fminf(acc[0], acc[1]), acc[2]);
    m = fminf(fminf(m, acc[3]), acc[4]);
    m = fminf(fminf(m, acc[5]), acc[6]);
    m = fminf(fminf(m, acc[7]), acc[8]);
    m = fminf(fminf(m, acc[9]), acc[10]);
    m = fminf(fminf(m, acc[11]), acc[12]);
    m = fminf(fminf(m, acc[13]), acc[14]);
    return fminf(m, acc[15]);
}

// grid = (gx, B), block = 128 (2 waves), LDS = 2 waves * (2 arrays * 32 rows * 272 B + a 2-KB patch with WRITE_HALF).
// partial: [B][32 tokens][nslots][kTopPartial], nslots = waves * 2 halves.
template <bool WRITE_HALF>
static __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void centroid_top_bf16x3_kernel(
    const uint16_t* __restrict__ Chi, const uint16_t* __restrict__ Clo, const float* __restrict__ Q,
    ValIdx* __restrict__ partial, uint32_t* __restrict__ cells16, int K, int T, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    unsigned char* my = lds16 + wave * (2 * 32 * kRowBytes16);
    // B operand: token i, dims 64h + 8s + j (the k index is a permutation of the dims, same for A)
    u32x4 qh[8], ql[8];
    {
        const float* qrow = Q + ((size_t)b * T + (i < T ? i : T - 1)) * kDim + 64 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
            const float4 a = *reinterpret_cast<const float4*>(qrow + 8 * s);
            const float4 c = *reinterpret_cast<const float4*>(qrow + 8 * s + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                split_bf16_pair(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
                if (i >= T) { hh[j] = 0u; ll[j] = 0u; }      // tokens past T: zero operand rows
            }
            qh[s] = u32x4{hh[0], hh[1], hh[2], hh[3]};
            ql[s] = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
    }
    float bv[kTopPartial];
    int bi[kTopPartial];
#pragma unroll
    for (int p = 0; p < kTopPartial; ++p) { bv[p] = kNegInf; bi[p] = 0x7fffffff; }
    const int waves_total = gridDim.x * 2;
    int tile = blockIdx.x * 2 + wave;
    const int prow = lane >> 4, pchunk = lane & 15;   // loader: instruction m covers rows 4m..4m+3, 16 chunks each
#define CLB_REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
    // WRITE_HALF: the sixteen tile loads are issued and waited for by hand, as in the batched kernels below -- behind them
    // are exactly the two table stores of a tile, so the next tile is waited for with vmcnt(2) (at the loop bottom: the
    // last, clamped request has then landed too, before its destination registers die) instead of the vmcnt(0) hipcc
    // places at the loop top when stores sit in branches, which also waits for the write acknowledgements
#define CLB_PF_DECL(m) u32x4 ph##m, pl##m;
#define CLB_PF_LOAD(m)                                                                        \
    {                                                                                         \
        int c = tl * 32 + 4 * m + prow;                                                       \
        c = c < K ? c : K - 1;                                                                \
        const uint16_t* a0_ = Chi + (size_t)c * kDim + 8 * pchunk;                            \
        const uint16_t* a1_ = Clo + (size_t)c * kDim + 8 * pchunk;                            \
        if (WRITE_HALF) {                                                                     \
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"  \
                         : "=&v"(ph##m), "=&v"(pl##m) : "v"(a0_), "v"(a1_) : "memory");       \
        } else {                                                                              \
            ph##m = *reinterpret_cast<const u32x4*>(a0_);                                     \
            pl##m = *reinterpret_cast<const u32x4*>(a1_);                                     \
        }                                                                                     \
    }
#define CLB_PF_TIE(m) , "+v"(ph##m), "+v"(pl##m)
#define CLB_PF_WAIT(N)                                                                        \
    if (WRITE_HALF) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ph0), "+v"(pl0) CLB_PF_TIE(1) CLB_PF_TIE(2) CLB_PF_TIE(3) \
                                 CLB_PF_TIE(4) CLB_PF_TIE(5) CLB_PF_TIE(6) CLB_PF_TIE(7) :: "memory");
#define CLB_PF_STORE(m)                                                                                          \
    *reinterpret_cast<u32x4*>(my + (4 * m + prow) * kRowBytes16 + 16 * pchunk) = ph##m;                          \
    *reinterpret_cast<u32x4*>(my + 32 * kRowBytes16 + (4 * m + prow) * kRowBytes16 + 16 * pchunk) = pl##m;
    CLB_REP8(CLB_PF_DECL)
    {
        const int tl = tile < n_tiles ? tile : n_tiles - 1;
        CLB_REP8(CLB_PF_LOAD)
    }
    CLB_PF_WAIT(0)
    // the wave's 2-KB transposition patch (behind the two waves' tile buffers) and the spill block behind the table
    unsigned char* patch = lds16 + 2 * (2 * 32 * kRowBytes16) + wave * 2048;
    unsigned char* spill = reinterpret_cast<unsigned char*>(cells16 + (size_t)gridDim.y * K * 16);
    while (tile < n_tiles) {
        const int c0 = tile * 32;
        CLB_REP8(CLB_PF_STORE)
        const int next = tile + waves_total;
        {
            const int tl = next < n_tiles ? next : n_tiles - 1;
            CLB_REP8(CLB_PF_LOAD)
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const u32x4 ah = *reinterpret_cast<const u32x4*>(my + i * kRowBytes16 + 16 * (8 * h + s));
            const u32x4 al = *reinterpret_cast<const u32x4*>(my + 32 * kRowBytes16 + i * kRowBytes16 + 16 * (8 * h + s));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, qh[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, ql[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, qh[s]), acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        topn_insert_lazy<kTopPartial>(bv, bi, group_max16(acc, c0, h, K), 2 * tile + h);
        if (WRITE_HALF) {
            // the tile's 32 x 32 scores leave as one 2-KB block of fp16 rows [centroid][token], transposed through the
            // wave's LDS patch: two full-width stores (rows past K to the spill block) instead of sixteen dword stores
            // in branches -- a store instruction costs the CU's texture-address unit ~70 cycles whatever its width
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
                *reinterpret_cast<__half*>(patch + cl * 64 + i * 2) = __float2half_rn(acc[r]);
            }
            __builtin_amdgcn_wave_barrier();
            unsigned char* dst = reinterpret_cast<unsigned char*>(cells16 + ((size_t)b * K + c0) * 16);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const int off = part * 1024 + lane * 16;   // centroid c0 + off / 64
                unsigned char* to = c0 + (off >> 6) < K ? dst + off : spill + off;
                *reinterpret_cast<uint4*>(to) = *reinterpret_cast<const uint4*>(patch + off);
            }
        }
        CLB_PF_WAIT(2)
        tile = next;
    }
#undef CLB_REP8
#undef CLB_PF_DECL
#undef CLB_PF_LOAD
#undef CLB_PF_TIE
#undef CLB_PF_WAIT
#undef CLB_PF_STORE
    const int slot = (blockIdx.x * 2 + wave) * 2 + h;
    const int nslots = gridDim.x * 4;
    ValIdx* out = partial + (((size_t)b * 32 + i) * nslots + slot) * kTopPartial;     // [query][token][slot][entry]
#pragma unroll
    for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv[p], bi[p]};
}

// Batched variant (B >= 8).  The kernel above re-reads the whole split centroid table once per query (64 MB x B
// through L2: 2 GB per 32-query batch, which held it at the L2/HBM rate instead of the MFMA rate).  Here a
// work-group of 4 waves stages each 32-centroid tile (hi and lo halves) ONCE in LDS, double-buffered, and every
// wave scores it against its own two queries, whose split bf16 operands stay in registers: 8 queries per tile
// load, the same three MFMA products per accumulator in the same order (so the error bound is unchanged).
// grid = (gx, ceil(B / 8)), block = 256, LDS = 2 buffers * 2 arrays * 32 rows * 272 B + 4 waves * 2 KB.
// partial: [B][32 tokens][nslots = gx * 2 halves][kTopPartial].
constexpr int kMqQueries = 8;

// BIAS (index build): `bias[c]` (padded to a multiple of 32 entries) is added to every score of centroid c before
// the group maximum is taken -- with bias = -||c||^2 / 2 the ranking is that of the k-means distance
// (nearest_centroids in codec.hip).  q_rows = number of valid rows of Q (rows past it re-read the last one; for the
// search path q_rows = B*T).
template <bool WRITE_HALF, bool BIAS = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void centroid_top_bf16x3_mq_kernel(
    const uint16_t* __restrict__ Chi, const uint16_t* __restrict__ Clo, const float* __restrict__ Q,
    ValIdx* __restrict__ partial, uint32_t* __restrict__ cells16, int K, int T, int B, int n_tiles,
    const float* __restrict__ bias = nullptr, int64_t q_rows = -1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int bq0 = blockIdx.y * kMqQueries + wave * 2;   // this wave's queries: bq0, bq0 + 1
    u32x4 qh[2][8], ql[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = bq0 + q < B ? bq0 + q : B - 1;      // past the batch: a duplicate whose results are dropped
        int64_t qr = (int64_t)b * T + (i < T ? i : T - 1);
        if (q_rows >= 0 && qr >= q_rows) qr = q_rows - 1;
        const float* qrow = Q + (size_t)qr * kDim + 64 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
            const float4 a = *reinterpret_cast<const float4*>(qrow + 8 * s);
            const float4 c = *reinterpret_cast<const float4*>(qrow + 8 * s + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                split_bf16_pair(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
                if (i >= T) { hh[j] = 0u; ll[j] = 0u; }      // tokens past T: zero operand rows
            }
            qh[q][s] = u32x4{hh[0], hh[1], hh[2], hh[3]};
            ql[q][s] = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
    }
    float bv0[kTopPartial], bv1[kTopPartial];
    int bi0[kTopPartial], bi1[kTopPartial];
#pragma unroll
    for (int p = 0; p < kTopPartial; ++p) { bv0[p] = bv1[p] = kNegInf; bi0[p] = bi1[p] = 0x7fffffff; }
    // loader: thread tid moves 16-B chunk (tid & 15) of rows (tid >> 4) and 16 + (tid >> 4), hi and lo
    const int prow = threadIdx.x >> 4, pchunk = threadIdx.x & 15;
    u32x4 ph0, ph1, pl0, pl1;
#define CLB_MQ_LOAD(TL)                                                                                   \
    {                                                                                                     \
        int c0_ = (TL) * 32 + prow, c1_ = c0_ + 16;                                                       \
        c0_ = c0_ < K ? c0_ : K - 1;                                                                      \
        c1_ = c1_ < K ? c1_ : K - 1;                                                                      \
        const uint16_t* a0_ = Chi + (size_t)c0_ * kDim + 8 * pchunk;                                       \
        const uint16_t* a1_ = Clo + (size_t)c0_ * kDim + 8 * pchunk;                                       \
        const uint16_t* a2_ = Chi + (size_t)c1_ * kDim + 8 * pchunk;                                       \
        const uint16_t* a3_ = Clo + (size_t)c1_ * kDim + 8 * pchunk;                                       \
        if (WRITE_HALF) {   /* hand-issued and hand-waited (CLB_MQ_WAIT): see the stores at the end of the loop */ \
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"           \
                         "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"                \
                         : "=&v"(ph0), "=&v"(pl0), "=&v"(ph1), "=&v"(pl1)                                   \
                         : "v"(a0_), "v"(a1_), "v"(a2_), "v"(a3_) : "memory");                             \
        } else {                                                                                          \
            ph0 = *reinterpret_cast<const u32x4*>(a0_);                                                   \
            pl0 = *reinterpret_cast<const u32x4*>(a1_);                                                   \
            ph1 = *reinterpret_cast<const u32x4*>(a2_);                                                   \
            pl1 = *reinterpret_cast<const u32x4*>(a3_);                                                   \
        }                                                                                                 \
    }
    // the four tile loads have landed once at most N younger vector-memory operations are outstanding
#define CLB_MQ_WAIT(N)                                                                                    \
    if (WRITE_HALF) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ph0), "+v"(pl0), "+v"(ph1), "+v"(pl1) :: "memory");
    int tile = blockIdx.x;
    CLB_MQ_LOAD(tile < n_tiles ? tile : n_tiles - 1)
    CLB_MQ_WAIT(0)
    int buf = 0;
    while (tile < n_tiles) {            // uniform over the work-group: tile depends on blockIdx only
        unsigned char* my = lds16 + buf * (2 * 32 * kRowBytes16);
        *reinterpret_cast<u32x4*>(my + prow * kRowBytes16 + 16 * pchunk) = ph0;
        *reinterpret_cast<u32x4*>(my + (16 + prow) * kRowBytes16 + 16 * pchunk) = ph1;
        *reinterpret_cast<u32x4*>(my + (32 + prow) * kRowBytes16 + 16 * pchunk) = pl0;
        *reinterpret_cast<u32x4*>(my + (48 + prow) * kRowBytes16 + 16 * pchunk) = pl1;
        const int next = tile + gridDim.x;
        CLB_MQ_LOAD(next < n_tiles ? next : n_tiles - 1)
        // one barrier per tile: the buffer written now was last read two iterations ago, before the previous barrier
        __syncthreads();
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(my + i * kRowBytes16 + 16 * (8 * h + s)));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(my + (32 + i) * kRowBytes16 + 16 * (8 * h + s)));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(bf16x8, qh[0][s]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(bf16x8, qh[1][s]), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, ql[0][s]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, ql[1][s]), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, qh[0][s]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, qh[1][s]), acc1, 0, 0, 0);
        }
        const int c0 = tile * 32;
        if (BIAS) {
            // this lane's rows: c0 + (r & 3) + 8 (r >> 2) + 4 h -- four aligned float4 of the bias row
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + c0 + 8 * qd + 4 * h);
                acc0[4 * qd] += b4.x; acc0[4 * qd + 1] += b4.y; acc0[4 * qd + 2] += b4.z; acc0[4 * qd + 3] += b4.w;
                acc1[4 * qd] += b4.x; acc1[4 * qd + 1] += b4.y; acc1[4 * qd + 2] += b4.z; acc1[4 * qd + 3] += b4.w;
            }
        }
        topn_insert_lazy<kTopPartial>(bv0, bi0, group_max16(acc0, c0, h, K), 2 * tile + h);
        topn_insert_lazy<kTopPartial>(bv1, bi1, group_max16(acc1, c0, h, K), 2 * tile + h);
        if (WRITE_HALF) {
            // the tile's 32 x 32 scores leave as one contiguous 2-KB block of fp16 rows [centroid][token]: transposed through
            // a per-wave LDS patch (16 ds_write_b16 + 2 ds_read_b128) so that the wave issues 2 full-width stores
            // instead of 16 quarter-filled ones
            unsigned char* patch = lds16 + 2 * (2 * 32 * kRowBytes16) + wave * 2048;
            const int pos = i;                                 // halfword of token i inside its centroid's 64 B
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
                    *reinterpret_cast<__half*>(patch + cl * 64 + pos * 2) = __float2half_rn(q ? acc1[r] : acc0[r]);
                }
                __builtin_amdgcn_wave_barrier();
                const int b = bq0 + q;
                unsigned char* dst = reinterpret_cast<unsigned char*>(cells16 + ((size_t)b * K + c0) * 16);
                // UNCONDITIONAL stores: rows past K and queries past B go to the 2-KB spill block behind the table
                // (approx_cells_bytes) instead of being branched around, so that every iteration issues exactly four
                // stores behind the next tile's four loads.  Left to hipcc, the loop waits with vmcnt(0) at its top
                // -- for the write acknowledgements of the tile just stored, not only for the prefetched loads (it
                // cannot count stores inside branches, and merges the loop entry, where nothing is younger than the
                // loads).  The loads are therefore issued by hand (CLB_MQ_LOAD) and waited for with vmcnt(4) at the
                // bottom of the loop: the stores drain under the next tile's MFMAs (0.1205 -> 0.112 ms per batch of 32)
                unsigned char* spill = reinterpret_cast<unsigned char*>(cells16 + (size_t)B * K * 16);
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int off = part * 1024 + lane * 16;   // centroid c0 + off / 64
                    unsigned char* to = (b < B && c0 + (off >> 6) < K) ? dst + off : spill + off;
                    *reinterpret_cast<uint4*>(to) = *reinterpret_cast<const uint4*>(patch + off);
                }
            }
        }
        CLB_MQ_WAIT(4)
        tile = next;
        buf ^= 1;
    }
#undef CLB_MQ_LOAD
#undef CLB_MQ_WAIT
    const int slot = blockIdx.x * 2 + h;
    const int nslots = gridDim.x * 2;
    if (bq0 < B) {
        ValIdx* out = partial + (((size_t)bq0 * 32 + i) * nslots + slot) * kTopPartial;   // [query][token][slot][entry]
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv0[p], bi0[p]};
    }
    if (bq0 + 1 < B) {
        ValIdx* out = partial + (((size_t)(bq0 + 1) * 32 + i) * nslots + slot) * kTopPartial;
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv1[p], bi1[p]};
    }
}

// "Two teams" variant for batches of 16+ queries (round 3).  In the kernel above every wave alternates an MFMA phase
// (48 MFMAs: 1 536 cycles of the matrix pipe) with an epilogue (group maxima, fp16 conversion, LDS transposition, table
// stores), and the work-group barrier per tile keeps the two waves that share a SIMD in step: both want the matrix
// pipe, then both leave it idle -- the pipe was 41 % busy.  s_memtime stamps around the phases (profiles/
// r03_experiments.md) put a wave's tile at ~1 830 cycles of MFMA phase, ~2 500 of epilogue -- 1 220 of them waiting to
// ISSUE its four table stores behind those of the other waves (the CU's texture-address unit takes ~70 cycles per store
// instruction whatever its width) -- and ~500 at the staging barrier.  Three changes:
//   * a work-group has EIGHT waves = two teams of four; wave w and wave w + 4 share a SIMD.  Both teams run
//     MFMA(tile) -> epilogue(tile), but team 1 meets the barrier between the two and team 0 after them: between two
//     barriers team 0 runs MFMA(k), epilogue(k) while team 1 runs epilogue(k-1), MFMA(k), so on every SIMD one wave
//     multiplies while the other converts and transposes.  Still one barrier per tile;
//   * the four table stores of a tile are issued one every twelve MFMAs of the wave's NEXT MFMA phase (data and
//     addresses wait in registers), so a team's sixteen stores no longer arrive at the texture-address unit together;
//   * sixteen queries per staged tile instead of eight: half the L2 reads of the split table.
// The arithmetic per (query, tile) is unchanged, so are the error bound and the partial-list layout.
// 0.1205 -> 0.106 ms per 32 queries x 131 072 centroids together with the hand-counted waits (both kernels) and the
// hardware bf16 split of the query operands.
// grid = (gx, ceil(B / 16)), block = 512, LDS = 2 buffers * 2 arrays * 32 rows * 272 B + 8 waves * 4 KB (66 KB).
constexpr int kTeamQueries = 16;
typedef _Float16 f16x8_c __attribute__((ext_vector_type(8)));

// X1 = true (round 5, "f16x1"): Chi is the fp16 table, Clo is not read -- one v_mfma_f32_32x32x16_f16 per 16 dims and query
// instead of three bf16 ones (see to_f16_kernel); one tile load per stage (the hand-counted waits stay: behind it are still
// exactly the four stores of an MFMA phase).  Same tiles, lists, table layout and stores.
// RANGE (round 6): every token's running minimum and maximum over the work-group's tiles go to rangep, [query][token][gridDim.x]
// {min, max} (token_range_kernel's comment): eight v_min3 per query and tile beside the group maximum the lists take anyway --
// +5 % on the kernel, so only batches whose table will be requantised track it.
template <bool X1, bool RANGE = false>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void centroid_top_bf16x3_teams_kernel(
    const uint16_t* __restrict__ Chi, const uint16_t* __restrict__ Clo, const float* __restrict__ Q,
    ValIdx* __restrict__ partial, uint32_t* __restrict__ cells16, int K, int T, int B, int n_tiles,
    float2* __restrict__ rangep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int team = wave >> 2;
    const int i = lane & 31, h = lane >> 5;
    const int bq0 = blockIdx.y * kTeamQueries + wave * 2;   // this wave's queries: bq0, bq0 + 1
    // loader: thread tid moves 16-B chunk (tid & 15) of row (tid >> 4), hi and lo.  Hand-issued and hand-waited, as in
    // the kernel above: every epilogue issues exactly four stores, so the two loads of a tile have landed once at most
    // four younger vector-memory operations are outstanding
    const int prow = threadIdx.x >> 4, pchunk = threadIdx.x & 15;
    u32x4 ph, pl = {0u, 0u, 0u, 0u};
#define CLB_TM_LOAD(TL)                                                                                   \
    {                                                                                                     \
        int c_ = (TL) * 32 + prow;                                                                        \
        c_ = c_ < K ? c_ : K - 1;                                                                         \
        const uint16_t* a0_ = Chi + (size_t)c_ * kDim + 8 * pchunk;                                       \
        const uint16_t* a1_ = Clo + (size_t)c_ * kDim + 8 * pchunk;                                       \
        if (X1) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ph) : "v"(a0_) : "memory");        \
        else asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"            \
                     : "=&v"(ph), "=&v"(pl) : "v"(a0_), "v"(a1_) : "memory");                             \
    }
#define CLB_TM_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ph), "+v"(pl) :: "memory");
    // the first tile is requested before the query operands are split
    int tile = blockIdx.x, buf = 0;
    CLB_TM_LOAD(tile < n_tiles ? tile : n_tiles - 1)
    u32x4 qh[2][8], ql[2][X1 ? 1 : 8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = bq0 + q < B ? bq0 + q : B - 1;      // past the batch: a duplicate whose results are dropped
        const float* qrow = Q + ((size_t)b * T + (i < T ? i : T - 1)) * kDim + 64 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
            const float4 a = *reinterpret_cast<const float4*>(qrow + 8 * s);
            const float4 c = *reinterpret_cast<const float4*>(qrow + 8 * s + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (X1) { hh[j] = pack_f16(v[2 * j], v[2 * j + 1]); ll[j] = 0u; }
                else split_bf16_pair(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
                if (i >= T) { hh[j] = 0u; ll[j] = 0u; }      // tokens past T: zero operand rows
            }
            qh[q][s] = u32x4{hh[0], hh[1], hh[2], hh[3]};
            ql[q][X1 ? 0 : s] = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
    }
    float bv0[kTopPartial], bv1[kTopPartial];
    int bi0[kTopPartial], bi1[kTopPartial];
#pragma unroll
    for (int p = 0; p < kTopPartial; ++p) { bv0[p] = bv1[p] = kNegInf; bi0[p] = bi1[p] = 0x7fffffff; }
    // 48 MFMAs: the staged tile against this wave's two queries
    // (-DCLB_ABL_CENTROID_X1, tuning builds only: ONE product per fp32 product -- what a single-fp16-product table would cost)
#ifdef CLB_ABL_CENTROID_X1
#define CLB_TM_LO_PRODUCTS(S)      /* (the lo plane is still loaded and staged: an upper bound of the single-product kernel's time) */
#else
#define CLB_TM_LO_PRODUCTS(S)                                                                             \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(bf16x8, qh[0][S]), acc0, 0, 0, 0); \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(bf16x8, qh[1][S]), acc1, 0, 0, 0); \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, ql[0][X1 ? 0 : S]), acc0, 0, 0, 0); \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, ql[1][X1 ? 0 : S]), acc1, 0, 0, 0);
#endif
#define CLB_TM_STORE(DATA, ADDR) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(ADDR), "v"(DATA));
#define CLB_TM_MFMA(MY)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }                  \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                   \
            const u32x4 aw_ = *reinterpret_cast<const u32x4*>((MY) + i * kRowBytes16 + 16 * (8 * h + s));     \
            if (X1) {                                                                                     \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_c, aw_), __builtin_bit_cast(f16x8_c, qh[0][s]), acc0, 0, 0, 0); \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_c, aw_), __builtin_bit_cast(f16x8_c, qh[1][s]), acc1, 0, 0, 0); \
            } else {                                                                                      \
            const bf16x8 ah = __builtin_bit_cast(bf16x8, aw_);                                            \
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>((MY) + (32 + i) * kRowBytes16 + 16 * (8 * h + s))); \
            CLB_TM_LO_PRODUCTS(s)                                                                         \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, qh[0][s]), acc0, 0, 0, 0); \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, qh[1][s]), acc1, 0, 0, 0); \
            }                                                                                             \
            /* the previous tile's four table stores, one every twelve MFMAs (see the epilogue) */            \
            if (s == 0) CLB_TM_STORE(o0, a0)                                                              \
            if (s == 2) CLB_TM_STORE(o1, a1)                                                              \
            if (s == 4) CLB_TM_STORE(o2, a2)                                                              \
            if (s == 6) CLB_TM_STORE(o3, a3)                                                              \
        }                                                                                                 \
    }
    // epilogue of tile TL: group lists, then the 32 x 32 scores of each query become one 2-KB block of fp16 rows
    // [centroid][token] (transposed through the wave's LDS patches) in o0..o3, with the addresses of its four 16-B-per-lane
    // stores in a0..a3 (rows past K and queries past B: the spill block)
#define CLB_TM_EPI(TL)                                                                                    \
    {                                                                                                     \
        const int c0 = (TL) * 32;                                                                         \
        const float g0_ = group_max16(acc0, c0, h, K), g1_ = group_max16(acc1, c0, h, K);                 \
        topn_insert_lazy<kTopPartial>(bv0, bi0, g0_, 2 * (TL) + h);                                       \
        topn_insert_lazy<kTopPartial>(bv1, bi1, g1_, 2 * (TL) + h);                                       \
        if (RANGE) {    /* the token's range (rows past K are copies of row K - 1: they cannot widen it) */ \
            rmax0 = fmaxf(rmax0, g0_); rmax1 = fmaxf(rmax1, g1_);                                         \
            rmin0 = fminf(rmin0, group_min16(acc0)); rmin1 = fminf(rmin1, group_min16(acc1));             \
        }                                                                                                 \
        __builtin_amdgcn_wave_barrier();                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
            const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;                                                \
            *reinterpret_cast<__half*>(patch + cl * 64 + i * 2) = __float2half_rn(acc0[r]);               \
            *reinterpret_cast<__half*>(patch + 2048 + cl * 64 + i * 2) = __float2half_rn(acc1[r]);        \
        }                                                                                                 \
        __builtin_amdgcn_wave_barrier();                                                                  \
        o0 = *reinterpret_cast<const u32x4*>(patch + lane * 16);                                          \
        o1 = *reinterpret_cast<const u32x4*>(patch + 1024 + lane * 16);                                   \
        o2 = *reinterpret_cast<const u32x4*>(patch + 2048 + lane * 16);                                   \
        o3 = *reinterpret_cast<const u32x4*>(patch + 3072 + lane * 16);                                   \
        unsigned char* d0_ = reinterpret_cast<unsigned char*>(cells16 + ((size_t)bq0 * K + c0) * 16) + lane * 16;  \
        unsigned char* d1_ = d0_ + (size_t)K * 64;                                                        \
        unsigned char* sp_ = spill + lane * 16;                                                           \
        const bool in0_ = c0 + (lane >> 2) < K, in1_ = c0 + 16 + (lane >> 2) < K;   /* centroid of the 16-B piece */ \
        a0 = bq0 < B && in0_ ? d0_ : sp_;                                                                 \
        a1 = bq0 < B && in1_ ? d0_ + 1024 : sp_ + 1024;                                                   \
        a2 = bq0 + 1 < B && in0_ ? d1_ : sp_;                                                             \
        a3 = bq0 + 1 < B && in1_ ? d1_ + 1024 : sp_ + 1024;                                               \
    }
    unsigned char* patch = lds16 + 2 * (2 * 32 * kRowBytes16) + wave * 4096;      // one 2-KB patch per query
    unsigned char* spill = reinterpret_cast<unsigned char*>(cells16 + (size_t)B * K * 16);
    float rmin0 = __builtin_inff(), rmin1 = rmin0, rmax0 = -__builtin_inff(), rmax1 = rmax0;
    f32x16 acc0, acc1;
    // the table stores of a tile are issued during the NEXT tile's MFMA phase (data in o0..o3, addresses in a0..a3; the
    // first phase stores zeros to the spill block, the last tile's stores follow the loop)
    u32x4 o0 = {0u, 0u, 0u, 0u}, o1 = o0, o2 = o0, o3 = o0;
    unsigned char *a0 = spill + lane * 16, *a1 = a0 + 1024, *a2 = a0, *a3 = a1;
    // Both teams run  MFMA(tile) -> epilogue(tile)  per iteration; what differs is WHERE in the iteration a team stages
    // the next tile and meets the other at the barrier: team 0 after its epilogue, team 1 between its MFMAs and its
    // epilogue.  Between two barriers team 0 therefore runs MFMA(k), epilogue(k) and team 1 epilogue(k-1), MFMA(k).
    // The buffer written before barrier k+1 held tile k-1, whose last reads (MFMA(k-1), either team) precede barrier k.
    // stage: the registers hold the tile after the staged one (requested one stage ago); behind those two loads are
    // exactly the four stores of one MFMA phase, hence vmcnt(4)
#define CLB_TM_STAGE()                                                                                    \
    {                                                                                                     \
        CLB_TM_WAIT(4)                                                                                    \
        unsigned char* nb_ = lds16 + (buf ^ 1) * (2 * 32 * kRowBytes16);                                  \
        *reinterpret_cast<u32x4*>(nb_ + prow * kRowBytes16 + 16 * pchunk) = ph;                           \
        if (!X1) *reinterpret_cast<u32x4*>(nb_ + (32 + prow) * kRowBytes16 + 16 * pchunk) = pl;           \
        const int after_ = tile + 2 * (int)gridDim.x;                                                     \
        CLB_TM_LOAD(after_ < n_tiles ? after_ : n_tiles - 1)                                              \
        __syncthreads();                                                                                  \
    }
    if (tile < n_tiles) {
        // (tied to the last operand words, or hipcc moves the wait in front of the operand split)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ph), "+v"(pl), "+v"(qh[0][7]), "+v"(qh[1][7]) :: "memory");
        *reinterpret_cast<u32x4*>(lds16 + prow * kRowBytes16 + 16 * pchunk) = ph;
        if (!X1) *reinterpret_cast<u32x4*>(lds16 + (32 + prow) * kRowBytes16 + 16 * pchunk) = pl;
        const int second = tile + (int)gridDim.x;
        CLB_TM_LOAD(second < n_tiles ? second : n_tiles - 1)
        __syncthreads();
    }
    while (tile < n_tiles) {            // uniform over the work-group: tile depends on blockIdx only
        const unsigned char* my = lds16 + buf * (2 * 32 * kRowBytes16);
        CLB_TM_MFMA(my)
        if (team == 1) CLB_TM_STAGE()
        CLB_TM_EPI(tile)
        if (team == 0) CLB_TM_STAGE()
        tile += gridDim.x;
        buf ^= 1;
    }
    // the last stage requested one more (clamped) tile: its two loads must have landed before their destination
    // registers die -- hipcc does not know they are in flight and would hand the registers to the code below
    CLB_TM_WAIT(0)
    CLB_TM_STORE(o0, a0) CLB_TM_STORE(o1, a1) CLB_TM_STORE(o2, a2) CLB_TM_STORE(o3, a3)    // the last tile's
#undef CLB_TM_STAGE
#undef CLB_TM_LOAD
#undef CLB_TM_WAIT
#undef CLB_TM_MFMA
#undef CLB_TM_LO_PRODUCTS
#undef CLB_TM_STORE
#undef CLB_TM_EPI
    const int slot = blockIdx.x * 2 + h;
    const int nslots = gridDim.x * 2;
    if (bq0 < B) {
        ValIdx* out = partial + (((size_t)bq0 * 32 + i) * nslots + slot) * kTopPartial;   // [query][token][slot][entry]
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv0[p], bi0[p]};
    }
    if (bq0 + 1 < B) {
        ValIdx* out = partial + (((size_t)(bq0 + 1) * 32 + i) * nslots + slot) * kTopPartial;
#pragma unroll
        for (int p = 0; p < kTopPartial; ++p) out[p] = ValIdx{bv1[p], bi1[p]};
    }
    if (!RANGE) return;
    // the work-group's range of token i, both lane halves combined (a work-group without a tile leaves +inf / -inf)
    rmin0 = fminf(rmin0, __shfl_xor(rmin0, 32, 64)); rmax0 = fmaxf(rmax0, __shfl_xor(rmax0, 32, 64));
    rmin1 = fminf(rmin1, __shfl_xor(rmin1, 32, 64)); rmax1 = fmaxf(rmax1, __shfl_xor(rmax1, 32, 64));
    if (h == 0 && bq0 < B) rangep[((size_t)bq0 * 32 + i) * gridDim.x + blockIdx.x] = make_float2(rmin0, rmax0);
    if (h == 0 && bq0 + 1 < B) rangep[((size_t)(bq0 + 1) * 32 + i) * gridDim.x + blockIdx.x] = make_float2(rmin1, rmax1);
}

// One wave per (token, query).  The partial lists hold GROUPS (16 centroids of one tile, see group_max16) keyed by
// their largest approximate score.  g2 = the 2nd largest group maximum over all lists is a lower bound of a2, the
// token's 2nd best approximate score, so every centroid with an approximate score >= a2 - 2 eps_c lives in a group
// whose maximum is >= g2 - 2 eps_c.  Those groups (normally 2) are compacted to LDS in two streaming sweeps over the
// lists (they stay in L2) and all their centroids are re-scored with the canonical fp32 fmaf chain, 16 lanes per
// group; the exact top-2 by (score desc, index asc) comes out.  A list whose LAST entry qualifies may have dropped a
// qualifying group, and more than kTopRefine qualifying groups do not fit the wave: both cases fall back to the
// exhaustive canonical scan in this wave (mass ties only).  grid = (32, B), block = 64.
static __global__ __launch_bounds__(64) void top_refine_kernel(const ValIdx* __restrict__ partial,
                                                              const float* __restrict__ C,
                                                              const float* __restrict__ Q, int T, int K,
                                                              int nslots, float cn_max, int* __restrict__ sel,
                                                              int* __restrict__ redo_flag, float dc_max = 0.f) {
    static_assert(kTopPartial == 4, "a partial list is read as two 16-byte halves");
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (t >= T) {
        if (lane == 0) { sel[((size_t)b * 32 + t) * 2] = 0; sel[((size_t)b * 32 + t) * 2 + 1] = 0; }
        return;
    }
    __shared__ float qs[kDim];
    __shared__ ValIdx cand[kTopRefineCap];
    const float* q = Q + ((size_t)b * T + t) * kDim;
    qs[lane] = q[lane];
    qs[lane + 64] = q[lane + 64];
    // a token's lists are contiguous ([query][token][slot][entry]): the two sweeps below read 32 B per lane, coalesced
    // (with the slots outermost every lane fetched its own 1-KB-strided piece: 18 us per query, now 10)
    const ValIdx* lists = partial + ((size_t)b * 32 + t) * nslots * kTopPartial;
    const size_t slot_stride = (size_t)kTopPartial;
    // sweep 1: the lane's two best entries.  Lists are sorted, so only their first two entries can matter here;
    // last_max = the largest LAST entry (the overflow test below).
    float b1v = kNegInf, b2v = kNegInf, last_max = kNegInf;
    int b1i = 0x7fffffff, b2i = 0x7fffffff;
#pragma unroll 4
    for (int sl = lane; sl < nslots; sl += 64) {
        const uint4 lo = *reinterpret_cast<const uint4*>(lists + (size_t)sl * slot_stride);
        const float last = lists[(size_t)sl * slot_stride + kTopPartial - 1].v;
        const float v0 = __uint_as_float(lo.x), v1 = __uint_as_float(lo.z);
        const int i0 = (int)lo.y, i1 = (int)lo.w;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = e ? v1 : v0;
            const int i = e ? i1 : i0;
            const bool gt1 = better(v, i, b1v, b1i), gt2 = better(v, i, b2v, b2i);
            b2v = gt1 ? b1v : (gt2 ? v : b2v);
            b2i = gt1 ? b1i : (gt2 ? i : b2i);
            b1v = gt1 ? v : b1v;
            b1i = gt1 ? i : b1i;
        }
        last_max = fmaxf(last_max, last);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float o1v = __shfl_xor(b1v, o, 64), o2v = __shfl_xor(b2v, o, 64);
        const int o1i = __shfl_xor(b1i, o, 64), o2i = __shfl_xor(b2i, o, 64);
        // merge two sorted pairs: the best two of {b1, b2, o1, o2}
        const bool mine = better(b1v, b1i, o1v, o1i);
        const float w1v = mine ? b1v : o1v, l1v = mine ? o1v : b1v;      // winner / loser of the firsts
        const int w1i = mine ? b1i : o1i, l1i = mine ? o1i : b1i;
        const float s2v = mine ? b2v : o2v;                              // the winner's own second
        const int s2i = mine ? b2i : o2i;
        const bool sec = better(l1v, l1i, s2v, s2i);
        b1v = w1v; b1i = w1i;
        b2v = sec ? l1v : s2v;
        b2i = sec ? l1i : s2i;
        last_max = fmaxf(last_max, __shfl_xor(last_max, o, 64));
    }
    __syncthreads();
    float qq = qs[lane] * qs[lane] + qs[lane + 64] * qs[lane + 64];
    const float d0 = qs[lane] - round_f16(qs[lane]), d1 = qs[lane + 64] - round_f16(qs[lane + 64]);
    float dd = d0 * d0 + d1 * d1;                    // ||q - fp16(q)||^2: the query side of the single-product kernel's error
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qq += __shfl_xor(qq, o, 64); dd += __shfl_xor(dd, o, 64); }
    const float eps_c = kEpsSafety * centroid_product_bound(sqrtf(qq) * 1.001f, sqrtf(dd) * 1.001f, cn_max, dc_max);
    const float thr = b2v - 2.f * eps_c;
    // sweep 2: every entry >= thr goes to the LDS candidate list (order is irrelevant: they are re-scored exactly)
    int total = 0;
    for (int base = 0; base < nslots; base += 64) {
        const int sl = base + lane;
        uint4 lo = make_uint4(0xff800000u, 0x7fffffffu, 0xff800000u, 0x7fffffffu), hi = lo;
        if (sl < nslots) {
            lo = *reinterpret_cast<const uint4*>(lists + (size_t)sl * slot_stride);
            hi = *reinterpret_cast<const uint4*>(lists + (size_t)sl * slot_stride + 2);
        }
        const float v[4] = {__uint_as_float(lo.x), __uint_as_float(lo.z), __uint_as_float(hi.x), __uint_as_float(hi.z)};
        const int id[4] = {(int)lo.y, (int)lo.w, (int)hi.y, (int)hi.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool take = id[e] != 0x7fffffff && v[e] >= thr;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
            if (m == 0) break;                                   // sorted lists: later entries are smaller
            const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
            if (take && pos < kTopRefineCap) cand[pos] = ValIdx{v[e], id[e]};
            total += __popcll(m);
        }
    }
    __syncthreads();
    // a list holds more than kTopPartial groups' worth of history only if the lane saw that many tiles
    const bool overflow = total > kTopRefineCap || last_max >= thr;
    float tv[2] = {kNegInf, kNegInf};
    int ti[2] = {0x7fffffff, 0x7fffffff};
    if (!overflow) {
        // 16 lanes per qualifying group: lane (grp, rr) re-scores centroid rr of group grp with the chain the fp32
        // MFMA kernel performs; kTopRefine groups per round (normally one round: two groups qualify)
        for (int g0 = 0; g0 < total; g0 += kTopRefine) {
        const int grp = g0 + (lane >> 4), rr = lane & 15;
        const int gid = grp < total ? cand[grp].i : 0;
        const int id = (gid >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (gid & 1);
        if (grp < total && id < K) {
            const float4* c4 = reinterpret_cast<const float4*>(C + (size_t)id * kDim);
            float4 cr[32];
#pragma unroll
            for (int m = 0; m < 32; ++m) cr[m] = c4[m];               // all loads in flight, then the ordered chain
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                a = fmaf(cr[m].x, qs[4 * m], a);
                a = fmaf(cr[m].y, qs[4 * m + 1], a);
                a = fmaf(cr[m].z, qs[4 * m + 2], a);
                a = fmaf(cr[m].w, qs[4 * m + 3], a);
            }
            topn_insert<2>(tv, ti, a, id);
        }
        }
    } else {
        for (int c = lane; c < K; c += 64) {
            const float* cr = C + (size_t)c * kDim;
            float a = 0.f;
            for (int d = 0; d < kDim; ++d) a = fmaf(cr[d], qs[d], a);
            topn_insert<2>(tv, ti, a, c);
        }
        if (lane == 0) atomicAdd(&redo_flag[b], 1);   // statistics only
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v0 = __shfl_xor(tv[0], o, 64), v1 = __shfl_xor(tv[1], o, 64);
        const int i0 = __shfl_xor(ti[0], o, 64), i1 = __shfl_xor(ti[1], o, 64);
        topn_insert<2>(tv, ti, v0, i0);
        topn_insert<2>(tv, ti, v1, i1);
    }
    if (lane == 0) {
        sel[((size_t)b * 32 + t) * 2] = ti[0];
        sel[((size_t)b * 32 + t) * 2 + 1] = ti[1] == 0x7fffffff ? ti[0] : ti[1];
    }
}

// ---- cells fp32 [K][Tpad=32] -> fp16 [K][32] in token order.  grid = (blocks, B), block = 256 -----------------
static __global__ __launch_bounds__(256) void cells_to_half_kernel(const float* __restrict__ cells,
                                                                  uint32_t* __restrict__ cells16, int K) {
    const int b = blockIdx.y;
    const float* src = cells + (size_t)b * K * 32;
    uint32_t* dst = cells16 + (size_t)b * K * 16;
    const int64_t total = (int64_t)K * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i >> 4;
        const int t = (int)(i & 15);
        const float lo = src[c * 32 + 2 * t], hi = src[c * 32 + 2 * t + 1];
        const __half2 h = __floats2half2_rn(lo, hi);
        dst[i] = *reinterpret_cast<const uint32_t*>(&h);
    }
}

// -------------------------------------------------------------------------------------------------------------
// Pass 1.  A wave walks its candidate passages in steps of 32 embeddings; one step is one 32 x 32 tile
// S[e][t] = ( X[code_e][t] + sum_d w[idx_e,d] * Q[t][d] ) * inv_norm[e]  on v_mfma_f32_32x32x16:
//   * Q.r: A (32 embeddings x 16 dims per k-step) = fp16 bucket weights expanded from the packed residual through a
//     2-KB LDS table (one ds_read_b64 per residual byte: 4 dims); lane (r = lane & 31, h = lane >> 5) owns bytes
//     16h .. 16h+15 of embedding r -- ONE 16-byte load per lane and step -- i.e. dims 64h + 8s + j in k-step s;
//     B = fp16(Q) resident in registers in the same dim order (8 k-steps x 4 VGPRs);
//   * X (the fp16 centroid scores of S1, rows of 32 tokens = 64 B): two more MFMAs (f16 inputs) against a 0/1
//     selection matrix ADD the gathered row into the accumulator: lane (r, h) fetches tokens 8h..8h+7 and
//     16+8h..16+8h+7 of row code_r as two 16-byte loads, which are exactly the A fragments of those MFMAs (k = token).
//     No fp16->fp32 conversion, no transposition, and 2 wide gather instructions per 32 embeddings where the 16-row
//     kernel of round 1 issued 8 dword gathers (that kernel was bound by its request count, not its bytes);
//   * inv_norm: one dword per lane (embedding r), moved to the accumulator layout (lane = token, register = row)
//     through a 128-byte per-wave LDS patch: 1 ds_write_b32 + 4 ds_read_b128;
//   * epilogue: 16 v_mul_f32 + 8 v_max3 per lane; at a passage's last step the two lane halves are combined and the
//     32 per-token maxima summed.
// Rows past the end of a passage (tail step) are DUPLICATES of the passage's last row -- the lane clamps its row
// index before it forms any address, so residual, code, score row and inv_norm all belong to that row -- and a
// duplicate cannot change a maximum: no masking anywhere, and no bytes fetched from behind the passage.
// Software pipeline, three steps in flight per wave: A(i+2) residual + code loads; G(i+1) score-row gather + inv_norm
// (both need A's data: the code, resp. nothing); C(i) compute.  The loop body is branch-free apart from the
// end-of-passage store; steps past the end of the wave's work are redirected to embedding 0 and discarded.
// Work-groups are dealt to queries by `blockIdx.x % 8` (the label of the XCD they share under round-robin placement
// -- a speed heuristic only): the work-groups of one XCD gather from ONE query's score table at a time.
// grid = 8 * wg_per_group (1-D) or (G, B) (2-D: few passages per query), block = kApproxThreads (12 waves).
// -------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StepTag {      // wave-uniform description of a step
    int j;            // candidate slot (ROWS: list position), -1 = dummy
    int rows;         // valid rows (1..32)
    int last;         // 1 = last step of its passage; 3 = ... which is also the last passage of the wave's 64-passage chunk
    int base;         // index of the step's first embedding inside its passage
};

constexpr int kStepRows = 32;      // embeddings per step; codes0 / residuals / inv_norm are padded by this many entries

// fp32 -> fp16 rounded toward -inf (a lower bound of x that is at most one fp16 ulp away)
__device__ __forceinline__ uint32_t f32_to_f16_floor(float x) {
    const __half hr = __float2half_rn(x);
    uint32_t bits = *reinterpret_cast<const uint16_t*>(&hr);
    if (__half2float(hr) > x) bits = (bits & 0x8000u) ? bits + 1u : (bits == 0u ? 0x8001u : bits - 1u);
    return bits;
}

// LDS table offset of residual byte N of word R: (byte << 8) | lane8 in ONE VALU op (v_perm_b32: byte N of R into
// byte 1, byte 0 of lane8 into byte 0, zeros above)
template <int N>
__device__ __forceinline__ uint32_t lut_offset(uint32_t R, uint32_t lane8) {
    return __builtin_amdgcn_perm(R, lane8, 0x0c0c0000u | ((4u + N) << 8));
}

// wave64 helpers on DPP / permlane (no LDS traffic): sum of lanes 0..31 delivered in lanes 16..31; max over the two
// lane halves delivered in every lane
#define CLB_DPP_F(V, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (V)), (CTRL), 0xf, 0xf, true))
__device__ __forceinline__ float sum_lanes_0_31(float s) {
    s += CLB_DPP_F(s, 0xB1);     // quad_perm [1,0,3,2]: lane ^ 1
    s += CLB_DPP_F(s, 0x4E);     // quad_perm [2,3,0,1]: lane ^ 2
    s += CLB_DPP_F(s, 0x141);    // row_half_mirror: the other quad of the 8
    s += CLB_DPP_F(s, 0x140);    // row_mirror: the other 8 of the row; every lane of a row now holds the row's sum
    // row_bcast15 into rows 1 and 3: lanes 16..31 receive row 0's sum (lanes of rows 0 and 2 receive 0)
    const float below = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x142, 0xa, 0xf, false));
    return s + below;
}
// (hipcc drops the second result of __builtin_amdgcn_permlane32_swap(m, m): one ds_bpermute per passage instead)
__device__ __forceinline__ float max_lane_halves(float m) { return fmaxf(m, __shfl_xor(m, 32, 64)); }

#ifndef CLB_APPROX_WAVES
#define CLB_APPROX_WAVES 12
#endif
#ifdef CLB_APPROX_SPLIT_LUT
constexpr bool kSplitLut = true;    // experiment: the 16 table reads of a step in two halves (16 VGPRs less)
#else
constexpr bool kSplitLut = false;
#endif
// experiment builds (make SUF=_prio EXTRA=-DCLB_APPROX_PRIO=1): the wave raises its issue priority for the ten dependent MFMAs of a step
#if defined(CLB_APPROX_PRIO) && CLB_APPROX_PRIO
#define CLB_APPROX_PRIO_UP __builtin_amdgcn_s_setprio(CLB_APPROX_PRIO);
#define CLB_APPROX_PRIO_DOWN __builtin_amdgcn_s_setprio(0);
#else
#define CLB_APPROX_PRIO_UP
#define CLB_APPROX_PRIO_DOWN
#endif
// waves per SIMD the register budget is set for: the work-group's own by default; an experiment build may ask for more than its
// waves need (make SUF=_w8 EXTRA='-DCLB_APPROX_WAVES=8 -DCLB_APPROX_MINOCC=3': two waves per SIMD within the registers of three,
// which leaves a third of every SIMD's register file to the kernels of the OTHER batch in flight)
#ifndef CLB_APPROX_MINOCC
#define CLB_APPROX_MINOCC (CLB_APPROX_WAVES / 4)
#endif
constexpr int kApproxThreads = 64 * CLB_APPROX_WAVES;   // 12 waves per work-group = 3 per SIMD, one work-group per CU
constexpr int kApproxLdsLut = 256 * 256;            // 256 entries x 32 lane slots x 8 B

// ROWS = false: pass 1 over every candidate; besides the score it leaves tokmax[b][slot][32] = the per-token maxima
//        a_t = max_j A[t][j] of every candidate passage as fp16 rounded DOWN.
// ROWS = true:  pass 2 preparation over the passages in `list` -- which embeddings of a listed passage can hold a
//        per-token maximum?  With |A - S| <= e (e = eps_pair[b]) the exact argmax j* of token t satisfies
//        A[t][j*] >= S[t][j*] - e >= S[t][ja] - e >= a_t - 2e (ja = the approximate argmax).  So the exact kernel only
//        has to decompress and multiply the rows J = { j : exists t, A[t][j] >= a_t - 2e } -- typically a third of the
//        passage -- and still finds the identical per-token maxima, hence the identical fp32 score.  The sweep
//        recomputes A with the same pipeline (bit-identical values), compares against the stored (floored, so only
//        more permissive) a_t and writes rowmask[b][list position][4] x 64 bits; passages longer than kMaxMaskedRows
//        embeddings are ignored downstream (the exact kernel then takes every row).  Comparisons are written
//        !(v < lo): a NaN or an infinite window (guarded query, select_margin_kernel) selects the row.
// ABL != 0: ablation variants for the roofline analysis (instantiated only in -DCLB_ABLATIONS builds; results are
// wrong by design): 1 no score-row gather; 2 gather from a 64-KB window of the table (always L2-hot); 3 no residual
// stream; 4 plain (temporal) stream loads; 5 no LUT expansion / MFMA; 6 v_pk_mul_f32 instead of v_mul_f32; 7 no memory
// access in the loop at all; 8 no result stores (what the passage-end stores and the waits they widen cost).
// Round 6: 11 = the walk over the passages in closed form (every passage three steps, no v_readlane header look-ups, a third of
// the scalar instructions): the most a step-descriptor table read with s_load could save (profiles/r06_experiments.md section 2).
// Round 5 (profiles/r05_pass1_ablations.jsonl): 10 = the row-mask sweep folded into pass 1 (every step compares its 16 values
// against the RUNNING per-token maximum and the passage's 256-bit mask is stored at its last step: what that costs the
// dominant kernel).  (Variant 9 of that round, the timing side of the 8-bit score rows, became CELL8.)
// GL = 1 ("LDS-DMA gather", round 3): the score rows reach the wave through LDS instead of VGPRs.  Four ADJACENT lanes
// fetch the 64 bytes of one row with global_load_lds_dwordx4 (16 rows per instruction, 2 instructions per step) -- one
// L1 tag look-up per row where the VGPR form (lane (r, h) fetching 2 x 16 B of row r, the lanes of a row 32 apart)
// costs four: the texture-address unit was the busiest unit of the kernel (TA_BUSY 75 %) whenever the codes of a
// passage are not id-adjacent (uniform codes, a k-means-built index).  The DMA writes lane l's 16 bytes at slot + 16 l,
// so WHICH (row, piece) a lane fetches decides the LDS image: row m of an instruction at 64 m, its four pieces rotated by
// (m >> 2) & 3 -- the ds_read_b128 of lane (r, h) (the A fragments of the two score-row MFMAs) is then conflict-free.
// The code of row m is taken from lane m with one ds_bpermute per instruction.  A ring of three 2-KB slots per wave
// (one per step in flight).  The two DMAs of a step are issued by inline assembly, NOT by __builtin_amdgcn_global_load_lds:
// hipcc orders every LDS instruction that may touch DMA-written memory -- here the ring reads AND the ds_bpermutes --
// behind ALL pending DMAs with s_waitcnt vmcnt(0), which drains the stage-A loads issued a moment earlier and exposes
// the full memory latency in every step.  Stage C instead waits for its own slot with s_waitcnt vmcnt(4): the VMEM
// operations issued after a step's two DMAs are at least the two loads of the next stage A and the two DMAs of the next
// stage G (the passage-end stores only add to that), and vector memory operations complete in order.  The compiler's
// own vmcnt arithmetic for the VGPR loads does not see the DMAs; its waits are therefore stronger than needed by the
// DMAs in between, never weaker.  A slot is re-filled three stages after its reads were waited for (lgkmcnt) -- both
// in program order, and the "memory" clobbers keep the compiler from moving the ring reads across either.
// PIPE = 1 (round-3 experiment, instantiated in -DCLB_ABLATIONS builds only): the epilogue of step i-1 (16 multiplies by
// inv_norm, the maxima, the row-mask bits) is issued in the shadow of step i's MFMA chain -- a wave issues in order and
// the ten MFMAs of a step depend on each other (SQ_WAIT_INST_ANY: 36 % of the wave cycles).  hipcc does interleave the
// two streams, but the second accumulator set takes the kernel to the 168-VGPR cap of three waves per SIMD (7-24 spilled
// registers) and the pass got SLOWER on every workload (0.662 -> 0.672 ms, uniform codes 1.46 -> 1.56, built index
// 0.653 -> 0.699; the row sweep 0.069 -> 0.107): the pass is not short of issue slots, it waits on memory.
// CELL8 = true (round 6): the score rows are 32-byte rows of 8-bit cells (token_range_kernel's comment).  Lane (r, h) fetches
// ONE 16-byte piece (tokens 16h .. 16h+15 of row code_r) and expands it with eight v_perm_b32 to the fp16 values 1024 + cell
// (0x6400 | cell), which the two selection MFMAs add into an accumulator that starts at k_t - 1024 (lane = token): after
// them it holds cell + k_t.  The query operand is fp16(Q_t / step_t), so the eight Q.r MFMAs add Q_t.r / step_t and
// acc * step_t is the token's score: the positive per-token factor commutes with the maximum over a passage's embeddings and
// is applied once per passage (the row sweep applies it per value: the same product for the row that holds the maximum).
// GL = 1: two adjacent lanes fetch a row -- ONE DMA instruction per step, a 1-KB ring slot, vmcnt(3).
template <bool ROWS, int ABL = 0, int GL = 0, int PIPE = 0, bool CELL8 = false>
static __global__ __launch_bounds__(kApproxThreads, CLB_APPROX_MINOCC) void score_approx32_kernel(
    const float* __restrict__ weights, const uint32_t* __restrict__ codeinv, const uint8_t* __restrict__ residuals,
    int cbits, float inv_lo, float inv_step, const float* __restrict__ Q, const uint32_t* __restrict__ cells16,
    const uint2* __restrict__ cand_hdr, const int* __restrict__ ncand, float* __restrict__ scores, int K, int T,
    int B, size_t cand_cap, uint16_t* __restrict__ tokmax, const int* __restrict__ list,
    const int* __restrict__ nlist, const float* __restrict__ eps_pair, unsigned long long* __restrict__ rowmask,
    const float4* __restrict__ tscale = nullptr) {
    const uint32_t cmask = (1u << cbits) - 1u;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int x = blockIdx.x & 7;             // XCD group label
    const int wg = blockIdx.x >> 3;           // index inside the group
    const int wg_per_group = gridDim.x >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // byte LUT in LDS: entry v = fp16 bucket weights of the 4 dims packed in residual byte v (LSB-first 2-bit fields),
    // replicated once per lane slot (lane & 31) at v * 256 + slot * 8: every lane of a ds_read_b64 group reads its own
    // bank pair, so the 16 table reads of a step are conflict-free whatever the bytes are (a single 2-KB table costs
    // ~2 extra LDS cycles per read on random bytes and made the LDS pipe the busiest unit of the kernel)
    __shared__ __attribute__((aligned(16))) unsigned char lut_s[kApproxLdsLut];
    __shared__ __attribute__((aligned(16))) float invx[kApproxThreads / 64][2 * kStepRows];   // two patches (PIPE)
    static_assert(GL == 0 || ABL == 0, "the ablation variants exist for the VGPR gather only");
    static_assert(!CELL8 || ABL == 0, "the ablation variants exist for the fp16 score rows only");
    constexpr int kSlotBytes = CELL8 ? 1024 : 2048;      // 32 rows x 32 B / 64 B
    constexpr int kRingBytes = 3 * kSlotBytes;           // three steps in flight, per wave
    static_assert(GL == 0 || CLB_APPROX_WAVES <= 12, "the 3-slot ring of the LDS-DMA form fits 160 KB of LDS up to 12 waves");
    __shared__ __attribute__((aligned(16))) unsigned char ring_s[GL ? (kApproxThreads / 64) * kRingBytes : 16];
    // Results of up to 16 consecutive passages of a wave wait here (32 fp16 token maxima + the fp32 score each) and leave as
    // TWO full-width stores (round 6).  A wave used to issue a 2-byte-per-lane and a one-lane store at every passage end, inside
    // a branch: ~0.03 ms of the 0.66-ms pass went into them and into the vector-memory waits they widen (ablation variant 8).
    // A wave therefore owns a CONTIGUOUS range of candidate slots now (it used to take every stride-th one), so that the
    // results of consecutive passages are consecutive in memory.
    constexpr int kStageBytes = 16 * 64 + 64;
    __shared__ __attribute__((aligned(16))) unsigned char stage_s[ROWS ? 16 : (kApproxThreads / 64) * kStageBytes];
    unsigned char* mystage = stage_s + (ROWS ? 0 : wave * kStageBytes);
    for (int i = threadIdx.x; i < 256 * 32; i += kApproxThreads) {
        const int v = i >> 5;
        *reinterpret_cast<uint2*>(lut_s + (size_t)i * 8) =
            make_uint2(pack_f16(weights[v & 3], weights[(v >> 2) & 3]), pack_f16(weights[(v >> 4) & 3], weights[(v >> 6) & 3]));
    }
    __syncthreads();
    const char* lut = reinterpret_cast<const char*>(lut_s);
    float* myinv = invx[wave];
    const uint32_t lane8 = 8u * (uint32_t)r;
    // GL: lane l = 4 m + q of a DMA instruction fetches, for row m (rows 0..15 / 16..31), the piece that belongs at
    // position q of the rotated row image; lane (r, h) later reads pieces h and 2 + h of row r
    unsigned char* myring = ring_s + (GL ? wave * kRingBytes : 0);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)myring;
    const uint32_t gl_poff = 16u * (((uint32_t)lane - (((uint32_t)lane >> 4) & 3u)) & 3u);   // piece = (q - rot(m)) & 3, rot(m) = (m >> 2) & 3
    const int gl_src0 = (int)((uint32_t)lane & ~3u);            // ds_bpermute byte address of lane m = lane >> 2
    const int gl_src1 = gl_src0 + 64;                           // lane 16 + m
    const uint32_t gl_rot = ((uint32_t)r >> 2) & 3u;
    const uint32_t gl_x0 = (uint32_t)r * 64u + 16u * (((uint32_t)h + gl_rot) & 3u);
    const uint32_t gl_x1 = (uint32_t)r * 64u + 16u * (((uint32_t)h + 2u + gl_rot) & 3u);
    // CELL8 + GL: lane l = 2 m + q of the ONE DMA fetches piece q ^ rot8(m) of row m (the code of lane m), rot8(m) = (m >> 3) & 1;
    // it lands at 32 m + 16 q, and lane (r, h) reads its piece h at 32 r + 16 (h ^ rot8(r)): 16 consecutive lanes then cover
    // all 64 banks once (rows 0..7 at 32 r, rows 8..15 at 32 r + 16)
    const int gl8_src = (int)(((uint32_t)lane >> 1) << 2);
    const uint32_t gl8_poff = 16u * (((uint32_t)lane ^ ((uint32_t)lane >> 4)) & 1u);
    const uint32_t gl8_x = (uint32_t)r * 32u + 16u * (((uint32_t)h ^ ((uint32_t)r >> 3)) & 1u);

    // selection matrices of the two score-row MFMAs: B1[k][col] = (col == k), B2[k][col] = (col == 16 + k), with
    // k = 8h + j held by lane (col = r, h) in element j
    f16x8 sel1, sel2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // (CELL8: the lane's one piece holds tokens 16h .. 16h+15 -- its low half feeds the first MFMA, its high half the second)
        sel1[j] = (r == (CELL8 ? 16 * h + j : 8 * h + j)) ? (_Float16)1.0f : (_Float16)0.0f;
        sel2[j] = (r == (CELL8 ? 16 * h + 8 + j : 16 + 8 * h + j)) ? (_Float16)1.0f : (_Float16)0.0f;
    }

    int b_first, b_step, sub, nsub;
    if (B >= 8) { b_first = x; b_step = 8; sub = 0; nsub = 1; }
    else { b_first = x % B; b_step = B * 8 /* one query per group */; sub = x / B; nsub = (8 - b_first + B - 1) / B; }
    // 2-D launch (grid = (G, B)): work-group blockIdx.x of query blockIdx.y -- used when a query has few passages
    // (the row-mask sweep; small shards), which are better spread over few, long-running waves
    const bool grid2d = gridDim.y > 1;
    const int wg_count = grid2d ? (int)gridDim.x : wg_per_group;
    const int wg_index = grid2d ? (int)blockIdx.x : wg;
    if (grid2d) { b_first = blockIdx.y; b_step = B; sub = 0; nsub = 1; }

    for (int b = b_first; b < B; b += b_step) {
        // B operand: fp16 Q[t = r][64h + 8s + j], k-steps s = 0..7
        u32x4 qb[8];
        // CELL8: token r's step (score units per cell) and its inverse, the factor of the fp16 query operand
        const float4 ts = CELL8 ? tscale[(size_t)b * 32 + r] : make_float4(1.f, 1.f, 0.f, 0.f);
        const float stp = ts.x;
        const float acc_init = CELL8 ? ts.z - 1024.0f : 0.f;
        {
            const float* qrow = Q + ((size_t)b * T + (r < T ? r : T - 1)) * kDim + 64 * h;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float4 lo = *reinterpret_cast<const float4*>(qrow + 8 * s);
                float4 hi = *reinterpret_cast<const float4*>(qrow + 8 * s + 4);
                if (r >= T) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
                if (CELL8) {     // (the same products query_bound measures the fp16 rounding of)
                    lo.x *= ts.y; lo.y *= ts.y; lo.z *= ts.y; lo.w *= ts.y;
                    hi.x *= ts.y; hi.y *= ts.y; hi.z *= ts.y; hi.w *= ts.y;
                }
                qb[s] = u32x4{pack_f16(lo.x, lo.y), pack_f16(lo.z, lo.w), pack_f16(hi.x, hi.y), pack_f16(hi.z, hi.w)};
            }
        }
        const uint2* hdr = cand_hdr + (size_t)b * cand_cap;
        // uniform base + 32-bit per-lane byte offset (code * 64 + 16h): the scalar-base form of global_load
        const char* c16 = reinterpret_cast<const char*>(cells16 + (size_t)b * K * (CELL8 ? 8 : 16));
        const uint32_t h16 = 16u * (uint32_t)h;
        float* out = scores + (size_t)b * cand_cap;
        uint16_t* tmax = tokmax + (size_t)b * cand_cap * 32 + r;
        const int* lst = ROWS ? list + (size_t)b * cand_cap : nullptr;
        unsigned long long* rmask = ROWS ? rowmask + (size_t)b * cand_cap * 4 : nullptr;
        const float window = ROWS ? 2.f * eps_pair[b] : 0.f;
        const int n = ROWS ? nlist[b] : ncand[b];
        // this wave's contiguous share of the query's n passages
        const int n_waves = wg_count * (kApproxThreads / 64) * nsub;
        const int my_wave = (sub * wg_count + wg_index) * (kApproxThreads / 64) + wave;
        const int per_wave = (n + n_waves - 1) / n_waves;
        const int j_first = min(n, my_wave * per_wave), j_end = min(n, j_first + per_wave);
        unsigned long long wm0 = 0, wm1 = 0, wm2 = 0, wm3 = 0;   // ROWS: the current passage's mask (wave-uniform)

        // ---- wave-uniform iterator over the steps of passages j0, j0 + 1, ... ---------------------------------
        // The headers {first embedding, length} of the wave's next 64 passages sit in one VGPR pair (lane k =
        // k-th passage) and are extracted with v_readlane: no memory wait at a passage switch.
        for (int j0 = j_first; j0 < j_end; j0 += 64) {
        const int jl = j0 + lane;
        int slot_l = jl < j_end ? jl : j0;                   // candidate slot of this lane's passage
        if (ROWS) slot_l = lst[slot_l];
        const uint2 hv = hdr[slot_l];
        const int nd = j_end - j0 < 64 ? j_end - j0 : 64;    // passages here
        int it_k = 0;
        uint32_t it_off = __builtin_amdgcn_readlane(hv.x, 0);
        int it_len = (int)__builtin_amdgcn_readlane(hv.y, 0);
        int it_slot = __builtin_amdgcn_readlane(slot_l, 0);
        int it_base = 0;
        // ABL 11: every passage of the chunk 82 embeddings long (the headline mean; 3 steps, mean 3.05), their first embeddings evenly
        // spread over the chunk's real span: the memory pattern of the real walk without its per-passage header look-ups
        const uint32_t abl_stride = ABL == 11 ? ((uint32_t)__builtin_amdgcn_readlane(hv.x, nd - 1) - it_off) / (uint32_t)nd : 0u;
        if (ABL == 11) it_len = 82;
        const uint32_t abl_bound = ABL == 11 ? (uint32_t)__builtin_amdgcn_readlane(hv.x, nd - 1) : 0u;   // (the arrays are padded by a step)

        // stage A: residual bytes (16 B / lane), code and inv_norm of the lane's row (dword each), ROWS: stored maximum
#define CLB_STAGE_A(RB, CV, PM, TAG)                                                                    \
    {                                                                                                       \
        const bool live = it_k < nd;                                                                        \
        const uint32_t e0 = live ? (ABL == 11 ? min(it_off + (uint32_t)it_base, abl_bound) : it_off + (uint32_t)it_base) : 0u; \
        const int left = live ? it_len - it_base : kStepRows;                                               \
        const int rows = left < kStepRows ? left : kStepRows;                                               \
        if (ROWS) PM = tmax[(size_t)(live ? it_slot : 0) * 32];   /* the passage's stored maximum of token r */ \
        const uint32_t rr = min((uint32_t)r, (uint32_t)(rows - 1));   /* tail lanes duplicate the last row */ \
        /* wave-uniform 64-bit bases + 32-bit lane offsets: the scalar-base form of global_load        */ \
        const uint8_t* rbase_ = residuals + (size_t)e0 * 32;                                                \
        const uint32_t* cbase_ = codeinv + (size_t)e0;                                                      \
        const uint32_t roff_ = rr * 32u + h16;                                                              \
        /* the three streams are read once per query: non-temporal loads keep them from evicting the score  */ \
        /* table, which the gathers want in L2 (measured: 0.779 -> 0.762 ms)                               */ \
        if (ABL == 3 || ABL == 7) RB = u32x4{e0 * 2654435761u + rr, e0 ^ h16, e0 + 77u * rr, e0 * 40503u};  \
        else if (ABL == 4) RB = *reinterpret_cast<const u32x4*>(rbase_ + roff_);                            \
        else RB = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rbase_ + roff_));               \
        if (ABL == 7) CV = (e0 * 97u + rr) & 131071u;                                                       \
        else if (ABL == 4) CV = cbase_[rr];                                                                 \
        else CV = __builtin_nontemporal_load(cbase_ + rr);      /* code | quantised inv_norm */              \
        TAG.j = live ? j0 + it_k : -1;                                                                      \
        TAG.rows = rows;                                                                                    \
        TAG.last = left <= kStepRows ? (it_k + 1 >= nd ? 3 : 1) : 0;                                        \
        TAG.base = it_base;                                                                                 \
        it_base += kStepRows;                                                                               \
        if (ABL == 11) {        /* closed-form walk: no header look-ups (what a step-descriptor table could save at most) */ \
            if (TAG.last) { it_k += 1; it_off += abl_stride; it_base = 0; }                                 \
        } else                                                                                              \
        if (TAG.last) {                                                                                     \
            it_k += 1;                                                                                      \
            const int kk = it_k < 64 ? it_k : 63;                                                           \
            it_off = __builtin_amdgcn_readlane(hv.x, kk);                                                   \
            it_len = (int)__builtin_amdgcn_readlane(hv.y, kk);                                              \
            it_slot = __builtin_amdgcn_readlane(slot_l, kk);                                                \
            it_base = 0;                                                                                    \
        }                                                                                                   \
    }
        // stage G: the score row of the lane's embedding: tokens 8h..8h+7 and 16+8h..16+8h+7 (fp16), 2 x 16 B
#define CLB_STAGE_G(CV, X0, X1, SLOT)                                                                       \
    if (GL && CELL8) {                                                                                      \
        const uint32_t ca_ = (uint32_t)__builtin_amdgcn_ds_bpermute(gl8_src, (int)CV) & cmask;              \
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"                                  \
                     :: "v"(c16 + ((ca_ << 5) + gl8_poff)), "s"(ring_lds + (SLOT) * 1024u) : "memory");     \
    } else if (GL) {                                                                                        \
        const uint32_t ca_ = (uint32_t)__builtin_amdgcn_ds_bpermute(gl_src0, (int)CV) & cmask;              \
        const uint32_t cb_ = (uint32_t)__builtin_amdgcn_ds_bpermute(gl_src1, (int)CV) & cmask;              \
        /* hand-issued (see the kernel header): M0 = LDS byte address of the slot, lane l lands at M0 + 16 l  */ \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, off\n\t"                            \
                     "s_add_u32 m0, m0, 0x400\n\tglobal_load_lds_dwordx4 %1, off"                          \
                     :: "v"(c16 + ((ca_ << 6) + gl_poff)), "v"(c16 + ((cb_ << 6) + gl_poff)),               \
                        "s"(ring_lds + (SLOT) * 2048u) : "memory", "scc");                                   \
    } else {                                                                                                \
        const char* row_ = CELL8 ? c16 + (((CV & cmask) << 5) + h16)                                        \
                                 : c16 + (((ABL == 2 ? (CV & 1023u) : (CV & cmask)) << 6) + h16);           \
        if (ABL == 1 || ABL == 7) { X0 = u32x4{CV, CV, CV, CV}; X1 = X0; }                                  \
        else if (CELL8) X0 = *reinterpret_cast<const u32x4*>(row_);      /* 16 one-byte cells: tokens 16h .. 16h+15 */ \
        else {                                                                                              \
        X0 = *reinterpret_cast<const u32x4*>(row_);                                                         \
        X1 = *reinterpret_cast<const u32x4*>(row_ + 32);                                                    \
        }                                                                                                   \
    }
#define CLB_LUT(W, N) (*reinterpret_cast<const uint2*>(lut + lut_offset<N>(W, lane8)))
#define CLB_STAGE_CM(RB, CV, X0, X1, SLOT, ACC, INVB)                                                       \
    {                                                                                                       \
        if (GL && CELL8) {                                                                                  \
            /* one DMA per step: behind it are at least the next stage A's two loads and the next stage G's DMA */ \
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                \
            X0 = *reinterpret_cast<const u32x4*>(myring + (SLOT) * 1024 + gl8_x);                           \
        } else if (GL) {                                                                                    \
            /* this step's two DMAs have landed once at most the 4 youngest VMEM operations are pending     */ \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                \
            X0 = *reinterpret_cast<const u32x4*>(myring + (SLOT) * 2048 + gl_x0);                           \
            X1 = *reinterpret_cast<const u32x4*>(myring + (SLOT) * 2048 + gl_x1);                           \
        }                                                                                                   \
        /* inv_norm: lane layout (row = r) -> accumulator layout (register i = row (i&3) + 8(i>>2) + 4h)  */ \
        __builtin_amdgcn_wave_barrier();                                                                    \
        myinv[(INVB) * kStepRows + r] = fmaf((float)(CV >> cbits), inv_step, inv_lo);                                            \
        __builtin_amdgcn_wave_barrier();                                                                    \
        /* residual byte -> 4 fp16 bucket weights through the LDS table; k-step s = bytes 2s, 2s+1.  All 16  */ \
        /* reads are issued before the first MFMA (an LDS read takes longer than an MFMA: interleaved one  */ \
        /* pair ahead, as the compiler schedules them on its own, the MFMA chain runs at the LDS latency)   */ \
        uint2 tl[16];                                                                                       \
        if (ABL != 5) {                                                                                     \
            tl[0] = CLB_LUT(RB[0], 0); tl[1] = CLB_LUT(RB[0], 1); tl[2] = CLB_LUT(RB[0], 2); tl[3] = CLB_LUT(RB[0], 3);     \
            tl[4] = CLB_LUT(RB[1], 0); tl[5] = CLB_LUT(RB[1], 1); tl[6] = CLB_LUT(RB[1], 2); tl[7] = CLB_LUT(RB[1], 3);     \
            if (!kSplitLut) {                                                                               \
            tl[8] = CLB_LUT(RB[2], 0); tl[9] = CLB_LUT(RB[2], 1); tl[10] = CLB_LUT(RB[2], 2); tl[11] = CLB_LUT(RB[2], 3);   \
            tl[12] = CLB_LUT(RB[3], 0); tl[13] = CLB_LUT(RB[3], 1); tl[14] = CLB_LUT(RB[3], 2); tl[15] = CLB_LUT(RB[3], 3); \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        CLB_APPROX_PRIO_UP                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) ACC[i] = acc_init;                                   \
        if (CELL8) {         /* byte b -> fp16 1024 + b (0x6400 | b): two cells per v_perm_b32 */            \
            const u32x4 raw_ = X0;                                                                          \
            X0 = u32x4{__builtin_amdgcn_perm(0x64646464u, raw_[0], 0x04010400u), __builtin_amdgcn_perm(0x64646464u, raw_[0], 0x04030402u), \
                       __builtin_amdgcn_perm(0x64646464u, raw_[1], 0x04010400u), __builtin_amdgcn_perm(0x64646464u, raw_[1], 0x04030402u)}; \
            X1 = u32x4{__builtin_amdgcn_perm(0x64646464u, raw_[2], 0x04010400u), __builtin_amdgcn_perm(0x64646464u, raw_[2], 0x04030402u), \
                       __builtin_amdgcn_perm(0x64646464u, raw_[3], 0x04010400u), __builtin_amdgcn_perm(0x64646464u, raw_[3], 0x04030402u)}; \
        }                                                                                                   \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, X0), sel1, ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, X1), sel2, ACC, 0, 0, 0);    \
        if (ABL == 5) { ACC[0] += __uint_as_float(RB[0] ^ RB[1]); ACC[1] += __uint_as_float(RB[2] ^ RB[3]); } \
        else {                                                                                              \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) {                                              \
                if (kSplitLut && s_ == 4) {      /* second half of the table reads, into the registers of the first */ \
                    __builtin_amdgcn_sched_barrier(0);                                                      \
                    tl[0] = CLB_LUT(RB[2], 0); tl[1] = CLB_LUT(RB[2], 1); tl[2] = CLB_LUT(RB[2], 2); tl[3] = CLB_LUT(RB[2], 3);   \
                    tl[4] = CLB_LUT(RB[3], 0); tl[5] = CLB_LUT(RB[3], 1); tl[6] = CLB_LUT(RB[3], 2); tl[7] = CLB_LUT(RB[3], 3);   \
                    __builtin_amdgcn_sched_barrier(0);                                                      \
                }                                                                                           \
                const int ti_ = kSplitLut ? 2 * (s_ & 3) : 2 * s_;                                          \
                ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                               \
                    __builtin_bit_cast(f16x8, u32x4{tl[ti_].x, tl[ti_].y, tl[ti_ + 1].x, tl[ti_ + 1].y}),   \
                    __builtin_bit_cast(f16x8, qb[s_]), ACC, 0, 0, 0);                                       \
            }                                                                                               \
        }                                                                                                   \
        CLB_APPROX_PRIO_DOWN                                                                                \
    }
#define CLB_STAGE_E(ACC, INVB, PM, TAG)                                                                     \
    {                                                                                                       \
        f32x4 iq[4];                                                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                       \
            iq[q] = *reinterpret_cast<const f32x4*>(myinv + (INVB) * kStepRows + 8 * q + 4 * h);                                 \
        /* plain v_mul_f32: packed f32 ops (v_pk_mul_f32) do not overlap the MFMAs of the co-resident waves  */ \
        /* (tools/microbench/issue_overlap: 3 per MFMA gap cost 27 cycles, 6 plain multiplies cost 3); the  */ \
        /* empty asm keeps the SLP vectoriser from re-packing them.  ABL 6: the packed form, for comparison */ \
        float v[16];                                                                                        \
        if (ABL != 6) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { v[i] = ACC[i] * iq[i >> 2][i & 3]; asm volatile("" : "+v"(v[i])); } } \
        else                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; i += 2) {    /* v_pk_mul_f32: two rows per instruction */ \
            const f32x2 p_ = f32x2{ACC[i], ACC[i + 1]} * f32x2{iq[i >> 2][i & 3], iq[i >> 2][(i & 3) + 1]}; \
            v[i] = p_[0]; v[i + 1] = p_[1];                                                                 \
        }                                                                                                   \
        if (ROWS) {                                                                                         \
            const __half pmh = *reinterpret_cast<const __half*>(&PM);                                       \
            const float lo = r < T ? __half2float(pmh) - window : __builtin_inff();   /* tokens past T select nothing */ \
            /* per lane (token): bit i of lm = accumulator register i is inside the window; OR over the 32 tokens of */ \
            /* each lane half with 4 DPP steps + 2 readlanes (16 wave-wide ballots cost ~130 scalar instructions per */ \
            /* step, and a wave issues one instruction per ~5.7 cycles whatever its type)                            */ \
            uint32_t lm = 0;                                                                                \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) lm |= !((CELL8 ? v[i] * stp : v[i]) < lo) ? (1u << i) : 0u; \
            lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0xB1, 0xf, 0xf, true);    /* lane ^ 1 */     \
            lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0x4E, 0xf, 0xf, true);    /* lane ^ 2 */     \
            lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0x141, 0xf, 0xf, true);   /* row_half_mirror */ \
            lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0x140, 0xf, 0xf, true);   /* row_mirror */   \
            const uint32_t m0_ = __builtin_amdgcn_readlane(lm, 0) | __builtin_amdgcn_readlane(lm, 16);      \
            const uint32_t m1_ = __builtin_amdgcn_readlane(lm, 32) | __builtin_amdgcn_readlane(lm, 48);     \
            /* register i of lane half h is row (i & 3) + 8 (i >> 2) + 4 h: spread the nibbles            */ \
            uint32_t bits = ((m0_ & 0xFu) | ((m0_ & 0xF0u) << 4) | ((m0_ & 0xF00u) << 8) | ((m0_ & 0xF000u) << 12)) | \
                            (((m1_ & 0xFu) | ((m1_ & 0xF0u) << 4) | ((m1_ & 0xF00u) << 8) | ((m1_ & 0xF000u) << 12)) << 4); \
            if (TAG.rows < kStepRows) bits &= (1u << TAG.rows) - 1u;   /* duplicates of the last row */     \
            const unsigned long long wbits = (unsigned long long)bits << (TAG.base & 32);                   \
            const int wi = TAG.base >> 6;                                                                   \
            wm0 |= wi == 0 ? wbits : 0ull;                                                                  \
            wm1 |= wi == 1 ? wbits : 0ull;                                                                  \
            wm2 |= wi == 2 ? wbits : 0ull;                                                                  \
            wm3 |= wi == 3 ? wbits : 0ull;                                                                  \
            if (TAG.last) {                                                                                 \
                if (lane == 0 && TAG.j >= 0) {                                                              \
                    unsigned long long* o = rmask + (size_t)TAG.j * 4;                                      \
                    o[0] = wm0; o[1] = wm1; o[2] = wm2; o[3] = wm3;                                         \
                }                                                                                           \
                wm0 = wm1 = wm2 = wm3 = 0;                                                                  \
            }                                                                                               \
        } else {                                                                                            \
            float m01 = fmaxf(fmaxf(v[0], v[1]), v[2]);                                                     \
            float m23 = fmaxf(fmaxf(v[3], v[4]), v[5]);                                                     \
            float m45 = fmaxf(fmaxf(v[6], v[7]), v[8]);                                                     \
            float m67 = fmaxf(fmaxf(v[9], v[10]), v[11]);                                                   \
            float m89 = fmaxf(fmaxf(v[12], v[13]), v[14]);                                                  \
            m01 = fmaxf(fmaxf(m01, m23), m45);                                                              \
            m67 = fmaxf(fmaxf(m67, m89), v[15]);                                                            \
            mx = fmaxf(fmaxf(mx, m01), m67);                                                                \
            if (ABL == 10) {     /* the sweep's window test against the RUNNING maximum, mask kept per passage */ \
                const float lo = r < T ? mx - 0.09f : __builtin_inff();                                     \
                uint32_t lm = 0;                                                                            \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) lm |= !(v[i] < lo) ? (1u << i) : 0u;         \
                lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0xB1, 0xf, 0xf, true);              \
                lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0x4E, 0xf, 0xf, true);              \
                lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0x141, 0xf, 0xf, true);             \
                lm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lm, 0x140, 0xf, 0xf, true);             \
                const uint32_t m0_ = __builtin_amdgcn_readlane(lm, 0) | __builtin_amdgcn_readlane(lm, 16);  \
                const uint32_t m1_ = __builtin_amdgcn_readlane(lm, 32) | __builtin_amdgcn_readlane(lm, 48); \
                uint32_t bits = ((m0_ & 0xFu) | ((m0_ & 0xF0u) << 4) | ((m0_ & 0xF00u) << 8) | ((m0_ & 0xF000u) << 12)) | \
                                (((m1_ & 0xFu) | ((m1_ & 0xF0u) << 4) | ((m1_ & 0xF00u) << 8) | ((m1_ & 0xF000u) << 12)) << 4); \
                if (TAG.rows < kStepRows) bits &= (1u << TAG.rows) - 1u;                                    \
                const unsigned long long wbits = (unsigned long long)bits << (TAG.base & 32);               \
                const int wi = TAG.base >> 6;                                                               \
                wm0 |= wi == 0 ? wbits : 0ull; wm1 |= wi == 1 ? wbits : 0ull;                               \
                wm2 |= wi == 2 ? wbits : 0ull; wm3 |= wi == 3 ? wbits : 0ull;                               \
            }                                                                                               \
            if (TAG.last) {                                                                                 \
                if (ABL == 10) {     /* 32 bytes per candidate passage into the (slot-indexed) row-mask buffer */ \
                    if (lane == 0 && TAG.j >= 0) {                                                          \
                        unsigned long long* o = rowmask + ((size_t)b * cand_cap + TAG.j) * 4;               \
                        o[0] = wm0; o[1] = wm1; o[2] = wm2; o[3] = wm3;                                     \
                    }                                                                                       \
                    wm0 = wm1 = wm2 = wm3 = 0;                                                              \
                }                                                                                           \
                mx = max_lane_halves(mx);                                                                   \
                if (CELL8) mx *= stp;                /* cells -> score units, once per passage and token */    \
                const float sum = sum_lanes_0_31(r < T ? mx : 0.f);     /* valid in lanes 16..31 */         \
                if (ABL == 8) { if (sum == 12345.678f) out[0] = mx; }    /* ablation: no result stores */     \
                else if (TAG.j >= 0) {                                                                      \
                    /* into the wave's staging block: passage p = (j - j0) & 15 of the current group of 16 */ \
                    const int p_ = (TAG.j - j0) & 15;                                                       \
                    if (h == 0) *reinterpret_cast<uint16_t*>(mystage + p_ * 64 + 2 * r) = (uint16_t)f32_to_f16_floor(mx); \
                    if (lane == 16) *reinterpret_cast<float*>(mystage + 1024 + 4 * p_) = sum;               \
                    if (p_ == 15 || TAG.last == 3) {    /* the group is full, or the chunk ends: p_ + 1 passages leave */ \
                        const int jb_ = TAG.j - p_;                                                         \
                        __builtin_amdgcn_wave_barrier();                                                    \
                        const u32x4 tk_ = *reinterpret_cast<const u32x4*>(mystage + 16 * lane);             \
                        const float sc_ = *reinterpret_cast<const float*>(mystage + 1024 + 4 * (lane & 15)); \
                        if (lane < 4 * (p_ + 1))                                                            \
                            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(tokmax + ((size_t)b * cand_cap + jb_) * 32) + 16 * lane) = tk_; \
                        if (lane <= p_) out[jb_ + lane] = sc_;                                              \
                        __builtin_amdgcn_wave_barrier();                                                    \
                    }                                                                                       \
                }                                                                                           \
                mx = kNegInf;                                                                               \
            }                                                                                               \
        }                                                                                                   \
    }

        float mx = kNegInf;
        u32x4 rb0, rb1, rb2, xa0, xa1, xa2, xb0, xb1, xb2;
        uint32_t cv0, cv1, cv2, cw0, cw1, cw2;       // cv: as loaded (stage A); cw: the copy stage C dequantises
        uint16_t pm0 = 0, pm1 = 0, pm2 = 0;
        StepTag t0, t1, t2;
        f32x16 acc0, acc1;
        CLB_STAGE_A(rb0, cv0, pm0, t0);
        CLB_STAGE_A(rb1, cv1, pm1, t1);
        CLB_STAGE_G(cv0, xa0, xb0, 0); cw0 = cv0;
        if (!PIPE) {
            while (t0.j >= 0) {
                CLB_STAGE_A(rb2, cv2, pm2, t2);
                CLB_STAGE_G(cv1, xa1, xb1, 1); cw1 = cv1;
                CLB_STAGE_CM(rb0, cw0, xa0, xb0, 0, acc0, 0);
                CLB_STAGE_E(acc0, 0, pm0, t0);
                CLB_STAGE_A(rb0, cv0, pm0, t0);
                CLB_STAGE_G(cv2, xa2, xb2, 2); cw2 = cv2;
                CLB_STAGE_CM(rb1, cw1, xa1, xb1, 1, acc0, 0);
                CLB_STAGE_E(acc0, 0, pm1, t1);
                CLB_STAGE_A(rb1, cv1, pm1, t1);
                CLB_STAGE_G(cv0, xa0, xb0, 0); cw0 = cv0;
                CLB_STAGE_CM(rb2, cw2, xa2, xb2, 2, acc0, 0);
                CLB_STAGE_E(acc0, 0, pm2, t2);
            }
        } else {
            // tp / pmp: tag and stored maximum of the step whose epilogue is pending; the dummy "previous step" of a
            // chunk's first iteration ends a passage nobody stores (j < 0), which also resets mx and the row mask
            StepTag tp; tp.j = -1; tp.rows = kStepRows; tp.last = 1; tp.base = 0;
            uint16_t pmp = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            for (;;) {
                if (t0.j < 0) { CLB_STAGE_E(acc1, 1, pmp, tp); break; }
                CLB_STAGE_A(rb2, cv2, pm2, t2);
                CLB_STAGE_G(cv1, xa1, xb1, 1); cw1 = cv1;
                CLB_STAGE_CM(rb0, cw0, xa0, xb0, 0, acc0, 0);
                CLB_STAGE_E(acc1, 1, pmp, tp);
                tp = t0; pmp = pm0;
                CLB_STAGE_A(rb0, cv0, pm0, t0);
                CLB_STAGE_G(cv2, xa2, xb2, 2); cw2 = cv2;
                CLB_STAGE_CM(rb1, cw1, xa1, xb1, 1, acc1, 1);
                CLB_STAGE_E(acc0, 0, pmp, tp);
                tp = t1; pmp = pm1;
                CLB_STAGE_A(rb1, cv1, pm1, t1);
                CLB_STAGE_G(cv0, xa0, xb0, 0); cw0 = cv0;
                CLB_STAGE_CM(rb2, cw2, xa2, xb2, 2, acc0, 0);
                CLB_STAGE_E(acc1, 1, pmp, tp);
                tp = t2; pmp = pm2;
                if (t0.j < 0) { CLB_STAGE_E(acc0, 0, pmp, tp); break; }
                CLB_STAGE_A(rb2, cv2, pm2, t2);
                CLB_STAGE_G(cv1, xa1, xb1, 1); cw1 = cv1;
                CLB_STAGE_CM(rb0, cw0, xa0, xb0, 0, acc1, 1);
                CLB_STAGE_E(acc0, 0, pmp, tp);
                tp = t0; pmp = pm0;
                CLB_STAGE_A(rb0, cv0, pm0, t0);
                CLB_STAGE_G(cv2, xa2, xb2, 2); cw2 = cv2;
                CLB_STAGE_CM(rb1, cw1, xa1, xb1, 1, acc0, 0);
                CLB_STAGE_E(acc1, 1, pmp, tp);
                tp = t1; pmp = pm1;
                CLB_STAGE_A(rb1, cv1, pm1, t1);
                CLB_STAGE_G(cv0, xa0, xb0, 0); cw0 = cv0;
                CLB_STAGE_CM(rb2, cw2, xa2, xb2, 2, acc1, 1);
                CLB_STAGE_E(acc0, 0, pmp, tp);
                tp = t2; pmp = pm2;
            }
        }
        }   // chunk of 64 passages
#undef CLB_STAGE_A
#undef CLB_STAGE_G
#undef CLB_STAGE_CM
#undef CLB_STAGE_E
#undef CLB_LUT
    }
}

// How often do two consecutive embeddings (passages are stored sorted by code) read the same 128-byte line of the
// fp16 score table, i.e. codes c, c' with c >> 1 == c' >> 1?  Decides pass 1's gather form at index load.
static __global__ __launch_bounds__(256) void code_adjacency_kernel(const uint32_t* __restrict__ codes0, int64_t n,
                                                                   unsigned long long* __restrict__ count) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * 256)
        c += (codes0[i] >> 1) == (codes0[i - 1] >> 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// -------------------------------------------------------------------------------------------------------------
// Selection: tau = k-th largest approximate score; list = { slot : approx[slot] >= tau - 2 eps } in ascending slot
// order.  One workgroup (1024 threads) per query.  thresh[b] = {tau, eps} is kept for inspection.
// -------------------------------------------------------------------------------------------------------------
struct ApproxConsts {
    float cn_max;   // max ||centroid||
    float rn_max;   // sqrt(dim) * max |bucket weight|  (>= ||r|| of every embedding)
    float inv_max;  // max inv_norm
    float rb_max;   // max over the shard's embeddings of ||r'||, r' = the fp16-rounded residual vector
    float dw_rn;    // sqrt(dim) * max_b |fp16(w_b) - w_b|  (>= ||r - r'|| of every embedding)
    float inv_qerr; // max |dequantised inv_norm - inv_norm| = half a quantisation step of the packed code|inv word
    float dc_max;   // max ||c - fp16(c)||: the centroid side of the single-fp16-product score table (0: that kernel is not in use)
};

// The error bound of one query (see the header of this file): eps_t bounds |approx - canonical| of ONE (token, embedding)
// score, eps_sum the same for a passage score (T tokens); `unsafe` = the fp16 score table cannot be trusted for this
// query.  Called by all 1 024 threads of a work-group (two barriers); s_qn / s_dq are two shared floats.
// tscale != nullptr: the batch's score table came from the batched centroid kernel, which measured every token's score range
// (token_range_kernel: {step_t, 1 / step_t, k_t, A_t = max_c |score|}).  The fp16 table's storage error is then 2^-11 A_t.
// cell8: the table holds 8-bit cells requantised FROM that fp16 table, and pass 1 multiplies fp16(Q_t / step_t):
//   cells:  computed score vs canonical (the product bound) + the fp16 rounding (2^-11 A_t) + half a step + the rounding of
//           fma / rint / k_t: 0.5005 * max_t step_t + 8 u A_t
//   Q.r:    Q_t . r - step_t * (fp16(q'_t) . r')  =  (Q_t - step_t fp16(q'_t)) . r' + Q_t . (r - r'),  q'_t = fl(Q_t * (1 / step_t)):
//           ||Q_t - step_t fp16(q'_t)|| <= step_t * ||q'_t - fp16(q'_t)|| (measured: dq8) + 2 u ||Q_t|| (the two roundings of q'_t)
//   MFMA:   the accumulator starts at k_t - 1024 and holds cell + k_t after the two selection products; the partial sums of
//           the 130 accumulations stay below |k_t| + 255 + |q'_t . r'| in cell units = A_t + 255 step_t + qn rn in score
//           units (|k_t| step_t = |lo_t| <= A_t): 2 * 130 * u of that, and 1280 u step_t for the start value's own rounding
// and the query is unsafe when a token's range is not a pair of finite numbers or its scaled operand could leave the fp16 range.
struct QueryBound { float eps_t, eps_sum; bool unsafe; };
__device__ __forceinline__ QueryBound query_bound(const float* __restrict__ Q, int b, int T, const ApproxConsts& ac,
                                                  float* s_qn, float* s_dq, const float4* __restrict__ tscale = nullptr,
                                                  float* s_aux = nullptr /* 4 shared floats when tscale is given */,
                                                  bool cell8 = false) {
    const int tid = threadIdx.x;
    // qn = max_t ||Q_t||  (plain fp32 sum, upper-bounded by the 1.001 factor below)
    if (tid == 0) { *s_qn = 0.f; *s_dq = 0.f; }
    if (tscale && tid < 4) s_aux[tid] = 0.f;     // [0] max_t step_t ||q'_t - fp16(q'_t)||, [1] max_t step_t, [2] bad-token flag, [3] max_t A_t
    __syncthreads();
    {   // 32 threads per token, one float4 each (T <= 32 in this mode).  dq = max_t ||Q_t - fp16(Q_t)||: what the
        // fp16 query operand of pass 1 really loses (at most 2^-12 ||Q_t|| in the normal range)
        const int t = tid >> 5, part = tid & 31;
        float a = 0.f, dd = 0.f, d8 = 0.f;
        if (t < T) {
            const float4 v = *reinterpret_cast<const float4*>(Q + ((size_t)b * T + t) * kDim + 4 * part);
            a = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
            const float dx = v.x - round_f16(v.x), dy = v.y - round_f16(v.y);
            const float dz = v.z - round_f16(v.z), dw = v.w - round_f16(v.w);
            dd = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, dw * dw)));
            if (tscale && cell8) {
                const float4 ts = tscale[(size_t)b * 32 + t];
                const float sx = v.x * ts.y, sy = v.y * ts.y, sz = v.z * ts.y, sw = v.w * ts.y;   // pass 1's operand before its fp16 rounding
                const float ex = sx - round_f16(sx), ey = sy - round_f16(sy), ez = sz - round_f16(sz), ew = sw - round_f16(sw);
                d8 = fmaf(ex, ex, fmaf(ey, ey, fmaf(ez, ez, ew * ew)));
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); dd += __shfl_xor(dd, o, 64); d8 += __shfl_xor(d8, o, 64); }
        if (t < T && part == 0) {
            atomicMax(reinterpret_cast<unsigned int*>(s_qn), __float_as_uint(sqrtf(a) * 1.001f));
            atomicMax(reinterpret_cast<unsigned int*>(s_dq), __float_as_uint(sqrtf(dd) * 1.001f));
            if (tscale) {
                const float4 ts = tscale[(size_t)b * 32 + t];
                atomicMax(reinterpret_cast<unsigned int*>(s_aux), __float_as_uint(ts.x * sqrtf(d8) * 1.001f));
                atomicMax(reinterpret_cast<unsigned int*>(s_aux + 1), __float_as_uint(ts.x));
                atomicMax(reinterpret_cast<unsigned int*>(s_aux + 3), __float_as_uint(ts.w));
                // no usable range (token_range_kernel wrote zeros), or a scaled operand |q'_t| <= ||Q_t|| / step_t that could
                // leave the fp16 range
                if (!(ts.x > 0.f) || !(sqrtf(a) * ts.y < 3.0e4f)) s_aux[2] = 1.f;
            }
        }
    }
    __syncthreads();
    const float u = 5.9604645e-08f;  // 2^-24
    const float qn = *s_qn;
    // fp16 storage: relative 2^-11 in the normal range, absolute 2^-25 below it (subnormal spacing 2^-24)
    const float e_cells = kEpsSafety * centroid_product_bound(qn, *s_dq, ac.cn_max, ac.dc_max) + 4.8828125e-04f * qn * ac.cn_max + 2.9802322e-08f;
    // Q.r:  sum_d (Q_d w_d - Q'_d w'_d) = dQ . r' + Q . (r - r')  with Q', w' the fp16 operands (their products are exact
    // in fp32): <= dq * max ||r'|| + qn * sqrt(dim) * max_b |w_b - w'_b|, both factors measured (dq here, the
    // other two at index load) instead of the generic 2^-9 relative bounds; plus the fp32 accumulation of the MFMA
    float e_qr = 1.001f * (*s_dq * ac.rb_max + qn * ac.dw_rn) + 2.f * 128.f * u * qn * ac.rn_max;
    float e_tab = e_cells;
    bool bad8 = false;
    if (tscale) {
        // the measured magnitude of the computed scores (never above what the norms allow; NaN-safe: fminf returns the number)
        const float amax = s_aux[2] != 0.f ? qn * ac.cn_max : fminf(s_aux[3] * 1.0001f, qn * ac.cn_max);
        const float e_f16 = kEpsSafety * centroid_product_bound(qn, *s_dq, ac.cn_max, ac.dc_max) + 4.8828125e-04f * amax + 2.9802322e-08f;
        e_tab = e_f16;
        if (cell8) {
            const float dq8 = s_aux[0], step_max = s_aux[1];
            e_tab = e_f16 + 0.5005f * step_max + 8.f * u * amax;
            e_qr = 1.001f * ((dq8 + 2.f * u * qn) * ac.rb_max + qn * ac.dw_rn) +
                   2.f * 130.f * u * (amax + 255.f * step_max + qn * ac.rn_max) + 1280.f * u * step_max;
            bad8 = s_aux[2] != 0.f;
        }
    }
    // the packed inv_norm is off by at most inv_qerr: it scales P = X + Q.r, |P| <= qn (cn + rn), and the other terms
    const float eps_t = (ac.inv_max + ac.inv_qerr) * (e_tab + e_qr) + 1.01f * ac.inv_qerr * qn * (ac.cn_max + ac.rn_max) + 332.f * u * qn;
    QueryBound r;
    r.eps_t = eps_t;
    // Guard of the fp16 score table: its entries are bounded by qn * cn and must stay finite in fp16 (max 65504);
    // the bound itself must be a finite number.  A query that fails either test (un-normalised or non-finite Q,
    // huge centroid norms) is not pre-filtered at all: every candidate is listed and every row selected, i.e. it is
    // scored by the exact kernel alone, exactly as in mode 0.  (NaN-safe: written with negated comparisons.)
    r.unsafe = !(qn * ac.cn_max < 3.0e4f) || !(qn < 6.0e4f) /* the fp16 query operand */ || !(eps_t < 1.0e30f) || bad8;
    r.eps_sum = kEpsSafety * ((float)T * eps_t + 2.f * (float)T * (float)T * u * qn);
    return r;
}

static __global__ __launch_bounds__(1024) void select_margin_kernel(const float* __restrict__ scores,
                                                                   const int* __restrict__ ncand,
                                                                   const float* __restrict__ Q, int T, int k,
                                                                   size_t cand_cap, ApproxConsts ac,
                                                                   int* __restrict__ list, int* __restrict__ nlist,
                                                                   float* __restrict__ thresh,
                                                                   float* __restrict__ eps_pair,
                                                                   const float* __restrict__ tau_in = nullptr,
                                                                   int coarse_tau = 0,
                                                                   const float4* __restrict__ tscale = nullptr,
                                                                   int cell8 = 0, int dbg_stop = 0 /* tuning builds: leave after phase N */) {
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ float s_aux[4];
    __shared__ int sh_scan[16];
    __shared__ int sh_big[2][8][16];
    __shared__ uint32_t s_prefix, s_kmin, s_kmax;
    __shared__ int s_remaining, s_run;
    __shared__ float s_qn, s_dq;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = ncand[b];
    const float* sc = scores + (size_t)b * cand_cap;
    int* lst = list + (size_t)b * cand_cap;
    if (tid == 0) { s_prefix = 0u; s_remaining = n < k ? n : k; s_run = 0; }
    // thread t owns the contiguous slots [t*chunk, (t+1)*chunk); up to kSelCache order keys stay in registers for the
    // radix passes and the compaction (the approximate scores are read once).  The loads are issued BEFORE the bound is
    // computed: they fly during its two barriers.  (Ownership c * 1024 + t -- coalesced loads -- was tried in round 6 and is
    // SLOWER, 37 against 31 us: every wave then walks all of the query's 4-KB pages.)
    const int chunk = (n + 1023) >> 10;
    const int i0 = tid * chunk;
    const bool cached = chunk <= kSelCache;   // uniform over the block
    uint32_t ckey[kSelCache];
    if (cached) {
#pragma unroll
        for (int c = 0; c < kSelCache; ++c) ckey[c] = (c < chunk && i0 + c < n) ? f32_order_key(sc[i0 + c]) : 0u;
    }
    const QueryBound qb = query_bound(Q, b, T, ac, &s_qn, &s_dq, tscale, s_aux, cell8 != 0);
    float thr = kNegInf, tau_f = kNegInf, eps = 0.f;
    const bool unsafe = qb.unsafe;
    if (tid == 0) eps_pair[b] = unsafe ? __builtin_inff() : kEpsSafety * qb.eps_t;
    if (kAblations && dbg_stop == 1) return;
    if (kAblations && dbg_stop == 2) {     // (the keys must be used, or the loads go away)
        uint32_t x = 0;
        if (cached) {
#pragma unroll
            for (int c = 0; c < kSelCache; ++c) x ^= ckey[c];
        }
        if (x == 0x12345u) nlist[b] = 0;
        return;
    }
#define CLB_SEL_FOR_EACH(...)                                                                   \
    if (cached) {                                                                               \
        _Pragma("unroll") for (int c = 0; c < kSelCache; ++c) {                                 \
            if (c >= chunk) break;                                                              \
            const int i = i0 + c;                                                               \
            const bool valid = i < n;                                                           \
            const uint32_t key = ckey[c];                                                       \
            __VA_ARGS__                                                                         \
        }                                                                                       \
    } else {   /* too many for the registers: strided ownership, so that the re-reads are coalesced */ \
        _Pragma("unroll 4") for (int c = 0; c < chunk; ++c) {                                   \
            const int i = c * 1024 + tid;                                                       \
            const bool valid = i < n;                                                           \
            const uint32_t key = valid ? f32_order_key(sc[i]) : 0u;                             \
            __VA_ARGS__                                                                         \
        }                                                                                       \
    }
    if (unsafe) {
        eps = __builtin_inff();          // thr stays -inf: everything is listed
    } else if (tau_in) {
        // sharded search, phase 2: tau is the GLOBAL k-th approximate score over all shards (global_tau_kernel); -inf
        // means fewer than k candidates exist anywhere, i.e. everything is listed
        eps = qb.eps_sum;
        tau_f = tau_in[b];
        thr = tau_f == kNegInf ? kNegInf : tau_f - 2.f * eps;
    } else if (n > k) {
        // coarse_tau (the unsharded search): two digit passes instead of four.  Any LOWER bound of the k-th approximate
        // score keeps the proof of the two-pass mode (it only lists more); the lower edge of the k-th key's 16-bit bin
        // is below it by at most 2^-16 of the score range (~5e-4 here against 2 eps ~ 0.09), and six block-wide
        // barriers of the 32-us kernel go away.  The sharded protocol publishes EXACT local thresholds and keeps all passes
        if (coarse_tau) { CLB_RADIX_SELECT_N(2) } else { CLB_RADIX_SELECT() }
        eps = qb.eps_sum;
        tau_f = f32_from_order_key(s_prefix);
        thr = tau_f - 2.f * eps;
    }
    if (kAblations && dbg_stop == 3) { if (tid == 0 && thr == 12345.f) nlist[b] = 0; return; }
    // ordered compaction of the slots with approx >= thr (all of them when n <= k): one block-wide scan of the
    // per-thread counts places every chunk in order
    if (!cached) {
        // large inputs: the same compaction in blocks of 8 x 1024 consecutive slots.  A thread's eight loads are in
        // flight together and a block costs ONE barrier (the wave counts alternate between two LDS buffers; the running
        // total lives in a register of every thread) -- with one load and three barriers per 1024 slots the load
        // latency of every block was exposed: 0.37 ms per batch at 96 k candidates per query.
        const int lane = tid & 63, wave = tid >> 6;
        int run = 0, it = 0;
        for (int base = 0; base < n; base += 8192, ++it) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * 1024 + tid;
                v[j] = i < n ? sc[i] : 0.f;
            }
            int pre[8];
            bool take[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * 1024 + tid;
                take[j] = i < n && !(v[j] < thr);   // NaN (unsafe query) is listed
                const unsigned long long m = __builtin_amdgcn_ballot_w64(take[j]);
                pre[j] = (int)__popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) sh_big[it & 1][j][wave] = (int)__popcll(m);
            }
            __syncthreads();
            int offs = run;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = lane < 16 ? sh_big[it & 1][j][lane] : 0;   // lanes 0..15: the 16 waves' counts of sub-block j
                int x = c;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const int y = __shfl_up(x, o, 64);
                    if (lane >= o) x += y;
                }
                const int wbase = __shfl(x - c, wave, 64);
                const int tot = __shfl(x, 15, 64);
                if (take[j]) lst[offs + wbase + pre[j]] = base + j * 1024 + tid;
                offs += tot;
            }
            run = offs;
        }
        if (tid == 0) s_run = run;
    } else {
        int cnt = 0;
        CLB_SEL_FOR_EACH((void)i; cnt += valid && !(f32_from_order_key(key) < thr);)
        const int lane = tid & 63, wave = tid >> 6;
        int xv = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(xv, o, 64);
            if (lane >= o) xv += y;
        }
        if (lane == 63) sh_scan[wave] = xv;
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) {
            const int sv = sh_scan[w2];
            if (w2 < wave) wbase += sv;
            tot += sv;
        }
        int pos = wbase + xv - cnt;
        CLB_SEL_FOR_EACH(if (valid && !(f32_from_order_key(key) < thr)) lst[pos++] = i;)
        if (tid == 0) s_run = tot;
        __syncthreads();
    }
#undef CLB_SEL_FOR_EACH
    if (tid == 0) {
        nlist[b] = s_run;
        thresh[2 * b] = tau_f;
        thresh[2 * b + 1] = eps;
    }
}

// -------------------------------------------------------------------------------------------------------------
// Wide selection: the same tau / list as select_margin_kernel, by kWideBlocks work-groups per query.  One CU needs
// 0.26 ms for the five radix passes over the ~96 k candidates a query has on a 10 M-passage shard (issue-bound);
// here every pass is its own launch of grid (kWideBlocks, B): the work-groups add the histogram of their segment to
// the query's global one.  No work-group waits for another and nothing is fenced: the launch boundary orders the
// passes, and every work-group of a launch re-derives the selection state (prefix, rank, digit position) from the
// finished histograms of the earlier passes (wide_replay: at most four 256-bin picks).  A first version let the last
// work-group to arrive pick the digit behind a device-scope fence: 8 192 waves x 7 launches of L2 write-backs made it
// 1.1 ms per batch.  Launch order: wide_minmax, wide_hist x 4 (a pass a query does not need returns at once),
// wide_count, wide_emit.  `WideSel` (one per query) must be zero before wide_minmax.
// -------------------------------------------------------------------------------------------------------------
constexpr int kWideBlocks = 16;
struct WideSel {
    uint32_t kmax, kmin_inv;      // max key, max of ~key
    int unsafe;
    float eps_sum;
    int cnt[kWideBlocks];
    int hist[4][256];
};
struct WideState {
    uint32_t prefix;
    int remaining, shift, width;
    bool done;
    float thr, tau, eps;
};

// slots [lo, hi) of work-group g: segments of whole 1 024-slot blocks
__device__ __forceinline__ void wide_segment(int n, int g, int& lo, int& hi) {
    const int seg = ((n + kWideBlocks * 1024 - 1) / (kWideBlocks * 1024)) * 1024;
    lo = g * seg;
    hi = lo + seg < n ? lo + seg : n;
    if (lo > n) lo = n;
}

// The selection state after the first `upto` radix passes, from what the earlier launches left in `w`.  Called by all
// threads of the work-group; hist = 256 ints of LDS (16-byte aligned), s_prefix / s_remaining shared words.
__device__ __forceinline__ WideState wide_replay(const WideSel& w, int n, int k, const float* __restrict__ tau_in, int b,
                                                 int upto, int* hist, uint32_t* s_prefix, int* s_remaining) {
    const int tid = threadIdx.x;
    WideState st;
    st.prefix = 0u; st.remaining = 0; st.shift = 0; st.width = 0; st.done = true;
    st.thr = kNegInf; st.tau = kNegInf; st.eps = 0.f;
    if (w.unsafe) {
        st.eps = __builtin_inff();                        // everything is listed
    } else if (tau_in) {                                  // sharded search, phase 2: the global threshold
        st.eps = w.eps_sum;
        st.tau = tau_in[b];
        st.thr = st.tau == kNegInf ? kNegInf : st.tau - 2.f * st.eps;
    } else if (n > k) {
        st.eps = w.eps_sum;
        const uint32_t gmax = w.kmax, gmin = ~w.kmin_inv;
        const uint32_t diff = gmin ^ gmax;
        if (diff == 0u) {
            st.prefix = gmax;
            st.tau = f32_from_order_key(gmax);
            st.thr = st.tau - 2.f * st.eps;
        } else {
            const int top = 31 - __clz((int)diff);
            st.shift = top > 7 ? top - 7 : 0;
            st.width = top - st.shift + 1;
            st.prefix = top == 31 ? 0u : (gmax & (0xffffffffu << (top + 1)));
            st.remaining = k;
            st.done = false;
            for (int p = 0; p < upto && !st.done; ++p) {  // uniform over the work-group
                __syncthreads();
                if (tid < 256) hist[tid] = w.hist[p][tid];
                __syncthreads();
                if (tid < 64) radix_pick(hist, st.remaining, st.prefix, st.shift, s_prefix, s_remaining);
                __syncthreads();
                st.prefix = *s_prefix;
                st.remaining = *s_remaining;
                if (st.shift == 0) {
                    st.tau = f32_from_order_key(st.prefix);
                    st.thr = st.tau - 2.f * st.eps;
                    st.done = true;
                } else {
                    const int ns = st.shift > 8 ? st.shift - 8 : 0;
                    st.width = st.shift - ns;
                    st.shift = ns;
                }
            }
            __syncthreads();
        }
    }
    return st;
}

// grid = (kWideBlocks, B), block = 1024
static __global__ __launch_bounds__(1024) void wide_minmax_kernel(const float* __restrict__ scores,
                                                                 const int* __restrict__ ncand,
                                                                 const float* __restrict__ Q, int T, size_t cand_cap,
                                                                 ApproxConsts ac, WideSel* __restrict__ wsel,
                                                                 float* __restrict__ eps_pair,
                                                                 const float4* __restrict__ tscale = nullptr,
                                                                 int cell8 = 0) {
    __shared__ uint32_t s_kmin, s_kmax;
    __shared__ float s_qn, s_dq, s_aux[4];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int n = ncand[b];
    const float* sc = scores + (size_t)b * cand_cap;
    WideSel& w = wsel[b];
    if (tid == 0) { s_kmin = 0xffffffffu; s_kmax = 0u; }
    __syncthreads();
    if (g == 0) {                                         // uniform over the work-group
        const QueryBound qb = query_bound(Q, b, T, ac, &s_qn, &s_dq, tscale, s_aux, cell8 != 0);
        if (tid == 0) {
            eps_pair[b] = qb.unsafe ? __builtin_inff() : kEpsSafety * qb.eps_t;
            w.unsafe = qb.unsafe ? 1 : 0;
            w.eps_sum = qb.eps_sum;
        }
    }
    int lo, hi;
    wide_segment(n, g, lo, hi);
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll 4
    for (int i = lo + tid; i < hi; i += 1024) {
        const uint32_t key = f32_order_key(sc[i]);
        kmin = key < kmin ? key : kmin;
        kmax = key > kmax ? key : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(kmin, o, 64), c = __shfl_xor(kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
    }
    if ((tid & 63) == 0) { atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); }
    __syncthreads();
    if (tid == 0 && lo < hi) { atomicMax(&w.kmax, s_kmax); atomicMax(&w.kmin_inv, ~s_kmin); }
}

// one radix pass; grid = (kWideBlocks, B), block = 1024
static __global__ __launch_bounds__(1024) void wide_hist_kernel(const float* __restrict__ scores,
                                                               const int* __restrict__ ncand, int k, size_t cand_cap,
                                                               WideSel* __restrict__ wsel, int pass) {
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_remaining;
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    WideSel& w = wsel[b];
    const int n = ncand[b];
    const WideState st = wide_replay(w, n, k, nullptr, b, pass, hist, &s_prefix, &s_remaining);
    if (st.done) return;                                  // uniform over the work-group
    const float* sc = scores + (size_t)b * cand_cap;
    const uint32_t prefix = st.prefix;
    const int shift = st.shift, width = st.width;
    const uint32_t himask = shift + width >= 32 ? 0u : (0xffffffffu << (shift + width));
    const uint32_t bmask = (1u << width) - 1u;
    __syncthreads();
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    int lo, hi;
    wide_segment(n, g, lo, hi);
#pragma unroll 4
    for (int i0 = lo; i0 < hi; i0 += 1024) {              // whole waves take part in the aggregated add
        const int i = i0 + tid;
        const bool valid = i < hi;
        const uint32_t key = valid ? f32_order_key(sc[i]) : 0u;
        hist_add_aggregated(hist, (key >> shift) & bmask, valid && (key & himask) == prefix);
    }
    __syncthreads();
    if (tid < 256 && hist[tid]) atomicAdd(&w.hist[pass][tid], hist[tid]);
}

// how many slots of the segment are listed; grid = (kWideBlocks, B), block = 1024
static __global__ __launch_bounds__(1024) void wide_count_kernel(const float* __restrict__ scores,
                                                                const int* __restrict__ ncand, int k, size_t cand_cap,
                                                                WideSel* __restrict__ wsel,
                                                                const float* __restrict__ tau_in) {
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_remaining, s_cnt;
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    WideSel& w = wsel[b];
    const int n = ncand[b];
    const float* sc = scores + (size_t)b * cand_cap;
    const float thr = wide_replay(w, n, k, tau_in, b, 4, hist, &s_prefix, &s_remaining).thr;
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int lo, hi;
    wide_segment(n, g, lo, hi);
    int c = 0;
#pragma unroll 4
    for (int i = lo + tid; i < hi; i += 1024) c += !(sc[i] < thr) ? 1 : 0;      // NaN (unsafe query) is listed
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((tid & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (tid == 0) w.cnt[g] = s_cnt;
}

// ordered compaction: the segment's listed slots follow those of the segments before it; grid = (kWideBlocks, B)
static __global__ __launch_bounds__(1024) void wide_emit_kernel(const float* __restrict__ scores,
                                                               const int* __restrict__ ncand, int k, size_t cand_cap,
                                                               const WideSel* __restrict__ wsel,
                                                               const float* __restrict__ tau_in, int* __restrict__ list,
                                                               int* __restrict__ nlist, float* __restrict__ thresh) {
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_remaining;
    __shared__ int sh_big[2][8][16];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const WideSel& w = wsel[b];
    const int n = ncand[b];
    const float* sc = scores + (size_t)b * cand_cap;
    int* lst = list + (size_t)b * cand_cap;
    const WideState st = wide_replay(w, n, k, tau_in, b, 4, hist, &s_prefix, &s_remaining);
    const float thr = st.thr;
    int run = 0, total = 0;
#pragma unroll
    for (int j = 0; j < kWideBlocks; ++j) {
        const int c = w.cnt[j];
        run += j < g ? c : 0;
        total += c;
    }
    if (g == 0 && tid == 0) {
        nlist[b] = total;
        thresh[2 * b] = st.tau;
        thresh[2 * b + 1] = st.eps;
    }
    int lo, hi;
    wide_segment(n, g, lo, hi);
    const int lane = tid & 63, wave = tid >> 6;
    int it = 0;
    for (int base = lo; base < hi; base += 8192, ++it) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = base + j * 1024 + tid;
            v[j] = i < hi ? sc[i] : 0.f;
        }
        int pre[8];
        bool take[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = base + j * 1024 + tid;
            take[j] = i < hi && !(v[j] < thr);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(take[j]);
            pre[j] = (int)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) sh_big[it & 1][j][wave] = (int)__popcll(m);
        }
        __syncthreads();
        int offs = run;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane < 16 ? sh_big[it & 1][j][lane] : 0;
            int x = c;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const int y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            const int wbase = __shfl(x - c, wave, 64);
            const int tot = __shfl(x, 15, 64);
            if (take[j]) lst[offs + wbase + pre[j]] = base + j * 1024 + tid;
            offs += tot;
        }
        run = offs;
    }
}

// -------------------------------------------------------------------------------------------------------------
// Sharded search (one shard per GPU).  A shard has to hand the merge ITS k best passages, so on its own it must
// cut at the shard-local k-th approximate score -- with 8 shards that lists (and re-scores exactly) 8x more
// passages than the global top-k needs.  Exchange between the two phases: every shard publishes its k largest
// approximate scores per query (local_top_kernel); the k-th largest of their union IS the global k-th approximate
// score (the global top-k by approximate score is a subset of the union of the local ones), and the proof of the
// two-pass mode then holds with that tau: every member of the exact global top-k has approx >= tau - 2 eps.
// -------------------------------------------------------------------------------------------------------------
// out[b][0..k): the approximate scores >= the local tau (at most k of them; ties beyond k are equal values and can be
// dropped), padded with -inf.  grid = B, block = 1024.
static __global__ __launch_bounds__(1024) void local_top_kernel(const float* __restrict__ scores,
                                                               const int* __restrict__ list,
                                                               const int* __restrict__ nlist,
                                                               const float* __restrict__ thresh, int k,
                                                               size_t cand_cap, float* __restrict__ out) {
    __shared__ int s_cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const float tau = thresh[2 * b];                       // -inf when the shard has at most k candidates
    const float* sc = scores + (size_t)b * cand_cap;
    const int* lst = list + (size_t)b * cand_cap;
    float* o = out + (size_t)b * k;
    const int n = nlist[b];
    for (int i = tid; i < n; i += 1024) {
        const float v = sc[lst[i]];
        if (v >= tau) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < k) o[pos] = v;
        }
    }
    __syncthreads();
    const int cnt = s_cnt < k ? s_cnt : k;
    for (int i = cnt + tid; i < k; i += 1024) o[i] = kNegInf;
}

// tau_glob[b] = the k-th largest of the n_shards * k gathered values all_top[shard][b][i] (-inf when fewer than k
// finite values exist).  grid = B, block = 1024.
static __global__ __launch_bounds__(1024) void global_tau_kernel(const float* __restrict__ all_top, int n_shards,
                                                                int B, int k, float* __restrict__ tau_glob) {
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ uint32_t s_prefix, s_kmin, s_kmax;
    __shared__ int s_remaining;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = n_shards * k;
    const int chunk = (n + 1023) >> 10;
#define CLB_SEL_FOR_EACH(...)                                                                   \
    for (int c = 0; c < chunk; ++c) {                                                           \
        const int i_ = c * 1024 + tid;                                                          \
        const bool valid = i_ < n;                                                              \
        const uint32_t key = valid ? f32_order_key(all_top[((size_t)(i_ / k) * B + b) * k + (i_ % k)]) : 0u; \
        __VA_ARGS__                                                                             \
    }
    if (tid == 0) { s_prefix = 0u; s_remaining = k; }
    __syncthreads();
    CLB_RADIX_SELECT()
#undef CLB_SEL_FOR_EACH
    if (tid == 0) tau_glob[b] = f32_from_order_key(s_prefix);
}

static __global__ void max_abs_kernel(const float* __restrict__ v, int n, unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(v[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}
// max_b |w_b - fp16(w_b)| with the device's own conversion (the one the LUT of pass 1 is built with); a weight beyond
// the fp16 range makes the bound infinite (such an index is searched exactly)
static __global__ void max_f16_err_kernel(const float* __restrict__ v, int n, unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float w = v[i];
        m = fmaxf(m, fabsf(w) < 6.0e4f ? fabsf(round_f16(w) - w) : __builtin_inff());
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(m));
}
static __global__ void max_row_norm_kernel(const float* __restrict__ C, int K, unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int d = 0; d < kDim; ++d) a = fmaf(C[(size_t)c * kDim + d], C[(size_t)c * kDim + d], a);
        m = fmaxf(m, sqrtf(a) * 1.001f);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

// index-load: the packed code | inv_norm words of pass 1 + the constants of the error bound.
// d_codeinv == nullptr: constants only (index build).  On return *inv_lo / *inv_step describe the quantisation.
inline int build_approx_tables(hipStream_t st, const float* dC, const float* dW, const uint32_t* dCodes0,
                               const uint8_t* dRes, int64_t n_emb, int K, uint32_t* d_codeinv /*null: constants only*/,
                               int cbits, int n_weights, ApproxConsts* out, float* inv_lo = nullptr,
                               float* inv_step = nullptr) {
    DevBuf tmp, inv;
    CLB_TRY(tmp.alloc(6 * sizeof(unsigned int)));
    CLB_HIP(hipMemsetAsync(tmp.p, 0, 4 * sizeof(unsigned int), st));
    CLB_HIP(hipMemsetAsync(static_cast<char*>(tmp.p) + 5 * sizeof(unsigned int), 0, sizeof(unsigned int), st));
    CLB_HIP(hipMemsetAsync(static_cast<char*>(tmp.p) + 4 * sizeof(unsigned int), 0x7f, sizeof(unsigned int), st));  // +huge
    unsigned int* bits = tmp.as<unsigned int>();
    const bool pack = n_emb > 0 && d_codeinv;
    if (pack) {
        CLB_TRY(inv.alloc(sizeof(float) * n_emb));
        const int grid = (int)std::min<int64_t>(4096, (n_emb + 63) / 64);
        hipLaunchKernelGGL(inv_norm_kernel, dim3(grid), dim3(256), 0, st, dC, dW, dCodes0, dRes, n_emb, inv.as<float>(), bits,
                           bits + 3, bits + 4);
    }
    hipLaunchKernelGGL(max_abs_kernel, dim3(1), dim3(64), 0, st, dW, n_weights, bits + 1);
    hipLaunchKernelGGL(max_f16_err_kernel, dim3(1), dim3(64), 0, st, dW, n_weights, bits + 5);
    hipLaunchKernelGGL(max_row_norm_kernel, dim3(std::max(1, std::min(1024, K / 256))), dim3(256), 0, st, dC, K, bits + 2);
    DevBuf dcb;
    CLB_TRY(dcb.alloc(sizeof(unsigned int)));
    CLB_HIP(hipMemsetAsync(dcb.p, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(max_row_f16_err_kernel, dim3(std::max(1, std::min(1024, K / 4))), dim3(256), 0, st, dC, K, dcb.as<unsigned int>());
    unsigned int hdc = 0;
    CLB_HIP(hipMemcpyAsync(&hdc, dcb.p, sizeof hdc, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipGetLastError());
    unsigned int h[6];
    CLB_HIP(hipMemcpyAsync(h, bits, sizeof h, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    float f[6];
    memcpy(f, h, sizeof f);
    out->inv_max = f[0] * 1.0001f;
    out->rn_max = sqrtf((float)kDim) * f[1] * 1.0001f;
    out->cn_max = f[2];
    const float dw = f[5];      // max_b |w_b - fp16(w_b)|
    out->dw_rn = sqrtf((float)kDim) * dw * 1.0001f;
    // no inv_norm pass (constants only): fall back to the generic bound ||r'|| <= sqrt(dim) * max |w| * (1 + 2^-11)
    out->rb_max = pack ? sqrtf(f[3]) * 1.0001f : out->rn_max * 1.0005f;
    out->inv_qerr = 0.f;
    memcpy(&out->dc_max, &hdc, sizeof(float));       // the caller zeroes it when the single-product centroid kernel is off
    if (pack) {
        // at most 20 bits for inv_norm: every level is then an exact float (a 26-bit qmax rounds UP as a float and the
        // top level would overflow the word) and the step is already far below the other error terms
        const uint32_t qmax = (1u << std::min(32 - cbits, 20)) - 1u;
        const float lo = f[4], hi = f[0];
        const float step = hi > lo ? (hi - lo) / (float)qmax : 0.f;
        // half a step plus the rounding of the (de)quantisation arithmetic itself
        out->inv_qerr = 0.5f * step * 1.001f + 4.f * 5.9604645e-08f * hi;
        hipLaunchKernelGGL(pack_code_inv_kernel, dim3((unsigned)((n_emb + 255) / 256)), dim3(256), 0, st, dCodes0,
                           inv.as<float>(), n_emb, cbits, lo, step > 0.f ? 1.0f / step : 0.f, qmax, d_codeinv);
        CLB_HIP(hipGetLastError());
        CLB_HIP(hipStreamSynchronize(st));
        if (inv_lo) *inv_lo = lo;
        if (inv_step) *inv_step = step;
    }
    return CLB_OK;
}

}  // namespace clb
