// encoder_kernels.hpp -- BERT encoder forward + ColBERT projection for gfx950 (fp32, f32 MFMA GEMMs).
// Replaces `doc(bert, linear, ids, mask)` (src/modelling/checkpoint.jl:21-25): Transformers.jl's HGFBertModel
// (post-LN BERT: embeddings -> N x [self-attention, add&norm, GELU feed-forward, add&norm]) followed by
// Layers.Dense(hidden -> dim).  The arithmetic lives in un-vendored Transformers.jl / NeuralAttentionlib (parity
// unpinned); it is restated here from the published BERT definition the HuggingFace checkpoint format
// implies, and tested against an independent fp32 reference of the same definition.
#pragma once
#include "common.hpp"
#include "search_kernels.hpp"

namespace clb {

// -------------------------------------------------------------------------------------------------------------
// Strided, batched fp32 GEMM on v_mfma_f32_32x32x2_f32:  C[z](m, n) = sum_k A[z](m, k) * B[z](k, n)  (+ epilogue)
//   A element (m, k) at A + z_off_a + m*lda + k           (k contiguous)
//   B element (k, n) at B + z_off_b + n*ldb_n + k*ldb_k   (torch Linear weight [out][in]: ldb_n = in, ldb_k = 1)
//   C element (m, n) at C + z_off_c + m*ldc + n
// batch z = (zo, zi) with zi < zi_count: offset = zo*stride_o + zi*stride_i for each operand (documents x heads).
// Workgroup = 4 waves = 64 x 64 output tile (wave (wr, wc) owns a 32 x 32 block), K-step 32 staged through LDS.
// Epilogue: EPI_BIAS adds bias[n]; EPI_GELU applies the erf GELU after the bias; EPI_RESID adds R(m, n) (same
// layout as C); scale multiplies the accumulator first (attention scores).
// -------------------------------------------------------------------------------------------------------------
enum { EPI_BIAS = 1, EPI_GELU = 2, EPI_RESID = 4 };

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; const float* R;
    int M, N, K;
    int64_t lda, ldb_n, ldb_k, ldc;
    int zi_count;
    int64_t a_so, a_si, b_so, b_si, c_so, c_si;
    float scale;
    int epi;
    int ksplit;        // > 1 (tiled kernel, unbatched only): blockIdx.z = K slice; raw partial sums go to C + z*M*N
};

constexpr int kGemmKT = 32;           // K per staged step
constexpr int kGemmLd = kGemmKT + 1;  // LDS row stride (floats): stride-33 rows -> conflict-free column reads

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

static __global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[64 * kGemmLd];
    __shared__ float Bs[64 * kGemmLd];
    const int z = blockIdx.z;
    const int zo = z / g.zi_count, zi = z % g.zi_count;
    const float* A = g.A + zo * g.a_so + zi * g.a_si;
    const float* B = g.B + zo * g.b_so + zi * g.b_si;
    float* C = g.C + zo * g.c_so + zi * g.c_si;
    const float* R = g.R ? g.R + zo * g.c_so + zi * g.c_si : nullptr;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += kGemmKT) {
        // stage A[64][32] and B[64 (n)][32 (k)]: thread t loads row t/4 (two passes of 32 rows... 8 elements each)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int e = tid + 256 * p;          // 0..2047 = 64 rows x 32 k
            const int row = e >> 5, kk = e & 31;
            const int m = m0 + row, n = n0 + row, k = k0 + kk;
            As[row * kGemmLd + kk] = (m < g.M && k < g.K) ? A[(int64_t)m * g.lda + k] : 0.f;
            Bs[row * kGemmLd + kk] = (n < g.N && k < g.K) ? B[(int64_t)n * g.ldb_n + (int64_t)k * g.ldb_k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kGemmKT / 2; ++s) {
            const float a = As[(wr * 32 + i) * kGemmLd + 2 * s + h];
            const float b = Bs[(wc * 32 + i) * kGemmLd + 2 * s + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
    const int n = n0 + wc * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M && n < g.N) {
            float v = acc[r] * g.scale;
            if (g.epi & EPI_BIAS) v += g.bias[n];
            if (g.epi & EPI_GELU) v = gelu_erf(v);
            if (g.epi & EPI_RESID) v += R[(int64_t)m * g.ldc + n];
            C[(int64_t)m * g.ldc + n] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// The same GEMM for the shapes that carry the encoder's flops (activations x torch Linear weights: both operands
// contiguous in k, K % 32 == 0, 16-byte aligned rows).  The loop issues little besides v_mfma_f32_32x32x2_f32: a wave
// owns WM x WN accumulator tiles of 32 x 32 and feeds 4 MFMAs of a tile from ONE ds_read_b128 per operand (lane half
// h owns k in [16h, 16h+16) of the 32-deep step, i.e. the k index of the MFMA is a permutation of the tile's k, applied
// to A and B alike); the next step's global loads are issued before the MFMAs and written to the other LDS buffer
// after them (one barrier per step).
// Per 32-deep step and wave: 4 (WM + WN) LDS reads for 16 WM WN MFMAs of 64 cycles.
// Work-group = 2 x 2 waves = (64 WM) x (64 WN) output tile.  LDS rows are 36 floats: the 16 lanes of a ds_read_b128
// group then touch 16 different 16-byte slots.
// -------------------------------------------------------------------------------------------------------------
constexpr int kG2Ld = 36;

template <int WM, int WN>
static __global__ __launch_bounds__(256) void gemm_f32_tiled_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    extern __shared__ __attribute__((aligned(16))) float g2lds[];      // 2 buffers x (BM + BN) rows x 36 floats
    const int z = blockIdx.z;
    const bool split = g.ksplit > 1;
    const int zo = split ? 0 : z / g.zi_count, zi = split ? 0 : z % g.zi_count;
    const int kslice = split ? g.K / g.ksplit : g.K;           // a multiple of 32 (host)
    const float* A = g.A + zo * g.a_so + zi * g.a_si + (split ? (int64_t)z * kslice : 0);
    const float* B = g.B + zo * g.b_so + zi * g.b_si + (split ? (int64_t)z * kslice : 0);
    float* C = g.C + zo * g.c_so + zi * g.c_si + (split ? (int64_t)z * g.M * g.ldc : 0);
    const float* R = g.R ? g.R + zo * g.c_so + zi * g.c_si : nullptr;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    // loader: thread t moves float4 (t & 7) of rows (t >> 3) + 32 p  (p < BM / 32 for A, < BN / 32 for B)
    const int lrow = tid >> 3, lq = tid & 7;
    f32x4 pa[BM / 32], pb[BN / 32];
    const int M_ = g.M, N_ = g.N, K_ = kslice;
    const int64_t lda_ = g.lda, ldbn_ = g.ldb_n;
    // (macros, not lambdas: a by-reference capture of the by-value argument struct forces it into scratch memory)
#define CLB_G2_LOAD(K0)                                                                                      \
    {                                                                                                        \
        _Pragma("unroll") for (int p = 0; p < BM / 32; ++p) {                                                \
            int m = m0 + lrow + 32 * p;                                                                      \
            m = m < M_ ? m : M_ - 1;                                                                         \
            pa[p] = *reinterpret_cast<const f32x4*>(A + (int64_t)m * lda_ + (K0) + 4 * lq);                 \
        }                                                                                                    \
        _Pragma("unroll") for (int p = 0; p < BN / 32; ++p) {                                                \
            int n = n0 + lrow + 32 * p;                                                                      \
            n = n < N_ ? n : N_ - 1;                                                                         \
            pb[p] = *reinterpret_cast<const f32x4*>(B + (int64_t)n * ldbn_ + (K0) + 4 * lq);                \
        }                                                                                                    \
    }
#define CLB_G2_STORE(BUF)                                                                                    \
    {                                                                                                        \
        float* As_ = g2lds + (BUF) * (BM + BN) * kG2Ld;                                                      \
        float* Bs_ = As_ + BM * kG2Ld;                                                                       \
        _Pragma("unroll") for (int p = 0; p < BM / 32; ++p)                                                  \
            *reinterpret_cast<f32x4*>(As_ + (lrow + 32 * p) * kG2Ld + 4 * lq) = pa[p];                      \
        _Pragma("unroll") for (int p = 0; p < BN / 32; ++p)                                                  \
            *reinterpret_cast<f32x4*>(Bs_ + (lrow + 32 * p) * kG2Ld + 4 * lq) = pb[p];                      \
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    CLB_G2_LOAD(0)
    CLB_G2_STORE(0)
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K_; k0 += 32) {
        const bool more = k0 + 32 < K_;
        if (more) CLB_G2_LOAD(k0 + 32)
        const float* As = g2lds + buf * (BM + BN) * kG2Ld + (wr * 32 * WM + i) * kG2Ld + 16 * h;
        const float* Bs = g2lds + buf * (BM + BN) * kG2Ld + BM * kG2Ld + (wc * 32 * WN + i) * kG2Ld + 16 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 av[WM], bv[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) av[a] = *reinterpret_cast<const f32x4*>(As + a * 32 * kG2Ld + 4 * j);
#pragma unroll
            for (int b = 0; b < WN; ++b) bv[b] = *reinterpret_cast<const f32x4*>(Bs + b * 32 * kG2Ld + 4 * j);
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][0], bv[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][1], bv[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][2], bv[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][3], bv[b][3], acc[a][b], 0, 0, 0);
                }
        }
        if (more) CLB_G2_STORE(buf ^ 1)
        __syncthreads();
        buf ^= 1;
    }
#undef CLB_G2_LOAD
#undef CLB_G2_STORE
    // C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const int n = n0 + (wc * WN + b) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wr * WM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M && n < g.N) {
                    float v = acc[a][b][r];
                    if (!split) {
                        v = v * g.scale;
                        if (g.epi & EPI_BIAS) v += g.bias[n];
                        if (g.epi & EPI_GELU) v = gelu_erf(v);
                        if (g.epi & EPI_RESID) v += R[(int64_t)m * g.ldc + n];
                    }
                    C[(int64_t)m * g.ldc + n] = v;
                }
            }
        }
}

// -------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate on gfx950: 512
// matrix-pipe cycles per 32 x 32 x 16 block against 32 per bf16 MFMA).  Every fp32 operand is split into NS bf16
// planes, x = p0 + p1 (+ p2) with p0 = RN_bf16(x), p1 = RN_bf16(x - p0), p2 = RN_bf16(x - p0 - p1) -- the
// subtractions are exact in fp32 -- and the product is the sum of the plane products whose magnitude matters, every
// one exact in the fp32 accumulator, smallest terms first:
//   NS = 2 ("bf16x3"): a*b ~= a1*b0 + a0*b1 + a0*b0                            3 MFMAs, error < 2^-15 |a||b| per product
//   NS = 3 ("bf16x6"): a*b ~= a2*b0 + a0*b2 + a1*b1 + a1*b0 + a0*b1 + a0*b0    6 MFMAs, error < 2^-22 |a||b|: three planes
//                      hold all 24 significant bits of an fp32 number, the dropped terms are below 2^-24 |a||b| each
// Both operands stay fp32 in HBM/L2 and are split by the loader threads while a tile is staged (v_cvt_pk_bf16_f32 +
// exact subtractions: plain VALU work that hides under the MFMAs of the co-resident waves,
// tools/microbench/issue_overlap): the kernel is bound by the bytes it pulls through L2 -- measured: pre-split
// planes in memory (6 B per element instead of 4) made it slower, not faster -- so nothing but fp32 is ever read.
// LDS: one buffer of NS planes per operand, rows of 32 bf16 = 64 B without padding; 16-byte chunk c of row r sits at
// chunk position c ^ ((r >> 2) & 3), so the 16 lanes of a ds_read_b128 group (16 consecutive rows, one chunk) touch
// 16 different 16-byte bank groups.  Work-group = WGM x WGN waves, wave tile (32 WM) x (32 WN), K step 32 = two k16
// MFMA groups; the next step's global loads are issued before the MFMAs and written after them (two barriers per step;
// a second work-group or the second wave of every SIMD covers them).
// -------------------------------------------------------------------------------------------------------------
struct Gemm3Args {
    const float* A; const float* B; float* C; const float* bias; const float* R;
    int M, N, K;
    int64_t lda, ldb, ldc;
    float scale;
    int epi;
    int ksplit;        // > 1: blockIdx.z = K slice; raw partial sums go to C + z*M*ldc
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// four consecutive k of one row -> 8 bytes in each of the NS planes (plane stride PL bytes)
template <int NS>
__device__ __forceinline__ void split_store4(const f32x4 v, unsigned char* d, int PL) {
    const uint32_t h0 = cvt_pk_bf16(v[0], v[1]), h1 = cvt_pk_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
    const float r0 = v[0] - __uint_as_float(h0 << 16), r1 = v[1] - __uint_as_float(h0 & 0xffff0000u);
    const float r2 = v[2] - __uint_as_float(h1 << 16), r3 = v[3] - __uint_as_float(h1 & 0xffff0000u);
    const uint32_t l0 = cvt_pk_bf16(r0, r1), l1 = cvt_pk_bf16(r2, r3);
    *reinterpret_cast<uint2*>(d + PL) = make_uint2(l0, l1);
    if (NS == 3) {
        const uint32_t t0 = cvt_pk_bf16(r0 - __uint_as_float(l0 << 16), r1 - __uint_as_float(l0 & 0xffff0000u));
        const uint32_t t1 = cvt_pk_bf16(r2 - __uint_as_float(l1 << 16), r3 - __uint_as_float(l1 & 0xffff0000u));
        *reinterpret_cast<uint2*>(d + 2 * PL) = make_uint2(t0, t1);
    }
}

// DB = true (round 3, the small tiles of query batches): TWO LDS tile buffers.  A query batch gives these GEMMs 1-2
// work-groups per CU (384 x 4 waves on 1 024 SIMDs), so nothing covered the two barriers per step of the single-buffer
// loop: a wave spent 39 % of its cycles waiting and its 224 VALU instructions of operand splitting and its 24 MFMAs
// took turns (PMC: MFMA busy 20 % of the wave cycles).  Here the split of tile k+1 (registers -> the other buffer) and
// the MFMAs of tile k (this buffer) sit in one basic block with one barrier per step, so the vector work issues in the
// shadow of the matrix pipe, and tile k+2 is requested as soon as the registers are free.
template <int WGM, int WGN, int WM, int WN, int NS, bool DB = false>
static __global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16split_kernel(Gemm3Args g) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
    constexpr int PA = BM * 64, PB = BN * 64;                              // bytes per plane
    constexpr int RPP = NT / 8;                                            // rows covered by one loader pass
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the loader pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char g3lds[];  // [A planes | B planes]
    const int z = blockIdx.z;
    const bool split = g.ksplit > 1;
    const int K_ = split ? g.K / g.ksplit : g.K;                // a multiple of 32 (host)
    const float* A = g.A + (split ? (int64_t)z * K_ : 0);
    const float* B = g.B + (split ? (int64_t)z * K_ : 0);
    float* C = g.C + (split ? (int64_t)z * g.M * g.ldc : 0);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave / WGN, wc = wave % WGN;
    const int i = lane & 31, h = lane >> 5;
    const int M_ = g.M, N_ = g.N;
    const int64_t lda_ = g.lda, ldb_ = g.ldb;
    // loader: thread t moves float4 (t & 7) of rows (t >> 3) + RPP p of A and of B
    const int lrow = tid >> 3, lq = tid & 7;
    f32x4 pa[BM / RPP], pb[BN / RPP];
    constexpr int BUFB = NS * (PA + PB);                                   // bytes of one tile buffer
#define CLB_G3_LOAD(K0) CLB_G3_LOAD_(K0, pa, pb)
#define CLB_G3_STORE() CLB_G3_STORE_(pa, pb, 0)
#define CLB_G3_STORE_AT(OFF) CLB_G3_STORE_(pa, pb, OFF)
#define CLB_G3_LOAD_(K0, pa, pb)                                                                                      \
    {                                                                                                        \
        _Pragma("unroll") for (int p = 0; p < BM / RPP; ++p) {                                               \
            int m = m0 + lrow + RPP * p;                                                                     \
            m = m < M_ ? m : M_ - 1;                                                                         \
            pa[p] = *reinterpret_cast<const f32x4*>(A + (int64_t)m * lda_ + (K0) + 4 * lq);                 \
        }                                                                                                    \
        _Pragma("unroll") for (int p = 0; p < BN / RPP; ++p) {                                               \
            int n = n0 + lrow + RPP * p;                                                                     \
            n = n < N_ ? n : N_ - 1;                                                                         \
            pb[p] = *reinterpret_cast<const f32x4*>(B + (int64_t)n * ldb_ + (K0) + 4 * lq);                 \
        }                                                                                                    \
    }
#define CLB_G3_STORE_(pa, pb, OFF)                                                                           \
    {                                                                                                        \
        _Pragma("unroll") for (int p = 0; p < BM / RPP; ++p) {                                               \
            const int row_ = lrow + RPP * p;                                                                 \
            split_store4<NS>(pa[p], g3lds + (OFF) + row_ * 64 + (((lq >> 1) ^ ((row_ >> 2) & 3)) << 4) + ((lq & 1) << 3), PA); \
        }                                                                                                    \
        _Pragma("unroll") for (int p = 0; p < BN / RPP; ++p) {                                               \
            const int row_ = lrow + RPP * p;                                                                 \
            split_store4<NS>(pb[p], g3lds + (OFF) + NS * PA + row_ * 64 + (((lq >> 1) ^ ((row_ >> 2) & 3)) << 4) + ((lq & 1) << 3), PB); \
        }                                                                                                    \
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    CLB_G3_LOAD(0)
    CLB_G3_STORE()
    __syncthreads();
    const int sw = (i >> 2) & 3;                  // the swizzle of this lane's rows (tile bases are multiples of 32)
    const unsigned char* As = g3lds + (wr * 32 * WM + i) * 64;
    const unsigned char* Bs = g3lds + NS * PA + (wc * 32 * WN + i) * 64;
    auto compute_step = [&](int boff = 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = (((2 * s + h) ^ sw) << 4) + boff;
            bf16x8 av[NS][WM], bv[NS][WN];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
#pragma unroll
                for (int a = 0; a < WM; ++a) av[q][a] = *reinterpret_cast<const bf16x8*>(As + q * PA + a * 32 * 64 + off);
#pragma unroll
                for (int b = 0; b < WN; ++b) bv[q][b] = *reinterpret_cast<const bf16x8*>(Bs + q * PB + b * 32 * 64 + off);
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    f32x16 c = acc[a][b];
                    if (NS == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2][a], bv[0][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][a], bv[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1][a], bv[1][b], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1][a], bv[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][a], bv[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][a], bv[0][b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
    };
    if (!DB) {
        for (int k0 = 0; k0 < K_; k0 += 32) {
            const bool more = k0 + 32 < K_;
            if (more) CLB_G3_LOAD(k0 + 32)
            compute_step();
            __syncthreads();              // every wave has read the tile
            if (more) CLB_G3_STORE()
            __syncthreads();
        }
    } else {
        if (K_ > 32) CLB_G3_LOAD(32)                  // tile 1 waits in registers
        int cur = 0;
        // One step: all fragments of tile k leave this buffer FIRST (the compiler will not move an LDS read above an LDS
        // write it cannot tell apart), then tile k+1 goes registers -> the other buffer (every wave left it at the barrier
        // below) and tile k+2 is requested as soon as the registers are free; the MFMAs depend on registers only and
        // interleave with that vector work.  STORE / LOAD are compile-time flags: a branch inside the step would split
        // the basic block and the MFMAs would queue up behind the split again (the last two steps are peeled instead).
#define CLB_G3_DB_STEP(K0, DO_STORE, DO_LOAD)                                                                \
        {                                                                                                    \
            bf16x8 av[2][NS][WM], bv[2][NS][WN];                                                             \
            const int boff = cur * BUFB;                                                                     \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                  \
                const int off = (((2 * s + h) ^ sw) << 4) + boff;                                            \
                _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                             \
                    _Pragma("unroll") for (int a = 0; a < WM; ++a)                                           \
                        av[s][q][a] = *reinterpret_cast<const bf16x8*>(As + q * PA + a * 32 * 64 + off);     \
                    _Pragma("unroll") for (int b = 0; b < WN; ++b)                                           \
                        bv[s][q][b] = *reinterpret_cast<const bf16x8*>(Bs + q * PB + b * 32 * 64 + off);     \
                }                                                                                            \
            }                                                                                                \
            if (DO_STORE) CLB_G3_STORE_AT((cur ^ 1) * BUFB)                                                  \
            if (DO_LOAD) CLB_G3_LOAD((K0) + 64)                                                              \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                    \
                _Pragma("unroll") for (int a = 0; a < WM; ++a)                                               \
                    _Pragma("unroll") for (int b = 0; b < WN; ++b) {                                         \
                        f32x16 c = acc[a][b];                                                                \
                        if (NS == 3) {                                                                       \
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][2][a], bv[s][0][b], c, 0, 0, 0); \
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][0][a], bv[s][2][b], c, 0, 0, 0); \
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][1][a], bv[s][1][b], c, 0, 0, 0); \
                        }                                                                                    \
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][1][a], bv[s][0][b], c, 0, 0, 0);   \
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][0][a], bv[s][1][b], c, 0, 0, 0);   \
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s][0][a], bv[s][0][b], c, 0, 0, 0);   \
                        acc[a][b] = c;                                                                       \
                    }                                                                                        \
            __syncthreads();                                                                                 \
            cur ^= 1;                                                                                        \
        }
        int k0 = 0;
        for (; k0 + 64 < K_; k0 += 32) CLB_G3_DB_STEP(k0, true, true)
        if (k0 + 32 < K_) { CLB_G3_DB_STEP(k0, true, false) k0 += 32; }
        CLB_G3_DB_STEP(k0, false, false)
#undef CLB_G3_DB_STEP
    }
#undef CLB_G3_LOAD
#undef CLB_G3_STORE
#undef CLB_G3_STORE_AT
#undef CLB_G3_LOAD_
#undef CLB_G3_STORE_
    // C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const int n = n0 + (wc * WN + b) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wr * WM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M && n < g.N) {
                    float v = acc[a][b][r];
                    if (!split) {
                        v = v * g.scale;
                        if (g.epi & EPI_BIAS) v += g.bias[n];
                        if (g.epi & EPI_GELU) v = gelu_erf(v);
                        if (g.epi & EPI_RESID) v += g.R[(int64_t)m * g.ldc + n];
                    }
                    C[(int64_t)m * g.ldc + n] = v;
                }
            }
        }
}

// split-K second pass: C = epilogue(sum over the K slices, in slice order -- deterministic)
static __global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int ksplit,
                                                                       int64_t M, int N, float* __restrict__ C,
                                                                       const float* __restrict__ bias,
                                                                       const float* __restrict__ R, float scale, int epi) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int n = (int)(idx % N);
    float v = part[idx];
    for (int z = 1; z < ksplit; ++z) v = v + part[(size_t)z * M * N + idx];
    v = v * scale;
    if (epi & EPI_BIAS) v += bias[n];
    if (epi & EPI_GELU) v = gelu_erf(v);
    if (epi & EPI_RESID) v += R[idx];
    C[idx] = v;
}

// split-K second pass fused with the LayerNorm that follows it (the attention-output and FFN-output Linears of a short
// batch): one 256-thread work-group per output row (a wave per row leaves a query batch's 1 024 rows one wave per
// SIMD, all latency: 19.7 us against 13.6 for the two separate launches), the row stays in registers between the
// reduction and the normalisation.  The element arithmetic is that of gemm_splitk_reduce_kernel; mean and variance are
// summed per wave, then over the four waves in order.  N <= 256 * NR.
template <int NR>
static __global__ __launch_bounds__(256) void gemm_splitk_reduce_ln_kernel(const float* __restrict__ part, int ksplit,
                                                                          int64_t M, int N, float* __restrict__ C,
                                                                          const float* __restrict__ bias,
                                                                          const float* __restrict__ R, float scale, int epi,
                                                                          const float* __restrict__ gamma,
                                                                          const float* __restrict__ beta, float eps) {
    __shared__ float red[2][4];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t slice = (size_t)M * N;
    float v[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int n = tid + 256 * j;
        v[j] = n < N ? part[t * N + n] : 0.f;
    }
    for (int z = 1; z < ksplit; ++z) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int n = tid + 256 * j;
            if (n < N) v[j] = v[j] + part[z * slice + t * N + n];
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int n = tid + 256 * j;
        if (n < N) {
            float x = v[j] * scale;
            if (epi & EPI_BIAS) x += bias[n];
            if (epi & EPI_GELU) x = gelu_erf(x);
            if (epi & EPI_RESID) x += R[t * N + n];
            v[j] = x;
            sum += x;
        }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[0][wave] = sum;
    __syncthreads();
    const float mean = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) / (float)N;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NR; ++j)
        if (tid + 256 * j < N) { const float c = v[j] - mean; var += c * c; }
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    if (lane == 0) red[1][wave] = var;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / (float)N + eps);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int n = tid + 256 * j;
        if (n < N) C[t * N + n] = (v[j] - mean) * rstd * gamma[n] + beta[n];
    }
}

// embeddings: word[id] + position[pos] + token_type[0], then LayerNorm.  One wave per token.  ids are the
// reference's 1-based Int32 ids (Julia), (L, N) column-major = token (l, n) at ids[l + L*n].
static __global__ __launch_bounds__(256) void embed_layernorm_kernel(const int32_t* __restrict__ ids, int64_t n_tok,
                                                                    int L, int H, int vocab,
                                                                    const float* __restrict__ word,
                                                                    const float* __restrict__ pos,
                                                                    const float* __restrict__ type0,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps,
                                                                    float* __restrict__ out, int* __restrict__ err) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tok) return;
    int id = ids[t] - 1;
    if (id < 0 || id >= vocab) { if (lane == 0) atomicOr(err, 1); id = 0; }
    const int l = (int)(t % L);
    float sum = 0.f;
    for (int d = lane; d < H; d += 64) {
        const float v = word[(int64_t)id * H + d] + pos[(int64_t)l * H + d] + type0[d];
        out[t * H + d] = v;
        sum += v;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)H;
    float var = 0.f;
    for (int d = lane; d < H; d += 64) {
        const float c = out[t * H + d] - mean;
        var += c * c;
    }
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var / (float)H + eps);
    for (int d = lane; d < H; d += 64) out[t * H + d] = (out[t * H + d] - mean) * rstd * gamma[d] + beta[d];
}

// in-place LayerNorm over rows of length H.  One wave per row.
static __global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, int64_t rows, int H,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= rows) return;
    float* row = x + t * H;
    float sum = 0.f;
    for (int d = lane; d < H; d += 64) sum += row[d];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)H;
    float var = 0.f;
    for (int d = lane; d < H; d += 64) { const float c = row[d] - mean; var += c * c; }
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var / (float)H + eps);
    for (int d = lane; d < H; d += 64) row[d] = (row[d] - mean) * rstd * gamma[d] + beta[d];
}

// masked softmax over the key axis of attention scores S[z][q][k] (z = document*heads + head), in place.
// Masked keys (bitmask == 0) get probability 0 -- GenericSequenceMask(bitmask), checkpoint.jl:24.  One wave per row.
static __global__ __launch_bounds__(256) void masked_softmax_kernel(float* __restrict__ S, int64_t rows, int L,
                                                                   int heads, const uint8_t* __restrict__ mask) {
    const int64_t rrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rrow >= rows) return;
    const int64_t z = rrow / L;
    const int64_t n = z / heads;
    float* s = S + rrow * L;
    const uint8_t* mk = mask + n * L;
    float mx = kNegInf;
    for (int k = lane; k < L; k += 64) mx = fmaxf(mx, mk[k] ? s[k] : kNegInf);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int k = lane; k < L; k += 64) {
        const float e = mk[k] ? expf(s[k] - mx) : 0.f;
        s[k] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int k = lane; k < L; k += 64) s[k] *= inv;
}

// -------------------------------------------------------------------------------------------------------------
// Fused self-attention for head size 64: softmax(Q K^T / sqrt(dh) + key mask) V of one (sequence, head) pair and one
// block of 32 query positions PER WAVE, entirely in registers -- no L x L score matrix in memory, no barrier, no LDS
// (the unfused path writes and re-reads N heads L^2 floats three times per layer: 13 of the 32 ms of a 64 x 300 passage
// batch).  fp32 MFMA (v_mfma_f32_32x32x2_f32) for both products:
//   * S^T tile = K_tile (32 keys x 64) . Q_blk^T (64 x 32 queries): computing the TRANSPOSE puts query i in lane
//     (i, h) with 16 of the tile's keys in its accumulator registers (key = 32 jt + (r & 3) + 8 (r >> 2) + 4 h), which
//     is exactly the A-operand layout of the second product -- no transposition of P between the two MFMA chains.
//     The summation index of an MFMA step is a dummy: lane half h carries features 32h + s at step s, so a lane's
//     32 operand values are one contiguous 128-byte piece of its Q / K row (8 dwordx4 loads).
//   * row maximum / sum: in-lane over the NT tiles' registers, then one exchange between the two lane halves;
//     masked keys (bitmask == 0, GenericSequenceMask(bitmask), checkpoint.jl:24) and keys past L get probability 0.
//   * O (32 x 64) = P V: A = the normalised probabilities straight from the accumulator registers, B = V[key][d]
//     read as 128-byte row pieces (lane = d), two 32-column output tiles.
// NT = key tiles of 32 (L <= 32 NT).  grid = (ceil(L / 32), heads, N), block = 64.
// -------------------------------------------------------------------------------------------------------------
template <int NT>
static __global__ __launch_bounds__(64) void attention_fused_kernel(const float* __restrict__ qkv,
                                                                    const uint8_t* __restrict__ mask,
                                                                    float* __restrict__ ctx, int L, int H, float scale) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, head = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int64_t ld = 3 * (int64_t)H;
    const float* base = qkv + n * L * ld + head * 64;
    const uint8_t* mk = mask + n * L;
    // B operand of the first product: Q[query q0 + i][32h + s]
    float qreg[32];
    {
        const int qrow = q0 + i < L ? q0 + i : L - 1;
        const f32x4* src = reinterpret_cast<const f32x4*>(base + qrow * ld + 32 * h);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 v = src[c];
            qreg[4 * c] = v[0]; qreg[4 * c + 1] = v[1]; qreg[4 * c + 2] = v[2]; qreg[4 * c + 3] = v[3];
        }
    }
    f32x16 st[NT];
    uint32_t valid[NT];          // wave-uniform: bit j = key 32 jt + j takes part
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int key = 32 * jt + i;
        const int krow = key < L ? key : L - 1;
        valid[jt] = (uint32_t)__builtin_amdgcn_ballot_w64(h == 0 && key < L && mk[krow] != 0);
        const f32x4* src = reinterpret_cast<const f32x4*>(base + H + krow * ld + 32 * h);
        f32x4 kf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) kf[c] = src[c];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s >> 2][s & 3], qreg[s], acc, 0, 0, 0);
        st[jt] = acc;
    }
    // softmax over the keys of query i (this lane and its partner in the other half hold them all)
    float mx = kNegInf;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kb = (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = (valid[jt] >> kb) & 1u;
            const float v = ok ? st[jt][r] * scale : kNegInf;
            st[jt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = st[jt][r];
            const float e = v > kNegInf ? expf(v - mx) : 0.f;
            st[jt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    // O = P V
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    const float* vbase = base + 2 * H + i;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float* vrow = vbase + (int64_t)(key < L ? key : L - 1) * ld;
            const float pr = st[jt][r] * inv;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pr, vrow[0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pr, vrow[32], o1, 0, 0, 0);
        }
    // o[r] = O[query q0 + (r & 3) + 8 (r >> 2) + 4 h][d = i (+ 32)]
    float* out = ctx + n * L * (int64_t)H + head * 64 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (q < L) {
            out[(int64_t)q * H] = o0[r];
            out[(int64_t)q * H + 32] = o1[r];
        }
    }
}

// The same attention with a running (online) softmax for long sequences: the key tiles are visited one after the
// other, only ONE 32 x 32 score tile lives in registers next to the 32 x 64 output accumulator, and whenever a query's
// running maximum rises the output rows of that query are rescaled by exp(m_old - m_new).  ~130 registers instead of
// 256+ (attention_fused_kernel keeps all NT tiles): three to four waves per SIMD cover each other's loads.  The
// rescale factor of query q lives in lane q (transposed score layout) but scales accumulator REGISTERS (output rows):
// it is fetched with v_readlane (two per register: the two lane halves hold different queries) and skipped for tiles
// that raise no maximum (wave-uniform test).  Same arithmetic per product as the register-resident kernel; the softmax
// differs from the two-pass form only by fp32 rounding (exp(a)exp(b) vs exp(a+b)).
// grid = (ceil(L / 32), heads, N), block = 64.
static __global__ __launch_bounds__(64, 3) void attention_online_kernel(const float* __restrict__ qkv,
                                                                        const uint8_t* __restrict__ mask,
                                                                        float* __restrict__ ctx, int L, int H, float scale) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, head = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int64_t ld = 3 * (int64_t)H;
    const float* base = qkv + n * L * ld + head * 64;
    const uint8_t* mk = mask + n * L;
    float qreg[32];
    {
        const int qrow = q0 + i < L ? q0 + i : L - 1;
        const f32x4* src = reinterpret_cast<const f32x4*>(base + qrow * ld + 32 * h);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 v = src[c];
            qreg[4 * c] = v[0]; qreg[4 * c + 1] = v[1]; qreg[4 * c + 2] = v[2]; qreg[4 * c + 3] = v[3];
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = kNegInf, l = 0.f;              // running maximum / sum of query i (l: this lane half's keys only)
    const float* vbase = base + 2 * H + i;
    const int nt = (L + 31) >> 5;
    for (int jt = 0; jt < nt; ++jt) {
        const int key = 32 * jt + i;
        const int krow = key < L ? key : L - 1;
        const uint32_t valid = (uint32_t)__builtin_amdgcn_ballot_w64(h == 0 && key < L && mk[krow] != 0);
        if (valid == 0u) continue;                                   // a tile of masked keys (padding): nothing to add
        const f32x4* src = reinterpret_cast<const f32x4*>(base + H + krow * ld + 32 * h);
        f32x4 kf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) kf[c] = src[c];
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s >> 2][s & 3], qreg[s], st, 0, 0, 0);
        float tmax = kNegInf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kb = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = ((valid >> kb) & 1u) ? st[r] * scale : kNegInf;
            st[r] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);                          // finite: the tile has a valid key
        const float alpha = m > kNegInf ? expf(m - m_new) : 0.f;     // 1 when the maximum did not move
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = st[r] > kNegInf ? expf(st[r] - m_new) : 0.f;
            st[r] = e;
            psum += e;
        }
        l = l * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(m_new > m) != 0ull) {       // some query's maximum rose: rescale its output row
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qa = (r & 3) + 8 * (r >> 2);               // the query of register r in lane half 0; +4 in half 1
                const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), qa));
                const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), qa + 4));
                const float a = h ? a1 : a0;
                o0[r] *= a;
                o1[r] *= a;
            }
        }
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float* vrow = vbase + (int64_t)(kk < L ? kk : L - 1) * ld;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vrow[0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vrow[32], o1, 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float* out = ctx + n * L * (int64_t)H + head * 64 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qa = (r & 3) + 8 * (r >> 2);
        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa));
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa + 4));
        const float sc = h ? s1 : s0;
        const int q = q0 + qa + 4 * h;
        if (q < L) {
            out[(int64_t)q * H] = o0[r] * sc;
            out[(int64_t)q * H + 32] = o1[r] * sc;
        }
    }
}

// (N*L, dim) row-major projection output -> the reference's (dim, L, N) column-major array is the same memory:
// element (d, l, n) at d + dim*(l + L*n) = row (l + L*n), column d.  So no transpose kernel is needed.

}  // namespace clb
