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
enum { EPI_BIAS = 1, EPI_GELU = 2, EPI_RESID = 4, EPI_QKV_ATT = 8 };   // EPI_QKV_ATT: gemm_planes2_kernel only (see GemmPArgs::Vt)

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

// GELU(x) = x/2 (1 + erf(x / sqrt 2)) needs erf to an ABSOLUTE accuracy near one fp32 ulp of 1, not a relative one:
// erf(t) = 1 - exp(q(t)) with q = log erfc as ONE degree-9 polynomial on [0, 4] (tools/fit_gelu_erf.py: max error 2.2e-9 in
// exact arithmetic, 1.1e-7 evaluated in fp32; erfc(4) = 1.5e-8 is below half an ulp of 1), nine FMAs + one v_exp_f32 instead
// of the two-branch library erff (~45 instructions on a divergent wave) -- the GELU epilogue of the FFN-in product of a
// passage batch was ~2/3 of a tile's main loop in vector instructions.
constexpr float kErfQ[10] = {2.14899565e-09f, -1.12837946f, -0.636615276f, -0.102803029f, 0.0192260593f, 4.69753249e-05f,
                             -0.00156761205f, 0.000583863817f, -0.000105056366f, 7.94943207e-06f};
__device__ __forceinline__ float erf_abs(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float q = kErfQ[9];
#pragma unroll
    for (int i = 8; i >= 0; --i) q = fmaf(q, t, kErfQ[i]);
    return copysignf(1.0f - __expf(q), x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_abs(x * 0.70710678118654752440f)); }

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

// -------------------------------------------------------------------------------------------------------------
// Round 4: the same split products with the operand split taken OUT of the GEMM.  PMC of the kernel above on a query
// batch: 224 VALU + 24 MFMA instructions per 32-deep step and wave, matrix pipe 20 % busy -- every work-group re-split the
// activation rows it shares with the N / BN other work-groups of its tile row, and the static weights were re-split on
// every call.  Here both operands arrive as 16-bit PLANES: the weights are split once at clb_encoder_create (and again
// when the GEMM mode changes), an activation is split once by the kernel that PRODUCES it (LayerNorm, GELU epilogue,
// attention output: 2-byte stores per plane instead of one 4-byte store), and the GEMM's loader does no arithmetic at all:
// every 16 rows x 64 bytes of a plane go global -> LDS with ONE global_load_lds_dwordx4 (LDS-DMA: no VGPRs, no ds_write),
// the lane -> chunk map of the instruction producing the swizzled LDS image of the kernel above (16-byte chunk c of row r
// at position c ^ ((r >> 2) & 3)).  A ring of STAGES tile buffers, one barrier per step: step k waits for its own DMAs
// (s_waitcnt vmcnt(DPW * (STAGES - 2)): vector memory operations complete in order), meets the barrier -- every wave has
// then left buffer (k - 1) % STAGES -- and refills that buffer with tile k + STAGES - 1 before its MFMAs.  The DMAs are
// hand-issued (inline asm): behind the builtin hipcc places s_waitcnt vmcnt(0) in front of every LDS read that may alias
// DMA-written memory, which would serialise the ring.  (The asm writes M0 without declaring it: hipcc rejects M0 in a
// clobber list as a reserved register, and nothing here keeps a value in M0 across a statement.)
// bf16 planes: the same products in the same order as the kernel above -- for equal tiles and K slices the results are
// bit-identical.  Output: fp32 C (bias / GELU / residual epilogue, or raw split-K partial sums) and / or the planes of the
// epilogue's result (Cp: the FFN intermediate is only ever read as planes and never exists in fp32).
// -------------------------------------------------------------------------------------------------------------
struct GemmPArgs {
    const uint16_t* A; const uint16_t* B;      // plane 0 of each operand, K-BLOCKED (see plane_index)
    int64_t a_plane, b_plane;                  // plane strides (elements)
    float* C; const float* bias; const float* R;
    uint16_t* Cp; int64_t c_plane;             // optional planes of the output, K-blocked with M rows (the next Linear's A)
    int M, N, K;
    int64_t ldc;
    int epi;
    int ksplit;                                // > 1: blockIdx.z = K slice; raw partial sums go to C + z*M*ldc
    float out_scale;                           // multiplies the accumulator first (PF_F16X2: the inverse operand scales; else 1)
    // EPI_QKV_ATT (the Q/K/V projection in front of attention_f16_kernel, PF_F16X2): columns [0, 2 att_H) -- Q and K -- go to
    // Cp as K-blocked planes of a (M x 2 att_H) matrix, columns [2 att_H, 3 att_H) -- V -- to Vt in the key-blocked layout of
    // vt_index; no fp32 copy exists
    uint16_t* Vt = nullptr; int64_t vt_plane = 0;
    int att_L = 0, att_H = 0, att_heads = 0;
    // packed batches (sequences of different lengths back to back, no padding rows): the sequence and the position of
    // every row; null: row m = sequence m / att_L, position m % att_L
    const int32_t* att_seq = nullptr; const int32_t* att_pos = nullptr;
    // LayerNorm folded AROUND the product (gemm_planes2_kernel<..., LN = true>; see "LayerNorm without a pass of its own"):
    //   ln_in     [M][ln_parts][2] partial (mean, M2 = sum of squared deviations from that mean) of the rows the LayerNorm is
    //             taken over, ln_width columns per part: A's rows when ln_u is set (fold), R's rows when r_gamma is set
    //   ln_u      fold: u[n] = sum_k gamma_k W[n][k]; `bias` then holds c[n] = sum_k beta_k W[n][k] + b[n] and the B planes hold
    //             gamma (.) W:  LN(a) . W^T + b  =  rstd (a . (gamma (.) W)^T - mean u) + c
    //   r_gamma / r_beta   the residual R holds RAW rows (a product's output before its LayerNorm): what is added is
    //             (R - mean) rstd gamma + beta
    //   stats_out [M][N / 64][2]: the partial (mean, M2) of THIS product's output rows, 64 columns per part
    const float* ln_in = nullptr; int ln_parts = 0, ln_width = 0; float ln_eps = 0.f;
    const float* ln_u = nullptr;
    const float* r_gamma = nullptr; const float* r_beta = nullptr;
    float* stats_out = nullptr;
};

// mean and 1 / sqrt(var + eps) of a row from its partial statistics (equal-width parts): Chan's combination -- the deviations
// are taken from each part's own mean and the parts' means from the row's, never E[x^2] - mean^2
constexpr int kLnMaxParts = 16;     // hidden <= 1 024 (64 columns per part)
// (all parts are requested before the first is used: a rolled load -> add loop is a chain of `parts` memory latencies -- it cost the
// consuming GEMMs 14 % when it was written that way)
__device__ __forceinline__ void ln_merge_stats(const f32x2_t* sv /* kLnMaxParts, in registers */, int parts, int width, float eps, float& mean, float& rstd) {
    float msum = 0.f, m2 = 0.f;
#pragma unroll
    for (int p = 0; p < kLnMaxParts; ++p) if (p < parts) { msum += sv[p][0]; m2 += sv[p][1]; }
    mean = msum / (float)parts;
    float dev = 0.f;
#pragma unroll
    for (int p = 0; p < kLnMaxParts; ++p) if (p < parts) { const float d = sv[p][0] - mean; dev = fmaf(d, d, dev); }
    const float var = (m2 + (float)width * dev) / (float)(parts * width);
    rstd = 1.0f / sqrtf(var + eps);
}

// V for attention_f16_kernel: [sequence][head][tile of 32 keys][d = 0..63][32 keys], the keys of a tile in the order in which
// the lanes of the score accumulator hold them -- lane half h, MFMA step u, slot j <-> key 16 u + 8 (j >> 2) + 4 h + (j & 3) sits
// at position 16 h + 8 u + j: the B fragment of P.V for lane (d, h) is 32 contiguous bytes
__device__ __host__ __forceinline__ int64_t vt_index(int64_t seq, int head, int heads, int NT, int t, int d) {
    const int jt = t >> 5, u = (t >> 4) & 1, w = t & 15;
    const int pos = 16 * ((w >> 2) & 1) + 8 * u + 4 * (w >> 3) + (w & 3);
    return ((((seq * heads + head) * NT + jt) * 64 + d) << 5) + pos;
}

// Plane layout, K-BLOCKED: element (row, k) of a (rows x K) operand sits at (k / 32) * rows * 32 + row * 32 + k % 32 --
// the 64-byte row pieces one 32-deep GEMM step needs from a tile's rows are CONTIGUOUS (a 64-row tile = one 4-KB run, every
// request a whole 128-byte line).  With plain row-major planes the same pieces are rows * K * 2 bytes apart: at K = 768 the
// stride is 1 536 B, at K = 3 072 it is 6 144 B, and the requests of a step pile onto a half / an eighth of an XCD's L2
// channels -- the first plane kernel of this round ran the query GEMMs at 4 TB/s of L2 traffic, no faster than the
// kernel it replaces.
__device__ __host__ __forceinline__ int64_t plane_index(int64_t row, int64_t k, int64_t rows) {
    return (k >> 5) * rows * 32 + row * 32 + (k & 31);
}

// x = p0 + p1 + p2 with p0 = RN_bf16(x), p1 = RN_bf16(x - p0), p2 = RN_bf16(x - p0 - p1): the split of split_store4
__device__ __forceinline__ void split3_bf16(float x, uint16_t& p0, uint16_t& p1, uint16_t& p2) {
    const uint32_t h = cvt_pk_bf16(x, 0.f) & 0xffffu;
    const float r = x - __uint_as_float(h << 16);
    const uint32_t l = cvt_pk_bf16(r, 0.f) & 0xffffu;
    const uint32_t t = cvt_pk_bf16(r - __uint_as_float(l << 16), 0.f) & 0xffffu;
    p0 = (uint16_t)h; p1 = (uint16_t)l; p2 = (uint16_t)t;
}
// Plane formats.  PF_BF16X2 / PF_BF16X3: two / three bf16 planes (bf16x3 / bf16x6 products).  PF_F16X2 (round 4): TWO fp16
// planes of x * scale -- h = RN_f16(x s), l = RN_f16(x s - h): round-to-nearest leaves |x s - h| <= 2^-12 |x s| and
// |x s - h - l| <= 2^-24 |x s|, i.e. two fp16 planes hold a whole fp32 significand (24 bits; the sign of l is the extra
// bit), so THREE products  al bh + ah bl + ah bh  (each exact in the fp32 accumulator, smallest first) reproduce the
// fp32 product to 2^-24 |a||b| -- what bf16x6 needs six products and three planes per operand for.  What fp16 lacks is
// exponent range: values are scaled by powers of two (exact) so that they sit well inside it -- activations by 2^4
// (|x| < 4 094 representable; below |x| = 2^-7 the low plane is an fp16 subnormal and the absolute error is 2^-29, the
// rounding fp32 itself applies to a value of 2^-5; the MFMA multiplies subnormal inputs exactly,
// tools/microbench/mfma_f16_denorm.hip), every weight matrix by the power of two that brings its largest entry to
// [2^13, 2^14); the epilogue multiplies the accumulator by the product of the inverse scales.  An activation beyond the
// range becomes Inf in the high plane and NaN downstream: the encode's epilogue reports non-finite output.
enum { PF_BF16X2 = 2, PF_BF16X3 = 3, PF_F16X2 = 16 };
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float kF16ActScale = 16.0f;
__device__ __forceinline__ void split2_f16(float xs, uint16_t& h, uint16_t& l) {
    const _Float16 hh = (_Float16)xs;                       // v_cvt_f16_f32, round to nearest even
    const _Float16 ll = (_Float16)(xs - (float)hh);         // the subtraction is exact in fp32
    h = __builtin_bit_cast(uint16_t, hh);
    l = __builtin_bit_cast(uint16_t, ll);
}
// p = plane 0 + plane_index(row, k, rows) of the element; fmt = one of PF_*
__device__ __forceinline__ void store_planes(uint16_t* p, int64_t plane, int fmt, float x) {
    uint16_t a, b, c;
    if (fmt == PF_F16X2) {
        split2_f16(x * kF16ActScale, a, b);
        p[0] = a; p[plane] = b;
        return;
    }
    split3_bf16(x, a, b, c);
    p[0] = a; p[plane] = b;
    if (fmt == PF_BF16X3) p[2 * plane] = c;
}

// four consecutive k of one row (k % 4 == 0: they share a 32-deep block): one 8-byte store per plane
__device__ __forceinline__ void store_planes4(uint16_t* p, int64_t plane, int fmt, const f32x4 v) {
    if (fmt == PF_F16X2) {
        uint16_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2_f16(v[j] * kF16ActScale, h[j], l[j]);
        *reinterpret_cast<uint2*>(p) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
        *reinterpret_cast<uint2*>(p + plane) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
        return;
    }
    uint16_t a[4], b[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3_bf16(v[j], a[j], b[j], c[j]);
    *reinterpret_cast<uint2*>(p) = make_uint2(a[0] | ((uint32_t)a[1] << 16), a[2] | ((uint32_t)a[3] << 16));
    *reinterpret_cast<uint2*>(p + plane) = make_uint2(b[0] | ((uint32_t)b[1] << 16), b[2] | ((uint32_t)b[3] << 16));
    if (fmt == PF_BF16X3) *reinterpret_cast<uint2*>(p + 2 * plane) = make_uint2(c[0] | ((uint32_t)c[1] << 16), c[2] | ((uint32_t)c[3] << 16));
}

// fp32 (rows x K, row-major) -> K-blocked planes, four elements per thread (weights at create time; activations whose
// producer is not fused)
static __global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes_,
                                                                 int64_t rows, int K, int64_t plane, int fmt, float scale,
                                                                 const float* __restrict__ colscale = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index of the row-major (rows x K) input
    if (i >= rows * (K / 4)) return;
    const int64_t row = i / (K / 4);
    const int k = (int)(i % (K / 4)) * 4;
    uint16_t* planes = planes_ + plane_index(row, k, rows);
    f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    if (colscale) v = v * *reinterpret_cast<const f32x4*>(colscale + k);   // gamma (.) W: one fp32 rounding, then the exact split
    if (fmt == PF_F16X2) {       // `scale`: kF16ActScale for activations, the matrix's own power of two for weights
        uint16_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2_f16(v[j] * scale, h[j], l[j]);
        *reinterpret_cast<uint2*>(planes) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
        *reinterpret_cast<uint2*>(planes + plane) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
        return;
    }
    const uint32_t h0 = cvt_pk_bf16(v[0], v[1]), h1 = cvt_pk_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(planes) = make_uint2(h0, h1);
    const float r0 = v[0] - __uint_as_float(h0 << 16), r1 = v[1] - __uint_as_float(h0 & 0xffff0000u);
    const float r2 = v[2] - __uint_as_float(h1 << 16), r3 = v[3] - __uint_as_float(h1 & 0xffff0000u);
    const uint32_t l0 = cvt_pk_bf16(r0, r1), l1 = cvt_pk_bf16(r2, r3);
    *reinterpret_cast<uint2*>(planes + plane) = make_uint2(l0, l1);
    if (fmt == PF_BF16X3) {
        const uint32_t t0 = cvt_pk_bf16(r0 - __uint_as_float(l0 << 16), r1 - __uint_as_float(l0 & 0xffff0000u));
        const uint32_t t1 = cvt_pk_bf16(r2 - __uint_as_float(l1 << 16), r3 - __uint_as_float(l1 & 0xffff0000u));
        *reinterpret_cast<uint2*>(planes + 2 * plane) = make_uint2(t0, t1);
    }
}

// max |colscale[k] W[n][k]| of a (rows x K) matrix -> the power-of-two scale of its folded fp16 planes
static __global__ __launch_bounds__(256) void max_abs_colscaled_kernel(const float* __restrict__ w, int64_t rows, int K,
                                                                       const float* __restrict__ colscale, unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    const int64_t n = rows * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i] * colscale[i % K]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

// u[n] = sum_k gamma_k W[n][k], c[n] = sum_k beta_k W[n][k] + b[n] (one wave per n: lane partial sums over k = lane, lane + 64,
// ..., then a fixed butterfly -- deterministic); out = [u (N) | c (N)]
static __global__ __launch_bounds__(256) void ln_fold_vectors_kernel(const float* __restrict__ w, int N, int K, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ bias,
                                                                     float* __restrict__ out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    float u = 0.f, c = 0.f;
    for (int k = lane; k < K; k += 64) { const float x = w[(int64_t)n * K + k]; u = fmaf(gamma[k], x, u); c = fmaf(beta[k], x, c); }
    for (int o = 32; o > 0; o >>= 1) { u += __shfl_xor(u, o, 64); c += __shfl_xor(c, o, 64); }
    if (lane == 0) { out[n] = u; out[N + n] = c + bias[n]; }
}

// number of work-groups to launch for gemm_planes_kernel's XCD-aware mapping (1-D grid)
inline int gemm_planes_grid(int M, int N, int bm, int bn, int ksplit) {
    const int TM = (M + bm - 1) / bm, TN = (N + bn - 1) / bn;
    const int64_t T = (int64_t)TM * TN * ksplit;
    return (int)(8 * ((T + 7) / 8));
}

// work-group -> (K slice z, n tile, m tile): XCD x (= blockIdx.x & 7 under round-robin placement: a speed heuristic, never
// needed for correctness) owns the contiguous range [x T / 8, (x + 1) T / 8) of the linear tile index t = c TM + m, c = the
// combined (K slice, n tile) index: every XCD gets T / 8 tiles to within one (dealing whole c ranges left XCDs with 3 against
// 2 n tiles on 18-tile-wide outputs, and two of eight XCDs idle on the 6-tile-wide ones of a passage batch), its L2 holds the
// weights of its ~TN / 8 n tiles, and co-resident work-groups share a weight tile.  Surplus work-groups return at once.
__device__ __forceinline__ bool gemm_planes_tile(int M, int N, int BM, int BN, int ksplit, int& z, int& m0, int& n0) {
    const int TM = (M + BM - 1) / BM, TN = (N + BN - 1) / BN;
    const int64_t T = (int64_t)TM * TN * ksplit;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int64_t t_lo = x * T / 8, t_hi = (x + 1) * T / 8;
    const int64_t t = t_lo + j;
    if (t >= t_hi) return false;
    const int c = (int)(t / TM);
    z = c / TN;
    n0 = (c % TN) * BN;
    m0 = (int)(t % TM) * BM;
    return true;
}

// ABL (tools/microbench/gemm_planes_bench.hip only): 1 = no DMAs (the MFMA / LDS-read side alone), 2 = no LDS reads and
// MFMAs (the DMA side alone); results are meaningless.  F16: the planes are PF_F16X2 (NS = 2), else bf16.
template <int WGM, int WGN, int WM, int WN, int NS, int STAGES, int ABL = 0, bool F16 = false>
static __global__ __launch_bounds__(64 * WGM * WGN) void gemm_planes_kernel(GemmPArgs g) {
    static_assert(!F16 || NS == 2, "the fp16 split has two planes");
    constexpr int PFMT = F16 ? PF_F16X2 : (NS == 3 ? PF_BF16X3 : PF_BF16X2);
    constexpr int NW = WGM * WGN;
    constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
    constexpr int PA = BM * 64, PB = BN * 64;                              // bytes per plane and stage
    constexpr int STAGEB = NS * (PA + PB);
    constexpr int NDMA = NS * (BM + BN) / 16;                              // wave-instructions per stage
    constexpr int DPW = (NDMA + NW - 1) / NW;       // a stage's DMAs dealt over the waves; a surplus slot repeats the last one
    static_assert(STAGES >= 2 && STAGES <= 4 && DPW * (STAGES - 2) <= 63, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char gplds[];
    // ---- work-group -> tile, XCD-aware (gemm_planes_tile).  Every XCD has its own 4-MB L2: dealt n-major over the whole
    // chip, every XCD would pull ALL of both operands through its L2 (FFN-out of a query batch: 33 MB of planes per XCD,
    // 264 MB from the Infinity Cache per GEMM, which then bounds it)
    const bool split = g.ksplit > 1;
    int z, m0, n0;
    if (!gemm_planes_tile(g.M, g.N, BM, BN, g.ksplit, z, m0, n0)) return;
    const int K_ = split ? g.K / g.ksplit : g.K;                // a multiple of 32 (host)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const int i = lane & 31, h = lane >> 5;
    float* C = g.C ? g.C + (split ? (int64_t)z * g.M * g.ldc : 0) : nullptr;
    // ---- loader: DMA instruction d = wave + NW j of a stage moves 16 rows x 64 B of one plane of one operand.  Lane l
    // lands at slot + 16 l: row 16 rb + (l >> 2), chunk position l & 3, i.e. it must FETCH chunk (l & 3) ^ ((l >> 4) & 3)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)gplds;
    const int64_t a_step = (int64_t)g.M * 64, b_step = (int64_t)g.N * 64;      // bytes from one 32-deep block to the next
    const char* a_base = reinterpret_cast<const char*>(g.A) + (split ? (int64_t)z * (K_ / 32) * a_step : 0);
    const char* b_base = reinterpret_cast<const char*>(g.B) + (split ? (int64_t)z * (K_ / 32) * b_step : 0);
    uint32_t voff[DPW];           // per-lane byte offset from the operand's base (plane, row, chunk)
    uint32_t loff[DPW];           // wave-uniform LDS offset inside a stage
    bool isb[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int d = wave + NW * j < NDMA ? wave + NW * j : NDMA - 1;
        const bool b_op = d >= NS * (BM / 16);
        const int dd = b_op ? d - NS * (BM / 16) : d;
        const int rows16 = b_op ? BN / 16 : BM / 16;
        const int q = dd / rows16, rb = dd % rows16;
        int row = (b_op ? n0 : m0) + 16 * rb + (lane >> 2);
        const int lim = b_op ? g.N : g.M;
        row = row < lim ? row : lim - 1;
        const uint32_t chunk = ((uint32_t)lane & 3u) ^ (((uint32_t)lane >> 4) & 3u);
        voff[j] = (uint32_t)(((int64_t)q * (b_op ? g.b_plane : g.a_plane) + (int64_t)row * 32) * 2) + chunk * 16u;
        loff[j] = (uint32_t)((b_op ? NS * PA : 0) + q * (b_op ? PB : PA) + rb * 1024);
        isb[j] = b_op;
    }
#define CLB_GP_ISSUE(KSTEP, BUF)                                                                                     \
    {                                                                                                                \
        const char* ab_ = a_base + (int64_t)(KSTEP) * a_step;                                                        \
        const char* bb_ = b_base + (int64_t)(KSTEP) * b_step;                                                        \
        if (ABL != 1) _Pragma("unroll") for (int j = 0; j < DPW; ++j)                                                \
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"                                        \
                         :: "v"(voff[j]), "s"(isb[j] ? bb_ : ab_), "s"(lds0 + (uint32_t)(BUF) * STAGEB + loff[j]) : "memory"); \
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nsteps = K_ / 32;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) CLB_GP_ISSUE(s, s)
    const int sw = (i >> 2) & 3;
    const unsigned char* As0 = gplds + (wr * 32 * WM + i) * 64;
    const unsigned char* Bs0 = gplds + NS * PA + (wc * 32 * WN + i) * 64;
    int buf = 0;
    for (int k = 0; k < nsteps; ++k) {
        // tile k has landed once at most the DMAs of the tiles issued after it are pending
        if (k + STAGES - 1 <= nsteps) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(DPW * (STAGES - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (k + STAGES - 1 < nsteps) {
            const int nb = buf == 0 ? STAGES - 1 : buf - 1;               // (k + STAGES - 1) % STAGES = (k - 1) % STAGES
            CLB_GP_ISSUE(k + STAGES - 1, nb)
        }
        const unsigned char* As = As0 + buf * STAGEB;
        const unsigned char* Bs = Bs0 + buf * STAGEB;
        if (ABL != 2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = ((2 * s + h) ^ sw) << 4;
            u32x4 av[NS][WM], bv[NS][WN];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
#pragma unroll
                for (int a = 0; a < WM; ++a) av[q][a] = *reinterpret_cast<const u32x4*>(As + q * PA + a * 32 * 64 + off);
#pragma unroll
                for (int b = 0; b < WN; ++b) bv[q][b] = *reinterpret_cast<const u32x4*>(Bs + q * PB + b * 32 * 64 + off);
            }
#define CLB_GP_MFMA(QA, QB)                                                                                            \
            c = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[QA][a]), __builtin_bit_cast(f16x8, bv[QB][b]), c, 0, 0, 0) \
                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[QA][a]), __builtin_bit_cast(bf16x8, bv[QB][b]), c, 0, 0, 0);
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    f32x16 c = acc[a][b];
                    if (NS == 3) { CLB_GP_MFMA(2, 0) CLB_GP_MFMA(0, 2) CLB_GP_MFMA(1, 1) }
                    CLB_GP_MFMA(1, 0) CLB_GP_MFMA(0, 1) CLB_GP_MFMA(0, 0)
                    acc[a][b] = c;
                }
#undef CLB_GP_MFMA
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
#undef CLB_GP_ISSUE
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
                        if (F16) v = v * g.out_scale;
                        if (g.epi & EPI_BIAS) v += g.bias[n];
                        if (g.epi & EPI_GELU) v = gelu_erf(v);
                        if (g.epi & EPI_RESID) v += g.R[(int64_t)m * g.ldc + n];
                        if (g.Cp) store_planes(g.Cp + plane_index(m, n, g.M), g.c_plane, PFMT, v);
                    }
                    if (C) C[(int64_t)m * g.ldc + n] = v;
                }
            }
        }
}

// The same product, second form (the default): (1) the accumulators are TRANSPOSED -- the MFMA takes the weight fragment as
// its row operand, so a lane owns ONE output row (m = lane & 31) and its registers run along n in groups of four consecutive
// columns: the epilogue is one 16-byte store (fp32) or one 8-byte store per plane for four outputs, where the first form
// issued 4-byte / 2-byte stores per element (the epilogue of a 19 200-row passage batch was 45 of 300 us); (2) the two
// 16-deep halves of a step are double-buffered in REGISTERS: the LDS reads of the next half are issued before the MFMAs of
// this one, and the single barrier of a step sits between the halves -- "tile k + 1 has landed and tile k has been read by
// everyone" -- after which the DMA of tile k + STAGES goes into the buffer just freed: the LDS latency is never exposed and a
// DMA has a whole step to land even with two buffers.  Requires N % 4 == 0 (host).
// LN = true (round 5): "LayerNorm without a pass of its own".  On a passage batch the stand-alone LayerNorm read and wrote
// 177 MB per call, 48.7 us x 24 = 8.6 % of the forward.  Here the Linear that PRODUCES a LayerNorm's input (attention output,
// FFN-out) stores the raw rows -- fp32 and planes -- and, per row and 64-column part, their (mean, M2) (stats_out: a lane owns
// one output row, so a part is 32 registers of this lane + 32 of its partner in the other lane half: two in-lane sums and two
// exchanges); the Linear that CONSUMES it (Q/K/V, FFN-in, the projection) multiplies the RAW planes with gamma (.) W and applies
//   LN(a) . W^T + b = rstd (a . (gamma (.) W)^T - mean u) + c,   u[n] = sum_k gamma_k W[n][k],  c[n] = sum_k beta_k W[n][k] + b[n]
// in its epilogue (two floats per row from ln_in, two vectors per column); the residual connection, which needs the normalised
// row itself, normalises the raw fp32 row it reads anyway (r_gamma / r_beta).  The flags are run-time (GemmPArgs) so that one
// instantiation per tile serves producer, consumer and both.
// LN = 1: the consuming side (ln_u / ln_in); LN = 2: the producing side (stats_out, and r_gamma / r_beta / ln_in for its residual).
// In both the statistics of the lane's rows are requested BEFORE the operand DMAs of the prologue and merged behind them (their
// latency hides under the first tile's), and the consumer loads the two column vectors of all its columns in one batch.
template <int WGM, int WGN, int WM, int WN, int NS, int STAGES, int ABL = 0, bool F16 = false, int LN = 0>
static __global__ __launch_bounds__(64 * WGM * WGN) void gemm_planes2_kernel(GemmPArgs g) {
    static_assert(!F16 || NS == 2, "the fp16 split has two planes");
    constexpr int PFMT = F16 ? PF_F16X2 : (NS == 3 ? PF_BF16X3 : PF_BF16X2);
    constexpr int NW = WGM * WGN;
    constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
    constexpr int PA = BM * 64, PB = BN * 64;
    constexpr int STAGEB = NS * (PA + PB);
    constexpr int NDMA = NS * (BM + BN) / 16;
    constexpr int DPW = (NDMA + NW - 1) / NW;
    static_assert(STAGES >= 2 && STAGES <= 5 && DPW * (STAGES - 1) <= 63, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char gplds[];
    const bool split = g.ksplit > 1;
    int z, m0, n0;
    if (!gemm_planes_tile(g.M, g.N, BM, BN, g.ksplit, z, m0, n0)) return;
    const int K_ = split ? g.K / g.ksplit : g.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const int i = lane & 31, h = lane >> 5;
    float* C = g.C ? g.C + (split ? (int64_t)z * g.M * g.ldc : 0) : nullptr;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)gplds;
    const int64_t a_step = (int64_t)g.M * 64, b_step = (int64_t)g.N * 64;
    const char* a_base = reinterpret_cast<const char*>(g.A) + (split ? (int64_t)z * (K_ / 32) * a_step : 0);
    const char* b_base = reinterpret_cast<const char*>(g.B) + (split ? (int64_t)z * (K_ / 32) * b_step : 0);
    uint32_t voff[DPW], loff[DPW];
    bool isb[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int d = wave + NW * j < NDMA ? wave + NW * j : NDMA - 1;
        const bool b_op = d >= NS * (BM / 16);
        const int dd = b_op ? d - NS * (BM / 16) : d;
        const int rows16 = b_op ? BN / 16 : BM / 16;
        const int q = dd / rows16, rb = dd % rows16;
        int row = (b_op ? n0 : m0) + 16 * rb + (lane >> 2);
        const int lim = b_op ? g.N : g.M;
        row = row < lim ? row : lim - 1;
        const uint32_t chunk = ((uint32_t)lane & 3u) ^ (((uint32_t)lane >> 4) & 3u);
        voff[j] = (uint32_t)(((int64_t)q * (b_op ? g.b_plane : g.a_plane) + (int64_t)row * 32) * 2) + chunk * 16u;
        loff[j] = (uint32_t)((b_op ? NS * PA : 0) + q * (b_op ? PB : PA) + rb * 1024);
        isb[j] = b_op;
    }
#define CLB_GP2_ISSUE(KSTEP, BUF)                                                                                    \
    {                                                                                                                \
        const char* ab_ = a_base + (int64_t)(KSTEP) * a_step;                                                        \
        const char* bb_ = b_base + (int64_t)(KSTEP) * b_step;                                                        \
        if (ABL != 1 && ABL != 3) _Pragma("unroll") for (int j = 0; j < DPW; ++j)                                                \
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"                                        \
                         :: "v"(voff[j]), "s"(isb[j] ? bb_ : ab_), "s"(lds0 + (uint32_t)(BUF) * STAGEB + loff[j]) : "memory"); \
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nsteps = K_ / 32;
    // LN: (mean, 1 / sqrt(var + eps)) of the rows this lane finishes -- A's rows (consumer) or the residual's (producer): the
    // partial statistics are requested here, in front of the prologue's DMAs (vector memory operations complete in order, so
    // the first tile's wait covers them), and merged right behind the DMAs' issue
    float ln_mean_[WM], ln_rstd_[WM];
    f32x2_t ln_sv[LN ? WM : 1][LN ? kLnMaxParts : 1];
    const bool ln_rows = LN != 0 && !split && g.ln_in != nullptr && (g.ln_u != nullptr || g.r_gamma != nullptr);
    if (LN != 0 && ln_rows) {
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            int m = m0 + (wr * WM + a) * 32 + i;
            m = m < g.M ? m : g.M - 1;
            const f32x2_t* st = reinterpret_cast<const f32x2_t*>(g.ln_in + (int64_t)m * g.ln_parts * 2);
#pragma unroll
            for (int p = 0; p < kLnMaxParts; ++p) ln_sv[a][p] = st[p < g.ln_parts ? p : g.ln_parts - 1];
        }
    }
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (s < nsteps) CLB_GP2_ISSUE(s, s)
#pragma unroll
    for (int a = 0; a < WM; ++a) { ln_mean_[a] = 0.f; ln_rstd_[a] = 1.f; }
    if (LN != 0 && ln_rows) {
#pragma unroll
        for (int a = 0; a < WM; ++a) ln_merge_stats(ln_sv[LN ? a : 0], g.ln_parts, g.ln_width, g.ln_eps, ln_mean_[a], ln_rstd_[a]);
    }
    const int sw = (i >> 2) & 3;
    const unsigned char* As0 = gplds + (wr * 32 * WM + i) * 64;
    const unsigned char* Bs0 = gplds + NS * PA + (wc * 32 * WN + i) * 64;
    u32x4 fa[2][NS][WM], fb[2][NS][WN];
#define CLB_GP2_READ(W, BUF, S)                                                                                       \
    if (ABL != 2) {                                                                                                   \
        const int off_ = ((2 * (S) + h) ^ sw) << 4;                                                                   \
        const unsigned char* as_ = As0 + (BUF) * STAGEB + off_;                                                       \
        const unsigned char* bs_ = Bs0 + (BUF) * STAGEB + off_;                                                       \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) {                                                              \
            _Pragma("unroll") for (int a = 0; a < WM; ++a) fa[W][q][a] = *reinterpret_cast<const u32x4*>(as_ + q * PA + a * 32 * 64); \
            _Pragma("unroll") for (int b = 0; b < WN; ++b) fb[W][q][b] = *reinterpret_cast<const u32x4*>(bs_ + q * PB + b * 32 * 64); \
        }                                                                                                             \
    }
    // D' = W-fragment (rows n) x A-fragment (columns m): registers along n, lanes along m
#define CLB_GP2_ONE(W, QA, QB)                                                                                        \
    c = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[W][QB][b]), __builtin_bit_cast(f16x8, fa[W][QA][a]), c, 0, 0, 0) \
            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[W][QB][b]), __builtin_bit_cast(bf16x8, fa[W][QA][a]), c, 0, 0, 0);
#define CLB_GP2_MFMA(W)                                                                                               \
    if (ABL != 2) {                                                                                                   \
        _Pragma("unroll") for (int a = 0; a < WM; ++a)                                                                \
            _Pragma("unroll") for (int b = 0; b < WN; ++b) {                                                          \
                f32x16 c = acc[a][b];                                                                                 \
                if (NS == 3) { CLB_GP2_ONE(W, 2, 0) CLB_GP2_ONE(W, 0, 2) CLB_GP2_ONE(W, 1, 1) }                       \
                CLB_GP2_ONE(W, 1, 0) CLB_GP2_ONE(W, 0, 1) CLB_GP2_ONE(W, 0, 0)                                        \
                acc[a][b] = c;                                                                                        \
            }                                                                                                         \
    }
    // tile 0 has landed once at most the groups issued after it are pending
    if (nsteps >= STAGES) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(DPW * (STAGES - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    CLB_GP2_READ(0, 0, 0)
    int buf = 0;
    for (int k = 0; k < nsteps; ++k) {
        CLB_GP2_READ(1, buf, 1)
        __builtin_amdgcn_sched_barrier(0);
        CLB_GP2_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        int nbuf = buf + 1 == STAGES ? 0 : buf + 1;
        if (k + 1 < nsteps) {
            // my reads of tile k are complete (lgkmcnt) and my DMAs of tile k + 1 have landed (vmcnt); after the barrier that
            // holds for every wave: buffer `buf` is free for tile k + STAGES and tile k + 1 can be read
            if (ABL == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (k + STAGES <= nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(DPW * (STAGES - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (k + STAGES < nsteps) CLB_GP2_ISSUE(k + STAGES, buf)
            CLB_GP2_READ(0, nbuf, 0)
            __builtin_amdgcn_sched_barrier(0);
        }
        CLB_GP2_MFMA(1)
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }
#undef CLB_GP2_ISSUE
#undef CLB_GP2_READ
#undef CLB_GP2_ONE
#undef CLB_GP2_MFMA
    // D' layout: column (lane & 31) = m, row (r & 3) + 8 * (r >> 2) + 4 * h = n: four consecutive n per register group
    const bool qkv_att = F16 && !split && (g.epi & EPI_QKV_ATT);
    const int att_nt = qkv_att ? (g.att_L + 31) >> 5 : 0;
    const bool ln_fold = LN == 1 && !split && g.ln_u != nullptr;      // the A rows are raw: fold their LayerNorm into this product
    const bool ln_res = LN == 2 && !split && g.r_gamma != nullptr;    // the residual rows are raw: normalise them on the fly
    const bool ln_out = LN == 2 && !split && g.stats_out != nullptr;  // leave the partial statistics of the output rows
    // consumer: u and c of ALL the lane's columns in one batch (they do not depend on the row: one exposed round trip per tile,
    // what the plain kernel pays for its bias)
    f32x4 col_u[LN == 1 ? WN : 1][4], col_c[LN == 1 ? WN : 1][4];
    if (LN == 1 && ln_fold) {
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + (wc * WN + b) * 32 + 8 * gq + 4 * h;
                const int nn = n < g.N ? n : 0;
                col_u[b][gq] = *reinterpret_cast<const f32x4*>(g.ln_u + nn);
                col_c[b][gq] = *reinterpret_cast<const f32x4*>(g.bias + nn);
            }
    }
#pragma unroll
    for (int a = 0; a < WM; ++a) {
        const int m = m0 + (wr * WM + a) * 32 + i;
        // (no early exit for rows past M when statistics are exchanged between the lane halves below: both halves hold the same m)
        if (m >= g.M) continue;
        int64_t vrow = 0;                 // EPI_QKV_ATT: vt_index of (this token, head 0, d 0)
        if (qkv_att) {
            const uint32_t seq = g.att_seq ? (uint32_t)g.att_seq[m] : (uint32_t)m / (uint32_t)g.att_L;
            vrow = vt_index(seq, 0, g.att_heads, att_nt, g.att_pos ? g.att_pos[m] : m - (int)seq * g.att_L, 0);
        }
        const float ln_mean = ln_mean_[a], ln_rstd = ln_rstd_[a];
        float psum = 0.f;                 // ln_out: this lane's sum over the current 64-column part
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            // LN kernels: everything the four groups of this 32-column tile read -- the per-column vectors and the residual row --
            // is requested BEFORE the first group is finished and stored: behind a store the compiler cannot hoist the next
            // group's loads (the pointers of GemmPArgs may alias), and 32 load -> use -> store rounds per row tile were a chain
            // of 32 memory latencies (the first fold kernels: +37 ... +55 us per launch)
            f32x4 pre_b[LN == 2 ? 4 : 1], pre_r[LN == 2 ? 4 : 1], pre_g[LN == 2 ? 4 : 1], pre_t[LN == 2 ? 4 : 1];
            if (LN == 2 && !split) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = n0 + (wc * WN + b) * 32 + 8 * gq + 4 * h;
                    const int nn = n < g.N ? n : 0;
                    if (g.epi & EPI_BIAS) pre_b[gq] = *reinterpret_cast<const f32x4*>(g.bias + nn);
                    if (g.epi & EPI_RESID) pre_r[gq] = *reinterpret_cast<const f32x4*>(g.R + (int64_t)m * g.ldc + nn);
                    if (ln_res) { pre_g[gq] = *reinterpret_cast<const f32x4*>(g.r_gamma + nn); pre_t[gq] = *reinterpret_cast<const f32x4*>(g.r_beta + nn); }
                }
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + (wc * WN + b) * 32 + 8 * gq + 4 * h;
                if (n >= g.N) continue;
                f32x4 v = {acc[a][b][4 * gq], acc[a][b][4 * gq + 1], acc[a][b][4 * gq + 2], acc[a][b][4 * gq + 3]};
                if (!split) {
                    if (F16) v = v * g.out_scale;
                    if (LN == 1 && ln_fold) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j] - ln_mean * col_u[b][gq][j], ln_rstd, col_c[b][gq][j]);
                    } else if (g.epi & EPI_BIAS) v = v + (LN == 2 ? pre_b[gq] : *reinterpret_cast<const f32x4*>(g.bias + n));
                    if (g.epi & EPI_GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
                    if (g.epi & EPI_RESID) {
                        f32x4 r = LN == 2 ? pre_r[gq] : *reinterpret_cast<const f32x4*>(g.R + (int64_t)m * g.ldc + n);
                        if (LN == 2 && ln_res) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) r[j] = fmaf((r[j] - ln_mean) * ln_rstd, pre_g[gq][j], pre_t[gq][j]);
                        }
                        v = v + r;
                    }
                    if (qkv_att) {
                        if (n < 2 * g.att_H) store_planes4(g.Cp + plane_index(m, n, g.M), g.c_plane, PFMT, v);
                        else {            // V: (head, d) of column n - 2H; the four d of this group are 64 bytes apart in a key row
                            const int d0 = n - 2 * g.att_H;
                            uint16_t* vp = g.Vt + vrow + (((int64_t)(d0 >> 6) * att_nt * 64 + (d0 & 63)) << 5);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                uint16_t hi, lo;
                                split2_f16(v[j] * kF16ActScale, hi, lo);
                                vp[32 * j] = hi;
                                vp[32 * j + g.vt_plane] = lo;
                            }
                        }
                    } else if (g.Cp) store_planes4(g.Cp + plane_index(m, n, g.M), g.c_plane, PFMT, v);
                    if (LN == 2 && ln_out) {   // keep the finished values for the second (deviation) sweep
                        acc[a][b][4 * gq] = v[0]; acc[a][b][4 * gq + 1] = v[1]; acc[a][b][4 * gq + 2] = v[2]; acc[a][b][4 * gq + 3] = v[3];
                        psum += (v[0] + v[1]) + (v[2] + v[3]);
                    }
                }
                if (C) *reinterpret_cast<f32x4*>(C + (int64_t)m * g.ldc + n) = v;
            }
            if (LN == 2 && ln_out && (b & 1)) {
                // part = the 64 columns of tiles b - 1, b: 32 values here, 32 in lane i of the other half (same row m)
                static_assert(LN != 2 || WN % 2 == 0, "a 64-column part is two adjacent 32-column tiles of one wave");
                const float mean = (psum + __shfl_xor(psum, 32, 64)) * (1.0f / 64.0f);
                float q = 0.f;
#pragma unroll
                for (int bb = b - 1; bb <= b; ++bb)
#pragma unroll
                    for (int r_ = 0; r_ < 16; ++r_) { const float d = acc[a][bb < 0 ? 0 : bb][r_] - mean; q = fmaf(d, d, q); }
                q += __shfl_xor(q, 32, 64);
                const int part = (n0 + (wc * WN + b - 1) * 32) >> 6;
                if (h == 0 && n0 + (wc * WN + b) * 32 < g.N)
                    *reinterpret_cast<f32x2_t*>(g.stats_out + ((int64_t)m * (g.N >> 6) + part) * 2) = f32x2_t{mean, q};
                psum = 0.f;
            }
        }
    }
}

// split-K second pass: C = epilogue(sum over the K slices, in slice order -- deterministic)
static __global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int ksplit,
                                                                       int64_t M, int N, float* __restrict__ C,
                                                                       const float* __restrict__ bias,
                                                                       const float* __restrict__ R, float scale, int epi,
                                                                       uint16_t* __restrict__ Cp = nullptr, int64_t c_plane = 0,
                                                                       int ns = 0, uint16_t* __restrict__ Vt = nullptr,
                                                                       int64_t vt_plane = 0, int att_L = 0, int att_H = 0,
                                                                       int att_heads = 0, const int32_t* __restrict__ att_seq = nullptr,
                                                                       const int32_t* __restrict__ att_pos = nullptr) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int n = (int)(idx % N);
    float v = 0.f;
    for (int z0 = 0; z0 < ksplit; z0 += 8) {      // eight slices requested at a time, added in slice order
        float pv[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) pv[z] = z0 + z < ksplit ? part[(size_t)(z0 + z) * M * N + idx] : 0.f;
#pragma unroll
        for (int z = 0; z < 8; ++z)
            if (z0 + z < ksplit) v = (z0 + z == 0) ? pv[z] : v + pv[z];
    }
    v = v * scale;
    if (epi & EPI_BIAS) v += bias[n];
    if (epi & EPI_GELU) v = gelu_erf(v);
    if (epi & EPI_RESID) v += R[idx];
    if (C) C[idx] = v;
    if (Vt) {      // EPI_QKV_ATT (PF_F16X2): Q | K planes of a (M x 2 att_H) matrix, V key-blocked -- attention_f16_kernel's operands
        const int64_t m = idx / N;
        if (n < 2 * att_H) store_planes(Cp + plane_index(m, n, M), c_plane, PF_F16X2, v);
        else {
            const int d0 = n - 2 * att_H;
            const int64_t seq = att_seq ? att_seq[m] : m / att_L;
            uint16_t hi, lo;
            split2_f16(v * kF16ActScale, hi, lo);
            const int64_t at = vt_index(seq, d0 >> 6, att_heads, (att_L + 31) >> 5, att_pos ? att_pos[m] : (int)(m - seq * att_L), d0 & 63);
            Vt[at] = hi;
            Vt[at + vt_plane] = lo;
        }
        return;
    }
    if (Cp) store_planes(Cp + plane_index(idx / N, n, M), c_plane, ns, v);   // the bf16 planes the next Linear reads
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
                                                                          const float* __restrict__ beta, float eps,
                                                                          uint16_t* __restrict__ Cp = nullptr, int64_t c_plane = 0,
                                                                          int ns = 0) {
    __shared__ float red[2][4];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t slice = (size_t)M * N;
    // every slice's value is requested before the first is used (ksplit <= 8: a rolled loop of load -> wait -> add made the
    // kernel a chain of ksplit memory latencies: 13 us for 1 024 rows); the additions stay in slice order
    float pv[8][NR];
#pragma unroll
    for (int z = 0; z < 8; ++z)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int n = tid + 256 * j;
            pv[z][j] = (z < ksplit && n < N) ? part[z * slice + t * N + n] : 0.f;
        }
    float rres[NR], rbias[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int n = tid + 256 * j;
        rres[j] = ((epi & EPI_RESID) && n < N) ? R[t * N + n] : 0.f;
        rbias[j] = ((epi & EPI_BIAS) && n < N) ? bias[n] : 0.f;
    }
    float v[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        v[j] = pv[0][j];
#pragma unroll
        for (int z = 1; z < 8; ++z)
            if (z < ksplit) v[j] = v[j] + pv[z][j];
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int n = tid + 256 * j;
        if (n < N) {
            float x = v[j] * scale;
            if (epi & EPI_BIAS) x += rbias[j];
            if (epi & EPI_GELU) x = gelu_erf(x);
            if (epi & EPI_RESID) x += rres[j];
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
        if (n < N) {
            const float y = (v[j] - mean) * rstd * gamma[n] + beta[n];
            C[t * N + n] = y;
            if (Cp) store_planes(Cp + plane_index(t, n, M), c_plane, ns, y);
        }
    }
}

// The same pass with four CONSECUTIVE columns per thread (N % 4 == 0, N <= 1 024): 16-byte loads of the slices, the
// residual and the LayerNorm parameters, one 16-byte store of the row and one 8-byte store per plane (the strided form
// above: 4-byte loads and 2-byte plane stores, 6.7 us per 1 024 x 768 pass; the element arithmetic is the same, the
// mean / variance partial sums associate differently).
static __global__ __launch_bounds__(256) void gemm_splitk_reduce_ln4_kernel(const float* __restrict__ part, int ksplit,
                                                                           int64_t M, int N, float* __restrict__ C,
                                                                           const float* __restrict__ bias,
                                                                           const float* __restrict__ R, float scale, int epi,
                                                                           const float* __restrict__ gamma,
                                                                           const float* __restrict__ beta, float eps,
                                                                           uint16_t* __restrict__ Cp, int64_t c_plane, int fmt) {
    __shared__ float red[2][4];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = 4 * tid;
    const bool live = n < N;
    const size_t slice = (size_t)M * N;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v = zero;
    for (int z0 = 0; z0 < ksplit; z0 += 8) {      // eight slices requested at a time, added in slice order (ksplit <= 32)
        f32x4 pv[8];
#pragma unroll
        for (int z = 0; z < 8; ++z)
            pv[z] = (z0 + z < ksplit && live) ? *reinterpret_cast<const f32x4*>(part + (z0 + z) * slice + t * N + n) : zero;
#pragma unroll
        for (int z = 0; z < 8; ++z)
            if (z0 + z < ksplit) v = (z0 + z == 0) ? pv[z] : v + pv[z];
    }
    const f32x4 rres = ((epi & EPI_RESID) && live) ? *reinterpret_cast<const f32x4*>(R + t * N + n) : zero;
    const f32x4 rbias = ((epi & EPI_BIAS) && live) ? *reinterpret_cast<const f32x4*>(bias + n) : zero;
    const f32x4 gm = live ? *reinterpret_cast<const f32x4*>(gamma + n) : zero;
    const f32x4 bt = live ? *reinterpret_cast<const f32x4*>(beta + n) : zero;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = v[j] * scale;
        if (epi & EPI_BIAS) x += rbias[j];
        if (epi & EPI_GELU) x = gelu_erf(x);
        if (epi & EPI_RESID) x += rres[j];
        v[j] = live ? x : 0.f;
        sum += v[j];
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[0][wave] = sum;
    __syncthreads();
    const float mean = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) / (float)N;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (live) { const float c = v[j] - mean; var += c * c; }
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    if (lane == 0) red[1][wave] = var;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / (float)N + eps);
    if (!live) return;
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = (v[j] - mean) * rstd * gm[j] + bt[j];
    *reinterpret_cast<f32x4*>(C + t * N + n) = y;
    if (Cp) store_planes4(Cp + plane_index(t, n, M), c_plane, fmt, y);
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
                                                                    float* __restrict__ out, int* __restrict__ err,
                                                                    uint16_t* __restrict__ outp = nullptr, int64_t o_plane = 0,
                                                                    int ns = 0, const int32_t* __restrict__ tok_pos = nullptr) {
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= n_tok) return;
    int id = ids[t] - 1;
    if (id < 0 || id >= vocab) { if (lane == 0) atomicOr(err, 1); id = 0; }
    const int l = tok_pos ? tok_pos[t] : (int)(t % L);       // packed batches carry every token's position
    if (H <= 64 * 16) {      // the row stays in registers between the three passes (it was written and re-read twice: 17 us)
        float v[16];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = lane + 64 * j;
            v[j] = d < H ? word[(int64_t)id * H + d] + pos[(int64_t)l * H + d] + type0[d] : 0.f;
            if (d < H) sum += v[j];
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)H;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (lane + 64 * j < H) { const float c = v[j] - mean; var += c * c; }
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        const float rstd = 1.0f / sqrtf(var / (float)H + eps);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = lane + 64 * j;
            if (d < H) {
                const float y = (v[j] - mean) * rstd * gamma[d] + beta[d];
                out[t * H + d] = y;
                if (outp) store_planes(outp + plane_index(t, d, n_tok), o_plane, ns, y);
            }
        }
        return;
    }
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
    for (int d = lane; d < H; d += 64) {
        const float y = (out[t * H + d] - mean) * rstd * gamma[d] + beta[d];
        out[t * H + d] = y;
        if (outp) store_planes(outp + plane_index(t, d, n_tok), o_plane, ns, y);
    }
}

// in-place LayerNorm over rows of length H.  One wave per row.
static __global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, int64_t rows, int H,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              uint16_t* __restrict__ xp = nullptr, int64_t x_plane = 0,
                                                              int ns = 0) {
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
    for (int d = lane; d < H; d += 64) {
        const float y = (row[d] - mean) * rstd * gamma[d] + beta[d];
        row[d] = y;
        if (xp) store_planes(xp + plane_index(t, d, rows), x_plane, ns, y);
    }
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
                                                                    float* __restrict__ ctx, int L, int H, float scale,
                                                                    uint16_t* __restrict__ ctxp = nullptr, int64_t c_plane = 0,
                                                                    int ns = 0) {
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
    const int64_t obase = n * L * (int64_t)H + head * 64 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (q < L) {
            if (ctxp) {      // the output projection reads bf16 planes (gemm_planes_kernel): no fp32 copy is kept
                const int64_t trow = n * L + q, rows_ = (int64_t)gridDim.z * L;
                store_planes(ctxp + plane_index(trow, head * 64 + i, rows_), c_plane, ns, o0[r]);
                store_planes(ctxp + plane_index(trow, head * 64 + 32 + i, rows_), c_plane, ns, o1[r]);
            } else {
                ctx[obase + (int64_t)q * H] = o0[r];
                ctx[obase + (int64_t)q * H + 32] = o1[r];
            }
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
                                                                        float* __restrict__ ctx, int L, int H, float scale,
                                                                        uint16_t* __restrict__ ctxp = nullptr, int64_t c_plane = 0,
                                                                        int ns = 0) {
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
    const int64_t obase = n * L * (int64_t)H + head * 64 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qa = (r & 3) + 8 * (r >> 2);
        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa));
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa + 4));
        const float sc = h ? s1 : s0;
        const int q = q0 + qa + 4 * h;
        if (q < L) {
            if (ctxp) {
                const int64_t trow = n * L + q, rows_ = (int64_t)gridDim.z * L;
                store_planes(ctxp + plane_index(trow, head * 64 + i, rows_), c_plane, ns, o0[r] * sc);
                store_planes(ctxp + plane_index(trow, head * 64 + 32 + i, rows_), c_plane, ns, o1[r] * sc);
            } else {
                ctx[obase + (int64_t)q * H] = o0[r] * sc;
                ctx[obase + (int64_t)q * H + 32] = o1[r] * sc;
            }
        }
    }
}

// The same online-softmax attention on the 16-bit matrix pipe (f16x3, the default with the f16x3 Linear layers): Q, K and V
// arrive as the fp16 hi / lo planes the Q/K/V projection's epilogue wrote (EPI_QKV_ATT: Q and K K-blocked -- a lane's 32
// features are one 64-byte piece per plane --, V key-blocked: vt_index), so no operand is split here and every load is a
// contiguous run across the wave.  S^T tile = K . Q^T: 4 steps of v_mfma_f32_32x32x16_f16 x 3 products (lo.hi, hi.lo, hi.hi)
// instead of 32 fp32 MFMAs; the probabilities (<= 1, scaled by 2^10 so that their low plane stays normal) are split in
// registers, P.V = 2 steps x 2 output halves x 3 products.  24 matrix instructions of 8 passes per 32 x 32 key tile against 64
// of 16 passes: the kernel is now bound by its exponentials and splits, not by the matrix pipe.  Operand scales (2^4 on Q, K
// and V, 2^10 on P) are divided out of the scores and of the output.  Tile layout, masks, running maximum / sum and the
// rescale are attention_online_kernel's.  grid = (ceil(L / 32), heads, N), block = 64.
// QB = query blocks of 32 per wave: 2 for long sequences -- the K / V fragments of a key tile are loaded ONCE for 64 queries
// (the kernel is bound by those loads: 16 KB per tile and wave through L2), at two instead of three waves per SIMD.
template <int QB>
static __global__ __launch_bounds__(64, QB == 1 ? 3 : 2) void attention_f16_kernel(const uint16_t* __restrict__ qk, int64_t qk_plane, int64_t rows,
                                                                                   const uint16_t* __restrict__ vt, int64_t vt_plane,
                                                                                   const uint8_t* __restrict__ mask, int L, int H, float scale,
                                                                                   uint16_t* __restrict__ ctxp, int64_t c_plane, int ns,
                                                                                   const int32_t* __restrict__ cu = nullptr) {
    // cu (packed batches): sequence n owns rows cu[n] .. cu[n + 1] - 1, every one of them attended (no mask array); the
    // key-blocked V buffer keeps its per-sequence stride of ceil(Lmax / 32) tiles (L = Lmax here)
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32 * QB, head = blockIdx.y, heads = gridDim.y;
    const int64_t n = blockIdx.z;
    const int nt_layout = (L + 31) >> 5;
    const int64_t row0 = cu ? (int64_t)cu[n] : n * L;
    if (cu) {
        L = cu[n + 1] - cu[n];
        if (q0 >= L) return;
    }
    const uint8_t* mk = cu ? nullptr : mask + n * L;
    const int nt = (L + 31) >> 5;
    u32x4 qh[QB][4], ql[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + 32 * qb + i;
        const int64_t qrow = row0 + (qi < L ? qi : L - 1);
        const uint16_t* qp = qk + plane_index(qrow, head * 64 + 32 * h, rows);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[qb][s] = reinterpret_cast<const u32x4*>(qp)[s];
            ql[qb][s] = reinterpret_cast<const u32x4*>(qp + qk_plane)[s];
        }
    }
    f32x16 o0[QB], o1[QB];
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = kNegInf; l[qb] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[qb][r] = 0.f; o1[qb][r] = 0.f; }
    }
    const float sscale = scale * (1.0f / (kF16ActScale * kF16ActScale)) * 1.44269504088896340736f;     // -> log2 domain
    constexpr float kPScale = 1024.0f;
    const uint16_t* vbase = vt + ((((n * heads + head) * nt_layout) * 64 + i) << 5) + 16 * h;
    for (int jt = 0; jt < nt; ++jt) {
        const int key = 32 * jt + i;
        const int krow = key < L ? key : L - 1;
        const uint32_t valid = (uint32_t)__builtin_amdgcn_ballot_w64(h == 0 && key < L && (!mk || mk[krow] != 0));
        if (valid == 0u) continue;
        const uint16_t* kp = qk + plane_index(row0 + krow, H + head * 64 + 32 * h, rows);
        u32x4 kh[4], kl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kh[s] = reinterpret_cast<const u32x4*>(kp)[s];
            kl[s] = reinterpret_cast<const u32x4*>(kp + qk_plane)[s];
        }
        // the V fragments of this tile: two output halves x (hi, lo) x two steps, requested before the score MFMAs
        const uint16_t* vp = vbase + ((int64_t)jt * 64 << 5);
        u32x4 vh[2][2], vl[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                vh[c][u] = reinterpret_cast<const u32x4*>(vp + (c * 32 << 5))[u];
                vl[c][u] = reinterpret_cast<const u32x4*>(vp + (c * 32 << 5) + vt_plane)[u];
            }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kl[s]), __builtin_bit_cast(f16x8, qh[qb][s]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kh[s]), __builtin_bit_cast(f16x8, ql[qb][s]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kh[s]), __builtin_bit_cast(f16x8, qh[qb][s]), st, 0, 0, 0);
            }
            // scores in the base-2 domain: s2 = s . scale . log2(e), so that a probability is ONE v_exp_f32 of a difference
            // (exp2(-inf) = 0 covers masked keys without a second select)
            float tmax = kNegInf;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kb = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = ((valid >> kb) & 1u) ? st[r] * sscale : kNegInf;
                st[r] = v;
                tmax = fmaxf(tmax, v);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m[qb], tmax);                      // finite: the tile has a valid key
            const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);   // m = -inf (first tile): 0; maximum unchanged: 1
            float psum = 0.f;
            u32x4 ph[2], pl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const float e0 = __builtin_amdgcn_exp2f(st[8 * u + 2 * j2] - m_new);
                    const float e1 = __builtin_amdgcn_exp2f(st[8 * u + 2 * j2 + 1] - m_new);
                    psum += e0;
                    psum += e1;
                    const f16x2 hh = {(_Float16)(e0 * kPScale), (_Float16)(e1 * kPScale)};
                    const f16x2 ll = {(_Float16)(e0 * kPScale - (float)hh[0]), (_Float16)(e1 * kPScale - (float)hh[1])};
                    ph[u][j2] = __builtin_bit_cast(uint32_t, hh);
                    pl[u][j2] = __builtin_bit_cast(uint32_t, ll);
                }
            l[qb] = l[qb] * alpha + psum;
            if (__builtin_amdgcn_ballot_w64(m_new > m[qb]) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qa = (r & 3) + 8 * (r >> 2);
                    const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), qa));
                    const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), qa + 4));
                    const float a = h ? a1 : a0;
                    o0[qb][r] *= a;
                    o1[qb][r] *= a;
                }
            }
            m[qb] = m_new;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pl[u]), __builtin_bit_cast(f16x8, vh[0][u]), o0[qb], 0, 0, 0);
                o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[u]), __builtin_bit_cast(f16x8, vl[0][u]), o0[qb], 0, 0, 0);
                o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[u]), __builtin_bit_cast(f16x8, vh[0][u]), o0[qb], 0, 0, 0);
                o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pl[u]), __builtin_bit_cast(f16x8, vh[1][u]), o1[qb], 0, 0, 0);
                o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[u]), __builtin_bit_cast(f16x8, vl[1][u]), o1[qb], 0, 0, 0);
                o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[u]), __builtin_bit_cast(f16x8, vh[1][u]), o1[qb], 0, 0, 0);
            }
        }
    }
    const int64_t rows_ = rows;                         // the ctx planes have as many rows as the Q | K planes
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lsum = l[qb] + __shfl_xor(l[qb], 32, 64);
        const float inv = lsum > 0.f ? 1.0f / (lsum * kPScale * kF16ActScale) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qa = (r & 3) + 8 * (r >> 2);
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa + 4));
            const float sc = h ? s1 : s0;
            const int q = q0 + 32 * qb + qa + 4 * h;
            if (q < L) {
                const int64_t trow = row0 + q;
                store_planes(ctxp + plane_index(trow, head * 64 + i, rows_), c_plane, ns, o0[qb][r] * sc);
                store_planes(ctxp + plane_index(trow, head * 64 + 32 + i, rows_), c_plane, ns, o1[qb][r] * sc);
            }
        }
    }
}

// Round 5: the same attention with the K / V tiles of a (sequence, head) SHARED by its query blocks through LDS.  In the kernel
// above every wave pulls all key tiles of its sequence through L2 on its own -- 16 KB per tile and wave, 1.2 GB per layer on a
// 64 x 300 passage batch, and the kernel is bound by those loads (halving them bought 20 %, cutting a third of its vector
// instructions 3 %).  Here a work-group of NW waves owns 32 QB NW queries of one (sequence, head) -- a whole 300-token passage
// at QB = 2, NW = 5 -- and stages every key tile ONCE: 2 planes x (K: 2 feature blocks x 32 keys x 64 B, V: 64 d x 64 B) = 16 KB,
// double-buffered (global -> registers while the previous tile is multiplied, registers -> LDS behind it, one barrier per tile);
// rows are padded to 80 bytes, so the ds_read_b128 of 16 consecutive lanes touch 64 different banks.  Every wave then reads
// exactly the fragments it used to load: the same MFMAs on the same operands in the same order -- bit-identical output.
// A wave whose queries lie past the end of a (packed) sequence still stages and meets the barriers.  grid = (ceil(L / (32 QB
// NW)), heads, N), block = 64 NW.
template <int QB, int NW>
static __global__ __launch_bounds__(64 * NW) void attention_f16_lds_kernel(const uint16_t* __restrict__ qk, int64_t qk_plane, int64_t rows,
                                                                          const uint16_t* __restrict__ vt, int64_t vt_plane,
                                                                          const uint8_t* __restrict__ mask, int L, int H, float scale,
                                                                          uint16_t* __restrict__ ctxp, int64_t c_plane, int ns,
                                                                          const int32_t* __restrict__ cu = nullptr) {
    constexpr int NT = 64 * NW;
    constexpr int kRow = 80;                              // padded LDS row (64 B of data)
    constexpr int kKBytes = 2 * 2 * 32 * kRow, kVBytes = 2 * 64 * kRow, kStage = kKBytes + kVBytes;     // 20 480 B per tile
    constexpr int kChunks = 1024;                         // 16-byte pieces of a tile: 512 of K, 512 of V
    constexpr int CPT = (kChunks + NT - 1) / NT;          // pieces per thread
    __shared__ __attribute__((aligned(16))) unsigned char tiles[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y, heads = gridDim.y;
    const int64_t n = blockIdx.z;
    const int nt_layout = (L + 31) >> 5;
    const int64_t row0 = cu ? (int64_t)cu[n] : n * L;
    if (cu) L = cu[n + 1] - cu[n];
    if ((int)blockIdx.x * NW * 32 * QB >= L) return;                     // the whole work-group lies past the sequence
    const int q0 = ((int)blockIdx.x * NW + wave) * 32 * QB;
    const bool active = q0 < L;
    const uint8_t* mk = cu ? nullptr : mask + n * L;
    // key tiles up to the last attended key (padding behind it is all masked: nothing to add, nothing to stage)
    int nt = (L + 31) >> 5;
    if (mk) {
        int last = -1;
        for (int k0 = lane; k0 < L; k0 += 64) last = mk[k0] ? k0 : last;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(last, o, 64); last = y > last ? y : last; }
        nt = (last + 32) >> 5;                                           // last = -1: no tile
    }
    // ---- staging: piece c of a tile -> (source address, LDS offset); tile jt adds its own strides
    u32x4 stg[CPT];
    auto load_tile = [&](int jt) {
#pragma unroll
        for (int p = 0; p < CPT; ++p) {
            const int c = tid + NT * p;
            if (c >= kChunks) break;
            const int plane = (c >> 8) & 1, ch = c & 3;
            if (c < 512) {
                const int blk = (c >> 7) & 1, r = (c >> 2) & 31;
                const int key = 32 * jt + r;
                const int krow = key < L ? key : L - 1;
                const uint16_t* src = qk + (int64_t)plane * qk_plane + plane_index(row0 + krow, H + head * 64 + 32 * blk, rows) + 8 * ch;
                stg[p] = *reinterpret_cast<const u32x4*>(src);
            } else {
                const int d = (c >> 2) & 63;
                const uint16_t* src = vt + (int64_t)plane * vt_plane + (((((n * heads + head) * nt_layout + jt) * 64) + d) << 5) + 8 * ch;
                stg[p] = *reinterpret_cast<const u32x4*>(src);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = tiles + buf * kStage;
#pragma unroll
        for (int p = 0; p < CPT; ++p) {
            const int c = tid + NT * p;
            if (c >= kChunks) break;
            const int plane = (c >> 8) & 1, ch = c & 3;
            if (c < 512) {
                const int blk = (c >> 7) & 1, r = (c >> 2) & 31;
                *reinterpret_cast<u32x4*>(base + ((plane * 2 + blk) * 32 + r) * kRow + 16 * ch) = stg[p];
            } else {
                const int d = (c >> 2) & 63;
                *reinterpret_cast<u32x4*>(base + kKBytes + (plane * 64 + d) * kRow + 16 * ch) = stg[p];
            }
        }
    };
    u32x4 qh[QB][4], ql[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + 32 * qb + i;
        const int64_t qrow = row0 + (qi < L ? qi : L - 1);
        const uint16_t* qp = qk + plane_index(qrow, head * 64 + 32 * h, rows);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[qb][s] = reinterpret_cast<const u32x4*>(qp)[s];
            ql[qb][s] = reinterpret_cast<const u32x4*>(qp + qk_plane)[s];
        }
    }
    f32x16 o0[QB], o1[QB];
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = kNegInf; l[qb] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[qb][r] = 0.f; o1[qb][r] = 0.f; }
    }
    const float sscale = scale * (1.0f / (kF16ActScale * kF16ActScale)) * 1.44269504088896340736f;     // -> log2 domain
    constexpr float kPScale = 1024.0f;
    if (nt > 0) { load_tile(0); store_tile(0); }
    __syncthreads();
    for (int jt = 0; jt < nt; ++jt) {
        const bool more = jt + 1 < nt;
        if (more) load_tile(jt + 1);                                    // in flight under this tile's products
        const int key = 32 * jt + i;
        const int krow = key < L ? key : L - 1;
        const uint32_t valid = (uint32_t)__builtin_amdgcn_ballot_w64(h == 0 && key < L && (!mk || mk[krow] != 0));
        if (active && valid != 0u) {
            const unsigned char* kb = tiles + (jt & 1) * kStage + (h * 32 + i) * kRow;
            const unsigned char* vb = tiles + (jt & 1) * kStage + kKBytes + i * kRow + 32 * h;
            // phase 1, every query block: scores, running maximum / sum, probabilities (split in registers).  The K fragments
            // are dead before the V fragments are read (from LDS there is no latency to hide by requesting them early): the
            // kernel stays under the 256 registers of two waves per SIMD
            u32x4 ph[QB][2], pl[QB][2];
            {
                u32x4 kh[4], kl[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    kh[s] = *reinterpret_cast<const u32x4*>(kb + 16 * s);
                    kl[s] = *reinterpret_cast<const u32x4*>(kb + 2 * 32 * kRow + 16 * s);
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x16 st;
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kl[s]), __builtin_bit_cast(f16x8, qh[qb][s]), st, 0, 0, 0);
                        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kh[s]), __builtin_bit_cast(f16x8, ql[qb][s]), st, 0, 0, 0);
                        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kh[s]), __builtin_bit_cast(f16x8, qh[qb][s]), st, 0, 0, 0);
                    }
                    float tmax = kNegInf;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kbit = (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float v = ((valid >> kbit) & 1u) ? st[r] * sscale : kNegInf;
                        st[r] = v;
                        tmax = fmaxf(tmax, v);
                    }
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                    const float m_new = fmaxf(m[qb], tmax);
                    const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);
                    float psum = 0.f;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j2 = 0; j2 < 4; ++j2) {
                            const float e0 = __builtin_amdgcn_exp2f(st[8 * u + 2 * j2] - m_new);
                            const float e1 = __builtin_amdgcn_exp2f(st[8 * u + 2 * j2 + 1] - m_new);
                            psum += e0;
                            psum += e1;
                            const f16x2 hh = {(_Float16)(e0 * kPScale), (_Float16)(e1 * kPScale)};
                            const f16x2 ll = {(_Float16)(e0 * kPScale - (float)hh[0]), (_Float16)(e1 * kPScale - (float)hh[1])};
                            ph[qb][u][j2] = __builtin_bit_cast(uint32_t, hh);
                            pl[qb][u][j2] = __builtin_bit_cast(uint32_t, ll);
                        }
                    l[qb] = l[qb] * alpha + psum;
                    if (__builtin_amdgcn_ballot_w64(m_new > m[qb]) != 0ull) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int qa = (r & 3) + 8 * (r >> 2);
                            const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), qa));
                            const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), qa + 4));
                            const float a = h ? a1 : a0;
                            o0[qb][r] *= a;
                            o1[qb][r] *= a;
                        }
                    }
                    m[qb] = m_new;
                }
            }
            // phase 2, every query block: P . V (the products of a block in the order of the kernel above)
            {
                u32x4 vh[2][2], vl[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        vh[c][u] = *reinterpret_cast<const u32x4*>(vb + c * 32 * kRow + 16 * u);
                        vl[c][u] = *reinterpret_cast<const u32x4*>(vb + (64 + c * 32) * kRow + 16 * u);
                    }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pl[qb][u]), __builtin_bit_cast(f16x8, vh[0][u]), o0[qb], 0, 0, 0);
                        o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[qb][u]), __builtin_bit_cast(f16x8, vl[0][u]), o0[qb], 0, 0, 0);
                        o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[qb][u]), __builtin_bit_cast(f16x8, vh[0][u]), o0[qb], 0, 0, 0);
                        o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pl[qb][u]), __builtin_bit_cast(f16x8, vh[1][u]), o1[qb], 0, 0, 0);
                        o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[qb][u]), __builtin_bit_cast(f16x8, vl[1][u]), o1[qb], 0, 0, 0);
                        o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ph[qb][u]), __builtin_bit_cast(f16x8, vh[1][u]), o1[qb], 0, 0, 0);
                    }
            }
        }
        if (more) store_tile((jt + 1) & 1);       // the other buffer: every wave left it at the barrier that ended tile jt - 1
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lsum = l[qb] + __shfl_xor(l[qb], 32, 64);
        const float inv = lsum > 0.f ? 1.0f / (lsum * kPScale * kF16ActScale) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qa = (r & 3) + 8 * (r >> 2);
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), qa + 4));
            const float sc = h ? s1 : s0;
            const int q = q0 + 32 * qb + qa + 4 * h;
            if (q < L) {
                const int64_t trow = row0 + q;
                store_planes(ctxp + plane_index(trow, head * 64 + i, rows), c_plane, ns, o0[qb][r] * sc);
                store_planes(ctxp + plane_index(trow, head * 64 + 32 + i, rows), c_plane, ns, o1[qb][r] * sc);
            }
        }
    }
}

// exclusive prefix sum of the N document lengths (one wave; N is a batch of passages) + their total
static __global__ __launch_bounds__(64) void doclens_scan_kernel(const int64_t* __restrict__ doclens, int N,
                                                                 int64_t* __restrict__ start, int64_t* __restrict__ total) {
    const int lane = threadIdx.x;
    const int per = (N + 63) / 64;
    int64_t mine = 0;
    for (int i = lane * per; i < N && i < (lane + 1) * per; ++i) mine += doclens[i];
    int64_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    int64_t run = incl - mine;
    for (int i = lane * per; i < N && i < (lane + 1) * per; ++i) { start[i] = run; run += doclens[i]; }
    if (lane == 63) *total = incl;
}

// (N*L, dim) row-major projection output -> the reference's (dim, L, N) column-major array is the same memory:
// element (d, l, n) at d + dim*(l + L*n) = row (l + L*n), column d.  So no transpose kernel is needed.

}  // namespace clb
