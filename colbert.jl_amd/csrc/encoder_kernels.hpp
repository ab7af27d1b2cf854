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
// contiguous in k, K % 32 == 0, 16-byte aligned rows).  On gfx950 a SIMD does not overlap an MFMA with other
// instructions of its waves (tools/microbench/mfma_overlap.hip: MFMA time and VALU/LDS time add), so the loop is built
// to issue as little as possible besides v_mfma_f32_32x32x2_f32: a wave owns WM x WN accumulator tiles of 32 x 32 and
// feeds 4 MFMAs of a tile from ONE ds_read_b128 per operand (lane half h owns k in [16h, 16h+16) of the 32-deep
// step, i.e. the k index of the MFMA is a permutation of the tile's k, applied to A and B alike); the next step's
// global loads are issued before the MFMAs and written to the other LDS buffer after them (one barrier per step).
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

// (N*L, dim) row-major projection output -> the reference's (dim, L, N) column-major array is the same memory:
// element (d, l, n) at d + dim*(l + L*n) = row (l + L*n), column d.  So no transpose kernel is needed.

}  // namespace clb
