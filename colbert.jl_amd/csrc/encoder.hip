// encoder.hip -- the Checkpoint encoder entry points of the C ABI: BERT forward + ColBERT projection
// (`doc`, src/modelling/checkpoint.jl:21-25) and the two encode paths with their epilogues
// (`_doc_embeddings_and_doclens` :27-52, `_query_embeddings` :54-71).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

#include "codec_kernels.hpp"
#include "encoder_kernels.hpp"

using namespace clb;

struct clb_encoder {
    int device = 0;
    int64_t vocab = 0, H = 0, layers = 0, heads = 0, I = 0, max_pos = 0, type_vocab = 0, dim = 0;
    float eps = 1e-12f;
    hipStream_t stream = nullptr;
    DevBuf weights;
    int attention_mode = 0;     // 0 = fused (register-resident up to 64 keys, online softmax beyond), 1 = register-resident
                                // for every length, 2 = the three-kernel path (comparison; always taken for head sizes != 64)
    int gemm_mode = 2;          // 0 = fp32 MFMA GEMMs, 1 = bf16x3, 2 = bf16x6 (bf16 MFMA products of split operands)
    // offsets (in floats) into the blob
    int64_t o_word = 0, o_pos = 0, o_type = 0, o_eg = 0, o_eb = 0, o_layer0 = 0, layer_stride = 0, o_lin_w = 0, o_lin_b = 0;
    // per-layer relative offsets
    int64_t r_wqkv = 0, r_bqkv = 0, r_wo = 0, r_bo = 0, r_g1 = 0, r_b1n = 0, r_w1 = 0, r_b1 = 0, r_w2 = 0, r_b2 = 0, r_g2 = 0, r_b2n = 0;
    // workspace
    DevBuf ids, mask, x, qkv, scores, ctx, hbuf, tmp, out, err, qmask, qlens, part;
    // per-stage HIP-event timing (clb_encoder_profile_*): off in timed runs
    bool prof_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending[8];
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[8] = {0};
    int64_t prof_launches[8] = {0};
};

namespace {
enum EncStage { ES_EMBED = 0, ES_QKV, ES_ATTENTION, ES_ATTN_OUT, ES_FFN_IN, ES_FFN_OUT, ES_PROJECTION, ES_EPILOGUE, ES_COUNT };
const char* kEncStageNames[ES_COUNT] = {"embed_layernorm", "linear_qkv", "attention", "linear_attn_out_ln", "linear_ffn_in_gelu",
                                        "linear_ffn_out_ln", "linear_projection", "epilogue"};
// events around one stage of the forward, on the stream the kernels are launched on
struct EncTimed {
    clb_encoder* e; int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(clb_encoder* e) {
        if (!e->prof_pool.empty()) { hipEvent_t v = e->prof_pool.back(); e->prof_pool.pop_back(); return v; }
        hipEvent_t v = nullptr;
        return hipEventCreate(&v) == hipSuccess ? v : nullptr;
    }
    EncTimed(clb_encoder* e_, int id_, hipStream_t st_) : e(e_), id(id_), st(st_) {
        if (!e->prof_on) return;
        a = get(e); b = get(e);
        if (a && b) (void)hipEventRecord(a, st);
    }
    ~EncTimed() {
        if (!e->prof_on || !a || !b) return;
        (void)hipEventRecord(b, st);
        e->prof_pending[id].push_back({a, b});
    }
};
}  // namespace

namespace {

inline int blocks_for(int64_t n, int bs = 256) { return (int)std::max<int64_t>(1, (n + bs - 1) / bs); }

int64_t expected_weights(const clb_encoder* e) {
    const int64_t H = e->H, I = e->I;
    const int64_t per_layer = 3 * H * H + 3 * H + H * H + H + 2 * H + I * H + I + H * I + H + 2 * H;
    return e->vocab * H + e->max_pos * H + e->type_vocab * H + 2 * H + e->layers * per_layer + e->dim * H + e->dim;
}

// `part`: scratch of at least 8 * M * N floats for the split-K path (may be null: no split)
void gemm(hipStream_t st, const float* A, const float* B, float* C, const float* bias, const float* R, int M, int N, int K,
          int64_t lda, int64_t ldb_n, int64_t ldb_k, int64_t ldc, int epi, float scale = 1.0f, int zo = 1, int zi = 1,
          int64_t a_so = 0, int64_t a_si = 0, int64_t b_so = 0, int64_t b_si = 0, int64_t c_so = 0, int64_t c_si = 0,
          float* part = nullptr) {
    GemmArgs g{A, B, C, bias, R, M, N, K, lda, ldb_n, ldb_k, ldc, zi, a_so, a_si, b_so, b_si, c_so, c_si, scale, epi, 1};
    // the tiled kernel needs both operands contiguous in k with 16-byte aligned rows and whole 32-deep steps
    const bool tiled = ldb_k == 1 && K % 32 == 0 && lda % 4 == 0 && ldb_n % 4 == 0 && a_so % 4 == 0 && a_si % 4 == 0 &&
                       b_so % 4 == 0 && b_si % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && M >= 1 && N >= 1;
    if (!tiled) {
        hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64, zo * zi), dim3(256), 0, st, g);
        return;
    }
    // the largest work-group tile that still gives the 256 CUs a few work-groups each; short activations (a batch of
    // queries: M ~ 1000) get 64 x 64 tiles and, for the narrow outputs (N = hidden), a deterministic split over K
    auto wgs = [&](int bm, int bn) { return (int64_t)((N + bn - 1) / bn) * ((M + bm - 1) / bm) * zo * zi; };
    if (wgs(128, 128) >= 384) {
        hipLaunchKernelGGL((gemm_f32_tiled_kernel<2, 2>), dim3((N + 127) / 128, (M + 127) / 128, zo * zi), dim3(256),
                           2 * 256 * kG2Ld * sizeof(float), st, g);
    } else if (wgs(64, 128) >= 384 && N >= 128) {
        hipLaunchKernelGGL((gemm_f32_tiled_kernel<1, 2>), dim3((N + 127) / 128, (M + 63) / 64, zo * zi), dim3(256),
                           2 * 192 * kG2Ld * sizeof(float), st, g);
    } else {
        int ks = 1;
        if (part && zo * zi == 1 && ldc == N) {
            while (ks < 8 && wgs(64, 64) * ks < 512 && K % (ks * 2 * 32) == 0 && K / (ks * 2) >= 256) ks *= 2;
        }
        if (ks > 1) {
            g.ksplit = ks; g.C = part;
            hipLaunchKernelGGL((gemm_f32_tiled_kernel<1, 1>), dim3((N + 63) / 64, (M + 63) / 64, ks), dim3(256),
                               2 * 128 * kG2Ld * sizeof(float), st, g);
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks_for((int64_t)M * N)), dim3(256), 0, st, part, ks,
                               (int64_t)M, N, C, bias, R, scale, epi);
        } else {
            hipLaunchKernelGGL((gemm_f32_tiled_kernel<1, 1>), dim3((N + 63) / 64, (M + 63) / 64, zo * zi), dim3(256),
                               2 * 128 * kG2Ld * sizeof(float), st, g);
        }
    }
}

// activations (T x K, fp32) x Linear weight (N x K, fp32) on the split-bf16 kernels (both operands are split into
// bf16 planes while they are staged).  Falls back to the fp32 MFMA GEMM for shapes the kernel does not take.
struct LnArgs { const float* gamma; const float* beta; float eps; };
// small tiles (query batches): two LDS tile buffers, one barrier per step (COLBERT_ENCODER_DOUBLE_BUFFER=0: the
// single-buffer loop, for comparison)
static const bool g_double_buffer = [] { const char* v = getenv("COLBERT_ENCODER_DOUBLE_BUFFER"); return !v || atoi(v) != 0; }();

// ln != null: the caller applies a LayerNorm to the output next; returns true when it was applied here (split-K path:
// fused into the reduction pass)
template <int NS>
bool linear_split(hipStream_t st, Gemm3Args g, float* part, const LnArgs* ln) {
    const int M = g.M, N = g.N, K = g.K;
    auto wgs = [&](int bm, int bn) { return (int64_t)((N + bn - 1) / bn) * ((M + bm - 1) / bm); };
    auto lds = [](int bm, int bn) { return (size_t)(NS * (bm + bn) * 64); };
    // 128 x 128 per 4-wave work-group, two or three work-groups per CU (measured: 8-wave 128 x 256 / 256 x 128 tiles with
    // one work-group per CU are 10 % slower -- nothing covers their barrier phases)
    if (wgs(128, 128) >= 384) {
        hipLaunchKernelGGL((gemm_bf16split_kernel<2, 2, 2, 2, NS>), dim3((N + 127) / 128, (M + 127) / 128, 1), dim3(256), lds(128, 128), st, g);
    } else if (wgs(64, 128) >= 384 && N >= 128) {
        if (g_double_buffer) {
            // 72 KB of dynamic LDS: above the 64-KB default limit of a launch
            allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16split_kernel<2, 2, 1, 2, NS, true>),
                              2 * NS * (64 + 128) * 64);
            hipLaunchKernelGGL((gemm_bf16split_kernel<2, 2, 1, 2, NS, true>), dim3((N + 127) / 128, (M + 63) / 64, 1), dim3(256), 2 * lds(64, 128), st, g);
        }
        else
            hipLaunchKernelGGL((gemm_bf16split_kernel<2, 2, 1, 2, NS>), dim3((N + 127) / 128, (M + 63) / 64, 1), dim3(256), lds(64, 128), st, g);
    } else {
        int ks = 1;
        if (part) {
            // K slices of at least 256 (192 for the few-tile outputs such as the hidden -> dim projection of a query batch)
            const int min_slice = wgs(64, 64) < 64 ? 192 : 256;
            while (ks < 8 && wgs(64, 64) * ks < 512 && K % (ks * 2 * 32) == 0 && K / (ks * 2) >= min_slice) ks *= 2;
        }
        float* C = g.C;
        if (ks > 1) { g.ksplit = ks; g.C = part; }
        if (g_double_buffer)
            hipLaunchKernelGGL((gemm_bf16split_kernel<2, 2, 1, 1, NS, true>), dim3((N + 63) / 64, (M + 63) / 64, ks), dim3(256), 2 * lds(64, 64), st, g);
        else
            hipLaunchKernelGGL((gemm_bf16split_kernel<2, 2, 1, 1, NS>), dim3((N + 63) / 64, (M + 63) / 64, ks), dim3(256), lds(64, 64), st, g);
        if (ks > 1) {
            if (ln && N <= 1024) {
                if (N <= 768)
                    hipLaunchKernelGGL(gemm_splitk_reduce_ln_kernel<3>, dim3(M), dim3(256), 0, st, part, ks, (int64_t)M, N,
                                       C, g.bias, g.R, 1.0f, g.epi, ln->gamma, ln->beta, ln->eps);
                else
                    hipLaunchKernelGGL(gemm_splitk_reduce_ln_kernel<4>, dim3(M), dim3(256), 0, st, part, ks, (int64_t)M, N,
                                       C, g.bias, g.R, 1.0f, g.epi, ln->gamma, ln->beta, ln->eps);
                return true;
            }
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks_for((int64_t)M * N)), dim3(256), 0, st, part, ks,
                               (int64_t)M, N, C, g.bias, g.R, 1.0f, g.epi);
        }
    }
    return false;
}

// Linear (+ optional LayerNorm of the output, in place)
void linear(clb_encoder* e, hipStream_t st, const float* A, const float* Wt, float* C, const float* bias, const float* R,
            int M, int N, int K, int epi, float* part, const LnArgs* ln = nullptr) {
    const bool ok = e->gemm_mode != 0 && K % 32 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)Wt % 16 == 0) && M >= 1 && N >= 1;
    if (!ok) {
        gemm(st, A, Wt, C, bias, R, M, N, K, K, K, 1, N, epi, 1.0f, 1, 1, 0, 0, 0, 0, 0, 0, part);
    } else {
        Gemm3Args g{A, Wt, C, bias, R, M, N, K, K, K, N, 1.0f, epi, 1};
        const bool done = e->gemm_mode == 1 ? linear_split<2>(st, g, part, ln) : linear_split<3>(st, g, part, ln);
        if (done) return;
    }
    if (ln) hipLaunchKernelGGL(layernorm_kernel, dim3(blocks_for(M, 4)), dim3(256), 0, st, C, (int64_t)M, N, ln->gamma, ln->beta, ln->eps);
}

// forward for N sequences of length L; ids / mask are device pointers; result in e->out ((N*L) x dim).
// sync = false: everything is only enqueued on `st` (an out-of-vocabulary id is then clamped silently).
int forward(clb_encoder* e, int64_t L, int64_t N, hipStream_t st, const int32_t* d_ids, const uint8_t* d_mask,
            bool sync = true) {
    const int64_t T = L * N, H = e->H, I = e->I, heads = e->heads, dh = H / heads;
    const float* W = e->weights.as<float>();
    CLB_TRY(e->x.ensure(sizeof(float) * T * H));
    CLB_TRY(e->qkv.ensure(sizeof(float) * T * 3 * H));
    const bool fused = dh == 64 && L <= 512 && e->attention_mode != 2;
    if (!fused) CLB_TRY(e->scores.ensure(sizeof(float) * N * heads * L * L));
    CLB_TRY(e->ctx.ensure(sizeof(float) * T * H));
    CLB_TRY(e->hbuf.ensure(sizeof(float) * T * I));
    CLB_TRY(e->tmp.ensure(sizeof(float) * T * H));
    CLB_TRY(e->out.ensure(sizeof(float) * T * e->dim));
    const bool short_batch = T <= 4096;         // split-K scratch only where it can be used (query batches)
    if (short_batch) CLB_TRY(e->part.ensure(sizeof(float) * 8 * T * H));
    float* part = short_batch ? e->part.as<float>() : nullptr;
    CLB_TRY(e->err.ensure(sizeof(int)));
    CLB_HIP(hipMemsetAsync(e->err.p, 0, sizeof(int), st));
    {
        EncTimed tm(e, ES_EMBED, st);
        hipLaunchKernelGGL(embed_layernorm_kernel, dim3(blocks_for(T, 4)), dim3(256), 0, st, d_ids, T, (int)L,
                           (int)H, (int)e->vocab, W + e->o_word, W + e->o_pos, W + e->o_type, W + e->o_eg, W + e->o_eb, e->eps,
                           e->x.as<float>(), e->err.as<int>());
    }
    float* x = e->x.as<float>();
    float* qkv = e->qkv.as<float>();
    float* sc = e->scores.as<float>();
    float* ctx = e->ctx.as<float>();
    float* hb = e->hbuf.as<float>();
    float* tmp = e->tmp.as<float>();
    const float inv_sqrt = 1.0f / std::sqrt((float)dh);
    for (int64_t l = 0; l < e->layers; ++l) {
        const float* P = W + e->o_layer0 + l * e->layer_stride;
        // q, k, v projections in one GEMM: (T x H) . (3H x H)^T
        { EncTimed tm(e, ES_QKV, st);
        linear(e, st, x, P + e->r_wqkv, qkv, P + e->r_bqkv, nullptr, (int)T, (int)(3 * H), (int)H, EPI_BIAS, nullptr); }
        EncTimed* t_att = new EncTimed(e, ES_ATTENTION, st);
        if (fused) {
            // softmax(Q K^T / sqrt(dh) + mask) V, one wave per (sequence, head, 32 queries), scores never leave registers
            const dim3 grid((unsigned)((L + 31) / 32), (unsigned)heads, (unsigned)N);
#define CLB_ATT(NT_) hipLaunchKernelGGL(attention_fused_kernel<NT_>, grid, dim3(64), 0, st, qkv, d_mask, ctx, (int)L, (int)H, inv_sqrt)
            if (L > 64 && e->attention_mode != 1)
                hipLaunchKernelGGL(attention_online_kernel, grid, dim3(64), 0, st, qkv, d_mask, ctx, (int)L, (int)H, inv_sqrt);
            else if (L <= 32) CLB_ATT(1); else if (L <= 64) CLB_ATT(2); else if (L <= 128) CLB_ATT(4); else if (L <= 192) CLB_ATT(6);
            else if (L <= 256) CLB_ATT(8); else if (L <= 320) CLB_ATT(10); else if (L <= 384) CLB_ATT(12); else CLB_ATT(16);
#undef CLB_ATT
        } else {
            // scores[n, head] = Q K^T / sqrt(dh)
            gemm(st, qkv, qkv + H, sc, nullptr, nullptr, (int)L, (int)L, (int)dh, 3 * H, 3 * H, 1, L, 0, inv_sqrt, (int)N, (int)heads,
                 L * 3 * H, dh, L * 3 * H, dh, heads * L * L, L * L);
            hipLaunchKernelGGL(masked_softmax_kernel, dim3(blocks_for(N * heads * L, 4)), dim3(256), 0, st, sc, N * heads * L, (int)L,
                               (int)heads, d_mask);
            // context[n, head] = P V   (B(k = key, n = dim) = V[key][dim]: ldb_k = 3H, ldb_n = 1)
            gemm(st, sc, qkv + 2 * H, ctx, nullptr, nullptr, (int)L, (int)dh, (int)L, L, 1, 3 * H, H, 0, 1.0f, (int)N, (int)heads,
                 heads * L * L, L * L, L * 3 * H, dh, L * H, dh);
        }
        delete t_att;
        // attention output + residual, LayerNorm
        const LnArgs ln1{P + e->r_g1, P + e->r_b1n, e->eps}, ln2{P + e->r_g2, P + e->r_b2n, e->eps};
        { EncTimed tm(e, ES_ATTN_OUT, st);
        linear(e, st, ctx, P + e->r_wo, tmp, P + e->r_bo, x, (int)T, (int)H, (int)H, EPI_BIAS | EPI_RESID, part, &ln1); }
        // feed-forward: GELU(x W1^T + b1) W2^T + b2 + residual, LayerNorm
        { EncTimed tm(e, ES_FFN_IN, st);
        linear(e, st, tmp, P + e->r_w1, hb, P + e->r_b1, nullptr, (int)T, (int)I, (int)H, EPI_BIAS | EPI_GELU, nullptr); }
        { EncTimed tm(e, ES_FFN_OUT, st);
        linear(e, st, hb, P + e->r_w2, x, P + e->r_b2, tmp, (int)T, (int)H, (int)I, EPI_BIAS | EPI_RESID, part, &ln2); }
    }
    // ColBERT projection: Layers.Dense(hidden -> dim)
    { EncTimed tm(e, ES_PROJECTION, st);
    linear(e, st, x, W + e->o_lin_w, e->out.as<float>(), W + e->o_lin_b, nullptr, (int)T, (int)e->dim, (int)H, EPI_BIAS, part); }
    CLB_HIP(hipGetLastError());
    if (!sync) return CLB_OK;
    int herr = 0;
    CLB_HIP(hipMemcpyAsync(&herr, e->err.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    if (herr) return fail(CLB_EBOUNDS, "token id outside the vocabulary (ids are 1-based, 1..%lld)", (long long)e->vocab);
    return CLB_OK;
}

int upload_inputs(clb_encoder* e, const int32_t* ids, const uint8_t* mask, int64_t L, int64_t N) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    if (L < 1 || N < 1) return fail(CLB_EARGUMENT, "empty batch");
    if (L > e->max_pos) return fail(CLB_EBOUNDS, "sequence length %lld exceeds max_position_embeddings %lld", (long long)L, (long long)e->max_pos);
    CLB_TRY(use_device(e->device));
    CLB_TRY(e->ids.ensure(sizeof(int32_t) * L * N));
    CLB_TRY(e->mask.ensure((size_t)L * N));
    CLB_HIP(hipMemcpyAsync(e->ids.p, ids, sizeof(int32_t) * L * N, hipMemcpyHostToDevice, e->stream));
    CLB_HIP(hipMemcpyAsync(e->mask.p, mask, (size_t)L * N, hipMemcpyHostToDevice, e->stream));
    return CLB_OK;
}

}  // namespace

extern "C" {

int clb_encoder_create(int device, int64_t vocab, int64_t hidden, int64_t layers, int64_t heads, int64_t intermediate,
                       int64_t max_pos, int64_t type_vocab, int64_t dim, float ln_eps, const float* weights,
                       int64_t n_weights, clb_encoder** out) {
    if (!out) return fail(CLB_EARGUMENT, "out is null");
    *out = nullptr;
    if (vocab < 1 || hidden < 1 || layers < 0 || heads < 1 || hidden % heads || intermediate < 1 || max_pos < 1 ||
        type_vocab < 1 || dim < 1)
        return fail(CLB_EARGUMENT, "invalid encoder shape");
    clb_encoder* e = new clb_encoder();
    e->device = device; e->vocab = vocab; e->H = hidden; e->layers = layers; e->heads = heads; e->I = intermediate;
    e->max_pos = max_pos; e->type_vocab = type_vocab; e->dim = dim; e->eps = ln_eps;
    if (expected_weights(e) != n_weights) {
        const long long want = (long long)expected_weights(e);
        delete e;
        return fail(CLB_EDIMENSION, "weight blob has %lld floats, this architecture needs %lld", (long long)n_weights, want);
    }
    const int64_t H = hidden, I = intermediate;
    int64_t o = 0;
    e->o_word = o; o += vocab * H;
    e->o_pos = o; o += max_pos * H;
    e->o_type = o; o += type_vocab * H;
    e->o_eg = o; o += H;
    e->o_eb = o; o += H;
    e->o_layer0 = o;
    int64_t r = 0;
    e->r_wqkv = r; r += 3 * H * H;
    e->r_bqkv = r; r += 3 * H;
    e->r_wo = r; r += H * H;
    e->r_bo = r; r += H;
    e->r_g1 = r; r += H;
    e->r_b1n = r; r += H;
    e->r_w1 = r; r += I * H;
    e->r_b1 = r; r += I;
    e->r_w2 = r; r += H * I;
    e->r_b2 = r; r += H;
    e->r_g2 = r; r += H;
    e->r_b2n = r; r += H;
    e->layer_stride = r;
    o += layers * r;
    e->o_lin_w = o; o += dim * H;
    e->o_lin_b = o; o += dim;
    int rc = use_device(device);
    if (rc) { delete e; return rc; }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail(CLB_EHIP, "hipStreamCreate failed"); }
    if ((rc = upload(e->weights, weights, sizeof(float) * n_weights, e->stream)) || hipStreamSynchronize(e->stream) != hipSuccess) {
        clb_encoder_destroy(e);
        return rc ? rc : fail(CLB_EHIP, "weight upload failed");
    }
    *out = e;
    return CLB_OK;
}

int clb_encoder_destroy(clb_encoder* e) {
    if (!e) return CLB_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    for (auto& v : e->prof_pending) for (auto& ab : v) { (void)hipEventDestroy(ab.first); (void)hipEventDestroy(ab.second); }
    for (auto ev : e->prof_pool) (void)hipEventDestroy(ev);
    delete e;
    return CLB_OK;
}

int clb_encoder_set_gemm_mode(clb_encoder* e, int mode) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    if (mode < 0 || mode > 2) return fail(CLB_EARGUMENT, "gemm mode %d: 0 = fp32 MFMA, 1 = bf16x3, 2 = bf16x6", mode);
    e->gemm_mode = mode;
    return CLB_OK;
}

int clb_encoder_set_attention_mode(clb_encoder* e, int mode) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    if (mode < 0 || mode > 2) return fail(CLB_EARGUMENT, "attention mode %d: 0 = fused, 1 = register-resident, 2 = three kernels", mode);
    e->attention_mode = mode;
    return CLB_OK;
}

int clb_encode(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N, float* out) {
    CLB_TRY(upload_inputs(e, integer_ids, bitmask, L, N));
    CLB_TRY(forward(e, L, N, e->stream, e->ids.as<int32_t>(), e->mask.as<uint8_t>()));
    CLB_HIP(hipMemcpy(out, e->out.p, sizeof(float) * L * N * e->dim, hipMemcpyDeviceToHost));
    return CLB_OK;
}

int clb_encode_docs(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N,
                    const int64_t* skiplist, int64_t n_skip, float* out_embs, int64_t* doclens, int64_t* n_out) {
    CLB_TRY(upload_inputs(e, integer_ids, bitmask, L, N));
    CLB_TRY(forward(e, L, N, e->stream, e->ids.as<int32_t>(), e->mask.as<uint8_t>()));
    hipStream_t st = e->stream;
    DevBuf dSkip, dMask, dLens, dStart, dOut;
    CLB_TRY(upload(dSkip, skiplist, sizeof(int64_t) * std::max<int64_t>(n_skip, 1), st));
    CLB_TRY(dMask.alloc((size_t)L * N));
    CLB_TRY(dLens.alloc(sizeof(int64_t) * N));
    hipLaunchKernelGGL(epilogue_mask_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, st, e->ids.as<int32_t>(), (int)L, (int)N,
                       dSkip.as<int64_t>(), (int)n_skip, dMask.as<uint8_t>(), dLens.as<int64_t>());
    CLB_HIP(hipMemcpyAsync(doclens, dLens.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> start((size_t)N);
    int64_t run = 0;
    for (int64_t i = 0; i < N; ++i) { start[i] = run; run += doclens[i]; }
    *n_out = run;
    if (run == 0) return CLB_OK;
    CLB_TRY(upload(dStart, start.data(), sizeof(int64_t) * N, st));
    CLB_TRY(dOut.alloc(sizeof(float) * e->dim * run));
    hipLaunchKernelGGL(epilogue_normalize_kernel, dim3(blocks_for(L * N, 64)), dim3(64), 0, st, e->out.as<float>(), (int)e->dim,
                       (int)L, (int)N, dMask.as<uint8_t>(), dStart.as<int64_t>(), dOut.as<float>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(out_embs, dOut.p, sizeof(float) * e->dim * run, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    return CLB_OK;
}

int clb_encode_queries(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N,
                       const int64_t* skiplist, int64_t n_skip, float* out) {
    CLB_TRY(upload_inputs(e, integer_ids, bitmask, L, N));
    CLB_TRY(forward(e, L, N, e->stream, e->ids.as<int32_t>(), e->mask.as<uint8_t>()));
    hipStream_t st = e->stream;
    DevBuf dSkip, dMask, dLens, dOut;
    CLB_TRY(upload(dSkip, skiplist, sizeof(int64_t) * std::max<int64_t>(n_skip, 1), st));
    CLB_TRY(dMask.alloc((size_t)L * N));
    CLB_TRY(dLens.alloc(sizeof(int64_t) * N));
    CLB_TRY(dOut.alloc(sizeof(float) * e->dim * L * N));
    hipLaunchKernelGGL(epilogue_mask_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, st, e->ids.as<int32_t>(), (int)L, (int)N,
                       dSkip.as<int64_t>(), (int)n_skip, dMask.as<uint8_t>(), dLens.as<int64_t>());
    hipLaunchKernelGGL(epilogue_normalize_kernel, dim3(blocks_for(L * N, 64)), dim3(64), 0, st, e->out.as<float>(), (int)e->dim,
                       (int)L, (int)N, dMask.as<uint8_t>(), (const int64_t*)nullptr, dOut.as<float>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(out, dOut.p, sizeof(float) * e->dim * L * N, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    return CLB_OK;
}

int clb_encode_queries_device(clb_encoder* e, const int32_t* d_integer_ids, const uint8_t* d_bitmask, int64_t L, int64_t N,
                              const int64_t* d_skiplist, int64_t n_skip, float* d_out, void* hip_stream) {
    if (!e || !d_integer_ids || !d_bitmask || !d_out) return fail(CLB_EARGUMENT, "null argument");
    if (L < 1 || N < 1) return fail(CLB_EARGUMENT, "empty batch");
    if (L > e->max_pos) return fail(CLB_EBOUNDS, "sequence length %lld exceeds max_position_embeddings %lld", (long long)L, (long long)e->max_pos);
    CLB_TRY(use_device(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    CLB_TRY(e->qmask.ensure((size_t)L * N));
    CLB_TRY(e->qlens.ensure(sizeof(int64_t) * N));
    CLB_TRY(forward(e, L, N, st, d_integer_ids, d_bitmask, /*sync=*/false));
    {
        EncTimed tm(e, ES_EPILOGUE, st);
        if (e->dim % 4 == 0)
            hipLaunchKernelGGL(epilogue_query_fused_kernel, dim3(blocks_for(L * N * 4, 256)), dim3(256), 0, st, e->out.as<float>(),
                               (int)e->dim, (int64_t)(L * N), d_integer_ids, d_skiplist, (int)n_skip, d_out);
        else {
            hipLaunchKernelGGL(epilogue_mask_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, st, d_integer_ids, (int)L, (int)N,
                               d_skiplist, (int)n_skip, e->qmask.as<uint8_t>(), e->qlens.as<int64_t>());
            hipLaunchKernelGGL(epilogue_normalize_kernel, dim3(blocks_for(L * N, 64)), dim3(64), 0, st, e->out.as<float>(), (int)e->dim,
                               (int)L, (int)N, e->qmask.as<uint8_t>(), (const int64_t*)nullptr, d_out);
        }
    }
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

int clb_encoder_check_last_ids(clb_encoder* e) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    if (!e->err.p) return CLB_OK;                     // nothing has been encoded yet
    CLB_TRY(use_device(e->device));
    int herr = 0;
    CLB_HIP(hipDeviceSynchronize());                  // the asynchronous encode may run on any of the caller's streams
    CLB_HIP(hipMemcpy(&herr, e->err.p, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) return fail(CLB_EBOUNDS, "token id outside the vocabulary (ids are 1-based, 1..%lld)", (long long)e->vocab);
    return CLB_OK;
}

int clb_encoder_profile_enable(clb_encoder* e, int on) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    e->prof_on = on != 0;
    return CLB_OK;
}

int clb_encoder_profile_read(clb_encoder* e, const char** names, double* total_ms, int64_t* launches, int cap) {
    if (!e || !names || !total_ms || !launches) return -1;
    if (use_device(e->device) != CLB_OK) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    int n = 0;
    for (int id = 0; id < ES_COUNT; ++id) {
        for (auto& ab : e->prof_pending[id]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ab.first, ab.second) == hipSuccess) { e->prof_ms[id] += ms; e->prof_launches[id] += 1; }
            e->prof_pool.push_back(ab.first); e->prof_pool.push_back(ab.second);
        }
        e->prof_pending[id].clear();
        if (n < cap && e->prof_launches[id]) {
            names[n] = kEncStageNames[id]; total_ms[n] = e->prof_ms[id]; launches[n] = e->prof_launches[id];
            ++n;
        }
        e->prof_ms[id] = 0; e->prof_launches[id] = 0;
    }
    return n;
}

}  // extern "C"
