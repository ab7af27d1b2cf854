// encoder.hip -- the Checkpoint encoder entry points of the C ABI: BERT forward + ColBERT projection
// (`doc`, src/modelling/checkpoint.jl:21-25) and the two encode paths with their epilogues
// (`_doc_embeddings_and_doclens` :27-52, `_query_embeddings` :54-71).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <vector>

#include "codec_kernels.hpp"
#include "sort.hpp"
#include "encoder_kernels.hpp"

using namespace clb;

#ifdef CLB_ABLATIONS
namespace clb { bool launch_planes2_wide(hipStream_t st, int ln_mode, const GemmPArgs& g, unsigned grid); }   // encoder_big.hip (tuning builds)
#else
namespace clb { inline bool launch_planes2_wide(hipStream_t, int, const GemmPArgs&, unsigned) { return false; } }
#endif

struct clb_encoder {
    int device = 0;
    int64_t vocab = 0, H = 0, layers = 0, heads = 0, I = 0, max_pos = 0, type_vocab = 0, dim = 0;
    float eps = 1e-12f;
    hipStream_t stream = nullptr;
    DevBuf weights;
    int attention_mode = 0;     // 0 = fused: the fp16-plane kernel behind the f16x3 Linear layers (attention_f16_kernel), else
                                // fp32 MFMA (register-resident up to 64 keys, online softmax beyond); 1 = fp32 register-resident
                                // for every length, 2 = the three-kernel path (comparison; always taken for head sizes != 64),
                                // 3 = fused on the fp32 MFMA whatever the GEMM mode (comparison); 5 = as 0 with the K / V tiles of a
                                // (sequence, head) staged once in LDS for all its query blocks (round 5: bit-identical to 0 and
                                // measured SLOWER -- 1.42 -> 1.92 ms per 64 x 300 batch --, kept for comparison)
    int gemm_mode = 3;          // 0 = fp32 MFMA GEMMs, 1 = bf16x3, 2 = bf16x6, 3 = f16x3 (MFMA products of split operands)
    // offsets (in floats) into the blob
    int64_t o_word = 0, o_pos = 0, o_type = 0, o_eg = 0, o_eb = 0, o_layer0 = 0, layer_stride = 0, o_lin_w = 0, o_lin_b = 0;
    // per-layer relative offsets
    int64_t r_wqkv = 0, r_bqkv = 0, r_wo = 0, r_bo = 0, r_g1 = 0, r_b1n = 0, r_w1 = 0, r_b1 = 0, r_w2 = 0, r_b2 = 0, r_g2 = 0, r_b2n = 0;
    // Linear weights as bf16 planes (split once at create: plane q of blob element o_layer0 + j at wplanes + q * wp_plane + j)
    DevBuf wplanes;
    int64_t wp_plane = 0;
    int wp_fmt = 0;             // PF_* format the weight planes currently hold (re-split when the GEMM mode changes)
    std::vector<float> wscale;  // PF_F16X2: the power of two every weight matrix was multiplied by (4 per layer + projection)
    // "LayerNorm without a pass of its own" (gemm_planes2_kernel<LN>): gamma (.) W planes of the Linears that consume a LayerNorm
    // (Q/K/V of layers >= 1 with the previous layer's second LayerNorm, FFN-in with the layer's first, the projection with the
    // last), their scales, and the vectors u[n] = sum_k gamma_k W[n][k], c[n] = sum_k beta_k W[n][k] + b[n] (PF_F16X2 only)
    DevBuf wplanes_f, lnvec, stats1, stats2;
    int64_t wpf_plane = 0;
    std::vector<int64_t> f_qkv, f_w1;   // element offsets of the folded matrices inside a plane of wplanes_f (per layer)
    int64_t f_lin = 0;
    std::vector<int64_t> v_qkv, v_w1;   // float offsets of (u | c) inside lnvec (per layer): u at off, c at off + N
    int64_t v_lin = 0;
    std::vector<float> wscale_f;        // 2 per layer (Q/K/V, FFN-in) + projection
    bool fold_ready = false;
    int ln_fold = -1;                   // -1 (default): whenever the batch is long enough to run without split-K (14.34 -> 13.93 ms per
                                        // 64 x 300 passage batch, profiles/r05_experiments.md); 0: never; 1: always (tests)
    bool planes = false;        // the Linear layers read pre-split bf16 planes (gemm_planes_kernel); COLBERT_ENCODER_PLANES=0: off
    // workspace
    DevBuf ids, mask, x, qkv, scores, ctx, hbuf, tmp, out, err, qmask, qlens, part, pkeep, prank, scan_tmp;
    DevBuf xp, ctxp, tmpp, hbp; // bf16 planes of the activations the Linear layers read (written by their producers)
    DevBuf qkp, vtp;            // attention_f16_kernel's operands: Q | K planes (T x 2H, K-blocked) and V key-blocked (vt_index)
    int64_t vt_L = 0, vt_N = 0; // the shape vtp was last cleared for (key slots past L are never written: they must stay finite)
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
static const bool g_double_buffer = [] { const char* v = CLB_ENV("COLBERT_ENCODER_DOUBLE_BUFFER"); return !v || atoi(v) != 0; }();

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

// ---- Linear layers on pre-split bf16 planes (gemm_planes_kernel) ----------------------------------------------------
struct PlanCfg { int bm, bn, stages, ks; };
struct AttOut { uint16_t* qk; int64_t qk_plane; uint16_t* vt; int64_t vt_plane; int L, H, heads; const int32_t* seq; const int32_t* pos; };   // EPI_QKV_ATT targets
// a packed batch: N sequences back to back without padding rows (device arrays: position and sequence of every row, row offsets)
// `pos`: index of the position embedding; `rank` (null = pos): the row's rank inside its sequence, which is what the attention
// kernel and the key-blocked V buffer go by -- they differ when a mask has holes (host path), not for prefix masks
struct Packed { const int32_t* pos; const int32_t* seq; const int32_t* cu; int64_t rows; const int32_t* rank = nullptr; };
enum LinRole { LR_QKV = 0, LR_ATTN_OUT, LR_FFN_IN, LR_FFN_OUT, LR_PROJ, LR_COUNT };
// COLBERT_ENC_PLAN="qkv=64x128x3x1,attn_out=64x64x3x4,...": tile / ring depth / K split per Linear role (tuning runs)
static const PlanCfg* plan_override(int role) {
    static PlanCfg cfg[LR_COUNT];
    static bool have[LR_COUNT] = {false, false, false, false, false};
    static const bool parsed = [] {
        const char* v = CLB_ENV("COLBERT_ENC_PLAN");
        if (!v) return true;
        static const char* names[LR_COUNT] = {"qkv", "attn_out", "ffn_in", "ffn_out", "proj"};
        std::string sv(v);
        size_t pos = 0;
        while (pos < sv.size()) {
            size_t end = sv.find(',', pos);
            if (end == std::string::npos) end = sv.size();
            const std::string item = sv.substr(pos, end - pos);
            const size_t eq = item.find('=');
            if (eq != std::string::npos)
                for (int r = 0; r < LR_COUNT; ++r)
                    if (item.substr(0, eq) == names[r]) {
                        PlanCfg c{0, 0, 0, 0};
                        if (sscanf(item.c_str() + eq + 1, "%dx%dx%dx%d", &c.bm, &c.bn, &c.stages, &c.ks) == 4) { cfg[r] = c; have[r] = true; }
                    }
            pos = end + 1;
        }
        return true;
    }();
    (void)parsed;
    return have[role] ? &cfg[role] : nullptr;
}

// COLBERT_ENC_GEMM_FORM=1: the first form of the plane GEMM (element-wise epilogue, LDS reads not pipelined) -- comparison runs
static bool planes_first_form() {
    static const bool v = [] { const char* e = CLB_ENV("COLBERT_ENC_GEMM_FORM"); return e && atoi(e) == 1; }();
    return kAblations && v;
}

// COLBERT_ENC_ATT_QB=1: one query block per wave in attention_f16_kernel whatever the length -- comparison runs
static bool att_qb2() {
    static const bool v = [] { const char* e = CLB_ENV("COLBERT_ENC_ATT_QB"); return !(e && atoi(e) == 1); }();
    return v;
}

// what a Linear needs to fold a LayerNorm around itself (GemmPArgs' ln_* fields)
struct LnFold {
    const float* ln_in = nullptr; int parts = 0, width = 0; float eps = 0.f;     // statistics of A's rows (u set) or of R's rows (r_gamma set)
    const float* u = nullptr;                                                     // fold: `bias` of the call then is c
    const float* r_gamma = nullptr; const float* r_beta = nullptr;
    float* stats_out = nullptr;
};

// the LN instantiations: 1 = consumer (64 x 64 for the projection's 128 columns, 128 x 128 behind GELU, 128 x 256 in front of the
// attention), 2 = producer (the big plain tiles)
// COLBERT_ENC_WIDE_WAVES=1: the 256 x 256 tile as four waves of 128 x 128 (encoder_big.hip: a third fewer LDS reads per MFMA,
// accumulators in AGPRs) instead of eight of 64 x 128 -- bit-identical and SLOWER (a 64 x 300 batch 16.6 against 13.9 ms: with
// one wave per SIMD nothing covers the barrier and the DMA wait of every step); kept for comparison runs (tools/r5_wide_waves.py)
static bool wide_waves() {
    static const bool v = [] { const char* e = CLB_ENV("COLBERT_ENC_WIDE_WAVES"); return e && atoi(e) == 1; }();
    return kAblations && v;
}

bool launch_planes_ln(hipStream_t st, const PlanCfg& c, const GemmPArgs& g) {
    const dim3 grid((unsigned)gemm_planes_grid(g.M, g.N, c.bm, c.bn, c.ks));
    const size_t lds = (size_t)c.stages * 2 * (c.bm + c.bn) * 64;
    const int mode = g.ln_u ? 1 : 2;
    if (c.bm == 256 && c.bn == 256 && c.stages == 2 && wide_waves()) return launch_planes2_wide(st, mode, g, grid.x);
#define CLB_GPL_CASE(MODE_, BM_, BN_, ST_, WGM_, WGN_, WM_, WN_)                                                      \
    if (mode == MODE_ && c.bm == BM_ && c.bn == BN_ && c.stages == ST_) {                                             \
        auto kern = gemm_planes2_kernel<WGM_, WGN_, WM_, WN_, 2, ST_, 0, true, MODE_>;                                \
        if (lds > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);                        \
        hipLaunchKernelGGL(kern, grid, dim3(64 * WGM_ * WGN_), lds, st, g);                                           \
        return true;                                                                                                  \
    }
    CLB_GPL_CASE(1, 64, 64, 2, 2, 2, 1, 1) CLB_GPL_CASE(1, 128, 128, 2, 2, 2, 2, 2) CLB_GPL_CASE(1, 128, 256, 2, 2, 4, 2, 2)
    CLB_GPL_CASE(2, 128, 128, 2, 2, 2, 2, 2) CLB_GPL_CASE(2, 128, 256, 2, 2, 4, 2, 2) CLB_GPL_CASE(2, 256, 256, 2, 4, 2, 2, 4)
#undef CLB_GPL_CASE
    return false;
}

// (the first form of the plane GEMM, gemm_planes_kernel, is a comparison kernel: tuning builds; it also takes N % 4 != 0, which
// no Linear of the encoder has -- linear_planes falls back to the fp32 GEMM for such a shape)
#ifdef CLB_ABLATIONS
#define CLB_PLANES_KERNEL(SECOND, WGM_, WGN_, WM_, WN_, NS_, ST_, F16_) \
    ((SECOND) ? gemm_planes2_kernel<WGM_, WGN_, WM_, WN_, NS_, ST_, 0, F16_> : gemm_planes_kernel<WGM_, WGN_, WM_, WN_, NS_, ST_, 0, F16_>)
#else
#define CLB_PLANES_KERNEL(SECOND, WGM_, WGN_, WM_, WN_, NS_, ST_, F16_) (gemm_planes2_kernel<WGM_, WGN_, WM_, WN_, NS_, ST_, 0, F16_>)
#endif
template <int NS, bool F16>
bool launch_planes(hipStream_t st, const PlanCfg& c, const GemmPArgs& g) {
    const dim3 grid((unsigned)gemm_planes_grid(g.M, g.N, c.bm, c.bn, c.ks));
    const size_t lds = (size_t)c.stages * NS * (c.bm + c.bn) * 64;
    const bool second = g.N % 4 == 0 && !planes_first_form();
    if (NS == 2 && F16 && second && c.bm == 256 && c.bn == 256 && c.stages == 2 && wide_waves()) return launch_planes2_wide(st, 0, g, grid.x);
#define CLB_GP_CASE(BM_, BN_, ST_, WGM_, WGN_, WM_, WN_)                                                              \
    if (c.bm == BM_ && c.bn == BN_ && c.stages == ST_) {                                                              \
        auto kern = CLB_PLANES_KERNEL(second, WGM_, WGN_, WM_, WN_, NS, ST_, F16);                                    \
        if (lds > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);                        \
        hipLaunchKernelGGL(kern, grid, dim3(64 * WGM_ * WGN_), lds, st, g);                                           \
        return true;                                                                                                  \
    }
    CLB_GP_CASE(64, 64, 2, 2, 2, 1, 1) CLB_GP_CASE(64, 64, 3, 2, 2, 1, 1) CLB_GP_CASE(64, 64, 4, 2, 2, 1, 1)
    CLB_GP_CASE(64, 128, 2, 2, 2, 1, 2) CLB_GP_CASE(64, 128, 3, 2, 2, 1, 2)
    CLB_GP_CASE(128, 64, 2, 2, 2, 2, 1) CLB_GP_CASE(128, 64, 3, 2, 2, 2, 1)
    CLB_GP_CASE(128, 128, 2, 2, 2, 2, 2) CLB_GP_CASE(128, 128, 3, 2, 2, 2, 2)
    if (NS == 2) { CLB_GP_CASE(128, 256, 2, 2, 4, 2, 2) CLB_GP_CASE(128, 256, 3, 2, 4, 2, 2) CLB_GP_CASE(256, 256, 2, 4, 2, 2, 4) }   // eight waves, two planes (96 / 144 / 128 KB)
#undef CLB_GP_CASE
    return false;
}

// The tile of a long-activation Linear (passage batches), among 256 x 256 (eight waves, one work-group per CU), 128 x 256 (four
// waves, one per CU) and 128 x 128 (two work-groups per CU): the one with the smallest  rounds x work-groups-per-CU x tile area x
// cost-per-flop  -- ROUNDS matter as much as the tile: a 22 386-row packed batch gives a 768-wide output 264 tiles of 256 x 256,
// one more than eight per XCD... 1.03 rounds that take two (measured: FFN-out 5.95 ms per batch against 3.45 at 19 200 rows, where
// 225 tiles fit one round).  Cost per flop relative to 256 x 256 from the round-4 sweeps (half / a quarter of the operand reuse).
static PlanCfg pick_long_tile(int M, int N, bool allow_256, bool allow_128x256) {
    struct Cand { int bm, bn, per_cu; double cost; bool ok; };
    const Cand cands[3] = {{256, 256, 1, 1.00, allow_256}, {128, 256, 1, 1.04, allow_128x256}, {128, 128, 2, 1.09, true}};
    PlanCfg best{128, 128, 2, 1};
    double best_t = 1e300;
    for (const Cand& c : cands) {
        if (!c.ok) continue;
        const int64_t tiles = (int64_t)((N + c.bn - 1) / c.bn) * ((M + c.bm - 1) / c.bm);
        const int64_t rounds = (tiles + 256 * c.per_cu - 1) / (256 * c.per_cu);
        const double t = (double)rounds * c.per_cu * c.bm * c.bn * c.cost;
        if (t < best_t) { best_t = t; best = PlanCfg{c.bm, c.bn, 2, 1}; }
    }
    return best;
}

// A planes (M x K) . W planes (N x K)^T -> C fp32 and / or Cp planes, epilogue bias / GELU / residual, optional LayerNorm
// of the output (then both C and, if given, Cp hold the normalised rows).  part: split-K scratch (8 * M * N floats) or null.
inline int plane_format(int gemm_mode) { return gemm_mode == 3 ? PF_F16X2 : gemm_mode == 1 ? PF_BF16X2 : PF_BF16X3; }

// wscale: the power of two the weight planes were scaled by (PF_F16X2; 1 otherwise)
void linear_planes(clb_encoder* e, hipStream_t st, int role, const uint16_t* Ap, int64_t a_plane, const uint16_t* Wp, float wscale,
                   float* C, uint16_t* Cp, int64_t c_plane, const float* bias, const float* R, int M, int N, int K, int epi,
                   float* part, const LnArgs* ln, const AttOut* att = nullptr, const LnFold* lf = nullptr, int64_t b_plane = 0) {
    const int fmt = plane_format(e->gemm_mode);
    const int NS = fmt == PF_BF16X3 ? 3 : 2;
    const float out_scale = fmt == PF_F16X2 ? 1.0f / (kF16ActScale * wscale) : 1.0f;
    auto wgs = [&](int bm, int bn) { return (int64_t)((N + bn - 1) / bn) * ((M + bm - 1) / bm); };
    PlanCfg c;
    if (lf) {
        // a Linear with a LayerNorm folded around it: never split over K (the statistics are taken from finished rows), the
        // big-tile rule of long activations below, 128 x 128 at least when it produces statistics (a part = two 32-wide tiles
        // of one wave); the narrow projection (N = dim) only consumes: 64 x 64
        const bool wide = N % 4 == 0 && !(epi & EPI_GELU);
        c = N <= 128 && !lf->stats_out ? PlanCfg{64, 64, 2, 1}
            : wgs(128, 128) >= 256 ? pick_long_tile(M, N, wide && !att && !lf->u, wide) : PlanCfg{128, 128, 2, 1};
        part = nullptr;
    }
    else if (const PlanCfg* o = plan_override(role)) c = *o;
    else if (M <= 64 && part) {
        // ONE query (search(searcher, query::String, k), 32 rows): every Linear is a weight stream with one tile row.  Split
        // over K until the chip is full (at least three 32-deep steps per slice) with a four-buffer ring: a work-group's
        // whole slice is in flight from its prologue, i.e. ONE memory round trip per work-group instead of one per step
        // (24 dependent round trips made each of these launches 24 us)
        c = {64, 64, 4, 1};
        const int tn = (N + 63) / 64;
        while (c.ks < 32 && K % (c.ks * 2 * 32) == 0 && K / (c.ks * 2) >= 96 && tn * c.ks * 2 <= 1024) c.ks *= 2;
    }
    else if (wgs(128, 128) >= 256)      // one 128 x 128 tile per CU and more (sweeps at M = 2 560 ... 19 200: r04_encoder_plan_sweeps.txt)
        // long activations (passage batches; tools/microbench/gemm_planes_bigm_bench.hip): enough tiles to fill the chip, the
        // widest tile that still leaves two waves per SIMD -- 256 x 256 on eight waves of 64 x 128 (two planes: 128 KB of LDS;
        // half the operand bytes of 128 x 128 through L2; not behind the Q/K/V projection, whose epilogue scatters V), 128 x 256, else 128 x 128.
        // A GELU epilogue is ~2/3 of a tile's main loop in vector instructions: two 128 x 128 work-groups per CU overlap one's
        // epilogue with the other's loop, a single large one cannot (FFN-in of 64 x 300 passages: 322 against 358 us)
        {
            const bool wide = NS == 2 && N % 4 == 0 && !(epi & EPI_GELU) && !planes_first_form();
            c = pick_long_tile(M, N, wide && !att, wide);
            // round 6 (profiles/r06_encoder_plan_n128.txt, 128 queries = 4 096 rows): the Q/K/V projection of a few thousand rows
            // is 576 tiles of 128 x 128 -- two work-groups per CU take them in 1.1 rounds whose tail costs half a round, where
            // the 288 tiles of 128 x 256 the rule above picks take two full ones (0.85 against 1.02 ms over the 12 layers)
            if (att && wgs(128, 128) < 1024) c = PlanCfg{128, 128, 2, 1};
        }
    else if (part && wgs(128, 128) >= 128 && K % 64 == 0 && K / 2 >= 384)
        // ... and its hidden-wide outputs (attention output, FFN-out: 192 tiles of 128 x 128 at 4 096 rows) fill the chip as
        // 128 x 128 tiles over TWO K slices, the reduction fused into the LayerNorm pass that follows
        // (gemm_splitk_reduce_ln4_kernel): 0.47 / 0.99 ms against 0.58 / 1.26 with the 64 x 64 tiles of the query-batch rule
        c = PlanCfg{128, 128, 2, 2};
    else {
        // A query batch (M ~ 1 000): 64 x 64 tiles with a two-tile ring = 48 KB of LDS, three work-groups per CU.  Measured
        // (tools/microbench/gemm_planes_bench.hip): the loop is bound by MFMA issue (three 32 x 32 tiles per SIMD at the
        // ~1.75 GHz the chip holds) and by the ~65 GB/s an XCD's L2 delivers to one CU, both ~20 us for the 1024 x 2304 x 768
        // product; larger tiles move fewer bytes but leave CUs without a work-group, deeper rings cost the third resident
        // work-group.  Narrow outputs (N = hidden) are split over K until every CU has three work-groups.
        c = {64, 64, 2, 1};
        if (part) {
            // K slices of at least 12 steps (6 for the few-tile hidden -> dim projection), and a three-tile ring for the short
            // loops of a split product (in-encoder sweep, f16x3: attention output 64x64x3 ks 2: 18.3 us against 20.2 at x2 ks 4;
            // FFN-out x3 ks 4: 32.8 against 33.5; ks 2 or 8 there: 45 / 42)
            const int min_slice = wgs(64, 64) < 64 ? 192 : 384;
            while (c.ks < 8 && wgs(64, 64) * c.ks < 768 && K % (c.ks * 2 * 32) == 0 && K / (c.ks * 2) >= min_slice) c.ks *= 2;
            if (c.ks > 1) c.stages = 3;
        }
    }
    if (!part || K % (c.ks * 32) != 0) c.ks = 1;
    GemmPArgs g{Ap, Wp, a_plane, b_plane ? b_plane : e->wp_plane, C, bias, R, Cp, c_plane, M, N, K, N, epi, c.ks, out_scale};
    if (lf) {
        g.ln_in = lf->ln_in; g.ln_parts = lf->parts; g.ln_width = lf->width; g.ln_eps = lf->eps;
        g.ln_u = lf->u; g.r_gamma = lf->r_gamma; g.r_beta = lf->r_beta; g.stats_out = lf->stats_out;
    }
    if (att) {      // Q | K planes and key-blocked V instead of an fp32 matrix (split over K: written by the reduce pass)
        g.C = nullptr; g.Cp = att->qk; g.c_plane = att->qk_plane; g.epi |= EPI_QKV_ATT;
        g.Vt = att->vt; g.vt_plane = att->vt_plane; g.att_L = att->L; g.att_H = att->H; g.att_heads = att->heads;
        g.att_seq = att->seq; g.att_pos = att->pos;
    }
    // an output that is normalised next keeps its planes for the LayerNorm kernel to write
    if (ln) g.Cp = nullptr;
    if (c.ks > 1) { g.C = part; g.Cp = nullptr; }
    auto go = [&](const PlanCfg& cc) {
        if (lf) return launch_planes_ln(st, cc, g);           // PF_F16X2 only (forward() folds in no other mode)
        return fmt == PF_F16X2 ? launch_planes<2, true>(st, cc, g) : NS == 2 ? launch_planes<2, false>(st, cc, g) : launch_planes<3, false>(st, cc, g);
    };
    if (!go(c)) {
        if (lf) { (void)fail(CLB_EUNSUPPORTED, "no LayerNorm-folding GEMM for a %d x %d tile", c.bm, c.bn); return; }
        c = {64, 64, 2, c.ks}; (void)go(c);
    }
    if (c.ks > 1) {
        if (ln && N <= 1024) {
            if (N % 4 == 0)
                hipLaunchKernelGGL(gemm_splitk_reduce_ln4_kernel, dim3(M), dim3(256), 0, st, part, c.ks, (int64_t)M, N, C, bias, R,
                                   out_scale, epi, ln->gamma, ln->beta, ln->eps, Cp, c_plane, fmt);
            else if (N <= 768)
                hipLaunchKernelGGL(gemm_splitk_reduce_ln_kernel<3>, dim3(M), dim3(256), 0, st, part, c.ks, (int64_t)M, N, C, bias, R,
                                   out_scale, epi, ln->gamma, ln->beta, ln->eps, Cp, c_plane, fmt);
            else
                hipLaunchKernelGGL(gemm_splitk_reduce_ln_kernel<4>, dim3(M), dim3(256), 0, st, part, c.ks, (int64_t)M, N, C, bias, R,
                                   out_scale, epi, ln->gamma, ln->beta, ln->eps, Cp, c_plane, fmt);
            return;
        }
        if (att)
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks_for((int64_t)M * N)), dim3(256), 0, st, part, c.ks, (int64_t)M, N,
                               (float*)nullptr, bias, R, out_scale, epi, att->qk, att->qk_plane, fmt, att->vt, att->vt_plane, att->L, att->H,
                               att->heads, att->seq, att->pos);
        else
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks_for((int64_t)M * N)), dim3(256), 0, st, part, c.ks, (int64_t)M, N,
                           C, bias, R, out_scale, epi, ln ? (uint16_t*)nullptr : Cp, c_plane, fmt);
    }
    if (ln) hipLaunchKernelGGL(layernorm_kernel, dim3(blocks_for(M, 4)), dim3(256), 0, st, C, (int64_t)M, N, ln->gamma, ln->beta,
                               ln->eps, Cp, c_plane, fmt);
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

// Every Linear weight (N x K, torch layout) into K-blocked planes of format `fmt` at its own blob offset.  PF_F16X2: each
// matrix is first scaled by the power of two that brings its largest |entry| into [2^13, 2^14) (exact; the GEMM epilogue
// divides it out again) -- fp16 carries 5 exponent bits, and an unscaled N(0, 0.02) weight would leave its low plane in the
// subnormals.
int split_weights(clb_encoder* e, int fmt) {
    const int64_t H = e->H, I = e->I, n_lin = e->wp_plane;
    struct Mat { int64_t off, rows, cols; };
    std::vector<Mat> mats;
    for (int64_t l = 0; l < e->layers; ++l) {
        const int64_t lo = e->o_layer0 + l * e->layer_stride;
        mats.push_back({lo + e->r_wqkv, 3 * H, H}); mats.push_back({lo + e->r_wo, H, H});
        mats.push_back({lo + e->r_w1, I, H}); mats.push_back({lo + e->r_w2, H, I});
    }
    mats.push_back({e->o_lin_w, e->dim, H});
    e->wscale.assign(e->layers * 4 + 4, 1.0f);
    hipStream_t st = e->stream;
    if (fmt == PF_F16X2) {
        DevBuf mx;
        CLB_TRY(mx.alloc(sizeof(unsigned int) * mats.size()));
        CLB_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned int) * mats.size(), st));
        for (size_t i = 0; i < mats.size(); ++i)
            hipLaunchKernelGGL(max_abs_kernel, dim3(256), dim3(256), 0, st, e->weights.as<float>() + mats[i].off,
                               (int)(mats[i].rows * mats[i].cols), mx.as<unsigned int>() + i);
        std::vector<unsigned int> bits(mats.size());
        CLB_HIP(hipMemcpyAsync(bits.data(), mx.p, sizeof(unsigned int) * mats.size(), hipMemcpyDeviceToHost, st));
        CLB_HIP(hipStreamSynchronize(st));
        for (size_t i = 0; i < mats.size(); ++i) {
            float m;
            memcpy(&m, &bits[i], sizeof m);
            int ex = 0;
            if (m > 0.f && m <= FLT_MAX) (void)std::frexp(m, &ex);          // m = f * 2^ex, f in [0.5, 1)
            e->wscale[i] = std::ldexp(1.0f, std::max(-100, std::min(100, 14 - ex)));   // m * scale in [2^13, 2^14)
        }
    }
    for (size_t i = 0; i < mats.size(); ++i)
        hipLaunchKernelGGL(split_planes_kernel, dim3(blocks_for(mats[i].rows * mats[i].cols / 4)), dim3(256), 0, st,
                           e->weights.as<float>() + mats[i].off, e->wplanes.as<uint16_t>() + (mats[i].off - e->o_layer0), mats[i].rows,
                           (int)mats[i].cols, n_lin, fmt, e->wscale[i]);
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipStreamSynchronize(st));
    e->wp_fmt = fmt;
    // ---- the folded operands of "LayerNorm without a pass of its own" (gemm_planes2_kernel<LN>): PF_F16X2 only
    e->fold_ready = false;
    // (a row's statistics travel as H / 64 partial (mean, M2) pairs, at most kLnMaxParts of them: hidden sizes up to 1 024 --
    // a wider model keeps its LayerNorm passes)
    if (fmt == PF_F16X2 && e->layers >= 1 && H % 64 == 0 && H / 64 <= kLnMaxParts && H % 4 == 0 && e->dim % 4 == 0) {
        const int64_t L = e->layers;
        struct Fold { int64_t w_off, rows, g_off, b_off, bias_off, p_off, v_off; };
        std::vector<Fold> folds;
        e->f_qkv.assign(L, -1); e->f_w1.assign(L, -1); e->v_qkv.assign(L, -1); e->v_w1.assign(L, -1);
        int64_t po = 0, vo = 0;
        for (int64_t l = 0; l < L; ++l) {
            const int64_t lo = e->o_layer0 + l * e->layer_stride, lp = lo - e->layer_stride;
            if (l >= 1) {      // Q/K/V of layer l consumes the second LayerNorm of layer l - 1
                e->f_qkv[l] = po; e->v_qkv[l] = vo;
                folds.push_back({lo + e->r_wqkv, 3 * H, lp + e->r_g2, lp + e->r_b2n, lo + e->r_bqkv, po, vo});
                po += 3 * H * H; vo += 2 * 3 * H;
            }
            e->f_w1[l] = po; e->v_w1[l] = vo;       // FFN-in consumes the layer's first LayerNorm
            folds.push_back({lo + e->r_w1, I, lo + e->r_g1, lo + e->r_b1n, lo + e->r_b1, po, vo});
            po += I * H; vo += 2 * I;
        }
        {   // the projection consumes the last layer's second LayerNorm
            const int64_t ll = e->o_layer0 + (L - 1) * e->layer_stride;
            e->f_lin = po; e->v_lin = vo;
            folds.push_back({e->o_lin_w, e->dim, ll + e->r_g2, ll + e->r_b2n, e->o_lin_b, po, vo});
            po += e->dim * H; vo += 2 * e->dim;
        }
        if (po * 4 < ((int64_t)1 << 31)) {
            e->wpf_plane = po;
            CLB_TRY(e->wplanes_f.ensure(sizeof(uint16_t) * 2 * po));
            CLB_TRY(e->lnvec.ensure(sizeof(float) * vo));
            DevBuf mx;
            CLB_TRY(mx.alloc(sizeof(unsigned int) * folds.size()));
            CLB_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned int) * folds.size(), st));
            const float* W = e->weights.as<float>();
            for (size_t i = 0; i < folds.size(); ++i)
                hipLaunchKernelGGL(max_abs_colscaled_kernel, dim3(256), dim3(256), 0, st, W + folds[i].w_off, folds[i].rows, (int)H,
                                   W + folds[i].g_off, mx.as<unsigned int>() + i);
            std::vector<unsigned int> bits(folds.size());
            CLB_HIP(hipMemcpyAsync(bits.data(), mx.p, sizeof(unsigned int) * folds.size(), hipMemcpyDeviceToHost, st));
            CLB_HIP(hipStreamSynchronize(st));
            e->wscale_f.assign(folds.size(), 1.0f);
            for (size_t i = 0; i < folds.size(); ++i) {
                float m;
                memcpy(&m, &bits[i], sizeof m);
                int ex = 0;
                if (m > 0.f && m <= FLT_MAX) (void)std::frexp(m, &ex);
                e->wscale_f[i] = std::ldexp(1.0f, std::max(-100, std::min(100, 14 - ex)));
                hipLaunchKernelGGL(split_planes_kernel, dim3(blocks_for(folds[i].rows * H / 4)), dim3(256), 0, st, W + folds[i].w_off,
                                   e->wplanes_f.as<uint16_t>() + folds[i].p_off, folds[i].rows, (int)H, po, fmt, e->wscale_f[i],
                                   W + folds[i].g_off);
                hipLaunchKernelGGL(ln_fold_vectors_kernel, dim3(blocks_for(folds[i].rows, 4)), dim3(256), 0, st, W + folds[i].w_off,
                                   (int)folds[i].rows, (int)H, W + folds[i].g_off, W + folds[i].b_off, W + folds[i].bias_off,
                                   e->lnvec.as<float>() + folds[i].v_off);
            }
            CLB_HIP(hipGetLastError());
            CLB_HIP(hipStreamSynchronize(st));
            e->fold_ready = true;
        }
    }
    return CLB_OK;
}

// index into wscale_f (the order split_weights folds in): Q/K/V of layer l >= 1, FFN-in of layer l, the projection last
static inline size_t fold_scale_index(int64_t l, int which /* 0 = Q/K/V (l >= 1), 1 = FFN-in */) { return (size_t)(l == 0 ? 0 : 2 * l - 1 + which); }

// forward for N sequences of length L; ids / mask are device pointers; result in e->out ((N*L) x dim).
// sync = false: everything is only enqueued on `st` (an out-of-vocabulary id is then clamped silently).
// whether a batch of at most `rows_max` rows and sequences up to L can run packed (the conditions of the fp16-plane attention)
static bool can_pack(const clb_encoder* e, int64_t L, int64_t rows_max = 0) {
    const int64_t H = e->H, I = e->I;
    return e->planes && e->gemm_mode == 3 && H % 32 == 0 && I % 32 == 0 && e->heads > 0 && H / e->heads == 64 && L <= 512 &&
           (e->attention_mode == 0 || e->attention_mode == 5) && !planes_first_form() && rows_max * std::max(H, I) * 6 < ((int64_t)1 << 31);
}

// pk (packed batch): d_ids holds pk->rows token ids, L is the longest sequence, d_mask is unused; needs the fp16-plane attention.
int forward(clb_encoder* e, int64_t L, int64_t N, hipStream_t st, const int32_t* d_ids, const uint8_t* d_mask,
            bool sync = true, const Packed* pk = nullptr) {
    const int64_t T = pk ? pk->rows : L * N, H = e->H, I = e->I, heads = e->heads, dh = H / heads;
    const float* W = e->weights.as<float>();
    CLB_TRY(e->x.ensure(sizeof(float) * T * H));
    CLB_TRY(e->qkv.ensure(sizeof(float) * T * 3 * H));
    const bool fused = dh == 64 && L <= 512 && e->attention_mode != 2;
    if (!fused) CLB_TRY(e->scores.ensure(sizeof(float) * N * heads * L * L));
    CLB_TRY(e->ctx.ensure(sizeof(float) * T * H));
    CLB_TRY(e->hbuf.ensure(sizeof(float) * T * I));
    CLB_TRY(e->tmp.ensure(sizeof(float) * T * H));
    CLB_TRY(e->out.ensure(sizeof(float) * T * e->dim));
    // the Linear layers read bf16 planes: activations are split once, by their producer (three planes are always written;
    // bf16x3 reads the first two)
    const bool P = e->planes && e->gemm_mode != 0 && H % 32 == 0 && I % 32 == 0 && T * std::max(H, I) * 6 < ((int64_t)1 << 31);
    const int64_t hp = T * H, ip = T * I;      // plane strides of the activation planes
    if (P) {
        CLB_TRY(e->xp.ensure(sizeof(uint16_t) * 3 * hp));
        CLB_TRY(e->ctxp.ensure(sizeof(uint16_t) * 3 * hp));
        CLB_TRY(e->tmpp.ensure(sizeof(uint16_t) * 3 * hp));
        CLB_TRY(e->hbp.ensure(sizeof(uint16_t) * 3 * ip));
    }
    const int PF = plane_format(e->gemm_mode);
    if (P && e->wp_fmt != PF) CLB_TRY(split_weights(e, PF));
    // attention on fp16 planes: the Q/K/V projection writes them (second GEMM form, never split over K)
    const bool att16 = P && PF == PF_F16X2 && fused && (e->attention_mode == 0 || e->attention_mode == 5) && H % 4 == 0 && !planes_first_form();
    const int64_t ntile = (L + 31) / 32, qk_plane = T * 2 * H, vt_plane = N * heads * ntile * 64 * 32;
    if (pk && !att16)
        return fail(CLB_EARGUMENT, "a packed batch needs the fp16-plane attention (head size 64, f16x3 Linear layers, attention mode 0)");
    if (att16) {
        CLB_TRY(e->qkp.ensure(sizeof(uint16_t) * 2 * qk_plane));
        const void* before = e->vtp.p;
        CLB_TRY(e->vtp.ensure(sizeof(uint16_t) * 2 * vt_plane));
        // packed: the key slots between a sequence's end and Lmax are never written either -- cleared per call (~20 us)
        if (pk || e->vtp.p != before || e->vt_L != L || e->vt_N != N) {
            CLB_HIP(hipMemsetAsync(e->vtp.p, 0, sizeof(uint16_t) * 2 * vt_plane, st));
            e->vt_L = L; e->vt_N = N;
        }
    }
    const AttOut att_out{e->qkp.as<uint16_t>(), qk_plane, e->vtp.as<uint16_t>(), vt_plane, (int)L, (int)H, (int)heads, pk ? pk->seq : nullptr,
                         pk ? (pk->rank ? pk->rank : pk->pos) : nullptr};
    uint16_t* xp = e->xp.as<uint16_t>(); uint16_t* ctxp = e->ctxp.as<uint16_t>();
    uint16_t* tmpp = e->tmpp.as<uint16_t>(); uint16_t* hbp = e->hbp.as<uint16_t>();
    const uint16_t* WP = e->wplanes.as<uint16_t>();
    auto wp = [&](int64_t blob_off) { return WP + (blob_off - e->o_layer0); };
    auto ws = [&](int64_t layer, int which) { return e->wscale.empty() ? 1.0f : e->wscale[(size_t)(layer * 4 + which)]; };
    const bool short_batch = T <= 4096;         // split-K scratch only where it can be used (query batches)
    // "LayerNorm without a pass of its own": long batches (no split-K anywhere), f16x3 planes, second GEMM form
    const bool fold = P && PF == PF_F16X2 && e->fold_ready && !planes_first_form() && e->layers >= 1 &&
                      (e->ln_fold == 1 || (e->ln_fold < 0 && !short_batch)) && !plan_override(LR_ATTN_OUT) && !plan_override(LR_FFN_OUT);
    const int ln_parts = (int)(H / 64);
    if (fold) {
        CLB_TRY(e->stats1.ensure(sizeof(float) * 2 * ln_parts * T));
        CLB_TRY(e->stats2.ensure(sizeof(float) * 2 * ln_parts * T));
    }
    const uint16_t* WPF = e->wplanes_f.as<uint16_t>();
    const float* LV = e->lnvec.as<float>();
    float* st1 = e->stats1.as<float>(); float* st2 = e->stats2.as<float>();
    const bool tiny = T <= 64;                  // one query: every Linear is split over K (up to 32 slices)
    if (short_batch) CLB_TRY(e->part.ensure(sizeof(float) * (tiny ? 32 * T * std::max(3 * H, I) : 8 * T * H)));
    float* part = short_batch ? e->part.as<float>() : nullptr;
    float* part_wide = tiny ? part : nullptr;   // scratch for the wide outputs (QKV, FFN-in), which only a single query splits
    if (!e->err.p) {      // the flag is STICKY: set by any encode since the last check, cleared by whoever reads it
        CLB_TRY(e->err.ensure(sizeof(int)));
        CLB_HIP(hipMemsetAsync(e->err.p, 0, sizeof(int), st));
    }
    {
        EncTimed tm(e, ES_EMBED, st);
        hipLaunchKernelGGL(embed_layernorm_kernel, dim3(blocks_for(T, 4)), dim3(256), 0, st, d_ids, T, (int)L,
                           (int)H, (int)e->vocab, W + e->o_word, W + e->o_pos, W + e->o_type, W + e->o_eg, W + e->o_eb, e->eps,
                           e->x.as<float>(), e->err.as<int>(), P ? xp : (uint16_t*)nullptr, hp, PF, pk ? pk->pos : nullptr);
    }
    float* x = e->x.as<float>();
    float* qkv = e->qkv.as<float>();
    float* sc = e->scores.as<float>();
    float* ctx = e->ctx.as<float>();
    float* hb = e->hbuf.as<float>();
    float* tmp = e->tmp.as<float>();
    const float inv_sqrt = 1.0f / std::sqrt((float)dh);
    for (int64_t l = 0; l < e->layers; ++l) {
        const float* P_ = W + e->o_layer0 + l * e->layer_stride;
        // q, k, v projections in one GEMM: (T x H) . (3H x H)^T
        const int64_t lo = e->o_layer0 + l * e->layer_stride;       // blob offset of this layer's parameters
        const float* Pp_ = l >= 1 ? W + e->o_layer0 + (l - 1) * e->layer_stride : nullptr;     // the previous layer's parameters
        { EncTimed tm(e, ES_QKV, st);
        if (fold && l >= 1) {   // x holds the RAW output of the previous FFN-out: its LayerNorm is folded into this product
            const LnFold lf{st2, ln_parts, 64, e->eps, LV + e->v_qkv[l], nullptr, nullptr, nullptr};
            linear_planes(e, st, LR_QKV, xp, hp, WPF + e->f_qkv[l], e->wscale_f[fold_scale_index(l, 0)], qkv, nullptr, 0, LV + e->v_qkv[l] + 3 * H, nullptr,
                          (int)T, (int)(3 * H), (int)H, EPI_BIAS, nullptr, nullptr, att16 ? &att_out : nullptr, &lf, e->wpf_plane);
        }
        else if (P) linear_planes(e, st, LR_QKV, xp, hp, wp(lo + e->r_wqkv), ws(l, 0), qkv, nullptr, 0, P_ + e->r_bqkv, nullptr, (int)T, (int)(3 * H),
                             (int)H, EPI_BIAS, part_wide, nullptr, att16 ? &att_out : nullptr);
        else linear(e, st, x, P_ + e->r_wqkv, qkv, P_ + e->r_bqkv, nullptr, (int)T, (int)(3 * H), (int)H, EPI_BIAS, nullptr); }
        EncTimed* t_att = new EncTimed(e, ES_ATTENTION, st);
        if (fused) {
            // softmax(Q K^T / sqrt(dh) + mask) V, one wave per (sequence, head, 32 queries), scores never leave registers
            const dim3 grid((unsigned)((L + 31) / 32), (unsigned)heads, (unsigned)N);
            uint16_t* cp_ = P ? ctxp : nullptr;
#define CLB_ATT(NT_) hipLaunchKernelGGL(attention_fused_kernel<NT_>, grid, dim3(64), 0, st, qkv, d_mask, ctx, (int)L, (int)H, inv_sqrt, cp_, hp, PF)
#ifdef CLB_ABLATIONS
            if (att16 && L > 32 && e->attention_mode == 5) {
                // the query blocks of a (sequence, head) share its K / V tiles through LDS (attention_f16_lds_kernel): NW waves
                // of QB blocks per work-group -- a whole sequence up to 512 tokens at QB = 2; bit-identical to the kernels below
                const int QB = L >= 128 && att_qb2() ? 2 : 1;
                int NW = (int)((L + 32 * QB - 1) / (32 * QB));
                // QB = 2 needs ~290 registers: up to four waves run one per SIMD with the AGPRs as overflow; five and more share
                // SIMDs (256 registers) and spill unless the staging is spread over eight waves (idle ones only stage)
                NW = NW > 8 ? 8 : NW == 7 ? 8 : NW < 2 ? 2 : (QB == 2 && NW >= 5) ? 8 : NW;
                const dim3 gridl((unsigned)((L + 32 * QB * NW - 1) / (32 * QB * NW)), (unsigned)heads, (unsigned)N);
#define CLB_ATTL(QB_, NW_)                                                                                              \
    if (QB == QB_ && NW == NW_)                                                                                         \
        hipLaunchKernelGGL((attention_f16_lds_kernel<QB_, NW_>), gridl, dim3(64 * NW_), 0, st, att_out.qk, qk_plane, T, att_out.vt, vt_plane, \
                           d_mask, (int)L, (int)H, inv_sqrt, ctxp, hp, PF, pk ? pk->cu : nullptr);
                CLB_ATTL(1, 2) CLB_ATTL(1, 3) CLB_ATTL(1, 4) CLB_ATTL(2, 2) CLB_ATTL(2, 3) CLB_ATTL(2, 4) CLB_ATTL(2, 5) CLB_ATTL(2, 6) CLB_ATTL(2, 8)
                CLB_ATTL(1, 5) CLB_ATTL(1, 6) CLB_ATTL(1, 8)
#undef CLB_ATTL
            } else
#endif
            if (att16 && L >= 128 && att_qb2()) {     // long sequences: two query blocks per wave share a key tile's K / V fragments
                const dim3 grid2((unsigned)((L + 63) / 64), (unsigned)heads, (unsigned)N);
                hipLaunchKernelGGL(attention_f16_kernel<2>, grid2, dim3(64), 0, st, att_out.qk, qk_plane, T, att_out.vt, vt_plane, d_mask, (int)L,
                                   (int)H, inv_sqrt, ctxp, hp, PF, pk ? pk->cu : nullptr);
            } else if (att16)
                hipLaunchKernelGGL(attention_f16_kernel<1>, grid, dim3(64), 0, st, att_out.qk, qk_plane, T, att_out.vt, vt_plane, d_mask, (int)L,
                                   (int)H, inv_sqrt, ctxp, hp, PF, pk ? pk->cu : nullptr);
            else if (L > 64 && e->attention_mode != 1)
                hipLaunchKernelGGL(attention_online_kernel, grid, dim3(64), 0, st, qkv, d_mask, ctx, (int)L, (int)H, inv_sqrt, cp_, hp, PF);
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
            if (P) hipLaunchKernelGGL(split_planes_kernel, dim3(blocks_for(hp / 4)), dim3(256), 0, st, ctx, ctxp, T, (int)H, hp, PF, kF16ActScale);
        }
        delete t_att;
        // attention output + residual, LayerNorm
        const LnArgs ln1{P_ + e->r_g1, P_ + e->r_b1n, e->eps}, ln2{P_ + e->r_g2, P_ + e->r_b2n, e->eps};
        { EncTimed tm(e, ES_ATTN_OUT, st);
        if (fold) {     // raw rows + their partial statistics; the residual is x, raw too from layer 1 on (normalised on the fly)
            const LnFold lf{l >= 1 ? st2 : nullptr, ln_parts, 64, e->eps, nullptr, l >= 1 ? Pp_ + e->r_g2 : nullptr, l >= 1 ? Pp_ + e->r_b2n : nullptr, st1};
            linear_planes(e, st, LR_ATTN_OUT, ctxp, hp, wp(lo + e->r_wo), ws(l, 1), tmp, tmpp, hp, P_ + e->r_bo, x, (int)T, (int)H, (int)H,
                          EPI_BIAS | EPI_RESID, nullptr, nullptr, nullptr, &lf);
        }
        else if (P) linear_planes(e, st, LR_ATTN_OUT, ctxp, hp, wp(lo + e->r_wo), ws(l, 1), tmp, tmpp, hp, P_ + e->r_bo, x, (int)T, (int)H, (int)H,
                             EPI_BIAS | EPI_RESID, part, &ln1);
        else linear(e, st, ctx, P_ + e->r_wo, tmp, P_ + e->r_bo, x, (int)T, (int)H, (int)H, EPI_BIAS | EPI_RESID, part, &ln1); }
        // feed-forward: GELU(x W1^T + b1) W2^T + b2 + residual, LayerNorm.  On the plane path the (T x I) intermediate
        // exists only as the bf16 planes the second Linear reads
        { EncTimed tm(e, ES_FFN_IN, st);
        if (fold) {
            const LnFold lf{st1, ln_parts, 64, e->eps, LV + e->v_w1[l], nullptr, nullptr, nullptr};
            linear_planes(e, st, LR_FFN_IN, tmpp, hp, WPF + e->f_w1[l], e->wscale_f[fold_scale_index(l, 1)], nullptr, hbp, ip, LV + e->v_w1[l] + I, nullptr,
                          (int)T, (int)I, (int)H, EPI_BIAS | EPI_GELU, nullptr, nullptr, nullptr, &lf, e->wpf_plane);
        }
        else if (P) linear_planes(e, st, LR_FFN_IN, tmpp, hp, wp(lo + e->r_w1), ws(l, 2), nullptr, hbp, ip, P_ + e->r_b1, nullptr, (int)T, (int)I, (int)H,
                             EPI_BIAS | EPI_GELU, part_wide, nullptr);
        else linear(e, st, tmp, P_ + e->r_w1, hb, P_ + e->r_b1, nullptr, (int)T, (int)I, (int)H, EPI_BIAS | EPI_GELU, nullptr); }
        { EncTimed tm(e, ES_FFN_OUT, st);
        if (fold) {     // the residual is the raw attention-output row: its (first) LayerNorm is applied on the fly
            const LnFold lf{st1, ln_parts, 64, e->eps, nullptr, P_ + e->r_g1, P_ + e->r_b1n, st2};
            linear_planes(e, st, LR_FFN_OUT, hbp, ip, wp(lo + e->r_w2), ws(l, 3), x, xp, hp, P_ + e->r_b2, tmp, (int)T, (int)H, (int)I,
                          EPI_BIAS | EPI_RESID, nullptr, nullptr, nullptr, &lf);
        }
        else if (P) linear_planes(e, st, LR_FFN_OUT, hbp, ip, wp(lo + e->r_w2), ws(l, 3), x, xp, hp, P_ + e->r_b2, tmp, (int)T, (int)H, (int)I,
                             EPI_BIAS | EPI_RESID, part, &ln2);
        else linear(e, st, hb, P_ + e->r_w2, x, P_ + e->r_b2, tmp, (int)T, (int)H, (int)I, EPI_BIAS | EPI_RESID, part, &ln2); }
    }
    // ColBERT projection: Layers.Dense(hidden -> dim)
    { EncTimed tm(e, ES_PROJECTION, st);
    if (fold) {         // x holds the raw output of the last FFN-out
        const LnFold lf{st2, ln_parts, 64, e->eps, LV + e->v_lin, nullptr, nullptr, nullptr};
        linear_planes(e, st, LR_PROJ, xp, hp, WPF + e->f_lin, e->wscale_f.back(), e->out.as<float>(), nullptr, 0, LV + e->v_lin + e->dim, nullptr, (int)T,
                      (int)e->dim, (int)H, EPI_BIAS, nullptr, nullptr, nullptr, &lf, e->wpf_plane);
    }
    else if (P) linear_planes(e, st, LR_PROJ, xp, hp, wp(e->o_lin_w), ws(e->layers, 0), e->out.as<float>(), nullptr, 0, W + e->o_lin_b, nullptr, (int)T,
                         (int)e->dim, (int)H, EPI_BIAS, part, nullptr);
    else linear(e, st, x, W + e->o_lin_w, e->out.as<float>(), W + e->o_lin_b, nullptr, (int)T, (int)e->dim, (int)H, EPI_BIAS, part); }
    CLB_HIP(hipGetLastError());
    if (!sync) return CLB_OK;
    int herr = 0;
    CLB_HIP(hipMemcpyAsync(&herr, e->err.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipMemsetAsync(e->err.p, 0, sizeof(int), st));
    CLB_HIP(hipStreamSynchronize(st));
    if (herr & 1) return fail(CLB_EBOUNDS, "token id outside the vocabulary (ids are 1-based, 1..%lld)", (long long)e->vocab);
    // bit 2 can only have been left by an EARLIER asynchronous encode nobody has checked yet (this call's epilogue has not run):
    // it was read and cleared with bit 1 above, so it is reported here rather than dropped
    if (herr & 2) return fail(CLB_EDOMAIN, "non-finite encoder output in an earlier asynchronous encode on this handle (an activation "
                                           "outside the range of the f16 operand split? clb_encoder_set_gemm_mode(e, 2) selects bf16x6)");
    return CLB_OK;
}

// waits for `st`, then reads and clears the non-finite-output bit of the sticky flag (set by the epilogue kernels)
int finish_checked(clb_encoder* e, hipStream_t st) {
    int herr = 0;
    CLB_HIP(hipMemcpyAsync(&herr, e->err.p, sizeof(int), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipMemsetAsync(e->err.p, 0, sizeof(int), st));
    CLB_HIP(hipStreamSynchronize(st));
    if (herr & 2) return fail(CLB_EDOMAIN, "non-finite encoder output (an activation outside the range of the f16 operand split? "
                                           "clb_encoder_set_gemm_mode(e, 2) selects the bf16x6 split)");
    return CLB_OK;
}

// ---- the document epilogue of a PACKED batch (checkpoint.jl:37-52): every row is an attended token; kept rows (id not in the
// skiplist) are normalised and written to their rank among the kept rows -- the sequences follow one another, so the global
// rank IS the compacted column -- and doclens are differences of that rank at the sequence boundaries
__global__ void packed_keep_kernel(const int32_t* __restrict__ ids, int64_t rows, const int64_t* __restrict__ skip, int nskip,
                                   uint32_t* __restrict__ keep) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == rows) keep[j] = 0u;       // the pad slot exclusive_scan_u32 reads (it scans rows + 1 inputs: out[rows] = total)
    if (j >= rows) return;
    const int64_t id = ids[j];
    bool k = true;
    for (int s = 0; s < nskip; ++s) k = k && (id != skip[s]);
    keep[j] = k ? 1u : 0u;
}
__global__ void packed_doclens_kernel(const uint32_t* __restrict__ rank /*rows + 1*/, const int32_t* __restrict__ cu, int N,
                                      int64_t* __restrict__ doclens, int64_t* __restrict__ n_out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) doclens[n] = (int64_t)rank[cu[n + 1]] - (int64_t)rank[cu[n]];
    if (n == 0) *n_out = (int64_t)rank[cu[N]];
}
__global__ void packed_normalize_kernel(const float* __restrict__ D, int dim, int64_t rows, const uint32_t* __restrict__ keep,
                                        const uint32_t* __restrict__ rank, float* __restrict__ out, int* __restrict__ err) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= rows) return;
    const float* x = D + j * dim;
    const float n2 = sumsq_canonical(x, dim);
    if (!(n2 <= FLT_MAX)) atomicOr(err, 2);
    if (!keep[j]) return;
    float* o = out + (size_t)rank[j] * dim;
    const float den = sqrtf(n2) + FLT_EPSILON;              // epilogue_normalize_kernel's arithmetic
    for (int d = 0; d < dim; ++d) o[d] = x[d] / den;
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
    {   // the Linear weights as bf16 planes, split ONCE (the region from the first layer to the end of the blob: biases and
        // LayerNorm parameters ride along unused).  1.5x the bytes of the fp32 region.
        const char* v = CLB_ENV("COLBERT_ENCODER_PLANES");
        const int64_t n_lin = n_weights - e->o_layer0;
        // (every Linear's N must be a multiple of 4 for the plane GEMM of the product library: H, 3H and I are; dim may not be)
        e->planes = (!v || atoi(v) != 0) && H % 32 == 0 && I % 32 == 0 && e->o_layer0 % 4 == 0 && n_lin * 6 < ((int64_t)1 << 31) &&
                    (kAblations || e->dim % 4 == 0);
        if (e->planes) {
            e->wp_plane = n_lin;
            if ((rc = e->wplanes.alloc(sizeof(uint16_t) * 3 * n_lin)) || (rc = split_weights(e, plane_format(e->gemm_mode)))) {
                clb_encoder_destroy(e);
                return rc;
            }
        }
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
    if (mode < 0 || mode > 3) return fail(CLB_EARGUMENT, "gemm mode %d: 0 = fp32 MFMA, 1 = bf16x3, 2 = bf16x6, 3 = f16x3", mode);
    e->gemm_mode = mode;
    return CLB_OK;
}

int clb_encoder_set_ln_fold(clb_encoder* e, int mode) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    if (mode < -1 || mode > 1) return fail(CLB_EARGUMENT, "LayerNorm folding: -1 = long batches only (default), 0 = never, 1 = always");
    if (mode == 1 && e->H / 64 > kLnMaxParts)
        return fail(CLB_EUNSUPPORTED, "LayerNorm folding carries at most %d partial statistics per row: hidden sizes up to %d (this model: %lld)",
                    kLnMaxParts, 64 * kLnMaxParts, (long long)e->H);
    e->ln_fold = mode;
    return CLB_OK;
}

int clb_encoder_set_attention_mode(clb_encoder* e, int mode) {
    if (!e) return fail(CLB_EARGUMENT, "null encoder");
    if (mode == 5 && !kAblations)
        return fail(CLB_EUNSUPPORTED, "attention mode 5 (K / V tiles shared through LDS: bit-identical to mode 0 and slower) is built into tuning libraries only (make ABLATIONS=1)");
    if (mode < 0 || mode > 5 || mode == 4)
        return fail(CLB_EARGUMENT, "attention mode %d: 0 = fused, 1 = register-resident, 2 = three kernels, 3 = fused on the fp32 MFMA, "
                                   "5 = fused with the K / V tiles of a (sequence, head) shared through LDS", mode);
    e->attention_mode = mode;
    return CLB_OK;
}

int clb_encode(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N, float* out) {
    CLB_TRY(upload_inputs(e, integer_ids, bitmask, L, N));
    CLB_TRY(forward(e, L, N, e->stream, e->ids.as<int32_t>(), e->mask.as<uint8_t>()));
    CLB_HIP(hipMemcpy(out, e->out.p, sizeof(float) * L * N * e->dim, hipMemcpyDeviceToHost));
    return CLB_OK;
}

// The host entry point packs its batch itself when it may: every unattended token must be one the skiplist drops (the [PAD]
// padding of tensorize_docs is; a caller's own mask need not be) and the encoder must be able to (can_pack) -- then the rows
// the output never sees are not computed.  Attended tokens keep their positions, so any mask shape packs, not only prefixes.
static int encode_docs_packed_host(clb_encoder* e, const int32_t* ids, const uint8_t* mask, int64_t L, int64_t N, const int64_t* skiplist,
                                   int64_t n_skip, float* out_embs, int64_t* doclens, int64_t* n_out, bool* done) {
    *done = false;
    if (L * N < 1) return CLB_OK;
    std::vector<int32_t> pid, ppos, pseq, cu((size_t)N + 1, 0);      // ids, positions and sequence of every attended row; row offsets
    pid.reserve((size_t)L * N); ppos.reserve((size_t)L * N); pseq.reserve((size_t)L * N);
    for (int64_t n = 0; n < N; ++n) {
        int64_t last = -1;
        for (int64_t l = 0; l < L; ++l) {
            const int32_t id = ids[n * L + l];
            if (mask[n * L + l]) { pid.push_back(id); ppos.push_back((int32_t)l); pseq.push_back((int32_t)n); last = l; continue; }
            bool dropped = false;
            for (int64_t k = 0; k < n_skip; ++k) dropped = dropped || (int64_t)id == skiplist[k];
            if (!dropped) return CLB_OK;             // an unattended token the reference keeps: the padded path computes it
        }
        if (last < 0) return CLB_OK;                 // a sequence without attended tokens: padded path
        cu[(size_t)n + 1] = (int32_t)pid.size();
    }
    const int64_t rows = (int64_t)pid.size();
    // positions are those of the padded layout, but the key-blocked V buffer is indexed by the RANK of a token in its sequence
    // (attention_f16_kernel walks a sequence's rows): re-number per sequence for V / attention, keep `ppos` for the embeddings
    std::vector<int32_t> prank((size_t)rows);
    int64_t longest = 0;
    for (int64_t n = 0; n < N; ++n) {
        for (int32_t r = cu[(size_t)n]; r < cu[(size_t)n + 1]; ++r) prank[(size_t)r] = r - cu[(size_t)n];
        longest = std::max<int64_t>(longest, cu[(size_t)n + 1] - cu[(size_t)n]);
    }
    hipStream_t st = e->stream;
    DevBuf dIds, dPos, dRank, dSeq, dCu, dSkip, dOut, dLens, dN;
    CLB_TRY(upload(dIds, pid.data(), sizeof(int32_t) * rows, st));
    CLB_TRY(upload(dPos, ppos.data(), sizeof(int32_t) * rows, st));
    CLB_TRY(upload(dRank, prank.data(), sizeof(int32_t) * rows, st));
    CLB_TRY(upload(dSeq, pseq.data(), sizeof(int32_t) * rows, st));
    CLB_TRY(upload(dCu, cu.data(), sizeof(int32_t) * (N + 1), st));
    CLB_TRY(upload(dSkip, skiplist, sizeof(int64_t) * std::max<int64_t>(n_skip, 1), st));
    CLB_TRY(dOut.alloc(sizeof(float) * e->dim * rows));
    CLB_TRY(dLens.alloc(sizeof(int64_t) * N));
    CLB_TRY(dN.alloc(sizeof(int64_t)));
    CLB_TRY(e->pkeep.ensure(sizeof(uint32_t) * (rows + 1)));
    CLB_TRY(e->prank.ensure(sizeof(uint32_t) * (rows + 1)));
    const Packed pk{dPos.as<int32_t>(), dSeq.as<int32_t>(), dCu.as<int32_t>(), rows, dRank.as<int32_t>()};
    CLB_TRY(forward(e, longest, N, st, dIds.as<int32_t>(), nullptr, /*sync=*/true, &pk));
    hipLaunchKernelGGL(packed_keep_kernel, dim3(blocks_for(rows + 1, 256)), dim3(256), 0, st, dIds.as<int32_t>(), rows, dSkip.as<int64_t>(),
                       (int)n_skip, e->pkeep.as<uint32_t>());
    CLB_TRY(exclusive_scan_u32(e->pkeep.as<uint32_t>(), e->prank.as<uint32_t>(), (size_t)rows, st, &e->scan_tmp));
    hipLaunchKernelGGL(packed_doclens_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, st, e->prank.as<uint32_t>(), dCu.as<int32_t>(), (int)N,
                       dLens.as<int64_t>(), dN.as<int64_t>());
    hipLaunchKernelGGL(packed_normalize_kernel, dim3(blocks_for(rows, 64)), dim3(64), 0, st, e->out.as<float>(), (int)e->dim, rows,
                       e->pkeep.as<uint32_t>(), e->prank.as<uint32_t>(), dOut.as<float>(), e->err.as<int>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(doclens, dLens.p, sizeof(int64_t) * N, hipMemcpyDeviceToHost, st));
    CLB_HIP(hipMemcpyAsync(n_out, dN.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    CLB_HIP(hipStreamSynchronize(st));
    if (*n_out > 0) CLB_HIP(hipMemcpyAsync(out_embs, dOut.p, sizeof(float) * e->dim * (*n_out), hipMemcpyDeviceToHost, st));
    *done = true;
    return finish_checked(e, st);
}

int clb_encode_docs(clb_encoder* e, const int32_t* integer_ids, const uint8_t* bitmask, int64_t L, int64_t N,
                    const int64_t* skiplist, int64_t n_skip, float* out_embs, int64_t* doclens, int64_t* n_out) {
    if (e && integer_ids && bitmask && skiplist && out_embs && doclens && n_out && L >= 1 && N >= 1 && L <= e->max_pos && can_pack(e, L, L * N)) {
        CLB_TRY(use_device(e->device));
        bool done = false;
        CLB_TRY(encode_docs_packed_host(e, integer_ids, bitmask, L, N, skiplist, n_skip, out_embs, doclens, n_out, &done));
        if (done) return CLB_OK;
    }
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
                       (int)L, (int)N, dMask.as<uint8_t>(), dStart.as<int64_t>(), dOut.as<float>(), e->err.as<int>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(out_embs, dOut.p, sizeof(float) * e->dim * run, hipMemcpyDeviceToHost, st));
    return finish_checked(e, st);
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
                       (int)L, (int)N, dMask.as<uint8_t>(), (const int64_t*)nullptr, dOut.as<float>(), e->err.as<int>());
    CLB_HIP(hipGetLastError());
    CLB_HIP(hipMemcpyAsync(out, dOut.p, sizeof(float) * e->dim * L * N, hipMemcpyDeviceToHost, st));
    return finish_checked(e, st);
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
                               (int)e->dim, (int64_t)(L * N), d_integer_ids, d_skiplist, (int)n_skip, d_out, e->err.as<int>());
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

int clb_encode_docs_device(clb_encoder* e, const int32_t* d_integer_ids, const uint8_t* d_bitmask, int64_t L, int64_t N,
                           const int64_t* d_skiplist, int64_t n_skip, float* d_out_embs, int64_t* d_doclens, int64_t* d_n_out,
                           void* hip_stream) {
    if (!e || !d_integer_ids || !d_bitmask || !d_out_embs || !d_doclens || !d_n_out) return fail(CLB_EARGUMENT, "null argument");
    if (L < 1 || N < 1) return fail(CLB_EARGUMENT, "empty batch");
    if (L > e->max_pos) return fail(CLB_EBOUNDS, "sequence length %lld exceeds max_position_embeddings %lld", (long long)L, (long long)e->max_pos);
    CLB_TRY(use_device(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    CLB_TRY(e->qmask.ensure((size_t)L * N));
    CLB_TRY(e->qlens.ensure(sizeof(int64_t) * N));          // here: the exclusive scan of the document lengths
    CLB_TRY(forward(e, L, N, st, d_integer_ids, d_bitmask, /*sync=*/false));
    {
        EncTimed tm(e, ES_EPILOGUE, st);
        hipLaunchKernelGGL(epilogue_mask_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, st, d_integer_ids, (int)L, (int)N, d_skiplist,
                           (int)n_skip, e->qmask.as<uint8_t>(), d_doclens);
        hipLaunchKernelGGL(doclens_scan_kernel, dim3(1), dim3(64), 0, st, d_doclens, (int)N, e->qlens.as<int64_t>(), d_n_out);
        hipLaunchKernelGGL(epilogue_normalize_kernel, dim3(blocks_for(L * N, 64)), dim3(64), 0, st, e->out.as<float>(), (int)e->dim,
                           (int)L, (int)N, e->qmask.as<uint8_t>(), e->qlens.as<int64_t>(), d_out_embs, e->err.as<int>());
    }
    CLB_HIP(hipGetLastError());
    return CLB_OK;
}

int clb_encode_docs_packed_device(clb_encoder* e, const int32_t* d_ids, const int32_t* d_pos, const int32_t* d_seq, const int32_t* d_cu,
                                  int64_t N, int64_t Lmax, int64_t rows, const int64_t* d_skiplist, int64_t n_skip, float* d_out_embs,
                                  int64_t* d_doclens, int64_t* d_n_out, void* hip_stream) {
    if (!e || !d_ids || !d_pos || !d_seq || !d_cu || !d_out_embs || !d_doclens || !d_n_out) return fail(CLB_EARGUMENT, "null argument");
    if (N < 1 || Lmax < 1 || rows < N || rows > N * Lmax) return fail(CLB_EARGUMENT, "packed batch: N >= 1, N <= rows <= N * Lmax");
    if (Lmax > e->max_pos) return fail(CLB_EBOUNDS, "sequence length %lld exceeds max_position_embeddings %lld", (long long)Lmax, (long long)e->max_pos);
    CLB_TRY(use_device(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    CLB_TRY(e->pkeep.ensure(sizeof(uint32_t) * (rows + 1)));
    CLB_TRY(e->prank.ensure(sizeof(uint32_t) * (rows + 1)));
    const Packed pk{d_pos, d_seq, d_cu, rows};
    CLB_TRY(forward(e, Lmax, N, st, d_ids, nullptr, /*sync=*/false, &pk));
    {
        EncTimed tm(e, ES_EPILOGUE, st);
        hipLaunchKernelGGL(packed_keep_kernel, dim3(blocks_for(rows + 1, 256)), dim3(256), 0, st, d_ids, rows, d_skiplist, (int)n_skip,
                           e->pkeep.as<uint32_t>());
        CLB_TRY(exclusive_scan_u32(e->pkeep.as<uint32_t>(), e->prank.as<uint32_t>(), (size_t)rows, st, &e->scan_tmp));
        hipLaunchKernelGGL(packed_doclens_kernel, dim3(blocks_for(N, 64)), dim3(64), 0, st, e->prank.as<uint32_t>(), d_cu, (int)N, d_doclens,
                           d_n_out);
        hipLaunchKernelGGL(packed_normalize_kernel, dim3(blocks_for(rows, 64)), dim3(64), 0, st, e->out.as<float>(), (int)e->dim, rows,
                           e->pkeep.as<uint32_t>(), e->prank.as<uint32_t>(), d_out_embs, e->err.as<int>());
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
    CLB_HIP(hipMemset(e->err.p, 0, sizeof(int)));     // sticky until read: covers EVERY encode since the previous check
    if (herr & 1) return fail(CLB_EBOUNDS, "token id outside the vocabulary (ids are 1-based, 1..%lld)", (long long)e->vocab);
    if (herr & 2) return fail(CLB_EDOMAIN, "non-finite encoder output (an activation outside the range of the f16 operand split? "
                                           "clb_encoder_set_gemm_mode(e, 2) selects the bf16x6 split)");
    return CLB_OK;
}

int clb_encoder_error_flag_device(clb_encoder* e, void** d_flag) {
    if (!e || !d_flag) return fail(CLB_EARGUMENT, "null argument");
    CLB_TRY(use_device(e->device));
    if (!e->err.p) {
        CLB_TRY(e->err.ensure(sizeof(int)));
        CLB_HIP(hipMemset(e->err.p, 0, sizeof(int)));
    }
    *d_flag = e->err.p;
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
