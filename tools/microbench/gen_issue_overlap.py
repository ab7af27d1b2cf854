#!/usr/bin/env python3
"""Generates issue_overlap.hip: does v_mfma_f32_32x32x16_bf16 overlap with VALU / LDS / SALU issue on gfx950?

The first version of this benchmark (mfma_overlap.hip) wrote its VALU filler in C++ and the compiler SLP-packed it
into v_pk_fma_f32, which MI355X_MICROARCH.md lists as an anti-lever beside MFMAs -- so its "no overlap" result said
nothing about ordinary VALU work.  Here every loop body is ONE asm volatile block: the instruction stream is exactly
what is written.

    python tools/microbench/gen_issue_overlap.py > tools/microbench/issue_overlap.hip
    hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_overlap.hip -o issue_overlap && ./issue_overlap
"""

NM = 10   # MFMAs per loop iteration (= one step of pass 1)

FILL = {
    "add": lambda i: f"v_add_u32 %[v{i % 8}], %[v{i % 8}], %[k]\\n",
    "perm": lambda i: f"v_perm_b32 %[v{i % 8}], %[v{i % 8}], %[k], %[sel]\\n",
    "fma": lambda i: f"v_fma_f32 %[v{i % 8}], %[v{i % 8}], %[kf], %[kf]\\n",
    "mul": lambda i: f"v_mul_f32 %[v{i % 8}], %[v{i % 8}], %[kf]\\n",
    "max3": lambda i: f"v_max3_f32 %[v{i % 8}], %[v{i % 8}], %[kf], %[v{(i + 1) % 8}]\\n",
    "pkmul": lambda i: f"v_pk_mul_f32 %[p{i % 4}], %[p{i % 4}], %[pk]\\n",
    "lds64": lambda i: f"ds_read_b64 %[d{i % 8}], %[addr] offset:{(i % 8) * 512}\\n",
    "lds128": lambda i: f"ds_read_b128 %[q{i % 4}], %[addr] offset:{(i % 4) * 1024}\\n",
    "salu": lambda i: f"s_add_u32 %[s{i % 4}], %[s{i % 4}], 3\\n",
}


def body(n_mfma, gap, tail=(), acc="acc", mfma="bf16"):
    """n_mfma MFMAs, each followed by the filler list `gap` ([(kind, count)]); then `tail` fillers."""
    out = []
    ctr = {}
    def emit(kind, count):
        for _ in range(count):
            i = ctr.get(kind, 0)
            ctr[kind] = i + 1
            out.append(FILL[kind](i))
    for _ in range(n_mfma):
        if mfma == "bf16":
            out.append(f"v_mfma_f32_32x32x16_bf16 %[{acc}], %[a], %[b], %[{acc}]\\n")
        elif mfma == "f32_16":      # the exact scorer's MFMA: two alternating accumulators as in score_exact_flat_kernel
            out.append(f"v_mfma_f32_16x16x4_f32 %[c{_ % 2}], %[kf], %[kf], %[c{_ % 2}]\\n")
        else:                        # f32_32: the fp32 GEMM's MFMA
            out.append(f"v_mfma_f32_32x32x2_f32 %[{acc}], %[kf], %[kf], %[{acc}]\\n")
        for kind, count in gap:
            emit(kind, count)
    for kind, count in tail:
        emit(kind, count)
    uses_lds = any(k.startswith("lds") for k, _ in list(gap) + list(tail))
    if uses_lds:
        out.append("s_waitcnt lgkmcnt(0)\\n")
    return "".join(f'            "{l}"\n' for l in out)


VARIANTS = [
    # name, n_mfma, gap fillers, tail fillers, agpr accumulator
    ("mfma10 (VGPR acc)", NM, [], [], False),
    ("mfma10 (AGPR acc)", NM, [], [], True),
    ("add60 only", 0, [], [("add", 60)], False),
    ("perm60 only", 0, [], [("perm", 60)], False),
    ("fma60 only", 0, [], [("fma", 60)], False),
    ("pkmul30 only", 0, [], [("pkmul", 30)], False),
    ("lds64x20 only", 0, [], [("lds64", 20)], False),
    ("salu50 only", 0, [], [("salu", 50)], False),
    ("mfma10 + 2 add/gap", NM, [("add", 2)], [], False),
    ("mfma10 + 4 add/gap", NM, [("add", 4)], [], False),
    ("mfma10 + 5 add/gap", NM, [("add", 5)], [], False),
    ("mfma10 + 6 add/gap", NM, [("add", 6)], [], False),
    ("mfma10 + 6 add/gap (AGPR acc)", NM, [("add", 6)], [], True),
    ("mfma10 + 8 add/gap", NM, [("add", 8)], [], False),
    ("mfma10 + 12 add/gap", NM, [("add", 12)], [], False),
    ("mfma10 + 6 perm/gap", NM, [("perm", 6)], [], False),
    ("mfma10 + 6 fma/gap", NM, [("fma", 6)], [], False),
    ("mfma10 + 6 mul/gap", NM, [("mul", 6)], [], False),
    ("mfma10 + 6 max3/gap", NM, [("max3", 6)], [], False),
    ("mfma10 + 3 pkmul/gap", NM, [("pkmul", 3)], [], False),
    ("mfma10 then add60", NM, [], [("add", 60)], False),
    ("mfma10 + 2 lds64/gap", NM, [("lds64", 2)], [], False),
    ("mfma10 + 5 salu/gap", NM, [("salu", 5)], [], False),
    ("mfma10 + 4 add + 2 lds64 + 4 salu /gap", NM, [("add", 4), ("lds64", 2), ("salu", 4)], [], False),
    ("mfma10 + 6 add + 2 lds64 + 5 salu /gap", NM, [("add", 6), ("lds64", 2), ("salu", 5)], [], False),
    # the fp32-input MFMAs (exact scorer: 16x16x4, two alternating accumulators; fp32 GEMM: 32x32x2)
    ("f32 16x16x4 x20", 20, [], [], False, "f32_16"),
    ("f32 16x16x4 x20 + 3 add/gap", 20, [("add", 3)], [], False, "f32_16"),
    ("f32 16x16x4 x20 + 6 add/gap", 20, [("add", 6)], [], False, "f32_16"),
    ("f32 16x16x4 x20 + 8 add/gap", 20, [("add", 8)], [], False, "f32_16"),
    ("f32 16x16x4 x20 + 4 fma + 1 lds64 /gap", 20, [("fma", 4), ("lds64", 1)], [], False, "f32_16"),
    ("add120 only", 0, [], [("add", 120)], False),
    ("f32 32x32x2 x10", NM, [], [], False, "f32_32"),
    ("f32 32x32x2 x10 + 6 add/gap", NM, [("add", 6)], [], False, "f32_32"),
    ("f32 32x32x2 x10 + 12 add/gap", NM, [("add", 12)], [], False, "f32_32"),
]

# two-role kernels: waves 0..3 of a work-group run role A, the others role B (one role-A wave per SIMD)
ROLES = [
    ("A: mfma10 | B: add60", (NM, [], []), (0, [], [("add", 60)])),
    ("A: mfma10 | B: perm40 + lds64x16 + salu30", (NM, [], []), (0, [], [("perm", 40), ("lds64", 16), ("salu", 30)])),
    ("A: mfma10 | B: mfma10", (NM, [], []), (NM, [], [])),
    ("A: f32 16x16x4 x20 | B: add120", (20, [], [], "acc", "f32_16"), (0, [], [("add", 120)])),
]

HEADER = r'''// GENERATED by gen_issue_overlap.py -- do not edit.  See that file for the purpose.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Regs {
    bf16x8 a, b; f32x16 acc; f32x4 c[2]; uint32_t v[8]; f32x2 p[4]; u32x2 d[8]; u32x4 q[4]; uint32_t s[4];
    uint32_t k, sel, addr; float kf; f32x2 pk;
};
__device__ __forceinline__ void init(Regs& R, const void* lds) {
    const int lane = threadIdx.x & 63;
    for (int j = 0; j < 8; ++j) { R.a[j] = (__bf16)(0.01f * ((lane * 7 + j) % 13)); R.b[j] = (__bf16)(0.02f * ((lane * 5 - j) % 11)); }
    for (int j = 0; j < 16; ++j) R.acc[j] = 0.f;
    for (int j = 0; j < 4; ++j) { R.c[0][j] = 0.f; R.c[1][j] = 0.f; }
    for (int j = 0; j < 8; ++j) { R.v[j] = lane * 3 + j; R.d[j] = u32x2{0u, 0u}; }
    for (int j = 0; j < 4; ++j) { R.p[j] = f32x2{1.f + lane, 2.f}; R.q[j] = u32x4{0u, 0u, 0u, 0u}; R.s[j] = j; }
    R.k = 0x01020304u + lane; R.sel = 0x07020500u; R.kf = 1.0001f; R.pk = f32x2{1.0001f, 0.9999f};
    R.addr = (uint32_t)(uintptr_t)lds + (uint32_t)lane * 8u;
}
__device__ __forceinline__ float fold(const Regs& R) {
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += R.acc[j];
    for (int j = 0; j < 4; ++j) s += R.c[0][j] + R.c[1][j];
    for (int j = 0; j < 8; ++j) s += (float)R.v[j] + (float)R.d[j][0] + (float)R.d[j][1];
    for (int j = 0; j < 4; ++j) s += R.p[j][0] + R.p[j][1] + (float)R.q[j][0] + (float)R.q[j][3] + (float)R.s[j];
    return s;
}
#define OPERANDS(ACC)                                                                                          \
    : [acc] ACC(R.acc), [c0] "+v"(R.c[0]), [c1] "+v"(R.c[1]), [v0] "+v"(R.v[0]), [v1] "+v"(R.v[1]), [v2] "+v"(R.v[2]), [v3] "+v"(R.v[3]),             \
      [v4] "+v"(R.v[4]), [v5] "+v"(R.v[5]), [v6] "+v"(R.v[6]), [v7] "+v"(R.v[7]),                               \
      [p0] "+v"(R.p[0]), [p1] "+v"(R.p[1]), [p2] "+v"(R.p[2]), [p3] "+v"(R.p[3]),                               \
      [d0] "+v"(R.d[0]), [d1] "+v"(R.d[1]), [d2] "+v"(R.d[2]), [d3] "+v"(R.d[3]),                               \
      [d4] "+v"(R.d[4]), [d5] "+v"(R.d[5]), [d6] "+v"(R.d[6]), [d7] "+v"(R.d[7]),                               \
      [q0] "+v"(R.q[0]), [q1] "+v"(R.q[1]), [q2] "+v"(R.q[2]), [q3] "+v"(R.q[3]),                               \
      [s0] "+s"(R.s[0]), [s1] "+s"(R.s[1]), [s2] "+s"(R.s[2]), [s3] "+s"(R.s[3])                                \
    : [a] "v"(R.a), [b] "v"(R.b), [k] "v"(R.k), [sel] "v"(R.sel), [kf] "v"(R.kf), [pk] "v"(R.pk),               \
      [addr] "v"(R.addr)                                                                                        \
    : "memory", "scc"

__device__ __forceinline__ void finish(const Regs& R, float* out, long long* cyc, long long t0, long long r0) {
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = fold(R);
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) / 64;
        cyc[2 * w] = t1 - t0; cyc[2 * w + 1] = r1 - r0;
    }
}
'''


def kernel(idx, n_mfma, gap, tail, agpr, mfma="bf16"):
    acc = '"+a"' if agpr else '"+v"'
    return f'''
__global__ __launch_bounds__(768) void k{idx}(float* out, long long* cyc, int iters) {{
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = i;
    __syncthreads();
    Regs R; init(R, lds);
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {{
        asm volatile(
{body(n_mfma, gap, tail, mfma=mfma)}            OPERANDS({acc}));
    }}
    finish(R, out, cyc, t0, r0);
}}
'''


def role_kernel(idx, ra, rb):
    return f'''
__global__ __launch_bounds__(768) void r{idx}(float* out, long long* cyc, int iters, int run_a, int run_b) {{
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = i;
    __syncthreads();
    Regs R; init(R, lds);
    const bool role_a = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (role_a) {{
        if (run_a) for (int it = 0; it < iters; ++it) {{
            asm volatile(
{body(*ra)}                OPERANDS("+v"));
        }}
    }} else {{
        if (run_b) for (int it = 0; it < iters; ++it) {{
            asm volatile(
{body(*rb)}                OPERANDS("+v"));
        }}
    }}
    finish(R, out, cyc, t0, r0);
}}
'''


MAIN_HEAD = r'''
template <typename F>
static void time_kernel(const char* name, int waves_per_simd, F launch) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 20000;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64) * 2);
    launch(blocks, threads, out, cyc, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(blocks, threads, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * (threads / 64) * 2);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> cy, ghz;
    for (size_t w = 0; w < h.size() / 2; ++w)
        if (h[2 * w + 1] > 0 && h[2 * w] > 1000) { cy.push_back((double)h[2 * w] / iters); ghz.push_back((double)h[2 * w] / h[2 * w + 1] * 0.1); }
    std::sort(cy.begin(), cy.end()); std::sort(ghz.begin(), ghz.end());
    const double c50 = cy.empty() ? 0 : cy[cy.size() / 2], g50 = ghz.empty() ? 0 : ghz[ghz.size() / 2];
    printf("%-52s w/SIMD %d: %8.1f cyc/iter (median wave) | clock %.2f GHz | wall %7.3f ms = %7.1f ns/iter\n", name,
           waves_per_simd, c50, g50, ms, ms * 1e6 / iters);
    fflush(stdout);
    hipFree(out); hipFree(cyc); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
'''


def main():
    print(HEADER)
    for i, v in enumerate(VARIANTS):
        print(kernel(i, *v[1:]))
    for i, (name, ra, rb) in enumerate(ROLES):
        print(role_kernel(i, ra, rb))
    print(MAIN_HEAD)
    for i, v in enumerate(VARIANTS):
        name = v[0]
        for w in (1, 2, 3):
            print(f'    time_kernel("{name}", {w}, [](int b, int t, float* o, long long* c, int it) {{ hipLaunchKernelGGL(k{i}, dim3(b), dim3(t), 0, 0, o, c, it); }});')
    for i, (name, ra, rb) in enumerate(ROLES):
        for w in (2, 3):
            for (ta, tb, tag) in ((1, 0, "A alone"), (0, 1, "B alone"), (1, 1, "A and B")):
                print(f'    time_kernel("{name} [{tag}]", {w}, [](int b, int t, float* o, long long* c, int it) {{ hipLaunchKernelGGL(r{i}, dim3(b), dim3(t), 0, 0, o, c, it, {ta}, {tb}); }});')
    print("    return 0;\n}")


if __name__ == "__main__":
    main()
