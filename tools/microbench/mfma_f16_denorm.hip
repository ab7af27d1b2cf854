// Does v_mfma_f32_32x32x16_f16 multiply fp16 SUBNORMAL inputs exactly (no flush to zero)?  Pass 1 of round 3 feeds fp16
// query values and fp16 bucket weights to it and bounds the rounding loss by what v_cvt_f16_f32 loses, subnormals kept.
//   hipcc --offload-arch=gfx950 -O3 mfma_f16_denorm.hip -o mfma_f16_denorm && ./mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* a_vals, const float* b_vals, float* out) {
    const int lane = threadIdx.x;
    // A[row = lane & 31][k = 8 (lane >> 5) + j] = a_vals[k] for every row; B[k][col = lane & 31] = b_vals[k]
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int kk = 8 * (lane >> 5) + j;
        a[j] = (_Float16)a_vals[kk];
        b[j] = (_Float16)b_vals[kk];
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

int main() {
    float ha[16], hb[16], *da, *db, *dout, r;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    struct { const char* name; float av, bv; } cases[] = {
        {"normal x normal", 0.5f, 0.25f},
        {"subnormal a (2^-20) x 1", 9.5367431640625e-07f, 1.0f},
        {"subnormal a (3 * 2^-24) x 1", 1.78813934326171875e-07f, 1.0f},
        {"subnormal a x subnormal b (2^-16 x 2^-16)", 1.52587890625e-05f, 1.52587890625e-05f},
        {"smallest subnormal 2^-24 x 0.04 (a bucket weight)", 5.9604644775390625e-08f, 0.0399780273f},
    };
    int bad = 0;
    for (auto& c : cases) {
        for (int i = 0; i < 16; ++i) { ha[i] = i == 3 ? c.av : 0.f; hb[i] = i == 3 ? c.bv : 0.f; }
        hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
        hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
        const float want = (float)((double)__half2float(__float2half_rn(c.av)) * (double)__half2float(__float2half_rn(c.bv)));
        printf("%-52s got %.10e want %.10e %s\n", c.name, r, want, r == want ? "exact" : "DIFFERENT");
        bad += r != want;
    }
    printf(bad ? "fp16 subnormals are NOT multiplied exactly\n" : "fp16 subnormal inputs are multiplied exactly\n");
    return bad;
}
