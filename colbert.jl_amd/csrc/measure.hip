// measure.hip -- measurement support of bench.py (include/colbert_hip.h: clb_measure_copy_rate, clb_measure_read_rate): what the
// memory system of THIS device delivers NOW, next to the data-sheet peak the roofline record divides by.  Not part
// of the search path (and outside the source hash that ties profiles/pmc_summary*.json to the search kernels).
#include <algorithm>

#include "common.hpp"

using namespace clb;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int FORM>
static __global__ __launch_bounds__(256) void copy_rate_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (FORM == 0) {
        for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
        return;
    }
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a, b, c, d;
        if (FORM == 1) { a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride]; }
        else {
            a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + stride);
            c = __builtin_nontemporal_load(src + i + 2 * stride); d = __builtin_nontemporal_load(src + i + 3 * stride);
        }
        if (FORM == 1) { dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d; }
        else {
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// read-only stream: every lane folds what it loads (four 16-byte pieces in flight); the store never happens for the fill pattern
// of the buffer, but the compiler cannot know that, so the loads stay
static __global__ __launch_bounds__(256) void read_rate_kernel(const u32x4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc += a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ d[0] ^ d[1] ^ d[2] ^ d[3];
    }
    for (; i < n16; i += stride) { const u32x4 a = src[i]; acc += a[0] ^ a[1] ^ a[2] ^ a[3]; }
    if (acc == 0x12345u) sink[blockIdx.x] = acc;
}

}  // namespace

extern "C" {

// What a plain stream reaches on THIS device at THIS moment: device-to-device copies of `bytes`, `reps` times between two HIP
// events after one untimed pass, by three forms of a 16-bytes-per-lane grid-stride kernel (FORM 0: non-temporal loads, plain
// stores, one piece per lane and iteration; 1: plain loads and stores, four pieces in flight; 2: non-temporal both ways, four in
// flight) and by the runtime's own hipMemcpyAsync -- the best of the four is reported (which one wins differs from box to box).
// bench.py quotes pass 1's achieved bandwidth against it next to the 8 TB/s of the data sheet.
int clb_measure_copy_rate(int device, int64_t bytes, int reps, double* gb_per_s) {
    if (!gb_per_s || bytes < 4096 || reps < 1) return fail(CLB_EARGUMENT, "copy rate: bytes >= 4096, reps >= 1, a result pointer");
    CLB_TRY(use_device(device));
    const size_t n16 = (size_t)bytes / 16;
    DevBuf a, b;
    CLB_TRY(a.alloc(n16 * 16));
    CLB_TRY(b.alloc(n16 * 16));
    CLB_HIP(hipMemset(a.p, 0x5a, n16 * 16));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CLB_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(CLB_EHIP, "hipEventCreate failed"); }
    const dim3 grid(256 * 8);      // eight work-groups per CU
    const u32x4* src = a.as<u32x4>();
    u32x4* dst = b.as<u32x4>();
    auto once = [&](int form) {
        if (form == 0) hipLaunchKernelGGL(copy_rate_kernel<0>, grid, dim3(256), 0, nullptr, src, dst, n16);
        else if (form == 1) hipLaunchKernelGGL(copy_rate_kernel<1>, grid, dim3(256), 0, nullptr, src, dst, n16);
        else if (form == 2) hipLaunchKernelGGL(copy_rate_kernel<2>, grid, dim3(256), 0, nullptr, src, dst, n16);
        else (void)hipMemcpyAsync(b.p, a.p, n16 * 16, hipMemcpyDeviceToDevice, nullptr);
    };
    double best = 0.0;
    bool ok = true;
    for (int form = 0; form < 4 && ok; ++form) {
        once(form);
        (void)hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps; ++r) once(form);
        (void)hipEventRecord(e1, nullptr);
        float ms = 0.f;
        ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && hipGetLastError() == hipSuccess && ms > 0.f;
        if (ok) best = std::max(best, 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9);      // bytes read + bytes written
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return fail(CLB_EHIP, "copy rate: timing failed");
    *gb_per_s = best;
    return CLB_OK;
}

// The read-only counterpart: `reps` passes of a kernel that only loads `bytes` (pass 1 is all reads: its actual HBM traffic per
// second is quoted against this number, its algorithmic bytes against the copy rate and the data sheet).
int clb_measure_read_rate(int device, int64_t bytes, int reps, double* gb_per_s) {
    if (!gb_per_s || bytes < 4096 || reps < 1) return fail(CLB_EARGUMENT, "read rate: bytes >= 4096, reps >= 1, a result pointer");
    CLB_TRY(use_device(device));
    const size_t n16 = (size_t)bytes / 16;
    DevBuf a, sink;
    CLB_TRY(a.alloc(n16 * 16));
    CLB_TRY(sink.alloc(sizeof(uint32_t) * 256 * 8));
    CLB_HIP(hipMemset(a.p, 0x5a, n16 * 16));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CLB_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(CLB_EHIP, "hipEventCreate failed"); }
    const dim3 grid(256 * 8);
    hipLaunchKernelGGL(read_rate_kernel, grid, dim3(256), 0, nullptr, (const u32x4*)a.as<u32x4>(), n16, sink.as<uint32_t>());
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(read_rate_kernel, grid, dim3(256), 0, nullptr, (const u32x4*)a.as<u32x4>(), n16, sink.as<uint32_t>());
    (void)hipEventRecord(e1, nullptr);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && hipGetLastError() == hipSuccess && ms > 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return fail(CLB_EHIP, "read rate: timing failed");
    *gb_per_s = (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;
    return CLB_OK;
}

}  // extern "C"
