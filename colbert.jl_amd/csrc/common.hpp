// common.hpp -- shared host/device helpers of libcolbert_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/colbert_hip.h"

namespace clb {

// ---- error plumbing: no C++ exception crosses the ABI ----------------------------------------------
inline std::string& last_error() {
    static thread_local std::string e;
    return e;
}
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define CLB_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return clb::fail(CLB_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define CLB_TRY(expr)           \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

// Selects the device; fails loudly when there is none (no CPU fallback anywhere in this library).
inline int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(CLB_EHIP, "no HIP device available (hipGetDeviceCount: %s)",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(CLB_EHIP, "device %d out of range (0..%d)", device, n - 1);
    CLB_HIP(hipSetDevice(device));
    return CLB_OK;
}

// RAII device buffer (host-side bookkeeping only)
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(CLB_ENOMEM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return CLB_OK;
    }
    int ensure(size_t n) { return n <= bytes && p ? CLB_OK : alloc(n); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline int upload(DevBuf& b, const void* host, size_t bytes, hipStream_t st = nullptr) {
    CLB_TRY(b.alloc(bytes));
    if (bytes) CLB_HIP(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, st));
    return CLB_OK;
}

constexpr float kNegInf = -__builtin_huge_valf();

// A launch may use more than 64 KB of dynamic LDS only after the limit of that kernel has been raised -- per DEVICE
// (a process may hold handles on several GPUs, one host thread each), so the raise is remembered per (kernel, device).
inline void allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!done.insert({kernel, dev}).second) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipGetLastError();
}

// Launch-geometry knobs used while tuning: the shipped library compiles them to their defaults; a build with
// -DCLB_ABLATIONS reads CLB_DEBUG_* from the environment instead (make ABLATIONS=1).
// Comparison switches read from the environment (COLBERT_ENC_PLAN, COLBERT_PASS1_GATHER, ...): tuning builds only.  The product
// library answers "not set" without looking, and the kernels only those switches select are not compiled into it.
#ifdef CLB_ABLATIONS
#define CLB_ENV(NAME) getenv(NAME)
constexpr bool kAblations = true;
#else
#define CLB_ENV(NAME) (static_cast<const char*>(nullptr))
constexpr bool kAblations = false;
#endif
#ifdef CLB_ABLATIONS
inline int tuning_knob(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#define CLB_KNOB(NAME, DFLT) (clb::tuning_knob(NAME, DFLT))      // re-read on every call: a sweep can change it
#else
#define CLB_KNOB(NAME, DFLT) (DFLT)
#endif

// ---- device helpers ------------------------------------------------------------------------------
// float -> unsigned key with the same ordering (larger float <=> larger key); -0.0 < +0.0 here, which
// only matters for ties between zeros and is applied identically by the oracle-facing comparison
// because scores are compared as floats first (see topk kernels).
__device__ __forceinline__ uint32_t f32_order_key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

}  // namespace clb
