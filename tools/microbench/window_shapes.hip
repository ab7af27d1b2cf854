// Micro-benchmark for the windowed pass 1 (round 3): what do its LOADS cost, with no arithmetic at all?
// A wave walks "steps" of 32 rows = 4 row groups x 8 rows; every row group streams its own chunk of CH rows (a
// passage's embeddings inside one code window: 32-B residual rows + 4-B code words at a pseudo-random place of a
// 1.8-GB buffer), and every row gathers one 64-B score row from the slice of the score table its XCD owns
// (`slice_bytes` per blockIdx % 8: L2-resident when <= ~2 MB).  Gather forms:
//   PAIR32  lane (r, h) reads 2 x 16 B of row r (round-2 pass 1)
//   QUADV   4 adjacent lanes read one row, to VGPRs (2 instructions per 32 rows)
//   QUADL   the same through LDS-DMA (global_load_lds_dwordx4) into a per-wave ring, then 2 ds_read_b128 per lane
// PART: at the end of every chunk the wave stores one 128-B line (the 32 per-token partial maxima of a chunk).
//   hipcc --offload-arch=gfx950 -O3 window_shapes.hip -o window_shapes && ./window_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { NOG = 0, PAIR32 = 1, QUADV = 2, QUADL = 3, B32V = 4, B32L = 5 };   // B32*: 32-byte rows (8-bit scores)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return x;
}

constexpr int kWaves = 12;
constexpr int kRing = 4;             // LDS ring slots per wave (QUADL), 2 KB each

template <int GATHER, int CH /* rows per chunk, multiple of 8; 0 = one contiguous stream */, bool STREAM, bool PART, int UNROLL = 4>
__global__ __launch_bounds__(64 * kWaves) void k(const unsigned char* __restrict__ tables, size_t slice_bytes,
                                                const unsigned char* __restrict__ resid,
                                                const uint32_t* __restrict__ codes, uint32_t total_rows,
                                                uint32_t steps_per_wave, float* __restrict__ part,
                                                uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[(GATHER == QUADL || GATHER == B32L) ? kWaves * kRing * 2048 : 16];
    const uint32_t lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * kWaves + wv;
    const unsigned char* table = tables + (size_t)(blockIdx.x & 7) * slice_bytes;
    const uint32_t rmask = (uint32_t)(slice_bytes / ((GATHER == B32V || GATHER == B32L) ? 32 : 64)) - 1u;
    const uint32_t g = r >> 3, j = r & 7;
    constexpr uint32_t kStepsPerChunk = CH ? CH / 8 : 1;
    uint32_t acc = 0;
    const uint32_t s0 = wave * steps_per_wave;
    unsigned char* myring = ring + ((GATHER == QUADL || GATHER == B32L) ? wv * kRing * 2048 : 0);

    auto row_of_step = [&](uint32_t s) -> uint32_t {       // the residual row this lane streams in step s
        if (CH == 0) return s * 32u + r;
        // row group g of this wave walks chunks; group g is g steps out of phase with group 0
        const uint32_t t = s - s0 + g;
        const uint32_t chunk = t / kStepsPerChunk, k8 = t % kStepsPerChunk;
        const uint32_t cid = (wave * 4u + g) * 65536u + chunk;
        const uint32_t base = (uint32_t)(((uint64_t)hash32(cid) * (uint64_t)(total_rows - 256u)) >> 32);
        return base + k8 * 8u + j;
    };
    auto issue_gather_lds = [&](uint32_t s) {
        unsigned char* slot = myring + ((s - s0) % kRing) * 2048;
        const uint32_t q = lane & 3u, qd = lane >> 2;
        const unsigned char* p0 = table + (size_t)(hash32(s * 32u + qd) & rmask) * 64 + 16u * q;
        const unsigned char* p1 = table + (size_t)(hash32(s * 32u + 16u + qd) & rmask) * 64 + 16u * q;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p0,
                                         (void __attribute__((address_space(3)))*)slot, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p1,
                                         (void __attribute__((address_space(3)))*)(slot + 1024), 16, 0, 0);
    };

    auto issue_gather_b32 = [&](uint32_t s) {
        unsigned char* slot = myring + ((s - s0) % kRing) * 2048;
        const uint32_t p = lane >> 1, sh = (lane & 1u) ^ ((p >> 3) & 1u);
        const unsigned char* p0 = table + (size_t)(hash32(s * 32u + p) & rmask) * 32 + 16u * sh;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p0,
                                         (void __attribute__((address_space(3)))*)slot, 16, 0, 0);
    };
    if (GATHER == B32L) { issue_gather_b32(s0); issue_gather_b32(s0 + 1); }
    if (GATHER == QUADL) {
        // explicit pipeline: gathers of step s+2 are issued before the data of step s is read
        issue_gather_lds(s0);
        issue_gather_lds(s0 + 1);
    }
#pragma unroll UNROLL
    for (uint32_t s = s0; s < s0 + steps_per_wave; ++s) {
        if (STREAM) {
            const uint32_t row = row_of_step(s);
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(resid + (size_t)row * 32 + 16u * h));
            const uint32_t c = __builtin_nontemporal_load(codes + row);
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3] ^ c;
        }
        if (GATHER == PAIR32) {
            const unsigned char* p = table + (size_t)(hash32(s * 32u + r) & rmask) * 64 + 16u * h;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 32);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        } else if (GATHER == QUADV) {
            const uint32_t q = lane & 3u, qd = lane >> 2;
            const unsigned char* p0 = table + (size_t)(hash32(s * 32u + qd) & rmask) * 64 + 16u * q;
            const unsigned char* p1 = table + (size_t)(hash32(s * 32u + 16u + qd) & rmask) * 64 + 16u * q;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p0), b = *reinterpret_cast<const u32x4*>(p1);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        } else if (GATHER == QUADL) {
            if (s + 2 < s0 + steps_per_wave) issue_gather_lds(s + 2);
            const unsigned char* slot = myring + ((s - s0) % kRing) * 2048;
            // lane (r, h): the two pieces of row r it needs as MFMA A fragments (rotated inside the row: bank spread)
            const uint32_t rot = (r >> 2) & 3u;
            const u32x4 a = *reinterpret_cast<const u32x4*>(slot + r * 64 + ((2u * h + rot) & 3u) * 16u);
            const u32x4 b = *reinterpret_cast<const u32x4*>(slot + r * 64 + ((2u * h + 1u + rot) & 3u) * 16u);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        }
        if (GATHER == B32V) {
            const unsigned char* p = table + (size_t)(hash32(s * 32u + r) & rmask) * 32 + 16u * h;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p);
            acc ^= a[0] ^ a[3] ^ a[1] ^ a[2];
        } else if (GATHER == B32L) {
            if (s + 2 < s0 + steps_per_wave) issue_gather_b32(s + 2);
            const unsigned char* slot = myring + ((s - s0) % kRing) * 2048;
            const u32x4 a = *reinterpret_cast<const u32x4*>(slot + r * 32 + ((h ^ ((r >> 3) & 1u)) * 16u));
            acc ^= a[0] ^ a[3] ^ a[1] ^ a[2];
        }
        if (PART && CH) {
            // a chunk ends for row group g' when (s - s0 + g') % kStepsPerChunk == kStepsPerChunk - 1: one 128-B store
#pragma unroll
            for (uint32_t gg = 0; gg < 4; ++gg) {
                const uint32_t t = s - s0 + gg;
                if (t % kStepsPerChunk == kStepsPerChunk - 1u) {
                    const uint32_t slotid = (wave * 4u + gg) * (steps_per_wave / kStepsPerChunk + 2u) + t / kStepsPerChunk;
                    if (h == 0) part[(size_t)slotid * 32 + r] = __uint_as_float(acc);
                }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int GATHER, int CH, bool STREAM, bool PART, int UNROLL = 4>
static void run(const char* name, const unsigned char* tables, size_t slice_bytes, const unsigned char* resid,
                const uint32_t* codes, uint32_t total_rows, float* part, uint32_t* out) {
    const int blocks = 256, threads = 64 * kWaves;
    const uint32_t steps_per_wave = 512;                       // 256 x 12 x 512 x 32 = 50.3 M rows
    const double rows = (double)blocks * kWaves * steps_per_wave * 32;
    hipLaunchKernelGGL((k<GATHER, CH, STREAM, PART, UNROLL>), dim3(blocks), dim3(threads), 0, 0, tables, slice_bytes, resid, codes,
                       total_rows, steps_per_wave, part, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((k<GATHER, CH, STREAM, PART, UNROLL>), dim3(blocks), dim3(threads), 0, 0, tables, slice_bytes, resid,
                           codes, total_rows, steps_per_wave, part, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-58s slice %4.1f MB/XCD: %.3f ms for %.1f M rows (%.2f TB/s of 36-B rows)\n", name, slice_bytes / 1048576.0,
           ms, rows * 1e-6, rows * 36.0 / ms * 1e-9);
    fflush(stdout);
}

int main() {
    const uint32_t total_rows = 50331648u;
    unsigned char *tabs, *resid; uint32_t *codes, *out; float* part;
    hipMalloc(&tabs, (size_t)8 << 23); hipMalloc(&resid, (size_t)total_rows * 32); hipMalloc(&codes, (size_t)total_rows * 4);
    hipMalloc(&out, 64); hipMalloc(&part, (size_t)256 * kWaves * 4 * 600 * 128);
    hipMemset(tabs, 1, (size_t)8 << 23); hipMemset(resid, 2, (size_t)total_rows * 32); hipMemset(codes, 3, (size_t)total_rows * 4);
    hipDeviceSynchronize();
    const size_t MB = 1 << 20;
#define RUN(G, CH, S, P, SL) run<G, CH, S, P>(#G " CH=" #CH " stream=" #S " part=" #P, tabs, SL, resid, codes, total_rows, part, out)
    RUN(NOG, 0, true, false, 8 * MB);          // contiguous stream alone
    RUN(PAIR32, 0, true, false, 8 * MB);       // round-2 pass 1
    RUN(QUADV, 0, true, false, 8 * MB);
    RUN(QUADL, 0, true, false, 8 * MB);
    RUN(B32V, 0, true, false, 4 * MB);         // 8-bit scores: K = 131 072 -> 4 MB per query
    RUN(B32L, 0, true, false, 4 * MB);
    RUN(B32V, 0, false, false, 4 * MB);
    RUN(B32L, 0, false, false, 4 * MB);
    RUN(B32V, 0, true, false, 2 * MB);
    RUN(B32L, 0, true, false, 2 * MB);
    RUN(B32V, 0, true, false, 8 * MB);         // K = 262 144
    RUN(B32L, 0, true, false, 8 * MB);
    RUN(QUADV, 0, true, false, 4 * MB);
    RUN(QUADL, 0, true, false, 4 * MB);
#define RUNU(G, S, SL, U) run<G, 0, S, false, U>(#G " stream=" #S " unroll=" #U, tabs, SL, resid, codes, total_rows, part, out)
    RUNU(PAIR32, true, 8 * MB, 2); RUNU(PAIR32, true, 8 * MB, 8); RUNU(PAIR32, true, 8 * MB, 16);
    RUNU(QUADV, true, 8 * MB, 2); RUNU(QUADV, true, 8 * MB, 8); RUNU(QUADV, true, 8 * MB, 16);
    RUNU(PAIR32, true, 2 * MB, 2); RUNU(PAIR32, true, 2 * MB, 8); RUNU(PAIR32, true, 2 * MB, 16);
    RUNU(QUADV, true, 2 * MB, 2); RUNU(QUADV, true, 2 * MB, 8); RUNU(QUADV, true, 2 * MB, 16);
    RUNU(NOG, true, 8 * MB, 2); RUNU(NOG, true, 8 * MB, 8); RUNU(NOG, true, 8 * MB, 16);
    if (getenv("WINDOWS")) {
    RUN(NOG, 24, true, false, 8 * MB);         // 24-row chunks alone
    RUN(NOG, 24, true, true, 8 * MB);          // + partial-maximum stores
    RUN(QUADV, 24, true, false, 2 * MB);       // the windowed pass: W = 4
    RUN(QUADL, 24, true, true, 2 * MB);
    }
    return 0;
}
