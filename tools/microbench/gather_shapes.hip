// Micro-benchmark: what does the lane -> byte mapping of pass 1's loads cost in the texture-address / L1 path?
// A wave handles "steps" of 32 rows (64 B each) picked pseudo-randomly from a score table, optionally next to the
// 1-KB residual stream of the step; the same rows are fetched with three lane mappings:
//   PAIR32 (pass 1 today): lane (r = l & 31, h = l >> 5) reads bytes [16h, 16h+16) and [32+16h, 48+16h) of row r:
//                          the two lanes of a row are 32 lanes apart, every lane is its own 16-byte request;
//   QUAD:                  4 adjacent lanes read the 64 contiguous bytes of one row, 16 rows per instruction;
//   PAIRADJ:               2 adjacent lanes read 32 contiguous bytes of one row, 32 rows per instruction.
// Residual stream: SPLIT (today: lane (r, h) reads bytes [32r + 16h, +16)) or CONTIG (lane l reads [16l, +16)).
// Tables: 8 x 8 MB, work-group b gathers from table b % 8 (the XCD it lands on under round-robin placement: pass 1's
// situation, one query's table per XCD), or one 4-GiB table (every row an HBM miss).
//   hipcc --offload-arch=gfx950 -O3 gather_shapes.hip -o gather_shapes && ./gather_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { PAIR32 = 0, QUAD = 1, PAIRADJ = 2, NOGATHER = 3 };
enum { NOSTREAM = 0, SPLIT = 1, CONTIG = 2 };

__device__ __forceinline__ uint32_t row_of(uint32_t g, uint32_t mask) {      // the g-th row of the launch
    uint32_t x = g * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return (uint32_t)(((uint64_t)x * (uint64_t)(mask + 1u)) >> 32);      // uniform in [0, rows)
}

template <int SHAPE, int STREAM>
__global__ __launch_bounds__(768) void k(const unsigned char* __restrict__ tables, size_t table_bytes, int per_xcd,
                                         const unsigned char* __restrict__ stream, uint32_t steps_per_wave,
                                         uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned char* table = tables + (per_xcd ? (size_t)(blockIdx.x & 7) * table_bytes : 0);
    const uint32_t mask = (uint32_t)(table_bytes / 64) - 1u;
    uint32_t acc = 0;
    const uint32_t s0 = wave * steps_per_wave;
#pragma unroll 4
    for (uint32_t s = s0; s < s0 + steps_per_wave; ++s) {
        const uint32_t g0 = s * 32u;
        if (STREAM != NOSTREAM) {
            const size_t off = (size_t)s * 1024 + (STREAM == SPLIT ? r * 32u + 16u * h : lane * 16u);
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(stream + off));
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
        if (SHAPE == PAIR32) {
            const unsigned char* p = table + (size_t)row_of(g0 + r, mask) * 64 + 16u * h;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 32);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        } else if (SHAPE == QUAD) {
            const uint32_t q = lane & 3u, qd = lane >> 2;
            const unsigned char* p0 = table + (size_t)row_of(g0 + 2u * qd, mask) * 64 + 16u * q;
            const unsigned char* p1 = table + (size_t)row_of(g0 + 2u * qd + 1u, mask) * 64 + 16u * q;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p0), b = *reinterpret_cast<const u32x4*>(p1);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        } else if (SHAPE == PAIRADJ) {
            const unsigned char* p = table + (size_t)row_of(g0 + (lane >> 1), mask) * 64 + 16u * (lane & 1u);
            const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 32);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int SHAPE, int STREAM>
static void run(const char* name, const unsigned char* tables, size_t table_bytes, int per_xcd, const unsigned char* stream,
                uint32_t* out) {
    const int blocks = 256 * 8, threads = 768;
    const uint32_t steps_per_wave = 64;                        // 2048 WG x 12 waves x 64 steps x 32 rows = 50.3 M rows
    const double rows = (double)blocks * 12 * steps_per_wave * 32;
    hipLaunchKernelGGL((k<SHAPE, STREAM>), dim3(blocks), dim3(threads), 0, 0, tables, table_bytes, per_xcd, stream, steps_per_wave, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((k<SHAPE, STREAM>), dim3(blocks), dim3(threads), 0, 0, tables, table_bytes, per_xcd, stream, steps_per_wave, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %s: %.3f ms for %.1f M rows = %.1f G rows/s\n", name, per_xcd ? "8 x 8 MB (one per XCD)" : "one 4-GiB table      ",
           ms, rows * 1e-6, rows / ms * 1e-6);
    fflush(stdout);
}


// Phased access: the rows of a step come from a 4-MB window of the XCD's table set; every wave switches window every
// `period` steps on its own clock (no barrier: the waves of an XCD drift apart as in a real kernel), `leak`/256 of the
// rows come from the sibling window (the other half of the same 8-MB table).  Models pass 1 walking a query's
// candidates in two code-range phases.
template <int SHAPE, int STREAM>
__global__ __launch_bounds__(768) void kphase(const unsigned char* __restrict__ tables, size_t xcd_bytes, uint32_t period,
                                              uint32_t leak, const unsigned char* __restrict__ stream,
                                              uint32_t steps_per_wave, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned char* base = tables + (size_t)(blockIdx.x & 7) * xcd_bytes;
    const uint32_t wrows = (4u << 20) / 64u, nwin = (uint32_t)(xcd_bytes / (4u << 20));
    uint32_t acc = 0;
    const uint32_t s0 = wave * steps_per_wave;
    // waves start a little out of step with each other (up to a quarter period)
    const uint32_t skew = (wave * 2654435761u >> 24) * period / 1024u;
#pragma unroll 4
    for (uint32_t s = s0; s < s0 + steps_per_wave; ++s) {
        const uint32_t g0 = s * 32u;
        const uint32_t win = ((s - s0 + skew) / period) % nwin;
        if (STREAM != NOSTREAM) {
            const size_t off = (size_t)s * 1024 + (STREAM == SPLIT ? r * 32u + 16u * h : lane * 16u);
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(stream + off));
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
        auto row_addr = [&](uint32_t g) -> const unsigned char* {
            uint32_t x = g * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            const uint32_t w = ((x >> 24) < leak) ? (win ^ 1u) : win;
            const uint32_t row = (uint32_t)(((uint64_t)(x * 40503u + 77u) * wrows) >> 32);
            return base + ((size_t)w * wrows + row) * 64;
        };
        if (SHAPE == PAIR32) {
            const unsigned char* p = row_addr(g0 + r) + 16u * h;
            const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 32);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        } else if (SHAPE == QUAD) {
            const uint32_t q = lane & 3u, qd = lane >> 2;
            const u32x4 a = *reinterpret_cast<const u32x4*>(row_addr(g0 + 2u * qd) + 16u * q);
            const u32x4 b = *reinterpret_cast<const u32x4*>(row_addr(g0 + 2u * qd + 1u) + 16u * q);
            acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int SHAPE, int STREAM>
static void run_phase(const char* name, const unsigned char* tables, size_t xcd_bytes, uint32_t period, uint32_t leak,
                      const unsigned char* stream, uint32_t* out) {
    const int blocks = 256, threads = 768;                     // one work-group per CU, as pass 1
    const uint32_t steps_per_wave = 512;                       // 256 x 12 x 512 x 32 = 50.3 M rows
    const double rows = (double)blocks * 12 * steps_per_wave * 32;
    hipLaunchKernelGGL((kphase<SHAPE, STREAM>), dim3(blocks), dim3(threads), 0, 0, tables, xcd_bytes, period, leak, stream, steps_per_wave, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((kphase<SHAPE, STREAM>), dim3(blocks), dim3(threads), 0, 0, tables, xcd_bytes, period, leak, stream, steps_per_wave, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-40s period %4u steps, leak %3u/256: %.3f ms for %.1f M rows\n", name, period, leak, ms, rows * 1e-6);
    fflush(stdout);
}

int main() {
    const size_t big = (size_t)4 << 30, small = (size_t)8 << 20;
    unsigned char *tb, *ts, *st; uint32_t* out;
    hipMalloc(&tb, big); hipMalloc(&ts, small * 8); hipMalloc(&out, 64);
    const size_t stream_bytes = (size_t)256 * 8 * 12 * 64 * 1024;      // 1 KB per step
    hipMalloc(&st, stream_bytes);
    hipMemset(tb, 1, big); hipMemset(ts, 1, small * 8); hipMemset(st, 2, stream_bytes);
    hipDeviceSynchronize();
    for (int per_xcd = 1; per_xcd >= 0; --per_xcd) {
        const unsigned char* t = per_xcd ? ts : tb;
        const size_t tbytes = per_xcd ? small : big;
        run<PAIR32, NOSTREAM>("gather PAIR32", t, tbytes, per_xcd, st, out);
        run<QUAD, NOSTREAM>("gather QUAD", t, tbytes, per_xcd, st, out);
        run<PAIRADJ, NOSTREAM>("gather PAIRADJ", t, tbytes, per_xcd, st, out);
        run<PAIR32, SPLIT>("gather PAIR32 + residual stream SPLIT", t, tbytes, per_xcd, st, out);
        run<PAIR32, CONTIG>("gather PAIR32 + residual stream CONTIG", t, tbytes, per_xcd, st, out);
        run<QUAD, SPLIT>("gather QUAD + residual stream SPLIT", t, tbytes, per_xcd, st, out);
        run<QUAD, CONTIG>("gather QUAD + residual stream CONTIG", t, tbytes, per_xcd, st, out);
        run<PAIRADJ, CONTIG>("gather PAIRADJ + residual stream CONTIG", t, tbytes, per_xcd, st, out);
    }
    // per-XCD table size sweep: how much does an L2-resident working set save (L2 = 4 MB per XCD)?
    for (size_t mb : {1, 2, 3, 4, 6, 8}) {
        char name[96];
        // row mask needs a power of two: 3 and 6 MB use the next power of two masked down by a multiply
        snprintf(name, sizeof name, "PAIR32 + stream SPLIT, %zu MB table / XCD", mb);
        run<PAIR32, SPLIT>(name, ts, mb << 20, 1, st, out);
        snprintf(name, sizeof name, "QUAD + stream CONTIG, %zu MB table / XCD", mb);
        run<QUAD, CONTIG>(name, ts, mb << 20, 1, st, out);
        snprintf(name, sizeof name, "PAIR32 alone, %zu MB table / XCD", mb);
        run<PAIR32, NOSTREAM>(name, ts, mb << 20, 1, st, out);
    }
    {   // phased walk over 8 x 64 MB (16 windows of 4 MB per XCD)
        unsigned char* tp; hipMalloc(&tp, (size_t)512 << 20); hipMemset(tp, 3, (size_t)512 << 20); hipDeviceSynchronize();
        for (uint32_t period : {64u, 16u, 1u})
            for (uint32_t leak : {0u, 33u, 128u}) {
                run_phase<PAIR32, SPLIT>("phased PAIR32 + stream SPLIT", tp, (size_t)64 << 20, period, leak, st, out);
                run_phase<QUAD, SPLIT>("phased QUAD + stream SPLIT", tp, (size_t)64 << 20, period, leak, st, out);
                run_phase<QUAD, CONTIG>("phased QUAD + stream CONTIG", tp, (size_t)64 << 20, period, leak, st, out);
            }
        run_phase<QUAD, NOSTREAM>("phased QUAD alone", tp, (size_t)64 << 20, 64, 33, st, out);
        run_phase<PAIR32, NOSTREAM>("phased PAIR32 alone", tp, (size_t)64 << 20, 64, 33, st, out);
    }
    run<QUAD, NOSTREAM>("QUAD alone 2 MB", ts, (size_t)2 << 20, 1, st, out);
    run<QUAD, SPLIT>("QUAD + stream SPLIT 4 MB", ts, (size_t)4 << 20, 1, st, out);
    run<PAIR32, CONTIG>("PAIR32 + stream CONTIG 4 MB", ts, (size_t)4 << 20, 1, st, out);
    run<NOGATHER, SPLIT>("residual stream SPLIT alone", ts, small, 1, st, out);
    run<NOGATHER, CONTIG>("residual stream CONTIG alone", ts, small, 1, st, out);
    return 0;
}
