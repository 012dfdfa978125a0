// common.h -- shared helpers of libdmet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dmet.h"

namespace dmet {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumXcd = 8;       // MI355X: 8 XCDs, blocks are dealt round-robin over them
constexpr float kKnnSentinel = 1e10f;  // upstream torch_cluster initial best distance

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define DMET_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::dmet::set_error(__VA_ARGS__);     \
            return -22;                         \
        }                                       \
    } while (0)

#define DMET_LAUNCH_CHECK(name)                                   \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return ::dmet::hip_fail(e__, name); \
    } while (0)

static inline hipStream_t as_stream(dmet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Experiment / A-B switches are read on EVERY call (never cached in a static): entry points that share a switch then
// always agree, also when a test changes the variable inside one process.
static inline bool env_is(const char *name, const char *value)
{
    const char *e = getenv(name);
    return e && strcmp(e, value) == 0;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Event that owns node i: the b with ptr[b] <= i < ptr[b+1] (empty events are skipped naturally).
__device__ __forceinline__ int find_event(const int64_t *__restrict__ ptr, int B, int64_t i)
{
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): remap so that the workgroups of one XCD
// cover one contiguous chunk of the grid (rows an event shares then stay in one L2).
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk)
{
    // bijective remap: blocks sharing an XCD (bid % 8) get one contiguous chunk of the grid
    const int q = nblk / kNumXcd, rm = nblk % kNumXcd;
    const int xcd = bid % kNumXcd, idx = bid / kNumXcd;
    const int base = (xcd < rm) ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q;
    return base + idx;
}

// ---- deferred weight-gradient sums (csrc/finalize.hip) -----------------------------------------------------------------
// The backward kernels of the EdgeConv's dense layer, the encoder and the output head leave per-workgroup partial sums
// of their weight gradients and a small second launch adds them up.  Nothing but the optimizer reads those sums, so a
// caller may ask for all of them to be formed by ONE launch at the end of the backward pass
// (dmet_finalize_defer_begin / dmet_finalize_flush): the entry points then queue a descriptor instead of launching.
constexpr int kEcbPartialFloats = 2 * 1024 + 32;    // csrc/edgeconv_bwd.hip: gP^T x | gQ^T x | column sums of gP
constexpr int kEncPartialFloats = 3 * 1024 + 64;    // csrc/encoder.hip: three 32 x 32 tiles | dba | dbk | dbc
constexpr int kHeadPartialFloats = 1024 + 64;       // csrc/head.hip: A^T emb tile | gW2 | gb2 | gb1
enum { kDeferEdgeConv = 0, kDeferEncoder = 1, kDeferHead = 2 };
struct DeferDesc {
    int kind;
    const float *partial;   // [nparts][k*PartialFloats]
    int64_t nparts;
    float *out[9];          // edgeconv: gW, gb (may be null) | encoder: Wc bc Wk bk Wa ba Echg Epdg Epv | head: gW1 gb1 gW2 gb2
};
bool defer_push(const DeferDesc &d);   // true: queued, the caller launches no finalize step of its own

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace dmet
