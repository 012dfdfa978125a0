// finalize.hip -- the weight-gradient sums of a whole backward pass in ONE launch (gfx950).
//
// /root/reference/train.py:51-52: `loss.backward(); optimizer.step()`.  The three per-node backward kernels that carry
// weights (EdgeConv dense layer x conv_depth, encoder, output head) each end in a one-to-98-workgroup launch that adds the
// per-workgroup partials of the weight gradients in a fixed order: four dependent launches of ~5-7 us per step at
// BASELINE configs[1] whose results only the optimizer reads.  Between dmet_finalize_defer_begin() and
// dmet_finalize_flush() (any thread of the process) those entry points queue a descriptor instead; the flush forms every queued sum in
// one launch, each with the additions of its stand-alone kernel in the same order (same bits).  The partial buffers live
// in the callers' workspaces: they must stay untouched until the flush has run on the stream.
#include "common.h"

#include <mutex>

namespace dmet {
namespace {

constexpr int kDeferMax = 8;

struct DeferState {
    bool active = false;
    int n = 0;
    DeferDesc q[kDeferMax];
};
// process-wide, not per thread: loss.backward() runs the queued calls on the autograd engine's device thread, the harness
// begins and flushes on the thread that called it
DeferState g_defer;
std::mutex g_defer_mutex;

struct FinalizeArgs {
    int n;
    int first_block[kDeferMax + 1];
    DeferDesc d[kDeferMax];
};

__host__ __device__ constexpr int defer_blocks(int kind)
{
    return kind == kDeferEdgeConv ? 33 : kind == kDeferEncoder ? kEncPartialFloats / 32 : kHeadPartialFloats / 32;
}

// A block owns 32 consecutive elements of one queued gradient set; its 32 thread groups each add every 32nd partial
// (independent loads in flight), then group 0 adds the 32 group sums in order and routes the element to its parameter
// gradient -- statement for statement what edgeconv_linear_bwd_finalize_kernel, encode_bwd_finalize_kernel and
// head_bwd_finalize_kernel do (tests/test_gpu_parity.py compares the bits).
__global__ __launch_bounds__(1024) void finalize_groups_kernel(const FinalizeArgs a)
{
    __shared__ float red0[32][33], red1[32][33];
    // the block's descriptor, picked with static indices (no dynamically indexed copy of the argument array).  The launch
    // takes ~11 us either way: 198 workgroups reading ~17 MB of partials that were written up to a millisecond earlier
    DeferDesc d = a.d[0];
    int first = 0;
#pragma unroll
    for (int i = 1; i < kDeferMax; ++i)
        if (i < a.n && (int)blockIdx.x >= a.first_block[i]) { d = a.d[i]; first = a.first_block[i]; }
    const int blk = (int)blockIdx.x - first;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const float *__restrict__ partial = d.partial;
    const int64_t nparts = d.nparts;
    float s0 = 0.0f, s1 = 0.0f;
    if (d.kind == kDeferEdgeConv) {
        const bool bias = blk == 32;
        const int idx = bias ? (2048 + e) : (blk * 32 + e);
        if (bias) {
#pragma unroll 16
            for (int64_t w = grp; w < nparts; w += 32) s0 += partial[w * kEcbPartialFloats + idx];
        } else {
#pragma unroll 16
            for (int64_t w = grp; w < nparts; w += 32) {
                s0 += partial[w * kEcbPartialFloats + idx];
                s1 += partial[w * kEcbPartialFloats + 1024 + idx];
            }
        }
        red0[grp][e] = s0; red1[grp][e] = s1;
        __syncthreads();
        if (grp != 0) return;
        s0 = 0.0f; s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 32; ++q) { s0 += red0[q][e]; s1 += red1[q][e]; }
        float *gW = d.out[0], *gb = d.out[1];
        if (bias) {
            if (gb) gb[e] = s0;
        } else {
            gW[(idx >> 5) * 64 + (idx & 31)] = s0;
            gW[(idx >> 5) * 64 + 32 + (idx & 31)] = s1 - s0;
        }
        return;
    }
    const int stride = d.kind == kDeferEncoder ? kEncPartialFloats : kHeadPartialFloats;
    const int idx = blk * 32 + e;
#pragma unroll 8
    for (int64_t w = grp; w < nparts; w += 32) s0 += partial[w * stride + idx];
    red0[grp][e] = s0;
    __syncthreads();
    if (grp != 0) return;
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red0[q][e];
    if (d.kind == kDeferEncoder) {
        float *Wc = d.out[0], *bc = d.out[1], *Wk = d.out[2], *bk = d.out[3], *Wa = d.out[4], *ba = d.out[5];
        float *Echg = d.out[6], *Epdg = d.out[7], *Epv = d.out[8];
        const int t = idx / 1024, r = (idx % 1024) / 32, c = idx % 32;
        if (t == 0) Wa[r * 32 + c] = s;
        else if (t == 1) {
            if (r < 16 && c < 24) Wk[r * 24 + c] = s;
            else if (r >= 16 && c >= 24) Wc[(r - 16) * 8 + (c - 24)] = s;
        } else if (t == 2) {
            if (r < 3 && c < 8) Echg[r * 8 + c] = s;
            else if (r >= 3 && r < 10 && c >= 8 && c < 16) Epdg[(r - 3) * 8 + (c - 8)] = s;
            else if (r >= 10 && r < 18 && c >= 16 && c < 24) Epv[(r - 10) * 8 + (c - 16)] = s;
        } else if (r == 0) ba[c] = s;
        else if (c < 16) bk[c] = s;
        else bc[c - 16] = s;
    } else {
        float *gW1 = d.out[0], *gb1 = d.out[1], *gW2 = d.out[2], *gb2 = d.out[3];
        if (idx < 1024) {
            const int r = idx >> 5, c = idx & 31;
            if (r < 16) gW1[r * 32 + c] = s;
        } else {
            const int l = idx - 1024;
            if (l < 16) gW2[l] = s;
            else if (l == 16) gb2[0] = s;
            else if (l >= 32 && l < 32 + 16) gb1[l - 32] = s;
        }
    }
}

}  // namespace

bool defer_push(const DeferDesc &d)
{
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    if (!g_defer.active || g_defer.n >= kDeferMax) return false;
    g_defer.q[g_defer.n++] = d;
    return true;
}

}  // namespace dmet

using namespace dmet;

extern "C" int dmet_finalize_defer_begin(void)
{
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    // sums still queued belong to gradients that nobody has formed yet: dropping them silently would leave those gradients
    // unwritten for good
    DMET_REQUIRE(!(g_defer.active && g_defer.n > 0),
                 "dmet_finalize_defer_begin: %d weight-gradient sums of the previous deferral are still queued (dmet_finalize_flush first)",
                 g_defer.n);
    g_defer.active = true;
    g_defer.n = 0;
    return 0;
}

extern "C" int dmet_finalize_pending(void)
{
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    return g_defer.active ? g_defer.n : -1;
}

extern "C" int dmet_finalize_flush(dmet_stream_t stream)
{
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    const int n = g_defer.active ? g_defer.n : 0;
    g_defer.active = false;
    g_defer.n = 0;
    if (n == 0) return 0;
    FinalizeArgs a;
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        a.first_block[i] = blocks;
        a.d[i] = g_defer.q[i];
        blocks += defer_blocks(g_defer.q[i].kind);
    }
    for (int i = n; i <= kDeferMax; ++i) a.first_block[i] = blocks;
    hipLaunchKernelGGL(finalize_groups_kernel, dim3((unsigned)blocks), dim3(1024), 0, as_stream(stream), a);
    DMET_LAUNCH_CHECK("finalize_groups_kernel");
    return 0;
}
