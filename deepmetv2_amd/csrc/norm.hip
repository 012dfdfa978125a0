// norm.hip -- N3: BatchNorm1d over the nodes (+ the residual add that follows it in the model), gfx950.
//
// /root/reference/model/graph_met_network.py:32,39,58,66: `emb = bn_all(encode_all(..))` and, per convolution,
// `emb = emb + bn(conv(emb, ..))` with nn.BatchNorm1d(hidden_dim) over N ~ 3e5 nodes.  Stock torch runs 3 kernels
// forward (statistics, transform, add) and 3 backward at ~30-55 us each for [N,32]; all of them are plain HBM streams.
// Here: forward = column statistics (fixed row ranges per workgroup, partials combined in order in double: bitwise
// reproducible) + one streaming kernel y = (x - mean) * (gamma * invstd) + beta (+ residual); backward = the two
// column sums (sum g, sum g * xhat) + one streaming kernel.  H (channels) is a multiple of 4, at most 64.
#include "common.h"

namespace dmet {
namespace {

constexpr int kBnBlocks = 512;     // workgroups of the reduction kernels (fixed: the summation order must not depend on N)
constexpr int kBnThreads = 256;

// Column sums of up to two per-element quantities over a contiguous row range per workgroup.
//   MODE 0 (forward):  s0 = sum (x - shift),  s1 = sum (x - shift)^2     (shift = first row: tames cancellation)
//   MODE 1 (backward): s0 = sum g,            s1 = sum g * (x - mean) * invstd
// Thread t owns the float4 column group (t % (H/4)) and every (256 / (H/4))-th row of the range; the workgroup's
// partials are reduced through LDS in a fixed order and written to partial[block][2][H].
template <int MODE>
__global__ __launch_bounds__(kBnThreads) void bn_reduce_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                               int64_t N, int H, const float *__restrict__ stat_a,
                                                               const float *__restrict__ stat_b, int64_t rows_per_block,
                                                               float *__restrict__ partial)
{
    __shared__ float4 red0[kBnThreads], red1[kBnThreads];
    const int h4 = H / 4;
    const int rpb = kBnThreads / h4;               // rows per pass
    const int tid = threadIdx.x;
    const int c4 = tid % h4, rr = tid / h4;
    const int64_t lo = (int64_t)blockIdx.x * rows_per_block, hi = min(N, lo + rows_per_block);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    float4 sa = a, sb = make_float4(1.f, 1.f, 1.f, 1.f);
    if (rr < rpb) {
        if (MODE == 0) sa = reinterpret_cast<const float4 *>(x)[c4];                      // shift: row 0
        else { sa = reinterpret_cast<const float4 *>(stat_a)[c4]; sb = reinterpret_cast<const float4 *>(stat_b)[c4]; }
#pragma unroll 4   // four rows in flight per thread (the adds keep their order: same bits)
        for (int64_t i = lo + rr; i < hi; i += rpb) {
            const float4 v = reinterpret_cast<const float4 *>(x + i * H)[c4];
            if (MODE == 0) {
                const float dx = v.x - sa.x, dy = v.y - sa.y, dz = v.z - sa.z, dw = v.w - sa.w;
                a.x += dx; a.y += dy; a.z += dz; a.w += dw;
                b.x = __builtin_fmaf(dx, dx, b.x); b.y = __builtin_fmaf(dy, dy, b.y);
                b.z = __builtin_fmaf(dz, dz, b.z); b.w = __builtin_fmaf(dw, dw, b.w);
            } else {
                const float4 gv = reinterpret_cast<const float4 *>(g + i * H)[c4];
                a.x += gv.x; a.y += gv.y; a.z += gv.z; a.w += gv.w;
                b.x = __builtin_fmaf(gv.x, (v.x - sa.x) * sb.x, b.x); b.y = __builtin_fmaf(gv.y, (v.y - sa.y) * sb.y, b.y);
                b.z = __builtin_fmaf(gv.z, (v.z - sa.z) * sb.z, b.z); b.w = __builtin_fmaf(gv.w, (v.w - sa.w) * sb.w, b.w);
            }
        }
    }
    red0[tid] = a; red1[tid] = b;
    __syncthreads();
    if (tid < h4) {
        float4 s0 = red0[tid], s1 = red1[tid];
        for (int r = 1; r < rpb; ++r) {
            const float4 p = red0[r * h4 + tid], q = red1[r * h4 + tid];
            s0.x += p.x; s0.y += p.y; s0.z += p.z; s0.w += p.w;
            s1.x += q.x; s1.y += q.y; s1.z += q.z; s1.w += q.w;
        }
        float4 *out = reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * 2 * H);
        out[tid] = s0;
        out[h4 + tid] = s1;
    }
}

// Combine the workgroup partials of one column pair in a fixed order: 16 thread groups each add every 16th partial
// (in double), then the 16 group sums are added in order.  1024 threads, c = tid & 63, group = tid >> 6.
__device__ __forceinline__ void bn_combine(const float *__restrict__ partial, int nblocks, int H, double &s0, double &s1)
{
    __shared__ double g0[16][64], g1[16][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    if (c < H)
#pragma unroll 8   // independent loads: keep eight in flight (the sum order is unchanged)
        for (int blk = grp; blk < nblocks; blk += 16) {
            a += (double)partial[(int64_t)blk * 2 * H + c];
            b += (double)partial[(int64_t)blk * 2 * H + H + c];
        }
    g0[grp][c] = a; g1[grp][c] = b;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (grp == 0)
        for (int q = 0; q < 16; ++q) { s0 += g0[q][c]; s1 += g1[q][c]; }
}

// forward statistics: mean, invstd (biased variance), running statistics (unbiased variance, like torch)
__global__ __launch_bounds__(1024) void bn_fwd_finalize_kernel(const float *__restrict__ partial, int nblocks,
                                                             const float *__restrict__ x, int64_t N, int H, float eps,
                                                             float momentum, float *__restrict__ running_mean,
                                                             float *__restrict__ running_var,
                                                             float *__restrict__ save_mean,
                                                             float *__restrict__ save_invstd,
                                                             int64_t *__restrict__ num_batches_tracked)
{
    double s0, s1;
    bn_combine(partial, nblocks, H, s0, s1);
    const int c = threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;   // torch bumps it with a kernel of its own
    if (c >= H) return;                                    // threads >= 64 (other groups) leave here too
    const double shift = (double)x[c];
    const double m = s0 / (double)N;                       // mean of (x - shift)
    double var = s1 / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    const double mean = shift + m;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// eval mode: mean = running_mean, invstd = 1 / sqrt(running_var + eps)
__global__ __launch_bounds__(64) void bn_eval_stats_kernel(const float *__restrict__ running_mean,
                                                           const float *__restrict__ running_var, int H, float eps,
                                                           float *__restrict__ save_mean,
                                                           float *__restrict__ save_invstd)
{
    const int c = threadIdx.x;
    if (c >= H) return;
    save_mean[c] = running_mean[c];
    save_invstd[c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
}

// y = (x - mean) * (gamma * invstd) + beta (+ residual); with mean = running_mean, invstd from running_var in eval mode
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                       int64_t N, int H, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, float *__restrict__ y)
{
    const int h4 = H / 4;
    const int64_t total = N * h4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % h4);
        const float4 v = reinterpret_cast<const float4 *>(x)[t];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[c4], is = reinterpret_cast<const float4 *>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[c4], be = reinterpret_cast<const float4 *>(beta)[c4];
        float4 o;
        o.x = (v.x - mu.x) * (ga.x * is.x) + be.x; o.y = (v.y - mu.y) * (ga.y * is.y) + be.y;
        o.z = (v.z - mu.z) * (ga.z * is.z) + be.z; o.w = (v.w - mu.w) * (ga.w * is.w) + be.w;
        if (res) {
            const float4 r = reinterpret_cast<const float4 *>(res)[t];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4 *>(y)[t] = o;
    }
}

// backward sums -> g_gamma = sum g xhat, g_beta = sum g (written), and their means for the element kernel
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nblocks, int64_t N,
                                                             int H, float *__restrict__ g_gamma,
                                                             float *__restrict__ g_beta, float *__restrict__ mean_g,
                                                             float *__restrict__ mean_gx)
{
    double s0, s1;
    bn_combine(partial, nblocks, H, s0, s1);
    const int c = threadIdx.x;
    if (c >= H) return;
    g_beta[c] = (float)s0;
    g_gamma[c] = (float)s1;
    mean_g[c] = (float)(s0 / (double)N);
    mean_gx[c] = (float)(s1 / (double)N);
}

// g_x = gamma * invstd * (g - mean_g - xhat * mean_gx)      (training-mode batch statistics)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                           int64_t N, int H, const float *__restrict__ gamma,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const float *__restrict__ mean_g,
                                                           const float *__restrict__ mean_gx, float *__restrict__ gx)
{
    const int h4 = H / 4;
    const int64_t total = N * h4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % h4);
        const float4 v = reinterpret_cast<const float4 *>(x)[t], gv = reinterpret_cast<const float4 *>(g)[t];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[c4], is = reinterpret_cast<const float4 *>(invstd)[c4];
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[c4];
        const float4 mg = reinterpret_cast<const float4 *>(mean_g)[c4], mx = reinterpret_cast<const float4 *>(mean_gx)[c4];
        float4 o;
        o.x = ga.x * is.x * (gv.x - mg.x - (v.x - mu.x) * is.x * mx.x);
        o.y = ga.y * is.y * (gv.y - mg.y - (v.y - mu.y) * is.y * mx.y);
        o.z = ga.z * is.z * (gv.z - mg.z - (v.z - mu.z) * is.z * mx.z);
        o.w = ga.w * is.w * (gv.w - mg.w - (v.w - mu.w) * is.w * mx.w);
        reinterpret_cast<float4 *>(gx)[t] = o;
    }
}

inline int bn_blocks(int64_t N, int64_t *rows_per_block)
{
    int64_t rpb = (N + kBnBlocks - 1) / kBnBlocks;
    if (rpb < 1) rpb = 1;
    *rows_per_block = rpb;
    return (int)((N + rpb - 1) / rpb);
}

inline bool bn_shape_ok(int H) { return H >= 4 && H <= 64 && (H % 4) == 0; }

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_bn_workspace_bytes(int64_t N, int H)
{
    (void)N;
    if (!bn_shape_ok(H)) return 0;
    return sizeof(float) * ((size_t)kBnBlocks * 2 * H + 2 * (size_t)H) + 512;
}

extern "C" int dmet_bn_fwd_tracked_f32(const float *x, const float *residual, int64_t N, int H, const float *gamma,
                                       const float *beta, float eps, float momentum, float *running_mean,
                                       float *running_var, int64_t *num_batches_tracked, int training, float *y,
                                       float *save_mean, float *save_invstd, void *ws, size_t ws_bytes,
                                       dmet_stream_t stream)
{
    DMET_REQUIRE(bn_shape_ok(H), "dmet_bn_fwd_f32: H=%d must be a multiple of 4 in [4,64]", H);
    DMET_REQUIRE(N >= 0, "dmet_bn_fwd_f32: N=%lld", (long long)N);
    if (N == 0) return 0;
    DMET_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && ws, "dmet_bn_fwd_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && aligned16(save_mean) &&
                     aligned16(save_invstd) && (!residual || aligned16(residual)),
                 "dmet_bn_fwd_f32: pointers must be 16-byte aligned");
    DMET_REQUIRE(ws_bytes >= dmet_bn_workspace_bytes(N, H), "dmet_bn_fwd_f32: workspace too small");
    DMET_REQUIRE(training || (running_mean && running_var), "dmet_bn_fwd_f32: eval mode needs running statistics");
    DMET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dmet_bn_fwd_f32: running_mean/var go together");
    hipStream_t st = as_stream(stream);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    if (training) {
        int64_t rpb;
        const int nb = bn_blocks(N, &rpb);
        hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(nb), dim3(kBnThreads), 0, st, x, (const float *)nullptr, N, H,
                           (const float *)nullptr, (const float *)nullptr, rpb, partial);
        DMET_LAUNCH_CHECK("bn_reduce_kernel<0>");
        hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nb, x, N, H, eps, momentum,
                           running_mean, running_var, save_mean, save_invstd, num_batches_tracked);
        DMET_LAUNCH_CHECK("bn_fwd_finalize_kernel");
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(1), dim3(64), 0, st, (const float *)running_mean,
                           (const float *)running_var, H, eps, save_mean, save_invstd);
        DMET_LAUNCH_CHECK("bn_eval_stats_kernel");
    }
    const int64_t total = N * (H / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, residual, N, H, gamma, beta,
                       (const float *)save_mean, (const float *)save_invstd, y);
    DMET_LAUNCH_CHECK("bn_apply_kernel");
    return 0;
}

extern "C" int dmet_bn_stats_f32(const float *x, int64_t N, int H, float eps, float momentum, float *running_mean,
                                 float *running_var, int64_t *num_batches_tracked, float *save_mean, float *save_invstd,
                                 void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(bn_shape_ok(H), "dmet_bn_stats_f32: H=%d must be a multiple of 4 in [4,64]", H);
    DMET_REQUIRE(N > 0, "dmet_bn_stats_f32: N=%lld", (long long)N);
    DMET_REQUIRE(x && save_mean && save_invstd && ws, "dmet_bn_stats_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(save_mean) && aligned16(save_invstd), "dmet_bn_stats_f32: pointers must be 16-byte aligned");
    DMET_REQUIRE(ws_bytes >= dmet_bn_workspace_bytes(N, H), "dmet_bn_stats_f32: workspace too small");
    DMET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dmet_bn_stats_f32: running_mean/var go together");
    hipStream_t st = as_stream(stream);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    int64_t rpb;
    const int nb = bn_blocks(N, &rpb);
    hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(nb), dim3(kBnThreads), 0, st, x, (const float *)nullptr, N, H,
                       (const float *)nullptr, (const float *)nullptr, rpb, partial);
    DMET_LAUNCH_CHECK("bn_reduce_kernel<0>");
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nb, x, N, H, eps, momentum, running_mean,
                       running_var, save_mean, save_invstd, num_batches_tracked);
    DMET_LAUNCH_CHECK("bn_fwd_finalize_kernel");
    return 0;
}

extern "C" int dmet_bn_fwd_f32(const float *x, const float *residual, int64_t N, int H, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                               int training, float *y, float *save_mean, float *save_invstd, void *ws, size_t ws_bytes,
                               dmet_stream_t stream)
{
    return dmet_bn_fwd_tracked_f32(x, residual, N, H, gamma, beta, eps, momentum, running_mean, running_var, nullptr,
                                   training, y, save_mean, save_invstd, ws, ws_bytes, stream);
}

extern "C" int dmet_bn_bwd_f32(const float *x, const float *g_y, int64_t N, int H, const float *gamma,
                               const float *save_mean, const float *save_invstd, float *g_x, float *g_gamma,
                               float *g_beta, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(bn_shape_ok(H), "dmet_bn_bwd_f32: H=%d must be a multiple of 4 in [4,64]", H);
    DMET_REQUIRE(N > 0, "dmet_bn_bwd_f32: N=%lld", (long long)N);
    DMET_REQUIRE(x && g_y && gamma && save_mean && save_invstd && g_x && g_gamma && g_beta && ws,
                 "dmet_bn_bwd_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(g_y) && aligned16(g_x) && aligned16(gamma) && aligned16(save_mean) &&
                     aligned16(save_invstd),
                 "dmet_bn_bwd_f32: pointers must be 16-byte aligned");
    DMET_REQUIRE(ws_bytes >= dmet_bn_workspace_bytes(N, H), "dmet_bn_bwd_f32: workspace too small");
    hipStream_t st = as_stream(stream);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    float *mean_g = partial + (size_t)kBnBlocks * 2 * H, *mean_gx = mean_g + H;
    int64_t rpb;
    const int nb = bn_blocks(N, &rpb);
    hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(nb), dim3(kBnThreads), 0, st, x, g_y, N, H, save_mean, save_invstd,
                       rpb, partial);
    DMET_LAUNCH_CHECK("bn_reduce_kernel<1>");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nb, N, H, g_gamma, g_beta, mean_g,
                       mean_gx);
    DMET_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    const int64_t total = N * (H / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, g_y, N, H, gamma, save_mean,
                       save_invstd, (const float *)mean_g, (const float *)mean_gx, g_x);
    DMET_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return 0;
}

extern "C" int dmet_bn_bwd_stats_f32(const float *x, const float *g_y, int64_t N, int H, const float *save_mean,
                                     const float *save_invstd, float *g_gamma, float *g_beta, float *mean_g,
                                     float *mean_gx, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(bn_shape_ok(H), "dmet_bn_bwd_stats_f32: H=%d must be a multiple of 4 in [4,64]", H);
    DMET_REQUIRE(N > 0, "dmet_bn_bwd_stats_f32: N=%lld", (long long)N);
    DMET_REQUIRE(x && g_y && save_mean && save_invstd && g_gamma && g_beta && mean_g && mean_gx && ws,
                 "dmet_bn_bwd_stats_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(g_y) && aligned16(save_mean) && aligned16(save_invstd),
                 "dmet_bn_bwd_stats_f32: pointers must be 16-byte aligned");
    DMET_REQUIRE(ws_bytes >= dmet_bn_workspace_bytes(N, H), "dmet_bn_bwd_stats_f32: workspace too small");
    hipStream_t st = as_stream(stream);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    int64_t rpb;
    const int nb = bn_blocks(N, &rpb);
    hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(nb), dim3(kBnThreads), 0, st, x, g_y, N, H, save_mean, save_invstd,
                       rpb, partial);
    DMET_LAUNCH_CHECK("bn_reduce_kernel<1>");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nb, N, H, g_gamma, g_beta, mean_g,
                       mean_gx);
    DMET_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    return 0;
}

extern "C" int dmet_bn_apply_f32(const float *x, const float *residual, int64_t N, int H, const float *gamma,
                                 const float *beta, const float *mean, const float *invstd, float *y,
                                 dmet_stream_t stream)
{
    DMET_REQUIRE(bn_shape_ok(H), "dmet_bn_apply_f32: H=%d must be a multiple of 4 in [4,64]", H);
    DMET_REQUIRE(N >= 0, "dmet_bn_apply_f32: N=%lld", (long long)N);
    if (N == 0) return 0;
    DMET_REQUIRE(x && gamma && beta && mean && invstd && y, "dmet_bn_apply_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && aligned16(mean) &&
                     aligned16(invstd) && (!residual || aligned16(residual)),
                 "dmet_bn_apply_f32: pointers must be 16-byte aligned");
    const int64_t total = N * (H / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, residual, N, H, gamma,
                       beta, mean, invstd, y);
    DMET_LAUNCH_CHECK("bn_apply_kernel");
    return 0;
}

extern "C" int dmet_bn_eval_stats_f32(const float *running_mean, const float *running_var, int H, float eps,
                                      float *save_mean, float *save_invstd, dmet_stream_t stream)
{
    DMET_REQUIRE(bn_shape_ok(H), "dmet_bn_eval_stats_f32: H=%d must be a multiple of 4 in [4,64]", H);
    DMET_REQUIRE(running_mean && running_var && save_mean && save_invstd, "dmet_bn_eval_stats_f32: null pointer");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(1), dim3(64), 0, as_stream(stream), running_mean, running_var, H, eps,
                       save_mean, save_invstd);
    DMET_LAUNCH_CHECK("bn_eval_stats_kernel");
    return 0;
}
