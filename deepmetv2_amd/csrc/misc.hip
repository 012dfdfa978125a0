// misc.hip -- error plumbing, K4 (per-event MET reduction), batch->ptr, reverse index (gfx950).
#include <stdarg.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

namespace dmet {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return -(1000 + (int)e);
}

namespace {

// ---- K4: one workgroup per event; lane-strided partials -> wavefront butterfly -> waves summed in order ------
// 1024 threads with four rows in flight each: an event of 4 500 nodes is two batches of loads per thread instead of the 18
// dependent round trips of a 256-thread loop (11.9 -> 5.5 us at 64 x 4500; the sum is still a fixed-shape tree)
constexpr int kMetThreads = 1024;

template <int NV>
__global__ __launch_bounds__(kMetThreads) void event_sum_kernel(const float *__restrict__ w,
                                                                 const float *__restrict__ x, int64_t x_stride,
                                                                 const int64_t *__restrict__ ptr,
                                                                 float *__restrict__ out)
{
    __shared__ float part[NV][kMetThreads / kWave];
    const int b = blockIdx.x;
    const int64_t lo = ptr[b], hi = ptr[b + 1];
    float s[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = 0.0f;
#pragma unroll 4   // independent loads in flight; the adds keep their order
    for (int64_t i = lo + threadIdx.x; i < hi; i += kMetThreads) {
        if (NV == 2) {
            const float wi = w[i];
            s[0] += wi * x[i * x_stride + 0];
            if (NV > 1) s[NV - 1] += wi * x[i * x_stride + 1];
        } else {
            s[0] += w[i];
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = wave_sum(s[v]);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) part[v][wv] = s[v];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float t = 0.0f;
            for (int q = 0; q < kMetThreads / kWave; ++q) t += part[v][q];
            out[(int64_t)b * NV + v] = t;
        }
    }
}

// scale (optional, one float on the device): g_met is multiplied by it first, rounded to fp32 like the separate
// `g_met * g_loss` of the autograd chain it replaces
__global__ __launch_bounds__(256) void met_bwd_kernel(const float *__restrict__ g_met, const float *__restrict__ scale,
                                                       const float *__restrict__ x, int64_t x_stride,
                                                       const int64_t *__restrict__ ptr, int B, int64_t N,
                                                       float *__restrict__ g_w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int b = find_event(ptr, B, i);
    float gx = g_met[2 * b], gy = g_met[2 * b + 1];
    if (scale) { const float sc = scale[0]; gx = gx * sc; gy = gy * sc; }
    g_w[i] = gx * x[i * x_stride] + gy * x[i * x_stride + 1];
}

__global__ __launch_bounds__(256) void batch_to_ptr_kernel(const int64_t *__restrict__ batch, int64_t N, int B,
                                                            int64_t *__restrict__ ptr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    // ptr[b] = first i with batch[i] >= b ; written by the thread sitting on each boundary
    const int64_t prev = (i == 0) ? -1 : batch[i - 1];
    const int64_t cur = (i == N) ? (int64_t)B : batch[i];
    for (int64_t b = prev + 1; b <= cur && b <= B; ++b) ptr[b] = i;
}

// rev_ptr from the sorted source keys: rev_ptr[j] = first sorted position whose key >= j
__global__ __launch_bounds__(256) void rev_ptr_kernel(const uint32_t *__restrict__ keys_sorted, int64_t E,
                                                       int64_t N, uint32_t mask, int32_t *__restrict__ rev_ptr)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > E) return;
    // keys >= N are the "-1 = no neighbour" entries (all ones under the key mask); clamp them to N
    int64_t prev = (p == 0) ? -1 : (int64_t)(keys_sorted[p - 1] & mask);
    int64_t cur = (p == E) ? N : (int64_t)(keys_sorted[p] & mask);
    if (prev > N) prev = N;
    if (cur > N) cur = N;
    for (int64_t j = prev + 1; j <= cur; ++j) rev_ptr[j] = (int32_t)p;
}

inline unsigned bit_length(uint64_t v)
{
    unsigned n = 0;
    while (v) { ++n; v >>= 1; }
    return n;
}

// ---- H1: the optimizer step of train.py:52,75 (torch.optim.AdamW) on ONE flat fp32 parameter tensor ----------------
// One workgroup walks the tensor (the model has 6 641 parameters: a grid would need a second launch, or a grid barrier,
// for the step state).  The step state lives on the device so that the launch replays inside a hipGraph: step[0] (float,
// like torch's capturable state) and bias_pow[2] = beta1^t, beta2^t kept as running products in double (a double pow()
// per launch cost 3 us of one thread's time; t multiplications lose t x 2^-53).  Thread 0 advances the state and hands the
// bias corrections to the others through LDS while their first elements are already in flight.
__global__ __launch_bounds__(1024) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                     float *__restrict__ m, float *__restrict__ v,
                                                     float *__restrict__ step, double *__restrict__ bias_pow, int64_t n,
                                                     double lr, double beta1, double beta2, double eps, double weight_decay,
                                                     const double *__restrict__ lr_dev)
{
    // the learning rate may live on the device (a scheduler such as train.py:76's ReduceLROnPlateau changes it between
    // steps; a launch replayed from a hipGraph must see the new value): a wave-uniform (scalar) load
    if (lr_dev) lr = lr_dev[0];
    __shared__ float sh[2];
    int64_t i = threadIdx.x;
    float gi = 0.f, pi = 0.f, mi = 0.f, vi = 0.f;
    if (i < n) { gi = g[i]; pi = p[i]; mi = m[i]; vi = v[i]; }
    if (threadIdx.x == 0) {
        step[0] = step[0] + 1.0f;
        const double p1 = bias_pow[0] * beta1, p2 = bias_pow[1] * beta2;
        bias_pow[0] = p1; bias_pow[1] = p2;
        sh[0] = (float)(lr / (1.0 - p1));
        sh[1] = (float)(1.0 / sqrt(1.0 - p2));
    }
    __syncthreads();
    const float step_size = sh[0], inv_sqrt_bc2 = sh[1];
    // hyper-parameters arrive as doubles (python floats): 1 - beta2 formed in fp32 would be off by 5e-5 of itself
    const float decay = (float)(1.0 - lr * weight_decay), omb1 = (float)(1.0 - beta1), b2 = (float)beta2,
                omb2 = (float)(1.0 - beta2), epsf = (float)eps;
    while (i < n) {
        const int64_t nx = i + blockDim.x;
        float gn = 0.f, pn = 0.f, mn = 0.f, vn = 0.f;
        if (nx < n) { gn = g[nx]; pn = p[nx]; mn = m[nx]; vn = v[nx]; }
        pi *= decay;                                               // decoupled weight decay
        mi = mi + (gi - mi) * omb1;                                // lerp, like torch's fused kernel
        vi = b2 * vi + omb2 * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + epsf;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        i = nx; gi = gn; pi = pn; mi = mn; vi = vn;
    }
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" int dmet_version(void) { return DMET_VERSION; }
extern "C" const char *dmet_last_error(void) { return g_err; }

extern "C" int dmet_device_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0 ? 1 : 0;
}

extern "C" int dmet_met_reduce_f32(const float *w, const float *x, int64_t x_stride, const int64_t *ptr, int B,
                                   float *met, dmet_stream_t stream)
{
    DMET_REQUIRE(B >= 0 && x_stride >= 2, "dmet_met_reduce_f32: bad sizes (B=%d, x_stride=%lld)", B, (long long)x_stride);
    if (B == 0) return 0;
    DMET_REQUIRE(w && x && ptr && met, "dmet_met_reduce_f32: null pointer");
    hipLaunchKernelGGL((event_sum_kernel<2>), dim3((unsigned)B), dim3(kMetThreads), 0, as_stream(stream), w, x,
                       x_stride, ptr, met);
    DMET_LAUNCH_CHECK("met_reduce_kernel");
    return 0;
}

extern "C" int dmet_segment_sum_1d_f32(const float *src, const int64_t *ptr, int B, float *out, dmet_stream_t stream)
{
    DMET_REQUIRE(B >= 0, "dmet_segment_sum_1d_f32: B<0");
    if (B == 0) return 0;
    DMET_REQUIRE(src && ptr && out, "dmet_segment_sum_1d_f32: null pointer");
    hipLaunchKernelGGL((event_sum_kernel<1>), dim3((unsigned)B), dim3(kMetThreads), 0, as_stream(stream), src,
                       (const float *)nullptr, (int64_t)0, ptr, out);
    DMET_LAUNCH_CHECK("segment_sum_1d_kernel");
    return 0;
}

// loss = 0.5 * mean_b((met_x + true_x)^2 + (met_y + true_y)^2)  (model/net.py:58-61) and d loss / d met in one launch;
// one workgroup, fixed summation order.
__global__ __launch_bounds__(256) void met_loss_kernel(const float *__restrict__ met, const float *__restrict__ truth,
                                                        int64_t truth_stride, int B, float *__restrict__ loss,
                                                        float *__restrict__ g_met)
{
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const float invB = 1.0f / (float)B;
    float s = 0.0f;
    for (int b = tid; b < B; b += 256) {
        const float rx = met[2 * b] + truth[b * truth_stride], ry = met[2 * b + 1] + truth[b * truth_stride + 1];
        s += rx * rx + ry * ry;
        g_met[2 * b] = rx * invB;
        g_met[2 * b + 1] = ry * invB;
    }
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) loss[0] = 0.5f * red[0] * invB;
}

extern "C" int dmet_met_loss_strided_f32(const float *met, const float *truth, int64_t truth_stride, int B, float *loss,
                                         float *g_met, dmet_stream_t stream)
{
    DMET_REQUIRE(B > 0 && truth_stride >= 2, "dmet_met_loss_f32: B=%d truth_stride=%lld", B, (long long)truth_stride);
    DMET_REQUIRE(met && truth && loss && g_met, "dmet_met_loss_f32: null pointer");
    hipLaunchKernelGGL(met_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), met, truth, truth_stride, B, loss, g_met);
    DMET_LAUNCH_CHECK("met_loss_kernel");
    return 0;
}

extern "C" int dmet_met_loss_f32(const float *met, const float *truth, int B, float *loss, float *g_met,
                                 dmet_stream_t stream)
{
    return dmet_met_loss_strided_f32(met, truth, 2, B, loss, g_met, stream);
}

extern "C" int dmet_met_reduce_bwd_scaled_f32(const float *g_met, const float *scale, const float *x, int64_t x_stride,
                                              const int64_t *ptr, int B, int64_t N, float *g_w, dmet_stream_t stream)
{
    DMET_REQUIRE(B >= 0 && N >= 0 && x_stride >= 2, "dmet_met_reduce_bwd_f32: bad sizes");
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(g_met && x && ptr && g_w, "dmet_met_reduce_bwd_f32: null pointer");
    hipLaunchKernelGGL(met_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, as_stream(stream), g_met, scale, x,
                       x_stride, ptr, B, N, g_w);
    DMET_LAUNCH_CHECK("met_bwd_kernel");
    return 0;
}

extern "C" int dmet_met_reduce_bwd_f32(const float *g_met, const float *x, int64_t x_stride, const int64_t *ptr,
                                       int B, int64_t N, float *g_w, dmet_stream_t stream)
{
    return dmet_met_reduce_bwd_scaled_f32(g_met, nullptr, x, x_stride, ptr, B, N, g_w, stream);
}

extern "C" int dmet_adamw_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step,
                              double *bias_pow, int64_t n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, dmet_stream_t stream)
{
    DMET_REQUIRE(n >= 0, "dmet_adamw_f32: n=%lld", (long long)n);
    if (n == 0) return 0;
    DMET_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && bias_pow, "dmet_adamw_f32: null pointer");
    hipLaunchKernelGGL(adamw_kernel, dim3(1), dim3(1024), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, step,
                       bias_pow, n, lr, beta1, beta2, eps, weight_decay, (const double *)nullptr);
    DMET_LAUNCH_CHECK("adamw_kernel");
    return 0;
}

extern "C" int dmet_adamw_lr_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step,
                                 double *bias_pow, int64_t n, const double *lr_dev, double beta1, double beta2, double eps,
                                 double weight_decay, dmet_stream_t stream)
{
    DMET_REQUIRE(n >= 0, "dmet_adamw_lr_f32: n=%lld", (long long)n);
    if (n == 0) return 0;
    DMET_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && bias_pow && lr_dev, "dmet_adamw_lr_f32: null pointer");
    hipLaunchKernelGGL(adamw_kernel, dim3(1), dim3(1024), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, step,
                       bias_pow, n, 0.0, beta1, beta2, eps, weight_decay, lr_dev);
    DMET_LAUNCH_CHECK("adamw_kernel");
    return 0;
}

extern "C" int dmet_batch_to_ptr(const int64_t *batch, int64_t N, int B, int64_t *ptr, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && B >= 0, "dmet_batch_to_ptr: bad sizes");
    DMET_REQUIRE(ptr && (batch || N == 0), "dmet_batch_to_ptr: null pointer");
    hipLaunchKernelGGL(batch_to_ptr_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       batch, N, B, ptr);
    DMET_LAUNCH_CHECK("batch_to_ptr_kernel");
    return 0;
}

// ---- reverse index: stable radix sort of the table positions by the source id they hold (rocPRIM) ---------
static size_t rocprim_sort_bytes(int64_t E, unsigned end_bit)
{
    size_t bytes = 0;
    uint32_t *kin = nullptr, *kout = nullptr;
    int32_t *vout = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, rocprim::counting_iterator<int32_t>(0), vout,
                                             (size_t)E, 0u, end_bit, (hipStream_t)0);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return bytes;
}

extern "C" size_t dmet_reverse_index_workspace_bytes(int64_t M, int64_t num_keys)
{
    if (M <= 0 || num_keys <= 0) return 0;
    size_t sort_bytes = rocprim_sort_bytes(M, bit_length((uint64_t)num_keys));
    if (sort_bytes == 0) sort_bytes = (size_t)M * 16 + (1u << 20);  // conservative when no device is visible
    return (size_t)M * sizeof(uint32_t) + sort_bytes + 1024;
}

extern "C" int dmet_reverse_index(const int32_t *keys, int64_t M, int64_t num_keys, int32_t *rev_ptr,
                                  int32_t *rev_pos, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(M >= 0 && num_keys >= 0, "dmet_reverse_index: bad sizes");
    DMET_REQUIRE(M < (int64_t)2147483647 && num_keys < (int64_t)2147483647, "dmet_reverse_index: sizes exceed int32");
    DMET_REQUIRE(rev_ptr, "dmet_reverse_index: null pointer");
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        hipError_t e = hipMemsetAsync(rev_ptr, 0, sizeof(int32_t) * (size_t)(num_keys + 1), st);
        if (e != hipSuccess) return hip_fail(e, "dmet_reverse_index memset");
        return 0;
    }
    DMET_REQUIRE(keys && rev_pos && ws, "dmet_reverse_index: null pointer");
    const unsigned end_bit = bit_length((uint64_t)num_keys);  // -1 keeps all ones under the mask and sorts last
    const uint32_t mask = (end_bit >= 32) ? 0xFFFFFFFFu : ((1u << end_bit) - 1u);
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u;
    uint32_t *keys_sorted = reinterpret_cast<uint32_t *>(base);
    void *tmp = reinterpret_cast<void *>((base + (size_t)M * sizeof(uint32_t) + 255u) & ~(uintptr_t)255u);
    size_t need = rocprim_sort_bytes(M, end_bit);
    const size_t used = (reinterpret_cast<uintptr_t>(tmp) - reinterpret_cast<uintptr_t>(ws)) + need;
    DMET_REQUIRE(need > 0 && used <= ws_bytes, "dmet_reverse_index: workspace too small (%zu needed, %zu given)", used,
                 ws_bytes);
    hipError_t e = rocprim::radix_sort_pairs(tmp, need, reinterpret_cast<const uint32_t *>(keys), keys_sorted,
                                             rocprim::counting_iterator<int32_t>(0), rev_pos, (size_t)M, 0u, end_bit,
                                             st);
    if (e != hipSuccess) return hip_fail(e, "rocprim::radix_sort_pairs");
    hipLaunchKernelGGL(rev_ptr_kernel, dim3((unsigned)((M + 1 + 255) / 256)), dim3(256), 0, st, keys_sorted, M,
                       num_keys, mask, rev_ptr);
    DMET_LAUNCH_CHECK("rev_ptr_kernel");
    return 0;
}

// ---- neighbour table -> edge list (N1/N2: what knn_graph / radius_graph hand back to the caller) ---------------------
// 8 lanes per row (a wavefront handles 8 rows): reads of 32 contiguous bytes per row and step instead of one row per
// lane (64 cache lines per load instruction).  deg[i] = number of valid (>= 0) entries among the first cnt[i] (or all
// k) slots of row i.
__global__ __launch_bounds__(256) void table_degree_kernel(const int32_t *__restrict__ nbr,
                                                           const int32_t *__restrict__ cnt, int64_t N, int k,
                                                           int32_t *__restrict__ deg)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 3;
    const int l = (int)(t & 7);
    const bool live = i < N;
    const int64_t ii = live ? i : N - 1;
    const int m = live ? (cnt ? min(k, cnt[ii]) : k) : 0;
    const int32_t *row = nbr + ii * k;
    int d = 0;
    for (int s0 = 0; __any(s0 < m); s0 += 8) {
        const int s = s0 + l;
        d += (s < m && row[s] >= 0) ? 1 : 0;
    }
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
    if (live && l == 0) deg[i] = d;
}

// Edge e = rowptr[i] + (rank of slot s among the valid slots of row i): first[e] / second[e] = (source, target) of the
// edge as int64 (flow source_to_target) or swapped; src32 / tgt32 (optional) = the same as int32.
__global__ __launch_bounds__(256) void table_edges_kernel(const int32_t *__restrict__ nbr,
                                                          const int32_t *__restrict__ cnt,
                                                          const int32_t *__restrict__ rowptr, int64_t N, int k,
                                                          int swap, int64_t *__restrict__ first,
                                                          int64_t *__restrict__ second, int32_t *__restrict__ src32,
                                                          int32_t *__restrict__ tgt32)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 3;
    const int l = (int)(t & 7);
    const int lane = threadIdx.x & 63;
    const bool live = i < N;
    const int64_t ii = live ? i : N - 1;
    const int m = live ? (cnt ? min(k, cnt[ii]) : k) : 0;
    const int32_t *row = nbr + ii * k;
    int64_t e = rowptr[ii];
    // a row the caller sized at the table's full width is copied slot for slot: a table ASSUMED to have no empty slot
    // (kNN with self loops over events of at least k nodes: no edge count is fetched from the device) whose row is short
    // after all (a non-finite query) hands out -1 there -- a defined, loudly invalid index, never uninitialised memory
    const bool verbatim = live && (rowptr[ii + 1] - e) == k;
    for (int s0 = 0; __any(s0 < m); s0 += 8) {
        const int s = s0 + l;
        const int32_t j = (s < m) ? row[s] : -1;
        const unsigned long long ball = __ballot(j >= 0 || (verbatim && s < m));
        const unsigned grp = (unsigned)(ball >> (lane & ~7)) & 0xffu;      // the row's 8 lanes
        if (j >= 0 || (verbatim && s < m)) {
            const int64_t pos = e + __popc(grp & ((1u << l) - 1u));
            if (first) { first[pos] = swap ? ii : (int64_t)j; second[pos] = swap ? (int64_t)j : ii; }
            if (src32) { src32[pos] = j; tgt32[pos] = (int32_t)ii; }
        }
        e += __popc(grp);
    }
}

extern "C" int dmet_table_degree(const int32_t *nbr, const int32_t *cnt, int64_t N, int k, int32_t *deg,
                                 dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && k >= 1, "dmet_table_degree: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(nbr && deg, "dmet_table_degree: null pointer");
    hipLaunchKernelGGL(table_degree_kernel, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, as_stream(stream), nbr, cnt,
                       N, k, deg);
    DMET_LAUNCH_CHECK("table_degree_kernel");
    return 0;
}

extern "C" int dmet_table_edges(const int32_t *nbr, const int32_t *cnt, const int32_t *rowptr, int64_t N, int k,
                                int swap, int64_t *first, int64_t *second, int32_t *src32, int32_t *tgt32,
                                dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && k >= 1, "dmet_table_edges: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(nbr && rowptr && ((first && second) || (src32 && tgt32)), "dmet_table_edges: null pointer");
    DMET_REQUIRE((first == nullptr) == (second == nullptr) && (src32 == nullptr) == (tgt32 == nullptr),
                 "dmet_table_edges: outputs come in pairs");
    hipLaunchKernelGGL(table_edges_kernel, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, as_stream(stream), nbr, cnt,
                       rowptr, N, k, swap, first, second, src32, tgt32);
    DMET_LAUNCH_CHECK("table_edges_kernel");
    return 0;
}
