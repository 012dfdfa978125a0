// dense.hip -- tall-skinny reductions of the per-node dense layers' backward pass (gfx950).
//
// The model around the graph operators (/root/reference/model/graph_met_network.py:15-32,41-44) is a chain of
// tiny per-node layers over N ~ 3e5 nodes.  Their weight gradients are C[Ha,Hb] = A^T B with A = grad_out[N,Ha],
// B = input[N,Hb], Ha,Hb <= 64: a reduction over N that library GEMMs handle badly (measured 0.3-0.55 ms each) and
// whose embedding flavour (A = one-hot(index)) costs torch > 1 ms through a sort.  Here both are one kernel:
//   stage 1: every wavefront owns a fixed contiguous row range and accumulates it with fp32 MFMAs
//            (v_mfma_f32_32x32x2_f32: K = two rows per instruction, operands are coalesced 128-B row reads);
//   stage 2: the per-wavefront partial tiles are summed in wavefront order.
// Fixed ranges + fixed order => bitwise reproducible, no float atomics.
#include "common.h"

namespace dmet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXtyWavesPerBlock = 4;

// A source: dense rows (float) or one-hot rows generated from an int64 index vector.
template <bool ONEHOT>
__device__ __forceinline__ float load_a(const void *__restrict__ A, int64_t row, int64_t N, int Ha, int col)
{
    if (row >= N) return 0.0f;
    if (ONEHOT) {
        const int64_t idx = reinterpret_cast<const int64_t *>(A)[row];
        return (idx == (int64_t)col) ? 1.0f : 0.0f;
    }
    return (col < Ha) ? reinterpret_cast<const float *>(A)[row * Ha + col] : 0.0f;
}

template <int MT, int NT, bool ONEHOT>
__global__ __launch_bounds__(64 * kXtyWavesPerBlock) void xty_partial_kernel(const void *__restrict__ A,
                                                                             const float *__restrict__ Bm,
                                                                             int64_t N, int Ha, int Hb,
                                                                             int64_t rows_per_wave,
                                                                             float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * kXtyWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r_lo = wave * rows_per_wave;
    const int64_t r_hi = min(N, r_lo + rows_per_wave);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    for (int64_t r = r_lo; r < r_hi; r += 8) {
        float av[4][MT], bv[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = r + 2 * u + h;
            const bool ok = row < r_hi;
#pragma unroll
            for (int m = 0; m < MT; ++m) av[u][m] = ok ? load_a<ONEHOT>(A, row, N, Ha, m * 32 + c) : 0.0f;
#pragma unroll
            for (int n = 0; n < NT; ++n)
                bv[u][n] = (ok && n * 32 + c < Hb) ? Bm[row * Hb + n * 32 + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][m], bv[u][n], acc[m][n], 0, 0, 0);
    }
    // the block's wavefronts are summed in wavefront order through LDS: partial[block][a][b], a < MT*32, b < NT*32
    __shared__ float red[kXtyWavesPerBlock - 1][MT * NT * 16 * 64];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[wv - 1][((m * NT + n) * 16 + e) * 64 + lane] = acc[m][n][e];
    }
    __syncthreads();
    if (wv == 0) {
        float *out = partial + (int64_t)blockIdx.x * (MT * 32) * (NT * 32);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[m][n][e];
#pragma unroll
                    for (int q = 0; q < kXtyWavesPerBlock - 1; ++q) v += red[q][((m * NT + n) * 16 + e) * 64 + lane];
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    out[(int64_t)(m * 32 + row) * (NT * 32) + n * 32 + c] = v;
                }
    }
}

// Stage 2: 16 outputs per workgroup; 16 lanes per output each sum a strided subset of the partials in order, then
// the 16 sub-sums are added in lane order (fixed shape => reproducible).
__global__ __launch_bounds__(256) void xty_reduce_kernel(const float *__restrict__ partial, int64_t nparts, int Ha,
                                                          int Hb, int pa, int pb, float *__restrict__ C)
{
    __shared__ float sub[16][17];
    const int oi = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + oi;
    float s = 0.0f;
    if (o < Ha * Hb) {
        const int a = o / Hb, b = o - a * Hb;
        for (int64_t w = part; w < nparts; w += 16) s += partial[w * (int64_t)pa * pb + (int64_t)a * pb + b];
    }
    sub[part][oi] = s;
    __syncthreads();
    if (part == 0 && o < Ha * Hb) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sub[q][oi];
        C[o] = t;
    }
}

struct XtyPlan {
    int64_t nwaves, rows_per_wave;
    int mt, nt;
};

inline XtyPlan plan_xty(int64_t N, int Ha, int Hb)
{
    XtyPlan p;
    p.mt = (Ha + 31) / 32;
    p.nt = (Hb + 31) / 32;
    int64_t waves = 1024;
    int64_t rpw = (N + waves - 1) / waves;
    rpw = (rpw + 7) / 8 * 8;
    if (rpw < 8) rpw = 8;
    p.rows_per_wave = rpw;
    p.nwaves = (N + rpw - 1) / rpw;
    p.nwaves = (p.nwaves + kXtyWavesPerBlock - 1) / kXtyWavesPerBlock * kXtyWavesPerBlock;
    return p;
}

template <bool ONEHOT>
int launch_xty(const void *A, const float *Bm, int64_t N, int Ha, int Hb, float *C, void *ws, size_t ws_bytes,
               hipStream_t st, const char *who)
{
    DMET_REQUIRE(N >= 0 && Ha >= 1 && Ha <= 64 && Hb >= 1 && Hb <= 64, "%s: sizes out of range (Ha=%d, Hb=%d <= 64)", who,
                 Ha, Hb);
    DMET_REQUIRE(C, "%s: null output", who);
    if (N == 0) {
        hipError_t e = hipMemsetAsync(C, 0, sizeof(float) * (size_t)Ha * Hb, st);
        if (e != hipSuccess) return hip_fail(e, who);
        return 0;
    }
    DMET_REQUIRE(A && Bm && ws, "%s: null pointer", who);
    const XtyPlan p = plan_xty(N, Ha, Hb);
    const size_t need = sizeof(float) * (size_t)p.nwaves * (p.mt * 32) * (p.nt * 32);
    DMET_REQUIRE(ws_bytes >= need + 256, "%s: workspace too small", who);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    const dim3 grid((unsigned)(p.nwaves / kXtyWavesPerBlock)), block(64 * kXtyWavesPerBlock);
#define DMET_XTY(MT_, NT_)                                                                                   \
    hipLaunchKernelGGL((xty_partial_kernel<MT_, NT_, ONEHOT>), grid, block, 0, st, A, Bm, N, Ha, Hb,        \
                       p.rows_per_wave, partial)
    if (p.mt == 1 && p.nt == 1) DMET_XTY(1, 1);
    else if (p.mt == 1 && p.nt == 2) DMET_XTY(1, 2);
    else if (p.mt == 2 && p.nt == 1) DMET_XTY(2, 1);
    else DMET_XTY(2, 2);
#undef DMET_XTY
    DMET_LAUNCH_CHECK(who);
    hipLaunchKernelGGL(xty_reduce_kernel, dim3((unsigned)((Ha * Hb + 15) / 16)), dim3(256), 0, st, partial,
                       p.nwaves / kXtyWavesPerBlock, Ha, Hb, p.mt * 32, p.nt * 32, C);
    DMET_LAUNCH_CHECK(who);
    return 0;
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_xty_workspace_bytes(int64_t N, int Ha, int Hb)
{
    if (N <= 0 || Ha <= 0 || Hb <= 0 || Ha > 64 || Hb > 64) return 0;
    const XtyPlan p = plan_xty(N, Ha, Hb);
    return sizeof(float) * (size_t)p.nwaves * (p.mt * 32) * (p.nt * 32) + 512;
}

extern "C" int dmet_xty_f32(const float *A, const float *Bm, int64_t N, int Ha, int Hb, float *C, void *ws,
                            size_t ws_bytes, dmet_stream_t stream)
{
    return launch_xty<false>(A, Bm, N, Ha, Hb, C, ws, ws_bytes, as_stream(stream), "dmet_xty_f32");
}

extern "C" int dmet_onehot_xty_f32(const int64_t *index, const float *Bm, int64_t N, int R, int Hb, float *C,
                                   void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    return launch_xty<true>(index, Bm, N, R, Hb, C, ws, ws_bytes, as_stream(stream), "dmet_onehot_xty_f32");
}
