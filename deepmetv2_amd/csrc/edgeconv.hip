// edgeconv.hip -- K2/K3: EdgeConv message + aggregation kernels (gfx950).
//
// Replaces torch_geometric.nn.EdgeConv.forward = MessagePassing.propagate -> index_select x2 -> cat -> nn ->
// torch_scatter.scatter(max)   (constructed /root/reference/model/graph_met_network.py:36-38, invoked :65/:63).
//
// Fused path (nn == Linear(2H -> H), aggr == 'max', fixed-width neighbour table):
//     W.[x_i || x_j - x_i] + b  ==  (W1 - W2).x_i + b  +  W2.x_j          (exact algebra; fp32 reassociation only)
//   node_linear_split : P = x.(W1-W2)^T + b, Q = x.W2^T   per NODE, on the fp32 matrix cores
//                       (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain) -- 16x fewer flops than per edge
//   gather_max        : out[i] = P[i] + max_s Q[nbr[i,s]]  -- the HBM/L2-bound "gather + scatter_max" kernel
// Un-fused path (arbitrary nn / arbitrary edge list): edge_features -> user nn -> segment_max / segment_sum.
#include <stdlib.h>

#include <hip/hip_bf16.h>

#include <type_traits>

#include "common.h"
#include "nls_body.h"

namespace dmet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------
// node_linear_split: the node-level dense layer; the wavefront-level body lives in nls_body.h (shared with the kNN
// filter launch, whose trailing workgroups can run it behind the graph build: dmet_knn_local_dense_f32).
// ---------------------------------------------------------------------------------------------------------
template <int HIN, int HOUT, bool SLICED = false>
__global__ __launch_bounds__(256) void node_linear_split_kernel(const float *__restrict__ x, int64_t N,
                                                                 const float *__restrict__ W,
                                                                 const float *__restrict__ bias,
                                                                 float *__restrict__ P, float *__restrict__ Q)
{
    __shared__ __attribute__((aligned(16))) float tpose[SLICED ? 4 : 1][SLICED ? kNlsLdsFloats : 1];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    node_linear_split_wave<HIN, HOUT, SLICED>(x, N, W, bias, P, Q, tpose[SLICED ? wv : 0], wave, nwaves, threadIdx.x & 63);
}

// the same with the rows FORMED as residual + BatchNorm(raw) and written to aff.y (NlsAffine, csrc/nls_body.h)
template <int HIN, int HOUT, bool SLICED>
__global__ __launch_bounds__(256) void node_linear_split_bn_kernel(NlsAffine aff, int64_t N, const float *__restrict__ W,
                                                                    const float *__restrict__ bias,
                                                                    float *__restrict__ P, float *__restrict__ Q)
{
    __shared__ __attribute__((aligned(16))) float tpose[SLICED ? 4 : 1][SLICED ? kNlsLdsFloats : 1];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    node_linear_split_wave<HIN, HOUT, SLICED, true>(nullptr, N, W, bias, P, Q, tpose[SLICED ? wv : 0], wave, nwaves,
                                                    threadIdx.x & 63, aff);
}

// ---------------------------------------------------------------------------------------------------------
// bf16 variant of the node-level dense layer (BASELINE configs[2]): x and the split weights are rounded to bf16
// (RNE) and multiplied on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate); P stays fp32 (it is the
// node's own row, read once), Q is STORED as bf16 because it is the table that is gathered k times per node.
// Operand maps (32x32x16): lane l (r = l&31, h = l>>5) holds A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7.
// ---------------------------------------------------------------------------------------------------------
template <int HIN, int HOUT>
__global__ __launch_bounds__(256) void node_linear_split_bf16_kernel(const float *__restrict__ x, int64_t N,
                                                                      const float *__restrict__ W,
                                                                      const float *__restrict__ bias,
                                                                      float *__restrict__ P,
                                                                      unsigned short *__restrict__ Qh)
{
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    node_linear_split_bf16_wave<HIN, HOUT>(x, N, W, bias, P, Qh, wave, nwaves, threadIdx.x & 63);   // nls_body.h
}

// gather+max over a bf16 Q table (H = 32): 4 lanes per node, 8 channels (16 B of bf16) per lane.
template <bool WITH_ARG, int K4>
__global__ __launch_bounds__(256) void gather_max_bf16q_kernel(const float *__restrict__ P,
                                                                const unsigned short *__restrict__ Qh,
                                                                const int32_t *__restrict__ nbr, int64_t N,
                                                                float *__restrict__ out, uint8_t *__restrict__ arg)
{
    constexpr int H = 32, LPN = 4, NPB = 256 / LPN;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t node = (int64_t)bid * NPB + threadIdx.x / LPN;
    const int c8 = threadIdx.x % LPN;
    if (node >= N) return;
    const int4 *row4 = reinterpret_cast<const int4 *>(nbr + node * (4 * K4));
    const uint4 *Q16 = reinterpret_cast<const uint4 *>(Qh);    // 8 bf16 per uint4; 4 per row
    int4 idv[K4];
#pragma unroll
    for (int q = 0; q < K4; ++q) idv[q] = row4[q];
    const float4 p0 = reinterpret_cast<const float4 *>(P)[node * (H / 4) + 2 * c8];
    const float4 p1 = reinterpret_cast<const float4 *>(P)[node * (H / 4) + 2 * c8 + 1];
    const float ninf = -__builtin_inff();
    float best[8];
    int a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { best[c] = ninf; a[c] = 255; }
    bool any = false;
#pragma unroll
    for (int q0 = 0; q0 < K4; q0 += 2) {
        uint4 v[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j[4] = {idv[q0 + q].x, idv[q0 + q].y, idv[q0 + q].z, idv[q0 + q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[q][u] = (j[u] >= 0) ? Q16[(int64_t)j[u] * LPN + c8] : make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);  // -inf pairs
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j[4] = {idv[q0 + q].x, idv[q0 + q].y, idv[q0 + q].z, idv[q0 + q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * (q0 + q) + u;
                any = any || (j[u] >= 0);
                const unsigned w[4] = {v[q][u].x, v[q][u].y, v[q][u].z, v[q][u].w};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float lo = __uint_as_float(w[m] << 16), hi = __uint_as_float(w[m] & 0xffff0000u);
                    if (lo > best[2 * m]) { best[2 * m] = lo; a[2 * m] = s; }
                    if (hi > best[2 * m + 1]) { best[2 * m + 1] = hi; a[2 * m + 1] = s; }
                }
            }
        }
    }
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
    if (any) {
        o0 = make_float4(p0.x + best[0], p0.y + best[1], p0.z + best[2], p0.w + best[3]);
        o1 = make_float4(p1.x + best[4], p1.y + best[5], p1.z + best[6], p1.w + best[7]);
    }
    reinterpret_cast<float4 *>(out)[node * (H / 4) + 2 * c8] = o0;
    reinterpret_cast<float4 *>(out)[node * (H / 4) + 2 * c8 + 1] = o1;
    if (WITH_ARG) {
        uint2 pk;
        if (any) {
            pk.x = (unsigned)a[0] | ((unsigned)a[1] << 8) | ((unsigned)a[2] << 16) | ((unsigned)a[3] << 24);
            pk.y = (unsigned)a[4] | ((unsigned)a[5] << 8) | ((unsigned)a[6] << 16) | ((unsigned)a[7] << 24);
        } else {
            pk.x = pk.y = 0xffffffffu;
        }
        reinterpret_cast<uint2 *>(arg)[node * LPN + c8] = pk;
    }
}

// ---------------------------------------------------------------------------------------------------------
// gather_max (L2-gather form): H/4 lanes per node, each lane owns 4 channels; neighbours' Q rows come from the
// XCD's L2 (block ids are remapped so one XCD works on a contiguous window of events at a time).
// ---------------------------------------------------------------------------------------------------------
template <int H, bool WITH_ARG>
__global__ __launch_bounds__(256) void gather_max_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                          const int32_t *__restrict__ nbr,
                                                          const int32_t *__restrict__ cnt, int64_t N, int kmax,
                                                          float *__restrict__ out, uint8_t *__restrict__ arg)
{
    constexpr int LPN = H / 4;               // lanes per node
    constexpr int NPB = 256 / LPN;           // nodes per block
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t node = (int64_t)bid * NPB + threadIdx.x / LPN;
    const int c4 = threadIdx.x % LPN;
    if (node >= N) return;
    const int32_t *row = nbr + node * kmax;
    // cnt (optional): only the first cnt[node] slots of a row can hold neighbours (radius tables are 255 wide
    // but ~36 deep: walking the padding would cost 7x the useful work)
    const int k = cnt ? min(kmax, cnt[node]) : kmax;
    const float4 *Q4 = reinterpret_cast<const float4 *>(Q);
    const float ninf = -__builtin_inff();
    float4 best = make_float4(ninf, ninf, ninf, ninf);
    int a0 = 255, a1 = 255, a2 = 255, a3 = 255;
    bool any = false;
    for (int s0 = 0; s0 < k; s0 += 4) {
        int32_t j[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = (s0 + u < k) ? row[s0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = (j[u] >= 0) ? Q4[(int64_t)j[u] * LPN + c4] : make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            any = any || (j[u] >= 0);
            if (v[u].x > best.x) { best.x = v[u].x; a0 = s0 + u; }
            if (v[u].y > best.y) { best.y = v[u].y; a1 = s0 + u; }
            if (v[u].z > best.z) { best.z = v[u].z; a2 = s0 + u; }
            if (v[u].w > best.w) { best.w = v[u].w; a3 = s0 + u; }
        }
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any) {
        const float4 p = reinterpret_cast<const float4 *>(P)[node * LPN + c4];
        o = make_float4(p.x + best.x, p.y + best.y, p.z + best.z, p.w + best.w);
    }
    reinterpret_cast<float4 *>(out)[node * LPN + c4] = o;
    if (WITH_ARG) {
        uchar4 a = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
        if (!any) a = make_uchar4(255, 255, 255, 255);
        reinterpret_cast<uchar4 *>(arg)[node * LPN + c4] = a;
    }
}

// Deep-MLP variant for k == 4*K4 (the kNN tables: k = 8/16/32): the kernel above is latency bound (ids -> 4 gathers
// -> compare, 4 rows in flight per lane); here a lane loads all of its node's ids as int4 and has up to 16 row
// gathers in flight before the compare chain starts.  Same results (same compare order).
// ONLY_BIG (dmet_gather_max_mixed_f32): only the nodes of events too large for the LDS image are computed here -- the
// LDS-resident kernel takes the others in the same call (per-event, not per-batch, eligibility on ragged batches).
template <int H, bool WITH_ARG, int K4, bool ONLY_BIG = false>
__global__ __launch_bounds__(256) void gather_max_mlp_kernel(const float *__restrict__ P,
                                                              const float *__restrict__ Q,
                                                              const int32_t *__restrict__ nbr, int64_t N,
                                                              float *__restrict__ out, uint8_t *__restrict__ arg,
                                                              const int64_t *__restrict__ ptr = nullptr, int B = 0,
                                                              int lds_rows = 0)
{
    constexpr int LPN = H / 4;               // lanes per node
    constexpr int NPB = 256 / LPN;           // nodes per block
    constexpr int BATCH = (K4 < 4) ? K4 : 4; // int4 groups (= 4 rows each) gathered together
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t node = (int64_t)bid * NPB + threadIdx.x / LPN;
    const int c4 = threadIdx.x % LPN;
    if (node >= N) return;
    if (ONLY_BIG) {
        // a block covers NPB consecutive nodes: almost always one event (block-uniform answer from its first / last node)
        const int64_t first = (int64_t)bid * NPB, last = min(N, first + NPB) - 1;
        const int eb0 = find_event(ptr, B, first), eb1 = find_event(ptr, B, last);
        int eb = eb0;
        if (eb0 != eb1) eb = find_event(ptr, B, node);
        if (ptr[eb + 1] - ptr[eb] + 1 <= lds_rows) return;
    }
    const int4 *row4 = reinterpret_cast<const int4 *>(nbr + node * (4 * K4));
    const float4 *Q4 = reinterpret_cast<const float4 *>(Q);
    int4 idv[K4];
#pragma unroll
    for (int q = 0; q < K4; ++q) idv[q] = row4[q];
    const float4 p = reinterpret_cast<const float4 *>(P)[node * LPN + c4];
    const float ninf = -__builtin_inff();
    float4 best = make_float4(ninf, ninf, ninf, ninf);
    int a0 = 255, a1 = 255, a2 = 255, a3 = 255;
    bool any = false;
#pragma unroll
    for (int q0 = 0; q0 < K4; q0 += BATCH) {
        float4 v[BATCH][4];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int j[4] = {idv[q0 + q].x, idv[q0 + q].y, idv[q0 + q].z, idv[q0 + q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[q][u] = (j[u] >= 0) ? Q4[(int64_t)j[u] * LPN + c4] : make_float4(ninf, ninf, ninf, ninf);
        }
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            const int j[4] = {idv[q0 + q].x, idv[q0 + q].y, idv[q0 + q].z, idv[q0 + q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * (q0 + q) + u;
                any = any || (j[u] >= 0);
                if (v[q][u].x > best.x) { best.x = v[q][u].x; a0 = s; }
                if (v[q][u].y > best.y) { best.y = v[q][u].y; a1 = s; }
                if (v[q][u].z > best.z) { best.z = v[q][u].z; a2 = s; }
                if (v[q][u].w > best.w) { best.w = v[q][u].w; a3 = s; }
            }
        }
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any) o = make_float4(p.x + best.x, p.y + best.y, p.z + best.z, p.w + best.w);
    reinterpret_cast<float4 *>(out)[node * LPN + c4] = o;
    if (WITH_ARG) {
        uchar4 a = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
        if (!any) a = make_uchar4(255, 255, 255, 255);
        reinterpret_cast<uchar4 *>(arg)[node * LPN + c4] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------
// gather_max (LDS-resident form): one workgroup per (event, 8-channel slice).  The slice of Q for the whole event
// (n_b x 32 B, 144 KB at 4500 nodes) is staged in the CU's 160 KB LDS once; the k neighbour rows of every node are
// then gathered from LDS instead of L2.  Global traffic per workgroup is pure streaming: Q slice, neighbour ids,
// P slice in; out (+arg) slice out.  The H/8 slice workgroups of one event get block ids 8 apart, i.e. land on
// one XCD together (speed only), so the 128-B lines they share are fetched into that XCD's L2 once.
// Events too large for the LDS budget fall back to gathering from global memory inside the same kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSliceC = 8;                         // channels per slice
constexpr int kLdsGatherThreads = 1024;
constexpr int kLdsGatherBytes = 160 * 1024;        // whole LDS of a gfx950 CU
constexpr int kLdsGatherRows = kLdsGatherBytes / (kSliceC * 4);  // 5120 rows: events up to 5119 nodes (+ the -inf row)

// One (value, slot) update of the running maximum: best = max(best, v) keeping the LOWEST slot on ties (strict >),
// as three VALU ops on VCC.  Written as asm so the 64 compares of a node are not hoisted into 64 live SGPR masks
// (hipcc then spills them through v_writelane / v_readlane, which dominated the kernel).
#define DMET_MAX_ARG(best, a, v, slot)                                                                            \
    asm("v_cmp_gt_f32 vcc, %2, %0\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %3, vcc"          \
        : "+v"(best), "+v"(a) : "v"(v), "n"(slot) : "vcc")
#define DMET_MAX_ONLY(best, v) asm("v_max_f32 %0, %0, %1" : "+v"(best) : "v"(v))

// The two lanes of a node need the same k ids: each loads one half of the row and the halves are swapped between
// the lane pair with DPP (quad_perm [1,0,3,2]) -- half the id requests through the texture path.
__device__ __forceinline__ int dpp_swap_pair(int v)
{
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);   // quad_perm: lane ^ 1
}

template <int K4>
__device__ __forceinline__ void load_ids_pair(int4 (&ids)[K4], const int4 *__restrict__ row4, int half)
{
    if (K4 % 2 != 0) {
#pragma unroll
        for (int q = 0; q < K4; ++q) ids[q] = row4[q];
        return;
    }
    int4 mine[K4 / 2];
#pragma unroll
    for (int q = 0; q < K4 / 2; ++q) mine[q] = row4[half * (K4 / 2) + q];
#pragma unroll
    for (int q = 0; q < K4 / 2; ++q) {
        int4 other;
        other.x = dpp_swap_pair(mine[q].x); other.y = dpp_swap_pair(mine[q].y);
        other.z = dpp_swap_pair(mine[q].z); other.w = dpp_swap_pair(mine[q].w);
        ids[q] = half ? other : mine[q];
        ids[K4 / 2 + q] = half ? mine[q] : other;
    }
}

// Event-local uint16 table (dmet_knn_local_f32): a row is 2k bytes = K4 dwords per lane of the pair; half the id bytes
// of the int32 table, and the ids need no `- lo`.
template <int K4>
__device__ __forceinline__ void load_ids16_pair(unsigned (&w)[K4], const uint16_t *__restrict__ row, int half)
{
    const unsigned *src = reinterpret_cast<const unsigned *>(row) + half * K4;
    if (K4 % 4 == 0) {
#pragma unroll
        for (int q = 0; q < K4 / 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else if (K4 == 2) {
        const uint2 v = *reinterpret_cast<const uint2 *>(src);
        w[0] = v.x; w[1] = v.y;
    } else {
        // k = 20 (the reference's own default, model/graph_met_network.py:63): rows of 40 bytes, a lane's half starts on a
        // 4-byte boundary only -- plain dword loads
#pragma unroll
        for (int q = 0; q < K4; ++q) w[q] = src[q];
    }
}

// K4 = number of int4 id loads per node (k == 4*K4).  LDS image: rows 0..n-1 = the event's Q slice, row n = -inf
// (ids < 0 and anything outside the event map to it, so the gather needs no per-neighbour branch).  Per node all
// 4*K4 LDS reads are issued before the compare chain; the next node's ids and P slice are prefetched meanwhile.
// SLICED: P and Q are the slice-major tables of dmet_node_linear_split_sliced_f32 ([H/8][N][8]); out / arg stay
// row-major.
// One segment of work: nodes [i0, i1) of event b, slice sl (the whole event's Q slice is staged either way).
template <bool WITH_ARG, int K4, int GML_MODE, bool IDS16, bool SLICED, int ROWS = kLdsGatherRows, int THREADS = kLdsGatherThreads>
__device__ __forceinline__ void gather_max_lds_segment(
    float4 *__restrict__ qs, const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ nbr,
    const uint16_t *__restrict__ nbr16, const int64_t *__restrict__ ptr, int k, int H,
    float *__restrict__ out, uint8_t *__restrict__ arg, int64_t N, int skip_big, const int b, const int sl, const int i0,
    const int i1)
{
    static_assert(K4 >= 1 && K4 <= 8, "k = 4 K4 <= 32");
    constexpr int RPI = THREADS / 2;                                // rows per iteration (2 lanes per node)
    const int lo = (int)ptr[b], hi = (int)ptr[b + 1];
    const int n = hi - lo;
    if (n <= 0 || i0 >= i1) return;
    const int h4 = H / 4;                       // float4s per full row
    const float4 *Q4 = reinterpret_cast<const float4 *>(Q);
    const float4 *P4 = reinterpret_cast<const float4 *>(P);
    const int half = threadIdx.x & 1;
    const int r0 = i0 + (threadIdx.x >> 1);
    const int col4 = sl * 2 + half;             // float4 column of this lane inside a full row
    const float ninf = -__builtin_inff();
    // float4 index of this lane's 4 channels of node `i` in P / Q (row-major, or slice-major [H/8][N][8])
    auto pq_at = [&](const int64_t i) -> int64_t { return SLICED ? ((int64_t)sl * N + i) * 2 + half : i * h4 + col4; };

    if (n + 1 > ROWS) {
        if (skip_big) return;   // dmet_gather_max_mixed_f32: the L2-form kernel of the same call takes this event
        // event too large for the LDS image: same arithmetic, rows gathered from global memory (L2)
        for (int r = r0; r < i1; r += RPI) {
            const int64_t node = lo + r;
            float4 best = make_float4(ninf, ninf, ninf, ninf);
            int a0 = 255, a1 = 255, a2 = 255, a3 = 255;
            for (int s = 0; s < k; ++s) {
                const int j = nbr[node * k + s];
                if (j < 0) continue;
                const float4 v = Q4[pq_at(j)];
                if (v.x > best.x) { best.x = v.x; a0 = s; }
                if (v.y > best.y) { best.y = v.y; a1 = s; }
                if (v.z > best.z) { best.z = v.z; a2 = s; }
                if (v.w > best.w) { best.w = v.w; a3 = s; }
            }
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a0 != 255) {
                const float4 p = P4[pq_at(node)];
                o = make_float4(p.x + best.x, p.y + best.y, p.z + best.z, p.w + best.w);
            }
            reinterpret_cast<float4 *>(out)[node * h4 + col4] = o;
            if (WITH_ARG)
                reinterpret_cast<uchar4 *>(arg)[node * h4 + col4] =
                    make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
        }
        return;
    }

    // stage the Q slice with LDS-DMA (global_load_lds_dwordx4: per-lane source address, LDS destination =
    // wave-uniform base + lane*16, no registers, every chunk of the wave in flight at once); 32 rows per chunk
    if (GML_MODE != 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int nchunk = (n + 31) / 32;
        for (int c = wave; c < nchunk; c += THREADS / 64) {
            int row = 32 * c + (lane >> 1);
            row = min(row, n - 1);   // tail lanes re-read the last row into rows >= n (row n is rewritten below)
            const float4 *src = SLICED ? Q4 + ((int64_t)sl * N + lo + row) * 2 + (lane & 1)
                                       : Q4 + (int64_t)(lo + row) * h4 + sl * 2 + (lane & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(qs + 64 * c), 16, 0, 0);
        }
    }
    // first node's ids and P slice
    int4 ids[K4];
    unsigned idw[K4];
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < i1) {
        if constexpr (IDS16) {
            load_ids16_pair<K4>(idw, nbr16 + (int64_t)(lo + r0) * k, half);
        } else {
            const int4 *row4 = reinterpret_cast<const int4 *>(nbr + (int64_t)(lo + r0) * k);
            load_ids_pair<K4>(ids, row4, half);
        }
        pv = P4[pq_at(lo + r0)];
    }
    __builtin_amdgcn_s_waitcnt(0);   // LDS-DMA is counted by vmcnt and is not covered by the barrier itself
    __syncthreads();
    if (threadIdx.x < 2) qs[n * 2 + threadIdx.x] = make_float4(ninf, ninf, ninf, ninf);   // the -inf row
    __syncthreads();

    for (int r = r0; r < i1; r += RPI) {
        const int64_t node = lo + r;
        unsigned off[4 * K4];
        if constexpr (IDS16) {
            // this lane holds one half of the row (K4 dwords = 2*K4 ids); the other half comes from the pair lane
#pragma unroll
            for (int q = 0; q < K4; ++q) {
                const unsigned mine = idw[q], other = (unsigned)dpp_swap_pair((int)mine);
                const unsigned a = half ? other : mine, c = half ? mine : other;   // a: ids 2q.., c: ids 2K4+2q..
                off[2 * q] = min(a & 0xFFFFu, (unsigned)n) * 2 + half;
                off[2 * q + 1] = min(a >> 16, (unsigned)n) * 2 + half;
                off[2 * K4 + 2 * q] = min(c & 0xFFFFu, (unsigned)n) * 2 + half;
                off[2 * K4 + 2 * q + 1] = min(c >> 16, (unsigned)n) * 2 + half;
            }
        } else {
#pragma unroll
            for (int q = 0; q < K4; ++q) {
                off[4 * q + 0] = min((unsigned)(ids[q].x - lo), (unsigned)n) * 2 + half;
                off[4 * q + 1] = min((unsigned)(ids[q].y - lo), (unsigned)n) * 2 + half;
                off[4 * q + 2] = min((unsigned)(ids[q].z - lo), (unsigned)n) * 2 + half;
                off[4 * q + 3] = min((unsigned)(ids[q].w - lo), (unsigned)n) * 2 + half;
            }
        }
        const float4 p = pv;
        if (r + RPI < i1) {
            if constexpr (IDS16) {
                load_ids16_pair<K4>(idw, nbr16 + (node + RPI) * k, half);
            } else {
                const int4 *row4 = reinterpret_cast<const int4 *>(nbr + (node + RPI) * k);
                load_ids_pair<K4>(ids, row4, half);
            }
            pv = P4[pq_at(node + RPI)];
        }
        float bx = ninf, by = ninf, bz = ninf, bw = ninf;
        int a0 = 255, a1 = 255, a2 = 255, a3 = 255;
        if (WITH_ARG && GML_MODE != 2 && K4 <= 4) {
            // all 4*K4 rows of the node in registers (k <= 16; wider tables keep the chain below); max by v_max3 (half an op per value), then the winning slot by an
            // equality scan from the last slot down, so the LOWEST slot among equal maxima is written last (R4):
            // 2.5 VALU ops per value instead of the 3 of a compare/select/select chain
            float4 v[4 * K4];
#pragma unroll
            for (int u = 0; u < 4 * K4; ++u) v[u] = qs[off[u]];
#pragma unroll
            for (int u = 0; u < 4 * K4; u += 2) {
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(bx) : "v"(v[u].x), "v"(v[u + 1].x));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(by) : "v"(v[u].y), "v"(v[u + 1].y));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(bz) : "v"(v[u].z), "v"(v[u + 1].z));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(bw) : "v"(v[u].w), "v"(v[u + 1].w));
            }
            // FOUR slots x four channels per asm statement (32 instructions): hipcc puts an `s_nop 0` behind every asm
            // statement that clobbers vcc -- with one statement per (slot, channel) that was 64 s_nops next to 333 useful
            // instructions in this loop body (round 3, found in the ISA: 19 % of the wavefront's issue slots).
            // The scan runs from the last slot down, so the LOWEST slot among equal maxima is written last (R4).
#define DMET_EQ1(V_, B_, A_, S_) "v_cmp_eq_f32 vcc, " V_ ", " B_ "\n\tv_cndmask_b32_e64 " A_ ", " A_ ", " S_ ", vcc\n\t"
#define DMET_EQ_SLOT(X_, Y_, Z_, W_, S_) DMET_EQ1(X_, "%[bx]", "%[a0]", S_) DMET_EQ1(Y_, "%[by]", "%[a1]", S_) \
                                         DMET_EQ1(Z_, "%[bz]", "%[a2]", S_) DMET_EQ1(W_, "%[bw]", "%[a3]", S_)
#define DMET_EQ_GROUP(G_)                                                                                          \
    asm(DMET_EQ_SLOT("%[x3]", "%[y3]", "%[z3]", "%[w3]", "%[s3]") DMET_EQ_SLOT("%[x2]", "%[y2]", "%[z2]", "%[w2]", "%[s2]") \
        DMET_EQ_SLOT("%[x1]", "%[y1]", "%[z1]", "%[w1]", "%[s1]") DMET_EQ_SLOT("%[x0]", "%[y0]", "%[z0]", "%[w0]", "%[s0]") \
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)                                                \
        : [bx] "v"(bx), [by] "v"(by), [bz] "v"(bz), [bw] "v"(bw),                                                   \
          [x0] "v"(v[4 * (G_)].x), [y0] "v"(v[4 * (G_)].y), [z0] "v"(v[4 * (G_)].z), [w0] "v"(v[4 * (G_)].w),       \
          [x1] "v"(v[4 * (G_) + 1].x), [y1] "v"(v[4 * (G_) + 1].y), [z1] "v"(v[4 * (G_) + 1].z), [w1] "v"(v[4 * (G_) + 1].w), \
          [x2] "v"(v[4 * (G_) + 2].x), [y2] "v"(v[4 * (G_) + 2].y), [z2] "v"(v[4 * (G_) + 2].z), [w2] "v"(v[4 * (G_) + 2].w), \
          [x3] "v"(v[4 * (G_) + 3].x), [y3] "v"(v[4 * (G_) + 3].y), [z3] "v"(v[4 * (G_) + 3].z), [w3] "v"(v[4 * (G_) + 3].w), \
          [s0] "n"(4 * (G_)), [s1] "n"(4 * (G_) + 1), [s2] "n"(4 * (G_) + 2), [s3] "n"(4 * (G_) + 3)               \
        : "vcc")
            if constexpr (K4 <= 4) {      // (this branch is taken for K4 <= 4 only; wider tables never instantiate the asm)
                if constexpr (K4 >= 4) DMET_EQ_GROUP(3);
                if constexpr (K4 >= 3) DMET_EQ_GROUP(2);
                if constexpr (K4 >= 2) DMET_EQ_GROUP(1);
                DMET_EQ_GROUP(0);
            }
#undef DMET_EQ_GROUP
#undef DMET_EQ_SLOT
#undef DMET_EQ1
            // a channel whose maximum is still -inf had no valid neighbour (all of them then): slot 255
            if (!(bx > ninf)) a0 = 255;
            if (!(by > ninf)) a1 = 255;
            if (!(bz > ninf)) a2 = 255;
            if (!(bw > ninf)) a3 = 255;
        } else {
#pragma unroll
        for (int q0 = 0; q0 < (GML_MODE == 2 ? 0 : K4); q0 += 2) {
            float4 v[8];
            constexpr int kLastSlot = 4 * K4 - 1;     // odd K4 (k = 20): the last round holds four slots, not eight
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = qs[off[min(4 * q0 + u, kLastSlot)]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (4 * q0 + u > kLastSlot) continue;
                if (WITH_ARG) {
                    switch (4 * q0 + u) {   // the slot must be an immediate
#define DMET_CASE(S_) case S_: DMET_MAX_ARG(bx, a0, v[u].x, S_); DMET_MAX_ARG(by, a1, v[u].y, S_); \
                               DMET_MAX_ARG(bz, a2, v[u].z, S_); DMET_MAX_ARG(bw, a3, v[u].w, S_); break;
                        DMET_CASE(0) DMET_CASE(1) DMET_CASE(2) DMET_CASE(3) DMET_CASE(4) DMET_CASE(5) DMET_CASE(6)
                        DMET_CASE(7) DMET_CASE(8) DMET_CASE(9) DMET_CASE(10) DMET_CASE(11) DMET_CASE(12)
                        DMET_CASE(13) DMET_CASE(14) DMET_CASE(15) DMET_CASE(16) DMET_CASE(17) DMET_CASE(18)
                        DMET_CASE(19) DMET_CASE(20) DMET_CASE(21) DMET_CASE(22) DMET_CASE(23) DMET_CASE(24)
                        DMET_CASE(25) DMET_CASE(26) DMET_CASE(27) DMET_CASE(28) DMET_CASE(29) DMET_CASE(30)
                        DMET_CASE(31)
#undef DMET_CASE
                    }
                } else if ((u & 1) == 0) {   // two rows per v_max3_f32 (4 K4 is even: a pair never straddles the end)
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(bx) : "v"(v[u].x), "v"(v[u + 1].x));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(by) : "v"(v[u].y), "v"(v[u + 1].y));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(bz) : "v"(v[u].z), "v"(v[u + 1].z));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(bw) : "v"(v[u].w), "v"(v[u + 1].w));
                }
            }
        }
        }
        const bool any = WITH_ARG ? (a0 != 255) : (bx > ninf);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any) o = make_float4(p.x + bx, p.y + by, p.z + bz, p.w + bw);
        reinterpret_cast<float4 *>(out)[node * h4 + col4] = o;
        if (WITH_ARG) {
            const uchar4 a = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
            reinterpret_cast<uchar4 *>(arg)[node * h4 + col4] = a;
        }
    }
}

inline int num_cus()
{
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            cached = cus;
        else
            cached = 256;
    }
    return cached;
}

// balanced == 0: one workgroup per (event, slice); slices of one event are 8 blocks apart (same XCD under round-robin
// placement).  With B * H/8 = 256 that is exactly one lock-step round of the 256 CUs (the image is the whole LDS of a
// CU: one workgroup per CU) -- and at any other B a partial or a second round (B = 65: 260 workgroups, twice the time).
// balanced != 0: the grid is one workgroup per CU; the work axis [event][slice][node] (length H/8 * N) is cut into
// gridDim.x equal ranges, the workgroups of an XCD taking one contiguous part of it.  A workgroup walks the
// (event, slice) segments of its range, staging the Q slice of each (a range that cuts an (event, slice) in two makes
// two workgroups stage it: the price of the balance).  At B * H/8 = 256 equal events both mappings coincide.
// ROWS / THREADS: rows of the LDS image and threads of the workgroup.  The full form (5 120 rows = 160 KB, 1 024 threads) is
// one workgroup per CU -- by the image AND by its registers (115 VGPRs: four wavefronts per SIMD).  For batches whose largest
// event fits 2 559 rows (dmet_gather_max_lds_sliced_cap_f32: the sizes real data has) the half form -- 80 KB, 512 threads --
// puts TWO workgroups on a CU: one stages while the other gathers.
template <bool WITH_ARG, int K4, int GML_MODE = 0, bool IDS16 = false, bool SLICED = false, int ROWS = kLdsGatherRows,
          int THREADS = kLdsGatherThreads>
__global__ __launch_bounds__(THREADS) void gather_max_lds_kernel(
    const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ nbr,
    const uint16_t *__restrict__ nbr16, const int64_t *__restrict__ ptr, int B, int k, int H,
    float *__restrict__ out, uint8_t *__restrict__ arg, int64_t N, int skip_big, int balanced)
{
    __shared__ __attribute__((aligned(16))) float4 qs[ROWS * 2];   // [n_b + 1][2] float4 = 8 channels/node
    const int nsl = H / kSliceC;
    if (!balanced) {
        const int grp = blockIdx.x / (kNumXcd * nsl), rem = blockIdx.x % (kNumXcd * nsl);
        const int b = grp * kNumXcd + (rem % kNumXcd);
        const int sl = rem / kNumXcd;
        if (b >= B) return;
        gather_max_lds_segment<WITH_ARG, K4, GML_MODE, IDS16, SLICED, ROWS, THREADS>(qs, P, Q, nbr, nbr16, ptr, k, H, out, arg, N,
                                                                                      skip_big, b, sl, 0, (int)(ptr[b + 1] - ptr[b]));
        return;
    }
    const int64_t L = (int64_t)nsl * N;
    const int c = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    int64_t pos = L * c / gridDim.x;
    const int64_t end = L * (c + 1) / gridDim.x;
    while (pos < end) {
        int b0 = 0, b1 = B;                // event of the position: largest b with nsl * ptr[b] <= pos
        while (b1 - b0 > 1) {
            const int mid = (b0 + b1) >> 1;
            if ((int64_t)nsl * ptr[mid] <= pos) b0 = mid; else b1 = mid;
        }
        const int64_t lo = ptr[b0];
        const int n = (int)(ptr[b0 + 1] - lo);
        const int64_t rel = pos - (int64_t)nsl * lo;       // < nsl * n
        const int sl = (int)(rel / n), i0 = (int)(rel - (int64_t)sl * n);
        const int i1 = (int)((end - pos) < (int64_t)(n - i0) ? i0 + (end - pos) : n);
        gather_max_lds_segment<WITH_ARG, K4, GML_MODE, IDS16, SLICED, ROWS, THREADS>(qs, P, Q, nbr, nbr16, ptr, k, H, out, arg, N,
                                                                                      skip_big, b0, sl, i0, i1);
        pos += i1 - i0;
        __syncthreads();                   // the image is restaged by the next segment
    }
}

// LDS-resident form for COUNTED tables (radius graphs: kmax = 255 slots, cnt[i] ~ 36 of them used): the same
// (event, 8-channel slice) workgroups and Q image as gather_max_lds_kernel; a lane pair walks the first cnt[i] slots
// of its node's row eight at a time (ids -> 8 LDS rows -> compare chain, strict > keeps the lowest slot on ties).
// ARGJ: instead of the winning SLOT (uint8) the kernel stores the winner's event-local node id (uint16, 0xFFFF =
// none) -- the backward scatter then needs no look-up in the 255-wide table (dmet_gather_max_bwd_j16_f32).
// `order` (optional): the event's nodes are processed in this order (table_order_kernel: by slot count), so that the
// lane pairs of a wavefront walk rows of similar depth.
// LOC16: the ids come from the event-local uint16 copy of the rows that the radius kernel writes
// (dmet_radius_windowed_local_f32: rows of stride16 ids, 16-byte aligned, every started chunk of 8 padded with 0xFFFF):
// one aligned 16-byte load per 8 slots.  Out of the 255-wide int32 table every lane pair reads its 1020-byte-strided
// row with two unaligned 16-byte loads per 8 slots, each touching 32 cache lines per wavefront, re-read once per
// slice: the texture path, not the vector ALU, bounded that form (103 -> 75 us at 64 x 4500 nodes, 35 ids per row).
template <bool WITH_ARG, bool SLICED, bool ARGJ = false, bool LOC16 = false>
__global__ __launch_bounds__(kLdsGatherThreads) void gather_max_lds_counted_kernel(
    const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ nbr,
    const int32_t *__restrict__ cnt, const int64_t *__restrict__ ptr, int B, int kmax, int H,
    float *__restrict__ out, uint8_t *__restrict__ arg, int64_t N, const int32_t *__restrict__ order = nullptr,
    const uint16_t *__restrict__ nbr16 = nullptr, int stride16 = 0)
{
    __shared__ __attribute__((aligned(16))) float4 qs[kLdsGatherRows * 2];   // [n_b + 1][2] float4 = 8 channels/node
    constexpr int RPI = kLdsGatherThreads / 2;
    const int nsl = H / kSliceC;
    const int grp = blockIdx.x / (kNumXcd * nsl), rem = blockIdx.x % (kNumXcd * nsl);
    const int b = grp * kNumXcd + (rem % kNumXcd);
    const int sl = rem / kNumXcd;
    if (b >= B) return;
    const int lo = (int)ptr[b], hi = (int)ptr[b + 1];
    const int n = hi - lo;
    if (n <= 0) return;
    const int h4 = H / 4;
    const float4 *Q4 = reinterpret_cast<const float4 *>(Q);
    const float4 *P4 = reinterpret_cast<const float4 *>(P);
    const int half = threadIdx.x & 1;
    const int r0 = threadIdx.x >> 1;
    const int col4 = sl * 2 + half;
    const float ninf = -__builtin_inff();
    const bool in_lds = n + 1 <= kLdsGatherRows;       // block-uniform; larger events gather from global memory (L2)
    const int64_t table_len = N * (int64_t)kmax;        // entries in the table
    // float4 index of this lane's 4 channels of node `i` in P / Q (row-major, or slice-major [H/8][N][8])
    auto pq_at = [&](const int64_t i) -> int64_t { return SLICED ? ((int64_t)sl * N + i) * 2 + half : i * h4 + col4; };
    if (in_lds) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int nchunk = (n + 31) / 32;
        for (int c = wave; c < nchunk; c += kLdsGatherThreads / 64) {
            int row = 32 * c + (lane >> 1);
            row = min(row, n - 1);
            const float4 *src = SLICED ? Q4 + ((int64_t)sl * N + lo + row) * 2 + (lane & 1)
                                       : Q4 + (int64_t)(lo + row) * h4 + sl * 2 + (lane & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(qs + 64 * c), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x < 2) qs[n * 2 + threadIdx.x] = make_float4(ninf, ninf, ninf, ninf);   // the -inf row
        __syncthreads();
    }
    // Round 3: the LDS / global choice is block-uniform, but as a run-time value inside the slot loop hipcc compiled it
    // as a branch pair PER SLOT -- every one of a chunk's eight LDS reads was followed by its own s_waitcnt (eight
    // serialised LDS round trips per chunk).  As a compile-time constant of the row loop the eight reads issue back to back.
    auto rows = [&](auto in_lds_c) __attribute__((always_inline)) {
    constexpr bool in_lds = decltype(in_lds_c)::value;
    for (int r = r0; r < n; r += RPI) {
        const int64_t node = lo + (order ? order[lo + r] : r);
        const int32_t *row = nbr + node * kmax;
        const int m = min(kmax, cnt[node]);
        const float4 p = P4[pq_at(node)];
        float4 best = make_float4(ninf, ninf, ninf, ninf);
        constexpr int kNone = ARGJ ? 0xFFFF : 255;
        int a0 = kNone, a1 = kNone, a2 = kNone, a3 = kNone;
        bool any = false;
        if (LOC16) {
            any = m > 0;
            const uint16_t *r16 = nbr16 + node * stride16;
            for (int s0 = 0; s0 < m; s0 += 8) {
                const uint4 w = *reinterpret_cast<const uint4 *>(r16 + s0);
                const unsigned jl[8] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16,
                                        w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16};   // 0xFFFF = none
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (in_lds) v[u] = qs[min(jl[u], (unsigned)n) * 2 + half];
                    else v[u] = (jl[u] != 0xFFFFu) ? Q4[pq_at(lo + (int64_t)jl[u])] : make_float4(ninf, ninf, ninf, ninf);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int tag = ARGJ ? (int)jl[u] : (s0 + u);
                    if (v[u].x > best.x) { best.x = v[u].x; a0 = tag; }
                    if (v[u].y > best.y) { best.y = v[u].y; a1 = tag; }
                    if (v[u].z > best.z) { best.z = v[u].z; a2 = tag; }
                    if (v[u].w > best.w) { best.w = v[u].w; a3 = tag; }
                }
            }
        } else
        for (int s0 = 0; s0 < m; s0 += 8) {
            int32_t j[8];
            float4 v[8];
            // 16-byte loads (rows are only 4-byte aligned: 255-wide tables); reading up to 7 slots past m stays
            // inside the table except for the very last rows, which take the scalar loads.  (Loading the next eight
            // ids ahead of the compare chain was measured: no gain, the loop is bound by its VALU / LDS work and the
            // spread of cnt inside a wavefront.)
            if (node * kmax + s0 + 8 <= table_len) {
                struct __attribute__((packed, aligned(4))) I4 { int32_t a, b, c, d; };
                const I4 lo4 = *reinterpret_cast<const I4 *>(row + s0), hi4 = *reinterpret_cast<const I4 *>(row + s0 + 4);
                j[0] = lo4.a; j[1] = lo4.b; j[2] = lo4.c; j[3] = lo4.d;
                j[4] = hi4.a; j[5] = hi4.b; j[6] = hi4.c; j[7] = hi4.d;
#pragma unroll
                for (int u = 0; u < 8; ++u) j[u] = (s0 + u < m) ? j[u] : -1;
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) j[u] = (s0 + u < m) ? row[s0 + u] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (in_lds) v[u] = qs[min((unsigned)(j[u] - lo), (unsigned)n) * 2 + half];
                else v[u] = (j[u] >= 0) ? Q4[pq_at(j[u])] : make_float4(ninf, ninf, ninf, ninf);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                any = any || (j[u] >= 0);
                const int tag = ARGJ ? (j[u] - lo) : (s0 + u);   // what a winner is remembered by
                if (v[u].x > best.x) { best.x = v[u].x; a0 = tag; }
                if (v[u].y > best.y) { best.y = v[u].y; a1 = tag; }
                if (v[u].z > best.z) { best.z = v[u].z; a2 = tag; }
                if (v[u].w > best.w) { best.w = v[u].w; a3 = tag; }
            }
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any) o = make_float4(p.x + best.x, p.y + best.y, p.z + best.z, p.w + best.w);
        reinterpret_cast<float4 *>(out)[node * h4 + col4] = o;
        if (WITH_ARG) {
            if (ARGJ) {
                ushort4 a = make_ushort4((unsigned short)a0, (unsigned short)a1, (unsigned short)a2, (unsigned short)a3);
                if (!any) a = make_ushort4(0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF);
                reinterpret_cast<ushort4 *>(arg)[node * h4 + col4] = a;
            } else {
                uchar4 a = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
                if (!any) a = make_uchar4(255, 255, 255, 255);
                reinterpret_cast<uchar4 *>(arg)[node * h4 + col4] = a;
            }
        }
    }
    };
    if (in_lds) rows(std::true_type{});
    else rows(std::false_type{});
}

// order[ptr[b] .. ptr[b+1]) = the event's local node indices grouped by slot count, deepest rows first (counting sort
// in LDS; the order inside a group is arbitrary and influences no result -- every node is computed independently).
__global__ __launch_bounds__(1024) void table_order_kernel(const int32_t *__restrict__ cnt, const int64_t *__restrict__ ptr,
                                                           int B, int32_t *__restrict__ order)
{
    __shared__ int hist[256], start[256];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int64_t lo = ptr[b], hi = ptr[b + 1];
    const int n = (int)(hi - lo), tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[255 - min(255, max(0, cnt[lo + i]))], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < 256; ++c) { start[c] = run; run += hist[c]; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int pos = atomicAdd(&start[255 - min(255, max(0, cnt[lo + i]))], 1);
        order[lo + pos] = i;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fully fused EdgeConv(Linear(64->32), max) over a fixed-width table: neighbour gather + edge MLP + max in ONE
// launch, no P/Q round trip through memory.  One workgroup per (event, 8-channel slice):
//   phase 1  [Q slice | P slice] = x_event . [W2^T | (W1-W2)^T + b] for the slice's 8 output channels with fp32
//            matrix cores (v_mfma_f32_16x16x4_f32: 16 nodes x 16 columns per instruction group, exact fp32);
//            the Q slice goes to LDS (n x 32 B), the P slice is parked in the output buffer;
//   phase 2  out[i] = P[i] + max_s Q_lds[nbr[i,s]]  (+ uint8 arg), gathers served by LDS.
// HBM-side traffic is the algorithmic minimum: x once per slice (4x through L2), ids, out (+arg).
// MFMA operand maps (16x16x4): lane l holds A[row l&15][k l>>4], B[k l>>4][col l&15]; D: col = l&15,
// row = 4*(l>>4) + reg.  k-step s, lane quarter kk = l>>4  <->  input feature f = 8*kk + s (contiguous per lane).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool WITH_ARG, int K4, int FMODE = 0>
__global__ __launch_bounds__(kLdsGatherThreads) void edgeconv_fused_lds_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ nbr, const int64_t *__restrict__ ptr, int B, int k,
    const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ out, uint8_t *__restrict__ arg)
{
    constexpr int H = 32, h4 = H / 4, nsl = H / kSliceC;
    __shared__ __attribute__((aligned(16))) float4 qs[kLdsGatherRows * 2];   // [n_b + 1][8 floats]
    constexpr int RPI = kLdsGatherThreads / 2;
    const int grp = blockIdx.x / (kNumXcd * nsl), rem = blockIdx.x % (kNumXcd * nsl);
    const int b = grp * kNumXcd + (rem % kNumXcd);
    const int sl = rem / kNumXcd;
    if (b >= B) return;
    const int lo = (int)ptr[b], hi = (int)ptr[b + 1];
    const int n = hi - lo;
    if (n <= 0) return;
    const int half = threadIdx.x & 1;
    const int r0 = threadIdx.x >> 1;
    const int col4 = sl * 2 + half;
    const float ninf = -__builtin_inff();

    if (n + 1 > kLdsGatherRows) {
        // event too large for the LDS image: direct per-edge evaluation (correct, slow; the host routes such
        // batches to the unfused kernels when it knows the event sizes)
        for (int r = r0; r < n; r += RPI) {
            const int64_t node = lo + r;
            float best[4] = {ninf, ninf, ninf, ninf};
            int a[4] = {255, 255, 255, 255};
#pragma unroll 1
            for (int s = 0; s < k; ++s) {
                const int j = nbr[node * k + s];
                if (j < 0) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = sl * 8 + half * 4 + c;
                    float acc = bias ? bias[o] : 0.0f;
#pragma unroll 1
                    for (int f = 0; f < H; ++f) {
                        const float xi = x[node * H + f], xj = x[(int64_t)j * H + f];
                        acc = __builtin_fmaf(W[o * 2 * H + f], xi, acc);
                        acc = __builtin_fmaf(W[o * 2 * H + H + f], xj - xi, acc);
                    }
                    if (acc > best[c]) { best[c] = acc; a[c] = s; }
                }
            }
            const bool any = a[0] != 255;
            reinterpret_cast<float4 *>(out)[node * h4 + col4] =
                any ? make_float4(best[0], best[1], best[2], best[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (WITH_ARG)
                reinterpret_cast<uchar4 *>(arg)[node * h4 + col4] =
                    make_uchar4((unsigned char)a[0], (unsigned char)a[1], (unsigned char)a[2], (unsigned char)a[3]);
        }
        return;
    }

    // ---- phase 1: per-node dense layer for this slice on the fp32 matrix cores ----
    if (FMODE != 1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int j = lane & 15, kk = lane >> 4;
        // B operand: column j < 8 -> Q (W2 row), j >= 8 -> P ((W1-W2) row); feature 8*kk + s at step s
        const int o = sl * 8 + (j & 7);
        float bw[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float w1 = W[o * 2 * H + 8 * kk + s], w2 = W[o * 2 * H + H + 8 * kk + s];
            bw[s] = (j < 8) ? w2 : (w1 - w2);
        }
        const float cinit = (j >= 8 && bias) ? bias[o] : 0.0f;
        float *qf = reinterpret_cast<float *>(qs);
        const int ntile = (n + 15) / 16;
        constexpr int NW = kLdsGatherThreads / 64;   // waves
        constexpr int TU = 4;                        // tiles in flight per wave (loads issued before any MFMA)
        for (int t0 = wave; t0 < ntile; t0 += NW * TU) {
            float4 v0[TU], v1[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int t = t0 + u * NW;
                const int row = min(16 * t + (lane & 15), n - 1);
                const float4 *src = reinterpret_cast<const float4 *>(x + (int64_t)(lo + row) * H + 8 * kk);
                v0[u] = src[0];
                v1[u] = src[1];
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int t = t0 + u * NW;
                if (t >= ntile) break;
                const float a[8] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w, v1[u].x, v1[u].y, v1[u].z, v1[u].w};
                f32x4 acc = {cinit, cinit, cinit, cinit};
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[s], acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rr = 16 * t + 4 * kk + e;
                    if (rr < n) {
                        if (j < 8) qf[rr * 8 + j] = acc[e];
                        else out[(int64_t)(lo + rr) * H + sl * 8 + (j - 8)] = acc[e];
                    }
                }
            }
        }
        if (threadIdx.x < 2) qs[n * 2 + threadIdx.x] = make_float4(ninf, ninf, ninf, ninf);   // the -inf row
    }
    // first node's ids (independent of phase 1)
    int4 ids[K4];
    if (r0 < n) {
        const int4 *row4 = reinterpret_cast<const int4 *>(nbr + (int64_t)(lo + r0) * k);
#pragma unroll
        for (int q = 0; q < K4; ++q) ids[q] = row4[q];
    }
    __threadfence_block();
    __syncthreads();

    // ---- phase 2: gather from LDS + max (+arg), P read back from the output buffer ----
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < n) pv = reinterpret_cast<const float4 *>(out)[(int64_t)(lo + r0) * h4 + col4];
    for (int r = r0; r < (FMODE == 2 ? 0 : n); r += RPI) {
        const int64_t node = lo + r;
        unsigned off[4 * K4];
#pragma unroll
        for (int q = 0; q < K4; ++q) {
            off[4 * q + 0] = min((unsigned)(ids[q].x - lo), (unsigned)n) * 2 + half;
            off[4 * q + 1] = min((unsigned)(ids[q].y - lo), (unsigned)n) * 2 + half;
            off[4 * q + 2] = min((unsigned)(ids[q].z - lo), (unsigned)n) * 2 + half;
            off[4 * q + 3] = min((unsigned)(ids[q].w - lo), (unsigned)n) * 2 + half;
        }
        const float4 p = pv;
        if (r + RPI < n) {
            const int4 *row4 = reinterpret_cast<const int4 *>(nbr + (node + RPI) * k);
#pragma unroll
            for (int q = 0; q < K4; ++q) ids[q] = row4[q];
            pv = reinterpret_cast<const float4 *>(out)[(node + RPI) * h4 + col4];
        }
        float bx = ninf, by = ninf, bz = ninf, bw2 = ninf;
        int a0 = 255, a1 = 255, a2 = 255, a3 = 255;
#pragma unroll
        for (int q0 = 0; q0 < K4; q0 += 2) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = qs[off[4 * q0 + u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (WITH_ARG) {
                    switch (4 * q0 + u) {
#define DMET_CASE(S_) case S_: DMET_MAX_ARG(bx, a0, v[u].x, S_); DMET_MAX_ARG(by, a1, v[u].y, S_); \
                               DMET_MAX_ARG(bz, a2, v[u].z, S_); DMET_MAX_ARG(bw2, a3, v[u].w, S_); break;
                        DMET_CASE(0) DMET_CASE(1) DMET_CASE(2) DMET_CASE(3) DMET_CASE(4) DMET_CASE(5) DMET_CASE(6)
                        DMET_CASE(7) DMET_CASE(8) DMET_CASE(9) DMET_CASE(10) DMET_CASE(11) DMET_CASE(12)
                        DMET_CASE(13) DMET_CASE(14) DMET_CASE(15) DMET_CASE(16) DMET_CASE(17) DMET_CASE(18)
                        DMET_CASE(19) DMET_CASE(20) DMET_CASE(21) DMET_CASE(22) DMET_CASE(23) DMET_CASE(24)
                        DMET_CASE(25) DMET_CASE(26) DMET_CASE(27) DMET_CASE(28) DMET_CASE(29) DMET_CASE(30)
                        DMET_CASE(31)
#undef DMET_CASE
                    }
                } else {
                    DMET_MAX_ONLY(bx, v[u].x); DMET_MAX_ONLY(by, v[u].y);
                    DMET_MAX_ONLY(bz, v[u].z); DMET_MAX_ONLY(bw2, v[u].w);
                }
            }
        }
        const bool any = WITH_ARG ? (a0 != 255) : (bx > ninf);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any) o = make_float4(p.x + bx, p.y + by, p.z + bz, p.w + bw2);
        reinterpret_cast<float4 *>(out)[node * h4 + col4] = o;
        if (WITH_ARG) {
            const uchar4 a = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
            reinterpret_cast<uchar4 *>(arg)[node * h4 + col4] = a;
        }
    }
}

// Backward of gather_max w.r.t. Q: deterministic walk of the reverse index (ascending table position).
template <int H>
__global__ __launch_bounds__(256) void gather_max_bwd_kernel(const float *__restrict__ g_out,
                                                              const uint8_t *__restrict__ arg,
                                                              const int32_t *__restrict__ rev_ptr,
                                                              const int32_t *__restrict__ rev_slot, int64_t N,
                                                              int k, float *__restrict__ gQ)
{
    constexpr int LPN = H / 4;
    constexpr int NPB = 256 / LPN;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t node = (int64_t)bid * NPB + threadIdx.x / LPN;
    const int c4 = threadIdx.x % LPN;
    if (node >= N) return;
    const int lo = rev_ptr[node], hi = rev_ptr[node + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rpos = lo; rpos < hi; ++rpos) {
        const int e = rev_slot[rpos];
        const int i = e / k;
        const int s = e - i * k;
        const uchar4 a = reinterpret_cast<const uchar4 *>(arg)[(int64_t)i * LPN + c4];
        const float4 g = reinterpret_cast<const float4 *>(g_out)[(int64_t)i * LPN + c4];
        acc.x += (a.x == s) ? g.x : 0.0f;
        acc.y += (a.y == s) ? g.y : 0.0f;
        acc.z += (a.z == s) ? g.z : 0.0f;
        acc.w += (a.w == s) ? g.w : 0.0f;
    }
    reinterpret_cast<float4 *>(gQ)[node * LPN + c4] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// un-fused pieces
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_features_kernel(const float *__restrict__ x,
                                                             const int32_t *__restrict__ src,
                                                             const int32_t *__restrict__ tgt, int64_t E, int H,
                                                             float *__restrict__ feat)
{
    const int h4 = H / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = gid / h4;
    const int c4 = (int)(gid - e * h4);
    if (e >= E) return;
    const float4 xi = reinterpret_cast<const float4 *>(x)[(int64_t)tgt[e] * h4 + c4];
    const float4 xj = reinterpret_cast<const float4 *>(x)[(int64_t)src[e] * h4 + c4];
    float4 *o = reinterpret_cast<float4 *>(feat) + e * (2 * h4);
    o[c4] = xi;
    o[h4 + c4] = make_float4(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z, xj.w - xi.w);
}

template <bool IS_MAX>
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float *__restrict__ msg,
                                                              const int32_t *__restrict__ rowptr, int64_t N,
                                                              int H, float *__restrict__ out,
                                                              int32_t *__restrict__ arg)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / H;
    const int c = (int)(gid - i * H);
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    if (IS_MAX) {
        float best = 0.0f;
        int a = -1;
        for (int e = lo; e < hi; ++e) {
            const float v = msg[(int64_t)e * H + c];
            if (a < 0 || v > best) { best = v; a = e; }
        }
        out[gid] = best;
        if (arg) arg[gid] = a;
    } else {
        float s = 0.0f;
        for (int e = lo; e < hi; ++e) s += msg[(int64_t)e * H + c];
        out[gid] = s;
    }
}

template <bool IS_MAX>
__global__ __launch_bounds__(256) void segment_reduce_bwd_kernel(const float *__restrict__ g_out,
                                                                  const int32_t *__restrict__ arg,
                                                                  const int32_t *__restrict__ rowptr, int64_t N,
                                                                  int H, float *__restrict__ g_msg)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / H;
    const int c = (int)(gid - i * H);
    if (i >= N) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    const float g = g_out[gid];
    const int a = IS_MAX ? arg[gid] : 0;
    for (int e = lo; e < hi; ++e) g_msg[(int64_t)e * H + c] = (!IS_MAX || a == e) ? g : 0.0f;
}

__global__ __launch_bounds__(256) void edge_features_bwd_kernel(const float *__restrict__ g_feat,
                                                                 const int32_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ srcptr,
                                                                 const int32_t *__restrict__ srcperm, int64_t N,
                                                                 int H, float *__restrict__ gx)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / H;
    const int c = (int)(gid - i * H);
    if (i >= N) return;
    float s = 0.0f;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        const float *g = g_feat + (int64_t)e * (2 * H);
        s += g[c] - g[H + c];
    }
    for (int rp = srcptr[i]; rp < srcptr[i + 1]; ++rp) {
        const int e = srcperm[rp];
        s += g_feat[(int64_t)e * (2 * H) + H + c];
    }
    gx[gid] = s;
}

template <int HIN, int HOUT, bool SLICED = false>
int launch_node_linear(const float *x, int64_t N, const float *W, const float *b, float *P, float *Q,
                       hipStream_t st)
{
    const int64_t ntiles = (N + 31) / 32;
    // one workgroup (four wavefronts) per CU, each wavefront walking ~9 tiles with the next tile's rows in flight: at
    // 2048 workgroups a wavefront had ONE tile, i.e. 64 weight loads and an exposed row load per 32 MFMAs (25.1 -> 20.6 us
    // in the training step at 288 000 nodes)
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > num_cus()) blocks = num_cus();
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((node_linear_split_kernel<HIN, HOUT, SLICED>), dim3((unsigned)blocks), dim3(256), 0, st, x, N, W,
                       b, P, Q);
    DMET_LAUNCH_CHECK("node_linear_split_kernel");
    return 0;
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_edgeconv_linear_workspace_bytes(int64_t N, int Hout)
{
    if (N <= 0 || Hout <= 0) return 0;
    return (size_t)N * Hout * sizeof(float) * 2 + 512;
}

extern "C" int dmet_node_linear_split_f32(const float *x, int64_t N, int Hin, int Hout, const float *W,
                                          const float *b, float *P, float *Q, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0, "dmet_node_linear_split_f32: N<0");
    if (N == 0) return 0;
    DMET_REQUIRE(x && W && P && Q, "dmet_node_linear_split_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(P) && aligned16(Q), "dmet_node_linear_split_f32: pointers must be 16-B aligned");
    hipStream_t st = as_stream(stream);
    if (Hin == 32 && Hout == 32) return launch_node_linear<32, 32>(x, N, W, b, P, Q, st);
    if (Hin == 64 && Hout == 64) return launch_node_linear<64, 64>(x, N, W, b, P, Q, st);
    if (Hin == 64 && Hout == 32) return launch_node_linear<64, 32>(x, N, W, b, P, Q, st);
    if (Hin == 32 && Hout == 64) return launch_node_linear<32, 64>(x, N, W, b, P, Q, st);
    set_error("dmet_node_linear_split_f32: unsupported (Hin,Hout)=(%d,%d); supported: 32/64", Hin, Hout);
    return -22;
}

extern "C" int dmet_node_linear_split_sliced_f32(const float *x, int64_t N, int Hin, int Hout, const float *W,
                                                 const float *b, float *P, float *Q, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0, "dmet_node_linear_split_sliced_f32: N<0");
    if (N == 0) return 0;
    DMET_REQUIRE(x && W && P && Q, "dmet_node_linear_split_sliced_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(P) && aligned16(Q), "dmet_node_linear_split_sliced_f32: pointers must be 16-B aligned");
    hipStream_t st = as_stream(stream);
    if (Hin == 32 && Hout == 32) return launch_node_linear<32, 32, true>(x, N, W, b, P, Q, st);
    if (Hin == 64 && Hout == 64) return launch_node_linear<64, 64, true>(x, N, W, b, P, Q, st);
    if (Hin == 64 && Hout == 32) return launch_node_linear<64, 32, true>(x, N, W, b, P, Q, st);
    if (Hin == 32 && Hout == 64) return launch_node_linear<32, 64, true>(x, N, W, b, P, Q, st);
    set_error("dmet_node_linear_split_sliced_f32: unsupported (Hin,Hout)=(%d,%d); supported: 32/64", Hin, Hout);
    return -22;
}

extern "C" int dmet_bn_node_linear_split_f32(const float *raw, const float *residual, const float *gamma, const float *beta,
                                             const float *mean, const float *invstd, float *y, int64_t N, int H,
                                             const float *W, const float *b, int sliced, float *P, float *Q,
                                             dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0, "dmet_bn_node_linear_split_f32: N<0");
    if (N == 0) return 0;
    DMET_REQUIRE(H == 32, "dmet_bn_node_linear_split_f32: H=%d (built for 32 -> 32)", H);
    DMET_REQUIRE(raw && gamma && beta && mean && invstd && y && W && P && Q, "dmet_bn_node_linear_split_f32: null pointer");
    DMET_REQUIRE(aligned16(raw) && aligned16(y) && aligned16(P) && aligned16(Q) && aligned16(gamma) && aligned16(beta) &&
                     aligned16(mean) && aligned16(invstd) && (!residual || aligned16(residual)),
                 "dmet_bn_node_linear_split_f32: pointers must be 16-B aligned");
    NlsAffine aff{raw, residual, gamma, beta, mean, invstd, y};
    const int64_t ntiles = (N + 31) / 32;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > num_cus()) blocks = num_cus();
    if (blocks < 1) blocks = 1;
    hipStream_t st = as_stream(stream);
    if (sliced)
        hipLaunchKernelGGL((node_linear_split_bn_kernel<32, 32, true>), dim3((unsigned)blocks), dim3(256), 0, st, aff, N, W, b, P, Q);
    else
        hipLaunchKernelGGL((node_linear_split_bn_kernel<32, 32, false>), dim3((unsigned)blocks), dim3(256), 0, st, aff, N, W, b, P, Q);
    DMET_LAUNCH_CHECK("node_linear_split_bn_kernel");
    return 0;
}

extern "C" int dmet_gather_max_counted_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt,
                                           int64_t N, int k, int H, float *out, uint8_t *arg, dmet_stream_t stream);

extern "C" int dmet_gather_max_f32(const float *P, const float *Q, const int32_t *nbr, const int64_t *ptr, int B,
                                   int64_t N, int k, int H, float *out, uint8_t *arg, dmet_stream_t stream)
{
    (void)ptr; (void)B;
    return dmet_gather_max_counted_f32(P, Q, nbr, nullptr, N, k, H, out, arg, stream);
}

extern "C" int dmet_gather_max_counted_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt,
                                           int64_t N, int k, int H, float *out, uint8_t *arg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_gather_max_f32: N out of range");
    DMET_REQUIRE(k >= 1 && k <= 255, "dmet_gather_max_f32: k=%d not in [1,255] (arg is uint8, 255 = none)", k);
    if (N == 0) return 0;
    DMET_REQUIRE(P && Q && nbr && out, "dmet_gather_max_f32: null pointer");
    DMET_REQUIRE(aligned16(P) && aligned16(Q) && aligned16(out), "dmet_gather_max_f32: pointers must be 16-B aligned");
    hipStream_t st = as_stream(stream);
#define DMET_GM(HH)                                                                                             \
    do {                                                                                                        \
        constexpr int NPB = 256 / (HH / 4);                                                                     \
        const int64_t blocks = (N + NPB - 1) / NPB;                                                             \
        if (arg)                                                                                                \
            hipLaunchKernelGGL((gather_max_kernel<HH, true>), dim3((unsigned)blocks), dim3(256), 0, st, P, Q,   \
                               nbr, cnt, N, k, out, arg);                                                       \
        else                                                                                                    \
            hipLaunchKernelGGL((gather_max_kernel<HH, false>), dim3((unsigned)blocks), dim3(256), 0, st, P, Q,  \
                               nbr, cnt, N, k, out, arg);                                                       \
    } while (0)
    if (H == 32 && !cnt && (k == 8 || k == 16 || k == 32) && aligned16(nbr)) {
        // the kNN tables of the hot path: deep-MLP variant
        const int64_t blocks = (N + 31) / 32;
#define DMET_GMM(K4_)                                                                                          \
        do {                                                                                                   \
            if (arg)                                                                                           \
                hipLaunchKernelGGL((gather_max_mlp_kernel<32, true, K4_>), dim3((unsigned)blocks), dim3(256),  \
                                   0, st, P, Q, nbr, N, out, arg);                                             \
            else                                                                                               \
                hipLaunchKernelGGL((gather_max_mlp_kernel<32, false, K4_>), dim3((unsigned)blocks), dim3(256), \
                                   0, st, P, Q, nbr, N, out, arg);                                             \
        } while (0)
        if (k == 8) DMET_GMM(2);
        else if (k == 16) DMET_GMM(4);
        else DMET_GMM(8);
#undef DMET_GMM
    }
    else if (H == 32) DMET_GM(32);
    else if (H == 64) DMET_GM(64);
    else if (H == 128) DMET_GM(128);
    else if (H == 16) DMET_GM(16);
    else {
        set_error("dmet_gather_max_f32: unsupported H=%d (16/32/64/128)", H);
        return -22;
    }
#undef DMET_GM
    DMET_LAUNCH_CHECK("gather_max_kernel");
    return 0;
}

extern "C" int dmet_edgeconv_fused_lds_f32(const float *x, const int32_t *nbr, const int64_t *ptr, int B, int64_t N,
                                           int k, int Hin, int Hout, const float *W, const float *b, float *out,
                                           uint8_t *arg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_edgeconv_fused_lds_f32: N out of range");
    DMET_REQUIRE(Hin == 32 && Hout == 32, "dmet_edgeconv_fused_lds_f32: only (Hin,Hout)=(32,32) is built, got (%d,%d)", Hin, Hout);
    DMET_REQUIRE(k == 8 || k == 16 || k == 32, "dmet_edgeconv_fused_lds_f32: k=%d must be 8, 16 or 32", k);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(x && nbr && ptr && W && out, "dmet_edgeconv_fused_lds_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(out) && aligned16(nbr), "dmet_edgeconv_fused_lds_f32: pointers must be 16-B aligned");
    const int nsl = 32 / kSliceC;
    const int64_t groups = (B + kNumXcd - 1) / kNumXcd;
    const int64_t blocks = groups * kNumXcd * nsl;
    hipStream_t st = as_stream(stream);
#define DMET_EFL(K4_)                                                                                          \
    do {                                                                                                       \
        if (arg)                                                                                               \
            hipLaunchKernelGGL((edgeconv_fused_lds_kernel<true, K4_>), dim3((unsigned)blocks),                 \
                               dim3(kLdsGatherThreads), 0, st, x, nbr, ptr, B, k, W, b, out, arg);             \
        else                                                                                                   \
            hipLaunchKernelGGL((edgeconv_fused_lds_kernel<false, K4_>), dim3((unsigned)blocks),                \
                               dim3(kLdsGatherThreads), 0, st, x, nbr, ptr, B, k, W, b, out, arg);             \
    } while (0)
#ifdef DMET_KNN_EXPERIMENT
    if (const char *e = arg ? getenv("DMET_EFL_MODE") : nullptr) {
        const int m = atoi(e);
        if (m == 1) hipLaunchKernelGGL((edgeconv_fused_lds_kernel<true, 4, 1>), dim3((unsigned)blocks), dim3(kLdsGatherThreads), 0, st, x, nbr, ptr, B, k, W, b, out, arg);
        else if (m == 2) hipLaunchKernelGGL((edgeconv_fused_lds_kernel<true, 4, 2>), dim3((unsigned)blocks), dim3(kLdsGatherThreads), 0, st, x, nbr, ptr, B, k, W, b, out, arg);
        else DMET_EFL(4);
        DMET_LAUNCH_CHECK("edgeconv_fused_lds_kernel");
        return 0;
    }
#endif
    if (k == 8) DMET_EFL(2);
    else if (k == 16) DMET_EFL(4);
    else DMET_EFL(8);
#undef DMET_EFL
    DMET_LAUNCH_CHECK("edgeconv_fused_lds_kernel");
    return 0;
}

extern "C" int dmet_node_linear_split_bf16(const float *x, int64_t N, int Hin, int Hout, const float *W, const float *b,
                                           float *P, uint16_t *Qh, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0, "dmet_node_linear_split_bf16: N<0");
    if (N == 0) return 0;
    DMET_REQUIRE(x && W && P && Qh, "dmet_node_linear_split_bf16: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(P) && aligned16(Qh), "dmet_node_linear_split_bf16: pointers must be 16-B aligned");
    DMET_REQUIRE(Hin == 32 && Hout == 32, "dmet_node_linear_split_bf16: only (Hin,Hout)=(32,32) is built, got (%d,%d)", Hin, Hout);
    const int64_t ntiles = (N + 31) / 32;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((node_linear_split_bf16_kernel<32, 32>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x,
                       N, W, b, P, Qh);
    DMET_LAUNCH_CHECK("node_linear_split_bf16_kernel");
    return 0;
}

extern "C" int dmet_gather_max_bf16q(const float *P, const uint16_t *Qh, const int32_t *nbr, int64_t N, int k, int H,
                                     float *out, uint8_t *arg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_gather_max_bf16q: N out of range");
    DMET_REQUIRE(H == 32, "dmet_gather_max_bf16q: only H=32 is built, got %d", H);
    DMET_REQUIRE(k == 8 || k == 16 || k == 32, "dmet_gather_max_bf16q: k=%d must be 8, 16 or 32", k);
    if (N == 0) return 0;
    DMET_REQUIRE(P && Qh && nbr && out, "dmet_gather_max_bf16q: null pointer");
    DMET_REQUIRE(aligned16(P) && aligned16(Qh) && aligned16(out) && aligned16(nbr), "dmet_gather_max_bf16q: pointers must be 16-B aligned");
    hipStream_t st = as_stream(stream);
    const int64_t blocks = (N + 63) / 64;
#define DMET_GMB(K4_)                                                                                           \
    do {                                                                                                        \
        if (arg)                                                                                                \
            hipLaunchKernelGGL((gather_max_bf16q_kernel<true, K4_>), dim3((unsigned)blocks), dim3(256), 0, st,  \
                               P, Qh, nbr, N, out, arg);                                                        \
        else                                                                                                    \
            hipLaunchKernelGGL((gather_max_bf16q_kernel<false, K4_>), dim3((unsigned)blocks), dim3(256), 0, st, \
                               P, Qh, nbr, N, out, arg);                                                        \
    } while (0)
    if (k == 8) DMET_GMB(2);
    else if (k == 16) DMET_GMB(4);
    else DMET_GMB(8);
#undef DMET_GMB
    DMET_LAUNCH_CHECK("gather_max_bf16q_kernel");
    return 0;
}

constexpr int kLdsGatherRowsHalf = kLdsGatherRows / 2;       // 2 560 rows = 80 KB
constexpr int kLdsGatherThreadsHalf = kLdsGatherThreads / 2; // 512 threads: two such workgroups per CU

static int gather_max_lds_impl(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr16,
                               const int64_t *ptr, int B, int64_t N, int k, int H, float *out, uint8_t *arg,
                               bool sliced, dmet_stream_t stream, int skip_big = 0, int64_t max_nodes = 0)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_gather_max_lds_f32: N out of range");
    DMET_REQUIRE(k >= 1 && k <= 255, "dmet_gather_max_lds_f32: k=%d not in [1,255]", k);
    DMET_REQUIRE(H >= kSliceC && H % kSliceC == 0 && H <= DMET_MAX_H, "dmet_gather_max_lds_f32: H=%d must be a multiple of %d", H,
                 kSliceC);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(P && Q && nbr && ptr && out, "dmet_gather_max_lds_f32: null pointer");
    DMET_REQUIRE(aligned16(P) && aligned16(Q) && aligned16(out) && aligned16(nbr) && aligned16(nbr16),
                 "dmet_gather_max_lds_f32: pointers must be 16-B aligned");
    const int nsl = H / kSliceC;
    const int64_t groups = (B + kNumXcd - 1) / kNumXcd;
    int64_t blocks = groups * kNumXcd * nsl;
    // Measured at 4500-node events on 256 CUs (tools/gather_sweep.py, us, one workgroup per (event, slice) / balanced):
    // B = 32: 35.8 / 29.5, 48: 37.2 / 41.0, 64: 39.6 / 40.8, 65: 56.6 / 45.0, 96: 65.2 / 67.2, 128: 69.8 / 94.2 -- a range
    // that cuts an (event, slice) stages its slice twice, and independent workgroups drift apart so that one CU's staging
    // hides under its neighbours' gathers, which the lock-step ranges of the balanced form do not.  It wins when the
    // (event, slice) units fill well under the chip or leave a short tail round.  DMET_GATHER_BALANCED=0/1 forces either.
    const int cus = num_cus();
    const int64_t units = (int64_t)B * nsl;
    int balanced = units * 16 <= (int64_t)cus * 9 || (units > cus && units % cus != 0 && units % cus <= cus / 8);
    if (const char *e = getenv("DMET_GATHER_BALANCED")) balanced = atoi(e) != 0;
    if (balanced) {
        blocks = cus;
        const int64_t most = ((int64_t)nsl * N + 63) / 64;
        if (blocks > most) blocks = most;
    }
    hipStream_t st = as_stream(stream);
    // batches of small events (the caller's hint; an event beyond it would take the in-kernel L2 path: slower, never wrong):
    // the half form, one workgroup per (event, slice), two per CU
    if (sliced && nbr16 && max_nodes > 0 && max_nodes + 1 <= kLdsGatherRowsHalf && (k == 8 || k == 16 || k == 20 || k == 32) &&
        !env_is("DMET_GATHER_HALF_IMAGE", "0")) {
        const int64_t hblocks = groups * kNumXcd * nsl;
#define DMET_GMH(K4_)                                                                                                  \
        do {                                                                                                           \
            if (arg)                                                                                                   \
                hipLaunchKernelGGL((gather_max_lds_kernel<true, K4_, 0, true, true, kLdsGatherRowsHalf, kLdsGatherThreadsHalf>), \
                                   dim3((unsigned)hblocks), dim3(kLdsGatherThreadsHalf), 0, st, P, Q, nbr, nbr16, ptr, B, k, H, \
                                   out, arg, N, skip_big, 0);                                                          \
            else                                                                                                       \
                hipLaunchKernelGGL((gather_max_lds_kernel<false, K4_, 0, true, true, kLdsGatherRowsHalf, kLdsGatherThreadsHalf>), \
                                   dim3((unsigned)hblocks), dim3(kLdsGatherThreadsHalf), 0, st, P, Q, nbr, nbr16, ptr, B, k, H, \
                                   out, arg, N, skip_big, 0);                                                          \
        } while (0)
        if (k == 8) DMET_GMH(2);
        else if (k == 16) DMET_GMH(4);
        else if (k == 20) DMET_GMH(5);
        else DMET_GMH(8);
#undef DMET_GMH
        DMET_LAUNCH_CHECK("gather_max_lds_kernel (80 KB image)");
        return 0;
    }
#define DMET_GML_LAUNCH(ARG_, K4_, I16_)                                                                          \
    do {                                                                                                          \
        if (sliced)                                                                                               \
            hipLaunchKernelGGL((gather_max_lds_kernel<ARG_, K4_, 0, I16_, true>), dim3((unsigned)blocks),         \
                               dim3(kLdsGatherThreads), 0, st, P, Q, nbr, nbr16, ptr, B, k, H, out, arg, N, skip_big, balanced); \
        else                                                                                                      \
            hipLaunchKernelGGL((gather_max_lds_kernel<ARG_, K4_, 0, I16_, false>), dim3((unsigned)blocks),        \
                               dim3(kLdsGatherThreads), 0, st, P, Q, nbr, nbr16, ptr, B, k, H, out, arg, N, skip_big, balanced); \
    } while (0)
#define DMET_GML(K4_)                                                                                          \
    do {                                                                                                       \
        if (arg) { if (nbr16) DMET_GML_LAUNCH(true, K4_, true); else DMET_GML_LAUNCH(true, K4_, false); }      \
        else { if (nbr16) DMET_GML_LAUNCH(false, K4_, true); else DMET_GML_LAUNCH(false, K4_, false); }        \
    } while (0)
#ifdef DMET_KNN_EXPERIMENT
    // phase timing (tools/build_variant.sh exp -DDMET_KNN_EXPERIMENT): DMET_GML_MODE=1 skips the staging of the Q
    // slice, =2 the LDS gather + compare chain (results are then meaningless)
    if (const char *e = (arg && nbr16 && k == 16) ? getenv("DMET_GML_MODE") : nullptr) {
        const int m = atoi(e);
        if (m == 1) hipLaunchKernelGGL((gather_max_lds_kernel<true, 4, 1, true>), dim3((unsigned)blocks), dim3(kLdsGatherThreads), 0, st, P, Q, nbr, nbr16, ptr, B, k, H, out, arg, N, skip_big, balanced);
        else if (m == 2) hipLaunchKernelGGL((gather_max_lds_kernel<true, 4, 2, true>), dim3((unsigned)blocks), dim3(kLdsGatherThreads), 0, st, P, Q, nbr, nbr16, ptr, B, k, H, out, arg, N, skip_big, balanced);
        else DMET_GML(4);
        DMET_LAUNCH_CHECK("gather_max_lds_kernel");
        return 0;
    }
#endif
    if (k == 8) DMET_GML(2);
    else if (k == 16) DMET_GML(4);
    else if (k == 20) DMET_GML(5);
    else if (k == 32) DMET_GML(8);
    else {
        DMET_REQUIRE(!sliced, "dmet_gather_max_lds_sliced_f32: k=%d (slice-major tables need k in {8,16,20,32})", k);
        return dmet_gather_max_f32(P, Q, nbr, ptr, B, N, k, H, out, arg, stream);  // other widths: L2 form
    }
#undef DMET_GML
#undef DMET_GML_LAUNCH
    DMET_LAUNCH_CHECK("gather_max_lds_kernel");
    return 0;
}

extern "C" int dmet_gather_max_lds_f32(const float *P, const float *Q, const int32_t *nbr, const int64_t *ptr, int B,
                                       int64_t N, int k, int H, float *out, uint8_t *arg, dmet_stream_t stream)
{
    return gather_max_lds_impl(P, Q, nbr, nullptr, ptr, B, N, k, H, out, arg, false, stream);
}

extern "C" int dmet_gather_max_lds16_f32(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr_local,
                                         const int64_t *ptr, int B, int64_t N, int k, int H, float *out,
                                         uint8_t *arg, dmet_stream_t stream)
{
    return gather_max_lds_impl(P, Q, nbr, nbr_local, ptr, B, N, k, H, out, arg, false, stream);
}

extern "C" int dmet_gather_max_mixed_f32(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr_local,
                                         const int64_t *ptr, int B, int64_t N, int k, int H, float *out,
                                         uint8_t *arg, dmet_stream_t stream)
{
    // per-EVENT choice of the gather form on ragged batches: events whose Q slice fits the LDS image go through the
    // LDS-resident kernel, the others through the L2-form kernel; both read the row-major tables, each skips the
    // other's events
    if (!(H == 32 && (k == 8 || k == 16 || k == 32)) || N == 0 || B == 0)
        return dmet_gather_max_f32(P, Q, nbr, ptr, B, N, k, H, out, arg, stream);
    const int rc = gather_max_lds_impl(P, Q, nbr, nbr_local, ptr, B, N, k, H, out, arg, false, stream, 1);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int64_t blocks = (N + 31) / 32;
#define DMET_GMM_BIG(K4_)                                                                                            \
    do {                                                                                                             \
        if (arg)                                                                                                     \
            hipLaunchKernelGGL((gather_max_mlp_kernel<32, true, K4_, true>), dim3((unsigned)blocks), dim3(256), 0,   \
                               st, P, Q, nbr, N, out, arg, ptr, B, kLdsGatherRows);                                  \
        else                                                                                                         \
            hipLaunchKernelGGL((gather_max_mlp_kernel<32, false, K4_, true>), dim3((unsigned)blocks), dim3(256), 0,  \
                               st, P, Q, nbr, N, out, arg, ptr, B, kLdsGatherRows);                                  \
    } while (0)
    if (k == 8) DMET_GMM_BIG(2);
    else if (k == 16) DMET_GMM_BIG(4);
    else DMET_GMM_BIG(8);
#undef DMET_GMM_BIG
    DMET_LAUNCH_CHECK("gather_max_mlp_kernel (large events)");
    return 0;
}

extern "C" int dmet_gather_max_lds_sliced_f32(const float *P, const float *Q, const int32_t *nbr,
                                              const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k,
                                              int H, float *out, uint8_t *arg, dmet_stream_t stream)
{
    return gather_max_lds_impl(P, Q, nbr, nbr_local, ptr, B, N, k, H, out, arg, true, stream);
}

extern "C" int dmet_gather_max_lds_sliced_cap_f32(const float *P, const float *Q, const int32_t *nbr,
                                                  const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k,
                                                  int H, float *out, uint8_t *arg, int64_t max_nodes, dmet_stream_t stream)
{
    DMET_REQUIRE(max_nodes >= 0, "dmet_gather_max_lds_sliced_cap_f32: max_nodes=%lld", (long long)max_nodes);
    return gather_max_lds_impl(P, Q, nbr, nbr_local, ptr, B, N, k, H, out, arg, true, stream, 0, max_nodes);
}

extern "C" int dmet_gather_max_counted_lds_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt,
                                               const int64_t *ptr, int B, int64_t N, int k, int H, int pq_sliced,
                                               float *out, uint8_t *arg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_gather_max_counted_lds_f32: N out of range");
    DMET_REQUIRE(k >= 1 && k <= 255, "dmet_gather_max_counted_lds_f32: k=%d not in [1,255]", k);
    DMET_REQUIRE(H >= kSliceC && H % kSliceC == 0 && H <= DMET_MAX_H,
                 "dmet_gather_max_counted_lds_f32: H=%d must be a multiple of %d", H, kSliceC);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(P && Q && nbr && cnt && ptr && out, "dmet_gather_max_counted_lds_f32: null pointer");
    DMET_REQUIRE(aligned16(P) && aligned16(Q) && aligned16(out), "dmet_gather_max_counted_lds_f32: pointers must be 16-B aligned");
    const int nsl = H / kSliceC;
    const int64_t groups = (B + kNumXcd - 1) / kNumXcd;
    const int64_t blocks = groups * kNumXcd * nsl;
    hipStream_t st = as_stream(stream);
#define DMET_GCL(ARG_, SL_)                                                                                     \
    hipLaunchKernelGGL((gather_max_lds_counted_kernel<ARG_, SL_>), dim3((unsigned)blocks), dim3(kLdsGatherThreads), 0, \
                       st, P, Q, nbr, cnt, ptr, B, k, H, out, arg, N)
    if (arg) { if (pq_sliced) DMET_GCL(true, true); else DMET_GCL(true, false); }
    else { if (pq_sliced) DMET_GCL(false, true); else DMET_GCL(false, false); }
#undef DMET_GCL
    DMET_LAUNCH_CHECK("gather_max_lds_counted_kernel");
    return 0;
}

extern "C" int dmet_table_order_by_count(const int32_t *cnt, const int64_t *ptr, int B, int64_t N, int32_t *order,
                                         dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && B >= 0, "dmet_table_order_by_count: bad sizes");
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(cnt && ptr && order, "dmet_table_order_by_count: null pointer");
    hipLaunchKernelGGL(table_order_kernel, dim3((unsigned)B), dim3(1024), 0, as_stream(stream), cnt, ptr, B, order);
    DMET_LAUNCH_CHECK("table_order_kernel");
    return 0;
}

extern "C" int dmet_gather_max_local_j16_f32(const float *P, const float *Q, const uint16_t *nbr16, int stride16,
                                             const int32_t *cnt, const int32_t *order, const int64_t *ptr, int B,
                                             int64_t N, int kmax, int H, int pq_sliced, float *out, uint16_t *argj,
                                             dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_gather_max_local_j16_f32: N out of range");
    DMET_REQUIRE(kmax >= 1 && kmax <= 255 && stride16 >= kmax && stride16 % 8 == 0,
                 "dmet_gather_max_local_j16_f32: kmax=%d stride16=%d", kmax, stride16);
    DMET_REQUIRE(H >= kSliceC && H % kSliceC == 0 && H <= DMET_MAX_H,
                 "dmet_gather_max_local_j16_f32: H=%d must be a multiple of %d", H, kSliceC);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(P && Q && nbr16 && cnt && ptr && out, "dmet_gather_max_local_j16_f32: null pointer");
    DMET_REQUIRE(aligned16(P) && aligned16(Q) && aligned16(out) && aligned16(nbr16) &&
                     (reinterpret_cast<uintptr_t>(argj) & 7u) == 0,
                 "dmet_gather_max_local_j16_f32: pointers must be 16-B (argj: 8-B) aligned");
    const int nsl = H / kSliceC;
    const int64_t groups = (B + kNumXcd - 1) / kNumXcd;
    const int64_t blocks = groups * kNumXcd * nsl;
    hipStream_t st = as_stream(stream);
    uint8_t *a8 = reinterpret_cast<uint8_t *>(argj);
    if (!argj) {      // inference: the maximum alone
        if (pq_sliced)
            hipLaunchKernelGGL((gather_max_lds_counted_kernel<false, true, true, true>), dim3((unsigned)blocks),
                               dim3(kLdsGatherThreads), 0, st, P, Q, nullptr, cnt, ptr, B, kmax, H, out, a8, N, order, nbr16,
                               stride16);
        else
            hipLaunchKernelGGL((gather_max_lds_counted_kernel<false, false, true, true>), dim3((unsigned)blocks),
                               dim3(kLdsGatherThreads), 0, st, P, Q, nullptr, cnt, ptr, B, kmax, H, out, a8, N, order, nbr16,
                               stride16);
        DMET_LAUNCH_CHECK("gather_max_lds_counted_kernel (uint16 rows, no winners)");
        return 0;
    }
    if (pq_sliced)
        hipLaunchKernelGGL((gather_max_lds_counted_kernel<true, true, true, true>), dim3((unsigned)blocks),
                           dim3(kLdsGatherThreads), 0, st, P, Q, nullptr, cnt, ptr, B, kmax, H, out, a8, N, order, nbr16,
                           stride16);
    else
        hipLaunchKernelGGL((gather_max_lds_counted_kernel<true, false, true, true>), dim3((unsigned)blocks),
                           dim3(kLdsGatherThreads), 0, st, P, Q, nullptr, cnt, ptr, B, kmax, H, out, a8, N, order, nbr16,
                           stride16);
    DMET_LAUNCH_CHECK("gather_max_lds_counted_kernel (uint16 rows)");
    return 0;
}

extern "C" int dmet_gather_max_counted_lds_j16_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt,
                                                   const int32_t *order, const int64_t *ptr, int B, int64_t N, int k,
                                                   int H, int pq_sliced, float *out, uint16_t *argj,
                                                   dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_gather_max_counted_lds_j16_f32: N out of range");
    DMET_REQUIRE(k >= 1 && k <= 255, "dmet_gather_max_counted_lds_j16_f32: k=%d not in [1,255]", k);
    DMET_REQUIRE(H >= kSliceC && H % kSliceC == 0 && H <= DMET_MAX_H,
                 "dmet_gather_max_counted_lds_j16_f32: H=%d must be a multiple of %d", H, kSliceC);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(P && Q && nbr && cnt && ptr && out && argj, "dmet_gather_max_counted_lds_j16_f32: null pointer");
    DMET_REQUIRE(aligned16(P) && aligned16(Q) && aligned16(out) && (reinterpret_cast<uintptr_t>(argj) & 7u) == 0,
                 "dmet_gather_max_counted_lds_j16_f32: pointers must be 16-B (argj: 8-B) aligned");
    const int nsl = H / kSliceC;
    const int64_t groups = (B + kNumXcd - 1) / kNumXcd;
    const int64_t blocks = groups * kNumXcd * nsl;
    hipStream_t st = as_stream(stream);
    uint8_t *a8 = reinterpret_cast<uint8_t *>(argj);
    if (pq_sliced)
        hipLaunchKernelGGL((gather_max_lds_counted_kernel<true, true, true>), dim3((unsigned)blocks), dim3(kLdsGatherThreads),
                           0, st, P, Q, nbr, cnt, ptr, B, k, H, out, a8, N, order);
    else
        hipLaunchKernelGGL((gather_max_lds_counted_kernel<true, false, true>), dim3((unsigned)blocks),
                           dim3(kLdsGatherThreads), 0, st, P, Q, nbr, cnt, ptr, B, k, H, out, a8, N, order);
    DMET_LAUNCH_CHECK("gather_max_lds_counted_kernel (winner ids)");
    return 0;
}

extern "C" int dmet_edgeconv_linear_max_fwd_f32(const float *x, const int32_t *nbr, const int64_t *ptr, int B,
                                                int64_t N, int k, int Hin, int Hout, const float *W,
                                                const float *b, float *out, uint8_t *arg, void *ws,
                                                size_t ws_bytes, dmet_stream_t stream)
{
    if (N == 0) return 0;
    DMET_REQUIRE(ws && ws_bytes >= dmet_edgeconv_linear_workspace_bytes(N, Hout),
                 "dmet_edgeconv_linear_max_fwd_f32: workspace too small");
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u;
    float *P = reinterpret_cast<float *>(base);
    float *Q = P + (size_t)N * Hout;
    int rc = dmet_node_linear_split_f32(x, N, Hin, Hout, W, b, P, Q, stream);
    if (rc) return rc;
    return dmet_gather_max_f32(P, Q, nbr, ptr, B, N, k, Hout, out, arg, stream);
}

extern "C" int dmet_gather_max_bwd_f32(const float *g_out, const uint8_t *arg, const int32_t *rev_ptr,
                                       const int32_t *rev_slot, int64_t N, int k, int H, float *gQ,
                                       dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && k >= 1, "dmet_gather_max_bwd_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(g_out && arg && rev_ptr && rev_slot && gQ, "dmet_gather_max_bwd_f32: null pointer");
    hipStream_t st = as_stream(stream);
#define DMET_GB(HH)                                                                                         \
    do {                                                                                                    \
        constexpr int NPB = 256 / (HH / 4);                                                                 \
        const int64_t blocks = (N + NPB - 1) / NPB;                                                         \
        hipLaunchKernelGGL((gather_max_bwd_kernel<HH>), dim3((unsigned)blocks), dim3(256), 0, st, g_out, arg, \
                           rev_ptr, rev_slot, N, k, gQ);                                                    \
    } while (0)
    if (H == 32) DMET_GB(32);
    else if (H == 64) DMET_GB(64);
    else if (H == 128) DMET_GB(128);
    else if (H == 16) DMET_GB(16);
    else {
        set_error("dmet_gather_max_bwd_f32: unsupported H=%d", H);
        return -22;
    }
#undef DMET_GB
    DMET_LAUNCH_CHECK("gather_max_bwd_kernel");
    return 0;
}

extern "C" int dmet_edge_features_f32(const float *x, const int32_t *src, const int32_t *tgt, int64_t E, int H,
                                      float *feat, dmet_stream_t stream)
{
    DMET_REQUIRE(E >= 0 && H > 0 && H % 4 == 0, "dmet_edge_features_f32: H=%d must be a positive multiple of 4", H);
    if (E == 0) return 0;
    DMET_REQUIRE(x && src && tgt && feat, "dmet_edge_features_f32: null pointer");
    const int64_t total = E * (H / 4);
    hipLaunchKernelGGL(edge_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       x, src, tgt, E, H, feat);
    DMET_LAUNCH_CHECK("edge_features_kernel");
    return 0;
}

extern "C" int dmet_segment_max_f32(const float *msg, const int32_t *rowptr, int64_t N, int H, float *out,
                                    int32_t *arg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && H > 0, "dmet_segment_max_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(rowptr && out, "dmet_segment_max_f32: null pointer");
    const int64_t total = N * H;
    hipLaunchKernelGGL((segment_reduce_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), msg, rowptr, N, H, out, arg);
    DMET_LAUNCH_CHECK("segment_max_kernel");
    return 0;
}

extern "C" int dmet_segment_sum_f32(const float *msg, const int32_t *rowptr, int64_t N, int H, float *out,
                                    dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && H > 0, "dmet_segment_sum_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(rowptr && out, "dmet_segment_sum_f32: null pointer");
    const int64_t total = N * H;
    hipLaunchKernelGGL((segment_reduce_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), msg, rowptr, N, H, out, (int32_t *)nullptr);
    DMET_LAUNCH_CHECK("segment_sum_kernel");
    return 0;
}

extern "C" int dmet_segment_max_bwd_f32(const float *g_out, const int32_t *arg, const int32_t *rowptr, int64_t N,
                                        int H, float *g_msg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && H > 0, "dmet_segment_max_bwd_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(g_out && arg && rowptr && g_msg, "dmet_segment_max_bwd_f32: null pointer");
    const int64_t total = N * H;
    hipLaunchKernelGGL((segment_reduce_bwd_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), g_out, arg, rowptr, N, H, g_msg);
    DMET_LAUNCH_CHECK("segment_max_bwd_kernel");
    return 0;
}

extern "C" int dmet_segment_sum_bwd_f32(const float *g_out, const int32_t *rowptr, int64_t N, int H,
                                        float *g_msg, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && H > 0, "dmet_segment_sum_bwd_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(g_out && rowptr && g_msg, "dmet_segment_sum_bwd_f32: null pointer");
    const int64_t total = N * H;
    hipLaunchKernelGGL((segment_reduce_bwd_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), g_out, (const int32_t *)nullptr, rowptr, N, H, g_msg);
    DMET_LAUNCH_CHECK("segment_sum_bwd_kernel");
    return 0;
}

extern "C" int dmet_edge_features_bwd_f32(const float *g_feat, const int32_t *rowptr, const int32_t *srcptr,
                                          const int32_t *srcperm, int64_t N, int H, float *gx,
                                          dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && H > 0, "dmet_edge_features_bwd_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(rowptr && srcptr && gx, "dmet_edge_features_bwd_f32: null pointer");
    const int64_t total = N * H;
    hipLaunchKernelGGL(edge_features_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), g_feat, rowptr, srcptr, srcperm, N, H, gx);
    DMET_LAUNCH_CHECK("edge_features_bwd_kernel");
    return 0;
}
