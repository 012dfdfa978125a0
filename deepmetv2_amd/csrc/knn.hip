// knn.hip -- K1: exact brute-force kNN graph build over ragged events, one launch (gfx950).
//
// Replaces torch_cluster.knn_graph / knn (call sites /root/reference/model/graph_met_network.py:63,
// model/dynamic_reduction_network.py:86,94).  Results are bit-identical to oracle/dmet_oracle.c:
//   R1  d(i,j) = chain of fmaf(diff, diff, acc) over the feature index, fp32, diff = x[j,c]-x[i,c]
//   R2  top-k by (d, j) lexicographic order == upstream's strict-'>' insertion in ascending j
//
// Work decomposition (fp32-VALU bound: 2 VALU ops per (query, candidate, feature)):
//   * one 64-lane wavefront per workgroup; every lane OWNS TQ query nodes whose features sit in registers;
//   * candidate rows of the event are staged into LDS in tiles and read back as wave-uniform (broadcast)
//     ds_read_b128, so one LDS read feeds 64*TQ lanes-queries (TQ = 2 keeps the LDS pipe at ~50%);
//   * selection is deferred: a lane whose distance beats its current k-th best appends (d, j) to its private
//     LDS queue; when any lane's queue is nearly full the whole wave drains its queues into the sorted top-k
//     lists, which live in an L2-resident global workspace between drains (keeps VGPRs for the distance loop);
//   * queries are assigned by global node index, so a wavefront may straddle two events: it then sweeps the
//     union of their candidate ranges and masks per lane (taken only by boundary wavefronts).
#include "common.h"

namespace dmet {
namespace {

constexpr int kTileC = 32;   // candidates per LDS tile
constexpr int kQMax = 12;    // per-lane pending queue capacity

template <int DP, int KP, int TQ>
struct KnnShared {
    float4 tile[(kTileC + 2) * DP / 4];  // +2 rows: the pipelined sweep reads up to two rows ahead
    uint2 queue[TQ][kQMax][kWave];
};

// Drain lane-private queue `qslot` into the sorted list of the lane's query (list kept in ws between drains).
template <int KP>
__device__ __attribute__((noinline)) float drain_queue(const uint2 (*queue)[kWave], int lane, int cnt, bool fresh,
                                                       bool valid, float *__restrict__ ld,
                                                       int32_t *__restrict__ lj)
{
    float d[KP];
    int32_t j[KP];
    if (fresh || !valid) {
#pragma unroll
        for (int p = 0; p < KP; ++p) { d[p] = kKnnSentinel; j[p] = -1; }
    } else {
#pragma unroll
        for (int p = 0; p < KP; p += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(ld + p);
            const int4 w = *reinterpret_cast<const int4 *>(lj + p);
            d[p] = v.x; d[p + 1] = v.y; d[p + 2] = v.z; d[p + 3] = v.w;
            j[p] = w.x; j[p + 1] = w.y; j[p + 2] = w.z; j[p + 3] = w.w;
        }
    }
    int maxcnt = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, off, 64));
    for (int s = 0; s < maxcnt; ++s) {
        if (s < cnt) {
            const uint2 e = queue[s][lane];
            const float nd = __uint_as_float(e.x);
            const int32_t nj = (int32_t)e.y;
            if (nd < d[KP - 1]) {
                // sorted insert; first position with d[p] > nd (strict) takes the new entry (R2)
#pragma unroll
                for (int p = KP - 1; p >= 1; --p) {
                    const bool mq = d[p - 1] > nd;
                    const bool mp = d[p] > nd;
                    const float dn = mq ? d[p - 1] : (mp ? nd : d[p]);
                    const int32_t jn = mq ? j[p - 1] : (mp ? nj : j[p]);
                    d[p] = dn;
                    j[p] = jn;
                }
                if (d[0] > nd) { d[0] = nd; j[0] = nj; }
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int p = 0; p < KP; p += 4) {
            *reinterpret_cast<float4 *>(ld + p) = make_float4(d[p], d[p + 1], d[p + 2], d[p + 3]);
            *reinterpret_cast<int4 *>(lj + p) = make_int4(j[p], j[p + 1], j[p + 2], j[p + 3]);
        }
    }
    return d[KP - 1];  // the new k-th best = the lane's new admission threshold
}

template <int DP, int KP, int TQ, bool EXACT_D>
__global__ __launch_bounds__(kWave, 2) void knn_kernel(const float *__restrict__ x,
                                                     const int64_t *__restrict__ ptr, int B, int64_t N, int D,
                                                     int k, int32_t *__restrict__ nbr,
                                                     float *__restrict__ dist, float *__restrict__ wsd,
                                                     int32_t *__restrict__ wsj)
{
    __shared__ KnnShared<DP, KP, TQ> sh;
    const int lane = threadIdx.x;
    const int64_t q_first = (int64_t)blockIdx.x * (kWave * TQ);
    if (q_first >= N) return;
    const int64_t q_last = min(N, q_first + kWave * TQ) - 1;

    // wave-uniform candidate range = union of the events this wavefront's queries live in
    const int b_first = find_event(ptr, B, q_first);
    const int b_last = find_event(ptr, B, q_last);
    const int clo = (int)ptr[b_first];
    const int chi = (int)ptr[b_last + 1];
    const bool single = (b_first == b_last);

    float q[TQ][DP];
    int64_t qi[TQ];
    bool valid[TQ];
    int lo[TQ], hi[TQ];
    float tau[TQ];
    int cnt[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        qi[t] = q_first + t * kWave + lane;
        valid[t] = qi[t] < N;
        const int64_t qq = valid[t] ? qi[t] : q_last;
        if (single) { lo[t] = clo; hi[t] = chi; }
        else { const int b = find_event(ptr, B, qq); lo[t] = (int)ptr[b]; hi[t] = (int)ptr[b + 1]; }
        if (!valid[t]) { lo[t] = 0; hi[t] = 0; }
        if (EXACT_D) {
#pragma unroll
            for (int c = 0; c < DP; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + qq * DP + c);
                q[t][c] = v.x; q[t][c + 1] = v.y; q[t][c + 2] = v.z; q[t][c + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < DP; ++c) q[t][c] = (c < D) ? x[qq * D + c] : 0.0f;
        }
        tau[t] = kKnnSentinel;
        cnt[t] = 0;
    }
    unsigned fresh = (1u << TQ) - 1u;  // wave-uniform: list of slot t not yet written to ws

    constexpr int kLd4 = kTileC * DP / 4;            // float4s per tile
    constexpr int kLdPerLane = (kLd4 + kWave - 1) / kWave;


    for (int c0 = clo; c0 < chi; c0 += kTileC) {
        const int cntc = min(kTileC, chi - c0);
        __syncthreads();  // every lane is done reading the previous tile
        if (EXACT_D) {
            const float4 *g = reinterpret_cast<const float4 *>(x + (int64_t)c0 * DP);
            const int n4 = cntc * (DP / 4);
#pragma unroll
            for (int m = 0; m < kLdPerLane; ++m) {
                const int idx = lane + m * kWave;
                if (idx < kLd4) sh.tile[idx] = (idx < n4) ? g[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            float *tl = reinterpret_cast<float *>(sh.tile);
            for (int e = lane; e < kTileC * DP; e += kWave) {
                const int c = e / DP, dd = e - c * DP;
                tl[e] = (c < cntc && dd < D) ? x[(int64_t)(c0 + c) * D + dd] : 0.0f;
            }
        }
        __syncthreads();

        // software-pipelined sweep: candidate rows A (cc) and B (cc+1) alternate between two register sets
        float4 rowA[DP / 4], rowB[DP / 4];
#pragma unroll
        for (int c4 = 0; c4 < DP / 4; ++c4) rowA[c4] = sh.tile[c4];
        for (int cc = 0; cc < cntc; cc += 2) {
            float accA[TQ], accB[TQ];
#pragma unroll
            for (int c4 = 0; c4 < DP / 4; ++c4) rowB[c4] = sh.tile[(cc + 1) * (DP / 4) + c4];  // broadcast reads
#pragma unroll
            for (int t = 0; t < TQ; ++t) accA[t] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < DP / 4; ++c4) {
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    float df;
                    df = rowA[c4].x - q[t][4 * c4 + 0]; accA[t] = __builtin_fmaf(df, df, accA[t]);
                    df = rowA[c4].y - q[t][4 * c4 + 1]; accA[t] = __builtin_fmaf(df, df, accA[t]);
                    df = rowA[c4].z - q[t][4 * c4 + 2]; accA[t] = __builtin_fmaf(df, df, accA[t]);
                    df = rowA[c4].w - q[t][4 * c4 + 3]; accA[t] = __builtin_fmaf(df, df, accA[t]);
                }
            }
#pragma unroll
            for (int c4 = 0; c4 < DP / 4; ++c4) rowA[c4] = sh.tile[(cc + 2) * (DP / 4) + c4];
#pragma unroll
            for (int t = 0; t < TQ; ++t) accB[t] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < DP / 4; ++c4) {
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    float df;
                    df = rowB[c4].x - q[t][4 * c4 + 0]; accB[t] = __builtin_fmaf(df, df, accB[t]);
                    df = rowB[c4].y - q[t][4 * c4 + 1]; accB[t] = __builtin_fmaf(df, df, accB[t]);
                    df = rowB[c4].z - q[t][4 * c4 + 2]; accB[t] = __builtin_fmaf(df, df, accB[t]);
                    df = rowB[c4].w - q[t][4 * c4 + 3]; accB[t] = __builtin_fmaf(df, df, accB[t]);
                }
            }
            // selection; a row past the event's end (odd tail / tile padding) has j >= hi and never passes
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = c0 + cc + u;
                    const float dj = u ? accB[t] : accA[t];
                    bool pass = dj < tau[t];
                    if (single) pass = pass && (j < hi[t]);
                    else pass = pass && (j >= lo[t]) && (j < hi[t]);
                    if (pass) {
                        sh.queue[t][cnt[t]][lane] = make_uint2(__float_as_uint(dj), (unsigned)j);
                        cnt[t]++;
                    }
                }
                if (__any(cnt[t] > kQMax - 2)) {
                    const int64_t qq = valid[t] ? qi[t] : 0;
                    tau[t] = drain_queue<KP>(sh.queue[t], lane, cnt[t], (fresh >> t) & 1u, valid[t], wsd + qq * KP,
                                             wsj + qq * KP);
                    cnt[t] = 0;
                    fresh &= ~(1u << t);
                }
            }
        }
    }

    // final drain + output of the first k entries
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        const int64_t qq = valid[t] ? qi[t] : 0;
        tau[t] = drain_queue<KP>(sh.queue[t], lane, cnt[t], (fresh >> t) & 1u, valid[t], wsd + qq * KP,
                                 wsj + qq * KP);
        if (valid[t]) {
            if (k == KP) {
#pragma unroll
                for (int p = 0; p < KP; p += 4) {
                    *reinterpret_cast<float4 *>(dist + qq * KP + p) =
                        *reinterpret_cast<const float4 *>(wsd + qq * KP + p);
                    *reinterpret_cast<int4 *>(nbr + qq * KP + p) =
                        *reinterpret_cast<const int4 *>(wsj + qq * KP + p);
                }
            } else {
                for (int p = 0; p < k; ++p) {
                    dist[qq * k + p] = wsd[qq * KP + p];
                    nbr[qq * k + p] = wsj[qq * KP + p];
                }
            }
        }
    }
}

template <int DP, int KP>
int launch_knn(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr, float *dist,
               float *wsd, int32_t *wsj, hipStream_t st)
{
    constexpr int TQ = (DP <= 32) ? 2 : 1;
    const int64_t per_block = kWave * TQ;
    const int64_t blocks = (N + per_block - 1) / per_block;
    if (D == DP && aligned16(x)) {
        hipLaunchKernelGGL((knn_kernel<DP, KP, TQ, true>), dim3((unsigned)blocks), dim3(kWave), 0, st, x, ptr, B, N,
                           D, k, nbr, dist, wsd, wsj);
    } else {
        hipLaunchKernelGGL((knn_kernel<DP, KP, TQ, false>), dim3((unsigned)blocks), dim3(kWave), 0, st, x, ptr, B,
                           N, D, k, nbr, dist, wsd, wsj);
    }
    DMET_LAUNCH_CHECK("knn_kernel");
    return 0;
}

template <int DP>
int dispatch_k(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr, float *dist,
               float *wsd, int32_t *wsj, hipStream_t st)
{
    if (k <= 8) return launch_knn<DP, 8>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    if (k <= 16) return launch_knn<DP, 16>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    if (k <= 32) return launch_knn<DP, 32>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    return launch_knn<DP, 64>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
}

inline int padded_k(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64; }

// ---- radius graph (N1): first max_nbr candidates in ascending index with d < r^2 ------------------------
// One lane per query, candidates broadcast from LDS exactly as above; no selection state beyond a counter.
template <int DP>
__global__ __launch_bounds__(kWave) void radius_kernel(const float *__restrict__ x,
                                                        const int64_t *__restrict__ ptr, int B, int64_t N, int D,
                                                        float r2, int max_nbr, int32_t *__restrict__ nbr,
                                                        int32_t *__restrict__ cntout)
{
    __shared__ float tile[kTileC * DP];
    const int lane = threadIdx.x;
    const int64_t q_first = (int64_t)blockIdx.x * kWave;
    if (q_first >= N) return;
    const int64_t q_last = min(N, q_first + kWave) - 1;
    const int b_first = find_event(ptr, B, q_first);
    const int b_last = find_event(ptr, B, q_last);
    const int clo = (int)ptr[b_first];
    const int chi = (int)ptr[b_last + 1];
    const int64_t qi = q_first + lane;
    const bool valid = qi < N;
    const int64_t qq = valid ? qi : q_last;
    int lo = 0, hi = 0;
    if (valid) { const int b = find_event(ptr, B, qq); lo = (int)ptr[b]; hi = (int)ptr[b + 1]; }
    float q[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) q[c] = (c < D) ? x[qq * D + c] : 0.0f;
    int cnt = 0;
    int32_t *row = nbr + qq * max_nbr;
    for (int c0 = clo; c0 < chi; c0 += kTileC) {
        const int cntc = min(kTileC, chi - c0);
        __syncthreads();
        for (int e = lane; e < kTileC * DP; e += kWave) {
            const int c = e / DP, dd = e - c * DP;
            tile[e] = (c < cntc && dd < D) ? x[(int64_t)(c0 + c) * D + dd] : 0.0f;
        }
        __syncthreads();
        for (int cc = 0; cc < cntc; ++cc) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < DP; ++c) {
                const float df = tile[cc * DP + c] - q[c];
                acc = __builtin_fmaf(df, df, acc);
            }
            const int j = c0 + cc;
            if (valid && j >= lo && j < hi && acc < r2 && cnt < max_nbr) { row[cnt] = j; ++cnt; }
        }
    }
    if (valid) {
        for (int p = cnt; p < max_nbr; ++p) row[p] = -1;
        cntout[qi] = cnt;
    }
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_knn_workspace_bytes(int64_t N, int B, int D, int k)
{
    (void)B; (void)D;
    if (N <= 0 || k <= 0 || k > DMET_MAX_K) return 0;
    const size_t per = (size_t)padded_k(k) * (sizeof(float) + sizeof(int32_t));
    return (size_t)N * per + 256;
}

extern "C" int dmet_knn_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                            float *dist, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_knn_f32: N=%lld out of range", (long long)N);
    DMET_REQUIRE(B >= 0, "dmet_knn_f32: B=%d", B);
    DMET_REQUIRE(k >= 1 && k <= DMET_MAX_K, "dmet_knn_f32: k=%d not in [1,%d]", k, DMET_MAX_K);
    DMET_REQUIRE(D >= 1 && D <= DMET_MAX_KNN_DIM, "dmet_knn_f32: D=%d not in [1,%d]", D, DMET_MAX_KNN_DIM);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(x && ptr && nbr && dist && ws, "dmet_knn_f32: null pointer");
    DMET_REQUIRE(ws_bytes >= dmet_knn_workspace_bytes(N, B, D, k), "dmet_knn_f32: workspace too small");
    const int KP = padded_k(k);
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u;
    float *wsd = reinterpret_cast<float *>(base);
    int32_t *wsj = reinterpret_cast<int32_t *>(base + (size_t)N * KP * sizeof(float));
    hipStream_t st = as_stream(stream);
    if (D <= 4) return dispatch_k<4>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    if (D <= 8) return dispatch_k<8>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    if (D <= 16) return dispatch_k<16>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    if (D <= 32) return dispatch_k<32>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
    return dispatch_k<64>(x, ptr, B, N, D, k, nbr, dist, wsd, wsj, st);
}

extern "C" int dmet_radius_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr,
                               int32_t *nbr, int32_t *cnt, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_radius_f32: N out of range");
    DMET_REQUIRE(D >= 1 && D <= 8, "dmet_radius_f32: D=%d not in [1,8]", D);
    DMET_REQUIRE(max_nbr >= 1, "dmet_radius_f32: max_nbr=%d", max_nbr);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(x && ptr && nbr && cnt, "dmet_radius_f32: null pointer");
    const float r2 = r * r;
    const int64_t blocks = (N + kWave - 1) / kWave;
    hipStream_t st = as_stream(stream);
    if (D <= 2)
        hipLaunchKernelGGL((radius_kernel<2>), dim3((unsigned)blocks), dim3(kWave), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, nbr, cnt);
    else if (D <= 4)
        hipLaunchKernelGGL((radius_kernel<4>), dim3((unsigned)blocks), dim3(kWave), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, nbr, cnt);
    else
        hipLaunchKernelGGL((radius_kernel<8>), dim3((unsigned)blocks), dim3(kWave), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, nbr, cnt);
    DMET_LAUNCH_CHECK("radius_kernel");
    return 0;
}
