// knn.hip -- K1: kNN graph build over ragged events (gfx950), and the radius graph (N1).
//
// Replaces torch_cluster.knn_graph / knn (call sites /root/reference/model/graph_met_network.py:63,
// model/dynamic_reduction_network.py:86,94).  Results are bit-identical to oracle/dmet_oracle.c:
//   R1  d(i,j) = chain of fmaf(diff, diff, acc) over the feature index, fp32, diff = x[j,c]-x[i,c]
//   R2  top-k by (d, j) lexicographic order == upstream's strict-'>' insertion in ascending j
// R1 / R2 define the RESULT, not the work done on pairs that cannot win.  Three paths, one stream, no host sync:
//   * D = 32 (the model's shape) or 64 (the DRN's), k <= 20: a matrix-core FILTER ranks all pairs approximately
//     (key = |x_j|^2 - 2 x_i.x_j), keeps a certified superset of every query's neighbours, and only those go through
//     the exact R1 chain and the (d, j) order.  Second form (filter2_wave, events of 2048..65536 nodes): single-term
//     fp16 operands on v_mfma_f32_32x32x16_f16 (2 MFMAs per 32 x 32 block and 32 features; the certificate carries the
//     fp16 rounding), per-tile hit masks, threshold from tile minima, a second in-wavefront sweep for queries that
//     miss the certificate by the slack alone; first form (filter1_wave): bf16 split x = h + m on
//     v_mfma_f32_32x32x16_bf16 (6 MFMAs per block), per-key queue; one launch (knn_filter12_kernel), each wavefront
//     takes the form its event calls for.  A query whose certificate does not hold is recomputed exactly.
//   * everything else, and the recomputation: the exact kernel below (knn_kernel).
//
// The exact kernel is fp32-VALU bound (a subtract and an fma per (query, candidate, feature); the difference form
// cannot go to the matrix cores without changing the rounding).  Measured on MI355X (tools/valu_micro.hip) this
// instruction mix saturates at ~75-80 TFLOP/s (3 flop per element) and needs packed math plus >= 3 wavefronts per
// SIMD to get there, which shapes the design:
//   * a work item = one 64-lane wavefront (workgroups are 4 independent wavefronts); every lane OWNS 2 query nodes
//     whose features sit in registers as float2 pairs, so the inner loop is v_pk_add_f32 / v_pk_fma_f32 (each half
//     is an exact IEEE op: same bits);
//   * candidate rows are staged into LDS in tiles and read back as wave-uniform (broadcast) ds_read_b128; two
//     candidates are in flight per iteration (two independent fma chains per lane);
//   * selection is deferred: a lane whose distance beats its current k-th best appends (d, j) to its private LDS
//     queue; when any lane's queue is nearly full the whole wave drains its queues into the sorted top-k lists,
//     which live in an L2-resident global workspace between drains (keeps VGPRs for the distance loop);
//   * query tiles never straddle events (a straddling wavefront would sweep two events: a 2x straggler); the
//     tile -> event map is a device-side plan (no host sync), events ordered longest first;
//   * load balance: with T tiles on S SIMDs the last (T mod S) tiles would leave most of the chip idle for a whole
//     sweep.  Those tail tiles are split over the candidate range into `split` sub-sweeps (dispatched last) whose
//     partial top-k lists a small merge kernel combines.
#include <stdlib.h>

#include "common.h"
#include "nls_body.h"

namespace dmet {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kTileC = 32;   // candidates per LDS tile
constexpr int kQMax = 8;     // per-lane pending queue capacity
constexpr int kMaxSplit = 8; // tail tiles are split into at most this many candidate sub-sweeps
constexpr int kWavesPerGroup = 4;  // independent wavefronts per workgroup (one per SIMD of the CU)

// LDS hand-off between the lanes of ONE wavefront: LDS requests of a wave execute in order, so only the compiler
// has to be kept from moving accesses across this point (no workgroup barrier: the group's waves are independent).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Four features x two candidates (A, B) x the lane's two queries in ONE asm block of 8 v_pk_add_f32 + 8 v_pk_fma_f32.
//   v_pk_add_f32 t, cand_pair, q  op_sel -> (c, c) + (-q.x, -q.y): the candidate feature is the low or high half of a
//   register pair, broadcast to both result halves with op_sel/op_sel_hi; the query pair is negated with neg_lo/neg_hi
//   (c + (-q) is the same IEEE result as c - q).
// Why asm: hipcc materialises every broadcast pair with v_mov (49 extra VALU per 128 useful ones) and serialises the
// fma chains behind the packed-fp32 RAW wait state; here each accumulator's chain stays in feature order (R1) and
// dependent packed ops are always separated by an independent one.
#define DMET_PKSUB_LO " op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define DMET_PKSUB_HI " op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
__device__ __forceinline__ void pk_dist_step4(f2 &accA, f2 &accB, f2 vxy, f2 vzw, f2 wxy, f2 wzw, f2 q0, f2 q1, f2 q2,
                                              f2 q3)
{
    f2 t0, t1, t2, t3;
    asm("v_pk_add_f32 %[t0], %[vxy], %[q0]" DMET_PKSUB_LO
        "v_pk_add_f32 %[t1], %[wxy], %[q0]" DMET_PKSUB_LO
        "v_pk_add_f32 %[t2], %[vxy], %[q1]" DMET_PKSUB_HI
        "v_pk_add_f32 %[t3], %[wxy], %[q1]" DMET_PKSUB_HI
        "v_pk_fma_f32 %[a], %[t0], %[t0], %[a]\n\t"
        "v_pk_fma_f32 %[b], %[t1], %[t1], %[b]\n\t"
        "v_pk_add_f32 %[t0], %[vzw], %[q2]" DMET_PKSUB_LO
        "v_pk_add_f32 %[t1], %[wzw], %[q2]" DMET_PKSUB_LO
        "v_pk_fma_f32 %[a], %[t2], %[t2], %[a]\n\t"
        "v_pk_fma_f32 %[b], %[t3], %[t3], %[b]\n\t"
        "v_pk_add_f32 %[t2], %[vzw], %[q3]" DMET_PKSUB_HI
        "v_pk_add_f32 %[t3], %[wzw], %[q3]" DMET_PKSUB_HI
        "v_pk_fma_f32 %[a], %[t0], %[t0], %[a]\n\t"
        "v_pk_fma_f32 %[b], %[t1], %[t1], %[b]\n\t"
        "v_pk_fma_f32 %[a], %[t2], %[t2], %[a]\n\t"
        "v_pk_fma_f32 %[b], %[t3], %[t3], %[b]"
        : [a] "+v"(accA), [b] "+v"(accB), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [vxy] "v"(vxy), [vzw] "v"(vzw), [wxy] "v"(wxy), [wzw] "v"(wzw), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2),
          [q3] "v"(q3));
}

template <int DP, int TQ>
struct KnnShared {
    float4 tile[(kTileC + 1) * DP / 4];  // +1 row kept at +inf: the paired sweep reads one row past the tile's last
    uint2 queue[TQ][kQMax][kWave];
};

// Drain a lane-private queue into the sorted list of the lane's query (list kept in ws between drains).
// Returns the lane's new admission threshold (its k-th best distance).
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
    return d[KP - 1];
}

#ifdef DMET_KNN_STAMP
// experiment only (tools/knn_trace.hip): per-workgroup start/end realtime stamps and hardware placement
__device__ unsigned long long g_knn_stamps[1 << 16][4];
#endif

// events the second matrix-core filter form sweeps (filter2_wave); the plan counts the others
// Smaller events take the first form: the threshold of the second form is the M-th smallest of the event's tile minima (M = 22
// tiles = 704 nodes at least), and below ~800 nodes the first form is the faster one (tools/knn_small_events.py: 64 events
// of 704 nodes 155 us against 241, of 800 nodes 168 against 133, of 1000 nodes 188 against 109, of 2000 nodes 265 against 140).
// Until round 3 the boundary stood at 2048 -- twice the build time for events of 1000..2047 nodes.
#ifndef DMET_F2_MIN_NODES
#define DMET_F2_MIN_NODES 800
#endif
constexpr int kF2MinNodes = DMET_F2_MIN_NODES;
constexpr int kFilterMaxSplit = 2;  // tail balancing of the filter: at most 2 candidate sub-sweeps (the re-rank assumes 2).
#ifndef DMET_F2_SPLIT_MIN_NODES
#define DMET_F2_SPLIT_MIN_NODES 2560
#endif
constexpr int kF2SplitMinNodes = DMET_F2_SPLIT_MIN_NODES;   // smaller second-form events are never cut into candidate sub-sweeps (see filter2_wave)
constexpr int kF2MaxNodes = 65536;  // tile number must fit 11 bits

// Device-side launch plan (no host synchronisation): tiles never straddle events, so tile -> event needs a prefix.
struct KnnPlan {
    int total_tiles;   // sum_b ceil(n_b / tile_queries)
    int n_full;        // tiles 0..n_full-1 sweep their whole event in one workgroup
    int split;         // tiles n_full.. are cut into `split` candidate sub-sweeps each (tail balancing)
    int form1_events;  // non-empty events outside the second filter form's size range (its kernels exit at once if 0)
};

struct KnnArgs {
    const float *x;
    const int64_t *ptr;
    int B;
    int64_t N;
    int D, k;
    int32_t *nbr;
    float *dist;
    uint16_t *nbr16;       // optional [N][k]: the same table as event-local ids (0xFFFF = none; meaningful for events
                           // of at most 65535 nodes)
    float *wsd;            // [N][KP] running lists of whole-sweep tiles
    int32_t *wsj;
    const KnnPlan *plan;
    const int32_t *order;     // [B] events longest first
    const int32_t *tile_ptr;  // [B+1] exclusive prefix of per-position tile counts
    float *psd;            // [(tile-n_full)*tile_queries + slot][split][KP] partial lists of split tiles
    int32_t *psj;
    const int32_t *flags;  // optional [tiles] = number of uncertified queries per tile (matrix-core path): only
    int flag_min;          // tiles with flags[tile] >= flag_min are computed here
    const int32_t *any;    // optional: total number of uncertified queries (0: nothing to do for any workgroup)
    // per-query fallback of the matrix-core path, run by the first requery_groups workgroups of the same launch (32
    // features): the flagged queries in flagging order, the event -> position map of the plans, queries per exact tile
    const int32_t *qlist;
    const int32_t *pos_of;
    int requery_groups;
    int requery_tile_queries;
};

__device__ __forceinline__ uint16_t local_id16(int32_t j, int ev_lo)
{
    return (uint16_t)(j >= 0 ? (unsigned)(j - ev_lo) : 0xFFFFu);
}

// One workgroup: events ordered LONGEST FIRST (a tile of an n-node event costs n candidates, so big events go out
// first and the split tail consists of the smallest ones), per-position tile counts -> exclusive prefix, then the
// tail-splitting plan for `simds` SIMDs: whole sweeps for the largest multiple of the SIMD count, the remaining
// tiles cut into sub-sweeps.  order[p] = event at position p; tile_ptr is indexed by position.
constexpr int kMaxSortedEvents = 4096;  // beyond this the O(B^2) ranking is skipped (identity order)

struct KnnPlanOut {
    int tile_queries, simds, max_split;
    int32_t *order, *pos_of, *tile_ptr;
    KnnPlan *plan;
};

// Event order of the plans.  Ranks are by size, longest first.  Workgroups are dealt round-robin over the 8 XCDs
// (each with its own L2), and the matrix-core filter maps each XCD's workgroups to ONE contiguous eighth of the tile
// list so that the tiles of an event -- which all sweep the same candidate records -- run on one XCD and re-read them
// from its L2 (the records of a batch do not fit any L2: left to round-robin every sweep comes from the Infinity
// Cache).  For the eighths to carry equal work on ragged batches the ranks are dealt to 8 bins in snake order
// (0..7, 7..0, ...) and the bins concatenated: every bin, hence every XCD, gets the same mix of event sizes and still
// runs its own events longest first.  Small batches keep the plain order.
__device__ __forceinline__ int xcd_dealt_position(int rank, int B)
{
    if (B < 4 * kNumXcd) return rank;
    const int c = rank % (2 * kNumXcd), g = rank / (2 * kNumXcd);
    const int bin = c < kNumXcd ? c : 2 * kNumXcd - 1 - c;
    const int idx = 2 * g + (c >= kNumXcd ? 1 : 0);
    const int rem = B % (2 * kNumXcd), full = B / (2 * kNumXcd);
    int start = 0;
    for (int b = 0; b < bin; ++b)      // sizes of the earlier bins
        start += 2 * full + (b < min(rem, kNumXcd) ? 1 : 0) + ((rem > kNumXcd && b > 2 * kNumXcd - 1 - rem) ? 1 : 0);
    return start + idx;
}

// blockIdx.x selects one of up to two plans (exact kernel: 128-query tiles; matrix-core filter: 64-query tiles)
__device__ __forceinline__ void knn_plan_body(const int64_t *__restrict__ ptr, int B, const KnnPlanOut &o)
{
    const int tile_queries = o.tile_queries, simds = o.simds, max_split = o.max_split;
    int32_t *__restrict__ order = o.order, *__restrict__ pos_of = o.pos_of, *__restrict__ tile_ptr = o.tile_ptr;
    KnnPlan *__restrict__ plan = o.plan;
    __shared__ int part[256];
    const int tid = threadIdx.x;
    if (B <= kMaxSortedEvents) {
        for (int b = tid; b < B; b += 256) {
            const int64_t nb = ptr[b + 1] - ptr[b];
            int rank = 0;
            for (int c = 0; c < B; ++c) {
                const int64_t nc = ptr[c + 1] - ptr[c];
                rank += (nc > nb || (nc == nb && c < b)) ? 1 : 0;
            }
            const int p = xcd_dealt_position(rank, B);
            order[p] = b;
            pos_of[b] = p;
        }
    } else {
        for (int b = tid; b < B; b += 256) { order[b] = b; pos_of[b] = b; }
    }
    __syncthreads();
    const int chunk = (B + 255) / 256;
    const int lo = min(B, tid * chunk), hi = min(B, lo + chunk);
    int sum = 0, nf1 = 0;
    for (int p = lo; p < hi; ++p) {
        const int b = order[p];
        const int64_t nb = ptr[b + 1] - ptr[b];
        sum += (int)((nb + tile_queries - 1) / tile_queries);
        nf1 += (nb > 0 && !(nb >= kF2MinNodes && nb <= kF2MaxNodes)) ? 1 : 0;
    }
    // exclusive prefix of the per-thread tile counts and the number of first-form events: wavefront scans + four
    // wavefront totals (a serial walk of thread 0 over 256 LDS cells was a third of the prep launch: it sits on the
    // critical path of every build)
    __shared__ int wave_tot[4], wave_f1[4];
    const int lane = tid & 63, wv = tid >> 6;
    int incl = sum, f1 = nf1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) f1 += __shfl_xor(f1, off, 64);
    if (lane == 63) wave_tot[wv] = incl;
    if (lane == 0) wave_f1[wv] = f1;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wave_tot[w];
    part[tid] = base + incl - sum;
    if (tid == 0) {
        const int form1_events = wave_f1[0] + wave_f1[1] + wave_f1[2] + wave_f1[3];
        const int tiles = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        int n_full = tiles, split = 1;
        // the tiles beyond the last full round of `simds` -- ALL tiles when the batch is small -- are cut into
        // candidate sub-sweeps so that they still fill the chip (unless the events are so small that a sub-sweep
        // would be a handful of candidates)
        const int full = (tiles / simds) * simds;
        const int rem = tiles - full;
        if (rem > 0 && B > 0 && (ptr[B] - ptr[0]) >= (int64_t)512 * B) {
            int f = simds / rem;
            if (f > max_split) f = max_split;
            if (f >= 2) { n_full = full; split = f; }
        }
        plan->total_tiles = tiles; plan->n_full = n_full; plan->split = split; plan->form1_events = form1_events;
        tile_ptr[B] = tiles;
    }
    __syncthreads();
    int run = part[tid];
    for (int p = lo; p < hi; ++p) {
        const int b = order[p];
        tile_ptr[p] = run;
        run += (int)((ptr[b + 1] - ptr[b] + tile_queries - 1) / tile_queries);
    }
    // When does cutting the tail's sweeps in two pay for SECOND-form events?  Each half has its own threshold from half
    // the tile minima (looser: more candidates to re-rank, a fixed cost per piece), so it pays only while the pieces
    // find idle SIMDs -- few tail tiles -- and the halves still have a threshold to speak of -- large events
    // (tools/knn_small_events.py, 2048 wavefront slots: 376 tiles of 3000-node events 106 us split / 154 whole, 568
    // tiles of 4500-node events 132 / 188, but 512 tiles of 2048-node events 115 / 106, 752 tiles of 3000-node events
    // 127 / 119, 1024 tiles of 2048-node events 140 / 118).  The tail holds the SMALLEST events (longest-first order),
    // so its first tile's event bounds them all.  First-form tails (events below kF2MinNodes) and the exact kernel's
    // plan keep the rule above.
    if (o.max_split == kFilterMaxSplit) {
        __syncthreads();
        if (tid == 0 && plan->split > 1) {
            const int first_tail = plan->n_full, rem = plan->total_tiles - plan->n_full;
            int plo = 0, phi = B;
            while (phi - plo > 1) {
                const int mid = (plo + phi) >> 1;
                if (tile_ptr[mid] <= first_tail) plo = mid; else phi = mid;
            }
            const int b = order[plo];
            const int64_t nb = ptr[b + 1] - ptr[b];
            // (the tile-count bound only for a batch that is ALL tail: behind full rounds -- 64 events of 3000 nodes, 960
            // tail tiles -- the halves still win, 248 us against 275)
#ifndef DMET_F2_SPLIT_REM16
#define DMET_F2_SPLIT_REM16 5
#endif
#ifndef DMET_F2_SPLIT_REM_ALLTAIL
#define DMET_F2_SPLIT_REM_ALLTAIL 1
#endif
            if (nb >= kF2MinNodes && (nb < kF2SplitMinNodes ||
                                      ((first_tail == 0 || !DMET_F2_SPLIT_REM_ALLTAIL) && rem * 16 > simds * DMET_F2_SPLIT_REM16))) {
                plan->n_full = plan->total_tiles;
                plan->split = 1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void knn_plan_kernel(const int64_t *__restrict__ ptr, int B, KnnPlanOut o0, KnnPlanOut o1)
{
    knn_plan_body(ptr, B, blockIdx.x == 0 ? o0 : o1);
}

// Position (in the longest-first order) that owns tile t: the p with tile_ptr[p] <= t < tile_ptr[p+1].
__device__ __forceinline__ int find_tile_event(const int32_t *__restrict__ tile_ptr, int B, int t)
{
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_ptr[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// Global -> registers copy of one candidate tile (cntc rows of DP floats; missing rows become +inf).
template <int DP, bool EXACT_D, int NLD>
__device__ __forceinline__ void load_tile(float4 (&pf)[NLD], const float *__restrict__ x, int D, int c0, int cntc,
                                          int lane)
{
    const float inf = __builtin_inff();
    if (EXACT_D) {
        const float4 *g = reinterpret_cast<const float4 *>(x + (int64_t)c0 * DP);
        const int n4 = cntc * (DP / 4);
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int idx = lane + m * kWave;
            pf[m] = (idx < n4) ? g[idx] : make_float4(inf, inf, inf, inf);
        }
    } else {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int flat = (lane + m * kWave) * 4 + e;
                const int c = flat / DP, dd = flat - c * DP;
                v[e] = (c < cntc) ? ((dd < D) ? x[(int64_t)(c0 + c) * D + dd] : 0.0f) : inf;
            }
            pf[m] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

__device__ __forceinline__ float chain_dist32(const float *__restrict__ xj, const float (&q)[32])
{
    const float4 *g = reinterpret_cast<const float4 *>(xj);
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 v = g[c];
        float df;
        df = v.x - q[4 * c + 0]; acc = __builtin_fmaf(df, df, acc);
        df = v.y - q[4 * c + 1]; acc = __builtin_fmaf(df, df, acc);
        df = v.z - q[4 * c + 2]; acc = __builtin_fmaf(df, df, acc);
        df = v.w - q[4 * c + 3]; acc = __builtin_fmaf(df, df, acc);
    }
    return acc;
}

// Uncertified queries of sparsely flagged tiles (1..kRequeryMax per 128-query tile; denser tiles go to the exact
// tile kernel): ONE WORKGROUP PER FLAGGED QUERY, taken from the list the flagging sites append to (round 2: the
// tile's workgroup used to take its queries one after the other, and a single straggler cost the build ~90 us).  The
// 256 lanes stride over the event's candidates with the exact R1 chain (four rows in flight per lane; distances cached
// in LDS when the event fits), then k rounds of "smallest (d, j) above the previous pick" (R2) on 64-bit
// (distance bits, j) words with a wavefront + cross-wavefront reduction.
constexpr int kRequeryMax = 8;
constexpr int kRequeryGroups = 512;    // workgroups of the launch: they exit at once while nothing is flagged

__device__ __forceinline__ unsigned long long requery_word(float d, int j)
{
    // a candidate at d >= 1e10 (or NaN) is never a neighbour (upstream's initial best distance, dmet_oracle.c:62)
    return d < kKnnSentinel ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j) : ~0ull;
}

// Runs as the first kRequeryGroups workgroups of the exact kernel's launch (its own launch cost 4.7 us per build while
// nothing is flagged); `cache` / `red` alias that kernel's LDS.
__device__ __forceinline__ void knn_requery_body(const KnnArgs &a, float *__restrict__ cache, const int cache_floats,
                                                 unsigned long long *__restrict__ red, const int group, const int ngroups)
{
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int64_t nf64 = *a.any;
    const int nflag = (int)(nf64 < a.N ? nf64 : a.N);
    const int k = a.k;
    for (int it = group; it < nflag; it += ngroups) {   // block-uniform
        const int q = a.qlist[it];
        int lo = 0, hi = a.B;                // event of q: the last b with ptr[b] <= q (empty events share a ptr value
        while (hi - lo > 1) {                // with their successor and are skipped by taking the last)
            const int mid = (lo + hi) >> 1;
            if (a.ptr[mid] <= q) lo = mid; else hi = mid;
        }
        const int ev = lo;
        const int ev_lo = (int)a.ptr[ev], ev_hi = (int)a.ptr[ev + 1];
        const int n = ev_hi - ev_lo;
        const int xt = a.tile_ptr[a.pos_of[ev]] + (q - ev_lo) / a.requery_tile_queries;
        if (a.flags[xt] > kRequeryMax) continue;   // a densely flagged tile: the exact tile kernel recomputes it
        const bool cached = n <= cache_floats;
        float qrow[32];
        {
            const float4 *g = reinterpret_cast<const float4 *>(a.x + (int64_t)q * 32);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = g[c];
                qrow[4 * c] = v.x; qrow[4 * c + 1] = v.y; qrow[4 * c + 2] = v.z; qrow[4 * c + 3] = v.w;
            }
        }
        __syncthreads();   // the previous query's cache / reduction slots are free
        if (cached) {
            // four candidate rows in flight per lane (clamped re-reads past the end, results unused)
            for (int j0 = tid; j0 < n; j0 += 4 * 256) {
                float4 r[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 *g = reinterpret_cast<const float4 *>(a.x + (int64_t)(ev_lo + min(j0 + 256 * u, n - 1)) * 32);
#pragma unroll
                    for (int c = 0; c < 8; ++c) r[u][c] = g[c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float dc = 0.0f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float df;
                        df = r[u][c].x - qrow[4 * c + 0]; dc = __builtin_fmaf(df, df, dc);
                        df = r[u][c].y - qrow[4 * c + 1]; dc = __builtin_fmaf(df, df, dc);
                        df = r[u][c].z - qrow[4 * c + 2]; dc = __builtin_fmaf(df, df, dc);
                        df = r[u][c].w - qrow[4 * c + 3]; dc = __builtin_fmaf(df, df, dc);
                    }
                    if (j0 + 256 * u < n) cache[j0 + 256 * u] = dc;
                }
            }
            __syncthreads();
        }
        unsigned long long last = 0ull;
        bool first = true;
        for (int r = 0; r < k; ++r) {
            unsigned long long best = ~0ull;
            for (int j = tid; j < n; j += 256) {
                const float d = cached ? cache[j] : chain_dist32(a.x + (int64_t)(ev_lo + j) * 32, qrow);
                const unsigned long long w = requery_word(d, ev_lo + j);
                if ((first || w > last) && w < best) best = w;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(best, off, 64);
                if (o < best) best = o;
            }
            if (lane == 0) red[wv] = best;
            __syncthreads();
            best = red[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) if (red[w] < best) best = red[w];
            __syncthreads();
            const bool found = best != ~0ull;   // block-uniform
            if (tid == 0) {
                const int bj = found ? (int)(unsigned)best : -1;
                a.nbr[(int64_t)q * k + r] = bj;
                if (a.nbr16) a.nbr16[(int64_t)q * k + r] = local_id16(bj, ev_lo);
                a.dist[(int64_t)q * k + r] = found ? __uint_as_float((unsigned)(best >> 32)) : kKnnSentinel;
            }
            if (!found) {
                if (tid == 0)
                    for (int rr = r + 1; rr < k; ++rr) {
                        a.nbr[(int64_t)q * k + rr] = -1;
                        a.dist[(int64_t)q * k + rr] = kKnnSentinel;
                        if (a.nbr16) a.nbr16[(int64_t)q * k + rr] = 0xFFFFu;
                    }
                break;
            }
            last = best;
            first = false;
        }
    }
}

template <int DP, int KP, int TQ, bool EXACT_D>
__global__ __launch_bounds__(kWave * kWavesPerGroup, 3) void knn_kernel(const KnnArgs a)
{
    // A workgroup is kWavesPerGroup INDEPENDENT wavefronts (one work item each, no workgroup barrier): the hardware
    // spreads a workgroup's waves over the CU's 4 SIMDs, so each SIMD receives one item of every resident workgroup
    // and whole-sweep and sub-sweep items mix evenly per SIMD (single-wave workgroups left some SIMDs with 3 whole
    // sweeps: measured 3.1 ms stragglers against a 2.4 ms median).
    __shared__ KnnShared<DP, TQ> sh_all[kWavesPerGroup];
    // (wave-uniform, and said so: without readfirstlane the tile, the event and every derived address are per-lane math)
    const int wv_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    KnnShared<DP, TQ> &sh = sh_all[wv_];
    int first_group = 0;
    if constexpr (DP == 32) {
        if (a.qlist) {    // (kernel-uniform) matrix-core path: the leading workgroups are the per-query fallback
            if (a.any && *a.any == 0) return;
            if ((int)blockIdx.x < a.requery_groups) {
                constexpr int kFloats = (int)((sizeof(sh_all) - 64) / sizeof(float));
                knn_requery_body(a, reinterpret_cast<float *>(sh_all), kFloats,
                                 reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(sh_all) + sizeof(sh_all) - 64),
                                 (int)blockIdx.x, a.requery_groups);
                return;
            }
            first_group = a.requery_groups;
        }
    }
    const int item = ((int)blockIdx.x - first_group) * kWavesPerGroup + wv_;
    const int lane = threadIdx.x & 63;
#ifdef DMET_KNN_STAMP
    if (lane == 0 && item < (1 << 16)) {
        g_knn_stamps[item][0] = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_knn_stamps[item][2] = hwid;
        g_knn_stamps[item][3] = xcc;
        g_knn_stamps[item][1] = 0;
    }
#endif
    const float *__restrict__ x = a.x;
    const int64_t *__restrict__ ptr = a.ptr;
    const int D = a.D;
    constexpr int QT = kWave * TQ;  // queries per tile

    // which (query tile, candidate sub-sweep) is this wavefront?  (plan lives in device memory)
    const int n_full = a.plan->n_full, split = a.plan->split, total = a.plan->total_tiles;
    int tile = item, sub = 0, nsub = 1;
    // as the fallback of the matrix-core path (flags given) every tile is swept whole: the few flagged tiles need no
    // balancing, and without partial lists the merge launch is not needed either
    if (item >= n_full && !a.flags) {
        const int r = item - n_full;
        tile = n_full + r / split;
        sub = r % split;
        nsub = split;
    }
    if (tile >= total) return;
    if (a.any && *a.any == 0) return;   // matrix-core path certified every query: nothing to recompute
    if (a.flags && a.flags[tile] < a.flag_min) return;
    const int pos = find_tile_event(a.tile_ptr, a.B, tile);
    const int ev = a.order[pos];
    const int ev_lo = (int)ptr[ev], ev_hi = (int)ptr[ev + 1];
    const int q_first = ev_lo + (tile - a.tile_ptr[pos]) * QT;

    // candidate range = the tile's own event (or one chunk of it for a split tile)
    int clo = ev_lo, chi = ev_hi;
    if (nsub > 1) {
        const int chunk = (((chi - clo) + nsub - 1) / nsub + 1) & ~1;  // even: candidate pairs never straddle chunks
        clo = min(chi, clo + sub * chunk);
        chi = min(chi, clo + chunk);
    }

    int qi[TQ];
    bool valid[TQ];
    float tau[TQ];
    int cnt[TQ];
    float *ld[TQ];
    int32_t *lj[TQ];
    f2 q2[(TQ == 2) ? DP : 1];
    float q1[(TQ == 1) ? DP : 1];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        qi[t] = q_first + t * kWave + lane;
        valid[t] = qi[t] < ev_hi;
        const int64_t qq = valid[t] ? qi[t] : ev_lo;
#pragma unroll
        for (int c = 0; c < DP; ++c) {
            float v;
            if (EXACT_D) v = x[qq * DP + c]; else v = (c < D) ? x[qq * D + c] : 0.0f;
            if (TQ == 2) { if (t == 0) q2[c].x = v; else q2[c].y = v; }
            else q1[c] = v;
        }
        tau[t] = valid[t] ? kKnnSentinel : -1.0f;  // a distance is never < -1: idle lanes admit nothing
        cnt[t] = 0;
        if (nsub == 1) {
            ld[t] = a.wsd + qq * KP;
            lj[t] = a.wsj + qq * KP;
        } else {
            const int64_t slot = (int64_t)(tile - n_full) * QT + t * kWave + lane;
            ld[t] = a.psd + (slot * nsub + sub) * KP;
            lj[t] = a.psj + (slot * nsub + sub) * KP;
        }
    }
    unsigned fresh = (1u << TQ) - 1u;  // wave-uniform: list of slot t not yet written to ws

    constexpr int kLd4 = kTileC * DP / 4;  // float4s per tile
    constexpr int kLdPerLane = (kLd4 + kWave - 1) / kWave;
    const float inf = __builtin_inff();
    const float4 inf4 = make_float4(inf, inf, inf, inf);
    // row kTileC is a permanent +inf row: the paired sweep may read one row past a full tile, and rows past a
    // partial tile are filled with +inf too, so such candidates get d = +inf and are never admitted
    for (int i = lane; i < DP / 4; i += kWave) sh.tile[kLd4 + i] = inf4;

    float4 pf[kLdPerLane];  // next tile, prefetched into registers while the current one is swept
#pragma unroll
    for (int m = 0; m < kLdPerLane; ++m) pf[m] = inf4;
    if (clo < chi) load_tile<DP, EXACT_D, kLdPerLane>(pf, x, D, clo, min(kTileC, chi - clo), lane);

    for (int c0 = clo; c0 < chi; c0 += kTileC) {
        const int cntc = min(kTileC, chi - c0);
        wave_sync();  // every lane is done reading the previous tile
#pragma unroll
        for (int m = 0; m < kLdPerLane; ++m) {
            const int idx = lane + m * kWave;
            if (idx < kLd4) sh.tile[idx] = pf[m];
        }
        wave_sync();
        if (c0 + kTileC < chi)
            load_tile<DP, EXACT_D, kLdPerLane>(pf, x, D, c0 + kTileC, min(kTileC, chi - c0 - kTileC), lane);

        for (int cc = 0; cc < cntc; cc += 2) {
            float dA[TQ], dB[TQ];
            if (TQ == 2) {
                f2 accA = {0.0f, 0.0f}, accB = {0.0f, 0.0f};
#pragma unroll
                for (int c4 = 0; c4 < DP / 4; ++c4) {
                    const float4 v = sh.tile[cc * (DP / 4) + c4];        // wave-uniform address: LDS broadcast
                    const float4 w = sh.tile[(cc + 1) * (DP / 4) + c4];
                    const f2 vxy = {v.x, v.y}, vzw = {v.z, v.w}, wxy = {w.x, w.y}, wzw = {w.z, w.w};
                    pk_dist_step4(accA, accB, vxy, vzw, wxy, wzw, q2[4 * c4 + 0], q2[4 * c4 + 1], q2[4 * c4 + 2],
                                  q2[4 * c4 + 3]);
                }
                dA[0] = accA.x; dB[0] = accB.x;
                dA[TQ - 1] = accA.y; dB[TQ - 1] = accB.y;
            } else {
                float accA = 0.0f, accB = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < DP / 4; ++c4) {
                    const float4 v = sh.tile[cc * (DP / 4) + c4];
                    const float4 w = sh.tile[(cc + 1) * (DP / 4) + c4];
                    float df;
                    df = v.x - q1[4 * c4 + 0]; accA = __builtin_fmaf(df, df, accA);
                    df = w.x - q1[4 * c4 + 0]; accB = __builtin_fmaf(df, df, accB);
                    df = v.y - q1[4 * c4 + 1]; accA = __builtin_fmaf(df, df, accA);
                    df = w.y - q1[4 * c4 + 1]; accB = __builtin_fmaf(df, df, accB);
                    df = v.z - q1[4 * c4 + 2]; accA = __builtin_fmaf(df, df, accA);
                    df = w.z - q1[4 * c4 + 2]; accB = __builtin_fmaf(df, df, accB);
                    df = v.w - q1[4 * c4 + 3]; accA = __builtin_fmaf(df, df, accA);
                    df = w.w - q1[4 * c4 + 3]; accB = __builtin_fmaf(df, df, accB);
                }
                dA[0] = accA; dB[0] = accB;
            }
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float dj = u ? dB[t] : dA[t];
                    bool pass = dj < tau[t];
#ifdef DMET_KNN_NOSELECT
                    pass = pass && (dj < -1.0f);   // experiment: distance sweep only
#endif
                    if (pass) {
                        sh.queue[t][cnt[t]][lane] = make_uint2(__float_as_uint(dj), (unsigned)(c0 + cc + u));
                        cnt[t]++;
                    }
                }
                if (__any(cnt[t] > kQMax - 2)) {
                    tau[t] = drain_queue<KP>(sh.queue[t], lane, cnt[t], (fresh >> t) & 1u, valid[t], ld[t], lj[t]);
                    if (!valid[t]) tau[t] = -1.0f;
                    cnt[t] = 0;
                    fresh &= ~(1u << t);
                }
            }
        }
    }
#ifdef DMET_KNN_STAMP
    if (lane == 0 && item < (1 << 16)) g_knn_stamps[item][1] = __builtin_amdgcn_s_memrealtime();
#endif
    // final drain; whole-sweep tiles also emit the first k entries
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        (void)drain_queue<KP>(sh.queue[t], lane, cnt[t], (fresh >> t) & 1u, valid[t], ld[t], lj[t]);
        if (valid[t] && nsub == 1) {
            const int64_t qq = qi[t];
            if (a.k == KP) {
#pragma unroll
                for (int p = 0; p < KP; p += 4) {
                    *reinterpret_cast<float4 *>(a.dist + qq * KP + p) = *reinterpret_cast<const float4 *>(ld[t] + p);
                    *reinterpret_cast<int4 *>(a.nbr + qq * KP + p) = *reinterpret_cast<const int4 *>(lj[t] + p);
                }
            } else {
                for (int p = 0; p < a.k; ++p) {
                    a.dist[qq * a.k + p] = ld[t][p];
                    a.nbr[qq * a.k + p] = lj[t][p];
                }
            }
            if (a.nbr16)
                for (int p = 0; p < a.k; ++p) a.nbr16[qq * a.k + p] = local_id16(lj[t][p], ev_lo);
        }
    }
}

// Merge the `split` sorted partial lists of every query of the split tiles: k steps of a `split`-way merge by
// (d, j); sentinels (1e10, -1) sort last.  One lane per query slot of the tail tiles (worst-case grid).
template <int KP>
__global__ __launch_bounds__(256) void knn_merge_kernel(const KnnArgs a, int tile_queries)
{
    if (a.any && *a.any == 0) return;
    const int n_full = a.plan->n_full, split = a.plan->split, total = a.plan->total_tiles;
    if (split <= 1) return;
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int tile = n_full + (int)(slot / tile_queries);
    if (tile >= total) return;
    if (a.flags && a.flags[tile] < a.flag_min) return;
    const int pos = find_tile_event(a.tile_ptr, a.B, tile);
    const int ev = a.order[pos];
    const int64_t qi = a.ptr[ev] + (int64_t)(tile - a.tile_ptr[pos]) * tile_queries + (slot % tile_queries);
    if (qi >= a.ptr[ev + 1]) return;
    const float *pd = a.psd + slot * split * KP;
    const int32_t *pj = a.psj + slot * split * KP;
    int head[kMaxSplit];
#pragma unroll
    for (int s = 0; s < kMaxSplit; ++s) head[s] = 0;
    for (int p = 0; p < a.k; ++p) {
        float bd = kKnnSentinel;
        int32_t bj = -1;
        int bs = -1;
#pragma unroll
        for (int s = 0; s < kMaxSplit; ++s) {
            if (s < split && head[s] < KP) {
                const float d = pd[s * KP + head[s]];
                const int32_t j = pj[s * KP + head[s]];
                if (j >= 0 && (bs < 0 || d < bd || (d == bd && j < bj))) { bd = d; bj = j; bs = s; }
            }
        }
#pragma unroll
        for (int s = 0; s < kMaxSplit; ++s) head[s] += (s == bs) ? 1 : 0;
        a.dist[qi * a.k + p] = bd;
        a.nbr[qi * a.k + p] = bj;
        if (a.nbr16) a.nbr16[qi * a.k + p] = local_id16(bj, (int)a.ptr[ev]);
    }
}

// ---- matrix-core filter + exact re-rank (D = 32) ---------------------------------------------------------------
// The difference form of R1 cannot run on the matrix cores, but it does not have to run for every pair.  With
//   key(i,j) = |x_j|^2 - 2 x_i.x_j   ( = d(i,j) - |x_i|^2 in exact arithmetic )
// a matrix-core sweep ranks the candidates of every query approximately.  Each fp32 feature is split into two bf16
// terms x = h + m + r (h = bf16(x), m = bf16(x - h), |r| <= 2^-18 |x|) and x_i.x_j is taken as h.h' + h.m' + m.h' with
// v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation): 6 MFMAs of 8 passes per 32x32 block instead of
// 16 fp32 MFMAs of 16 passes (measured on MI355X, tools/mfma_valu_micro.hip: MFMA passes and VALU instructions of a
// SIMD do NOT overlap, not even across wavefronts, so matrix-pipe time adds to the selection's VALU time).
// Error budget for a pair (a = |x_i|, b = |x_j|): dropped split terms 6.2 * 2^-18 a b, MFMA fp32 accumulation
// (<= 100 roundings) 3.3 * 2^-18 a b + 6e-6 b^2, squared norms 1.9e-6 (a^2 + b^2), the R1 chain itself 2.1e-6 (a + b)^2:
//   |d_chain(i,j) - (key(i,j) + |x_i|^2)|  <=  E(a, b) = 4e-5 a b + 1e-5 b^2 + 4e-6 a^2.
// Certification only has to rule out dropped candidates that could beat the k-th kept distance d_k: such a candidate
// has b <= R_i = a + sqrt(d_k) (otherwise d >= (b - a)^2 > d_k already), so the slack is e_i = 2 E(a, R_i) (2x margin)
// and depends on the query alone -- an outlier with a huge norm elsewhere in the event does not loosen it.
// R1/R2 stay the definition of the RESULT: every returned (d, j) comes from the exact fmaf chain and the (d, j) order;
// the expansion above only decides which pairs the exact chain is run for, under the proven bound.
//   1. knn_filter_kernel, sweep: every query (one lane) keeps the M = k + 4 smallest keys of its candidate range and
//      the candidates themselves (ties at the threshold included) in LDS.
//   2. exact re-rank: the R1 chain for the kept candidates, top-k by (d, j) (R2) -- in the tail of the filter kernel
//      for whole-sweep items, in knn_rerank_kernel for the tail tiles whose candidate range was split over two
//      work items.  It is THE exact answer iff nothing that could belong to the top k was dropped: a dropped
//      candidate has key >= the list's threshold tau, hence d >= tau + |x_i|^2 - e_i; if that exceeds the k-th
//      smallest exact distance among the kept candidates (for every partial list that saw at least M keys), the kept
//      top-k is the global top-k.
//   3. Queries that fail the test (exact ties beyond the list length, duplicates, lattices, events of more than
//      65535 nodes) are recomputed exactly: one workgroup per query when a 128-query tile has few of them, the exact
//      tile kernel above otherwise.
// Result: bit-identical output at a fraction of the VALU work.
constexpr int kFQ = 64;             // queries per filter work item: two 32-column MFMA blocks
// (4 sub-sweeps per tail tile were built and measured for the second form: every sub-sweep re-ranks its own ~27
// candidates and the 4-way merge took 60 us instead of 17: filter 452 -> 509 us, build 0.55 -> 0.65 ms.  Not kept.)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct KnnFilterArgs {
    const float *x;
    const int64_t *ptr;
    int B;
    int64_t N;
    int k;
    float *nrm;                 // [N] squared norms
    uint8_t *rec;               // candidate tile records (kRecBytes each): the bf16 split of 32 rows in MFMA operand
                                // order + their squared norms; event b starts at record (ptr[b] >> 5) + b
    const KnnPlan *plan;        // filter plan (kFQ-query tiles)
    const int32_t *order;
    const int32_t *pos_of;
    const int32_t *tile_ptr;
    float *psd;                 // split tiles: [(tile-n_full)*kFQ + slot][split][MS], MS = M kept (key, id) + threshold slot
    int32_t *psj;
    int32_t *nbr;
    float *dist;
    uint16_t *nbr16;            // optional event-local copy of nbr (see KnnArgs)
    int32_t *flags;             // [exact tiles] number of uncertified queries of the tile
    int32_t *any;               // total number of uncertified queries: the fallback kernels exit at once while it is 0
    uint8_t *qflag;             // [N] 1 = uncertified query
    int32_t *qlist;             // [N] the uncertified queries in the order they were flagged (any = their number)
    const int32_t *xtile_ptr;   // tile prefix of the exact kernel's plan (same event order)
    int xtile_queries;
    int form2;                  // 1: events of kF2MinNodes..kF2MaxNodes nodes are swept by the second form
    float slack_scale;          // certificate slack relative to the 32-feature bound (1.5 at 64 features)
    // rider (dmet_knn_local_dense_f32): workgroups first_rider .. gridDim.x - 1 of the filter launch compute the
    // node-level dense layer of the EdgeConv that consumes this graph (nls_body.h; 32 -> 32 features), rP == nullptr: none
    const float *rW, *rb;
    float *rP, *rQ;
    int r_sliced;               // layout: 0 row-major fp32, 1 slice-major fp32, 2 row-major with Q as bf16 bits
    int first_rider;
    int emit_coalesced;         // 1: nbr / dist / nbr16 are 16-byte aligned: whole-sweep items write their rows as 16-byte pieces
    int no_rerank;              // 1: knn_rerank_kernel is not launched (dmet_knn_size_hint: no first-form event); a first-form
                                // tail item that shows up anyway hands its queries to the exact kernel
};

// What the caller knows about the event sizes of the NEXT build on this thread (dmet_knn_size_hint; 0 = unknown).
struct KnnSizeHint {
    int min_nodes = 0, max_nodes = 0;
};
thread_local KnnSizeHint g_size_hint;

// The dense layer a build may carry (set by dmet_knn_local_dense_f32 around its call of the build on this thread).
struct KnnRider {
    const float *W = nullptr, *b = nullptr;
    float *P = nullptr, *Q = nullptr;
    int sliced = 0;
    bool done = false;
};
thread_local KnnRider g_rider;

// BatchNorm transform + residual fused into the prep launch (dmet_bn_knn_local_dense_f32): the build's input y is not
// there yet -- the prep kernel reads the rows it is made of, raw (the BatchNorm's input) and res, writes
//   y = (raw - mean) * (gamma * invstd) + beta + res      (the expression of bn_apply_kernel, same bits)
// and cuts its tile records from the values it just formed: one pass over the rows instead of two, one launch less.
struct KnnAffine {
    const float *raw = nullptr, *res = nullptr, *gamma = nullptr, *beta = nullptr, *mean = nullptr, *invstd = nullptr;
    bool done = false;
};
thread_local KnnAffine g_affine;
// rider workgroups per launch (DMET_KNN_RIDER_GROUPS: experiments; 128..1024 measured within 1 % of each other at
// 64 x 4500 nodes: the last round leaves ~1150 of the 2048 wavefront slots empty)
inline int rider_groups()
{
    static int cached = 0;
    if (cached == 0) {
        const char *e = getenv("DMET_KNN_RIDER_GROUPS");
        const int v = e ? atoi(e) : 0;
        cached = (v >= 1 && v <= 4096) ? v : 512;
    }
    return cached;
}

// A query whose result is not certified: counted per exact-kernel tile (dense tiles go to the exact tile kernel) and
// appended to the list the per-query fallback walks.  Every query is flagged at most once per call.
__device__ __forceinline__ void flag_query(const KnnFilterArgs &a, int q, int xtile)
{
    a.qflag[q] = 1;
    atomicAdd(a.flags + xtile, 1);   // a count: order-independent
    const int slot = atomicAdd(a.any, 1);
    if (slot < a.N) a.qlist[slot] = q;
}

__device__ __forceinline__ unsigned bf16_rne_bits(float f)   // finite inputs
{
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// ---- candidate tile records ---------------------------------------------------------------------------------------
// The filter sweeps an event's candidates in tiles of 32 rows.  A tile's A operands are stored the way a wavefront
// consumes them: frag[m][lane] = the 8 bf16 values lane (col = lane & 31, hh = lane >> 5) feeds to MFMA operand m
// (m = 0,1: high terms of features 0-15 / 16-31; m = 2,3: middle terms), i.e. features 16 (m & 1) + 8 hh + 0..7 of row
// col -- so one load instruction of a wavefront is 1 KB of contiguous memory (8 full cache lines) instead of 32-byte
// pieces of 32 different rows -- followed by the 32 squared norms.  Rows past the end of an event are zero with
// norm = +inf (key = +inf: never admitted), so the sweep needs no range checks.  Event b owns the records
// [(ptr[b] >> 5) + b, ... + ceil(n_b / 32)): monotone and disjoint without a prefix sum over the events.
constexpr int kRecFragBytes = 4 * kWave * 16;           // 4096: the four operand fragments of 32 features
// NH = number of 32-feature halves per row (1: D = 32, 2: D = 64, the hidden width of the reference's DRN,
// model/dynamic_reduction_network.py:40,86): a record holds NH x 4 fragments, then the 32 squared norms
constexpr int rec_bytes(int NH) { return kRecFragBytes * NH + 32 * 4; }
constexpr int kRecBytes = rec_bytes(1);                 // 4224 (33 cache lines)
constexpr int kRecBytesMax = rec_bytes(2);              // the workspace is carved for either width

__device__ __forceinline__ int64_t rec_base_tile(const int64_t *__restrict__ ptr, int b) { return (ptr[b] >> 5) + b; }

// Second-form events (kF2MinNodes..kF2MaxNodes nodes, when the second form is enabled) get SINGLE-TERM fp16 records
// instead of the bf16 split: frag[m][lane], m = 0..2 NH - 1, = the 8 fp16 values (round to nearest even) of features
// 16 m + 8 hh + 0..7 of row col, then the 32 squared norms at byte kRec16FragBytes * NH (the record stride stays
// rec_bytes(NH)).  Two MFMAs per 32 x 32 block instead of six; the certificate of the second form carries the fp16
// rounding (see f2_slack).  A row with a feature outside the range whose doubled value fits fp16 (|v| >= 16384, or not
// finite) is stored as zeros with norm -inf: its key is -inf for every query, so it is always a candidate of the exact
// re-rank and never dropped on the strength of an overflowed product (as a QUERY such a row is refused by the
// certificate through its true norm, kept in nrm[]).
constexpr int kRec16FragBytes = 2 * kWave * 16;         // 2048: the two fp16 operand fragments of 32 features
constexpr float kF16WideLimit = 16384.0f;

__device__ __forceinline__ bool f2_in_domain64(int64_t n) { return n >= kF2MinNodes && n <= kF2MaxNodes; }

// One wavefront per record: lane (col, hh) converts the 16 features of row col it will later feed to the MFMAs.
// Also writes the flat norm array (certificates) and clears the uncertified-query counters / flags (zero_bytes bytes
// at `zero`, 4-byte aligned) for the launches that follow, which saves a memset launch per call.
template <int NH, bool AFFINE = false>
__global__ __launch_bounds__(256) void knn_prep_kernel(const float *__restrict__ x, const int64_t *__restrict__ ptr,
                                                        int B, int64_t N, float *__restrict__ nrm,
                                                        uint8_t *__restrict__ rec, int64_t nrec,
                                                        uint32_t *__restrict__ zero, size_t zero_bytes, KnnPlanOut o0,
                                                        KnnPlanOut o1, int form2, KnnAffine af = KnnAffine{})
{
    static_assert(!AFFINE || NH == 1, "the fused BatchNorm transform is built for 32 features");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {
        const size_t words = zero_bytes >> 2, total = (size_t)gridDim.x * blockDim.x;
        for (size_t wd = (size_t)t; wd < words; wd += total) zero[wd] = 0u;
        if (t == 0)
            for (size_t bt = words << 2; bt < zero_bytes; ++bt) reinterpret_cast<uint8_t *>(zero)[bt] = 0;
    }
    // the last two workgroups compute the two launch plans (one launch less on the critical path of the build)
    if (blockIdx.x + 2 >= gridDim.x) {
        knn_plan_body(ptr, B, blockIdx.x + 2 == gridDim.x ? o0 : o1);
        return;
    }
    // wave-uniform, and said so: the search below then runs on the scalar unit (s_load through the constant cache)
    // instead of six dependent vector loads per wavefront
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile >= nrec) return;
    const int lane = (int)(t & 63), col = lane & 31, hh = lane >> 5;
    // event that owns the record: the last b with (ptr[b] >> 5) + b <= tile.  A 64-ary search across the lanes (one
    // vector load + ballot per level: one memory round trip for B <= 64, two up to 4096) instead of a binary search of
    // log2(B) DEPENDENT scalar loads -- with every wavefront of the grid resident at once the kernel lasts as long as one
    // wavefront's chain of round trips (measured: 16.5 -> 16.0 us at 64 events; the chain was not what bounds the kernel)
    int lo = 0, hi = B;                      // invariant: base(lo) <= tile < base(hi) (base(B) = +inf)
    {
        const int ln = (int)(threadIdx.x & 63);
        while (hi - lo > 1) {
            const int span = hi - lo, step = (span + 63) >> 6;           // candidates lo + step * (ln + 1), ln = 0..63
            const int cand = lo + step * (ln + 1);
            const bool ok = cand < hi && rec_base_tile(ptr, cand) <= tile;
            const unsigned long long m = __ballot(ok);                   // monotone: a prefix of the lanes
            const int cnt = __popcll(m);                                 // wave-uniform
            const int nlo = lo + step * cnt;
            const int nhi = min(hi, lo + step * (cnt + 1));
            lo = __builtin_amdgcn_readfirstlane(nlo);
            hi = __builtin_amdgcn_readfirstlane(nhi);
        }
    }
    const int64_t ev_lo = ptr[lo], n = ptr[lo + 1] - ev_lo;
    const int64_t li0 = (tile - rec_base_tile(ptr, lo)) * 32;
    if (li0 >= n) return;                   // a slot between two events: never read
    const bool live = li0 + col < n;
    const int64_t r = ev_lo + (live ? li0 + col : 0);
    const bool rec16 = form2 != 0 && f2_in_domain64(n);    // wave-uniform: one event per record
    float s = 0.0f;
    uint8_t *recp = rec + tile * rec_bytes(NH);
    float f[NH][16];
    if constexpr (NH == 1) {
        // Coalesced tile loads (a wavefront instruction = 1 KB of consecutive bytes: lane l takes float4 64 j + l of the
        // 4 KB tile, i.e. feature group l & 7 of row 8 j + (l >> 3)), the optional BatchNorm transform applied right there
        // (one feature group per lane: its constants are loaded once) and y stored the same way; the values then change
        // to the record layout (lane (col, hh): features 16 kb + 8 hh .. + 7 of row col) through the wavefront's LDS tile.
        // Until the third session the rows were read as 32-byte pieces in the record layout: 4 instructions that each
        // touch all 32 cache lines of the tile.
        __shared__ __attribute__((aligned(16))) float prep_tile[4][32 * 36];
        float *T = prep_tile[threadIdx.x >> 6];
        const int fg = lane & 7;
        float4 mu4, sc4, be4;
        if constexpr (AFFINE) {
            mu4 = reinterpret_cast<const float4 *>(af.mean)[fg];
            const float4 is4 = reinterpret_cast<const float4 *>(af.invstd)[fg], ga4 = reinterpret_cast<const float4 *>(af.gamma)[fg];
            be4 = reinterpret_cast<const float4 *>(af.beta)[fg];
            sc4 = make_float4(ga4.x * is4.x, ga4.y * is4.y, ga4.z * is4.z, ga4.w * is4.w);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int rowj = 8 * jj + (lane >> 3);
            const bool livej = li0 + rowj < n;
            const int64_t rj = ev_lo + (livej ? li0 + rowj : 0);
            float4 v;
            if constexpr (AFFINE) {
                // x is the OUTPUT here: y = (raw - mean) * (gamma * invstd) + beta (+ res), written for the live rows
                const float4 a = reinterpret_cast<const float4 *>(af.raw + rj * 32)[fg];
                v.x = (a.x - mu4.x) * sc4.x + be4.x; v.y = (a.y - mu4.y) * sc4.y + be4.y;
                v.z = (a.z - mu4.z) * sc4.z + be4.z; v.w = (a.w - mu4.w) * sc4.w + be4.w;
                if (af.res) {
                    const float4 q = reinterpret_cast<const float4 *>(af.res + rj * 32)[fg];
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                if (livej) reinterpret_cast<float4 *>(const_cast<float *>(x) + rj * 32)[fg] = v;
            } else {
                v = reinterpret_cast<const float4 *>(x + rj * 32)[fg];
            }
            *reinterpret_cast<float4 *>(&T[rowj * 36 + 4 * fg]) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wavefront's own LDS writes, in order
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 v0 = *reinterpret_cast<const float4 *>(&T[col * 36 + 16 * kb + 8 * hh]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&T[col * 36 + 16 * kb + 8 * hh + 4]);
            f[0][8 * kb + 0] = v0.x; f[0][8 * kb + 1] = v0.y; f[0][8 * kb + 2] = v0.z; f[0][8 * kb + 3] = v0.w;
            f[0][8 * kb + 4] = v1.x; f[0][8 * kb + 5] = v1.y; f[0][8 * kb + 6] = v1.z; f[0][8 * kb + 7] = v1.w;
        }
    } else {
#pragma unroll
        for (int half = 0; half < NH; ++half) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float4 *g = reinterpret_cast<const float4 *>(x + r * (32 * NH) + 32 * half + 16 * kb + 8 * hh);
                const float4 v0 = g[0], v1 = g[1];
                f[half][8 * kb + 0] = v0.x; f[half][8 * kb + 1] = v0.y; f[half][8 * kb + 2] = v0.z; f[half][8 * kb + 3] = v0.w;
                f[half][8 * kb + 4] = v1.x; f[half][8 * kb + 5] = v1.y; f[half][8 * kb + 6] = v1.z; f[half][8 * kb + 7] = v1.w;
            }
        }
    }
    bool wide = false;
#pragma unroll
    for (int half = 0; half < NH; ++half) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float v = live ? f[half][u] : 0.0f;
            f[half][u] = v;
            s = __builtin_fmaf(v, v, s);
            wide = wide || !(__builtin_fabsf(v) < kF16WideLimit);
        }
    }
    s += __shfl_xor(s, 32, 64);             // the row's other features: fixed order, deterministic
    if (rec16) {
        wide = wide || (__shfl_xor(wide ? 1 : 0, 32, 64) != 0);
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            uint4 *dst = reinterpret_cast<uint4 *>(recp + half * kRec16FragBytes);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f16x8 hv;
#pragma unroll
                for (int u = 0; u < 8; ++u) hv[u] = wide ? (_Float16)0.0f : (_Float16)f[half][8 * kb + u];   // v_cvt_f16_f32: RNE
                dst[kb * 64 + lane] = __builtin_bit_cast(uint4, hv);
            }
        }
        if (hh == 0) {
            reinterpret_cast<float *>(recp + kRec16FragBytes * NH)[col] =
                !live ? __builtin_inff() : (wide ? -__builtin_inff() : s);
            if (live) nrm[r] = s;
        }
        return;
    }
#pragma unroll
    for (int half = 0; half < NH; ++half) {
        unsigned h[16], m[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float v = f[half][u];
            h[u] = bf16_rne_bits(v);
            m[u] = bf16_rne_bits(v - __uint_as_float(h[u] << 16));   // the subtraction is exact
        }
        uint4 *dst = reinterpret_cast<uint4 *>(recp + half * kRecFragBytes);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            dst[kb * 64 + lane] = make_uint4(h[8 * kb] | (h[8 * kb + 1] << 16), h[8 * kb + 2] | (h[8 * kb + 3] << 16),
                                             h[8 * kb + 4] | (h[8 * kb + 5] << 16), h[8 * kb + 6] | (h[8 * kb + 7] << 16));
            dst[(2 + kb) * 64 + lane] = make_uint4(m[8 * kb] | (m[8 * kb + 1] << 16), m[8 * kb + 2] | (m[8 * kb + 3] << 16),
                                                   m[8 * kb + 4] | (m[8 * kb + 5] << 16), m[8 * kb + 6] | (m[8 * kb + 7] << 16));
        }
    }
    if (!live) s = __builtin_inff();
    if (hh == 0) {
        reinterpret_cast<float *>(recp + kRecFragBytes * NH)[col] = s;
        if (live) nrm[r] = s;
    }
}

// One 32(candidates) x 32(queries) block: acc = cinit + sum over both 16-feature k-blocks of  h.h' + h.m' + m.h'.
// Operand map of v_mfma_f32_32x32x16_bf16: lane (r = lane & 31, hh = lane >> 5) holds A[row r][k = 8 hh + 0..7].
// av / bv = {high k-block 0, high k-block 1, middle k-block 0, middle k-block 1}.
template <int NH = 1>
__device__ __forceinline__ f32x16 filter_block(const bf16x8 (&av)[4 * NH], const bf16x8 (&bv)[4 * NH], const f32x16 &cinit)
{
    f32x16 acc = cinit;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4 * h + 0], bv[4 * h + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4 * h + 1], bv[4 * h + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4 * h + 0], bv[4 * h + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4 * h + 1], bv[4 * h + 3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4 * h + 2], bv[4 * h + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4 * h + 3], bv[4 * h + 1], acc, 0, 0, 0);
    }
    return acc;
}

// one candidate tile record: the lane's 4 NH A operands and the squared norms of the 16 candidate rows it receives
// results for (accumulator seed; rows (e & 3) + 8 (e >> 2) + 4 hh)
template <int NH = 1>
__device__ __forceinline__ void filter_load(bf16x8 (&av)[4 * NH], f32x16 &cinit, const uint8_t *__restrict__ rec,
                                            int64_t tidx, int lane, int hh)
{
    const uint8_t *base = rec + tidx * rec_bytes(NH);
    const bf16x8 *g = reinterpret_cast<const bf16x8 *>(base);
#pragma unroll
    for (int m = 0; m < 4 * NH; ++m) av[m] = g[m * 64 + lane];
    const float4 *nr = reinterpret_cast<const float4 *>(base + kRecFragBytes * NH);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = nr[2 * q + hh];
        cinit[4 * q] = v.x; cinit[4 * q + 1] = v.y; cinit[4 * q + 2] = v.z; cinit[4 * q + 3] = v.w;
    }
}

// ---- selection state of one query (= one lane) ----------------------------------------------------------------
// After v_permlane32_swap of the two accumulators lane (c, hh) holds all 32 keys of query (block hh, column c), so
// a lane owns ONE query.  Per lane:
//   * tk[M]: the M smallest keys so far, sorted, in registers; inserting is a v_med3_f32 chain (one op per slot,
//     no payload to move);  tau = tk[M-1] is the admission threshold;
//   * an LDS queue of (key, j) pairs that doubles as the store of the kept candidates: admitted keys are appended;
//     when a lane runs out of room the wave drains: new entries update tk, then every lane compacts its queue in
//     place to the entries with key <= tau (at most M survive, ties aside).  No global-memory traffic until the end.
// A lane whose queue cannot be compacted below the refill mark (more than M candidates tied at tau) gives up
// (tau = -inf) and marks its list as overflowed: the re-rank flags such queries for the exact path.
template <int M>
struct FilterLane {
    float tk[M];
    float tau;
    int cnt;       // entries in the queue
    int kept;      // entries that survived the last compaction (already in tk)
    bool overflow;
};

constexpr int filter_list_len(int KP) { return KP + 4; }   // M = kept keys per list: result capacity KP (>= k) + 4
constexpr int filter_queue_len(int M) { return M + 28; }

// LDS queue of one wavefront: keys and 16-bit event-relative candidate ids in separate arrays (6 bytes per entry:
// QF = M + 28 slots per lane fit two wavefronts per SIMD).  Events of more than 65535 nodes do not fit the id and
// are handed to the exact kernel (every lane reports overflow).
template <int QF>
struct FilterQueue {
    unsigned key[QF][kWave];
    unsigned short id[QF][kWave];
};

template <int M>
__device__ __forceinline__ void filter_drain(FilterLane<M> &L, FilterQueue<filter_queue_len(M)> &Q, int lane)
{
    constexpr int QF = filter_queue_len(M);
    // 1. new entries -> sorted keys
    int maxnew = L.cnt - L.kept;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxnew = max(maxnew, __shfl_xor(maxnew, off, 64));
    for (int s = 0; s < maxnew; ++s) {
        const int idx = L.kept + s;
        if (idx < L.cnt) {
            const float key = __uint_as_float(Q.key[idx][lane]);
            if (key < L.tk[M - 1]) {
#pragma unroll
                for (int p = M - 1; p >= 1; --p) L.tk[p] = __builtin_amdgcn_fmed3f(L.tk[p - 1], key, L.tk[p]);
                L.tk[0] = fminf(L.tk[0], key);
            }
        }
    }
    const float tau = L.tk[M - 1];
    // 2. in-place compaction of every lane's queue to key <= tau
    int maxcnt = L.cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, off, 64));
    int out = 0;
    for (int s = 0; s < maxcnt; ++s) {
        if (s < L.cnt) {
            const unsigned kb = Q.key[s][lane];
            const unsigned short id = Q.id[s][lane];
            if (__uint_as_float(kb) <= tau) { Q.key[out][lane] = kb; Q.id[out][lane] = id; ++out; }
        }
    }
    L.cnt = out;
    L.kept = out;
    if (out > QF - 8) {          // cannot make room: too many candidates tied at tau
        L.overflow = true;
        L.cnt = 0; L.kept = 0;
        L.tau = -__builtin_inff();
    } else if (!L.overflow) {
        L.tau = tau;
    }
}

// 16 keys of one accumulator.  The push is branch-free (a compare, a carry add, the slot address and the id: four VALU
// ops per key; per-key branches cost more in scalar work and pipeline bubbles than they skip): the slot is always
// written and only kept when the key is admitted.
template <int M>
__device__ __forceinline__ void filter_select(FilterLane<M> &L, const f32x16 &acc, int jrel,
                                              FilterQueue<filter_queue_len(M)> &Q, int lane)
{
    constexpr int QF = filter_queue_len(M);
#if defined(DMET_FILTER_ABL) && (DMET_FILTER_ABL == 1 || DMET_FILTER_ABL == 2)
    // cycle-budget experiment (tools/knn_budget.sh): keys computed (and swapped), never looked at
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[e]));
    return;
#endif
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (__any(L.cnt > QF - 8)) filter_drain<M>(L, Q, lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = half * 8 + u;
            const float key = acc[e];
#if defined(DMET_FILTER_ABL) && DMET_FILTER_ABL == 3
            // experiment: the push's VALU work (compare, count, two slot addresses, id) without the LDS stores
            const unsigned ak = (unsigned)L.cnt * (kWave * 4u) + (unsigned)lane * 4u;
            const unsigned ai = (unsigned)L.cnt * (kWave * 2u) + (unsigned)lane * 2u;
            const unsigned idv = (unsigned)(jrel + (e & 3) + 8 * (e >> 2));
            asm volatile("" ::"v"(ak), "v"(ai), "v"(idv), "v"(key));
#else
            Q.key[L.cnt][lane] = __float_as_uint(key);
            Q.id[L.cnt][lane] = (unsigned short)(jrel + (e & 3) + 8 * (e >> 2));
#endif
            L.cnt += (key < L.tau) ? 1 : 0;
        }
#if defined(DMET_FILTER_ABL) && (DMET_FILTER_ABL == 3 || DMET_FILTER_ABL == 4)
        asm volatile("v_mov_b32 %0, 0" : "=v"(L.cnt) : "v"(L.cnt));   // experiment: queue never fills, no drains
#endif
    }
}


// events of the second form (filter2_wave below): kF2MinNodes .. kF2MaxNodes nodes (constants with the plan)
__device__ __forceinline__ bool f2_in_domain(int n) { return n >= kF2MinNodes && n <= kF2MaxNodes; }

// One wavefront's item of the first form (work item `group * kWavesPerGroup + wv` of the plan); Q is the wavefront's
// own LDS.  No workgroup barrier inside: wavefronts of one workgroup may run different forms (knn_filter12_kernel).
template <int KP>
__device__ __forceinline__ void filter1_wave(const KnnFilterArgs &a, FilterQueue<filter_queue_len(filter_list_len(KP))> &Q,
                                             int group, int wv, int lane)
{
    constexpr int M = filter_list_len(KP);
    constexpr int QF = filter_queue_len(M);
    constexpr int MS = (M + 1 + 3) & ~3;   // list stride in the workspace: M entries, then tau (d array) / overflow (j array)
    const int col = lane & 31, hh = lane >> 5;
    const int item = group * kWavesPerGroup + wv;
    const uint8_t *__restrict__ rec = a.rec;
    const int64_t *__restrict__ ptr = a.ptr;

    if (a.form2 && a.plan->form1_events == 0) return;      // every event is swept by the second form
    const int n_full = a.plan->n_full, split = a.plan->split, total = a.plan->total_tiles;
    int tile = item, sub = 0, nsub = 1;
    if (item >= n_full) {
        const int r = item - n_full;
        tile = n_full + r / split;
        sub = r % split;
        nsub = split;
    }
    if (tile >= total) return;
    const int pos = find_tile_event(a.tile_ptr, a.B, tile);
    const int ev = a.order[pos];
    const int ev_lo = (int)ptr[ev], ev_hi = (int)ptr[ev + 1];
    if (a.form2 && f2_in_domain(ev_hi - ev_lo)) return;   // swept by the second form
    const int q_first = ev_lo + (tile - a.tile_ptr[pos]) * kFQ;
    int clo = ev_lo, chi = ev_hi;
    if (nsub > 1) {
        const int chunk = (((chi - clo) + nsub - 1) / nsub + 31) & ~31;
        clo = min(chi, clo + sub * chunk);
        chi = min(chi, clo + chunk);
    }
    const bool fits = (ev_hi - ev_lo) <= 65535;   // 16-bit candidate ids

    // record of the event's first 32 candidates; the (32-aligned, event-relative) tile at c0 is rbase + (c0 - ev_lo) / 32
    const int64_t rbase = (ptr[ev] >> 5) + ev;
    const int64_t rlast = rbase + (ev_hi - ev_lo - 1) / 32;
    // B operands of both 32-query blocks: -2 * the query's bf16 terms (exact: sign flip and exponent + 1); the query
    // tiles are records too (a block past the end of the event reads the event's last record: idle lanes)
    bf16x8 bq[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t qrec = min(rbase + (q_first - ev_lo) / 32 + b, rlast);
        const bf16x8 *g = reinterpret_cast<const bf16x8 *>(rec + qrec * kRecBytes);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bf16x8 v = g[m * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float f = -2.0f * __uint_as_float(((unsigned)(unsigned short)v[u]) << 16);
                bq[b][m][u] = (short)(__float_as_uint(f) >> 16);
            }
        }
    }
    // the query this lane selects for (after the half-wave swap): block hh, column col
    const int myq = q_first + hh * 32 + col;
    const bool valid = myq < ev_hi;
    FilterLane<M> L;
#pragma unroll
    for (int p = 0; p < M; ++p) L.tk[p] = kKnnSentinel;
    L.tau = (valid && fits) ? kKnnSentinel : -__builtin_inff();
    L.cnt = 0; L.kept = 0;
    L.overflow = !(valid && fits);   // idle lanes never admit; oversized events are left to the exact kernel

    if (clo < chi && fits) {
        bf16x8 av[4], an[4];
        f32x16 ci, cn;
        int64_t tidx = rbase + (clo - ev_lo) / 32;
        filter_load(av, ci, rec, tidx, lane, hh);
        for (int c0 = clo; c0 < chi; c0 += 32) {
            const bool more = c0 + 32 < chi;
            if (more) filter_load(an, cn, rec, ++tidx, lane, hh);
            f32x16 acc0 = filter_block(av, bq[0], ci);
            f32x16 acc1 = filter_block(av, bq[1], ci);
            // lanes 32..63 of block 0 <-> lanes 0..31 of block 1: afterwards acc0 = rows {0-3, 8-11, ..} and
            // acc1 = rows {4-7, 12-15, ..} of THIS lane's query
#if !(defined(DMET_FILTER_ABL) && DMET_FILTER_ABL == 2)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[e]), __float_as_uint(acc1[e]),
                                                                false, false);
                acc0[e] = __uint_as_float(r[0]);
                acc1[e] = __uint_as_float(r[1]);
            }
#endif
            filter_select<M>(L, acc0, c0 - ev_lo, Q, lane);
            filter_select<M>(L, acc1, c0 - ev_lo + 4, Q, lane);
            if (more) {
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = an[m];
                ci = cn;
            }
        }
    }
    filter_drain<M>(L, Q, lane);
#if defined(DMET_FILTER_ABL) && DMET_FILTER_ABL == 5
    return;   // experiment: selection only, no exact re-rank / certificate
#endif
    if (nsub == 1) {
        // ---- whole-sweep items: exact re-rank right here, while the kept candidates still sit in LDS ----------------
        // Every lane owns one query and <= QF-8 kept candidates (all keys <= tau, ties included, so every dropped
        // candidate has key >= tau).  Round c handles candidate c of all 64 queries: the rows are fetched cooperatively
        // (8 lanes x 16 bytes per row: 8 cache lines per load instruction instead of 64) into the LDS space of the
        // keys (no longer needed), each lane runs the exact R1 chain on its row and inserts (d, j) into its sorted top-k.
        const int64_t qrow_id = valid ? myq : ev_lo;
        float qrow[32];
        {
            const float4 *g = reinterpret_cast<const float4 *>(a.x + qrow_id * 32);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = g[c];
                qrow[4 * c] = v.x; qrow[4 * c + 1] = v.y; qrow[4 * c + 2] = v.z; qrow[4 * c + 3] = v.w;
            }
        }
        float kd[KP];
        int32_t kj[KP];
#pragma unroll
        for (int p = 0; p < KP; ++p) { kd[p] = kKnnSentinel; kj[p] = -1; }
        constexpr int kRowPad = 36;
        static_assert(kWave * kRowPad <= QF * kWave, "row staging must fit the key array");
        float (*rows)[kRowPad] = reinterpret_cast<float (*)[kRowPad]>(&Q.key[0][0]);
        const int mycnt = valid ? L.cnt : 0;
        int maxcnt = mycnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, off, 64));
        // the loop is latency bound (load -> LDS -> chain per round): rows are fetched two rounds ahead into registers
        auto fetch = [&](int c, float4 (&pv)[8], int32_t &jout) {
            jout = (c < mycnt) ? ev_lo + (int32_t)Q.id[c < QF ? c : 0][lane] : -1;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int32_t jr = __shfl(jout, 8 * r + (lane >> 3), 64);
                pv[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (jr >= 0) pv[r] = reinterpret_cast<const float4 *>(a.x + (int64_t)jr * 32)[lane & 7];
            }
        };
        auto round = [&](const float4 (&pv)[8], int32_t j, bool have) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < 8; ++r) *reinterpret_cast<float4 *>(&rows[8 * r + (lane >> 3)][4 * (lane & 7)]) = pv[r];
            wave_sync();
            if (have) {
                float dc = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    const float4 v = *reinterpret_cast<const float4 *>(&rows[lane][4 * c4]);
                    float df;
                    df = v.x - qrow[4 * c4 + 0]; dc = __builtin_fmaf(df, df, dc);
                    df = v.y - qrow[4 * c4 + 1]; dc = __builtin_fmaf(df, df, dc);
                    df = v.z - qrow[4 * c4 + 2]; dc = __builtin_fmaf(df, df, dc);
                    df = v.w - qrow[4 * c4 + 3]; dc = __builtin_fmaf(df, df, dc);
                }
                // sorted insert by (d, j) (R2)
#pragma unroll
                for (int p = KP - 1; p >= 1; --p) {
                    const bool gq = kd[p - 1] > dc || (kd[p - 1] == dc && kj[p - 1] > j);
                    const bool gp = kd[p] > dc || (kd[p] == dc && kj[p] > j);
                    const float dn = gq ? kd[p - 1] : (gp ? dc : kd[p]);
                    const int32_t jn = gq ? kj[p - 1] : (gp ? j : kj[p]);
                    kd[p] = dn; kj[p] = jn;
                }
                if (kd[0] > dc || (kd[0] == dc && kj[0] > j)) { kd[0] = dc; kj[0] = j; }
            }
        };
        float4 pa[8], pb[8];
        int32_t ja = -1, jb = -1;
        fetch(0, pa, ja);
        fetch(1, pb, jb);
        for (int c = 0; c < maxcnt; c += 2) {
            {
                float4 cur[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) cur[r] = pa[r];
                const int32_t jc = ja;
                if (c + 2 < maxcnt) fetch(c + 2, pa, ja);
                round(cur, jc, c < mycnt);
            }
            if (c + 1 < maxcnt) {
                float4 cur[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) cur[r] = pb[r];
                const int32_t jc = jb;
                if (c + 3 < maxcnt) fetch(c + 3, pb, jb);
                round(cur, jc, c + 1 < mycnt);
            }
        }
        if (valid) {
            const int k = a.k;
            float kth = -1.0f;
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                if (p < k) {
                    a.nbr[(int64_t)myq * k + p] = kj[p];
                    a.dist[(int64_t)myq * k + p] = kd[p];
                }
                if (p == k - 1 && kj[p] >= 0) kth = kd[p];
            }
            if (a.nbr16) {
                uint16_t *r16 = a.nbr16 + (int64_t)myq * k;
                if ((k & 1) == 0) {   // two ids per dword store
#pragma unroll
                    for (int p = 0; p + 1 < KP; p += 2)
                        if (p < k)
                            reinterpret_cast<unsigned *>(r16)[p >> 1] =
                                (unsigned)local_id16(kj[p], ev_lo) | ((unsigned)local_id16(kj[p + 1], ev_lo) << 16);
                } else {
#pragma unroll
                    for (int p = 0; p < KP; ++p)
                        if (p < k) r16[p] = local_id16(kj[p], ev_lo);
                }
            }
            // certificate: a list that saw at least M keys dropped only keys >= tau
            const float tau = L.tk[M - 1];
            const float nx = a.nrm[myq];
            const float an = __builtin_sqrtf(nx) * 1.000001f;
            const float rn = an + __builtin_sqrtf(fmaxf(kth, 0.0f)) * 1.00002f;
            const float slack = 2.0f * (4e-5f * an * rn + 1e-5f * rn * rn + 4e-6f * an * an) + 1e-30f;
            const bool full = tau < kKnnSentinel;
            if (L.overflow || !fits || (full && !(tau + nx - slack > kth))) {
                flag_query(a, myq, a.xtile_ptr[pos] + (myq - ev_lo) / a.xtile_queries);
            }
        }
        return;
    }
    // ---- split (tail) items: hand the partial list to knn_rerank_kernel, which merges the sub-sweeps ---------------
    if (a.no_rerank) {   // the caller's size hint ruled this event out and the merge launch was dropped: exact path
        if (valid && sub == 0) flag_query(a, myq, a.xtile_ptr[pos] + (myq - ev_lo) / a.xtile_queries);
        return;
    }
    if (valid) {
        const int64_t slot = (int64_t)(tile - n_full) * kFQ + hh * 32 + col;
        float *ld = a.psd + (slot * nsub + sub) * MS;
        int32_t *lj = a.psj + (slot * nsub + sub) * MS;
        // keys below the threshold first, then ties at the threshold until the list is full (a dropped tie has
        // key == threshold, which is what the certification assumes of dropped candidates)
        const float tfin = L.tk[M - 1];
        int out = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int s = 0; s < L.cnt; ++s) {
                const float key = __uint_as_float(Q.key[s][lane]);
                if ((pass == 0 ? key < tfin : key == tfin) && out < M) {
                    ld[out] = key;
                    lj[out] = ev_lo + (int32_t)Q.id[s][lane];
                    ++out;
                }
            }
        for (; out < M; ++out) { ld[out] = kKnnSentinel; lj[out] = -1; }
        ld[M] = L.tk[M - 1];              // the list's admission threshold (sentinel while fewer than M keys were seen)
        lj[M] = (L.overflow || !fits) ? 1 : 0;
    }
}

// first form alone: DMET_KNN_FILTER=1 (every event), A/B timing
template <int KP>
__global__ __launch_bounds__(kWave * kWavesPerGroup, 2) void knn_filter_kernel(const KnnFilterArgs a)
{
    __shared__ FilterQueue<filter_queue_len(filter_list_len(KP))> queue_all[kWavesPerGroup];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    filter1_wave<KP>(a, queue_all[wv], (int)blockIdx.x, wv, lane);
}


// ---- filter, second form ("tile masks"): events of kF2MinNodes .. 65536 nodes ------------------------------------------
// Budget of the first form at 64 x 4500 x 32 (tools/knn_budget.sh, ablation builds): once the operands arrive as
// contiguous records the kernel is bound by the per-key push -- two LDS stores per key on a 64 B/clk store path plus
// ~5 VALU ops -- and by the drains that re-read the queue.  This form keeps the per-key work to a compare and a carry
// add (a 32-bit hit mask per lane and tile) and touches LDS once per TILE:
//   * tau of a query = the M-th smallest TILE MINIMUM seen so far (a v_min3 tree over the lane's 32 keys, then the
//     v_med3 insertion chain once per tile): every one of those M tiles holds a key <= tau, so at least M >= k
//     candidates lie at or below it, and it is refreshed every tile instead of every drain;
//   * mask bit r = key of candidate row r < tau (the value before this tile's update); a tile with a non-zero mask
//     appends ONE 8-byte entry {mask, tile minimum | tile number};
//   * entries whose tile minimum exceeds the current tau are dropped when a lane runs out of slots and once at the
//     end: what survives are the ~M tiles that hold the list's keys.  Every dropped candidate -- an unset bit, a tile
//     that was never appended, a dropped entry -- had key >= the tau of its time >= the final tau: the same
//     certificate as the first form with T = tk[M-1];
//   * the first kF2Defer tiles only feed tau; they are swept again at the end against the final tau (their masks
//     would otherwise be nearly full: tau is still the sentinel there);
//   * the exact re-rank walks the set bits of the surviving entries.
#ifndef DMET_F2_DEFER
#define DMET_F2_DEFER 32
#endif
constexpr int kF2Defer = DMET_F2_DEFER;   // tiles that only feed tau in the main sweep
// DMET_F2_LEAN (experiment builds, tools/knn_lean.sh): THREE wavefronts per SIMD -- 26 entry slots, no row staging area
// (13 KB of LDS per wavefront, 12 wavefronts per CU), registers capped at 168 by the launch bounds
#ifdef DMET_F2_LEAN
constexpr int kF2Slots = 26;
constexpr int kF2StageRows = 4;
constexpr int kF2WavesPerSimd = 3;
#else
constexpr int kF2Slots = 30;        // entries per lane
constexpr int kF2StageRows = kWave;
constexpr int kF2WavesPerSimd = 2;
#endif
constexpr int kF2RowF = 16;         // features staged per re-rank half round
constexpr unsigned kF2TileBits = 11u, kF2TileMask = (1u << kF2TileBits) - 1u;

struct F2Wave {
    uint2 ent[kF2Slots][kWave];              // 15 360 B
    float rows[kF2StageRows][kF2RowF + 4];   //  5 120 B: half rows of the re-rank (16-byte aligned, conflict-free b128)
};
#ifndef DMET_F2_LEAN
static_assert(sizeof(F2Wave) == 20480, "two workgroups of four wavefronts fill the CU's 160 KB exactly");
#else
static_assert(sizeof(F2Wave) * 12 <= 163840, "three workgroups of four wavefronts per CU");
#endif

template <int M>
struct F2Lane {
    float tk[M];     // the M smallest tile minima, sorted
    float tau;       // admission threshold (= tk[M-1]; -inf for idle lanes / after an overflow)
    int cnt;         // entries in the lane's queue
    bool overflow;
};

// Drop the entries whose tile minimum is above tau.  The stored minimum carries the tile number in its low 11 mantissa
// bits; clearing them is monotone in the float order (x <= y => trunc(x) <= trunc(y)), so "trunc(stored) <= trunc(tau)"
// keeps every tile with minimum <= tau and at most the tiles within 2^-12 |tau| above it.  (Round 2, second session:
// the comparison used to allow 2^-10 |tau| on either side; with M = 22 one query per build of the benchmark's
// embeddings had five tile minima inside that window, ended with 27 entries and was handed to the fallback.)
// `limit`: entries a lane may keep -- during a sweep it needs room to append before the next compaction, at the end
// every slot may be in use.
template <int M>
__device__ __forceinline__ void f2_compact(F2Lane<M> &L, F2Wave &S, int lane, int limit)
{
    const float tauT = __uint_as_float(__float_as_uint(L.tau) & ~kF2TileMask);
    int maxcnt = L.cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, off, 64));
    int out = 0;
    for (int s = 0; s < maxcnt; ++s) {
        if (s < L.cnt) {
            const uint2 e = S.ent[s][lane];
            if (!(__uint_as_float(e.y & ~kF2TileMask) > tauT)) { S.ent[out][lane] = e; ++out; }
        }
    }
    L.cnt = out;
    if (out > limit) {   // too many tiles tied at tau: leave the query to the exact path
        L.overflow = true;
        L.cnt = 0;
        L.tau = -__builtin_inff();
    }
}

// Operands of one candidate tile (A fragments + accumulator seed): single-term fp16 records (see knn_prep_kernel).
template <int NH = 1>
struct F2Ops {
    f16x8 a[2 * NH];
    f32x16 c;
};

// One 32(candidates) x 32(queries) block: acc = cinit + sum over the 16-feature k-blocks of h.h' (fp16 operands, fp32
// accumulate): 2 NH MFMAs.  Operand map of v_mfma_f32_32x32x16_f16: lane (r = lane & 31, hh = lane >> 5) holds
// A[row r][k = 8 hh + 0..7].
template <int NH = 1>
__device__ __forceinline__ f32x16 f2_block(const f16x8 (&av)[2 * NH], const f16x8 (&bv)[2 * NH], const f32x16 &cinit)
{
    f32x16 acc = cinit;
#pragma unroll
    for (int m = 0; m < 2 * NH; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[m], bv[m], acc, 0, 0, 0);
    return acc;
}

// one fp16 candidate tile record: the lane's 2 NH A operands and the squared norms of the 16 candidate rows it receives
// results for (accumulator seed; rows (e & 3) + 8 (e >> 2) + 4 hh)
template <int NH = 1>
__device__ __forceinline__ void f2_load(F2Ops<NH> &o, const uint8_t *__restrict__ rec, int64_t tidx, int lane, int hh)
{
    const uint8_t *base = rec + tidx * rec_bytes(NH);
    const f16x8 *g = reinterpret_cast<const f16x8 *>(base);
#pragma unroll
    for (int m = 0; m < 2 * NH; ++m) o.a[m] = g[m * 64 + lane];
    const float4 *nr = reinterpret_cast<const float4 *>(base + kRec16FragBytes * NH);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = nr[2 * q + hh];
        o.c[4 * q] = v.x; o.c[4 * q + 1] = v.y; o.c[4 * q + 2] = v.z; o.c[4 * q + 3] = v.w;
    }
}

// Certificate slack of the second form (dropped candidates had key >= T; true d >= T + |x_i|^2 - slack).  an >= |x_i|,
// rn >= |x_i| + sqrt(d_k): a candidate with a larger norm than rn is farther than d_k by the triangle inequality, so the
// bound only has to hold for |x_j| <= rn.  Terms:
//   * the bound of the split form (fp32 accumulation inside the MFMAs, squared norms, the R1 chain itself), `scale` x
//     (1.5 at 64 features: twice the products per key), with its 2 x margin;
//   * fp16 operands: each rounds with relative error <= 2^-11 (normal range; |v| < 16384 is guaranteed by the record
//     writer), the products are exact in fp32, so |x.x' - h.h'| <= (2^-10 + 2^-22) sum_c |x_c||x'_c| <= 1.0003 x 2^-10
//     |x||x'| (Cauchy-Schwarz), twice that on the key: 2^-9 an rn, taken as 1.96e-3 (> 1.0003 x 2^-9 = 1.9537e-3);
//   * fp16 subnormals (|v| < 2^-14): absolute error <= 2^-25 per feature, on the key <= 2 x 2^-25 x sqrt(D) (|x|+|x'|)
//     <= 4.8e-7 (an + rn) at D <= 64; taken as 6e-7.
__device__ __forceinline__ float f2_slack(float an, float rn, float scale)
{
    return 2.0f * scale * (4e-5f * an * rn + 1e-5f * rn * rn + 4e-6f * an * an) + 1.96e-3f * an * rn + 6e-7f * (an + rn) +
           1e-30f;
}

// threshold list length of the second form: the fp16 slack needs the M-th smallest tile minimum two ranks further out
// than the split form did (measured on the model's embeddings at k = 16: uncertified queries per 4500-node event
// ~10 at KP + 4, ~2 at KP + 5, ~0.1 at KP + 6); a lane keeps kF2Slots = 30 entries, so the widest list stays at 24
constexpr int f2_list_len(int KP) { return KP <= 16 ? KP + 6 : KP + 4; }

// candidate row (0..31) of hit-mask position p (counted from the most significant bit), see f2_tile
__device__ __forceinline__ int f2_mask_row(int p) { return (p & 3) + 8 * ((p & 15) >> 2) + 4 * (p >> 4); }

// One tile of a sweep, software-pipelined inside the wavefront: the 12 MFMAs of tile t + 1 (operands `use`) are
// issued between the vector instructions that select from tile t's keys (c0, c1, computed one call earlier), and the
// operands of tile t + 2 are loaded into `ld`.  On gfx950 independent VALU work of the same wavefront hides under an
// MFMA (tools/mfma_overlap_micro.hip: 12 v_add per 32x32x16 MFMA interleaved cost 62 cycles per slot against 85 when
// the two run in phases), but only if it is in program order between the MFMAs: the scheduler is told to emit
// 1 MFMA + 11 VALU groups.  UPD: the tile minima feed tk / tau;  REC: hit masks are recorded.
// INS: this call inserts min(carry, its tile minimum) into the threshold list (every second tile: the list then holds the
// M smallest minima of tile PAIRS -- still M groups that each contain a key <= tau); otherwise it only updates `carry`.
// NH = 2 (64 features): `use` and `ld` are the SAME operand set (two sets of eight fragments next to the sixteen of
// the queries do not fit the register file at two wavefronts per SIMD), reloaded right after its MFMAs were issued.
template <int M, bool UPD, bool REC, bool INS, int NH = 1>
__device__ __forceinline__ void f2_tile(F2Lane<M> &L, F2Wave &S, const uint8_t *__restrict__ rec, int64_t rbase, int t,
                                        int t_hi, f32x16 &c0, f32x16 &c1, f32x16 &n0, f32x16 &n1, const F2Ops<NH> &use,
                                        F2Ops<NH> &ld, const f16x8 (&bq)[2][2 * NH], int lane, int hh, bool alive,
                                        float &carry)
{
#if defined(DMET_F2_ABL) && DMET_F2_ABL >= 2
    constexpr bool kRec = false;     // cycle-budget experiment (tools/knn_budget2.sh)
#else
    constexpr bool kRec = REC;
#endif
#if defined(DMET_F2_ABL) && DMET_F2_ABL >= 3
    constexpr bool kUpd = false;
#else
    constexpr bool kUpd = UPD;
#endif
    if (kRec) {
        if (__any(L.cnt >= kF2Slots - 1)) f2_compact<M>(L, S, lane, kF2Slots - 3);
    }
#if defined(DMET_F2_SAMEREC)
    f2_load<NH>(ld, rec, rbase + (t & 1), lane, hh);   // experiment: operands always cache-resident
#else
    f2_load<NH>(ld, rec, rbase + min(t + 2, t_hi - 1), lane, hh);   // clamped: the last two calls re-read the last tile
#endif
    n0 = f2_block<NH>(use.a, bq[0], use.c);     // (s_setprio 1 around these was measured: 10 % slower)
    n1 = f2_block<NH>(use.a, bq[1], use.c);
    // The accumulators stay where the MFMAs left them: lane (col, hh) holds, for candidate rows (e & 3) + 8 (e >> 2) + 4 hh,
    // the keys of query (0, col) in c0 and of query (1, col) in c1 -- 16 keys of each of the two queries the lane PAIR
    // (col, 0), (col, 1) owns.  Every lane reduces both halves it holds (hit mask against the owner's threshold, minimum)
    // and the pair exchanges the REDUCED values: v_permlane32_swap(V0, V1) trades V0 of lanes 32..63 for V1 of lanes
    // 0..31, so with V0 = "my part for query (0, col)" and V1 = "my part for query (1, col)" every lane ends up with
    // V0 = the hh = 0 rows' part and V1 = the hh = 1 rows' part of ITS OWN query.  Three swaps per tile (thresholds,
    // masks, minima) instead of the sixteen that moved the accumulators themselves (second session of round 2; a swap
    // costs two issue slots and sat between the MFMA results and everything else).
    unsigned mask = 0u;
    if (kRec) {
        const auto tt = __builtin_amdgcn_permlane32_swap(__float_as_uint(L.tau), __float_as_uint(L.tau), false, false);
        const float t0 = __uint_as_float(tt[0]), t1 = __uint_as_float(tt[1]);   // thresholds of query (0, col) / (1, col)
        // Two VALU ops per key: key - tau, then v_alignbit shifts its sign bit into the mask (a NaN key may set a bit:
        // its exact distance is NaN and never enters the result).  Element e of a half ends up in bit 15 - e.
        // (v_pk_add_f32 for two rows at once was measured: no gain -- packed fp32 issues at half rate here)
        unsigned ma = 0u, mb = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            ma = __builtin_amdgcn_alignbit(ma, __float_as_uint(c0[e] - t0), 31);
            mb = __builtin_amdgcn_alignbit(mb, __float_as_uint(c1[e] - t1), 31);
        }
        const auto mm = __builtin_amdgcn_permlane32_swap(ma, mb, false, false);
        // bit 31 - p: p < 16 -> element p of the hh = 0 rows, else element p - 16 of the hh = 1 rows (f2_mask_row)
        mask = (mm[0] << 16) | mm[1];
    }
    float tmin = -__builtin_inff();   // deferred tiles: "never drop" (their tau is already final)
    if (kUpd) {
        float na = kKnnSentinel, nb = kKnnSentinel;   // (the start value also keeps a NaN key out of the v_med3 chain)
        // v_min3_f32 by hand: fminf() makes hipcc canonicalise every MFMA output with a v_max first (twice the ops)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(na) : "v"(c0[e]), "v"(c0[e + 1]));
            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(nb) : "v"(c1[e]), "v"(c1[e + 1]));
        }
        const auto nn = __builtin_amdgcn_permlane32_swap(__float_as_uint(na), __float_as_uint(nb), false, false);
        tmin = __uint_as_float(nn[0]);
        asm("v_min_f32 %0, %0, %1" : "+v"(tmin) : "v"(__uint_as_float(nn[1])));
    }
    if (kRec) {
        // one 8-byte entry per tile and lane, kept only when the mask is non-zero (branch-free append)
        const unsigned packed = (__float_as_uint(tmin) & ~kF2TileMask) | (unsigned)t;
        S.ent[L.cnt][lane] = make_uint2(mask, packed);
        L.cnt += (mask != 0u) ? 1 : 0;
    }
    if (kUpd) {
        if (INS) {
            // a tile that holds a forced candidate (key -inf: a row outside the fp16 range, see knn_prep_kernel) does
            // not vote for the threshold: its -inf would take a list slot without standing for a real key below tau
            float v = tmin < -3.0e38f ? kKnnSentinel : tmin;
            asm("v_min_f32 %0, %0, %1" : "+v"(v) : "v"(carry));     // (both operands are clamped to the sentinel: no NaN)
            carry = kKnnSentinel;
#pragma unroll
            for (int p = M - 1; p >= 1; --p) L.tk[p] = __builtin_amdgcn_fmed3f(L.tk[p - 1], v, L.tk[p]);
            asm("v_min_f32 %0, %0, %1" : "+v"(L.tk[0]) : "v"(v));
            if (alive && !L.overflow) L.tau = L.tk[M - 1];
        } else {
            carry = tmin;
        }
    }
#ifdef DMET_F2_SCHED
    // experiment: force 1 MFMA + 11 VALU groups (measured 4 % SLOWER than hipcc's own order at two wavefronts per SIMD)
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (kRec ? 11 : (kUpd ? 5 : 2)), 0);
    }
#endif
}

// One pass over the tiles [t_lo, t_hi) of the event whose first record is rbase.
template <int M, bool UPD, bool REC, int NH = 1>
__device__ __forceinline__ void f2_sweep(F2Lane<M> &L, F2Wave &S, const uint8_t *__restrict__ rec, int64_t rbase,
                                         int t_lo, int t_hi, const f16x8 (&bq)[2][2 * NH], int lane, int hh, bool alive)
{
    if (t_lo >= t_hi) return;
    F2Ops<NH> A, B;
    f2_load<NH>(A, rec, rbase + t_lo, lane, hh);
    f32x16 c0 = f2_block<NH>(A.a, bq[0], A.c);      // prologue: the first tile's keys
    f32x16 c1 = f2_block<NH>(A.a, bq[1], A.c);
    f32x16 n0, n1;
    f2_load<NH>(A, rec, rbase + min(t_lo + 1, t_hi - 1), lane, hh);
    float carry = kKnnSentinel;
    for (int t = t_lo; t < t_hi; t += 2) {
        // every tile inserts its own minimum (INS = true).  Inserting the minimum of tile PAIRS instead (half the
        // v_med3 chains) was tried: the threshold then admits up to 2M tiles, more than the 26 entries a lane can keep
        // -> 3 761 overflowed queries per launch and twice the kernel time
        f2_tile<M, UPD, REC, true, NH>(L, S, rec, rbase, t, t_hi, c0, c1, n0, n1, A, B, bq, lane, hh, alive, carry);
        if (t + 1 < t_hi)
            f2_tile<M, UPD, REC, true, NH>(L, S, rec, rbase, t + 1, t_hi, n0, n1, c0, c1, B, A, bq, lane, hh, alive, carry);
    }
}

// Plan group of a workgroup.  Whole-sweep tiles: the workgroups of one XCD take one contiguous eighth of the tile list
// (see xcd_dealt_position); the split tail tiles that follow stay interleaved over the XCDs.
__device__ __forceinline__ int filter_group(const KnnFilterArgs &a)
{
    const int full_groups = a.plan->n_full / kWavesPerGroup;   // n_full is a multiple of the SIMD count
    return (int)blockIdx.x < full_groups ? xcd_swizzle((int)blockIdx.x, full_groups) : (int)blockIdx.x;
}

// One wavefront's item of the second form; S is the wavefront's own LDS (no workgroup barrier inside).
template <int KP, int NH = 1>
__device__ __forceinline__ void filter2_wave(const KnnFilterArgs &a, F2Wave &S, int *tickets, int group, int wv, int lane)
{
    constexpr int M = f2_list_len(KP);
    constexpr int D = 32 * NH;
    constexpr int MS = (M + 1 + 3) & ~3;
    const int col = lane & 31, hh = lane >> 5;
    const uint8_t *__restrict__ rec = a.rec;
    const int64_t *__restrict__ ptr = a.ptr;

    const int n_full = a.plan->n_full, split = a.plan->split, total = a.plan->total_tiles;
    const int item = group * kWavesPerGroup + wv;
    int tile = item, sub = 0, nsub = 1;
    if (item >= n_full) {
        const int r = item - n_full;
        tile = n_full + r / split;
        sub = r % split;
        nsub = split;
    }
    if (tile >= total) return;
    const int pos = find_tile_event(a.tile_ptr, a.B, tile);
    const int ev = a.order[pos];
    const int ev_lo = (int)ptr[ev], ev_hi = (int)ptr[ev + 1];
    const int q_first = ev_lo + (tile - a.tile_ptr[pos]) * kFQ;
    if (!f2_in_domain(ev_hi - ev_lo)) {
        // the first form's events.  D = 64 has no first form: every query of such an event is handed to the exact
        // kernel (once per tile: a split tile comes by `split` times)
        if (NH != 1 && sub == 0) {
            const int q = q_first + lane;
            if (q < ev_hi) {
                flag_query(a, q, a.xtile_ptr[pos] + (q - ev_lo) / a.xtile_queries);
            }
        }
        return;
    }
    int clo = ev_lo, chi = ev_hi;
    bool idle_piece = false;
    if (nsub > 1) {
        if (ev_hi - ev_lo < kF2SplitMinNodes) {
            // a sub-sweep of fewer than ~2 M tiles has no threshold to speak of (it would hand most of its range to the
            // exact re-rank): the first piece sweeps the whole event, the others bring an empty list to the merge
            idle_piece = sub != 0;
        } else {
            const int chunk = (((chi - clo) + nsub - 1) / nsub + 31) & ~31;
            clo = min(chi, clo + sub * chunk);
            chi = min(chi, clo + chunk);
        }
    }
    const int64_t rbase = (ptr[ev] >> 5) + ev;
    const int64_t rlast = rbase + (ev_hi - ev_lo - 1) / 32;
    const int t_lo = (clo - ev_lo) / 32, t_hi = idle_piece ? t_lo : (chi - ev_lo + 31) / 32;

    const int myq = q_first + hh * 32 + col;
    const bool valid = myq < ev_hi;
    // Second attempt (whole-sweep items only).  A query whose certificate fails by the slack alone -- enough candidates,
    // but the threshold T too close to its k-th distance: T + |x|^2 - slack <= d_k -- does not need the exact kernels:
    // the wavefront sweeps the event once more with the FIXED threshold T* = d_k - |x|^2 + slack for those lanes
    // (-inf, i.e. nothing recorded, for the others), re-ranks what that admits and certifies against T*: every candidate
    // dropped by that sweep has key >= T*, hence d >= d_k >= the new k-th distance.  With fp16 operands ~0.1 queries per
    // 4500-node event take this road (one wavefront in ~200 pays a second, masks-only sweep) instead of ~100 us of
    // per-query fallback per build.
    float t_fix = -__builtin_inff();
    bool act = valid;              // lanes whose result this attempt writes and certifies
    for (int attempt = 0;; ++attempt) {
    F2Lane<M> L;
#pragma unroll
    for (int p = 0; p < M; ++p) L.tk[p] = kKnnSentinel;
    L.tau = attempt == 0 ? -__builtin_inff() : t_fix;     // first attempt: nothing is recorded before tk is full
    L.cnt = 0;
    L.overflow = false;
    {
        // the query operands live only as long as the sweeps (the re-rank needs the registers for the rows)
        f16x8 bq[2][2 * NH];      // -2 x the queries' fp16 fragments (exact: |h| <= 16384)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t qrec = min(rbase + (q_first - ev_lo) / 32 + b, rlast);
            const f16x8 *g = reinterpret_cast<const f16x8 *>(rec + qrec * rec_bytes(NH));
#pragma unroll
            for (int m = 0; m < 2 * NH; ++m) bq[b][m] = g[m * 64 + lane] * (_Float16)-2.0f;
        }
        if (attempt == 0) {
            const int t_def = min(t_hi, t_lo + kF2Defer);
            f2_sweep<M, true, false, NH>(L, S, rec, rbase, t_lo, t_def, bq, lane, hh, valid);     // tau only
            f2_sweep<M, true, true, NH>(L, S, rec, rbase, t_def, t_hi, bq, lane, hh, valid);
            f2_sweep<M, false, true, NH>(L, S, rec, rbase, t_lo, t_def, bq, lane, hh, valid);     // against the final tau
        } else {
            f2_sweep<M, false, true, NH>(L, S, rec, rbase, t_lo, t_hi, bq, lane, hh, valid);      // against T*
        }
    }
    f2_compact<M>(L, S, lane, kF2Slots);
#if defined(DMET_F2_ABL) && DMET_F2_ABL >= 1
    return;
#endif

    // ---- exact re-rank of the set bits (R1 chain, top-k by (d, j)), candidates fetched cooperatively ----------------
    const int64_t qrow_id = valid ? myq : ev_lo;
    float qrow[D];
    {
        const float4 *g = reinterpret_cast<const float4 *>(a.x + qrow_id * D);
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
            const float4 v = g[c];
            qrow[4 * c] = v.x; qrow[4 * c + 1] = v.y; qrow[4 * c + 2] = v.z; qrow[4 * c + 3] = v.w;
        }
    }
    // sorted top-KP as 64-bit words (distance bits << 32 | j): distances are >= +0, so the unsigned order IS the (d, j)
    // order of R2, one v_cmp_gt_u64 per slot and no branches.  Empty slots are (sentinel, 0): a candidate at exactly
    // the sentinel distance (or NaN / inf: larger bit patterns) is never inserted, like the oracle's strict '>'.
    unsigned long long kk[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) kk[p] = (unsigned long long)__float_as_uint(kKnnSentinel) << 32;
    const int nent = (act && !L.overflow) ? L.cnt : 0;
    int slot = 0;
    unsigned cmask = 0u;
    int ctile = 0;
    // next candidate of this lane (tiles as appended, mask order inside a tile), -1 when exhausted
    auto pop = [&]() __attribute__((always_inline)) -> int32_t {
        if (cmask == 0u && slot < nent) {
            const uint2 e = S.ent[slot][lane];
            cmask = e.x;
            ctile = (int)(e.y & kF2TileMask);
            ++slot;
        }
        int32_t j = -1;
        if (cmask != 0u) {
            const int r = __builtin_clz(cmask);
            cmask &= ~(0x80000000u >> r);
            j = ev_lo + ctile * 32 + f2_mask_row(r);   // < ev_hi: rows past the event's end have key = +inf and are never set
        }
        return j;
    };
    const float4 *x4 = reinterpret_cast<const float4 *>(a.x);
#ifdef DMET_RR_PRIO
    __builtin_amdgcn_s_setprio(DMET_RR_PRIO);   // experiment: the latency-bound phase issues ahead of the other wavefront's sweep
#endif
    if constexpr (NH != 1) {
        // D = 64: every lane fetches the row of its own candidate (sixteen 16-byte loads in flight) and runs the chain
        // on it -- none of the cooperative staging of the 32-wide form below, whose register budget (three rounds of
        // half rows in flight) does not carry over; the sweep, not this loop, is the larger part at this width
        for (;;) {
            const int32_t j = pop();
            if (!__any(j >= 0)) break;
            const float4 *row = x4 + (int64_t)(j >= 0 ? j : ev_lo) * (D / 4);
            float4 v[D / 4];
#pragma unroll
            for (int c = 0; c < D / 4; ++c) v[c] = row[c];
            float dc = 0.0f;
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                float df;
                df = v[c].x - qrow[4 * c + 0]; dc = __builtin_fmaf(df, df, dc);
                df = v[c].y - qrow[4 * c + 1]; dc = __builtin_fmaf(df, df, dc);
                df = v[c].z - qrow[4 * c + 2]; dc = __builtin_fmaf(df, df, dc);
                df = v[c].w - qrow[4 * c + 3]; dc = __builtin_fmaf(df, df, dc);
            }
            const unsigned long long nk =
                j >= 0 ? (((unsigned long long)__float_as_uint(dc) << 32) | (unsigned)j) : ~0ull;
            bool g[KP];
#pragma unroll
            for (int p = 0; p < KP; ++p) g[p] = kk[p] > nk;
#pragma unroll
            for (int p = KP - 1; p >= 1; --p) kk[p] = g[p - 1] ? kk[p - 1] : (g[p] ? nk : kk[p]);
            kk[0] = g[0] ? nk : kk[0];
        }
    } else {
    // rows are fetched half a row at a time (16 features = 64 bytes): load instruction 4 h + r brings half h of rows
    // 16 r + (lane >> 2), 16 bytes per lane; exhausted lanes re-read the event's first row (no branches, result unused)
    struct HalfRows { float4 v0, v1, v2, v3, v4, v5, v6, v7; };   // named members: stays in registers
#ifdef DMET_RR_FULLROW
    // experiment: one load instruction brings 8 WHOLE rows (8 lanes x 16 bytes = one 128-byte line per row) instead of 16
    // half rows -- half the line look-ups per round
    auto fetch = [&](int32_t j) __attribute__((always_inline)) -> HalfRows {
        const int32_t jc = j >= 0 ? j : ev_lo;
        const int64_t o = lane & 7;
        const int sub = lane >> 3;
        HalfRows R;
        R.v0 = x4[(int64_t)__shfl(jc, 0 + sub, 64) * 8 + o];
        R.v1 = x4[(int64_t)__shfl(jc, 8 + sub, 64) * 8 + o];
        R.v2 = x4[(int64_t)__shfl(jc, 16 + sub, 64) * 8 + o];
        R.v3 = x4[(int64_t)__shfl(jc, 24 + sub, 64) * 8 + o];
        R.v4 = x4[(int64_t)__shfl(jc, 32 + sub, 64) * 8 + o];
        R.v5 = x4[(int64_t)__shfl(jc, 40 + sub, 64) * 8 + o];
        R.v6 = x4[(int64_t)__shfl(jc, 48 + sub, 64) * 8 + o];
        R.v7 = x4[(int64_t)__shfl(jc, 56 + sub, 64) * 8 + o];
        return R;
    };
#else
    auto fetch = [&](int32_t j) __attribute__((always_inline)) -> HalfRows {
        const int32_t jc = j >= 0 ? j : ev_lo;
        const int64_t o = 4 * 0 + (lane & 3);
        const int64_t r0 = (int64_t)__shfl(jc, 0 + (lane >> 2), 64) * 8 + o;
        const int64_t r1 = (int64_t)__shfl(jc, 16 + (lane >> 2), 64) * 8 + o;
        const int64_t r2 = (int64_t)__shfl(jc, 32 + (lane >> 2), 64) * 8 + o;
        const int64_t r3 = (int64_t)__shfl(jc, 48 + (lane >> 2), 64) * 8 + o;
        HalfRows R;
        R.v0 = x4[r0]; R.v1 = x4[r1]; R.v2 = x4[r2]; R.v3 = x4[r3];
        R.v4 = x4[r0 + 4]; R.v5 = x4[r1 + 4]; R.v6 = x4[r2 + 4]; R.v7 = x4[r3 + 4];
        return R;
    };
#endif
    auto chain16 = [&](float dc, int h) __attribute__((always_inline)) -> float {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 v = *reinterpret_cast<const float4 *>(&S.rows[lane][4 * c4]);
            const int f0 = 16 * h + 4 * c4;
            float df;
            df = v.x - qrow[f0 + 0]; dc = __builtin_fmaf(df, df, dc);
            df = v.y - qrow[f0 + 1]; dc = __builtin_fmaf(df, df, dc);
            df = v.z - qrow[f0 + 2]; dc = __builtin_fmaf(df, df, dc);
            df = v.w - qrow[f0 + 3]; dc = __builtin_fmaf(df, df, dc);
        }
        return dc;
    };
    auto round = [&](const HalfRows &R, int32_t j) __attribute__((always_inline)) {
#ifdef DMET_RR_FULLROW
        float4 *dst = reinterpret_cast<float4 *>(&S.rows[lane >> 3][4 * (lane & 3)]);   // + 8 r rows per register
        constexpr int kS8 = 8 * (kF2RowF + 4) / 4;                                        // float4s per 8 rows
        const bool lowhalf = (lane & 4) == 0;
        wave_sync();
        if (lowhalf) {
            dst[0] = R.v0; dst[kS8] = R.v1; dst[2 * kS8] = R.v2; dst[3 * kS8] = R.v3;
            dst[4 * kS8] = R.v4; dst[5 * kS8] = R.v5; dst[6 * kS8] = R.v6; dst[7 * kS8] = R.v7;
        }
        wave_sync();
        float dc = chain16(0.0f, 0);
        wave_sync();
        if (!lowhalf) {
            dst[0] = R.v0; dst[kS8] = R.v1; dst[2 * kS8] = R.v2; dst[3 * kS8] = R.v3;
            dst[4 * kS8] = R.v4; dst[5 * kS8] = R.v5; dst[6 * kS8] = R.v6; dst[7 * kS8] = R.v7;
        }
        wave_sync();
        dc = chain16(dc, 1);
#else
        float4 *dst = reinterpret_cast<float4 *>(&S.rows[lane >> 2][4 * (lane & 3)]);   // + 16 r rows per register
        constexpr int kStride = 16 * (kF2RowF + 4) / 4;                                    // float4s per 16 rows
        wave_sync();
        dst[0] = R.v0; dst[kStride] = R.v1; dst[2 * kStride] = R.v2; dst[3 * kStride] = R.v3;
        wave_sync();
        float dc = chain16(0.0f, 0);
        wave_sync();
        dst[0] = R.v4; dst[kStride] = R.v5; dst[2 * kStride] = R.v6; dst[3 * kStride] = R.v7;
        wave_sync();
        dc = chain16(dc, 1);
#endif
        const unsigned long long nk =
            j >= 0 ? (((unsigned long long)__float_as_uint(dc) << 32) | (unsigned)j) : ~0ull;
        bool g[KP];
#pragma unroll
        for (int p = 0; p < KP; ++p) g[p] = kk[p] > nk;
#pragma unroll
        for (int p = KP - 1; p >= 1; --p) kk[p] = g[p - 1] ? kk[p - 1] : (g[p] ? nk : kk[p]);
        kk[0] = g[0] ? nk : kk[0];
    };
    // Rounds of rows in flight: three (KP <= 16) were chosen in round 2 -- the kernel has since grown to 256 VGPRs + 56
    // bytes of scratch per lane with them (-Rpass-analysis=kernel-resource-usage), i.e. spill traffic inside this
    // latency-bound loop; with two it needs 237 registers and no scratch and the build is 8-10 us faster (second session
    // of round 3; DMET_RR_THREE brings the third back for A/B)
#ifdef DMET_RR_THREE
    constexpr bool kThreeRounds = KP <= 16;
#else
    constexpr bool kThreeRounds = false;
#endif
    if constexpr (kThreeRounds) {
        // three rounds of rows in flight: the loop is bound by the gathers' latency
        int32_t ja = pop(), jb, jc;
        HalfRows pa = fetch(ja), pb, pc;
        jb = pop();
        pb = fetch(jb);
        for (;;) {
            if (!__any(ja >= 0)) break;
            jc = pop(); pc = fetch(jc);
            round(pa, ja);
            if (!__any(jb >= 0)) break;
            ja = pop(); pa = fetch(ja);
            round(pb, jb);
            if (!__any(jc >= 0)) break;
            jb = pop(); pb = fetch(jb);
            round(pc, jc);
        }
    } else {
        // the 20-wide list leaves registers for two rounds in flight
        int32_t ja = pop(), jb;
        HalfRows pa = fetch(ja), pb;
        for (;;) {
            if (!__any(ja >= 0)) break;
            jb = pop(); pb = fetch(jb);
            round(pa, ja);
            if (!__any(jb >= 0)) break;
            ja = pop(); pa = fetch(ja);
            round(pb, jb);
        }
    }
    }
#ifdef DMET_RR_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    float kd[KP];
    int32_t kj[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        kd[p] = __uint_as_float((unsigned)(kk[p] >> 32));
        kj[p] = kd[p] == kKnnSentinel ? -1 : (int32_t)(unsigned)kk[p];
    }
    const int k = a.k;
    const float tau = attempt == 0 ? L.tk[M - 1] : t_fix;
    // the query's rows of the three tables; returns its k-th distance (-1: fewer than k neighbours)
    // coop (whole-sweep items, called by ALL lanes): the 64 queries of the item own 64 consecutive rows of each table, i.e.
    // one contiguous block; written by the lanes themselves that is k 4-byte stores per lane and table, each instruction a
    // 4-byte piece of 64 different lines (switching the stores off measured 22 of the build's 420 us).  The rows go through
    // the wavefront's LDS (free by now) instead and leave as 16-byte pieces of consecutive addresses; rows_on masks the
    // rows that are written (lanes past the event's end; the second attempt rewrites only its own queries).
    auto emit = [&](const bool rows_on, const bool coop) __attribute__((always_inline)) -> float {
        float kth = -1.0f;
#pragma unroll
        for (int p = 0; p < KP; ++p)
            if (p == k - 1 && kj[p] >= 0) kth = kd[p];
        if constexpr (KP % 4 == 0 && KP <= 20) {
            if (coop && a.emit_coalesced && k == KP) {
                constexpr int RP = KP / 4;       // 16-byte pieces per row
                const unsigned long long on = __ballot(rows_on);
                unsigned *stg = reinterpret_cast<unsigned *>(&S);        // 64 rows x KP words <= 5 120 bytes per table
                const int64_t blk = (int64_t)q_first * KP;               // first word of the block in nbr / dist
                wave_sync();
#pragma unroll
                for (int q = 0; q < RP; ++q)
                    *reinterpret_cast<uint4 *>(stg + lane * KP + 4 * q) =
                        make_uint4((unsigned)kj[4 * q], (unsigned)kj[4 * q + 1], (unsigned)kj[4 * q + 2], (unsigned)kj[4 * q + 3]);
                wave_sync();
#pragma unroll
                for (int t = 0; t < RP; ++t) {
                    const int pi = t * 64 + lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(stg + 4 * pi);
                    if ((on >> (pi / RP)) & 1ull) *reinterpret_cast<uint4 *>(a.nbr + blk + 4 * pi) = v;
                }
                wave_sync();
#pragma unroll
                for (int q = 0; q < RP; ++q)
                    *reinterpret_cast<uint4 *>(stg + lane * KP + 4 * q) =
                        make_uint4(__float_as_uint(kd[4 * q]), __float_as_uint(kd[4 * q + 1]), __float_as_uint(kd[4 * q + 2]),
                                   __float_as_uint(kd[4 * q + 3]));
                wave_sync();
#pragma unroll
                for (int t = 0; t < RP; ++t) {
                    const int pi = t * 64 + lane;
                    const uint4 v = *reinterpret_cast<const uint4 *>(stg + 4 * pi);
                    if ((on >> (pi / RP)) & 1ull) *reinterpret_cast<uint4 *>(a.dist + blk + 4 * pi) = v;
                }
                if (a.nbr16) {
                    if constexpr (KP % 8 == 0) {
                        constexpr int RH = KP / 8;   // 16-byte pieces per uint16 row
                        wave_sync();
#pragma unroll
                        for (int q = 0; q < RH; ++q) {
                            uint4 w;
                            w.x = (unsigned)local_id16(kj[8 * q], ev_lo) | ((unsigned)local_id16(kj[8 * q + 1], ev_lo) << 16);
                            w.y = (unsigned)local_id16(kj[8 * q + 2], ev_lo) | ((unsigned)local_id16(kj[8 * q + 3], ev_lo) << 16);
                            w.z = (unsigned)local_id16(kj[8 * q + 4], ev_lo) | ((unsigned)local_id16(kj[8 * q + 5], ev_lo) << 16);
                            w.w = (unsigned)local_id16(kj[8 * q + 6], ev_lo) | ((unsigned)local_id16(kj[8 * q + 7], ev_lo) << 16);
                            *reinterpret_cast<uint4 *>(stg + lane * (KP / 2) + 4 * q) = w;
                        }
                        wave_sync();
#pragma unroll
                        for (int t = 0; t < RH; ++t) {
                            const int pi = t * 64 + lane;
                            const uint4 v = *reinterpret_cast<const uint4 *>(stg + 4 * pi);
                            if ((on >> (pi / RH)) & 1ull)
                                *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned *>(a.nbr16 + blk) + 4 * pi) = v;
                        }
                    } else if (rows_on) {
                        uint16_t *r16 = a.nbr16 + (int64_t)myq * k;
#pragma unroll
                        for (int p = 0; p + 1 < KP; p += 2)
                            reinterpret_cast<unsigned *>(r16)[p >> 1] =
                                (unsigned)local_id16(kj[p], ev_lo) | ((unsigned)local_id16(kj[p + 1], ev_lo) << 16);
                    }
                }
                wave_sync();       // the staging area is the next attempt's entry list
                return kth;
            }
        }
        if (!rows_on) return kth;
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            if (p < k) {
                a.nbr[(int64_t)myq * k + p] = kj[p];
                a.dist[(int64_t)myq * k + p] = kd[p];
            }
        }
        if (a.nbr16) {
            uint16_t *r16 = a.nbr16 + (int64_t)myq * k;
            if ((k & 1) == 0) {   // two ids per dword store
#pragma unroll
                for (int p = 0; p + 1 < KP; p += 2)
                    if (p < k)
                        reinterpret_cast<unsigned *>(r16)[p >> 1] =
                            (unsigned)local_id16(kj[p], ev_lo) | ((unsigned)local_id16(kj[p + 1], ev_lo) << 16);
            } else {
#pragma unroll
                for (int p = 0; p < KP; ++p)
                    if (p < k) r16[p] = local_id16(kj[p], ev_lo);
            }
        }
        return kth;
    };
    if (nsub == 1) {
        bool retry = false;
        const float kth = emit(act, true);
        if (act) {
            // certificate: every dropped candidate had key >= tau (see the header of this form).  Candidates were
            // dropped (tau below the sentinel) but fewer than k neighbours came back (kth < 0): not certified either
            const float nx = a.nrm[myq];
            const float an = __builtin_sqrtf(nx) * 1.000001f;
            const float rn = an + __builtin_sqrtf(fmaxf(kth, 0.0f)) * 1.00002f;
            const float slack = f2_slack(an, rn, NH == 1 ? 1.0f : 1.5f);
            const bool full = tau < kKnnSentinel;
            // a query whose own row is outside the fp16 range (or not finite) swept with zero operands: never certified
            const bool wideq = !(nx < kF16WideLimit * kF16WideLimit);
            const bool fail = L.overflow || wideq || (full && !(kth >= 0.0f && tau + nx - slack > kth));
            // slack-only failures get the second attempt: the smallest threshold that certifies this k-th distance,
            // nudged up by a few ulps of the largest term so that the same fp32 expression holds for it
            float ts = kth - nx + slack;
            ts += (__builtin_fabsf(ts) + nx + slack) * 4.8e-7f + 1e-30f;
            retry = fail && attempt == 0 && !L.overflow && !wideq && kth >= 0.0f && ts + nx - slack > kth &&
                    ts < kKnnSentinel;
            if (fail && !retry) {
                flag_query(a, myq, a.xtile_ptr[pos] + (myq - ev_lo) / a.xtile_queries);
#ifdef DMET_KNN_WHY
                a.qflag[myq] = (uint8_t)(1 | (L.overflow ? 2 : 0) | (wideq ? 4 : 0) | (kth < 0.0f ? 8 : 0) | (attempt ? 16 : 0) |
                                         (!(ts < kKnnSentinel) ? 32 : 0) | (!(ts + nx - slack > kth) ? 64 : 0));
                a.dist[(int64_t)myq * k + 0] = tau; a.dist[(int64_t)myq * k + 1] = kth; a.dist[(int64_t)myq * k + 2] = slack; a.dist[(int64_t)myq * k + 3] = (float)L.cnt;
#endif
            }
            t_fix = retry ? ts : -__builtin_inff();
        }
        if (!__any(retry)) return;
        act = retry;
        continue;
    }
    // ---- split (tail) items: the exact top-KP of this candidate range + its threshold go to global memory; the two
    // sub-sweeps of a tile are neighbouring wavefronts of ONE workgroup (items 4g + {0,1} and 4g + {2,3}: n_full is a
    // multiple of 4 and the split is 2), and the one that finishes second merges the other's list into its own and
    // certifies against both thresholds -- a ticket in LDS and workgroup-scope fences, no launch of its own (the
    // separate merge kernel took 17 us per build behind the whole filter grid).
    static_assert(kFilterMaxSplit == 2 && kWavesPerGroup % 2 == 0, "pairs of sub-sweeps share a workgroup");
    const int64_t fslot = (int64_t)(tile - n_full) * kFQ + hh * 32 + col;
    if (valid) {
        float *ld = a.psd + (fslot * nsub + sub) * MS;
        int32_t *lj = a.psj + (fslot * nsub + sub) * MS;
#pragma unroll
        for (int p = 0; p < KP; ++p) { ld[p] = kd[p]; lj[p] = kj[p]; }
        ld[M] = tau;
        lj[M] = L.overflow ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(&tickets[wv >> 1], 1);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == 0) return;                 // the other sub-sweep of this tile is still running: it will merge
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float tau_o = kKnnSentinel;
    int of_o = 0;
    if (valid) {
        const float *ld = a.psd + (fslot * nsub + (sub ^ 1)) * MS;
        const int32_t *lj = a.psj + (fslot * nsub + (sub ^ 1)) * MS;
        tau_o = ld[M];
        of_o = lj[M];
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const float od = ld[p];
            const int32_t oj = lj[p];
            // (d, j) pairs of the two candidate ranges are distinct; empty slots are never inserted
            const unsigned long long nk = oj >= 0 ? (((unsigned long long)__float_as_uint(od) << 32) | (unsigned)oj) : ~0ull;
            bool g[KP];
#pragma unroll
            for (int q = 0; q < KP; ++q) g[q] = kk[q] > nk;
#pragma unroll
            for (int q = KP - 1; q >= 1; --q) kk[q] = g[q - 1] ? kk[q - 1] : (g[q] ? nk : kk[q]);
            kk[0] = g[0] ? nk : kk[0];
        }
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            kd[p] = __uint_as_float((unsigned)(kk[p] >> 32));
            kj[p] = kd[p] == kKnnSentinel ? -1 : (int32_t)(unsigned)kk[p];
        }
        const float kth = emit(true, false);      // (inside a divergent branch: every lane writes its own rows)
        const float nx = a.nrm[myq];
        const float an = __builtin_sqrtf(nx) * 1.000001f;
        const float rn = an + __builtin_sqrtf(fmaxf(kth, 0.0f)) * 1.00002f;
        const float slack = f2_slack(an, rn, NH == 1 ? 1.0f : 1.5f);
        const bool wideq = !(nx < kF16WideLimit * kF16WideLimit);
        const bool fail_a = tau < kKnnSentinel && !(kth >= 0.0f && tau + nx - slack > kth);
        const bool fail_b = tau_o < kKnnSentinel && !(kth >= 0.0f && tau_o + nx - slack > kth);
        if (L.overflow || of_o != 0 || wideq || fail_a || fail_b)
            flag_query(a, myq, a.xtile_ptr[pos] + (myq - ev_lo) / a.xtile_queries);
    }
    return;
    }   // attempts
}

// Both forms in ONE launch: a wavefront takes the form its item's event calls for.  Batches that mix event sizes
// (configs[4]: 500-8000 nodes) used to pay a second, nearly empty launch for their events below kF2MinNodes -- a few
// hundred long serial items on an otherwise idle chip (216 us at 64 events) -- which now run beside the second form's
// items.  The wavefront number is wave-uniform, but only readfirstlane tells the compiler: without it the tile, the
// event, the loop counters and every record address are computed per lane on the vector ALU.
template <int KP, int NH = 1>
__global__ __launch_bounds__(kWave * kWavesPerGroup, kF2WavesPerSimd) void knn_filter12_kernel(const KnnFilterArgs a)
{
    union WaveLds {
        F2Wave f2;
#ifndef DMET_F2_LEAN
        FilterQueue<filter_queue_len(filter_list_len(KP))> f1;
#endif
    };
    __shared__ WaveLds sh_all[kWavesPerGroup];
    if constexpr (NH == 1) {
        // rider workgroups (behind every filter workgroup of the grid: dispatched last, into the slots of the last round)
        if (a.rP != nullptr && (int)blockIdx.x >= a.first_rider) {
            static_assert(sizeof(WaveLds) >= sizeof(float) * kNlsLdsFloats, "a wavefront's LDS holds the transposition tiles");
            const int rwv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            const int64_t wave = (int64_t)((int)blockIdx.x - a.first_rider) * kWavesPerGroup + rwv;
            const int64_t nwaves = (int64_t)((int)gridDim.x - a.first_rider) * kWavesPerGroup;
            float *tp = reinterpret_cast<float *>(&sh_all[rwv]);
            if (a.r_sliced == 2)
                node_linear_split_bf16_wave<32, 32>(a.x, a.N, a.rW, a.rb, a.rP, reinterpret_cast<unsigned short *>(a.rQ), wave,
                                                    nwaves, threadIdx.x & 63);
            else if (a.r_sliced == 1) node_linear_split_wave<32, 32, true>(a.x, a.N, a.rW, a.rb, a.rP, a.rQ, tp, wave, nwaves, threadIdx.x & 63);
            else node_linear_split_wave<32, 32, false>(a.x, a.N, a.rW, a.rb, a.rP, a.rQ, tp, wave, nwaves, threadIdx.x & 63);
            return;
        }
    }
    // arrival tickets of the sub-sweep pairs of split tiles: 4 x 20 480 bytes fill half the CU's LDS exactly, so they
    // live in the last two padding floats of wavefront 0's row staging area (bytes 20 472..20 479 of its block), which
    // neither the staging (features 0..15 of a row) nor the first form's queue (at most 19 968 bytes) ever touches;
    // cleared here, before any wavefront of the group can arrive
    static_assert(sizeof(WaveLds) == sizeof(F2Wave), "the union is sized by the second form");
#ifndef DMET_F2_LEAN
    static_assert(sizeof(FilterQueue<filter_queue_len(filter_list_len(KP))>) <= sizeof(F2Wave) - 8, "ticket bytes are free");
#endif
    static_assert(kWavesPerGroup / 2 <= 2, "two ticket words");
    int *tickets = reinterpret_cast<int *>(&sh_all[0].f2.rows[kF2StageRows - 1][kF2RowF + 2]);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < kWavesPerGroup / 2) tickets[threadIdx.x] = 0;
    __syncthreads();
    const int group = filter_group(a);
    filter2_wave<KP, NH>(a, sh_all[wv].f2, tickets, group, wv, lane);   // returns at once unless the item's event is a second-form event
#ifndef DMET_F2_LEAN
    if constexpr (NH == 1) filter1_wave<KP>(a, sh_all[wv].f1, group, wv, lane);   // likewise (32 features only)
#endif
}

// Exact R1 chain for the kept candidates of one query, top-k by (d, j), certification.  M lanes per query (one kept
// candidate each; split tiles take a second round), 64 / M queries per wavefront; workgroups of one XCD walk one
// contiguous range of queries so the candidate rows they gather stay in that XCD's L2.
template <int KP>
__global__ __launch_bounds__(256) void knn_rerank_kernel(const KnnFilterArgs a)
{
    constexpr int M = filter_list_len(KP);
    constexpr int MS = (M + 1 + 3) & ~3;
    constexpr int QPW = kWave / M;                       // queries per wavefront (3 for M = 20)
    constexpr int QPB = 4 * QPW;                         // per workgroup
    constexpr int EMAX = kFilterMaxSplit * M;            // entries per query at most
    __shared__ float sc[QPB][EMAX];
    __shared__ int32_t sj[QPB][EMAX];
    __shared__ float skth[QPB];
    __shared__ int sfail[QPB];
    __shared__ float qbuf[QPB][32];
    constexpr int PARTS = (kFQ + QPB - 1) / QPB;         // workgroups per 64-query filter tile
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qw = lane / M, l = lane - qw * M;          // query slot of the wavefront, lane within the query
    const int slot = wv * QPW + min(qw, QPW - 1);
    // workgroup -> (filter tile, part): the event lookup is per workgroup (wave-uniform: scalar loads), not per lane
    // no XCD remap here: the grid is a worst-case bound and the live tiles are its first few hundred workgroups,
    // which the round-robin dispatch already spreads over all XCDs
    const int bid = blockIdx.x;
    if (a.form2 && a.plan->form1_events == 0) return;
    const int n_full = a.plan->n_full, split = a.plan->split;
    const int ft = n_full + bid / PARTS, part = bid % PARTS;   // only the split (tail) tiles come here
    if (ft >= a.plan->total_tiles) return;
    const int pos = find_tile_event(a.tile_ptr, a.B, ft);
    const int ev = a.order[pos];
    const int64_t ev_lo = a.ptr[ev], ev_hi = a.ptr[ev + 1];
    if (a.form2 && f2_in_domain((int)(ev_hi - ev_lo))) return;   // merged inside the filter kernel
    const int qoff = part * QPB + wv * QPW + qw;         // query within the tile
    const int64_t q = ev_lo + (int64_t)(ft - a.tile_ptr[pos]) * kFQ + qoff;
    const bool active = qw < QPW && qoff < kFQ && q < ev_hi;
    const int64_t qq = active ? q : ev_lo;
    const int nsub = split;
    const int64_t fslot = (int64_t)(ft - n_full) * kFQ + (active ? qoff : 0);
    const float *bd = a.psd + fslot * nsub * MS;
    const int32_t *bj = a.psj + fslot * nsub * MS;
    const int E = nsub * M;

    // issue every independent load up front: the kernel is bound by its chain of dependent memory round trips
    int32_t myj[kFilterMaxSplit];
#pragma unroll
    for (int t = 0; t < kFilterMaxSplit; ++t) myj[t] = (active && t < nsub) ? bj[t * MS + l] : -1;
    float vtau = kKnnSentinel, vnx = 0.0f;
    bool voverflow = false;
    if (active && l < nsub) { vtau = bd[l * MS + M]; voverflow = bj[l * MS + M] != 0; vnx = a.nrm[qq]; }
    // the wavefront's query rows go through LDS (read back as broadcasts): 32 fewer VGPRs, twice the resident waves
    if (lane < QPW * 8) {
        const int w = lane >> 3;
        const int64_t qrow_id = min(ev_lo + (int64_t)(ft - a.tile_ptr[pos]) * kFQ + part * QPB + wv * QPW + w, ev_hi - 1);
        *reinterpret_cast<float4 *>(&qbuf[wv * QPW + w][4 * (lane & 7)]) =
            reinterpret_cast<const float4 *>(a.x + qrow_id * 32)[lane & 7];
    }
    float myc[kFilterMaxSplit];
    wave_sync();
#pragma unroll 1
    for (int t = 0; t < kFilterMaxSplit; ++t) {
        float ct = kKnnSentinel;
        int32_t j = (t == 0) ? myj[0] : myj[kFilterMaxSplit - 1];
        if (active && t < nsub) {
            const int idx = t * M + l;
            if (j >= 0) {
                const float4 *g = reinterpret_cast<const float4 *>(a.x + (int64_t)j * 32);
                float4 crow[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) crow[c] = g[c];
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4 v = crow[c];
                    const float4 qv = *reinterpret_cast<const float4 *>(&qbuf[slot][4 * c]);
                    float df;
                    df = v.x - qv.x; acc = __builtin_fmaf(df, df, acc);
                    df = v.y - qv.y; acc = __builtin_fmaf(df, df, acc);
                    df = v.z - qv.z; acc = __builtin_fmaf(df, df, acc);
                    df = v.w - qv.w; acc = __builtin_fmaf(df, df, acc);
                }
                ct = acc;
                // a candidate at or beyond the sentinel distance (or NaN) is never a neighbour (dmet_oracle.c:62: strict
                // '>' against the 1e10 the lists start from) -- it counts as a missing entry from here on
                if (!(ct < kKnnSentinel)) {
                    ct = kKnnSentinel;
                    j = -1;
                    if (t == 0) myj[0] = -1; else myj[kFilterMaxSplit - 1] = -1;
                }
            }
            sc[slot][idx] = ct;
            sj[slot][idx] = (j >= 0) ? j : (0x7fffffff - idx);   // missing entries sort last, all distinct
        }
        if (t == 0) myc[0] = ct; else myc[kFilterMaxSplit - 1] = ct;
    }
    if (active && l == 0) { skth[slot] = -1.0f; sfail[slot] = 0; }
    wave_sync();
    const int k = a.k;
#pragma unroll
    for (int t = 0; t < kFilterMaxSplit; ++t) {
        if (active && t < nsub) {
            const int idx = t * M + l;
            const float c = myc[t];
            const int32_t jj = (myj[t] >= 0) ? myj[t] : (0x7fffffff - idx);
            int rank = 0;
            for (int e = 0; e < E; ++e) {
                const float ce = sc[slot][e];
                const int32_t je = sj[slot][e];
                rank += (ce < c || (ce == c && je < jj)) ? 1 : 0;
            }
            if (rank < k) {
                a.nbr[q * k + rank] = myj[t];
                if (a.nbr16) a.nbr16[q * k + rank] = local_id16(myj[t], ev_lo);
                a.dist[q * k + rank] = (myj[t] >= 0) ? c : kKnnSentinel;
                if (rank == k - 1 && myj[t] >= 0) skth[slot] = c;
            }
        }
    }
    wave_sync();
    // certification (one lane per partial list): a list that saw at least M keys dropped only keys >= its threshold
    if (active && l < nsub) {
        const float kth = skth[slot];
        const float an = __builtin_sqrtf(vnx) * 1.000001f;
        const float rn = an + __builtin_sqrtf(fmaxf(kth, 0.0f)) * 1.00002f;
        const float slack = 2.0f * (4e-5f * an * rn + 1e-5f * rn * rn + 4e-6f * an * an) + 1e-30f;
        const bool full = vtau < kKnnSentinel;
        // kth < 0 (fewer than k kept candidates) cannot coincide with a full list (M >= k)
        if (voverflow || (full && !(vtau + vnx - slack > kth))) sfail[slot] = 1;
    }
    wave_sync();
    if (active && l == 0 && sfail[slot] != 0) {      // count the query once
        flag_query(a, (int)qq, a.xtile_ptr[pos] + (int)((qq - ev_lo) / a.xtile_queries));
    }
}

int num_simds()
{
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            cached = cus * 4;
        else
            cached = 1024;
    }
    return cached;
}

inline int padded_k(int k) { return k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64; }
constexpr int kMaxSimds = 4096;  // workspace bound for the split (tail) tiles: fewer than `simds` tiles

struct KnnWorkspace {
    KnnPlan *plan;
    int32_t *order;
    int32_t *pos_of;
    int32_t *tile_ptr;
    float *wsd;
    int32_t *wsj;
    float *psd;
    int32_t *psj;
    // matrix-core filter path
    KnnPlan *fplan;
    int32_t *forder;
    int32_t *fpos_of;
    int32_t *ftile_ptr;
    float *nrm;
    uint8_t *rec;         // candidate tile records
    int64_t nrec;
    int32_t *flags;       // flags[...], any[4] and qflag[N] are one zero-filled region
    int32_t *any;
    uint8_t *qflag;
    int32_t *qlist;       // [N] flagged queries (not cleared: `any` bounds it)
    size_t zero_bytes;
    size_t bytes;
};

inline int64_t exact_tiles_max(int64_t N, int B) { return (N + 63) / 64 + B + 1; }   // bound for 64- and 128-query tiles

inline KnnWorkspace carve_workspace(void *ws, int64_t N, int B, int KP)
{
    KnnWorkspace w;
    uintptr_t p = (reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u;
    auto take = [&](size_t nbytes) { uintptr_t r = p; p = (p + nbytes + 255u) & ~(uintptr_t)255u; return r; };
    // query slots of split (tail) tiles: fewer than kMaxSimds tiles, and never more than all tiles hold
    size_t split_q = (size_t)kMaxSimds * 128;
    if ((size_t)N + 128 * ((size_t)B + 1) < split_q) split_q = (size_t)N + 128 * ((size_t)B + 1);
    // the filter's split tiles use the same arrays: [slots][kFilterMaxSplit][KP + KP/4 + pad]
    size_t ps_elems = split_q * kMaxSplit * KP;
    {
        size_t fslots = (size_t)kMaxSimds * 2 * kFQ;
        if ((size_t)N + kFQ * ((size_t)B + 1) < fslots) fslots = (size_t)N + kFQ * ((size_t)B + 1);
        const size_t need = fslots * kFilterMaxSplit * (size_t)(2 * KP);
        if (need > ps_elems) ps_elems = need;
    }
    w.plan = reinterpret_cast<KnnPlan *>(take(sizeof(KnnPlan)));
    w.order = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ((size_t)B + 1)));
    w.pos_of = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ((size_t)B + 1)));
    w.tile_ptr = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ((size_t)B + 1)));
    w.wsd = reinterpret_cast<float *>(take(sizeof(float) * (size_t)N * KP));
    w.wsj = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)N * KP));
    w.psd = reinterpret_cast<float *>(take(sizeof(float) * ps_elems));
    w.psj = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ps_elems));
    w.fplan = reinterpret_cast<KnnPlan *>(take(sizeof(KnnPlan)));
    w.forder = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ((size_t)B + 1)));
    w.fpos_of = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ((size_t)B + 1)));
    w.ftile_ptr = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * ((size_t)B + 1)));
    w.nrm = reinterpret_cast<float *>(take(sizeof(float) * (size_t)N));
    w.nrec = (N >> 5) + B + 1;                               // event b owns records from (ptr[b] >> 5) + b
    w.rec = reinterpret_cast<uint8_t *>(take((size_t)w.nrec * kRecBytesMax));
    w.zero_bytes = sizeof(int32_t) * ((size_t)exact_tiles_max(N, B) + 4) + (size_t)N;
    w.flags = reinterpret_cast<int32_t *>(take(w.zero_bytes));
    w.any = w.flags + exact_tiles_max(N, B);
    w.qflag = reinterpret_cast<uint8_t *>(w.any + 4);
    w.qlist = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)N));
    w.bytes = (size_t)(p - reinterpret_cast<uintptr_t>(ws));
    return w;
}

// DMET_KNN_PATH=exact disables the matrix-core filter (tests and A/B timing); =filter_only (experiment builds only)
// skips the exact fallback of uncertified tiles; anything else: filter when eligible + fallback
inline int filter_mode()
{
    const char *e = getenv("DMET_KNN_PATH");   // read per call: tests switch paths inside one process
    if (e && strcmp(e, "exact") == 0) return 0;
#if defined(DMET_KNN_EXPERIMENT) || defined(DMET_FILTER_ABL) || defined(DMET_F2_ABL) || defined(DMET_F2_SAMEREC)
    // timing builds only (tools/knn_budget*.sh): skipping the exact fallback can return uncertified neighbours, so
    // the product library does not honour it
    if (e && strcmp(e, "filter_only") == 0) return 2;
#endif
    return 1;
}

// DMET_KNN_FILTER=1: first form (per-key queue) for every event (A/B timing, tests); default: second form where it applies
inline int filter_form2()
{
    const char *e = getenv("DMET_KNN_FILTER");
    return (e && strcmp(e, "1") == 0) ? 0 : 1;
}

// prep + filter (+ in-place re-rank) + re-rank of the split tail tiles, for result capacity KF >= k
template <int KF, int NH = 1>
int launch_filter(const KnnFilterArgs &f, const KnnWorkspace &w, int simds, const KnnPlanOut &px, const KnnPlanOut &pf,
                  hipStream_t st)
{
    const int slots = simds * kF2WavesPerSimd;   // filter wavefronts per SIMD
    bool affine = false;
    if constexpr (NH == 1) affine = g_affine.raw != nullptr && !g_affine.done;
    if (affine) {
        if constexpr (NH == 1)
            hipLaunchKernelGGL((knn_prep_kernel<1, true>), dim3((unsigned)((w.nrec * kWave + 255) / 256 + 2)), dim3(256), 0, st, f.x,
                               f.ptr, f.B, f.N, w.nrm, w.rec, w.nrec, reinterpret_cast<uint32_t *>(w.flags), w.zero_bytes, px,
                               pf, f.form2, g_affine);
        g_affine.done = true;
    } else {
        hipLaunchKernelGGL((knn_prep_kernel<NH>), dim3((unsigned)((w.nrec * kWave + 255) / 256 + 2)), dim3(256), 0, st, f.x,
                           f.ptr, f.B, f.N, w.nrm, w.rec, w.nrec, reinterpret_cast<uint32_t *>(w.flags), w.zero_bytes, px, pf,
                           f.form2, KnnAffine{});
    }
    DMET_LAUNCH_CHECK("knn_prep_kernel");
    const int64_t ftiles_max = (f.N + kFQ - 1) / kFQ + f.B;
    const int64_t fblocks = (ftiles_max + slots + kWavesPerGroup - 1) / kWavesPerGroup;
    if (f.form2) {
        KnnFilterArgs fr = f;
        int64_t grid = fblocks;
        if (NH == 1 && g_rider.P != nullptr && !g_rider.done) {
            fr.rW = g_rider.W; fr.rb = g_rider.b; fr.rP = g_rider.P; fr.rQ = g_rider.Q; fr.r_sliced = g_rider.sliced;
            fr.first_rider = (int)fblocks;
            grid = fblocks + rider_groups();
            g_rider.done = true;
        }
        hipLaunchKernelGGL((knn_filter12_kernel<KF, NH>), dim3((unsigned)grid), dim3(kWave * kWavesPerGroup), 0, st, fr);
        DMET_LAUNCH_CHECK("knn_filter12_kernel");
    } else {
        if constexpr (NH == 1) hipLaunchKernelGGL((knn_filter_kernel<KF>), dim3((unsigned)fblocks), dim3(kWave * kWavesPerGroup), 0, st, f);
        DMET_LAUNCH_CHECK("knn_filter_kernel");
    }
    constexpr int kRerankQpb = 4 * (kWave / filter_list_len(KF));
    constexpr int kRerankParts = (kFQ + kRerankQpb - 1) / kRerankQpb;
    // only the split tail tiles (fewer than `slots`) need the separate re-rank: whole sweeps re-rank in place
    const int64_t tail_max = ftiles_max < slots ? ftiles_max : slots;
    if constexpr (NH == 1) {    // tail tiles of the first form (32 features only)
        if (f.no_rerank) return 0;     // no first-form event in this batch (size hint): nothing to merge
        hipLaunchKernelGGL((knn_rerank_kernel<KF>), dim3((unsigned)(tail_max * kRerankParts)), dim3(256), 0, st, f);
        DMET_LAUNCH_CHECK("knn_rerank_kernel");
    }
    return 0;
}

template <int DP, int KP>
int launch_knn(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr, float *dist,
               uint16_t *nbr16, const KnnWorkspace &w, hipStream_t st)
{
    constexpr int TQ = (DP <= 32) ? 2 : 1;
    constexpr int QT = kWave * TQ;
    int simds = num_simds();
    if (simds > kMaxSimds) simds = kMaxSimds;
    bool use_filter = false;
    if constexpr ((DP == 32 || DP == 64) && KP <= 32)
        use_filter = D == DP && k <= 20 && aligned16(x) && filter_mode() != 0 && (DP == 32 || filter_form2());
    constexpr int NH = DP == 64 ? 2 : 1;   // 64 features (the DRN's hidden width): second filter form only
    const int slots = simds * kF2WavesPerSimd;   // filter wavefronts per SIMD
    const KnnPlanOut px{QT, simds, kMaxSplit, w.order, w.pos_of, w.tile_ptr, w.plan};
    const KnnPlanOut pf{kFQ, slots, kFilterMaxSplit, w.forder, w.fpos_of, w.ftile_ptr, w.fplan};
    if (!use_filter) {   // the filter path computes both plans inside its prep launch
        hipLaunchKernelGGL(knn_plan_kernel, dim3(1), dim3(256), 0, st, ptr, B, px, pf);
        DMET_LAUNCH_CHECK("knn_plan_kernel");
    }
    KnnArgs a{x, ptr, B, N, D, k, nbr, dist, nbr16, w.wsd, w.wsj, w.plan, w.order, w.tile_ptr, w.psd, w.psj, nullptr, 0, nullptr,
              nullptr, nullptr, 0, QT};
    // uncertified-query counters: zero for every call, so dmet_knn_fallback_stats is meaningful on any path (the
    // matrix-core path clears them in its prep kernel)
    if (!use_filter && hipMemsetAsync(w.flags, 0, w.zero_bytes, st) != hipSuccess)
        return hip_fail(hipGetLastError(), "hipMemsetAsync");

    // matrix-core filter + exact re-rank for the hot shapes (D = 32 or 64, k <= 20); the exact kernel then only recomputes
    // the tiles the re-rank could not certify
    if (use_filter) {
        KnnFilterArgs f{x, ptr, B, N, k, w.nrm, w.rec, w.fplan, w.forder, w.fpos_of, w.ftile_ptr,
                        w.psd, w.psj, nbr, dist, nbr16, w.flags, w.any, w.qflag, w.qlist, w.tile_ptr, QT, filter_form2(),
                        NH == 1 ? 1.0f : 1.5f, nullptr, nullptr, nullptr, nullptr, 0, 0,
                        (aligned16(nbr) && aligned16(dist) && aligned16(nbr16) && !env_is("DMET_KNN_EMIT", "lanes")) ? 1 : 0, 0};
        // every event is a second-form event (the caller says so): the first form's tail merge has nothing to do
        f.no_rerank = (f.form2 && g_size_hint.min_nodes >= kF2MinNodes && g_size_hint.max_nodes >= g_size_hint.min_nodes &&
                       g_size_hint.max_nodes <= kF2MaxNodes) ? 1 : 0;
        int rc = 0;
        if constexpr (DP == 32 || DP == 64) {
            if constexpr (KP == 8) rc = launch_filter<8, NH>(f, w, simds, px, pf, st);
            else if constexpr (KP == 16) rc = launch_filter<16, NH>(f, w, simds, px, pf, st);
            else if constexpr (KP == 32) rc = launch_filter<20, NH>(f, w, simds, px, pf, st);   // 16 < k <= 20 (checked above)
        }
        if (rc) return rc;
        if (filter_mode() == 2) return 0;
        a.flags = w.flags;
        a.any = w.any;
        if constexpr (NH == 1) {
            a.qlist = w.qlist;             // the per-query fallback rides in front of the exact kernel's grid
            a.pos_of = w.pos_of;
            a.requery_groups = kRequeryGroups;
            a.flag_min = kRequeryMax + 1;
        } else {
            a.flag_min = 1;    // no per-query fallback at 64 features: the exact kernel takes every flagged tile
        }
    }

    // worst-case grid (the plan is on the device): every event adds at most one partial tile, and splitting the
    // fewer-than-`simds` tail tiles adds fewer than `simds` workgroups; surplus workgroups exit at once
    const int64_t tiles_max = (N + QT - 1) / QT + B;
    const int64_t blocks = (tiles_max + simds + kWavesPerGroup - 1) / kWavesPerGroup + a.requery_groups;
    unsigned dyn = 0;
#ifdef DMET_KNN_EXPERIMENT
    if (const char *e = getenv("DMET_KNN_EXTRA_LDS")) dyn = (unsigned)atoi(e);
#endif
    if (D == DP && aligned16(x))
        hipLaunchKernelGGL((knn_kernel<DP, KP, TQ, true>), dim3((unsigned)blocks), dim3(kWave * kWavesPerGroup), dyn, st, a);
    else
        hipLaunchKernelGGL((knn_kernel<DP, KP, TQ, false>), dim3((unsigned)blocks), dim3(kWave * kWavesPerGroup), dyn, st, a);
    DMET_LAUNCH_CHECK("knn_kernel");
    if (!use_filter) {   // split tiles (the tail of a large batch, every tile of a small one) merge their partial lists
        int64_t slots = (int64_t)simds * QT;
        if (tiles_max * QT < slots) slots = tiles_max * QT;
        hipLaunchKernelGGL((knn_merge_kernel<KP>), dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, a, QT);
        DMET_LAUNCH_CHECK("knn_merge_kernel");
    }
    return 0;
}

template <int DP>
int dispatch_k(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr, float *dist,
               uint16_t *nbr16, void *ws, hipStream_t st)
{
    const int KP = padded_k(k);
    const KnnWorkspace w = carve_workspace(ws, N, B, KP);
    if (k <= 8) return launch_knn<DP, 8>(x, ptr, B, N, D, k, nbr, dist, nbr16, w, st);
    if (k <= 16) return launch_knn<DP, 16>(x, ptr, B, N, D, k, nbr, dist, nbr16, w, st);
    if (k <= 32) return launch_knn<DP, 32>(x, ptr, B, N, D, k, nbr, dist, nbr16, w, st);
    return launch_knn<DP, 64>(x, ptr, B, N, D, k, nbr, dist, nbr16, w, st);
}

// ---- radius graph (N1): first max_nbr candidates in ascending index with d < r^2 ------------------------
// One lane per query, four independent wavefronts per workgroup (no workgroup barrier).  Candidates are staged per
// wavefront in LDS as [pair][feature][2] so that one broadcast read yields a feature of two candidates in adjacent
// registers: the R1 chain then runs on v_pk_add_f32 / v_pk_fma_f32 (each half an exact IEEE op in feature order, same
// bits as the scalar oracle).  The kernel only writes the hits; unused slots are -1 from a memset (dmet_radius_f32) or left unwritten (counted forms).
// `skip_self` reproduces upstream's loop=False: the search limit counts the node itself, the node is not stored.
constexpr int kRadTile = 64;   // candidates per LDS tile and wavefront

template <int DP>
__global__ __launch_bounds__(kWave * 4) void radius_kernel(const float *__restrict__ x,
                                                            const int64_t *__restrict__ ptr, int B, int64_t N, int D,
                                                            float r2, int max_nbr, int skip_self,
                                                            int32_t *__restrict__ nbr, int32_t *__restrict__ cntout)
{
    __shared__ f2 tile_all[4][(kRadTile / 2) * DP];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f2 *tile = tile_all[wv];
    const int64_t q_first = ((int64_t)blockIdx.x * 4 + wv) * kWave;
    if (q_first >= N) return;
    const int64_t q_last = min(N, q_first + kWave) - 1;
    const int b_first = find_event(ptr, B, q_first);
    const int b_last = find_event(ptr, B, q_last);
    const int clo = (int)ptr[b_first];
    const int chi = (int)ptr[b_last + 1];
    const bool one_event = b_first == b_last;          // wave-uniform: no per-lane event window needed
    const int64_t qi = q_first + lane;
    const bool valid = qi < N;
    const int64_t qq = valid ? qi : q_last;
    int lo = clo, hi = chi;
    if (!one_event) { const int b = find_event(ptr, B, qq); lo = (int)ptr[b]; hi = (int)ptr[b + 1]; }
    f2 q[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) { const float v = (c < D) ? x[qq * D + c] : 0.0f; q[c].x = v; q[c].y = v; }
    int stored = 0, seen = valid ? 0 : max_nbr;        // idle lanes are "full" from the start
    int32_t *row = nbr + qq * max_nbr;
    // hits are rare per lane (a few per thousand pairs): the sweep of a 64-candidate tile only records them as bits
    // of a lane-private 64-bit mask (compare + select + or per candidate, no branch, no memory traffic); the set
    // bits are turned into row entries after the tile, in ascending candidate order
    for (int c0 = clo; c0 < chi; c0 += kRadTile) {
        const int cntc = min(kRadTile, chi - c0);
        wave_sync();
        for (int e = lane; e < kRadTile * DP; e += kWave) {
            const int c = e / DP, dd = e - c * DP;
            // rows past the range get a coordinate that is farther than any radius from everything
            const float v = (c < cntc) ? ((dd < D) ? x[(int64_t)(c0 + c) * D + dd] : 0.0f) : 3.0e18f;
            reinterpret_cast<float *>(tile)[((c >> 1) * DP + dd) * 2 + (c & 1)] = v;
        }
        wave_sync();
        if (!__any(seen < max_nbr)) break;             // every query of the wavefront is full
        unsigned m0 = 0u, m1 = 0u;
#pragma unroll
        for (int cc = 0; cc < kRadTile; cc += 2) {
            f2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < DP; ++c) {
                const f2 df = tile[(cc >> 1) * DP + c] - q[c];
                acc = __builtin_elementwise_fma(df, df, acc);
            }
            if (cc < 32) {
                m0 |= (acc.x < r2) ? (1u << cc) : 0u;
                m0 |= (acc.y < r2) ? (1u << (cc + 1)) : 0u;
            } else {
                m1 |= (acc.x < r2) ? (1u << (cc - 32)) : 0u;
                m1 |= (acc.y < r2) ? (1u << (cc - 31)) : 0u;
            }
        }
        unsigned long long mask = ((unsigned long long)m1 << 32) | m0;
        if (!one_event) {                               // keep the candidates of the lane's own event only
            const int a0 = max(lo - c0, 0), a1 = min(hi - c0, 64);
            const unsigned long long keep = (a1 <= a0) ? 0ull
                : ((a1 >= 64 ? ~0ull : ((1ull << a1) - 1ull)) & ~((1ull << a0) - 1ull));
            mask &= keep;
        }
        while (__any(mask != 0ull)) {
            if (mask != 0ull) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const int j = c0 + bit;
                if (seen < max_nbr) {
                    if (!(skip_self && j == (int)qq)) { row[stored] = j; ++stored; }
                    ++seen;
                }
            }
        }
    }
    if (valid) cntout[qi] = stored;
}


// ---- radius graph, windowed by the first coordinate --------------------------------------------------------------
// d(i,j) < r^2 needs |x0_i - x0_j| < r.  Queries are therefore PROCESSED in the order of their first coordinate (a
// wavefront = 64 neighbours in x0), and each wavefront walks its event's nodes in index order but only keeps those
// whose x0 lies in [min x0 - r, max x0 + r] of its queries (stream compaction: ballot + prefix count into a small
// index queue).  The kept candidates go through the same tile sweep as radius_kernel, still in ascending index
// order, so "the first max_nbr hits in index order" and the bits of every distance are unchanged -- only the
// candidates that cannot be hits are never multiplied out.  In (eta, phi) with r = 0.4 that is ~85 % of them.
constexpr int kRadBins = 1024;

// order[ptr[b] .. ptr[b+1]) = the node ids of event b grouped into kRadBins bins of x0 (ascending bins; the order
// inside a bin is arbitrary and does not influence any result, only which queries share a wavefront).
__global__ __launch_bounds__(kRadBins) void radius_order_kernel(const float *__restrict__ x,
                                                                const int64_t *__restrict__ ptr, int B, int D,
                                                                int32_t *__restrict__ order)
{
    constexpr int NT = kRadBins, NW = kRadBins / 64;
    __shared__ int hist[kRadBins];
    __shared__ float red_lo[NW], red_hi[NW];
    __shared__ int wave_tot[NW];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int64_t lo = ptr[b], hi = ptr[b + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int64_t i = lo + tid; i < hi; i += NT) {
        const float v = x[i * D];
        if (v == v && fabsf(v) < 3.0e38f) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
    if (lane == 0) { red_lo[wv] = mn; red_hi[wv] = mx; }
    hist[tid] = 0;
    __syncthreads();
    mn = red_lo[0]; mx = red_hi[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) { mn = fminf(mn, red_lo[w]); mx = fmaxf(mx, red_hi[w]); }
    const float scale = (mx > mn) ? (float)(kRadBins - 1) / (mx - mn) : 0.0f;
    auto bin_of = [&](float v) -> int {
        if (!(v == v)) return 0;
        const float t = (v - mn) * scale;
        if (!(t == t)) return 0;
        return t <= 0.0f ? 0 : (t >= (float)(kRadBins - 1) ? kRadBins - 1 : (int)t);
    };
    for (int64_t i = lo + tid; i < hi; i += NT) atomicAdd(&hist[bin_of(x[i * D])], 1);
    __syncthreads();
    // exclusive scan of the counters: one per thread, wavefront scan, then the wavefront totals
    const int mine = hist[tid];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wv; ++w) base += wave_tot[w];
    hist[tid] = base;
    __syncthreads();
    for (int64_t i = lo + tid; i < hi; i += NT) {
        const int pos = atomicAdd(&hist[bin_of(x[i * D])], 1);
        order[lo + pos] = (int32_t)i;
    }
}

constexpr int kRadQueue = 128;   // pending candidate ids per wavefront (a compaction step adds at most 64)

template <int DP>
__global__ __launch_bounds__(kWave * 4) void radius_window_kernel(const float *__restrict__ x,
                                                                   const int64_t *__restrict__ ptr, int B, int64_t N,
                                                                   int D, float r2, int max_nbr, int skip_self,
                                                                   const int32_t *__restrict__ order,
                                                                   int32_t *__restrict__ nbr,
                                                                   int32_t *__restrict__ cntout,
                                                                   uint16_t *__restrict__ nbr16, int stride16)
{
    __shared__ f2 tile_all[4][(kRadTile / 2) * DP];
    __shared__ int queue_all[4][kRadQueue];
    // hits leave the lane through 16-byte staging slots (4 int32 ids / 8 uint16 ids) and reach memory as one 16-byte
    // store per full slot: a wavefront's 64 scattered 2- or 4-byte stores per hit cost more than the distances
    __shared__ __attribute__((aligned(16))) int32_t stage32_all[4][kWave][4];
    __shared__ __attribute__((aligned(16))) uint16_t stage16_all[4][kWave][8];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f2 *tile = tile_all[wv];
    int *queue = queue_all[wv];
    int32_t *stage32 = stage32_all[wv][lane];
    uint16_t *stage16 = stage16_all[wv][lane];
    // wavefronts are aligned to events: event b owns the wavefront ids from (ptr[b] >> 6) + b on (strictly increasing
    // in b and at least ceil(n_b / 64) apart, so no prefix sum over the events is needed); surplus ids idle
    const int64_t w = (int64_t)blockIdx.x * 4 + wv;
    int b = 0;
    {
        int l = 0, h = B;                               // largest b with (ptr[b] >> 6) + b <= w
        while (h - l > 1) {
            const int mid = (l + h) >> 1;
            if ((ptr[mid] >> 6) + mid <= w) l = mid; else h = mid;
        }
        b = l;
    }
    const int clo = (int)ptr[b], chi = (int)ptr[b + 1];
    const int64_t p_first = clo + (w - ((int64_t)(clo >> 6) + b)) * kWave;   // positions in `order`
    if (p_first >= chi) return;
    const int64_t p_last = min((int64_t)chi, p_first + kWave) - 1;
    constexpr bool one_event = true;
    const int64_t pi = p_first + lane;
    const bool valid = pi < chi;
    const int64_t qq = order[valid ? pi : p_last];     // this lane's query (a node of event b)
    const int lo = clo, hi = chi;
    (void)N;
    f2 q[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) { const float v = (c < D) ? x[qq * D + c] : 0.0f; q[c].x = v; q[c].y = v; }
    // window of the first coordinate: a little wider than [min - r, max + r] so that rounding can only ADD candidates
    float wlo = -__builtin_inff(), whi = __builtin_inff();
    if (one_event) {
        float mn = q[0].x, mx = q[0].x;
        if (!(mn == mn)) { mn = __builtin_inff(); mx = -__builtin_inff(); }   // a NaN query has no hits anyway
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
        const float rr = __builtin_sqrtf(r2) * 1.00001f + 1e-30f;
        wlo = mn - rr - 1e-6f * fabsf(mn);
        whi = mx + rr + 1e-6f * fabsf(mx);
    }
    int stored = 0, seen = valid ? 0 : max_nbr;        // idle lanes are "full" from the start
    int32_t *row = nbr + qq * max_nbr;
    // optional second copy of the row as event-local uint16 ids (rows of stride16 ids, 16-byte aligned)
    uint16_t *row16 = nbr16 ? nbr16 + qq * stride16 : nullptr;
    int pending = 0;                                    // wave-uniform: ids waiting in the queue

    // sweep of one tile: the first `cntc` queue entries (ascending node ids)
    auto sweep = [&](const int cntc) {
        wave_sync();
        for (int e = lane; e < kRadTile * DP; e += kWave) {
            const int c = e / DP, dd = e - c * DP;
            // rows past the range get a coordinate that is farther than any radius from everything
            const float v = (c < cntc) ? ((dd < D) ? x[(int64_t)queue[c] * D + dd] : 0.0f) : 3.0e18f;
            reinterpret_cast<float *>(tile)[((c >> 1) * DP + dd) * 2 + (c & 1)] = v;
        }
        wave_sync();
        unsigned m0 = 0u, m1 = 0u;
#pragma unroll
        for (int cc = 0; cc < kRadTile; cc += 2) {
            f2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < DP; ++c) {
                const f2 df = tile[(cc >> 1) * DP + c] - q[c];
                acc = __builtin_elementwise_fma(df, df, acc);
            }
            if (cc < 32) {
                m0 |= (acc.x < r2) ? (1u << cc) : 0u;
                m0 |= (acc.y < r2) ? (1u << (cc + 1)) : 0u;
            } else {
                m1 |= (acc.x < r2) ? (1u << (cc - 32)) : 0u;
                m1 |= (acc.y < r2) ? (1u << (cc - 31)) : 0u;
            }
        }
        unsigned long long mask = ((unsigned long long)m1 << 32) | m0;
        while (__any(mask != 0ull)) {
            if (mask != 0ull) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const int j = queue[bit];
                if (seen < max_nbr && j >= lo && j < hi) {   // own event only (wavefronts that straddle two events)
                    if (!(skip_self && j == (int)qq)) {
                        stage32[stored & 3] = j;
                        if (row16) stage16[stored & 7] = (uint16_t)(j - lo);
                        ++stored;
                        if ((stored & 3) == 0 && nbr) {   // rows are only 4-byte aligned (255-wide tables)
                            struct __attribute__((packed, aligned(4))) I4 { int32_t a, b, c, d; };
                            const int4 v = *reinterpret_cast<const int4 *>(stage32);
                            *reinterpret_cast<I4 *>(row + stored - 4) = I4{v.x, v.y, v.z, v.w};
                        }
                        if (row16 && (stored & 7) == 0)
                            *reinterpret_cast<uint4 *>(row16 + stored - 8) = *reinterpret_cast<const uint4 *>(stage16);
                    }
                    ++seen;
                }
            }
        }
    };

    // the walk over the event's nodes is a chain of dependent steps (load a first coordinate, ballot, append, maybe
    // sweep): with every wavefront of the grid resident at once the kernel lasts as long as ONE wavefront's chain, so
    // the first coordinates of kRadAhead groups of 64 nodes are fetched together (151 -> 128 us for the whole table at
    // 64 x 4500 nodes).  Carrying both coordinates with the queued ids (D = 2), so that a sweep needs no gather from
    // memory at all, was measured too: no further gain -- the kernel is then half vector-ALU time (7 000 instructions
    // per wavefront: the hit loop, the pair distances, the compaction), half latency.
    constexpr int kRadAhead = 4;
    bool full = false;
    for (int c0 = clo; c0 < chi && !full; c0 += kWave * kRadAhead) {
        float v4[kRadAhead];
#pragma unroll
        for (int u = 0; u < kRadAhead; ++u) {
            const int c = c0 + u * kWave + lane;
            v4[u] = (c < chi) ? x[(int64_t)c * D] : __builtin_nanf("");   // a NaN is outside every window
        }
#pragma unroll
        for (int u = 0; u < kRadAhead; ++u) {
            if (c0 + u * kWave >= chi) break;
            if (!__any(seen < max_nbr)) { pending = 0; full = true; break; }   // every query of the wavefront is full
            const int c = c0 + u * kWave + lane;
            const bool keep = c < chi && v4[u] >= wlo && v4[u] <= whi;
            const unsigned long long km = __ballot(keep);
            if (keep) queue[pending + __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u))] = c;
            pending += __popcll(km);
            if (pending >= kRadTile) {
                sweep(kRadTile);
                wave_sync();
                const int rest = pending - kRadTile;       // < 64: move it to the front
                int moved = 0;
                if (lane < rest) moved = queue[kRadTile + lane];
                wave_sync();
                if (lane < rest) queue[lane] = moved;
                pending = rest;
            }
        }
    }
    if (pending > 0 && __any(seen < max_nbr)) sweep(pending);
    if (valid) {
        cntout[qq] = stored;
        if (nbr)
            for (int s = stored & ~3; s < stored; ++s) row[s] = stage32[s & 3];     // the unfinished slots
        if (row16 && (stored & 7) != 0) {   // the last started chunk of 8 reads as "no neighbour" beyond the row's end
            for (int s = stored & 7; s < 8; ++s) stage16[s] = 0xFFFFu;
            *reinterpret_cast<uint4 *>(row16 + (stored & ~7)) = *reinterpret_cast<const uint4 *>(stage16);
        }
    }
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_knn_workspace_bytes(int64_t N, int B, int D, int k)
{
    (void)D;
    if (N <= 0 || B < 0 || k <= 0 || k > DMET_MAX_K) return 0;
    return carve_workspace(nullptr, N, B, padded_k(k)).bytes + 512;
}

extern "C" int dmet_knn_local_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                                  float *dist, uint16_t *nbr16, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    struct HintScope { ~HintScope() { g_size_hint = KnnSizeHint{}; } } hint_scope;   // the hint describes one batch: it is spent by one build, whatever the outcome
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647 - 4096, "dmet_knn_f32/dmet_knn_local_f32: N=%lld out of range", (long long)N);
    DMET_REQUIRE(B >= 0, "dmet_knn_f32/dmet_knn_local_f32: B=%d", B);
    DMET_REQUIRE(k >= 1 && k <= DMET_MAX_K, "dmet_knn_f32/dmet_knn_local_f32: k=%d not in [1,%d]", k, DMET_MAX_K);
    DMET_REQUIRE(D >= 1 && D <= DMET_MAX_KNN_DIM, "dmet_knn_f32/dmet_knn_local_f32: D=%d not in [1,%d]", D, DMET_MAX_KNN_DIM);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(x && ptr && nbr && dist && ws, "dmet_knn_f32/dmet_knn_local_f32: null pointer");
    DMET_REQUIRE(ws_bytes >= dmet_knn_workspace_bytes(N, B, D, k), "dmet_knn_f32/dmet_knn_local_f32: workspace too small");
    DMET_REQUIRE(!nbr16 || (reinterpret_cast<uintptr_t>(nbr16) & 3u) == 0, "dmet_knn_local_f32: nbr_local must be 4-byte aligned");
    hipStream_t st = as_stream(stream);
    int rc;
    if (D <= 4) rc = dispatch_k<4>(x, ptr, B, N, D, k, nbr, dist, nbr16, ws, st);
    else if (D <= 8) rc = dispatch_k<8>(x, ptr, B, N, D, k, nbr, dist, nbr16, ws, st);
    else if (D <= 16) rc = dispatch_k<16>(x, ptr, B, N, D, k, nbr, dist, nbr16, ws, st);
    else if (D <= 32) rc = dispatch_k<32>(x, ptr, B, N, D, k, nbr, dist, nbr16, ws, st);
    else rc = dispatch_k<64>(x, ptr, B, N, D, k, nbr, dist, nbr16, ws, st);
    return rc;
}

extern "C" int dmet_knn_size_hint(int min_nodes, int max_nodes)
{
    DMET_REQUIRE(min_nodes >= 0 && max_nodes >= 0, "dmet_knn_size_hint: negative size");
    g_size_hint.min_nodes = min_nodes;
    g_size_hint.max_nodes = max_nodes;
    return 0;
}

extern "C" int dmet_knn_local_dense_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                                        float *dist, uint16_t *nbr16, const float *W, const float *bias, int layout,
                                        float *P, void *Q, int *dense_done, void *ws, size_t ws_bytes,
                                        dmet_stream_t stream)
{
    DMET_REQUIRE(dense_done, "dmet_knn_local_dense_f32: dense_done is null");
    *dense_done = 0;
    DMET_REQUIRE(W && P && Q, "dmet_knn_local_dense_f32: null pointer");
    DMET_REQUIRE(layout >= 0 && layout <= 2, "dmet_knn_local_dense_f32: layout=%d not in {0, 1, 2}", layout);
    const int sliced = layout;
    DMET_REQUIRE(aligned16(P) && aligned16(Q), "dmet_knn_local_dense_f32: P / Q must be 16-byte aligned");
    // the dense layer rides in the matrix-core filter launch (32 features); any other build leaves it to the caller
    g_rider = KnnRider{};
    if (D == 32 && N > 0 && B > 0 && aligned16(x)) {
        g_rider.W = W; g_rider.b = bias; g_rider.P = P; g_rider.Q = reinterpret_cast<float *>(Q); g_rider.sliced = sliced;
    }
    const int rc = dmet_knn_local_f32(x, ptr, B, N, D, k, nbr, dist, nbr16, ws, ws_bytes, stream);
    *dense_done = (rc == 0 && g_rider.done) ? 1 : 0;
    g_rider = KnnRider{};
    return rc;
}

extern "C" int dmet_bn_knn_local_dense_f32(const float *raw, const float *residual, const float *gamma, const float *beta,
                                           const float *mean, const float *invstd, float *y, const int64_t *ptr, int B,
                                           int64_t N, int D, int k, int32_t *nbr, float *dist, uint16_t *nbr16,
                                           const float *W, const float *bias, int layout, float *P, void *Q,
                                           int *dense_done, int *fused, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(fused && dense_done, "dmet_bn_knn_local_dense_f32: fused / dense_done is null");
    *fused = 0;
    *dense_done = 0;
    DMET_REQUIRE(raw && gamma && beta && mean && invstd && y, "dmet_bn_knn_local_dense_f32: null pointer");
    // only the matrix-core path has the prep launch the transform rides in: any other build leaves everything to the caller
    const bool eligible = D == 32 && k >= 1 && k <= 20 && N > 0 && B > 0 && filter_mode() != 0 && aligned16(raw) && aligned16(y) &&
                          aligned16(gamma) && aligned16(beta) && aligned16(mean) && aligned16(invstd) &&
                          (!residual || aligned16(residual));
    if (!eligible) return 0;
    g_affine = KnnAffine{};
    g_affine.raw = raw; g_affine.res = residual; g_affine.gamma = gamma; g_affine.beta = beta; g_affine.mean = mean;
    g_affine.invstd = invstd;
    int rc;
    if (W) rc = dmet_knn_local_dense_f32(y, ptr, B, N, D, k, nbr, dist, nbr16, W, bias, layout, P, Q, dense_done, ws, ws_bytes, stream);
    else rc = dmet_knn_local_f32(y, ptr, B, N, D, k, nbr, dist, nbr16, ws, ws_bytes, stream);
    const bool done = g_affine.done;
    g_affine = KnnAffine{};
    if (rc == 0 && !done) {
        set_error("dmet_bn_knn_local_dense_f32: the build did not take the matrix-core path it was checked for");
        return -22;
    }
    *fused = (rc == 0) ? 1 : 0;
    return rc;
}

extern "C" int dmet_knn_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                            float *dist, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    return dmet_knn_local_f32(x, ptr, B, N, D, k, nbr, dist, nullptr, ws, ws_bytes, stream);
}

extern "C" int dmet_knn_fallback_stats(const void *ws, int64_t N, int B, int D, int k, int64_t *out, dmet_stream_t stream)
{
    (void)D;
    DMET_REQUIRE(N > 0 && B > 0 && k >= 1 && k <= DMET_MAX_K && ws && out, "dmet_knn_fallback_stats: bad arguments");
    const KnnWorkspace w = carve_workspace(const_cast<void *>(ws), N, B, padded_k(k));
    const int64_t n = exact_tiles_max(N, B);
    int32_t *host = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)n));
    DMET_REQUIRE(host != nullptr, "dmet_knn_fallback_stats: out of host memory");
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemcpyAsync(host, w.flags, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(host); return hip_fail(e, "dmet_knn_fallback_stats"); }
    out[0] = 0; out[1] = 0;
    for (int64_t i = 0; i < n; ++i) { out[0] += host[i] != 0; out[1] += host[i]; }
    free(host);
    if (getenv("DMET_KNN_LIST_FLAGGED") && out[1] > 0) {   // developer aid: the first flagged queries, to stderr
        int32_t ids[16];
        const int64_t m = out[1] < 16 ? out[1] : 16;
        if (hipMemcpy(ids, w.qlist, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost) == hipSuccess)
            for (int64_t i = 0; i < m; ++i) {
                uint8_t why = 0;
                (void)hipMemcpy(&why, w.qflag + ids[i], 1, hipMemcpyDeviceToHost);
                fprintf(stderr, "[dmet] flagged query %d (qflag %d)\n", ids[i], (int)why);
            }
    }
    return 0;
}

static int radius_impl(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr,
                       int skip_self, bool fill, int32_t *nbr, int32_t *cnt, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_radius_f32: N out of range");
    DMET_REQUIRE(D >= 1 && D <= 8, "dmet_radius_f32: D=%d not in [1,8]", D);
    DMET_REQUIRE(max_nbr >= 1, "dmet_radius_f32: max_nbr=%d", max_nbr);
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(x && ptr && nbr && cnt, "dmet_radius_f32: null pointer");
    const float r2 = r * r;
    const int64_t blocks = (N + 4 * kWave - 1) / (4 * kWave);
    hipStream_t st = as_stream(stream);
    // empty slots are -1: one coalesced fill instead of per-lane tail stores (294 MB for 288 000 x 255: the counted
    // form leaves them unwritten, its consumers go by cnt)
    if (fill) {
        hipError_t me = hipMemsetAsync(nbr, 0xff, sizeof(int32_t) * (size_t)N * (size_t)max_nbr, st);
        if (me != hipSuccess) return hip_fail(me, "hipMemsetAsync(nbr)");
    }
    if (D <= 2)
        hipLaunchKernelGGL((radius_kernel<2>), dim3((unsigned)blocks), dim3(kWave * 4), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, skip_self, nbr, cnt);
    else if (D <= 4)
        hipLaunchKernelGGL((radius_kernel<4>), dim3((unsigned)blocks), dim3(kWave * 4), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, skip_self, nbr, cnt);
    else
        hipLaunchKernelGGL((radius_kernel<8>), dim3((unsigned)blocks), dim3(kWave * 4), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, skip_self, nbr, cnt);
    DMET_LAUNCH_CHECK("radius_kernel");
    return 0;
}

extern "C" int dmet_radius_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr,
                               int skip_self, int32_t *nbr, int32_t *cnt, dmet_stream_t stream)
{
    return radius_impl(x, ptr, B, N, D, r, max_nbr, skip_self, true, nbr, cnt, stream);
}

extern "C" int dmet_radius_counted_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r,
                                       int max_nbr, int skip_self, int32_t *nbr, int32_t *cnt, dmet_stream_t stream)
{
    return radius_impl(x, ptr, B, N, D, r, max_nbr, skip_self, false, nbr, cnt, stream);
}

extern "C" size_t dmet_radius_workspace_bytes(int64_t N)
{
    return N > 0 ? sizeof(int32_t) * (size_t)N + 512 : 0;
}

extern "C" int dmet_radius_windowed_local_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r,
                                              int max_nbr, int skip_self, int fill, int32_t *nbr, int32_t *cnt,
                                              uint16_t *nbr16, int stride16, void *ws, size_t ws_bytes,
                                              dmet_stream_t stream)
{
    DMET_REQUIRE(!nbr16 || (stride16 >= max_nbr && stride16 % 8 == 0 && aligned16(nbr16)),
                 "dmet_radius_windowed_local_f32: nbr16 rows need a 16-byte aligned stride of >= max_nbr ids (stride16=%d)",
                 stride16);
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647, "dmet_radius_windowed_f32: N out of range");
    DMET_REQUIRE(D >= 1 && D <= 8, "dmet_radius_windowed_f32: D=%d not in [1,8]", D);
    DMET_REQUIRE(max_nbr >= 1, "dmet_radius_windowed_f32: max_nbr=%d", max_nbr);
    if (N == 0 || B == 0) return 0;
    // nbr == NULL: only the uint16 rows are written (a caller whose consumers read those: the 255-wide int32 table is 294 MB
    // of address space at 288 000 nodes, its ~36 used slots per row 41 MB of 16-byte pieces: 12 of the kernel's 130 us)
    DMET_REQUIRE(x && ptr && cnt && ws && (nbr || (nbr16 && !fill)), "dmet_radius_windowed_f32: null pointer");
    DMET_REQUIRE(ws_bytes >= dmet_radius_workspace_bytes(N), "dmet_radius_windowed_f32: workspace too small");
    int32_t *order = reinterpret_cast<int32_t *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    const float r2 = r * r;
    const int64_t blocks = (N / kWave + B + 1 + 3) / 4;   // event-aligned wavefront ids, four per workgroup
    hipStream_t st = as_stream(stream);
    if (fill) {
        hipError_t me = hipMemsetAsync(nbr, 0xff, sizeof(int32_t) * (size_t)N * (size_t)max_nbr, st);
        if (me != hipSuccess) return hip_fail(me, "hipMemsetAsync(nbr)");
    }
    hipLaunchKernelGGL(radius_order_kernel, dim3((unsigned)B), dim3(kRadBins), 0, st, x, ptr, B, D, order);
    DMET_LAUNCH_CHECK("radius_order_kernel");
    if (D <= 2)
        hipLaunchKernelGGL((radius_window_kernel<2>), dim3((unsigned)blocks), dim3(kWave * 4), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, skip_self, order, nbr, cnt, nbr16, stride16);
    else if (D <= 4)
        hipLaunchKernelGGL((radius_window_kernel<4>), dim3((unsigned)blocks), dim3(kWave * 4), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, skip_self, order, nbr, cnt, nbr16, stride16);
    else
        hipLaunchKernelGGL((radius_window_kernel<8>), dim3((unsigned)blocks), dim3(kWave * 4), 0, st, x, ptr, B, N, D, r2,
                           max_nbr, skip_self, order, nbr, cnt, nbr16, stride16);
    DMET_LAUNCH_CHECK("radius_window_kernel");
    return 0;
}

extern "C" int dmet_radius_windowed_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r,
                                        int max_nbr, int skip_self, int fill, int32_t *nbr, int32_t *cnt, void *ws,
                                        size_t ws_bytes, dmet_stream_t stream)
{
    return dmet_radius_windowed_local_f32(x, ptr, B, N, D, r, max_nbr, skip_self, fill, nbr, cnt, nullptr, 0, ws, ws_bytes,
                                          stream);
}
