/*
 * dmet.h -- C ABI of libdmet_hip.so: the MI355X (gfx950) implementation of the graph operators on the
 * DeepMETv2 DynamicEdgeConv hot path.
 *
 * The reference (DeepMETv2, /root/reference) is pure Python; the kernels it executes for this path live in
 * the torch_cluster / torch_geometric / torch_scatter wheels.  Each entry point below names the reference
 * call site (file:line, relative to /root/reference) whose third-party operator it replaces.  The reference
 * side binds this library with ctypes; deepmetv2_amd/_lib.py is that binding.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every pointer is a DEVICE pointer unless it says "host".
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no host
 *    synchronisation, allocates nothing and keeps no global mutable state (re-entrant, graph-capturable).
 *    Scratch comes from the caller: ask dmet_*_workspace_bytes(), pass `ws`/`ws_bytes`.
 *  - return value: 0 = ok; < 0 = failure (-EINVAL = -22 for bad arguments, -(1000+hipError_t) for a HIP
 *    runtime error).  dmet_last_error() returns a thread-local message for the last failure.
 *  - events (graphs) are the ragged units: event b owns nodes ptr[b] .. ptr[b+1]-1 (int64, B+1 entries).
 *  - neighbour tables are fixed width: nbr[N, k] int32 GLOBAL node ids, -1 = no neighbour (short events).
 *  - all floating point is IEEE fp32 unless the name says otherwise.
 */
#ifndef DMET_H_
#define DMET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dmet_stream_t; /* hipStream_t */

#define DMET_VERSION 1
#define DMET_MAX_K 64       /* upstream torch_cluster rejects k > 100; this build supports k <= 64   */
#define DMET_MAX_KNN_DIM 64 /* feature width of the kNN space (hot path: 32; (eta,phi) graph: 2)      */
#define DMET_MAX_H 128      /* feature width of EdgeConv inputs/outputs (hot path: 32)                */

int dmet_version(void);
const char *dmet_last_error(void);
/* 1 if a HIP device is usable by this process, 0 otherwise (never fails). */
int dmet_device_available(void);

/* ---- K1: kNN graph build -------------------------------------------------------------------------
 * replaces torch_cluster.knn_graph / torch_cluster.knn
 *   call sites: model/graph_met_network.py:63, model/dynamic_reduction_network.py:86,94; inside PyG
 *   DynamicEdgeConv.forward.
 * For every node i: the k nodes j of the same event with the smallest squared L2 distance
 *   d(i,j) = sum_c fmaf(x[j,c]-x[i,c], x[j,c]-x[i,c], acc)  (sequential in c, fp32; rule R1)
 * ordered by (d, j) ascending; ties keep the lower j (rule R2); candidates at d >= 1e10 are never
 * selected (upstream sentinel).  Self is a candidate like any other (loop handling is host-side).
 * nbr[N,k] int32 (required), dist[N,k] fp32 (required): the k neighbours and their R1 distances.
 * One launch for all ragged events.  1 <= k <= DMET_MAX_K, 1 <= D <= DMET_MAX_KNN_DIM. */
size_t dmet_knn_workspace_bytes(int64_t N, int B, int D, int k);
int dmet_knn_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                 float *dist, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* Same call, additionally writing the table as event-local ids: nbr_local[N,k] uint16 = nbr - ptr[event], 0xFFFF
 * for an empty slot (4-byte aligned; may be NULL = dmet_knn_f32).  Written by the same kernels that write nbr, for
 * dmet_gather_max_lds16_f32.  Rows of events with more than 65535 nodes are unspecified. */
int dmet_knn_local_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                       float *dist, uint16_t *nbr_local, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* dmet_knn_local_f32 for the DynamicEdgeConv call shape (model/graph_met_network.py:63: the graph is built in the
 * space of the rows the convolution then consumes): the node-level dense layer of the fused EdgeConv,
 *   P = x.(W1-W2)^T + b, Q = x.W2^T   (W[32,64], D = 32),
 * depends on x only, like the graph, and is computed by trailing workgroups of the matrix-core filter launch: they are
 * dispatched last and fill the wavefront slots the build's last round leaves empty.  layout = 0: P, Q fp32 [N,32]
 * (dmet_node_linear_split_f32), 1: slice-major [4][N][8] (dmet_node_linear_split_sliced_f32), 2: bf16 operands on
 * the bf16 matrix cores, P fp32 [N,32], Q as bf16 bits [N,32] (dmet_node_linear_split_bf16).  *dense_done = 1: P and
 * Q were written (same bits as the call named); 0: this build took another path (D != 32, k > 20, the matrix-core
 * path switched off) and the caller launches the dense layer itself.  nbr / dist / nbr_local as above. */
int dmet_knn_local_dense_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, int k, int32_t *nbr,
                             float *dist, uint16_t *nbr_local, const float *W, const float *bias, int layout,
                             float *P, void *Q, int *dense_done, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* The block shape of model/graph_met_network.py:64-66, `emb = emb + bn(conv(emb))` followed by the next DynamicEdgeConv
 * on emb: the BatchNorm transform + residual add that PRODUCES the build's input is fused into the build's prep launch,
 *   y = (raw - mean) * (gamma * invstd) + beta (+ residual)     (the expression and bits of dmet_bn_fwd_f32's transform;
 *   mean / invstd from dmet_bn_stats_f32 or the running statistics),
 * which writes y[N,32] and cuts its tile records from the values it just formed: one pass over the rows instead of two
 * and one launch less per layer.  Then dmet_knn_local_dense_f32(y, ...) (W == NULL: dmet_knn_local_f32).  *fused = 1:
 * y, the graph (and P / Q when *dense_done) were written; *fused = 0: NOTHING was launched (the build would not take the
 * matrix-core path: D != 32, k > 20, switched off, misaligned) and the caller runs the transform and the build itself. */
int dmet_bn_knn_local_dense_f32(const float *raw, const float *residual, const float *gamma, const float *beta,
                                const float *mean, const float *invstd, float *y, const int64_t *ptr, int B, int64_t N,
                                int D, int k, int32_t *nbr, float *dist, uint16_t *nbr_local, const float *W,
                                const float *bias, int layout, float *P, void *Q, int *dense_done, int *fused, void *ws,
                                size_t ws_bytes, dmet_stream_t stream);

/* What the caller knows about the event sizes of the NEXT dmet_knn*_f32 call on this thread (a data loader has the sizes
 * on the host; ptr lives on the device): every event of that batch has min_nodes .. max_nodes nodes; 0 = unknown.  The
 * hint is spent by that one call.  Today it saves one launch: when every event takes the second filter form (800 ..
 * 65536 nodes) the merge launch of the first form's tail items is dropped.  A hint that is WRONG costs time, never
 * results: an event the hint ruled out sends the queries concerned through the exact kernel. */
int dmet_knn_size_hint(int min_nodes, int max_nodes);

/* Diagnostics of the matrix-core kNN path (D = 32 or 64, k <= 20): dmet_knn_f32 first ranks candidates with an MFMA
 * filter (fp16 operands for events of 800..65536 nodes, a bf16 split for smaller ones), re-ranks the kept ones with
 * the exact R1 chain and certifies every query; uncertified queries are recomputed exactly (one workgroup per query
 * when their 128-query tile has at most 8 of them, by the exact tile kernel otherwise).  Rows with a feature of
 * magnitude >= 16384 (or not finite) in an event of 2048..65536 nodes are outside the fp16 operand range: they stay
 * candidates of every query and are themselves recomputed exactly -- same results, slower.  For the LAST dmet_knn_f32 call on this workspace: out[0] = tiles holding an uncertified query,
 * out[1] = uncertified queries (both 0 when the exact kernel ran alone).  Synchronises the stream.
 * Environment: DMET_KNN_PATH=exact forces the exact kernel for everything. */
int dmet_knn_fallback_stats(const void *ws, int64_t N, int B, int D, int k, int64_t *out, dmet_stream_t stream);

/* ---- N1: radius graph build ----------------------------------------------------------------------
 * replaces torch_cluster.radius_graph   call sites: train.py:48, evaluate.py:88, plt_weight.py:122
 * For every node i: the FIRST max_nbr nodes j (ascending j) of the same event with d(i,j) < r*r
 * (strict, r*r formed in fp32).  nbr[N,max_nbr] int32 (-1 padded), cnt[N] int32 = entries stored in the row.
 * skip_self != 0 reproduces upstream's loop=False (call it with max_nbr = max_num_neighbors + 1): node i counts
 * towards the search limit when it is met but is not stored. */
int dmet_radius_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr, int skip_self,
                    int32_t *nbr, int32_t *cnt, dmet_stream_t stream);
/* Same, but slots >= cnt[i] of row i are left UNWRITTEN (no -1 fill of the max_nbr-wide table: the reference's
 * max_num_neighbors=255 rows are ~36 deep, so the fill is most of the table's bytes).  For consumers that go by cnt:
 * dmet_gather_max_counted_f32, dmet_table_degree / dmet_table_edges, the arg-addressed backward. */
int dmet_radius_counted_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr,
                            int skip_self, int32_t *nbr, int32_t *cnt, dmet_stream_t stream);
/* The same table (bit-identical: same candidates in the same order, same distance arithmetic) without multiplying out
 * all n^2 pairs of an event: queries are processed in the order of their FIRST coordinate, and a wavefront only keeps
 * the candidates whose first coordinate lies within r of its 64 queries' range (stream compaction in index order).
 * In (eta, phi) with r = 0.4 about 85 % of the pairs are never formed.  fill != 0: -1 fill as dmet_radius_f32,
 * fill == 0: as dmet_radius_counted_f32.  ws: dmet_radius_workspace_bytes(N) bytes (the processing order). */
size_t dmet_radius_workspace_bytes(int64_t N);
int dmet_radius_windowed_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr,
                             int skip_self, int fill, int32_t *nbr, int32_t *cnt, void *ws, size_t ws_bytes,
                             dmet_stream_t stream);
/* Same, and optionally a second copy of every row as EVENT-LOCAL uint16 ids (nbr16 != NULL: rows of stride16 ids,
 * stride16 % 8 == 0, stride16 >= max_nbr, 16-byte aligned; slots cnt[i] .. roundup8(cnt[i]) - 1 hold 0xFFFF, the rest
 * of the row is unwritten; meaningful for events of at most 65534 nodes) for dmet_gather_max_local_j16_f32.
 * With nbr16 given and fill == 0, nbr may be NULL: the int32 table is then not written at all (for callers whose
 * consumers read the uint16 rows -- 288 000 x 255 int32 slots are 294 MB of address space, the ~36 used slots per row
 * 41 MB of 16-byte pieces). */
int dmet_radius_windowed_local_f32(const float *x, const int64_t *ptr, int B, int64_t N, int D, float r, int max_nbr,
                                   int skip_self, int fill, int32_t *nbr, int32_t *cnt, uint16_t *nbr16, int stride16,
                                   void *ws, size_t ws_bytes, dmet_stream_t stream);

/* ---- K2+K3 fused: EdgeConv with nn = Linear(2*Hin -> Hout), aggr = 'max', fixed-width table ---------
 * replaces torch_geometric.nn.EdgeConv(nn=Sequential(Linear(2H,H)), aggr='max').forward
 *   constructed model/graph_met_network.py:36-38, invoked :65 (static graph) / :63 (dynamic kNN).
 *   out[i] = max_{s: nbr[i,s]>=0} ( W . [x_i || x_j - x_i] + b ),  j = nbr[i,s];  no neighbour -> 0 (R3)
 * computed through the exact split  W.[x_i || x_j-x_i] + b = (W1-W2).x_i + b + W2.x_j :
 *   step 1 (fp32 MFMA):  P[i] = (W1-W2).x_i + b,  Q[i] = W2.x_i           (tables [N,Hout] in ws)
 *   step 2 (gather+max): out[i] = P[i] + max_s Q[nbr[i,s]],  arg[i,c] = winning slot s (lowest on ties, R4)
 * W is torch Linear.weight layout [Hout, 2*Hin] row-major; arg[N,Hout] uint8 may be NULL (inference);
 * arg = 255 marks a node without neighbours.  Hin, Hout multiples of 32, <= DMET_MAX_H; k <= DMET_MAX_K. */
size_t dmet_edgeconv_linear_workspace_bytes(int64_t N, int Hout);
int dmet_edgeconv_linear_max_fwd_f32(const float *x, const int32_t *nbr, const int64_t *ptr, int B,
                                     int64_t N, int k, int Hin, int Hout, const float *W,
                                     const float *b, float *out, uint8_t *arg, void *ws,
                                     size_t ws_bytes, dmet_stream_t stream);
/* The same operator in ONE launch for events of up to 5119 nodes (Hin = Hout = 32, k in {8,16,32}): one workgroup
 * per (event, 8-channel slice) computes [Q slice | P slice] on the fp32 matrix cores (v_mfma_f32_16x16x4_f32),
 * keeps the Q slice of the whole event in LDS and gathers the k neighbour rows from there -- neighbour gather +
 * edge MLP + max with no P/Q round trip through memory.  Larger events are evaluated edge by edge (correct, slow). */
int dmet_edgeconv_fused_lds_f32(const float *x, const int32_t *nbr, const int64_t *ptr, int B, int64_t N, int k,
                                int Hin, int Hout, const float *W, const float *b, float *out, uint8_t *arg,
                                dmet_stream_t stream);
/* The two steps individually (step 2 is "the gather + scatter_max kernel" of BASELINE.json). */
int dmet_node_linear_split_f32(const float *x, int64_t N, int Hin, int Hout, const float *W,
                               const float *b, float *P, float *Q, dmet_stream_t stream);
int dmet_gather_max_f32(const float *P, const float *Q, const int32_t *nbr, const int64_t *ptr, int B,
                        int64_t N, int k, int H, float *out, uint8_t *arg, dmet_stream_t stream);
/* Slice-major pair: step 1 writes P and Q as [Hout/8][N][8] (the 8-channel slice of every node contiguous) and the
 * LDS-resident gather reads them that way -- its (event, slice) workgroups then stage Q and stream P as contiguous
 * bytes instead of 32-byte pieces of 128-byte rows.  out / arg stay row-major; nbr_local may be NULL; k in {8,16,20,32},
 * H % 8 == 0.  Same results as dmet_node_linear_split_f32 + dmet_gather_max_lds_f32. */
int dmet_node_linear_split_sliced_f32(const float *x, int64_t N, int Hin, int Hout, const float *W,
                                      const float *b, float *P, float *Q, dmet_stream_t stream);
/* The same dense layer with its input rows FORMED on the fly as y = residual + BatchNorm(raw) -- the block shape
 * `emb = emb + bn(conv(emb, edge_index))` of model/graph_met_network.py:65-66 followed by the next EdgeConv over the SAME
 * static graph (the reference's active flow, train.py:48: no kNN build whose prep launch could carry the transform):
 * y[N,H] is written (the block's output), P / Q are computed from the registers; y has the bits of dmet_bn_fwd_f32's
 * transform, P / Q those of dmet_node_linear_split[_sliced]_f32 on that y.  mean / invstd: dmet_bn_stats_f32 (training)
 * or dmet_bn_eval_stats_f32.  residual may be NULL.  H = 32 -> 32 only; sliced != 0: slice-major P / Q. */
int dmet_bn_node_linear_split_f32(const float *raw, const float *residual, const float *gamma, const float *beta,
                                  const float *mean, const float *invstd, float *y, int64_t N, int H, const float *W,
                                  const float *b, int sliced, float *P, float *Q, dmet_stream_t stream);
int dmet_gather_max_lds_sliced_f32(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr_local,
                                   const int64_t *ptr, int B, int64_t N, int k, int H, float *out, uint8_t *arg,
                                   dmet_stream_t stream);
/* Same with a HINT on the batch's largest event (max_nodes; 0 = unknown = the entry above).  The kernel above is one
 * workgroup per CU whatever the event size -- by its 160 KB image and by its registers -- so its staging phase overlaps with
 * nothing.  For batches whose largest event fits 2 559 rows -- what model/data_loader.py:67-90 yields on real data -- and
 * uint16 ids given, workgroups of 512 threads on 80 KB images run two to a CU (one stages while the other gathers).  An
 * event beyond the hint takes the in-kernel L2 path: slower, never wrong.  Same bits. */
int dmet_gather_max_lds_sliced_cap_f32(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr_local,
                                       const int64_t *ptr, int B, int64_t N, int k, int H, float *out, uint8_t *arg,
                                       int64_t max_nodes, dmet_stream_t stream);
/* Same contract with a per-node slot count: only the first cnt[i] slots of row i are examined (radius tables are
 * max_num_neighbors wide but a few dozen deep).  cnt may be NULL (= k for every node). */
int dmet_gather_max_counted_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt, int64_t N,
                                int k, int H, float *out, uint8_t *arg, dmet_stream_t stream);
/* Same contract, LDS-resident form: one workgroup per (event, 8-channel slice) stages that slice of Q for the
 * whole event in the CU's 160 KB LDS (events up to 5120 nodes; larger events gather from global memory inside the
 * same kernel).  Needs ptr/B and H % 8 == 0.  The faster form for events of a few thousand nodes. */
int dmet_gather_max_lds_f32(const float *P, const float *Q, const int32_t *nbr, const int64_t *ptr, int B,
                            int64_t N, int k, int H, float *out, uint8_t *arg, dmet_stream_t stream);
/* LDS-resident form of dmet_gather_max_counted_f32 (radius tables: kmax slots per row, the first cnt[i] used): the
 * (event, 8-channel slice) workgroups and LDS image of dmet_gather_max_lds_f32; events that do not fit gather from
 * global memory inside the same kernel.  Identical results. */
int dmet_gather_max_counted_lds_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt,
                                    const int64_t *ptr, int B, int64_t N, int k, int H, int pq_sliced, float *out,
                                    uint8_t *arg, dmet_stream_t stream);   /* pq_sliced != 0: P / Q are the slice-major
                                    tables of dmet_node_linear_split_sliced_f32 */
/* Winner-id form of dmet_gather_max_counted_lds_f32 for the reference's active flow (one 255-wide radius table per
 * batch, train.py:48): instead of the winning slot it stores, per (node, channel), the winner's EVENT-LOCAL node id
 * (argj[N,H] uint16, 0xFFFF = no neighbour; events of at most 65534 nodes), so that the backward scatter
 * (dmet_gather_max_bwd_j16_f32) needs no look-up in the wide table; `order` (optional, from dmet_table_order_by_count:
 * the event's local node indices grouped by slot count) makes the lane pairs of a wavefront walk rows of similar depth.
 * out is identical to dmet_gather_max_counted_f32. */
int dmet_table_order_by_count(const int32_t *cnt, const int64_t *ptr, int B, int64_t N, int32_t *order,
                              dmet_stream_t stream);
int dmet_gather_max_counted_lds_j16_f32(const float *P, const float *Q, const int32_t *nbr, const int32_t *cnt,
                                        const int32_t *order, const int64_t *ptr, int B, int64_t N, int k, int H,
                                        int pq_sliced, float *out, uint16_t *argj, dmet_stream_t stream);
/* Second form of the ids for the same gather: rows of event-local uint16 ids written by the radius kernel itself
 * (dmet_radius_windowed_local_f32: nbr16[N][stride16], stride16 a multiple of 8 and >= max_nbr, 16-byte aligned; the
 * last started chunk of 8 of every row is padded with 0xFFFF; events of at most 65534 nodes): one aligned 16-byte
 * load per 8 slots, no packing pass.  Identical out / argj.  argj == NULL (inference): the maximum alone. */
int dmet_gather_max_local_j16_f32(const float *P, const float *Q, const uint16_t *nbr16, int stride16,
                                  const int32_t *cnt, const int32_t *order, const int64_t *ptr, int B, int64_t N,
                                  int kmax, int H, int pq_sliced, float *out, uint16_t *argj, dmet_stream_t stream);
/* gQ[j,c] = sum of g_out[i,c] over the (i,c) whose winner is j = ptr[event] + argj[i,c]: the per-event LDS scatter
 * with exact integer sums of dmet_gather_max_bwd_lds_f32 (bitwise reproducible), H = 32. */
int dmet_gather_max_bwd_j16_f32(const float *g_out, const uint16_t *argj, const int64_t *ptr, int B, int64_t N, int H,
                                float *gQ, dmet_stream_t stream);
/* Same again with the table ALSO given as event-local uint16 ids (nbr_local from dmet_knn_local_f32; k in {8,16,20,32},
 * 16-byte aligned): events that fit the LDS image read their ids from it -- half the id bytes, and every one of the
 * H/8 slice workgroups of an event re-reads the ids, so this is a third of the kernel's L2 requests.  Larger events
 * read nbr as before.  Results are identical to dmet_gather_max_lds_f32. */
int dmet_gather_max_lds16_f32(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr_local,
                              const int64_t *ptr, int B, int64_t N, int k, int H, float *out, uint8_t *arg,
                              dmet_stream_t stream);
/* Ragged batches (BASELINE configs[4]: events of 500..8000 nodes): the gather form is chosen PER EVENT inside one call.
 * Events whose Q slice fits the LDS image (<= 5119 nodes) take the LDS-resident kernel, larger ones the L2-form kernel
 * (8 lanes per node, 16 row gathers in flight); both read the row-major P / Q tables and each skips the other's
 * events.  nbr_local may be NULL.  H = 32 and k in {8,16,20,32}; anything else is dmet_gather_max_f32.  Results are
 * identical to dmet_gather_max_f32.  Replaces the same reference lines as dmet_gather_max_f32
 * (model/graph_met_network.py:63,65 through torch_geometric.nn.EdgeConv / torch_scatter.scatter(max)). */
int dmet_gather_max_mixed_f32(const float *P, const float *Q, const int32_t *nbr, const uint16_t *nbr_local,
                              const int64_t *ptr, int B, int64_t N, int k, int H, float *out, uint8_t *arg,
                              dmet_stream_t stream);
/* Two-layer edge MLP on the bf16 matrix cores, fused with the aggregation (BASELINE configs[2]; the DRN call shape
 * model/dynamic_reduction_network.py:59-73,86-87): for nn = Linear(2 Hin, H1) - ELU - Linear(H1, H2) [- ELU] (act2 != 0)
 * over a fixed-width table nbr[N,k] (-1 = empty slot):
 *   out[i,:] = aggr_{s: nbr[i,s] >= 0} nn([x_i || x_{nbr[i,s]} - x_i]),  aggr = 0: max (0 for a node without neighbours,
 *   rule R3), 1: add.   W1[H1, 2 Hin], W2[H2, H1] row-major as torch.nn.Linear.weight; b1 / b2 may be NULL.
 * Inputs and weights are rounded to bf16, products accumulate in fp32, ELU and the aggregation are fp32: the bf16 bar
 * of rule R6 (rtol 2e-2) against the fp32 operators.  Nothing per-edge is written to memory.
 * Built for (Hin, H1, H2) in {(32, <= 64, 32 | 64), (64, <= 128, 64)} and k in {8, 16, 32}: dmet_edge_mlp2_supported.
 * Replaces torch_geometric.nn.EdgeConv.forward (index_select x2 -> cat -> nn -> scatter) for that `nn`. */
int dmet_edge_mlp2_supported(int Hin, int H1, int H2, int k);
int dmet_edge_mlp2_bf16(const float *x, int64_t N, int Hin, const int32_t *nbr, int k, const float *W1, const float *b1,
                        int H1, const float *W2, const float *b2, int H2, int act2, int aggr, float *out,
                        dmet_stream_t stream);
/* The same edge MLP followed by the BatchNorm1d(H2) that ends the DRN's `nn` (model/dynamic_reduction_network.py:59-70;
 * it normalises the per-edge MESSAGES, before the aggregation).  One pass over the edges: the affine map of the norm
 * commutes with the aggregation (sum: a S + cnt b; max: a max + b for a >= 0, a min + b otherwise), so the kernel writes
 * un-normalised aggregates plus per-wavefront partial sums of m and m^2 over the valid edges, a one-workgroup kernel turns
 * them into the per-channel (a, b) -- training != 0: batch statistics over the E valid edges (biased variance;
 * running_mean / running_var / num_batches_tracked, all optional, updated like torch.nn.BatchNorm1d); training == 0:
 * running statistics -- and a node-level kernel applies them.  Nodes without any neighbour give 0 (R3).
 * ws: dmet_edge_mlp2_bn_workspace_bytes(N, H2). */
size_t dmet_edge_mlp2_bn_workspace_bytes(int64_t N, int H2);
int dmet_edge_mlp2_bn_bf16(const float *x, int64_t N, int Hin, const int32_t *nbr, int k, const float *W1, const float *b1,
                           int H1, const float *W2, const float *b2, int H2, int act2, int aggr, const float *gamma,
                           const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                           int64_t *num_batches_tracked, int training, float *out, void *ws, size_t ws_bytes,
                           dmet_stream_t stream);
/* bf16 variant (BASELINE configs[2]): x and the split weights rounded to bf16 (RNE), multiplied on the bf16 matrix
 * cores with fp32 accumulation; P stays fp32, Q is stored as bf16 (raw bits) and gathered as 64-B rows.
 * Built for Hin = Hout = 32, k in {8,16,32}.  Backward is shared with the fp32 path (arg-based, fp32). */
int dmet_node_linear_split_bf16(const float *x, int64_t N, int Hin, int Hout, const float *W, const float *b,
                                float *P, uint16_t *Qh, dmet_stream_t stream);
int dmet_gather_max_bf16q(const float *P, const uint16_t *Qh, const int32_t *nbr, int64_t N, int k, int H,
                          float *out, uint8_t *arg, dmet_stream_t stream);
/* Backward of step 2 w.r.t. Q:  gQ[j,c] = sum over (i,s) with nbr[i,s]==j and arg[i,c]==s of g_out[i,c].
 * Deterministic (no float atomics): walks the reverse index rev_ptr[N+1] (int32), rev_slot[E] (int32,
 * entry = i*k+s, ascending inside a row) built by dmet_reverse_index(nbr, N*k, N, ...). */
int dmet_gather_max_bwd_f32(const float *g_out, const uint8_t *arg, const int32_t *rev_ptr,
                            const int32_t *rev_slot, int64_t N, int k, int H, float *gQ,
                            dmet_stream_t stream);
/* The same gQ without a reverse index (H = 32, node-sorted events as in ptr[B+1]; every nbr entry of a node lies in
 * the node's own event): a workgroup per (event, 4 channels) scatter-adds into LDS with 64-bit fixed-point integer
 * atomics -- integer sums are order-independent, hence bitwise reproducible; each term is rounded once to 2^-30 of the
 * slice's max |g_out|, the sums are exact.  Replaces the radix sort + L2 gather of the route above on the hot path. */
int dmet_gather_max_bwd_lds_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr, const int64_t *ptr, int B,
                                int64_t N, int k, int H, float *gQ, dmet_stream_t stream);
/* Same, reading the winning neighbour's id from the event-local uint16 table of dmet_knn_local_f32 (nbr_local, may
 * be NULL) for events of at most 65535 nodes -- 32-byte instead of 64-byte rows under the scattered 2-byte reads. */
int dmet_gather_max_bwd_lds16_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr,
                                  const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k, int H,
                                  float *gQ, dmet_stream_t stream);
/* The same two scatters with a HINT on the batch's largest event (max_nodes; 0 = unknown = the entries above): a
 * workgroup's accumulators are (event nodes) x 4 channels x 8 bytes of LDS, so for batches of small events -- what
 * model/data_loader.py:67-90 yields on real data, 1 000-2 500 candidates per event -- the workgroups are sized down
 * (<= 1 152 nodes: 256 threads, 36 KB; <= 2 304: 512 threads, 72 KB) and several share a CU.  Events larger than the
 * hint take more passes: slower, never wrong.  Same bits as the unhinted entries (integer sums). */
int dmet_gather_max_bwd_lds16_cap_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr,
                                      const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k, int H,
                                      float *gQ, int64_t max_nodes, dmet_stream_t stream);
int dmet_gather_max_bwd_j16_cap_f32(const float *g_out, const uint16_t *argj, const int64_t *ptr, int B, int64_t N, int H,
                                    float *gQ, int64_t max_nodes, dmet_stream_t stream);
/* The pair scatter -> node-level kernel with gQ handed over SLICE-MAJOR, gQs[8][N][4] floats: a scatter workgroup owns 4
 * channels of one event, so in this layout it writes ONE contiguous run instead of a 16-byte piece of every 128-byte
 * row (13 of the scatter's 46 us at 64 x 4500 nodes were that write-out), and the node-level kernel's loads stay whole
 * 128-byte lines either way.  Same sums, same bits; only the layout of the intermediate differs.  arg_is_j16: `arg` holds
 * uint16 winner ids (dmet_gather_max_bwd_j16_sliced_f32) instead of uint8 slots. */
int dmet_gather_max_bwd_sliced_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr, const uint16_t *nbr_local,
                                   const int64_t *ptr, int B, int64_t N, int k, int H, float *gQs, int64_t max_nodes,
                                   dmet_stream_t stream);
int dmet_gather_max_bwd_j16_sliced_f32(const float *g_out, const uint16_t *argj, const int64_t *ptr, int B, int64_t N,
                                       int H, float *gQs, int64_t max_nodes, dmet_stream_t stream);
int dmet_edgeconv_linear_bwd_sliced_f32(const float *x, const float *W, const float *g_out, const void *arg,
                                        int arg_is_j16, const float *gQs, const float *g_add, int64_t N, int H, float *gx,
                                        float *gW, float *gb, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* Reverse index: a stable sort of the positions 0..M-1 of an int32 key array by key value.
 *   rev_ptr[num_keys+1]: rev_pos[rev_ptr[j] .. rev_ptr[j+1]-1] = the positions holding key j, ascending.
 * Keys outside [0, num_keys) (the -1 "no neighbour" entries) sort last and are not indexed.
 * For a neighbour table pass keys = nbr, M = N*k, num_keys = N (position = i*k+s). */
size_t dmet_reverse_index_workspace_bytes(int64_t M, int64_t num_keys);
int dmet_reverse_index(const int32_t *keys, int64_t M, int64_t num_keys, int32_t *rev_ptr,
                       int32_t *rev_pos, void *ws, size_t ws_bytes, dmet_stream_t stream);

/* ---- K2 / K3 un-fused: arbitrary `nn`, arbitrary edge lists ------------------------------------------
 * replaces the PyG MessagePassing.propagate pieces around a user `nn`
 *   (model/dynamic_reduction_network.py:59-73,86-87: Linear-ELU-Linear-ELU-BN, aggr in {'add','max'}).
 * Edges are (src[e] -> tgt[e]) int32, grouped by target: rowptr[N+1] int32 with tgt[e]==i for
 * rowptr[i] <= e < rowptr[i+1].
 *   edge_features: feat[e] = [ x[tgt[e]] || x[src[e]] - x[tgt[e]] ]                      ([E, 2H])
 *   segment_max  : out[i,c] = max_e msg[e,c] (empty -> 0), arg[i,c] = winning e (lowest on ties), -1 if empty
 *   segment_sum  : out[i,c] = sum_e msg[e,c] in ascending e (deterministic)
 *   segment_max_bwd: g_msg[e,c] = (arg[tgt(e),c]==e) ? g_out[tgt(e),c] : 0
 *   edge_features_bwd: gx[i] = sum_{e in in(i)} (g_feat[e,:H] - g_feat[e,H:]) + sum_{e in out(i)} g_feat[e,H:]
 *                      out(i) walked through the by-source index srcptr[N+1], srcperm[E] (ascending e). */
int dmet_edge_features_f32(const float *x, const int32_t *src, const int32_t *tgt, int64_t E, int H,
                           float *feat, dmet_stream_t stream);
int dmet_segment_max_f32(const float *msg, const int32_t *rowptr, int64_t N, int H, float *out,
                         int32_t *arg, dmet_stream_t stream);
int dmet_segment_sum_f32(const float *msg, const int32_t *rowptr, int64_t N, int H, float *out,
                         dmet_stream_t stream);
int dmet_segment_max_bwd_f32(const float *g_out, const int32_t *arg, const int32_t *rowptr, int64_t N,
                             int H, float *g_msg, dmet_stream_t stream);
int dmet_segment_sum_bwd_f32(const float *g_out, const int32_t *rowptr, int64_t N, int H, float *g_msg,
                             dmet_stream_t stream);
int dmet_edge_features_bwd_f32(const float *g_feat, const int32_t *rowptr, const int32_t *srcptr,
                               const int32_t *srcperm, int64_t N, int H, float *gx,
                               dmet_stream_t stream);

/* ---- K4: per-event MET reduction -------------------------------------------------------------------
 * replaces the two torch_scatter.scatter_add calls at model/net.py:55-56 (and :132-133):
 *   met[b,0] = sum_{i in event b} w[i]*x[i*x_stride+0],  met[b,1] = sum w[i]*x[i*x_stride+1]
 * Deterministic fixed-shape tree (one workgroup per event; lane-strided partials, wavefront shuffle
 * tree, then waves in order).  bwd: g_w[i] = g_met[b,0]*px_i + g_met[b,1]*py_i. */
int dmet_met_reduce_f32(const float *w, const float *x, int64_t x_stride, const int64_t *ptr, int B,
                        float *met, dmet_stream_t stream);
int dmet_met_reduce_bwd_f32(const float *g_met, const float *x, int64_t x_stride, const int64_t *ptr,
                            int B, int64_t N, float *g_w, dmet_stream_t stream);
/* H1: torch.optim.AdamW's step (train.py:52,75; decoupled weight decay, no amsgrad) on one flat fp32 tensor, one launch.
 * Device-side step state, advanced by the launch (replayable in a hipGraph): step[0] = number of steps taken (float),
 * bias_pow[2] = beta1^step, beta2^step as running products in double (both 1.0 before the first step).
 * Hyper-parameters are doubles (python floats): 1 - beta is formed before the rounding to fp32, as torch does. */
int dmet_adamw_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, double *bias_pow,
                   int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                   dmet_stream_t stream);
/* Same with the learning rate read from device memory (lr_dev[0], double): the reference drives it with
 * ReduceLROnPlateau (train.py:76, stepped at train.py:58), and a launch captured in a hipGraph must see the change. */
int dmet_adamw_lr_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *step, double *bias_pow,
                      int64_t n, const double *lr_dev, double beta1, double beta2, double eps, double weight_decay,
                      dmet_stream_t stream);
/* loss[0] = 0.5 * mean_b((met[b,0] + truth[b,0])^2 + (met[b,1] + truth[b,1])^2)  (model/net.py:58-61) and
 * g_met[B,2] = d loss / d met, one launch, fixed summation order. */
int dmet_met_loss_f32(const float *met, const float *truth, int B, float *loss, float *g_met, dmet_stream_t stream);
/* The same two entries for the fused loss of model/net.py:49-62 (deepmetv2_amd.scatter.met_loss_from_weights): truth rows
 * of truth_stride floats (px, py in columns 0, 1 of the [B,11] target: no [B,2] copy), and the backward with g_met
 * multiplied by one device float first (the upstream gradient of the scalar loss: no separate multiply). */
int dmet_met_loss_strided_f32(const float *met, const float *truth, int64_t truth_stride, int B, float *loss,
                              float *g_met, dmet_stream_t stream);
int dmet_met_reduce_bwd_scaled_f32(const float *g_met, const float *scale, const float *x, int64_t x_stride,
                                   const int64_t *ptr, int B, int64_t N, float *g_w, dmet_stream_t stream);
/* Generic sorted-index form used by the scatter_add(src, batch) drop-in: out[b] = sum_{i in b} src[i]. */
int dmet_segment_sum_1d_f32(const float *src, const int64_t *ptr, int B, float *out,
                            dmet_stream_t stream);
/* Neighbour table -> edge list (what knn_graph / radius_graph return): deg[i] = valid entries of row i (among the
 * first cnt[i] slots when cnt is given); with rowptr = exclusive prefix sum of deg, dmet_table_edges writes the edges
 * grouped by target i in slot order: first/second [E] int64 = (source, target), or (target, source) when swap != 0,
 * and/or src32/tgt32 [E] int32. */
int dmet_table_degree(const int32_t *nbr, const int32_t *cnt, int64_t N, int k, int32_t *deg, dmet_stream_t stream);
int dmet_table_edges(const int32_t *nbr, const int32_t *cnt, const int32_t *rowptr, int64_t N, int k, int swap,
                     int64_t *first, int64_t *second, int32_t *src32, int32_t *tgt32, dmet_stream_t stream);
/* ptr[B+1] from a SORTED int64 batch vector (ptr[b] = first i with batch[i] >= b). */
int dmet_batch_to_ptr(const int64_t *batch, int64_t N, int B, int64_t *ptr, dmet_stream_t stream);

/* ---- N3 (first piece): weight gradients of the per-node dense layers ----------------------------------
 * The layers around the graph operators (model/graph_met_network.py:15-32,41-44) are tiny per-node Linear /
 * Embedding modules over N ~ 3e5 nodes; autograd's weight gradients for them are tall-skinny reductions
 *   C[Ha,Hb] = A^T B,  A = grad_out[N,Ha], B = input[N,Hb]        (torch Linear.weight layout [out,in])
 *   C[R,Hb]  = onehot(index[N])^T B                                (torch Embedding.weight gradient)
 * computed deterministically (fixed row ranges per wavefront, partials summed in order), Ha,Hb,R <= 64. */
size_t dmet_xty_workspace_bytes(int64_t N, int Ha, int Hb);
int dmet_xty_f32(const float *A, const float *B, int64_t N, int Ha, int Hb, float *C, void *ws,
                 size_t ws_bytes, dmet_stream_t stream);
int dmet_onehot_xty_f32(const int64_t *index, const float *B, int64_t N, int R, int Hb, float *C, void *ws,
                        size_t ws_bytes, dmet_stream_t stream);

/* ---- N3 (second piece): the per-node encoder as one kernel each way ---------------------------------------
 * model/graph_met_network.py:48-58 up to (not including) bn_all:
 *   h = ELU(Wa [ELU(Wk [Echg[chg+1] | Epdg[remap(|pdg|)] | Epv[pv]] + bk) | ELU(Wc x[:, :8] + bc)] + ba)
 * x[N,8] fp32 continuous columns with row stride x_stride floats (a view into the 11-column feature matrix is
 * fine); x_cat[N,3] int64 = (pdgId, charge, fromPV), the arguments of GraphMETNetwork.forward -- or x_cat == NULL:
 * the three columns are columns 8..10 of the same rows of x (x_stride >= 11), still as floats, and the `.long()` of
 * train.py:43 (truncation toward zero) happens inside the kernel.  Torch layouts
 * for the weights (Linear.weight [out,in], Embedding.weight [rows,8]).  Backward takes the forward output h and g_h = dL/dh and
 * writes (not accumulates) all nine parameter gradients, reduced over the nodes deterministically. */
int dmet_encode_fwd_f32(const float *x, int64_t x_stride, const int64_t *x_cat, int64_t N, const float *Wc, const float *bc,
                        const float *Wk, const float *bk, const float *Wa, const float *ba, const float *Echg,
                        const float *Epdg, const float *Epv, float *h, dmet_stream_t stream);
size_t dmet_encode_bwd_workspace_bytes(int64_t N);
int dmet_encode_bwd_f32(const float *x, int64_t x_stride, const int64_t *x_cat, int64_t N, const float *Wc, const float *bc,
                        const float *Wk, const float *bk, const float *Wa, const float *ba, const float *Echg,
                        const float *Epdg, const float *Epv, const float *h, const float *g_h, float *gWc,
                        float *gbc, float *gWk, float *gbk, float *gWa, float *gba, float *gEchg, float *gEpdg,
                        float *gEpv, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* dmet_encode_bwd_f32 behind bn_all (model/graph_met_network.py:58): g_y is the gradient with respect to the
 * BatchNorm's OUTPUT and the BatchNorm's backward transform g = gamma * invstd * (g_y - mean_g - (h - mean) * invstd *
 * mean_gx) (statistics from dmet_bn_bwd_stats_f32; h, the BatchNorm's input, is the encoder's own output) is applied
 * as the rows are loaded: the transform pass of dmet_bn_bwd_f32 disappears.  *fused = 0: nothing was launched
 * (DMET_ENCODER_BWD=valu, misaligned operands) and the caller keeps the two steps. */
int dmet_encode_bn_bwd_f32(const float *x, int64_t x_stride, const int64_t *xcat, int64_t N, const float *Wc,
                           const float *bc, const float *Wk, const float *bk, const float *Wa, const float *ba,
                           const float *Echg, const float *Epdg, const float *Epv, const float *h, const float *g_y,
                           const float *bn_gamma, const float *bn_mean, const float *bn_invstd, const float *bn_mean_g,
                           const float *bn_mean_gx, float *gWc, float *gbc, float *gWk, float *gbk, float *gWa,
                           float *gba, float *gEchg, float *gEpdg, float *gEpv, int *fused, void *ws, size_t ws_bytes,
                           dmet_stream_t stream);

/* ---- K5 (node level): backward of the fused EdgeConv dense layer, H = 32 -------------------------------------
 * With P = x (W1-W2)^T + b, Q = x W2^T, out_i = P_i + max_s Q[nbr[i,s]] (dmet_node_linear_split_f32 +
 * dmet_gather_max_f32) and gQ from dmet_gather_max_bwd_f32:
 *   gx = gP (W1-W2) + gQ W2,   gW[H,2H] = [gP^T x | gQ^T x - gP^T x],   gb = sum_i gP_i,
 * gP = g_out masked to 0 where arg == 255 (node without neighbour; arg may be NULL for dense tables).
 * One pass over the rows, fp32 MFMA, deterministic (fixed node ranges per wavefront, ordered partial sums). */
size_t dmet_edgeconv_linear_bwd_workspace_bytes(int64_t N, int H);
int dmet_edgeconv_linear_bwd_f32(const float *x, const float *W, const float *g_out, const uint8_t *arg,
                                 const float *gQ, int64_t N, int H, float *gx, float *gW, float *gb, void *ws,
                                 size_t ws_bytes, dmet_stream_t stream);
/* Same with gx += g_add[N,H] fused into the store (g_add may be NULL): the gradient that reaches x through the
 * residual connection of the block (graph_met_network.py:66: emb = emb + norm(conv(emb))), instead of a separate
 * elementwise add over the node table. */
int dmet_edgeconv_linear_bwd_add_f32(const float *x, const float *W, const float *g_out, const uint8_t *arg,
                                     const float *gQ, const float *g_add, int64_t N, int H, float *gx, float *gW,
                                     float *gb, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* Same for the winner-id form of the counted radius gather (dmet_gather_max_local_j16_f32 / _counted_lds_j16_f32:
 * train.py:48's radius graph feeding model/graph_met_network.py:65): argj[N,H] uint16 event-local winner ids,
 * 0xFFFF = the node had no neighbour at all (a query with a NaN / inf coordinate finds nobody, not even itself);
 * gP is masked to 0 there exactly as for arg == 255 above.  argj may be NULL (no masking). */
int dmet_edgeconv_linear_bwd_add_j16_f32(const float *x, const float *W, const float *g_out, const uint16_t *argj,
                                         const float *gQ, const float *g_add, int64_t N, int H, float *gx, float *gW,
                                         float *gb, void *ws, size_t ws_bytes, dmet_stream_t stream);

/* ---- N3 (third piece): BatchNorm1d over the nodes, optionally fused with the residual add ---------------
 * model/graph_met_network.py:32,39,58,66: bn_all(...) and emb + bn(conv(...)).  x[N,H] row-major, H a multiple of 4
 * up to 64.  training != 0: batch statistics (biased variance for the normalisation, unbiased for running_var, like
 * torch.nn.BatchNorm1d); running_mean/var (optional pair) are updated in place with `momentum`.  training == 0:
 * running statistics.  y = (x - mean) * gamma / sqrt(var + eps) + beta (+ residual when not NULL).
 * save_mean / save_invstd [H] are written for the backward.  Column sums use fixed row ranges per workgroup and are
 * combined in order: bitwise reproducible.  Backward (training statistics): g_x, g_gamma, g_beta (written). */
size_t dmet_bn_workspace_bytes(int64_t N, int H);
int dmet_bn_fwd_f32(const float *x, const float *residual, int64_t N, int H, const float *gamma, const float *beta,
                    float eps, float momentum, float *running_mean, float *running_var, int training, float *y,
                    float *save_mean, float *save_invstd, void *ws, size_t ws_bytes, dmet_stream_t stream);
/* Same; in training mode *num_batches_tracked (optional, int64 on the device: nn.BatchNorm1d's buffer) is incremented by
 * the statistics kernel instead of by a kernel launch of its own. */
int dmet_bn_fwd_tracked_f32(const float *x, const float *residual, int64_t N, int H, const float *gamma,
                            const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                            int64_t *num_batches_tracked, int training, float *y, float *save_mean, float *save_invstd,
                            void *ws, size_t ws_bytes, dmet_stream_t stream);
/* The statistics half of dmet_bn_bwd_f32 (column sums + finalize): g_gamma, g_beta and the two means the element
 * transform needs, for a caller that applies the transform elsewhere (dmet_encode_bn_bwd_f32). */
int dmet_bn_bwd_stats_f32(const float *x, const float *g_y, int64_t N, int H, const float *save_mean,
                          const float *save_invstd, float *g_gamma, float *g_beta, float *mean_g, float *mean_gx,
                          void *ws, size_t ws_bytes, dmet_stream_t stream);
/* Eval mode: save_mean = running_mean, save_invstd = 1 / sqrt(running_var + eps) (the constants dmet_bn_fwd_f32 uses with
 * training == 0), for callers that apply the transform elsewhere (dmet_bn_knn_local_dense_f32, dmet_bn_head_fwd_f32). */
int dmet_bn_eval_stats_f32(const float *running_mean, const float *running_var, int H, float eps, float *save_mean,
                           float *save_invstd, dmet_stream_t stream);
/* The transform half of dmet_bn_fwd_f32 with the statistics given (dmet_bn_stats_f32 / dmet_bn_eval_stats_f32):
 * y = (x - mean) * (gamma * invstd) + beta (+ residual when not NULL), the same kernel and therefore the same bits as
 * dmet_bn_fwd_f32 -- for a caller whose fused consumer (dmet_bn_knn_local_dense_f32, dmet_bn_head_fwd_f32) declined
 * after the statistics were computed (model/graph_met_network.py:58,66).  All pointers 16-byte aligned. */
int dmet_bn_apply_f32(const float *x, const float *residual, int64_t N, int H, const float *gamma, const float *beta,
                      const float *mean, const float *invstd, float *y, dmet_stream_t stream);
/* The statistics half of dmet_bn_fwd_tracked_f32 in training mode (column sums + finalize: save_mean, save_invstd,
 * running statistics, num_batches_tracked), for a caller that applies the transform elsewhere
 * (dmet_bn_knn_local_dense_f32 fuses it into the next layer's graph build). */
int dmet_bn_stats_f32(const float *x, int64_t N, int H, float eps, float momentum, float *running_mean, float *running_var,
                      int64_t *num_batches_tracked, float *save_mean, float *save_invstd, void *ws, size_t ws_bytes,
                      dmet_stream_t stream);
int dmet_bn_bwd_f32(const float *x, const float *g_y, int64_t N, int H, const float *gamma, const float *save_mean,
                    const float *save_invstd, float *g_x, float *g_gamma, float *g_beta, void *ws, size_t ws_bytes,
                    dmet_stream_t stream);

/* ---- N3 (fourth piece): the output head with its sigmoid ---------------------------------------------------
 * model/graph_met_network.py:41-44,67 + model/net.py:46:  out_i = sigmoid(W2 . ELU(W1 . emb_i + b1) + b2) with
 * emb[N,32], W1[16,32], b1[16], W2[1,16], b2[1] (torch layouts).  Backward takes the forward output and g_out[N] and
 * writes g_emb[N,32] and the four parameter gradients (reduced over the nodes deterministically). */
int dmet_head_fwd_f32(const float *emb, int64_t N, const float *W1, const float *b1, const float *W2, const float *b2,
                      float *out, dmet_stream_t stream);
/* The last block of the model, `emb = emb + bn(conv(emb))` followed by the head (model/graph_met_network.py:66-67): the
 * BatchNorm transform + residual add is formed inside the head's forward launch (expression and bits of
 * dmet_bn_fwd_f32's transform; statistics from dmet_bn_stats_f32), emb[N,32] is written for the backward.  *fused = 0:
 * nothing was launched (misaligned operands, DMET_HEAD_FWD=valu) and the caller keeps the two steps. */
int dmet_bn_head_fwd_f32(const float *raw, const float *residual, const float *gamma, const float *beta,
                         const float *mean, const float *invstd, float *emb, int64_t N, const float *W1,
                         const float *b1, const float *W2, const float *b2, float *out, int *fused,
                         dmet_stream_t stream);
size_t dmet_head_bwd_workspace_bytes(int64_t N);
int dmet_head_bwd_f32(const float *emb, int64_t N, const float *W1, const float *b1, const float *W2, const float *out,
                      const float *g_out, float *g_emb, float *gW1, float *gb1, float *gW2, float *gb2, void *ws,
                      size_t ws_bytes, dmet_stream_t stream);

/* ---- K5 / N3: the weight-gradient sums of a whole backward pass in one launch -----------------------------------
 * train.py:51-52 (`loss.backward(); optimizer.step()`): dmet_edgeconv_linear_bwd*_f32, dmet_encode_bwd_f32 /
 * dmet_encode_bn_bwd_f32 and dmet_head_bwd_f32 each end in a small second launch that adds the per-workgroup partials
 * of their parameter gradients in a fixed order; only the optimizer reads the results.  After
 * dmet_finalize_defer_begin() those calls -- from ANY thread of the process: an autograd engine runs them on a thread of
 * its own -- queue that step instead of launching it -- up to 8; a ninth
 * launches its own as before -- and dmet_finalize_flush(stream) forms every queued sum in ONE launch (the same
 * additions in the same order: same bits) and ends the deferral.  Contract: the gradient outputs and the workspaces of
 * the queued calls must stay allocated and untouched until the flush has been enqueued on the same stream; the outputs
 * hold garbage before it.  dmet_finalize_pending(): queued steps, -1 outside a deferral.  Beginning a deferral while sums
 * of an earlier one are still queued is refused (-22): they would never be formed. */
int dmet_finalize_defer_begin(void);
int dmet_finalize_pending(void);
int dmet_finalize_flush(dmet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMET_H_ */
