/*
 * dmet_oracle.c -- CPU restatement (the ORACLE) of the graph operators on the
 * DeepMETv2 DynamicEdgeConv hot path.  TEST INFRASTRUCTURE ONLY: nothing under
 * deepmetv2_amd/ may link, load or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party wheels
 * (torch_cluster / torch_scatter / torch_geometric, versions unpinned at
 * /root/reference/README.md:12-17) that are absent from /root/reference and
 * from this image, and the reference ships no tests or golden vectors
 * (SURVEY.md section 8c).  This file restates the published behaviour of those
 * operators (rules R1-R6 of SURVEY.md section 8a) and is anchored on the
 * reference's call sites:
 *   knn_graph   model/graph_met_network.py:63, model/dynamic_reduction_network.py:86,94
 *   radius_graph train.py:48, evaluate.py:88
 *   scatter_add model/net.py:55-56
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() is exact by
 * definition, so -march flags only change speed, never results).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DMET_ORACLE_MAX_K 128

/* R1: squared L2 accumulated sequentially over the feature index in fp32,
 * one fused multiply-add per feature, candidate minus query
 * (torch_cluster knn_cuda.cu inner loop, nvcc-contracted). */
static inline float sqdist_r1(const float *cand, const float *query, int D)
{
    float acc = 0.0f;
    for (int c = 0; c < D; ++c) {
        float diff = cand[c] - query[c];
        acc = fmaf(diff, diff, acc);
    }
    return acc;
}

/*
 * knn: for every query node i of every event b (nodes ptr[b]..ptr[b+1]-1) find
 * the kk nearest nodes of the same event.  R2: candidates scanned in ascending
 * index; a candidate takes slot p only if best[p] > d (strict), so ties keep the
 * lower index first; slots start at (1e10, -1) like the upstream kernel, so a
 * candidate at distance >= 1e10 is never selected and short events leave -1.
 * Output nbr[i*kk + p] = GLOBAL node index or -1; dist (optional) likewise.
 * No self-loop handling here: that is a host-side filter (see knn_graph in
 * oracle/ref_ops.py), exactly as upstream does it.
 */
int dmet_oracle_knn_f32(const float *x, const int64_t *ptr, int B, int D, int kk,
                        int32_t *nbr, float *dist)
{
    if (kk <= 0 || kk > DMET_ORACLE_MAX_K || D <= 0 || B < 0) return -22;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const int64_t lo = ptr[b], hi = ptr[b + 1];
        for (int64_t i = lo; i < hi; ++i) {
            float bd[DMET_ORACLE_MAX_K];
            int32_t bi[DMET_ORACLE_MAX_K];
            for (int e = 0; e < kk; ++e) { bd[e] = 1e10f; bi[e] = -1; }
            const float *q = x + i * (int64_t)D;
            for (int64_t j = lo; j < hi; ++j) {
                const float d = sqdist_r1(x + j * (int64_t)D, q, D);
                for (int e = 0; e < kk; ++e) {
                    if (bd[e] > d) {
                        for (int e2 = kk - 1; e2 > e; --e2) { bd[e2] = bd[e2 - 1]; bi[e2] = bi[e2 - 1]; }
                        bd[e] = d;
                        bi[e] = (int32_t)j;
                        break;
                    }
                }
            }
            memcpy(nbr + i * (int64_t)kk, bi, sizeof(int32_t) * (size_t)kk);
            if (dist) memcpy(dist + i * (int64_t)kk, bd, sizeof(float) * (size_t)kk);
        }
    }
    return 0;
}

/*
 * radius: all j of the same event with sqdist < r*r (strict), the FIRST
 * max_nbr of them in ascending index order (not the nearest), self included
 * (torch_cluster radius_cuda.cu).  The radius is squared in fp32.
 * Output nbr[i*max_nbr + p] = global index or -1; cnt[i] = number found.
 */
int dmet_oracle_radius_f32(const float *x, const int64_t *ptr, int B, int D, float r,
                           int max_nbr, int32_t *nbr, int32_t *cnt)
{
    if (max_nbr <= 0 || D <= 0 || B < 0) return -22;
    const float r2 = r * r;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const int64_t lo = ptr[b], hi = ptr[b + 1];
        for (int64_t i = lo; i < hi; ++i) {
            int c = 0;
            int32_t *row = nbr + i * (int64_t)max_nbr;
            for (int p = 0; p < max_nbr; ++p) row[p] = -1;
            for (int64_t j = lo; j < hi && c < max_nbr; ++j) {
                const float d = sqdist_r1(x + j * (int64_t)D, x + i * (int64_t)D, D);
                if (d < r2) row[c++] = (int32_t)j;
            }
            cnt[i] = c;
        }
    }
    return 0;
}

/*
 * Per-event MET sums (model/net.py:55-56): met[b] = (sum_i w_i*px_i, sum_i w_i*py_i)
 * with px = x[i*stride+0], py = x[i*stride+1].  Accumulated in fp64 so the
 * oracle is the order-free value every fp32 summation order is compared to (R6).
 */
int dmet_oracle_met_f64(const float *w, const float *x, int64_t stride, const int64_t *ptr,
                        int B, double *met)
{
    for (int b = 0; b < B; ++b) {
        double sx = 0.0, sy = 0.0;
        for (int64_t i = ptr[b]; i < ptr[b + 1]; ++i) {
            /* the product w*px is an fp32 product in the reference (weights*px is a torch fp32 mul) */
            sx += (double)(w[i] * x[i * stride + 0]);
            sy += (double)(w[i] * x[i * stride + 1]);
        }
        met[2 * b + 0] = sx;
        met[2 * b + 1] = sy;
    }
    return 0;
}

int dmet_oracle_version(void) { return 1; }
