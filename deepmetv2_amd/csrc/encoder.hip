// encoder.hip -- N3: the per-node encoder of the graph-MET model as two kernels (gfx950).
//
// /root/reference/model/graph_met_network.py:48-58 runs, per node, a chain of ~25 tiny torch kernels:
//   e_cont = ELU(Linear 8->16 (x_cont))
//   cls    = sequential remap of |pdgId| over (1,2,11,13,22,130,211) -> 0..6           (:52-54)
//   cat24  = [embed_charge[charge+1] | embed_pdgid[cls] | embed_pv[fromPV]]              (:49,:50,:55,:57)
//   e_cat  = ELU(Linear 24->16 (cat24))
//   h      = ELU(Linear 32->32 ([e_cat | e_cont]))                                       (:58, before bn_all)
// and autograd runs twice as many in backward.  Here: encode_fwd (one thread per node, weights broadcast from LDS)
// and encode_bwd (recomputes the activations, back-propagates per node, and reduces ALL parameter gradients over
// the nodes in-kernel: per 64-node chunk the per-node vectors are transposed through LDS and multiplied on the
// fp32 matrix cores, C = A^T B with K = nodes; fixed node ranges per wavefront, block partials summed in order by a
// second kernel => bitwise reproducible, no float atomics).
#include <stdlib.h>

#include "common.h"

namespace dmet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The weights are read straight from global memory with wave-uniform addresses: the compiler turns those into scalar
// loads (s_load_dwordx8/16 through the scalar cache), so a weight costs no VGPR, no LDS traffic and is an SGPR operand
// of the FMA.  (A first version staged them in LDS: the scheduler hoisted hundreds of ds_reads and spilled.)
#define ENC_PARAMS                                                                                                  \
    const float *__restrict__ Wc, const float *__restrict__ bc, const float *__restrict__ Wk,                      \
        const float *__restrict__ bk, const float *__restrict__ Wa, const float *__restrict__ ba,                  \
        const float *__restrict__ Echg, const float *__restrict__ Epdg, const float *__restrict__ Epv
#define ENC_ARGS Wc, bc, Wk, bk, Wa, ba, Echg, Epdg, Epv

__device__ __forceinline__ float elu(float z) { return z > 0.0f ? z : (__expf(z) - 1.0f); }

// categorical columns (x_cat[N,3] int64 = pdgId, charge, fromPV) -> table rows; the pdg remap is the reference's
// SEQUENTIAL torch.where chain, kept sequential so unexpected ids behave identically; indices are clamped into
// their tables (torch would raise on an out-of-range index).
// xcat == NULL: the three columns are columns 8..10 of the node's own row of x, still as floats -- the conversion of
// train.py:43 (`x[:, 8:].long()`, truncation toward zero) then happens here instead of in a kernel of its own.
__device__ __forceinline__ void cat_indices(const int64_t *__restrict__ xcat, const float *__restrict__ row, int64_t i,
                                            int &ichg, int &ipdg, int &ipv)
{
    long long c = xcat ? xcat[i * 3 + 0] : (long long)row[8];
    const long long chg = xcat ? xcat[i * 3 + 1] : (long long)row[9];
    const long long pv = xcat ? xcat[i * 3 + 2] : (long long)row[10];
    c = c < 0 ? -c : c;
    const long long table[7] = {1, 2, 11, 13, 22, 130, 211};
#pragma unroll
    for (int t = 0; t < 7; ++t) c = (c == table[t]) ? (long long)t : c;
    ipdg = (int)min(max(c, 0ll), 6ll);
    ichg = (int)min(max(chg + 1, 0ll), 2ll);
    ipv = (int)min(max(pv, 0ll), 7ll);
}

// forward chain for one node; pre-activations are returned because backward needs ELU'(z)
__device__ __forceinline__ void encode_node(ENC_PARAMS, const float (&xc)[8], int ichg, int ipdg, int ipv,
                                            float (&cat24)[24], float (&z1)[16], float (&z2)[16], float (&joint)[32],
                                            float (&z3)[32])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        cat24[e] = Echg[ichg * 8 + e];
        cat24[8 + e] = Epdg[ipdg * 8 + e];
        cat24[16 + e] = Epv[ipv * 8 + e];
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float a = bc[o];
#pragma unroll
        for (int f = 0; f < 8; ++f) a = __builtin_fmaf(Wc[o * 8 + f], xc[f], a);
        z1[o] = a;
        float b = bk[o];
#pragma unroll
        for (int f = 0; f < 24; ++f) b = __builtin_fmaf(Wk[o * 24 + f], cat24[f], b);
        z2[o] = b;
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) { joint[o] = elu(z2[o]); joint[16 + o] = elu(z1[o]); }   // [e_cat | e_cont]
#pragma unroll
    for (int o = 0; o < 32; ++o) {
        float a = ba[o];
#pragma unroll
        for (int f = 0; f < 32; ++f) a = __builtin_fmaf(Wa[o * 32 + f], joint[f], a);
        z3[o] = a;
    }
}

__global__ __launch_bounds__(256, 2) void encode_fwd_kernel(const float *__restrict__ x, int64_t x_stride,
                                                          const int64_t *__restrict__ xcat, int64_t N,
                                                          ENC_PARAMS, float *__restrict__ h)
{
    // one node per thread and NO loop: with a grid-stride loop the compiler hoists the (loop-invariant) scalar
    // weight loads out of it, runs out of SGPRs and shuttles 1.5k weights through v_writelane/v_readlane
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *row = x + i * x_stride;
    float xc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) xc[f] = row[f];
    int ichg, ipdg, ipv;
    cat_indices(xcat, row, i, ichg, ipdg, ipv);
    float cat24[24], z1[16], z2[16], joint[32], z3[32];
    encode_node(ENC_ARGS, xc, ichg, ipdg, ipv, cat24, z1, z2, joint, z3);
    float4 *o = reinterpret_cast<float4 *>(h + i * 32);
#pragma unroll
    for (int c = 0; c < 8; ++c)
        o[c] = make_float4(elu(z3[4 * c]), elu(z3[4 * c + 1]), elu(z3[4 * c + 2]), elu(z3[4 * c + 3]));
}

// ---- forward on the fp32 matrix cores (third session of round 2) ---------------------------------------------------
// encode_fwd_kernel above streams 1.5k weights per wavefront through the scalar cache: its vector ALU is 23 % busy and
// every wavefront spends half of its life in s_waitcnt (counters in DESIGN section 4 N3).  Here the weights are MFMA
// A operands held in registers for the whole launch, and the chain needs NO transposition between its layers:
//   C = W . X^T with v_mfma_f32_32x32x2_f32: A = W (row = output channel), B = X^T (column = node).
//   Lane (n = l & 31, hh = l >> 5) receives, for node n, the 16 output channels row(e) = (e & 3) + 8 (e >> 2) + 4 hh --
//   and a B operand is exactly that: lane (n, hh) supplies, at k-step s, feature f(s, hh) of node n.  With the k-steps of
//   the next product numbered like the accumulator registers, f(s, hh) = row(s), ELU(C) IS the next B operand.
// Product 1 (16 MFMAs): the block-diagonal [32 x 32] matrix [[Wk 0] [0 Wc]] on the input [cat24 | x_cont] gives
// [z2 | z1], i.e. `joint` in the reference's order (graph_met_network.py:57-58); product 2 (16 MFMAs): Wa.
// The two lanes of a node load the halves of its table rows and of its 8 continuous features that their k-steps need.
__global__ __launch_bounds__(256) void encode_fwd_mfma_kernel(const float *__restrict__ x, int64_t x_stride,
                                                               const int64_t *__restrict__ xcat, int64_t N, ENC_PARAMS,
                                                               float *__restrict__ h)
{
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t ntiles = (N + 31) / 32;
    float w12[16], w3[16], b12[16], b3[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int f = (s & 3) + 8 * (s >> 2) + 4 * hh;        // feature of k-step s in this half-wave = channel of register s
        // branch-free: one load from an address that always exists, masked afterwards (the guarded form compiled to 16
        // divergent branches, i.e. 16 memory round trips one after the other before the first tile)
        const float *src = r < 16 ? Wk + r * 24 + min(f, 23) : Wc + (r - 16) * 8 + max(f - 24, 0);
        const bool keep = r < 16 ? f < 24 : f >= 24;
        const float a = *src;
        w12[s] = keep ? a : 0.0f;
        w3[s] = Wa[r * 32 + f];
        const float *bsrc = f < 16 ? bk + f : bc + (f - 16);
        b12[s] = *bsrc;
        b3[s] = ba[f];
    }
    // two-stage pipeline over the wavefront's tiles: while tile t is multiplied, the table rows of tile t + 1 (whose
    // categorical columns arrived an iteration ago) and the raw row of tile t + 2 are in flight
    struct Raw { float c[4]; long long k0, k1, k2; };
    auto load_raw = [&](const int64_t tile, Raw &q) __attribute__((always_inline)) {
        const int64_t node = tile * 32 + r;
        const int64_t i = node < N ? node : N - 1;
        const float *row = x + i * x_stride;
#pragma unroll
        for (int u = 0; u < 4; ++u) q.c[u] = row[4 * hh + u];
        if (xcat) { q.k0 = xcat[i * 3 + 0]; q.k1 = xcat[i * 3 + 1]; q.k2 = xcat[i * 3 + 2]; }
        else { q.k0 = (long long)row[8]; q.k1 = (long long)row[9]; q.k2 = (long long)row[10]; }
    };
    auto load_tables = [&](const Raw &q, float (&in)[16]) __attribute__((always_inline)) {
        long long c = q.k0 < 0 ? -q.k0 : q.k0;    // cat_indices(): the reference's sequential remap, clamped into the tables
        const long long table[7] = {1, 2, 11, 13, 22, 130, 211};
#pragma unroll
        for (int t = 0; t < 7; ++t) c = (c == table[t]) ? (long long)t : c;
        const int ipdg = (int)min(max(c, 0ll), 6ll), ichg = (int)min(max(q.k1 + 1, 0ll), 2ll), ipv = (int)min(max(q.k2, 0ll), 7ll);
        const float4 e0 = *reinterpret_cast<const float4 *>(Echg + ichg * 8 + 4 * hh);
        const float4 e1 = *reinterpret_cast<const float4 *>(Epdg + ipdg * 8 + 4 * hh);
        const float4 e2 = *reinterpret_cast<const float4 *>(Epv + ipv * 8 + 4 * hh);
        in[0] = e0.x; in[1] = e0.y; in[2] = e0.z; in[3] = e0.w; in[4] = e1.x; in[5] = e1.y; in[6] = e1.z; in[7] = e1.w;
        in[8] = e2.x; in[9] = e2.y; in[10] = e2.z; in[11] = e2.w;
#pragma unroll
        for (int u = 0; u < 4; ++u) in[12 + u] = q.c[u];
    };
    Raw raw;
    float in[16], nxt[16];
    if (wave < ntiles) { load_raw(wave, raw); load_tables(raw, in); }
    if (wave + nwaves < ntiles) load_raw(wave + nwaves, raw);
    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t node = tile * 32 + r;
        if (tile + nwaves < ntiles) load_tables(raw, nxt);
        if (tile + 2 * nwaves < ntiles) load_raw(tile + 2 * nwaves, raw);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = b12[e];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w12[s], in[s], acc, 0, 0, 0);
        float joint[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) joint[e] = elu(acc[e]);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = b3[e];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w3[s], joint[s], acc, 0, 0, 0);
        if (node < N) {
            float4 *o = reinterpret_cast<float4 *>(h + node * 32 + 4 * hh);
#pragma unroll
            for (int g = 0; g < 4; ++g)     // channels 8 g + 4 hh .. + 3
                o[2 * g] = make_float4(elu(acc[4 * g]), elu(acc[4 * g + 1]), elu(acc[4 * g + 2]), elu(acc[4 * g + 3]));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) in[u] = nxt[u];
    }
}

// ---- backward -------------------------------------------------------------------------------------------------
// Gradient tiles (each 32x32, C = A^T B over nodes, on the fp32 matrix cores):
//   T0 = g_z3^T . joint                       -> dWa
//   T1 = [g_z2 | g_z1]^T . [cat24 | x_cont]   -> dWk = T1[0:16, 0:24], dWc = T1[16:32, 24:32]
//   T2 = S^T . [g_cat24 | 0]                  -> dEchg = T2[0:3, 0:8], dEpdg = T2[3:10, 8:16], dEpv = T2[10:18, 16:24]
// with S = [onehot(chg) (3) | onehot(pdg) (7) | onehot(pv) (8) | 0...] per node, generated on the fly from the
// packed table rows; bias gradients are column sums of the staged A tiles (lane (c, half) sums 32 nodes of column c).
// ELU'(z) is taken from the activation itself (1 if a > 0 else a + 1), so z3 = Wa.joint is never recomputed.
constexpr int kEncTiles = 3;
constexpr int kEncBwdWaves = 4;
constexpr int kEncPartial = kEncTiles * 1024 + 64;   // floats per workgroup partial: 3 tiles + [dba | dbk | dbc]

__device__ __forceinline__ void enc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// C += A^T B for a 64-node chunk held in LDS as A[64][33], B[64][33] (padded rows: conflict-free column reads)
__device__ __forceinline__ void tile_mma(f32x16 &acc, const float *__restrict__ A, const float *__restrict__ Bm, int lane)
{
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        const int node = 2 * s + hh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[node * 33 + c], Bm[node * 33 + c], acc, 0, 0, 0);
    }
}

// C += S^T B with S[node][c] = (field of packed[node] selected by c) == target(c)
__device__ __forceinline__ void tile_mma_onehot(f32x16 &acc, const int *__restrict__ packed, const float *__restrict__ Bm,
                                                int lane)
{
    const int c = lane & 31, hh = lane >> 5;
    int sh, mask, tgt;
    if (c < 3) { sh = 0; mask = 3; tgt = c; }
    else if (c < 10) { sh = 2; mask = 7; tgt = c - 3; }
    else if (c < 18) { sh = 5; mask = 7; tgt = c - 10; }
    else { sh = 0; mask = 0; tgt = 1; }
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        const int node = 2 * s + hh;
        const float a = (((packed[node] >> sh) & mask) == tgt) ? 1.0f : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bm[node * 33 + c], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ float column_half_sum(const float *__restrict__ A, int lane)
{
    const int c = lane & 31, hh = lane >> 5;
    float s = 0.0f;
#pragma unroll 8
    for (int n = 0; n < 32; ++n) s += A[(hh * 32 + n) * 33 + c];
    return s;
}

__global__ __launch_bounds__(64 * kEncBwdWaves, 2) void encode_bwd_kernel(const float *__restrict__ x, int64_t x_stride,
                                                                          const int64_t *__restrict__ xcat, int64_t N,
                                                                          ENC_PARAMS, const float *__restrict__ gh,
                                                                          const float *__restrict__ hout,
                                                                          int64_t nodes_per_wave,
                                                                          float *__restrict__ partial)
{
    __shared__ float bufA[kEncBwdWaves][64 * 33];
    __shared__ float bufB[kEncBwdWaves][64 * 33];
    __shared__ int bufI[kEncBwdWaves][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *A = bufA[wv], *Bm = bufB[wv];
    int *packed = bufI[wv];
    const int64_t wave = (int64_t)blockIdx.x * kEncBwdWaves + wv;
    const int64_t lo = wave * nodes_per_wave, hi = min(N, lo + nodes_per_wave);
    f32x16 acc[kEncTiles];
#pragma unroll
    for (int t = 0; t < kEncTiles; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    float bias3 = 0.0f, bias21 = 0.0f;

    for (int64_t base = lo; base < hi; base += 64) {
        // an opaque (always zero) scalar offset makes the weight addresses loop-variant: the loads stay scalar
        // (uniform address off a __restrict__ base) but cannot be hoisted out of this loop, where 2.5k weights
        // would overflow the SGPR file and be shuttled through v_writelane/v_readlane
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const float *__restrict__ wc = Wc + zero, *__restrict__ wk = Wk + zero, *__restrict__ wa = Wa + zero,
                                  *__restrict__ pbc = bc + zero, *__restrict__ pbk = bk + zero;
        const int64_t i = base + lane;
        const bool live = i < hi;
        const int64_t ii = live ? i : (hi - 1);
        const float *row = x + ii * x_stride;
        float xc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) xc[f] = row[f];
        int ichg, ipdg, ipv;
        cat_indices(xcat, row, ii, ichg, ipdg, ipv);
        packed[lane] = ichg | (ipdg << 2) | (ipv << 5);
        float cat24[24], joint[32];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cat24[e] = Echg[ichg * 8 + e];
            cat24[8 + e] = Epdg[ipdg * 8 + e];
            cat24[16 + e] = Epv[ipv * 8 + e];
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float a = pbc[o];
#pragma unroll
            for (int f = 0; f < 8; ++f) a = __builtin_fmaf(wc[o * 8 + f], xc[f], a);
            joint[16 + o] = elu(a);
            float b = pbk[o];
#pragma unroll
            for (int f = 0; f < 24; ++f) b = __builtin_fmaf(wk[o * 24 + f], cat24[f], b);
            joint[o] = elu(b);
        }
        // g_z3 = g_h * ELU'(z3), ELU' from h itself; stage A = g_z3, B = joint
        float gz[32];
        {
            const float4 *gp = reinterpret_cast<const float4 *>(gh + ii * 32);
            const float4 *hp = reinterpret_cast<const float4 *>(hout + ii * 32);
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                const float4 gv = gp[c4], hv = hp[c4];
                const float gg[4] = {gv.x, gv.y, gv.z, gv.w}, hh4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = hh4[u] > 0.0f ? 1.0f : (hh4[u] + 1.0f);
                    gz[4 * c4 + u] = live ? gg[u] * d : 0.0f;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) { A[lane * 33 + c] = gz[c]; Bm[lane * 33 + c] = joint[c]; }
        enc_wave_sync();
        tile_mma(acc[0], A, Bm, lane);           // g_z3^T joint
        bias3 += column_half_sum(A, lane);
        // back through encode_all: g_joint = Wa^T g_z3 (rows of Wa read sequentially), then the two ELUs
        float gj[32];
#pragma unroll
        for (int f = 0; f < 32; ++f) gj[f] = 0.0f;
#pragma unroll
        for (int o = 0; o < 32; ++o)
#pragma unroll
            for (int f = 0; f < 32; ++f) gj[f] = __builtin_fmaf(wa[o * 32 + f], gz[o], gj[f]);
#pragma unroll
        for (int f = 0; f < 32; ++f) gj[f] *= joint[f] > 0.0f ? 1.0f : (joint[f] + 1.0f);   // [g_z2 | g_z1]
        enc_wave_sync();
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            A[lane * 33 + c] = gj[c];
            Bm[lane * 33 + c] = (c < 24) ? cat24[c] : xc[c - 24];
        }
        enc_wave_sync();
        tile_mma(acc[1], A, Bm, lane);           // [g_z2|g_z1]^T [cat24|x_cont]
        bias21 += column_half_sum(A, lane);
        // g_cat24 = Wk^T g_z2
        float gc[24];
#pragma unroll
        for (int f = 0; f < 24; ++f) gc[f] = 0.0f;
#pragma unroll
        for (int o = 0; o < 16; ++o)
#pragma unroll
            for (int f = 0; f < 24; ++f) gc[f] = __builtin_fmaf(wk[o * 24 + f], gj[o], gc[f]);
        enc_wave_sync();
#pragma unroll
        for (int c = 0; c < 32; ++c) Bm[lane * 33 + c] = (c < 24) ? gc[c < 24 ? c : 0] : 0.0f;
        enc_wave_sync();
        tile_mma_onehot(acc[2], packed, Bm, lane);   // S^T [g_cat24|0]
        enc_wave_sync();
    }
    // one partial per WORKGROUP (3 tiles [32][32] then [dba(32) | dbk(16) dbc(16)]): the wavefronts' tiles meet in their
    // own LDS buffers (consumed by now) and are added in wavefront order
    const int c = lane & 31, hh = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
        A[r * 32 + c] = acc[0][e];
        A[1024 + r * 32 + c] = acc[2][e];
        Bm[r * 32 + c] = acc[1][e];
    }
    const float b3 = bias3 + __shfl_xor(bias3, 32), b21 = bias21 + __shfl_xor(bias21, 32);
    if (hh == 0) { Bm[1024 + c] = b3; Bm[1024 + 32 + c] = b21; }
    __syncthreads();
    float *out = partial + (int64_t)blockIdx.x * kEncPartial;
    for (int i = threadIdx.x; i < kEncPartial; i += 64 * kEncBwdWaves) {
        // element i of the partial: tile 0 -> A[0..], tile 1 -> Bm[0..], tile 2 -> A[1024..], tail -> Bm[1024..]
        const float *src = i < 1024 ? &bufA[0][i] : i < 2048 ? &bufB[0][i - 1024] : i < 3072 ? &bufA[0][i - 1024]
                                                                                              : &bufB[0][i - 2048];
        float t = src[0];
#pragma unroll
        for (int w = 1; w < kEncBwdWaves; ++w) t += src[w * 64 * 33];
        out[i] = t;
    }
}

// ---- backward with the per-node chain on the fp32 matrix cores (third session of round 2) --------------------------
// Same outputs and partial layout as encode_bwd_kernel; the recomputed activations and the two transposed products
// (g_joint = Wa^T g_z3, g_in = [[Wk 0] [0 Wc]]^T g_z12) run as MFMA chains in the accumulator layout of
// encode_fwd_mfma_kernel (lane (n, hh) holds the 16 channels (e & 3) + 8 (e >> 2) + 4 hh of node n; the weights are A
// operands in registers), 32 nodes per step; the K = nodes weight-gradient tiles still go through LDS (the operands of
// those products are indexed the other way round).  No scalar weight stream.
// Ordering of a wavefront's own LDS traffic WITHOUT draining its global loads: enc_wave_sync()'s wavefront-scope fences
// compile to s_waitcnt vmcnt(0) lgkmcnt(0), so every staging step waited for the prefetched rows of the next tiles.
// DS instructions of one wavefront complete in order; waiting for the LDS counter alone is enough.
__device__ __forceinline__ void enc_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void tile_mma32(f32x16 &acc, const float *__restrict__ A, const float *__restrict__ Bm, int lane)
{
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
        const int node = 2 * s + hh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[node * 33 + c], Bm[node * 33 + c], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void tile_mma_onehot32(f32x16 &acc, const int *__restrict__ packed, const float *__restrict__ Bm,
                                                  int lane)
{
    const int c = lane & 31, hh = lane >> 5;
    int sh, mask, tgt;
    if (c < 3) { sh = 0; mask = 3; tgt = c; }
    else if (c < 10) { sh = 2; mask = 7; tgt = c - 3; }
    else if (c < 18) { sh = 5; mask = 7; tgt = c - 10; }
    else { sh = 0; mask = 0; tgt = 1; }
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
        const int node = 2 * s + hh;
        const float a = (((packed[node] >> sh) & mask) == tgt) ? 1.0f : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bm[node * 33 + c], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ float column_half_sum32(const float *__restrict__ A, int lane)
{
    const int c = lane & 31, hh = lane >> 5;
    float s = 0.0f;
#pragma unroll 8
    for (int n = 0; n < 16; ++n) s += A[(hh * 16 + n) * 33 + c];
    return s;
}

// BNB (dmet_encode_bn_bwd_f32): `gh` is the gradient with respect to bn_all's OUTPUT; the BatchNorm's backward
// transform  g = gamma * invstd * (g_y - mean_g - (h - mean) * invstd * mean_gx)  (expression of bn_bwd_apply_kernel) is
// applied to the values as they are loaded -- h, the BatchNorm's input, is the encoder's own output, which this kernel
// reads anyway: the transform pass (17 us, 111 MB) disappears.
struct EncBnBwd {
    const float *gamma, *mean, *invstd, *mean_g, *mean_gx;
};

template <bool BNB = false>
__global__ __launch_bounds__(64 * kEncBwdWaves, 2) void encode_bwd_mfma_kernel(const float *__restrict__ x, int64_t x_stride,
                                                                               const int64_t *__restrict__ xcat, int64_t N,
                                                                               ENC_PARAMS, const float *__restrict__ gh,
                                                                               const float *__restrict__ hout,
                                                                               int64_t /*nodes_per_wave: tiles are dealt round-robin*/,
                                                                               float *__restrict__ partial,
                                                                               EncBnBwd bnb = EncBnBwd{})
{
    __shared__ float bufA[kEncBwdWaves][64 * 33];
    __shared__ float bufB[kEncBwdWaves][64 * 33];
    __shared__ int bufI[kEncBwdWaves][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    float *A = bufA[wv], *Bm = bufB[wv];
    int *packed = bufI[wv];
    // 32-node tiles dealt round-robin over the wavefronts (contiguous ranges of whole 64-node chunks left half of the
    // SIMDs with two wavefronts and half with one at 288 000 nodes)
    const int64_t wave = (int64_t)blockIdx.x * kEncBwdWaves + wv, nwaves = (int64_t)gridDim.x * kEncBwdWaves;
    const int64_t ntiles = (N + 31) / 32, hi = N;
    // A operands of the three chain products (branch-free loads, see encode_fwd_mfma_kernel) and the first bias
    float w12[16], w3t[16], w12t[16], b12[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int f = (s & 3) + 8 * (s >> 2) + 4 * hh;
        {   // W12[r][f]
            const float *src = r < 16 ? Wk + r * 24 + min(f, 23) : Wc + (r - 16) * 8 + max(f - 24, 0);
            const bool keep = r < 16 ? f < 24 : f >= 24;
            const float a = *src;
            w12[s] = keep ? a : 0.0f;
        }
        w3t[s] = Wa[f * 32 + r];                                   // Wa^T[r][f]
        {   // W12^T[r][f] = W12[f][r]
            const float *src = f < 16 ? Wk + f * 24 + min(r, 23) : Wc + (f - 16) * 8 + max(r - 24, 0);
            const bool keep = f < 16 ? r < 24 : r >= 24;
            const float a = *src;
            w12t[s] = keep ? a : 0.0f;
        }
        const float *bsrc = f < 16 ? bk + f : bc + (f - 16);
        b12[s] = *bsrc;
    }
    f32x16 acc[kEncTiles];
#pragma unroll
    for (int t = 0; t < kEncTiles; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    float bias3 = 0.0f, bias21 = 0.0f;

    // two-stage prefetch (as in encode_fwd_mfma_kernel): the table rows of tile t + 1 and the raw row of tile t + 2 are in
    // flight while tile t is processed
    struct Raw { float c[4]; long long k0, k1, k2; };
    auto load_raw = [&](const int64_t tile, Raw &q) __attribute__((always_inline)) {
        const int64_t nd = tile * 32 + r;
        const int64_t i = nd < N ? nd : N - 1;
        const float *row = x + i * x_stride;
#pragma unroll
        for (int u = 0; u < 4; ++u) q.c[u] = row[4 * hh + u];
        if (xcat) { q.k0 = xcat[i * 3 + 0]; q.k1 = xcat[i * 3 + 1]; q.k2 = xcat[i * 3 + 2]; }
        else { q.k0 = (long long)row[8]; q.k1 = (long long)row[9]; q.k2 = (long long)row[10]; }
    };
    auto load_tables = [&](const Raw &q, float (&in)[16], int &pk) __attribute__((always_inline)) {
        long long c = q.k0 < 0 ? -q.k0 : q.k0;    // cat_indices(): the reference's sequential remap, clamped into the tables
        const long long table[7] = {1, 2, 11, 13, 22, 130, 211};
#pragma unroll
        for (int t = 0; t < 7; ++t) c = (c == table[t]) ? (long long)t : c;
        const int ipdg = (int)min(max(c, 0ll), 6ll), ichg = (int)min(max(q.k1 + 1, 0ll), 2ll), ipv = (int)min(max(q.k2, 0ll), 7ll);
        pk = ichg | (ipdg << 2) | (ipv << 5);
        const float4 e0 = *reinterpret_cast<const float4 *>(Echg + ichg * 8 + 4 * hh);
        const float4 e1 = *reinterpret_cast<const float4 *>(Epdg + ipdg * 8 + 4 * hh);
        const float4 e2 = *reinterpret_cast<const float4 *>(Epv + ipv * 8 + 4 * hh);
        in[0] = e0.x; in[1] = e0.y; in[2] = e0.z; in[3] = e0.w; in[4] = e1.x; in[5] = e1.y; in[6] = e1.z; in[7] = e1.w;
        in[8] = e2.x; in[9] = e2.y; in[10] = e2.z; in[11] = e2.w;
#pragma unroll
        for (int u = 0; u < 4; ++u) in[12 + u] = q.c[u];
    };
    Raw raw;
    float in[16], nxt[16];
    int pk = 0, pk_nxt = 0;
    if (wave < ntiles) { load_raw(wave, raw); load_tables(raw, in, pk); }
    if (wave + nwaves < ntiles) load_raw(wave + nwaves, raw);
    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t node = tile * 32 + r;
        const bool live = node < hi;
        const int64_t ii = live ? node : (hi - 1);
        if (tile + nwaves < ntiles) load_tables(raw, nxt, pk_nxt);
        if (tile + 2 * nwaves < ntiles) load_raw(tile + 2 * nwaves, raw);
        if (hh == 0) packed[r] = pk;
        // g_z3 = g_h * ELU'(z3), ELU' from h itself: this lane's 16 channels of the node
        float gz[16];
        {
            const float4 *gp = reinterpret_cast<const float4 *>(gh + ii * 32 + 4 * hh);
            const float4 *hp = reinterpret_cast<const float4 *>(hout + ii * 32 + 4 * hh);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gv = gp[2 * g], hv = hp[2 * g];
                float gg[4] = {gv.x, gv.y, gv.z, gv.w};
                const float hh4[4] = {hv.x, hv.y, hv.z, hv.w};
                if constexpr (BNB) {
                    const int c0 = 8 * g + 4 * hh;      // this lane's channels of group g
                    const float4 ga = *reinterpret_cast<const float4 *>(bnb.gamma + c0), mu = *reinterpret_cast<const float4 *>(bnb.mean + c0);
                    const float4 is = *reinterpret_cast<const float4 *>(bnb.invstd + c0), mg = *reinterpret_cast<const float4 *>(bnb.mean_g + c0);
                    const float4 mx = *reinterpret_cast<const float4 *>(bnb.mean_gx + c0);
                    gg[0] = ga.x * is.x * (gg[0] - mg.x - (hh4[0] - mu.x) * is.x * mx.x);
                    gg[1] = ga.y * is.y * (gg[1] - mg.y - (hh4[1] - mu.y) * is.y * mx.y);
                    gg[2] = ga.z * is.z * (gg[2] - mg.z - (hh4[2] - mu.z) * is.z * mx.z);
                    gg[3] = ga.w * is.w * (gg[3] - mg.w - (hh4[3] - mu.w) * is.w * mx.w);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = hh4[u] > 0.0f ? 1.0f : (hh4[u] + 1.0f);
                    gz[4 * g + u] = live ? gg[u] * d : 0.0f;
                }
            }
        }
        // product 1: [z2 | z1] -> joint
        f32x16 c1;
#pragma unroll
        for (int e = 0; e < 16; ++e) c1[e] = b12[e];
#pragma unroll
        for (int s = 0; s < 16; ++s) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w12[s], in[s], c1, 0, 0, 0);
        float joint[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) joint[e] = elu(c1[e]);
        // tile 0: g_z3^T joint
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ch = (e & 3) + 8 * (e >> 2) + 4 * hh;
            A[r * 33 + ch] = gz[e];
            Bm[r * 33 + ch] = joint[e];
        }
        enc_lds_sync();
        tile_mma32(acc[0], A, Bm, lane);
        bias3 += column_half_sum32(A, lane);
        // product 2: g_joint = Wa^T g_z3, then the two ELUs -> [g_z2 | g_z1]
        f32x16 c2;
#pragma unroll
        for (int e = 0; e < 16; ++e) c2[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3t[s], gz[s], c2, 0, 0, 0);
        float gj[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) gj[e] = c2[e] * (joint[e] > 0.0f ? 1.0f : (joint[e] + 1.0f));
        enc_lds_sync();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ch = (e & 3) + 8 * (e >> 2) + 4 * hh;
            A[r * 33 + ch] = gj[e];
            Bm[r * 33 + ch] = in[e];
        }
        enc_lds_sync();
        tile_mma32(acc[1], A, Bm, lane);           // [g_z2|g_z1]^T [cat24|x_cont]
        bias21 += column_half_sum32(A, lane);
        // product 3: g_in = W12^T [g_z2 | g_z1]; its first 24 features are g_cat24
        f32x16 c3;
#pragma unroll
        for (int e = 0; e < 16; ++e) c3[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w12t[s], gj[s], c3, 0, 0, 0);
        enc_lds_sync();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ch = (e & 3) + 8 * (e >> 2) + 4 * hh;
            Bm[r * 33 + ch] = ch < 24 ? c3[e] : 0.0f;
        }
        enc_lds_sync();
        tile_mma_onehot32(acc[2], packed, Bm, lane);   // S^T [g_cat24|0]
        enc_lds_sync();
#pragma unroll
        for (int u = 0; u < 16; ++u) in[u] = nxt[u];
        pk = pk_nxt;
    }
    // one partial per WORKGROUP, as in encode_bwd_kernel
    const int c = lane & 31;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rr = (e & 3) + 8 * (e >> 2) + 4 * hh;
        A[rr * 32 + c] = acc[0][e];
        A[1024 + rr * 32 + c] = acc[2][e];
        Bm[rr * 32 + c] = acc[1][e];
    }
    const float b3 = bias3 + __shfl_xor(bias3, 32), b21 = bias21 + __shfl_xor(bias21, 32);
    if (hh == 0) { Bm[1024 + c] = b3; Bm[1024 + 32 + c] = b21; }
    __syncthreads();
    float *out = partial + (int64_t)blockIdx.x * kEncPartial;
    for (int i = threadIdx.x; i < kEncPartial; i += 64 * kEncBwdWaves) {
        const float *src = i < 1024 ? &bufA[0][i] : i < 2048 ? &bufB[0][i - 1024] : i < 3072 ? &bufA[0][i - 1024]
                                                                                              : &bufB[0][i - 2048];
        float t = src[0];
#pragma unroll
        for (int w = 1; w < kEncBwdWaves; ++w) t += src[w * 64 * 33];
        out[i] = t;
    }
}

struct EncGrads {
    float *Wc, *bc, *Wk, *bk, *Wa, *ba, *Echg, *Epdg, *Epv;
};

// Sum the wavefront partials in a fixed order and route each element to its parameter gradient.  A block owns 32
// consecutive elements; its 32 thread groups each sum every 32nd partial, then group 0 adds the 32 group sums in order.
__global__ __launch_bounds__(1024) void encode_bwd_finalize_kernel(const float *__restrict__ partial, int64_t nwaves,
                                                                    EncGrads g)
{
    __shared__ float red[32][33];
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    float s = 0.0f;
#pragma unroll 8   // independent loads: keep eight in flight (the sum order is unchanged)
    for (int64_t w = grp; w < nwaves; w += 32) s += partial[w * kEncPartial + idx];
    red[grp][e] = s;
    __syncthreads();
    if (grp != 0) return;
    s = 0.0f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red[q][e];
    const int t = idx / 1024, r = (idx % 1024) / 32, c = idx % 32;
    if (t == 0) g.Wa[r * 32 + c] = s;
    else if (t == 1) {
        if (r < 16 && c < 24) g.Wk[r * 24 + c] = s;
        else if (r >= 16 && c >= 24) g.Wc[(r - 16) * 8 + (c - 24)] = s;
    } else if (t == 2) {
        if (r < 3 && c < 8) g.Echg[r * 8 + c] = s;
        else if (r >= 3 && r < 10 && c >= 8 && c < 16) g.Epdg[(r - 3) * 8 + (c - 8)] = s;
        else if (r >= 10 && r < 18 && c >= 16 && c < 24) g.Epv[(r - 10) * 8 + (c - 16)] = s;
    } else if (r == 0) g.ba[c] = s;
    else if (c < 16) g.bk[c] = s;
    else g.bc[c - 16] = s;
}

inline int64_t enc_nodes_per_wave(int64_t N, int64_t *nwaves)
{
    const int64_t target_waves = 2048;   // 256 CUs x 8 resident wavefronts
    int64_t npw = (N + target_waves - 1) / target_waves;
    // rounded DOWN to whole 64-node chunks: at least target_waves wavefronts' worth of partials, so that the matrix-core
    // kernel (32-node tiles dealt round-robin, grid capped at 512 workgroups AND at the partial count) gets its full grid --
    // rounded up, 288 000 nodes gave 375 workgroups: 6 tiles per wavefront with half of the SIMDs holding two wavefronts
    // (12 tiles) against <= 5 per wavefront, two wavefronts on every SIMD
#ifdef DMET_ENC_NPW_UP
    npw = (npw + 63) / 64 * 64;
#else
    npw = npw / 64 * 64;
#endif
    if (npw < 64) npw = 64;
    int64_t nw = (N + npw - 1) / npw;
    nw = (nw + kEncBwdWaves - 1) / kEncBwdWaves * kEncBwdWaves;
    *nwaves = nw;
    return npw;
}

// set by dmet_encode_bn_bwd_f32 around its call of dmet_encode_bwd_f32 on this thread
thread_local EncBnBwd g_enc_bnb{nullptr, nullptr, nullptr, nullptr, nullptr};
thread_local bool g_enc_bnb_done = false;

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" int dmet_encode_fwd_f32(const float *x, int64_t x_stride, const int64_t *xcat, int64_t N, const float *Wc, const float *bc,
                                   const float *Wk, const float *bk, const float *Wa, const float *ba,
                                   const float *Echg, const float *Epdg, const float *Epv, float *h,
                                   dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && x_stride >= 8, "dmet_encode_fwd_f32: bad sizes");
    if (N == 0) return 0;
    DMET_REQUIRE(x && Wc && bc && Wk && bk && Wa && ba && Echg && Epdg && Epv && h, "dmet_encode_fwd_f32: null pointer");
    DMET_REQUIRE(xcat || x_stride >= 11, "dmet_encode_fwd_f32: x_cat == NULL needs rows of at least 11 columns (x_stride=%lld)", (long long)x_stride);
    DMET_REQUIRE(aligned16(h), "dmet_encode_fwd_f32: h must be 16-B aligned");
    const int64_t blocks = (N + 255) / 256;
    DMET_REQUIRE(blocks < (1ll << 31), "dmet_encode_fwd_f32: too many nodes");
    const int form = env_is("DMET_ENCODER_FWD", "valu") ? 0 : 1;    // valu: the scalar-weight kernel (experiments, A/B)
    if (form == 1 && aligned16(Echg) && aligned16(Epdg) && aligned16(Epv)) {
        const int64_t tiles = (N + 31) / 32;
        int64_t grid = (tiles + 3) / 4;
        int gmax = 512;
        if (const char *e = getenv("DMET_ENC_GRID")) { gmax = atoi(e); if (gmax < 1) gmax = 512; }
        if (grid > gmax) grid = gmax;       // 2048 wavefronts (three per SIMD fit), 4-5 tiles each at 288 000 nodes: the weights are loaded once per wavefront
        hipLaunchKernelGGL(encode_fwd_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, x_stride, xcat, N,
                           ENC_ARGS, h);
        DMET_LAUNCH_CHECK("encode_fwd_mfma_kernel");
        return 0;
    }
    hipLaunchKernelGGL(encode_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, x_stride, xcat, N, ENC_ARGS, h);
    DMET_LAUNCH_CHECK("encode_fwd_kernel");
    return 0;
}

extern "C" size_t dmet_encode_bwd_workspace_bytes(int64_t N)
{
    if (N <= 0) return 0;
    int64_t nw;
    (void)enc_nodes_per_wave(N, &nw);
    return sizeof(float) * (size_t)nw * kEncPartial + 512;
}

extern "C" int dmet_encode_bwd_f32(const float *x, int64_t x_stride, const int64_t *xcat, int64_t N, const float *Wc, const float *bc,
                                   const float *Wk, const float *bk, const float *Wa, const float *ba,
                                   const float *Echg, const float *Epdg, const float *Epv, const float *h,
                                   const float *g_h, float *gWc, float *gbc, float *gWk, float *gbk, float *gWa,
                                   float *gba, float *gEchg, float *gEpdg, float *gEpv, void *ws, size_t ws_bytes,
                                   dmet_stream_t stream)
{
    DMET_REQUIRE(N > 0 && x_stride >= 8, "dmet_encode_bwd_f32: bad sizes");
    DMET_REQUIRE((xcat || x_stride >= 11) && x && Wc && bc && Wk && bk && Wa && ba && Echg && Epdg && Epv && h && g_h && gWc && gbc && gWk && gbk &&
                     gWa && gba && gEchg && gEpdg && gEpv && ws,
                 "dmet_encode_bwd_f32: null pointer");
    DMET_REQUIRE(ws_bytes >= dmet_encode_bwd_workspace_bytes(N), "dmet_encode_bwd_f32: workspace too small");
    DMET_REQUIRE(aligned16(h) && aligned16(g_h), "dmet_encode_bwd_f32: h / g_h must be 16-B aligned");
    int64_t nw;
    const int64_t npw = enc_nodes_per_wave(N, &nw);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    hipStream_t st = as_stream(stream);
    const int form = env_is("DMET_ENCODER_BWD", "valu") ? 0 : 1;    // valu: the scalar-weight kernel (experiments, A/B)
    int64_t nparts = nw / kEncBwdWaves;    // workgroup partials the finalize kernel sums
    if (form == 1 && aligned16(Echg) && aligned16(Epdg) && aligned16(Epv) && aligned16(g_h) && aligned16(h)) {
        // 2048 wavefronts = two per SIMD everywhere, 32-node tiles dealt round-robin; never more workgroups than the
        // workspace holds partials for
        const int64_t tiles = (N + 31) / 32;
        int64_t grid = (tiles + kEncBwdWaves - 1) / kEncBwdWaves;
        if (grid > 512) grid = 512;
        if (grid > nparts) grid = nparts;
        nparts = grid;
        if (g_enc_bnb.gamma)
            hipLaunchKernelGGL(encode_bwd_mfma_kernel<true>, dim3((unsigned)grid), dim3(64 * kEncBwdWaves), 0, st, x, x_stride, xcat,
                               N, ENC_ARGS, g_h, h, npw, partial, g_enc_bnb);
        else
            hipLaunchKernelGGL(encode_bwd_mfma_kernel<false>, dim3((unsigned)grid), dim3(64 * kEncBwdWaves), 0, st, x, x_stride,
                               xcat, N, ENC_ARGS, g_h, h, npw, partial, EncBnBwd{});
        g_enc_bnb_done = g_enc_bnb.gamma != nullptr;
    } else {
        hipLaunchKernelGGL(encode_bwd_kernel, dim3((unsigned)(nw / kEncBwdWaves)), dim3(64 * kEncBwdWaves), 0, st, x, x_stride,
                           xcat, N, ENC_ARGS, g_h, h, npw, partial);
    }
    DMET_LAUNCH_CHECK("encode_bwd_kernel");
    EncGrads gr{gWc, gbc, gWk, gbk, gWa, gba, gEchg, gEpdg, gEpv};
    static_assert(kEncPartial % 32 == 0, "finalize blocks own 32 elements");
    static_assert(kEncPartial == kEncPartialFloats, "csrc/finalize.hip sums the same partial layout");
    if (defer_push(DeferDesc{kDeferEncoder, partial, nparts, {gWc, gbc, gWk, gbk, gWa, gba, gEchg, gEpdg, gEpv}})) return 0;
    hipLaunchKernelGGL(encode_bwd_finalize_kernel, dim3(kEncPartial / 32), dim3(1024), 0, st, partial, nparts, gr);
    DMET_LAUNCH_CHECK("encode_bwd_finalize_kernel");
    return 0;
}

extern "C" int dmet_encode_bn_bwd_f32(const float *x, int64_t x_stride, const int64_t *xcat, int64_t N, const float *Wc,
                                      const float *bc, const float *Wk, const float *bk, const float *Wa, const float *ba,
                                      const float *Echg, const float *Epdg, const float *Epv, const float *h,
                                      const float *g_y, const float *bn_gamma, const float *bn_mean, const float *bn_invstd,
                                      const float *bn_mean_g, const float *bn_mean_gx, float *gWc, float *gbc, float *gWk,
                                      float *gbk, float *gWa, float *gba, float *gEchg, float *gEpdg, float *gEpv,
                                      int *fused, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(fused, "dmet_encode_bn_bwd_f32: fused is null");
    *fused = 0;
    DMET_REQUIRE(bn_gamma && bn_mean && bn_invstd && bn_mean_g && bn_mean_gx, "dmet_encode_bn_bwd_f32: null pointer");
    const bool ok = !env_is("DMET_ENCODER_BWD", "valu") && N > 0 && aligned16(Echg) && aligned16(Epdg) && aligned16(Epv) && aligned16(g_y) &&
                    aligned16(h) && aligned16(bn_gamma) && aligned16(bn_mean) && aligned16(bn_invstd) && aligned16(bn_mean_g) &&
                    aligned16(bn_mean_gx);
    if (!ok) return 0;     // nothing launched: the caller applies the BatchNorm's backward transform and calls dmet_encode_bwd_f32
    g_enc_bnb = EncBnBwd{bn_gamma, bn_mean, bn_invstd, bn_mean_g, bn_mean_gx};
    g_enc_bnb_done = false;
    const int rc = dmet_encode_bwd_f32(x, x_stride, xcat, N, Wc, bc, Wk, bk, Wa, ba, Echg, Epdg, Epv, h, g_y, gWc, gbc, gWk, gbk, gWa,
                                       gba, gEchg, gEpdg, gEpv, ws, ws_bytes, stream);
    const bool done = g_enc_bnb_done;
    g_enc_bnb = EncBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr};
    g_enc_bnb_done = false;
    if (rc == 0 && !done) {
        set_error("dmet_encode_bn_bwd_f32: the matrix-core backward it was checked for did not run");
        return -22;
    }
    *fused = rc == 0 ? 1 : 0;
    return rc;
}
