// edgeconv_bwd.hip -- K5: node-level backward of the fused EdgeConv (Linear 2H -> H with max), H = 32, gfx950.
//
// After gather_max_bwd_kernel has turned the upstream gradient into gQ (the gradient of the gathered table
// Q = x.W2^T), what is left of the backward of /root/reference/model/graph_met_network.py:35-37
// (EdgeConv(nn=Sequential(Linear(2H, H)))) in the split form  out_i = P_i + max_s Q[nbr[i,s]],
// P = x.(W1-W2)^T + b,  is node-level dense algebra with gP = g_out (masked where the node had no neighbour):
//     gx = gP.(W1-W2) + gQ.W2                      [N,H]
//     gW = [ gP^T x  |  gQ^T x - gP^T x ]          [H,2H]   (torch Linear.weight layout)
//     gb = sum_i gP_i                               [H]
// Stock torch spends ~330 us per layer on it (two library GEMMs, two tall-skinny reductions, mask/copy/cat/sum
// elementwise kernels).  Here it is ONE pass over the rows: every wavefront walks 32-node chunks, stages gP, gQ and x
// of the chunk in LDS with coalesced loads, and runs fp32 MFMAs (32x32x2): gx of the chunk (K = 2H) goes straight
// back to memory, the weight-gradient tiles (K = nodes) stay in accumulators for the wavefront's whole node range.
// Fixed node ranges per wavefront + partials summed in order by a second kernel: bitwise reproducible, no atomics.
#include "common.h"

namespace dmet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH = 32;
constexpr int kChunk = 32;                 // nodes per MFMA row tile
constexpr int kPad = 36;                   // LDS row stride in floats (16-byte aligned rows, banks spread)
constexpr int kWavesPerBlock = 4;
constexpr int kPartial = 2 * 1024 + kH;    // per wavefront: gP^T x, gQ^T x, column sums of gP
constexpr int kFinGroups = 33;             // finalize: 32 groups of 32 weight-tile elements + the bias
constexpr int kFinChunks = 8;              // (workspace layout of the former two-level finalize, kept: the size query is ABI)

__device__ __forceinline__ void ecb_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ARG16: `arg` holds uint16 winner ids per (node, channel) (0xFFFF = no winner: the counted radius gather's form,
// dmet_gather_max_local_j16_f32) instead of uint8 winning slots (255 = none); the mask is the same.
template <bool ARG16>
__global__ __launch_bounds__(kWave * kWavesPerBlock, 2) void edgeconv_linear_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ g_out,
    const uint8_t *__restrict__ arg, const float *__restrict__ gQ, int64_t N, int64_t nodes_per_wave,
    const float *__restrict__ g_add, float *__restrict__ gx, float *__restrict__ partial,
    int gq_sliced /* gQ is slice-major [8][N][4] (dmet_gather_max_bwd_sliced_f32) instead of [N][32] */)
{
    __shared__ float sP[kWavesPerBlock][kChunk * kPad];
    __shared__ float sQ[kWavesPerBlock][kChunk * kPad];
    __shared__ float sX[kWavesPerBlock][kChunk * kPad];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 31, hh = lane >> 5;
    float *P = sP[wv], *Q = sQ[wv], *X = sX[wv];
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wv;
    const int64_t lo = wave * nodes_per_wave, hi = min(N, lo + nodes_per_wave);

    // B operand of the gx product: Wst[k][j], k = 0..63 over [gP | gQ] columns, j = input feature:
    //   k <  32: (W1 - W2)[out k][in j] = W[k][j] - W[k][32 + j];   k >= 32: W2[out k-32][in j] = W[k-32][32 + j]
    // MFMA step s uses k = 2 s + hh.
    float wst[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int k = 2 * s + hh;
        wst[s] = (k < kH) ? (W[k * 64 + c] - W[k * 64 + kH + c]) : W[(k - kH) * 64 + kH + c];
    }
    f32x16 accP, accQ;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accP[e] = 0.0f; accQ[e] = 0.0f; }
    float bsum = 0.0f;

    const int lr = lane >> 3, lp = lane & 7;      // loader role: row within a group of 8, float4 column
    // software pipeline: the rows of the NEXT chunk are loaded into registers while the matrix products of the
    // current one run (two wavefronts per SIMD do not hide a global round trip per chunk by themselves)
    float4 np[kChunk / 8], nq[kChunk / 8], nx[kChunk / 8];
    uint2 na[kChunk / 8];       // the four winning-slot bytes (.x; ARG16: four 16-bit ids in .x, .y) that go with np[g]
    // The slot bytes are only LOADED here; the mask is applied when the rows go to LDS.  (Round 2, second session:
    // masking inside the prefetch consumed the loaded values on the spot -- s_waitcnt vmcnt(0) after each of the four
    // row groups, i.e. four exposed memory round trips per 32-node chunk: 8 us per chunk, matrix pipe 28 % busy.)
    auto fetch = [&](const int64_t base) {
#pragma unroll
        for (int g = 0; g < kChunk / 8; ++g) {
            const int64_t i = base + g * 8 + lr;
            float4 vp = make_float4(0.f, 0.f, 0.f, 0.f), vq = vp, vx = vp;
            uint2 va = make_uint2(0u, 0u);
            if (i < hi) {
                vp = reinterpret_cast<const float4 *>(g_out + i * kH)[lp];
                vq = gq_sliced ? reinterpret_cast<const float4 *>(gQ)[(int64_t)lp * N + i]     // 8 lanes: 128 contiguous bytes either way
                               : reinterpret_cast<const float4 *>(gQ + i * kH)[lp];
                vx = reinterpret_cast<const float4 *>(x + i * kH)[lp];
                if (arg) {
                    if (ARG16) va = reinterpret_cast<const uint2 *>(arg + i * kH * 2)[lp];
                    else va.x = reinterpret_cast<const unsigned *>(arg + i * kH)[lp];
                }
            }
            np[g] = vp; nq[g] = vq; nx[g] = vx; na[g] = va;
        }
    };
    if (lo < hi) fetch(lo);
    for (int64_t base = lo; base < hi; base += kChunk) {
        ecb_wave_sync();                           // previous chunk's tiles are consumed
#pragma unroll
        for (int g = 0; g < kChunk / 8; ++g) {
            const int r = g * 8 + lr;
            // nodes without any neighbour produced 0 (R3): no gradient reaches P there (slot byte 255)
            if (ARG16) {
                if ((na[g].x & 0xFFFFu) == 0xFFFFu) np[g].x = 0.f;
                if ((na[g].x >> 16) == 0xFFFFu) np[g].y = 0.f;
                if ((na[g].y & 0xFFFFu) == 0xFFFFu) np[g].z = 0.f;
                if ((na[g].y >> 16) == 0xFFFFu) np[g].w = 0.f;
            } else {
                if ((na[g].x & 0xFFu) == 0xFFu) np[g].x = 0.f;
                if (((na[g].x >> 8) & 0xFFu) == 0xFFu) np[g].y = 0.f;
                if (((na[g].x >> 16) & 0xFFu) == 0xFFu) np[g].z = 0.f;
                if ((na[g].x >> 24) == 0xFFu) np[g].w = 0.f;
            }
            *reinterpret_cast<float4 *>(&P[r * kPad + 4 * lp]) = np[g];
            *reinterpret_cast<float4 *>(&Q[r * kPad + 4 * lp]) = nq[g];
            *reinterpret_cast<float4 *>(&X[r * kPad + 4 * lp]) = nx[g];
        }
        ecb_wave_sync();
        // the residual branch's gradient of THIS chunk as row segments (loader layout), used after the products
        float4 ga[kChunk / 8];
#pragma unroll
        for (int g = 0; g < kChunk / 8; ++g) {
            const int64_t i = base + g * 8 + lr;
            ga[g] = (g_add && i < hi) ? reinterpret_cast<const float4 *>(g_add + i * kH)[lp]
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (base + kChunk < hi) fetch(base + kChunk);
        // gx tile [32 nodes][32 features] = [gP | gQ] . Wst   (A operand: lane (node c, hh) supplies G[c][2 s + hh])
        f32x16 accx;
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(P[c * kPad + 2 * s + hh], wst[s], accx, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(Q[c * kPad + 2 * s + hh], wst[16 + s], accx, 0, 0, 0);
        // weight-gradient tiles: C[m][j] += sum_node G[node][m] x[node][j]   (K = nodes, step s uses node 2 s + hh)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int node = 2 * s + hh;
            const float xv = X[node * kPad + c];
            accP = __builtin_amdgcn_mfma_f32_32x32x2f32(P[node * kPad + c], xv, accP, 0, 0, 0);
            accQ = __builtin_amdgcn_mfma_f32_32x32x2f32(Q[node * kPad + c], xv, accQ, 0, 0, 0);
        }
        // bias gradient: column c of gP, half of the rows per lane half
        {
            float t = 0.0f;
#pragma unroll
            for (int n = 0; n < 16; ++n) t += P[(hh * 16 + n) * kPad + c];
            bsum += t;
        }
        // store gx (+ the residual branch's gradient): accx[e] = row (e&3) + 8 (e>>2) + 4 hh, column c, transposed
        // through the P tile (consumed by now) so that global memory sees 16-byte row segments
        ecb_wave_sync();
#pragma unroll
        for (int e = 0; e < 16; ++e) P[((e & 3) + 8 * (e >> 2) + 4 * hh) * kPad + c] = accx[e];
        ecb_wave_sync();
#pragma unroll
        for (int g = 0; g < kChunk / 8; ++g) {
            const int r = g * 8 + lr;
            const int64_t i = base + r;
            const float4 t = *reinterpret_cast<const float4 *>(&P[r * kPad + 4 * lp]);
            if (i < hi)
                reinterpret_cast<float4 *>(gx + i * kH)[lp] =
                    make_float4(t.x + ga[g].x, t.y + ga[g].y, t.z + ga[g].z, t.w + ga[g].w);
        }
    }
    // one partial per WORKGROUP: the four wavefronts' tiles meet in LDS (their own P / Q / X tiles, consumed by now) and
    // are added in wavefront order -- a quarter of the partial traffic and of the finalize kernel's reads
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
        P[r * 32 + c] = accP[e];
        Q[r * 32 + c] = accQ[e];
    }
    const float b = bsum + __shfl_xor(bsum, 32, 64);
    if (hh == 0) X[c] = b;
    __syncthreads();
    float *out = partial + (int64_t)blockIdx.x * kPartial;
    for (int i = threadIdx.x; i < kPartial; i += kWave * kWavesPerBlock) {
        const float *src = i < 1024 ? &sP[0][i] : i < 2048 ? &sQ[0][i - 1024] : &sX[0][i - 2048];
        float t = src[0];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) t += src[w * kChunk * kPad];
        out[i] = t;
    }
}

// gW[o][0:32] = sum gP^T x,  gW[o][32:64] = sum gQ^T x - sum gP^T x,  gb[o] = sum gP.  Partials (one per workgroup of the
// kernel above: 512 at 288 000 nodes) are added in a fixed order in ONE pass: workgroup cg owns 32 elements of both
// weight tiles (cg = 32: the bias); its 32 thread groups each take every 32nd partial (all loads of a thread in
// flight at once), the 32 group sums are added in order.  (Until the third session of round 2 the partials were cut
// into 8 ranges over 8 x 33 workgroups and the last one to finish -- a ticket -- added the range sums: a chain of four
// dependent device-scope round trips of ~1.6 us each, 10.7 us; with one partial per WORKGROUP there are few enough for
// one level.)
__global__ __launch_bounds__(1024) void edgeconv_linear_bwd_finalize_kernel(const float *__restrict__ partial,
                                                                             int64_t nparts, float *__restrict__ gW,
                                                                             float *__restrict__ gb)
{
    __shared__ float red0[32][33], red1[32][33];
    const int cg = blockIdx.x;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const bool bias = cg == 32;
    const int idx = bias ? (2048 + e) : (cg * 32 + e);
    float s0 = 0.0f, s1 = 0.0f;
    if (bias) {
#pragma unroll 16   // independent loads: all in flight (the sum order is unchanged)
        for (int64_t w = grp; w < nparts; w += 32) s0 += partial[w * kPartial + idx];
    } else {
#pragma unroll 16
        for (int64_t w = grp; w < nparts; w += 32) {
            s0 += partial[w * kPartial + idx];
            s1 += partial[w * kPartial + 1024 + idx];
        }
    }
    red0[grp][e] = s0; red1[grp][e] = s1;
    __syncthreads();
    if (grp != 0) return;
    s0 = 0.0f; s1 = 0.0f;
#pragma unroll
    for (int g = 0; g < 32; ++g) { s0 += red0[g][e]; s1 += red1[g][e]; }
    if (bias) {
        if (gb) gb[e] = s0;
    } else {
        gW[(idx >> 5) * 64 + (idx & 31)] = s0;
        gW[(idx >> 5) * 64 + 32 + (idx & 31)] = s1 - s0;
    }
}

inline int64_t ecb_nodes_per_wave(int64_t N, int64_t *nwaves)
{
    const int64_t target = 2048;                   // 256 CUs x 8 resident wavefronts
    int64_t npw = (N + target - 1) / target;
    npw = (npw + kChunk - 1) / kChunk * kChunk;
    if (npw < kChunk) npw = kChunk;
    int64_t nw = (N + npw - 1) / npw;
    nw = (nw + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
    *nwaves = nw;
    return npw;
}

// ---- gQ without a reverse index: per-event scatter into LDS with exact (order-independent) integer sums -----------
// gQ[j,c] = sum over (i,s) with nbr[i,s] == j and arg[i,c] == s of g_out[i,c].  The reverse-index route costs a stable
// radix sort of N*k keys per layer (~250 us at N*k = 4.6M) plus an L2 gather per reverse edge.  Here a workgroup owns
// (event, 4 channels): every node's winner j is looked up in its nbr row and the gradient is ADDED INTO LDS at
// [j][channel] as a 64-bit fixed-point integer.  Integer addition is associative, so the result does not depend on
// the order in which the lanes' atomics land: bitwise reproducible without any sorting.  The fixed-point scale is a
// power of two chosen per workgroup from max |g| over its slice (30 value bits, 33 bits of headroom: up to 2^33
// terms per cell); a term is rounded once to 2^-30 of that maximum, the sum is exact.
// Events larger than the LDS window are done in several passes over fewer channels / a range of j.
constexpr int kBwdThreads = 1024;
constexpr int kBwdCells = 18432;             // 64-bit cells in LDS (144 KB)

// J16: `arg` holds the winner's event-local node id per (node, channel) (uint16, 0xFFFF = none; written by
// dmet_gather_max_counted_lds_j16_f32) instead of the winning slot: no table look-up at all.
template <bool J16 = false, int THREADS = kBwdThreads, int CELLS = kBwdCells>
__global__ __launch_bounds__(THREADS) void gather_max_bwd_lds_kernel(const float *__restrict__ g_out,
                                                                          const uint8_t *__restrict__ arg,
                                                                          const int32_t *__restrict__ nbr,
                                                                          const uint16_t *__restrict__ nbr16,
                                                                          const int64_t *__restrict__ ptr, int B, int k,
                                                                          float *__restrict__ gQ, int64_t gq_sliced_n)
{
    // gq_sliced_n = N > 0: gQ is written slice-major, [8][N][4] -- the workgroup's 4 channels of its event are ONE
    // contiguous run (16-byte pieces of 128-byte rows otherwise: 13 of the kernel's 46 us at 64 x 4500 were the
    // write-out of such pieces); dmet_edgeconv_linear_bwd_sliced_f32 reads that layout
    // (the names of the full-size constants, shadowed: a workgroup of THREADS threads with CELLS accumulators)
    constexpr int kBwdThreads = THREADS, kBwdCells = CELLS;
    extern __shared__ unsigned long long cells[];
    __shared__ float red[kBwdThreads / kWave];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ev = bid >> 3, slice = bid & 7;
    if (ev >= B) return;
    const int c0 = 4 * slice;
    const int64_t lo = ptr[ev], hi = ptr[ev + 1];
    const int n = (int)(hi - lo);
    if (n <= 0) return;
    const int tid = threadIdx.x;
    const bool loc16 = nbr16 != nullptr && n <= 65535;   // event-local uint16 table (half the row bytes) when given

    // window: CH channels x JW nodes of accumulators
    const int CH = (n <= kBwdCells / 4) ? 4 : (n <= kBwdCells / 2) ? 2 : 1;
    // Events of 4 609 .. 9 216 nodes (the upper half of a ragged batch, BASELINE configs[4]): the window holds two of
    // the slice's four channels, so the event takes two accumulation passes -- but only ONE look-up pass: the first
    // pass resolves the winners of all four channels (slot -> id, the dependent loads this kernel waits on), scatters
    // channels 0..1 and keeps the ids of channels 2..3 (and the gradients) in registers for the second, which then is
    // LDS atomics only.  (Before: both passes walked arg / ids / g_out again and a large event cost twice its nodes.)
    constexpr int kBwdBig = 9;                        // rows per thread: 9 x 1024 = 9 216 = kBwdCells / 2
    if (CH == 2) {
        float4 g9[kBwdBig];
        unsigned jk[kBwdBig];
        float mb = 0.0f;
#pragma unroll
        for (int u = 0; u < kBwdBig; ++u) {
            const int i = tid + u * kBwdThreads;
            g9[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) g9[u] = reinterpret_cast<const float4 *>(g_out + (lo + i) * kH + c0)[0];
            mb = fmaxf(mb, fmaxf(fmaxf(fabsf(g9[u].x), fabsf(g9[u].y)), fmaxf(fabsf(g9[u].z), fabsf(g9[u].w))));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mb = fmaxf(mb, __shfl_xor(mb, off, 64));
        if ((tid & 63) == 0) red[tid >> 6] = mb;
        __syncthreads();
        mb = red[0];
#pragma unroll
        for (int w = 1; w < kBwdThreads / kWave; ++w) mb = fmaxf(mb, red[w]);
        int exb = 0;
        (void)frexpf(mb, &exb);
        exb = max(exb, -96);
        const float scale_b = ldexpf(1.0f, 30 - exb), inv_b = ldexpf(1.0f, exb - 30);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            for (int t = tid; t < n * 2; t += kBwdThreads) cells[t] = 0ull;
            __syncthreads();
            if (mb > 0.0f) {
#pragma unroll
                for (int u = 0; u < kBwdBig; ++u) {
                    const int i = tid + u * kBwdThreads;
                    if (i < n) {
                        const int64_t gi = lo + i;
                        unsigned ja = 0xFFFFu, jb = 0xFFFFu;       // winners of this pass's two channels (0xFFFF: none)
                        if (pass == 0) {
                            unsigned jj[4];
                            if (J16) {
                                const ushort4 a4 = reinterpret_cast<const ushort4 *>(arg)[(gi * kH + c0) >> 2];
                                jj[0] = a4.x; jj[1] = a4.y; jj[2] = a4.z; jj[3] = a4.w;
                            } else {
                                const uchar4 a4 = reinterpret_cast<const uchar4 *>(arg + gi * kH + c0)[0];
                                const unsigned as[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const unsigned sl = as[q] != 255u ? as[q] : 0u;       // clamped: the load is always issued
                                    const unsigned j = loc16 ? (unsigned)nbr16[gi * k + sl] : (unsigned)(nbr[gi * k + sl] - (int)lo);
                                    jj[q] = (as[q] != 255u && j < (unsigned)n) ? j : 0xFFFFu;      // (a -1 slot would wrap to >= n)
                                }
                            }
                            ja = jj[0]; jb = jj[1];
                            jk[u] = (jj[2] & 0xFFFFu) | (jj[3] << 16);
                        } else {
                            ja = jk[u] & 0xFFFFu; jb = jk[u] >> 16;
                        }
                        const float ga = pass == 0 ? g9[u].x : g9[u].z, gb2 = pass == 0 ? g9[u].y : g9[u].w;
                        if (ja < (unsigned)n) atomicAdd(&cells[ja * 2], (unsigned long long)(long long)__float2int_rn(ga * scale_b));
                        if (jb < (unsigned)n) atomicAdd(&cells[jb * 2 + 1], (unsigned long long)(long long)__float2int_rn(gb2 * scale_b));
                    }
                }
            }
            __syncthreads();
            for (int j = tid; j < n; j += kBwdThreads) {
                const float2 o = make_float2((float)(long long)cells[2 * j] * inv_b, (float)(long long)cells[2 * j + 1] * inv_b);
                if (gq_sliced_n > 0) *reinterpret_cast<float2 *>(gQ + ((int64_t)slice * gq_sliced_n + lo + j) * 4 + 2 * pass) = o;
                else *reinterpret_cast<float2 *>(gQ + (lo + j) * kH + c0 + 2 * pass) = o;
            }
        }
        return;
    }

    // scale: 2^(30 - e) with 2^e > max |g| over the slice (exactly representable, so the final rescale is exact)
    // the slice of g_out is needed twice (its maximum, then the terms): events of up to kBwdKeep * 1024 nodes keep it
    // in registers in between instead of reading the 16-byte pieces of the rows a second time
    constexpr int kBwdKeep = 5;
    const bool keep = n <= kBwdKeep * kBwdThreads;    // block-uniform
    float4 gk[kBwdKeep];
    float m = 0.0f;
    if (keep) {
#pragma unroll
        for (int u = 0; u < kBwdKeep; ++u) {
            const int i = tid + u * kBwdThreads;
            gk[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#if defined(DMET_BWD_ABL) && DMET_BWD_ABL == 5      // timing experiment: g_out read as if slice-major
            if (i < n) gk[u] = reinterpret_cast<const float4 *>(g_out)[(int64_t)slice * (gq_sliced_n > 0 ? gq_sliced_n : 288000) + lo + i];
#else
            if (i < n) gk[u] = reinterpret_cast<const float4 *>(g_out + (lo + i) * kH + c0)[0];
#endif
            m = fmaxf(m, fmaxf(fmaxf(fabsf(gk[u].x), fabsf(gk[u].y)), fmaxf(fabsf(gk[u].z), fabsf(gk[u].w))));
        }
    } else {
        for (int i = tid; i < n; i += kBwdThreads) {
            const float4 g = reinterpret_cast<const float4 *>(g_out + (lo + i) * kH + c0)[0];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < kBwdThreads / kWave; ++w) m = fmaxf(m, red[w]);
    int ex = 0;
    (void)frexpf(m, &ex);                       // m = f * 2^ex, f in [0.5, 1)
    ex = max(ex, -96);                          // gradients below 2^-96 everywhere: keep the scale finite
    const float scale = ldexpf(1.0f, 30 - ex), inv_scale = ldexpf(1.0f, ex - 30);

    const int JW = kBwdCells / CH;
    for (int cb = 0; cb < 4; cb += CH) {
        for (int j0 = 0; j0 < n; j0 += JW) {
            const int jn = min(JW, n - j0);
            __syncthreads();
            for (int t = tid; t < jn * CH; t += kBwdThreads) cells[t] = 0ull;
            __syncthreads();
#if defined(DMET_BWD_ABL) && DMET_BWD_ABL == 3
            if (m == 123.456f) {    // timing experiment: no look-up / accumulation phase at all
#else
            if (m > 0.0f) {
#endif
                int u_keep = 0;
                for (int i = tid; i < n; i += kBwdThreads, ++u_keep) {
                    const int64_t gi = lo + i;
                    unsigned as[4];
                    if (J16) {
                        const ushort4 a4 = reinterpret_cast<const ushort4 *>(arg)[(gi * kH + c0) >> 2];
                        as[0] = a4.x; as[1] = a4.y; as[2] = a4.z; as[3] = a4.w;
                    } else {
                        const uchar4 a4 = reinterpret_cast<const uchar4 *>(arg + gi * kH + c0)[0];
                        as[0] = a4.x; as[1] = a4.y; as[2] = a4.z; as[3] = a4.w;
                    }
                    float4 g4;
                    if (keep) {   // static register indices: select, do not index
                        g4 = gk[0];
#pragma unroll
                        for (int u = 1; u < kBwdKeep; ++u) if (u_keep == u) g4 = gk[u];
                    } else {
                        g4 = reinterpret_cast<const float4 *>(g_out + gi * kH + c0)[0];
                    }
                    const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (J16) {
                            if (u >= cb && u < cb + CH && as[u] != 0xFFFFu) {
                                const int j = (int)as[u] - j0;
                                if (j >= 0 && j < jn) {
                                    const long long q = (long long)__float2int_rn(gs[u] * scale);   // |g| scale < 2^30
                                    atomicAdd(&cells[j * CH + (u - cb)], (unsigned long long)q);
                                }
                            }
                        } else if (u >= cb && u < cb + CH && as[u] != 255) {
#ifdef DMET_KNN_EXPERIMENT
                            const int j = (k == 1 ? i : (loc16 ? (int)nbr16[gi * k + as[u]] : nbr[gi * k + as[u]] - (int)lo)) - j0;   // k == 1: upper bound of what skipping the id fetch would save
#else
                            const int j = (loc16 ? (int)nbr16[gi * k + as[u]] : nbr[gi * k + as[u]] - (int)lo) - j0;
#endif
                            if (j >= 0 && j < jn) {
                                // |g| scale < 2^30: one v_cvt_i32_f32 (round to nearest even, like __float2ll_rn) and a
                                // sign extension instead of the dozen instructions of a float -> int64 conversion
                                const long long q = (long long)__float2int_rn(gs[u] * scale);
#if defined(DMET_BWD_ABL) && DMET_BWD_ABL == 1
                                cells[j * CH + (u - cb)] = (unsigned long long)q;       // timing experiment: plain store
#elif defined(DMET_BWD_ABL) && DMET_BWD_ABL == 4
                                if (q == 0x7fffffffll) cells[j * CH + (u - cb)] = (unsigned long long)q;   // timing experiment: no LDS traffic
#else
                                atomicAdd(&cells[j * CH + (u - cb)], (unsigned long long)q);
#endif
                            }
                        }
                    }
                }
            }
            __syncthreads();
#if defined(DMET_BWD_ABL) && DMET_BWD_ABL == 2
            if (m == 123.456f)    // timing experiment: no write-out
#endif
            if (CH == 4) {   // one node per thread: 16-byte stores
                for (int j = tid; j < jn; j += kBwdThreads) {
                    const unsigned long long *cj = cells + 4 * j;
                    const float4 o = make_float4((float)(long long)cj[0] * inv_scale, (float)(long long)cj[1] * inv_scale,
                                                 (float)(long long)cj[2] * inv_scale, (float)(long long)cj[3] * inv_scale);
                    if (gq_sliced_n > 0) reinterpret_cast<float4 *>(gQ)[(int64_t)slice * gq_sliced_n + lo + j0 + j] = o;
                    else *reinterpret_cast<float4 *>(gQ + (lo + j0 + j) * kH + c0) = o;
                }
            } else {
                for (int t = tid; t < jn * CH; t += kBwdThreads) {
                    const int j = t / CH, u = t - j * CH;
                    const float o1 = (float)(long long)cells[t] * inv_scale;
                    if (gq_sliced_n > 0) gQ[((int64_t)slice * gq_sliced_n + lo + j0 + j) * 4 + cb + u] = o1;
                    else gQ[(lo + j0 + j) * kH + c0 + cb + u] = o1;
                }
            }
        }
    }
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_edgeconv_linear_bwd_workspace_bytes(int64_t N, int H)
{
    if (N <= 0 || H != kH) return 0;
    int64_t nw;
    (void)ecb_nodes_per_wave(N, &nw);
    return sizeof(float) * ((size_t)nw + kFinChunks) * kPartial + sizeof(int) * 64 + 512;
}

static int ecb_launch(const float *x, const float *W, const float *g_out, const uint8_t *arg, bool arg16, const float *gQ,
                      const float *g_add, int64_t N, int H, float *gx, float *gW, float *gb, void *ws, size_t ws_bytes,
                      dmet_stream_t stream, int gq_sliced = 0)
{
    DMET_REQUIRE(H == kH, "dmet_edgeconv_linear_bwd_f32: H=%d (only 32 is built)", H);
    DMET_REQUIRE(N > 0, "dmet_edgeconv_linear_bwd_f32: N=%lld", (long long)N);
    DMET_REQUIRE(x && W && g_out && gQ && gx && gW && ws, "dmet_edgeconv_linear_bwd_f32: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(g_out) && aligned16(gQ) && aligned16(gx) && aligned16(g_add) &&
                     (!arg || (reinterpret_cast<uintptr_t>(arg) & (arg16 ? 7u : 3u)) == 0),
                 "dmet_edgeconv_linear_bwd_f32: rows must be 16-byte aligned");
    DMET_REQUIRE(ws_bytes >= dmet_edgeconv_linear_bwd_workspace_bytes(N, H), "dmet_edgeconv_linear_bwd_f32: workspace too small");
    int64_t nw;
    const int64_t npw = ecb_nodes_per_wave(N, &nw);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    hipStream_t st = as_stream(stream);
    if (arg16)
        hipLaunchKernelGGL(edgeconv_linear_bwd_kernel<true>, dim3((unsigned)(nw / kWavesPerBlock)), dim3(kWave * kWavesPerBlock),
                           0, st, x, W, g_out, arg, gQ, N, npw, g_add, gx, partial, gq_sliced);
    else
        hipLaunchKernelGGL(edgeconv_linear_bwd_kernel<false>, dim3((unsigned)(nw / kWavesPerBlock)), dim3(kWave * kWavesPerBlock),
                           0, st, x, W, g_out, arg, gQ, N, npw, g_add, gx, partial, gq_sliced);
    DMET_LAUNCH_CHECK("edgeconv_linear_bwd_kernel");
    static_assert(kPartial == kEcbPartialFloats && kFinGroups == 33, "csrc/finalize.hip sums the same partial layout");
    if (defer_push(DeferDesc{kDeferEdgeConv, partial, nw / kWavesPerBlock, {gW, gb}})) return 0;   // summed by dmet_finalize_flush
    hipLaunchKernelGGL(edgeconv_linear_bwd_finalize_kernel, dim3(kFinGroups), dim3(1024), 0, st, partial, nw / kWavesPerBlock,
                       gW, gb);
    DMET_LAUNCH_CHECK("edgeconv_linear_bwd_finalize_kernel");
    return 0;
}

extern "C" int dmet_edgeconv_linear_bwd_add_f32(const float *x, const float *W, const float *g_out,
                                                const uint8_t *arg, const float *gQ, const float *g_add, int64_t N,
                                                int H, float *gx, float *gW, float *gb, void *ws, size_t ws_bytes,
                                                dmet_stream_t stream)
{
    return ecb_launch(x, W, g_out, arg, false, gQ, g_add, N, H, gx, gW, gb, ws, ws_bytes, stream);
}

extern "C" int dmet_edgeconv_linear_bwd_add_j16_f32(const float *x, const float *W, const float *g_out,
                                                    const uint16_t *argj, const float *gQ, const float *g_add, int64_t N,
                                                    int H, float *gx, float *gW, float *gb, void *ws, size_t ws_bytes,
                                                    dmet_stream_t stream)
{
    return ecb_launch(x, W, g_out, reinterpret_cast<const uint8_t *>(argj), true, gQ, g_add, N, H, gx, gW, gb, ws, ws_bytes,
                      stream);
}

extern "C" int dmet_edgeconv_linear_bwd_f32(const float *x, const float *W, const float *g_out, const uint8_t *arg,
                                            const float *gQ, int64_t N, int H, float *gx, float *gW, float *gb,
                                            void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    return dmet_edgeconv_linear_bwd_add_f32(x, W, g_out, arg, gQ, nullptr, N, H, gx, gW, gb, ws, ws_bytes, stream);
}

// One launch of the scatter with a workgroup sized for the batch's largest event: the accumulators of a workgroup are
// (event nodes) x 4 channels x 8 bytes of LDS, and a workgroup that reserves the whole CU's 144 KB for an event of 2 000
// nodes runs alone on its CU through its serial phases (maximum, clearing, write-out).  `max_nodes` is a HINT (0 =
// unknown): events larger than the chosen window take more passes, never a wrong result.
//   max_nodes <= 1 152: 256 threads,  36 KB -> four workgroups per CU;   <= 2 304: 512 threads, 72 KB -> two;
//   otherwise (and unknown): 1 024 threads, 144 KB.
// 128 events of 1000 / 2000 nodes: 29.2 -> 20.3 us / 43.8 -> 35.8 us (same bits: integer sums).
template <bool J16, int THREADS, int CELLS>
int bwd_scatter_launch_as(const float *g_out, const uint8_t *arg, const int32_t *nbr, const uint16_t *nbr16,
                          const int64_t *ptr, int B, int k, float *gQ, hipStream_t st, int64_t gq_sliced_n)
{
    static bool attr_set = false;      // one flag per instantiation
    const size_t lds = sizeof(unsigned long long) * (size_t)CELLS;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gather_max_bwd_lds_kernel<J16, THREADS, CELLS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(gather_max_bwd_lds_kernel)");
        attr_set = true;
    }
    hipLaunchKernelGGL((gather_max_bwd_lds_kernel<J16, THREADS, CELLS>), dim3((unsigned)B * 8u), dim3(THREADS), lds, st, g_out,
                       arg, nbr, nbr16, ptr, B, k, gQ, gq_sliced_n);
    DMET_LAUNCH_CHECK("gather_max_bwd_lds_kernel");
    return 0;
}

template <bool J16>
int bwd_scatter_launch(const float *g_out, const uint8_t *arg, const int32_t *nbr, const uint16_t *nbr16,
                       const int64_t *ptr, int B, int k, float *gQ, int64_t max_nodes, hipStream_t st, int64_t gq_sliced_n = 0)
{
    if (max_nodes > 0 && max_nodes <= kBwdCells / 16)
        return bwd_scatter_launch_as<J16, 256, kBwdCells / 4>(g_out, arg, nbr, nbr16, ptr, B, k, gQ, st, gq_sliced_n);
    if (max_nodes > 0 && max_nodes <= kBwdCells / 8)
        return bwd_scatter_launch_as<J16, 512, kBwdCells / 2>(g_out, arg, nbr, nbr16, ptr, B, k, gQ, st, gq_sliced_n);
    return bwd_scatter_launch_as<J16, kBwdThreads, kBwdCells>(g_out, arg, nbr, nbr16, ptr, B, k, gQ, st, gq_sliced_n);
}

// gQ slice-major, [8][N][4] floats (see the kernel): for the pair scatter -> dmet_edgeconv_linear_bwd_sliced_f32
extern "C" int dmet_gather_max_bwd_sliced_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr,
                                              const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k, int H,
                                              float *gQs, int64_t max_nodes, dmet_stream_t stream)
{
    DMET_REQUIRE(H == kH, "dmet_gather_max_bwd_sliced_f32: H=%d (only 32 is built)", H);
    DMET_REQUIRE(N >= 0 && B >= 0 && k >= 1 && k <= 255 && max_nodes >= 0, "dmet_gather_max_bwd_sliced_f32: bad sizes");
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(g_out && arg && nbr && ptr && gQs, "dmet_gather_max_bwd_sliced_f32: null pointer");
    DMET_REQUIRE(aligned16(g_out) && aligned16(gQs) && (reinterpret_cast<uintptr_t>(arg) & 3u) == 0,
                 "dmet_gather_max_bwd_sliced_f32: rows must be 16-byte aligned");
    return bwd_scatter_launch<false>(g_out, arg, nbr, nbr_local, ptr, B, k, gQs, max_nodes, as_stream(stream), N);
}

extern "C" int dmet_gather_max_bwd_j16_sliced_f32(const float *g_out, const uint16_t *argj, const int64_t *ptr, int B,
                                                  int64_t N, int H, float *gQs, int64_t max_nodes, dmet_stream_t stream)
{
    DMET_REQUIRE(H == kH, "dmet_gather_max_bwd_j16_sliced_f32: H=%d (only 32 is built)", H);
    DMET_REQUIRE(N >= 0 && B >= 0 && max_nodes >= 0, "dmet_gather_max_bwd_j16_sliced_f32: bad sizes");
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(g_out && argj && ptr && gQs, "dmet_gather_max_bwd_j16_sliced_f32: null pointer");
    DMET_REQUIRE(aligned16(g_out) && aligned16(gQs) && (reinterpret_cast<uintptr_t>(argj) & 7u) == 0,
                 "dmet_gather_max_bwd_j16_sliced_f32: rows must be 16-byte (argj: 8-byte) aligned");
    return bwd_scatter_launch<true>(g_out, reinterpret_cast<const uint8_t *>(argj), nullptr, nullptr, ptr, B, 1, gQs, max_nodes,
                                    as_stream(stream), N);
}

// dmet_edgeconv_linear_bwd_add_f32 / _add_j16_f32 (arg_is_j16) reading gQ in the slice-major layout those two write
extern "C" int dmet_edgeconv_linear_bwd_sliced_f32(const float *x, const float *W, const float *g_out, const void *arg,
                                                   int arg_is_j16, const float *gQs, const float *g_add, int64_t N, int H,
                                                   float *gx, float *gW, float *gb, void *ws, size_t ws_bytes,
                                                   dmet_stream_t stream)
{
    return ecb_launch(x, W, g_out, reinterpret_cast<const uint8_t *>(arg), arg_is_j16 != 0, gQs, g_add, N, H, gx, gW, gb, ws,
                      ws_bytes, stream, 1);
}

extern "C" int dmet_gather_max_bwd_lds16_cap_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr,
                                                 const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k,
                                                 int H, float *gQ, int64_t max_nodes, dmet_stream_t stream)
{
    DMET_REQUIRE(H == kH, "dmet_gather_max_bwd_lds_f32: H=%d (only 32 is built)", H);
    DMET_REQUIRE(N >= 0 && B >= 0 && k >= 1 && k <= 255 && max_nodes >= 0, "dmet_gather_max_bwd_lds_f32: bad sizes");
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(g_out && arg && nbr && ptr && gQ, "dmet_gather_max_bwd_lds_f32: null pointer");
    DMET_REQUIRE(aligned16(g_out) && aligned16(gQ) && (reinterpret_cast<uintptr_t>(arg) & 3u) == 0,
                 "dmet_gather_max_bwd_lds_f32: rows must be 16-byte aligned");
    return bwd_scatter_launch<false>(g_out, arg, nbr, nbr_local, ptr, B, k, gQ, max_nodes, as_stream(stream));
}

extern "C" int dmet_gather_max_bwd_lds16_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr,
                                             const uint16_t *nbr_local, const int64_t *ptr, int B, int64_t N, int k,
                                             int H, float *gQ, dmet_stream_t stream)
{
    return dmet_gather_max_bwd_lds16_cap_f32(g_out, arg, nbr, nbr_local, ptr, B, N, k, H, gQ, 0, stream);
}

extern "C" int dmet_gather_max_bwd_lds_f32(const float *g_out, const uint8_t *arg, const int32_t *nbr,
                                           const int64_t *ptr, int B, int64_t N, int k, int H, float *gQ,
                                           dmet_stream_t stream)
{
    return dmet_gather_max_bwd_lds16_f32(g_out, arg, nbr, nullptr, ptr, B, N, k, H, gQ, stream);
}

extern "C" int dmet_gather_max_bwd_j16_cap_f32(const float *g_out, const uint16_t *argj, const int64_t *ptr, int B,
                                               int64_t N, int H, float *gQ, int64_t max_nodes, dmet_stream_t stream)
{
    DMET_REQUIRE(H == kH, "dmet_gather_max_bwd_j16_f32: H=%d (only 32 is built)", H);
    DMET_REQUIRE(N >= 0 && B >= 0 && max_nodes >= 0, "dmet_gather_max_bwd_j16_f32: bad sizes");
    if (N == 0 || B == 0) return 0;
    DMET_REQUIRE(g_out && argj && ptr && gQ, "dmet_gather_max_bwd_j16_f32: null pointer");
    DMET_REQUIRE(aligned16(g_out) && aligned16(gQ) && (reinterpret_cast<uintptr_t>(argj) & 7u) == 0,
                 "dmet_gather_max_bwd_j16_f32: rows must be 16-byte (argj: 8-byte) aligned");
    return bwd_scatter_launch<true>(g_out, reinterpret_cast<const uint8_t *>(argj), nullptr, nullptr, ptr, B, 1, gQ, max_nodes,
                                    as_stream(stream));
}

extern "C" int dmet_gather_max_bwd_j16_f32(const float *g_out, const uint16_t *argj, const int64_t *ptr, int B, int64_t N,
                                           int H, float *gQ, dmet_stream_t stream)
{
    return dmet_gather_max_bwd_j16_cap_f32(g_out, argj, ptr, B, N, H, gQ, 0, stream);
}
