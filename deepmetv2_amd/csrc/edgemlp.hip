// edgemlp.hip -- K2 for a generic two-layer edge MLP on the bf16 matrix cores (gfx950), fused with the aggregation.
//
// Replaces, for  nn = Sequential(Linear(2H, H1), ELU, Linear(H1, H2)[, ELU])  over a fixed-width neighbour table,
//   torch_geometric.nn.EdgeConv.forward = index_select x2 -> cat([x_i, x_j - x_i]) -> nn -> scatter(max | add)
// i.e. the call shape of /root/reference/model/dynamic_reduction_network.py:59-73,86-87 (Linear-ELU-Linear-ELU) and
// BASELINE configs[2] ("bf16 with MFMA edge-MLP").  No [E, 2H] tensor, no [E, H1] tensor and no [E, H2] message tensor
// ever exists in HBM: a wavefront takes 32 edges (32 / k target nodes), builds their [x_i || x_j - x_i] rows as bf16
// MFMA operands straight from the node table, runs both dense layers with v_mfma_f32_32x32x16_bf16 (fp32 accumulate),
// applies ELU in fp32 and reduces the 32 messages to their targets with DPP row reductions.
//
// Both products are computed TRANSPOSED (channels x edges): z1^T = W1 . feat^T, z2^T = W2 . h1^T.  The accumulator of
// the first product then has the edge on the lane and the hidden channel in the registers, which is exactly the B
// operand the second product needs (it sums over the hidden channel): registers 8s..8s+7 converted to bf16 are the
// fragment of k-step s, no LDS round trip, no lane movement; the permuted k order that comes with it is absorbed into
// the way W2 is laid out in LDS.  Biases enter as the accumulators' initial values.
//   MFMA operand maps (32x32x16 bf16): lane (r = lane & 31, hh = lane >> 5) holds A[row r][k = 8 hh + j],
//   B[k = 8 hh + j][col r], j = 0..7;  C/D: col = r, row = (reg & 3) + 8 (reg >> 2) + 4 hh.
// Numerics: inputs and weights rounded to bf16 (RNE), products exact, fp32 accumulation, ELU / max / sum in fp32:
// the R6 bf16 bar (rtol 2e-2) against the fp32 oracle.
#include <type_traits>

#include "common.h"

namespace dmet {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float f)
{
    // hipcc turns this cast into v_cvt_pk_bf16_f32 (keeps a NaN a NaN; guides/MI355X_MICROARCH.md)
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ float elu1(float z)
{
    // ELU(alpha = 1): z > 0 ? z : exp(z) - 1
    const float e = __builtin_amdgcn_exp2f(z * 1.44269504088896341f) - 1.0f;
    return z > 0.0f ? z : e;
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// all-lanes reduction over the groups of `k` consecutive lanes (k = 8, 16, 32) that hold one target node's edges;
// every lane of a group ends up with the group's result (butterfly: same association in every lane, deterministic)
template <bool ADD>
__device__ __forceinline__ float group_reduce(float v, int k)
{
    auto op = [](float a, float b) { return ADD ? a + b : __builtin_fmaxf(a, b); };
    v = op(v, dpp<0xB1>(v));      // quad_perm [1,0,3,2]
    v = op(v, dpp<0x4E>(v));      // quad_perm [2,3,0,1]
    v = op(v, dpp<0x141>(v));     // row_half_mirror: lanes l <-> 7 - l of each 8
    if (k >= 16) v = op(v, dpp<0x140>(v));   // row_mirror: l <-> 15 - l of each 16
    if (k >= 32) v = op(v, __shfl_xor(v, 16, 64));
    return v;
}

template <int HIN, int H1P, int H2>
struct EdgeMlpLds {
    static constexpr int MB1 = H1P / 32, KS1 = 2 * HIN / 16, MB2 = H2 / 32, KS2 = H1P / 16;
    bf16x8 w1[MB1][KS1][kWave];   // A fragments of W1: row 32 mb + r, columns 16 kk + 8 hh + 0..7
    bf16x8 w2[MB2][KS2][kWave];   // A fragments of W2 in the k order of the converted accumulators
    float b1[H1P];
    float b2[H2];
};

// BN (the trailing BatchNorm1d of the DRN's edge MLP, model/dynamic_reduction_network.py:59-70: it normalises the
// MESSAGES, before the aggregation): a per-channel affine map m -> a m + b commutes with the aggregation --
// sum_s (a m_s + b) = a sum_s m_s + cnt b, and max_s (a m_s + b) = a max_s m_s + b for a >= 0, a min_s m_s + b for
// a < 0 -- so ONE pass over the edges suffices: it writes the un-normalised aggregates (sum + edge count, or max and
// min), and, for batch statistics, per-wavefront partial sums of m and m^2 over the valid edges; a one-workgroup kernel
// turns those into (a, b) and a node-level kernel applies them (edge_mlp2_bn_*_kernel below).
template <int HIN, int H1P, int H2, bool ADD, bool BN = false>
__global__ __launch_bounds__(256, 2) void edge_mlp2_kernel(const float *__restrict__ x, const int32_t *__restrict__ nbr,
                                                            int64_t N, int k, const float *__restrict__ W1,
                                                            const float *__restrict__ b1, int H1,
                                                            const float *__restrict__ W2, const float *__restrict__ b2,
                                                            int act2, float *__restrict__ out,
                                                            float *__restrict__ out2 = nullptr,
                                                            float *__restrict__ partial = nullptr)
{
    using L = EdgeMlpLds<HIN, H1P, H2>;
    constexpr int MB1 = L::MB1, KS1 = L::KS1, MB2 = L::MB2, KS2 = L::KS2, KH = HIN / 16;
    __shared__ L S;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    // ---- stage the weights once per workgroup, already in fragment order (bf16) ----------------------------------
    for (int idx = threadIdx.x; idx < MB1 * KS1 * kWave; idx += blockDim.x) {
        const int l = idx % kWave, kk = (idx / kWave) % KS1, mb = idx / (kWave * KS1);
        const int row = 32 * mb + (l & 31), col0 = 16 * kk + 8 * (l >> 5);
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (short)bf16_rne(row < H1 ? W1[(int64_t)row * (2 * HIN) + col0 + j] : 0.0f);
        S.w1[mb][kk][l] = f;
    }
    for (int idx = threadIdx.x; idx < MB2 * KS2 * kWave; idx += blockDim.x) {
        const int l = idx % kWave, ks = (idx / kWave) % KS2, mb2 = idx / (kWave * KS2);
        const int row = 32 * mb2 + (l & 31), mb1 = ks >> 1, s = ks & 1, lh = l >> 5;
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * mb1 + 16 * s + 8 * (j >> 2) + 4 * lh + (j & 3);   // hidden channel of element j
            f[j] = (short)bf16_rne(c < H1 ? W2[(int64_t)row * H1 + c] : 0.0f);
        }
        S.w2[mb2][ks][l] = f;
    }
    for (int c = threadIdx.x; c < H1P; c += blockDim.x) S.b1[c] = (c < H1 && b1) ? b1[c] : 0.0f;
    for (int c = threadIdx.x; c < H2; c += blockDim.x) S.b2[c] = b2 ? b2[c] : 0.0f;
    __syncthreads();

    const int npt = 32 / k;                                  // target nodes per 32-edge tile
    const int64_t ntiles = (N + npt - 1) / npt;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const float ninf = -__builtin_inff();
    // BN: this lane's sums of m and m^2 over the valid edges it holds, for channel r of each 32-channel block, + edge count
    // (round 3: the channel sits on the lane now, so ONE running sum per 32-channel block and lane -- over the edges the lane
    // holds: half of every tile's; the two halves meet at the end)
    float st1[BN ? MB2 : 1], st2[BN ? MB2 : 1], stc = 0.0f;
    if (BN) {
#pragma unroll
        for (int mb2 = 0; mb2 < MB2; ++mb2) { st1[mb2] = 0.0f; st2[mb2] = 0.0f; }
    }

    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t node = tile * npt + r / k;
        const int slot = r % k;
        const bool live = node < N;
        const int32_t j = live ? nbr[node * k + slot] : -1;
        const bool valid = j >= 0;
        const int64_t ni = live ? node : N - 1, nj = valid ? j : ni;
        // B operands of the first product: feat[edge r][16 kk + 8 hh + 0..7], feat = [x_i || x_j - x_i]
        bf16x8 fi[KH], fd[KH];
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            const float4 a0 = x4[ni * (HIN / 4) + 4 * kk + 2 * hh], a1 = x4[ni * (HIN / 4) + 4 * kk + 2 * hh + 1];
            const float4 c0 = x4[nj * (HIN / 4) + 4 * kk + 2 * hh], c1 = x4[nj * (HIN / 4) + 4 * kk + 2 * hh + 1];
            const float xi[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float xj[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                fi[kk][u] = (short)bf16_rne(xi[u]);
                fd[kk][u] = (short)bf16_rne(xj[u] - xi[u]);
            }
        }
        // ---- first layer, 32 hidden channels at a time; the converted accumulators are the second layer's B operands
        bf16x8 h1f[MB1][2];
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4 *>(&S.b1[32 * mb + 8 * q + 4 * hh]);
                acc[4 * q] = bv.x; acc[4 * q + 1] = bv.y; acc[4 * q + 2] = bv.z; acc[4 * q + 3] = bv.w;
            }
#pragma unroll
            for (int kk = 0; kk < KS1; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S.w1[mb][kk][lane], kk < KH ? fi[kk] : fd[kk - KH], acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 8; ++u) h1f[mb][s][u] = (short)bf16_rne(elu1(acc[8 * s + u]));
        }
        // ---- second layer + aggregation, 32 output channels at a time ---------------------------------------------
        // Round 3: the second product is computed UN-transposed, z2 = h1 . W2^T (rows = edges, columns = channels) -- the
        // converted accumulators of the first product are the A operand just as well as the B operand (lane (r, hh) holds
        // edge r, hidden channels 8 hh + j either way), and the W2 fragments staged as "A of W2" are "B of W2^T" register
        // for register.  The accumulator then has the CHANNEL on the lane and the 32 edges in the registers (edge
        // (e & 3) + 8 (e >> 2) + 4 hh), so the reduction over a node's k edges is a v_max3 / add chain over k / 2 registers
        // of the lane plus one exchange between the two lane halves (v_permlane32_swap) -- ~12 vector instructions per 32
        // channels where the transposed form needed a DPP butterfly per accumulator register (16 x 8 = 128).
        const unsigned vmask = (unsigned)__ballot(valid);          // bit r: edge r of the tile is valid (lanes 0..31)
        const bool all_valid = vmask == 0xFFFFFFFFu;               // wave-uniform: every table without empty slots
        const unsigned vsh = vmask >> (4 * hh);
#pragma unroll
        for (int mb2 = 0; mb2 < MB2; ++mb2) {
            f32x16 acc;
            const float bv = S.b2[32 * mb2 + r];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = bv;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1f[ks >> 1][ks & 1], S.w2[mb2][ks][lane], acc, 0, 0, 0);
            // messages of channel 32 mb2 + r for the edges this lane holds
            float m[16], mn[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = act2 ? elu1(acc[e]) : acc[e];
                const bool ve = all_valid || ((vsh >> ((e & 3) + 8 * (e >> 2))) & 1u);
                if (BN && partial) {
                    const float mv = ve ? m0 : 0.0f;
                    st1[mb2] += mv;
                    st2[mb2] = __builtin_fmaf(mv, mv, st2[mb2]);
                }
                m[e] = ve ? m0 : (ADD ? 0.0f : ninf);
                if (BN && !ADD) mn[e] = ve ? -m0 : ninf;
            }
            // per node: registers [n k/2, (n+1) k/2) of both lane halves (k is wave-uniform: three straight-line forms)
            float tot[4], tot2[4];
            auto fold = [&](auto rpn_c) __attribute__((always_inline)) {
                constexpr int RPN = decltype(rpn_c)::value;      // registers per node and lane: k / 2
#pragma unroll
                for (int n = 0; n < 16 / RPN; ++n) {
                    float a = m[n * RPN], b = (BN && !ADD) ? mn[n * RPN] : ninf;
#pragma unroll
                    for (int e = 1; e < RPN; ++e) {
                        a = ADD ? a + m[n * RPN + e] : __builtin_fmaxf(a, m[n * RPN + e]);
                        if (BN && !ADD) b = __builtin_fmaxf(b, mn[n * RPN + e]);
                    }
                    tot[n] = a; tot2[n] = b;
                }
            };
            tot[0] = tot[1] = tot[2] = tot[3] = 0.0f; tot2[0] = tot2[1] = tot2[2] = tot2[3] = 0.0f;
            if (k == 16) fold(std::integral_constant<int, 8>{});
            else if (k == 8) fold(std::integral_constant<int, 4>{});
            else fold(std::integral_constant<int, 16>{});
            // exchange between the halves: after swap(V0, V1) lanes 0..31 hold (own V0, upper V0), lanes 32..63 (lower V1,
            // own V1): with V0 = node 2p, V1 = node 2p + 1 the lower half ends up with node 2p, the upper with node 2p + 1
            auto combine = [&](float v0, float v1, bool add) -> float {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
                const float x0 = __uint_as_float(sw[0]), x1 = __uint_as_float(sw[1]);
                return add ? x0 + x1 : __builtin_fmaxf(x0, x1);
            };
            const int pairs = npt >= 2 ? npt / 2 : 1;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p >= pairs) break;
                const float v0 = tot[npt >= 2 ? 2 * p : 0], v1 = tot[npt >= 2 ? 2 * p + 1 : 0];
                float res = combine(v0, v1, ADD);
                float res2 = 0.0f;
                if (BN && !ADD) res2 = combine(tot2[npt >= 2 ? 2 * p : 0], tot2[npt >= 2 ? 2 * p + 1 : 0], false);
                const int64_t onode = tile * npt + (npt >= 2 ? 2 * p + hh : 0);
                const bool writer = onode < N && (npt >= 2 || hh == 0);
                if (!BN) res = (!ADD && res == ninf) ? 0.0f : res;     // a node without any neighbour aggregates to 0 (R3)
                if (writer) {
                    out[onode * H2 + 32 * mb2 + r] = res;
                    if (BN && !ADD) out2[onode * H2 + 32 * mb2 + r] = -res2;
                }
            }
        }
        if (BN) {
            // valid edges per node from the tile's mask (wave-uniform)
            if (ADD && r < npt && hh == 0) {
                const int64_t onode = tile * npt + r;
                const unsigned bits = k == 32 ? vmask : ((vmask >> (r * k)) & ((1u << k) - 1u));
                if (onode < N) out2[onode] = (float)__popc(bits);
            }
            if (hh == 0) stc += valid ? 1.0f : 0.0f;
        }
    }
    if (BN && partial) {
        // per-WORKGROUP partials [2][H2] + count: the two lane halves (same channel, the other half of the edges) are added
        // first, then the four wavefronts in order through LDS (one partial per wavefront made the one-workgroup finalize
        // kernel walk 8192 of them: 1 ms)
        __shared__ float pst[4][2 * H2 + 4];
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int mb2 = 0; mb2 < MB2; ++mb2) {
            const auto sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(st1[mb2]), __float_as_uint(st1[mb2]), false, false);
            const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(st2[mb2]), __float_as_uint(st2[mb2]), false, false);
            if (hh == 0) {
                pst[wv][32 * mb2 + r] = __uint_as_float(sa[0]) + __uint_as_float(sa[1]);
                pst[wv][H2 + 32 * mb2 + r] = __uint_as_float(sb[0]) + __uint_as_float(sb[1]);
            }
        }
        const float cc = group_reduce<true>(stc, 32);
        if (lane == 0) pst[wv][2 * H2] = cc;
        __syncthreads();
        float *pb = partial + (int64_t)blockIdx.x * (2 * H2 + 4);
        for (int t = threadIdx.x; t <= 2 * H2; t += blockDim.x) pb[t] = ((pst[0][t] + pst[1][t]) + pst[2][t]) + pst[3][t];
    }
}

// (a, b) of the trailing BatchNorm from the wavefront partials (training: batch statistics over the valid edges, biased
// variance; running statistics updated like torch.nn.BatchNorm1d) or from the running statistics (eval).  One workgroup,
// sums in double in a fixed order.
__global__ __launch_bounds__(1024) void edge_mlp2_bn_finalize_kernel(const float *__restrict__ partial, int64_t nwaves, int H2,
                                                                    const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, float eps,
                                                                    float momentum, float *__restrict__ running_mean,
                                                                    float *__restrict__ running_var,
                                                                    int64_t *__restrict__ num_batches_tracked,
                                                                    int training, float *__restrict__ ab)
{
    __shared__ double red[3][16][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;     // 16 groups, each every 16th partial
    double mean = 0.0, var = 1.0;
    if (training) {
        double s1 = 0.0, s2 = 0.0, cn = 0.0;
        const int cc = c < H2 ? c : 0;
#pragma unroll 8   // independent loads: keep eight in flight (the sum order is unchanged)
        for (int64_t w = grp; w < nwaves; w += 16) {
            const float *pw = partial + w * (int64_t)(2 * H2 + 4);
            s1 += (double)pw[cc]; s2 += (double)pw[H2 + cc];
            cn += (double)pw[2 * H2];
        }
        red[0][grp][c] = s1; red[1][grp][c] = s2; red[2][grp][c] = cn;
        __syncthreads();
        if (grp != 0 || c >= H2) return;
        s1 = 0.0; s2 = 0.0; cn = 0.0;
        for (int q = 0; q < 16; ++q) { s1 += red[0][q][c]; s2 += red[1][q][c]; cn += red[2][q][c]; }
        const double n = cn > 0.0 ? cn : 1.0;
        mean = s1 / n;
        var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        if (running_mean) {
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    } else {
        if (grp != 0 || c >= H2) return;
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    const double a = (double)(gamma ? gamma[c] : 1.0f) / sqrt(var + (double)eps);
    ab[c] = (float)a;
    ab[H2 + c] = (float)((double)(beta ? beta[c] : 0.0f) - mean * a);
}

// out = a * aggregate + b per channel: sums take cnt * b, maxima become minima under a negative scale, nodes without
// any neighbour stay 0 (R3)
template <bool ADD>
__global__ __launch_bounds__(256) void edge_mlp2_bn_apply_kernel(const float *__restrict__ agg0, const float *__restrict__ agg1,
                                                                 const float *__restrict__ ab, int64_t N, int H2,
                                                                 float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * H2) return;
    const int64_t i = t / H2;
    const int c = (int)(t - i * H2);
    const float a = ab[c], b = ab[H2 + c];
    float v;
    if (ADD) {
        const float cnt = agg1[i];
        v = cnt > 0.0f ? __builtin_fmaf(a, agg0[t], cnt * b) : 0.0f;
    } else {
        const float mx = agg0[t];
        v = mx == -__builtin_inff() ? 0.0f : __builtin_fmaf(a, a >= 0.0f ? mx : agg1[t], b);
    }
    out[t] = v;
}

template <int HIN, int H1P, int H2>
int launch_edge_mlp2(const float *x, const int32_t *nbr, int64_t N, int k, const float *W1, const float *b1, int H1,
                     const float *W2, const float *b2, int act2, int aggr, float *out, hipStream_t st)
{
    const int npt = 32 / k;
    const int64_t ntiles = (N + npt - 1) / npt;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;     // persistent: the weights are staged once per workgroup
    if (aggr == 0)
        hipLaunchKernelGGL((edge_mlp2_kernel<HIN, H1P, H2, false>), dim3((unsigned)blocks), dim3(256), 0, st, x, nbr, N, k, W1,
                           b1, H1, W2, b2, act2, out);
    else
        hipLaunchKernelGGL((edge_mlp2_kernel<HIN, H1P, H2, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, nbr, N, k, W1,
                           b1, H1, W2, b2, act2, out);
    DMET_LAUNCH_CHECK("edge_mlp2_kernel");
    return 0;
}

inline int64_t edge_mlp2_blocks(int64_t N, int k)
{
    const int npt = 32 / k;
    const int64_t ntiles = (N + npt - 1) / npt;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    return blocks;
}

template <int HIN, int H1P, int H2>
int launch_edge_mlp2_bn(const float *x, const int32_t *nbr, int64_t N, int k, const float *W1, const float *b1, int H1,
                        const float *W2, const float *b2, int act2, int aggr, float *agg0, float *agg1, float *partial,
                        hipStream_t st)
{
    const int64_t blocks = edge_mlp2_blocks(N, k);
    if (aggr == 0)
        hipLaunchKernelGGL((edge_mlp2_kernel<HIN, H1P, H2, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, nbr, N, k,
                           W1, b1, H1, W2, b2, act2, agg0, agg1, partial);
    else
        hipLaunchKernelGGL((edge_mlp2_kernel<HIN, H1P, H2, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, nbr, N, k,
                           W1, b1, H1, W2, b2, act2, agg0, agg1, partial);
    DMET_LAUNCH_CHECK("edge_mlp2_kernel (BatchNorm form)");
    return 0;
}

}  // namespace
}  // namespace dmet

using namespace dmet;

extern "C" size_t dmet_edge_mlp2_bn_workspace_bytes(int64_t N, int H2)
{
    if (N <= 0 || H2 <= 0) return 0;
    // two aggregates [N][H2] (the second one: minima, or the per-node edge counts), wavefront partials, (a, b)
    return sizeof(float) * (2 * (size_t)N * H2 + (size_t)2048 * (2 * H2 + 4) + 2 * (size_t)H2) + 1024;
}

extern "C" int dmet_edge_mlp2_bn_bf16(const float *x, int64_t N, int Hin, const int32_t *nbr, int k, const float *W1,
                                      const float *b1, int H1, const float *W2, const float *b2, int H2, int act2, int aggr,
                                      const float *gamma, const float *beta, float eps, float momentum,
                                      float *running_mean, float *running_var, int64_t *num_batches_tracked, int training,
                                      float *out, void *ws, size_t ws_bytes, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647 / 64, "dmet_edge_mlp2_bn_bf16: N out of range");
    DMET_REQUIRE(dmet_edge_mlp2_supported(Hin, H1, H2, k), "dmet_edge_mlp2_bn_bf16: unsupported shape Hin=%d H1=%d H2=%d k=%d",
                 Hin, H1, H2, k);
    DMET_REQUIRE(aggr == 0 || aggr == 1, "dmet_edge_mlp2_bn_bf16: aggr must be 0 (max) or 1 (add)");
    DMET_REQUIRE(training || (running_mean && running_var), "dmet_edge_mlp2_bn_bf16: eval mode needs running statistics");
    DMET_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dmet_edge_mlp2_bn_bf16: running_mean/var go together");
    if (N == 0) return 0;
    DMET_REQUIRE(x && nbr && W1 && W2 && out && ws, "dmet_edge_mlp2_bn_bf16: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(out), "dmet_edge_mlp2_bn_bf16: x and out must be 16-B aligned");
    DMET_REQUIRE(ws_bytes >= dmet_edge_mlp2_bn_workspace_bytes(N, H2), "dmet_edge_mlp2_bn_bf16: workspace too small");
    hipStream_t st = as_stream(stream);
    float *agg0 = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u);
    float *agg1 = agg0 + (size_t)N * H2;
    float *partial = agg1 + (size_t)N * H2;
    const int64_t nwaves = edge_mlp2_blocks(N, k);      // one partial per workgroup
    float *ab = partial + (size_t)2048 * (2 * H2 + 4);
    float *pp = training ? partial : nullptr;
    int rc;
    if (Hin == 32 && H2 == 32) rc = launch_edge_mlp2_bn<32, 64, 32>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, agg0, agg1, pp, st);
    else if (Hin == 32 && H2 == 64) rc = launch_edge_mlp2_bn<32, 64, 64>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, agg0, agg1, pp, st);
    else if (H1 <= 96) rc = launch_edge_mlp2_bn<64, 96, 64>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, agg0, agg1, pp, st);
    else rc = launch_edge_mlp2_bn<64, 128, 64>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, agg0, agg1, pp, st);
    if (rc) return rc;
    hipLaunchKernelGGL(edge_mlp2_bn_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float *)partial, nwaves, H2, gamma, beta,
                       eps, momentum, running_mean, running_var, num_batches_tracked, training, ab);
    DMET_LAUNCH_CHECK("edge_mlp2_bn_finalize_kernel");
    const int64_t total = N * H2;
    if (aggr == 0)
        hipLaunchKernelGGL((edge_mlp2_bn_apply_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const float *)agg0, (const float *)agg1, (const float *)ab, N, H2, out);
    else
        hipLaunchKernelGGL((edge_mlp2_bn_apply_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           (const float *)agg0, (const float *)agg1, (const float *)ab, N, H2, out);
    DMET_LAUNCH_CHECK("edge_mlp2_bn_apply_kernel");
    return 0;
}

extern "C" int dmet_edge_mlp2_supported(int Hin, int H1, int H2, int k)
{
    if (!(k == 8 || k == 16 || k == 32)) return 0;
    if (Hin == 32 && H1 >= 1 && H1 <= 64 && (H2 == 32 || H2 == 64)) return 1;
    if (Hin == 64 && H1 >= 1 && H1 <= 128 && H2 == 64) return 1;
    return 0;
}

extern "C" int dmet_edge_mlp2_bf16(const float *x, int64_t N, int Hin, const int32_t *nbr, int k, const float *W1,
                                   const float *b1, int H1, const float *W2, const float *b2, int H2, int act2, int aggr,
                                   float *out, dmet_stream_t stream)
{
    DMET_REQUIRE(N >= 0 && N < (int64_t)2147483647 / 64, "dmet_edge_mlp2_bf16: N out of range");
    DMET_REQUIRE(dmet_edge_mlp2_supported(Hin, H1, H2, k), "dmet_edge_mlp2_bf16: unsupported shape Hin=%d H1=%d H2=%d k=%d", Hin,
                 H1, H2, k);
    DMET_REQUIRE(aggr == 0 || aggr == 1, "dmet_edge_mlp2_bf16: aggr must be 0 (max) or 1 (add)");
    if (N == 0) return 0;
    DMET_REQUIRE(x && nbr && W1 && W2 && out, "dmet_edge_mlp2_bf16: null pointer");
    DMET_REQUIRE(aligned16(x) && aligned16(out), "dmet_edge_mlp2_bf16: x and out must be 16-B aligned");
    hipStream_t st = as_stream(stream);
    if (Hin == 32 && H2 == 32) return launch_edge_mlp2<32, 64, 32>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, out, st);
    if (Hin == 32 && H2 == 64) return launch_edge_mlp2<32, 64, 64>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, out, st);
    if (H1 <= 96) return launch_edge_mlp2<64, 96, 64>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, out, st);
    return launch_edge_mlp2<64, 128, 64>(x, nbr, N, k, W1, b1, H1, W2, b2, act2, aggr, out, st);
}
