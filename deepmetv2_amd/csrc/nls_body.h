// nls_body.h -- the wavefront-level body of the node-level dense layer of the fused EdgeConv (K2),
//   P = x.(W1-W2)^T + b,  Q = x.W2^T      (W = [W1 | W2] as torch's Linear(2H -> H).weight, /root/reference/model/graph_met_network.py:36)
// shared by node_linear_split_kernel (edgeconv.hip) and by the trailing "rider" workgroups of the kNN filter launch
// (knn.hip: the dense layer of a DynamicEdgeConv depends on x only, like the graph build, and fills the wavefront
// slots the build's last round leaves empty).  No workgroup barrier inside: a wavefront owns its tiles and its LDS.
#pragma once
#include <hip/hip_bf16.h>

#include "common.h"

namespace dmet {
namespace {

typedef float nls_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNlsTP = 36;                         // LDS row stride of the transposition tiles (SLICED only)
constexpr int kNlsLdsFloats = 2 * 32 * kNlsTP;     // per wavefront (SLICED only): one tile each for P and Q

// One wavefront computes [32 nodes] x [HOUT] for both P and Q with 32x32x2 fp32 MFMAs, tiles wave, wave + nwaves, ...
// MFMA operand maps (32x32x2): lane l holds A[row l&31][k l>>5], B[k l>>5][col l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// k-step s, half h  <->  input feature f = s + (HIN/2)*h, so a lane reads HIN/2 CONTIGUOUS floats of its node row.
// SLICED: P and Q are written slice-major, [HOUT/8][N][8] (the 8-channel slice of every node contiguous), which is how
// gather_max_lds_kernel's (event, slice) workgroups read them: their LDS staging and P reads become contiguous
// streams instead of 32-byte pieces of 128-byte rows.  `tp`: kNlsLdsFloats floats of LDS owned by this wavefront.
// AFFINE (round 3, the static flow: model/graph_met_network.py:65-66 -- no kNN prep launch for the transform to ride in):
// the rows are not read but FORMED, y = residual + BatchNorm(raw) with the expression of bn_apply_kernel (csrc/norm.hip:
// same bits), stored to aff.y (the block's output, needed by the residual branch and the backward) and fed to the matrix
// cores from the registers: one pass over the rows and one launch less than transform + dense layer.
struct NlsAffine {
    const float *raw, *res, *gamma, *beta, *mean, *invstd;
    float *y;
};

template <int HIN, int HOUT, bool SLICED, bool AFFINE = false>
__device__ __forceinline__ void node_linear_split_wave(const float *__restrict__ x, int64_t N,
                                                       const float *__restrict__ W, const float *__restrict__ bias,
                                                       float *__restrict__ P, float *__restrict__ Q,
                                                       float *__restrict__ tp, const int64_t wave, const int64_t nwaves,
                                                       const int lane, const NlsAffine aff = NlsAffine{})
{
    constexpr int KS = HIN / 2;     // k-steps
    constexpr int JT = HOUT / 32;   // output column tiles
    constexpr int TP = kNlsTP;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (N + 31) / 32;

    // B operands: column j = jt*32 + r of (W1-W2)^T and W2^T for feature f = s + KS*h
    float wd[JT][KS], w2[JT][KS], bj[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const float *wrow = W + (int64_t)(jt * 32 + r) * (2 * HIN);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float a = wrow[s + KS * h];
            const float b2 = wrow[HIN + s + KS * h];
            wd[jt][s] = a - b2;
            w2[jt][s] = b2;
        }
        bj[jt] = bias ? bias[jt * 32 + r] : 0.0f;
    }

    // the rows of the NEXT tile are loaded while the matrix products of the current one run (second session of round 2:
    // load / wait / 32 MFMAs / store left the matrix pipe 24 % busy with 43 % of the wave cycles in s_waitcnt)
    float4 nxt[KS / 4], nxr[AFFINE ? KS / 4 : 1];
    // AFFINE: this lane's KS features' constants (mean, gamma * invstd, beta), formed exactly as bn_apply_kernel forms them
    float4 cmu[AFFINE ? KS / 4 : 1], csc[AFFINE ? KS / 4 : 1], cbe[AFFINE ? KS / 4 : 1];
    if constexpr (AFFINE) {
#pragma unroll
        for (int s = 0; s < KS / 4; ++s) {
            const int f4 = (KS * h) / 4 + s;
            const float4 ga = reinterpret_cast<const float4 *>(aff.gamma)[f4], is = reinterpret_cast<const float4 *>(aff.invstd)[f4];
            cmu[s] = reinterpret_cast<const float4 *>(aff.mean)[f4];
            cbe[s] = reinterpret_cast<const float4 *>(aff.beta)[f4];
            csc[s] = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
        }
    }
    auto fetch = [&](const int64_t tile) __attribute__((always_inline)) {
        const int64_t node = tile * 32 + r;
        const int64_t nload = node < N ? node : N - 1;
        const float4 *src = reinterpret_cast<const float4 *>((AFFINE ? aff.raw : x) + nload * HIN + KS * h);
#pragma unroll
        for (int s = 0; s < KS / 4; ++s) nxt[s] = src[s];
        if constexpr (AFFINE) {
            if (aff.res) {
                const float4 *rs = reinterpret_cast<const float4 *>(aff.res + nload * HIN + KS * h);
#pragma unroll
                for (int s = 0; s < KS / 4; ++s) nxr[s] = rs[s];
            }
        }
    };
    if (wave < ntiles) fetch(wave);
    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        float a[KS];
#pragma unroll
        for (int s = 0; s < KS; s += 4) {
            float4 v = nxt[s / 4];
            if constexpr (AFFINE) {
                const float4 mu = cmu[s / 4], sc = csc[s / 4], be = cbe[s / 4];
                v.x = (v.x - mu.x) * sc.x + be.x; v.y = (v.y - mu.y) * sc.y + be.y;
                v.z = (v.z - mu.z) * sc.z + be.z; v.w = (v.w - mu.w) * sc.w + be.w;
                if (aff.res) {
                    const float4 rr = nxr[s / 4];
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                const int64_t node = tile * 32 + r;
                if (node < N) reinterpret_cast<float4 *>(aff.y + node * HIN + KS * h)[s / 4] = v;
            }
            a[s] = v.x; a[s + 1] = v.y; a[s + 2] = v.z; a[s + 3] = v.w;
        }
        if (tile + nwaves < ntiles) fetch(tile + nwaves);
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            nls_f32x16 accP, accQ;
#pragma unroll
            for (int e = 0; e < 16; ++e) { accP[e] = bj[jt]; accQ[e] = 0.0f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                accP = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wd[jt][s], accP, 0, 0, 0);
                accQ = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w2[jt][s], accQ, 0, 0, 0);
            }
            if constexpr (SLICED) {
                // slice-major rows are 32 bytes: go through LDS so that a lane stores 16 bytes and the 16 lanes of a
                // slice cover 8 consecutive nodes (256 contiguous bytes) instead of 32-byte pieces per store
                float *tP = tp, *tQ = tp + 32 * TP;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    tP[row * TP + r] = accP[e];
                    tQ[row * TP + r] = accQ[e];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int sl = lane >> 4, idx = lane & 15;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = p * 8 + (idx >> 1);
                    const int64_t n = tile * 32 + row;
                    if (n < N) {
                        const int64_t at = ((int64_t)(jt * 4 + sl) * N + n) * 8 + (idx & 1) * 4;
                        *reinterpret_cast<float4 *>(P + at) = *reinterpret_cast<const float4 *>(&tP[row * TP + sl * 8 + (idx & 1) * 4]);
                        *reinterpret_cast<float4 *>(Q + at) = *reinterpret_cast<const float4 *>(&tQ[row * TP + sl * 8 + (idx & 1) * 4]);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int64_t n = tile * 32 + row;
                    if (n < N) {
                        P[n * HOUT + jt * 32 + r] = accP[e];
                        Q[n * HOUT + jt * 32 + r] = accQ[e];
                    }
                }
            }
        }
    }
}


// bf16 variant of the node-level dense layer (BASELINE configs[2]): x and the split weights are rounded to bf16
// (RNE) and multiplied on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate); P stays fp32 (it is the
// node's own row, read once), Q is STORED as bf16 because it is the table that is gathered k times per node.
// Operand maps (32x32x16): lane l (r = l&31, h = l>>5) holds A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7.
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f)
{
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short *>(&h);
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

template <int HIN, int HOUT>
__device__ __forceinline__ void node_linear_split_bf16_wave(const float *__restrict__ x, int64_t N,
                                                            const float *__restrict__ W, const float *__restrict__ bias,
                                                            float *__restrict__ P, unsigned short *__restrict__ Qh,
                                                            const int64_t wave, const int64_t nwaves, const int lane)
{
    constexpr int KB = HIN / 16;    // k-blocks of 16 features
    constexpr int JT = HOUT / 32;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (N + 31) / 32;
    bf16x8 wd[JT][KB], w2[JT][KB];
    float bj[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const float *wrow = W + (int64_t)(jt * 32 + r) * (2 * HIN);
#pragma unroll
        for (int s = 0; s < KB; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = wrow[16 * s + 8 * h + j];
                const float b2 = wrow[HIN + 16 * s + 8 * h + j];
                wd[jt][s][j] = (short)f32_to_bf16_rne(a - b2);
                w2[jt][s][j] = (short)f32_to_bf16_rne(b2);
            }
        bj[jt] = bias ? bias[jt * 32 + r] : 0.0f;
    }
    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t node = tile * 32 + r;
        const int64_t nload = node < N ? node : N - 1;
        bf16x8 a[KB];
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            const float4 *src = reinterpret_cast<const float4 *>(x + nload * HIN + 16 * s + 8 * h);
            const float4 v0 = src[0], v1 = src[1];
            a[s][0] = (short)f32_to_bf16_rne(v0.x); a[s][1] = (short)f32_to_bf16_rne(v0.y);
            a[s][2] = (short)f32_to_bf16_rne(v0.z); a[s][3] = (short)f32_to_bf16_rne(v0.w);
            a[s][4] = (short)f32_to_bf16_rne(v1.x); a[s][5] = (short)f32_to_bf16_rne(v1.y);
            a[s][6] = (short)f32_to_bf16_rne(v1.z); a[s][7] = (short)f32_to_bf16_rne(v1.w);
        }
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            nls_f32x16 accP, accQ;
#pragma unroll
            for (int e = 0; e < 16; ++e) { accP[e] = bj[jt]; accQ[e] = 0.0f; }
#pragma unroll
            for (int s = 0; s < KB; ++s) {
                accP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], wd[jt][s], accP, 0, 0, 0);
                accQ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], w2[jt][s], accQ, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int64_t n = tile * 32 + row;
                if (n < N) {
                    P[n * HOUT + jt * 32 + r] = accP[e];
                    Qh[n * HOUT + jt * 32 + r] = f32_to_bf16_rne(accQ[e]);
                }
            }
        }
    }
}

}  // namespace
}  // namespace dmet
