// Micro-benchmark: do fp32 MFMA (32x32x2) and ordinary VALU work overlap on gfx950?
//   mode 0: MFMA only (chains of 16 dependent MFMAs)       mode 1: VALU only (N fma per MFMA slot)
//   mode 2: both, interleaved in one wave                   mode 3: both, alternating long phases (16 MFMA, then VALU)
// waves per SIMD set by the launch (blocks of 256 threads = 1 wave/SIMD each).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#ifdef USE_BF16
#define MFMA(a, b, acc) __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0)
#else
#define MFMA(a, b, acc) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#endif

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    float a = threadIdx.x * 0.001f, b = 1.0f + blockIdx.x * 1e-6f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(threadIdx.x + i); bb[i] = (short)(blockIdx.x + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + a;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = MFMA(a, b, acc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i % 8] = __builtin_fmaf(v[i % 8], b, a);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (MODE == 0 || MODE == 2) acc = MFMA(a, b, acc);
                if (MODE == 1 || MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) v[i % 8] = __builtin_fmaf(v[i % 8], b, a);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.0f;
    for (int e = 0; e < 16; ++e) s += acc[e];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char *name, int wps, float *out, int iters)
{
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s NV=%2d waves/SIMD=%d : %.3f ms  (%.1f cycles@2.1GHz per MFMA slot per wave)\n", name, NV, wps, ms,
           ms * 1e-3 * 2.1e9 / (iters * 16.0) / wps);
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 6>("mfma only", wps, out, iters);
        run<1, 6>("valu only", wps, out, iters);
        run<2, 6>("interleaved", wps, out, iters);
        run<3, 6>("phased (16 mfma | valu)", wps, out, iters);
        run<1, 14>("valu only", wps, out, iters);
        run<2, 14>("interleaved", wps, out, iters);
        run<3, 14>("phased (16 mfma | valu)", wps, out, iters);
    }
    return 0;
}
