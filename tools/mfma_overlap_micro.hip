// Does VALU work hide under bf16 MFMAs on gfx950?  Hand-pinned instruction streams (inline asm), s_memtime stamps.
//   stream per slot: [1 x v_mfma_f32_32x32x16_bf16 on accumulator (slot % NACC)] + [NV x independent VALU]
//   MODE 0: MFMA only   1: VALU only   2: both (MFMA, then its NV VALU)   3: phased (NM MFMAs, then NM*NV VALU)
// VALU flavours: KIND 0 = v_add_f32 on 8 independent registers, 1 = v_cmp_lt_f32 (SGPR pair) + v_addc_co_u32 pairs.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_micro.hip -o /tmp/mom && /tmp/mom
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, int NACC, int KIND>
__global__ __launch_bounds__(256, 2) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + (threadIdx.x & 3)); bb[i] = (short)(0x3f80 + (blockIdx.x & 1)); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x * 0.001f;
    unsigned m[4] = {0u, 0u, 0u, 0u};
    float tau = 3.5f + threadIdx.x * 0.01f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int NM = 12;
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < NM; ++s)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s % NACC]) : "v"(ab), "v"(bb));
#pragma unroll
            for (int s = 0; s < NM * NV; ++s) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[s % 8]) : "v"(tau));
                else {
                    unsigned long long c;
                    asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(c) : "v"(v[s % 8]), "v"(tau));
                    asm volatile("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(m[s % 4]) : "s"(c) : "vcc");
                    ++s;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NM; ++s) {
                if (MODE == 0 || MODE == 2)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s % NACC]) : "v"(ab), "v"(bb));
                if (MODE == 1 || MODE == 2) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(s * NV + i) % 8]) : "v"(tau));
                        else {
                            unsigned long long c;
                            asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(c) : "v"(v[(s * NV + i) % 8]), "v"(tau));
                            asm volatile("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(m[i % 4]) : "s"(c) : "vcc");
                            ++i;
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    s += (float)(m[0] + m[1] + m[2] + m[3]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NV, int NACC, int KIND>
void run(const char *name, int wps, float *out, unsigned long long *cyc, int iters)
{
    const int blocks = 256 * wps;
    hipLaunchKernelGGL((k<MODE, NV, NACC, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, NACC, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[4096];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
    const double per_slot = sum / (blocks * 4) / (iters * 12.0);
    printf("%-10s NV=%2d NACC=%d KIND=%d waves/SIMD=%d : %7.3f ms, %6.1f shader cycles per MFMA slot per wave (clock %.2f GHz)\n", name, NV, NACC,
           KIND, wps, ms, per_slot, sum / (blocks * 4) / (ms * 1e-3) / 1e9);
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * 256 * 512);
    hipMalloc(&cyc, sizeof(unsigned long long) * 4096);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 6, 1, 0>("mfma", wps, out, cyc, iters);
        run<0, 6, 2, 0>("mfma", wps, out, cyc, iters);
        run<1, 6, 2, 0>("valu", wps, out, cyc, iters);
        run<2, 6, 1, 0>("both", wps, out, cyc, iters);
        run<2, 6, 2, 0>("both", wps, out, cyc, iters);
        run<2, 4, 2, 0>("both", wps, out, cyc, iters);
        run<3, 6, 2, 0>("phased", wps, out, cyc, iters);
        run<1, 12, 2, 0>("valu", wps, out, cyc, iters);
        run<2, 12, 2, 0>("both", wps, out, cyc, iters);
        run<3, 12, 2, 0>("phased", wps, out, cyc, iters);
        run<1, 6, 2, 1>("valu", wps, out, cyc, iters);
        run<2, 6, 2, 1>("both", wps, out, cyc, iters);
        run<3, 6, 2, 1>("phased", wps, out, cyc, iters);
        run<1, 12, 2, 1>("valu", wps, out, cyc, iters);
        run<3, 12, 2, 1>("phased", wps, out, cyc, iters);
    }
    return 0;
}
