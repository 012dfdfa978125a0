// Micro-benchmark: inner distance loop variants (scalar / packed / packed+ILP2 / TQ4), LDS-broadcast candidates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int D = 32, TILE = 32;

template <int VAR>
__global__ __launch_bounds__(64, (VAR == 3 ? 3 : 5)) void k(const float* __restrict__ x, float* __restrict__ out, int iters) {
    __shared__ float4 tile[(TILE + 2) * D / 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < (TILE + 2) * D / 4; i += 64) tile[i] = reinterpret_cast<const float4*>(x)[i];
    __syncthreads();
    float best0 = 1e30f, best1 = 1e30f, best2 = 1e30f, best3 = 1e30f;
    if (VAR == 0) {  // scalar, TQ=2, one candidate at a time
        float q0[D], q1[D];
        for (int d = 0; d < D; ++d) { q0[d] = x[(blockIdx.x * 64 + lane) % 1000 * D + d]; q1[d] = x[(blockIdx.x * 64 + lane + 7) % 1000 * D + d]; }
        for (int it = 0; it < iters; ++it)
            for (int c = 0; c < TILE; ++c) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < D / 4; ++c4) {
                    const float4 v = tile[c * (D / 4) + c4];
                    float df;
                    df = v.x - q0[4*c4+0]; a0 = __builtin_fmaf(df, df, a0); df = v.x - q1[4*c4+0]; a1 = __builtin_fmaf(df, df, a1);
                    df = v.y - q0[4*c4+1]; a0 = __builtin_fmaf(df, df, a0); df = v.y - q1[4*c4+1]; a1 = __builtin_fmaf(df, df, a1);
                    df = v.z - q0[4*c4+2]; a0 = __builtin_fmaf(df, df, a0); df = v.z - q1[4*c4+2]; a1 = __builtin_fmaf(df, df, a1);
                    df = v.w - q0[4*c4+3]; a0 = __builtin_fmaf(df, df, a0); df = v.w - q1[4*c4+3]; a1 = __builtin_fmaf(df, df, a1);
                }
                best0 = fminf(best0, a0); best1 = fminf(best1, a1);
            }
    } else if (VAR == 1 || VAR == 2) {  // packed TQ=2 ; VAR 2: two candidates interleaved
        f2 q[D];
        for (int d = 0; d < D; ++d) { q[d].x = x[(blockIdx.x * 64 + lane) % 1000 * D + d]; q[d].y = x[(blockIdx.x * 64 + lane + 7) % 1000 * D + d]; }
        for (int it = 0; it < iters; ++it)
            for (int c = 0; c < TILE; c += (VAR == 2 ? 2 : 1)) {
                f2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
                for (int c4 = 0; c4 < D / 4; ++c4) {
                    const float4 v = tile[c * (D / 4) + c4];
                    float4 w = v;
                    if (VAR == 2) w = tile[(c + 1) * (D / 4) + c4];
                    f2 df;
                    df = (f2){v.x, v.x} - q[4*c4+0]; a = __builtin_elementwise_fma(df, df, a);
                    if (VAR == 2) { df = (f2){w.x, w.x} - q[4*c4+0]; b = __builtin_elementwise_fma(df, df, b); }
                    df = (f2){v.y, v.y} - q[4*c4+1]; a = __builtin_elementwise_fma(df, df, a);
                    if (VAR == 2) { df = (f2){w.y, w.y} - q[4*c4+1]; b = __builtin_elementwise_fma(df, df, b); }
                    df = (f2){v.z, v.z} - q[4*c4+2]; a = __builtin_elementwise_fma(df, df, a);
                    if (VAR == 2) { df = (f2){w.z, w.z} - q[4*c4+2]; b = __builtin_elementwise_fma(df, df, b); }
                    df = (f2){v.w, v.w} - q[4*c4+3]; a = __builtin_elementwise_fma(df, df, a);
                    if (VAR == 2) { df = (f2){w.w, w.w} - q[4*c4+3]; b = __builtin_elementwise_fma(df, df, b); }
                }
                best0 = fminf(best0, a.x); best1 = fminf(best1, a.y);
                if (VAR == 2) { best2 = fminf(best2, b.x); best3 = fminf(best3, b.y); }
            }
    } else if (VAR == 3) {  // packed TQ=4 (two f2 chains per candidate)
        f2 q[D], r[D];
        for (int d = 0; d < D; ++d) { q[d].x = x[(blockIdx.x * 64 + lane) % 1000 * D + d]; q[d].y = x[(blockIdx.x * 64 + lane + 7) % 1000 * D + d];
                                      r[d].x = x[(blockIdx.x * 64 + lane + 3) % 1000 * D + d]; r[d].y = x[(blockIdx.x * 64 + lane + 11) % 1000 * D + d]; }
        for (int it = 0; it < iters; ++it)
            for (int c = 0; c < TILE; ++c) {
                f2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
                for (int c4 = 0; c4 < D / 4; ++c4) {
                    const float4 v = tile[c * (D / 4) + c4];
                    f2 df;
                    df = (f2){v.x, v.x} - q[4*c4+0]; a = __builtin_elementwise_fma(df, df, a); df = (f2){v.x, v.x} - r[4*c4+0]; b = __builtin_elementwise_fma(df, df, b);
                    df = (f2){v.y, v.y} - q[4*c4+1]; a = __builtin_elementwise_fma(df, df, a); df = (f2){v.y, v.y} - r[4*c4+1]; b = __builtin_elementwise_fma(df, df, b);
                    df = (f2){v.z, v.z} - q[4*c4+2]; a = __builtin_elementwise_fma(df, df, a); df = (f2){v.z, v.z} - r[4*c4+2]; b = __builtin_elementwise_fma(df, df, b);
                    df = (f2){v.w, v.w} - q[4*c4+3]; a = __builtin_elementwise_fma(df, df, a); df = (f2){v.w, v.w} - r[4*c4+3]; b = __builtin_elementwise_fma(df, df, b);
                }
                best0 = fminf(best0, a.x); best1 = fminf(best1, a.y); best2 = fminf(best2, b.x); best3 = fminf(best3, b.y);
            }
    }
    out[blockIdx.x * 64 + lane] = best0 + best1 + best2 + best3;
}

template <int VAR> void run(const char* name, int waves_per_simd, int queries_per_lane, float* dx, float* dout) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 100;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(64), 0, 0, dx, dout, iters);
    hipEventRecord(a); hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(64), 0, 0, dx, dout, iters); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    double pairs = (double)blocks * 64 * queries_per_lane * iters * TILE;   // (query,candidate) pairs
    double tflops = pairs * D * 3 / (ms * 1e-3) / 1e12;
    double cyc_per_cand_wave = ms * 1e-3 * 2.4e9 / (iters * TILE) ;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s(3 flop)  %.0f cycles(2.4GHz) per candidate per wave-slot\n", name, waves_per_simd, ms, tflops, cyc_per_cand_wave);
}
int main() {
    float *dx, *dout; std::vector<float> h(1000 * D + 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMalloc(&dx, h.size() * 4); hipMalloc(&dout, 256 * 4 * 16 * 64 * 4); hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 5; ++w) {
        run<0>("scalar TQ2", w, 2, dx, dout);
        run<1>("packed TQ2", w, 2, dx, dout);
        run<2>("packed TQ2 ILP2", w, 2, dx, dout);
        if (w <= 3) run<3>("packed TQ4", w, 4, dx, dout);
    }
    return 0;
}
