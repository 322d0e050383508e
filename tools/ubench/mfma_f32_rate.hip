// Microbenchmark: what does the chip sustain on v_mfma_f32_32x32x2_f32 with NOTHING else in the way?  Every CU runs `waves` waves per SIMD, each
// issuing independent back-to-back MFMAs on 4 accumulators (the pattern of k_gemm_f32's inner loop) -- no LDS, no loads, no barriers.
// Prints TFLOP/s over the whole chip against the 157.3 nominal peak (256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz) and the s_memtime cycles
// per MFMA of one wave.   build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int waves = 1; waves <= 2; ++waves) {
        const int blocks = 256 * waves;                       // 256 threads = 4 waves = one per SIMD; `waves` blocks per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, 100, 0.5f, 0.25f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5f, 0.25f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double mfma = (double)blocks * 4 * iters * 64;
        const double tflops = mfma * 4096 / (ms * 1e-3) / 1e12;
        printf("%d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s = %.3f of 157.3; one wave: %.1f s_memtime ticks per MFMA (100 MHz ticks x24 = cycles at 2.4 GHz: %.1f)\n",
               waves, ms, tflops, tflops / 157.3, (double)c / (iters * 64.0), (double)c / (iters * 64.0) * 24.0);
    }
    return 0;
}
