// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: 4 independent accumulators per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters)
{
    v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void kfma(double *out, int iters)
{
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_fma(a, c0, b); c1 = __builtin_fma(a, c1, b); c2 = __builtin_fma(a, c2, b); c3 = __builtin_fma(a, c3, b);
        c4 = __builtin_fma(a, c4, b); c5 = __builtin_fma(a, c5, b); c6 = __builtin_fma(a, c6, b); c7 = __builtin_fma(a, c7, b);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
int main()
{
    double *d; (void)hipMalloc(&d, 8 * 256 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;   // 256 CUs x 4 SIMDs: one block of 4 waves per CU and wave-per-SIMD multiple
        for (int which = 0; which < 2; which++) {
            if (which == 0) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10); else hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, d, 10);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, d, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double n = which == 0 ? 4.0 * iters : 8.0 * iters;        // instructions per wave
            const double fma_per_instr = which == 0 ? 1024.0 : 64.0;
            const double tflops = 2.0 * fma_per_instr * n * blocks * 4 / (ms * 1e-3) / 1e12;
            printf("%s  waves/SIMD %d: %.3f ms, %.1f ns per instruction per wave, %.1f TFLOP/s\n", which == 0 ? "mfma_f64_16x16x4" : "v_fma_f64        ", wps, ms,
                   ms * 1e6 / n, tflops);
        }
    }
    return 0;
}
