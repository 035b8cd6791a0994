// fp64 VALU micro-benchmarks for gfx950: throughput of fma/mul/add/rcp/sqrt/rsq chains and accuracy of v_rcp_f64.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp64_ubench fp64_ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_thr(double *out, int iters, double seed)
{
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0;
    double a4 = a0 + 4.0, a5 = a0 + 5.0, a6 = a0 + 6.0, a7 = a0 + 7.0;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; i++) {
#define APPLY(v) \
        if (OP == 0) v = __builtin_fma(v, b, c); \
        else if (OP == 1) v = v * b; \
        else if (OP == 2) v = v + c; \
        else if (OP == 3) v = __builtin_amdgcn_rcp(v); \
        else if (OP == 4) v = __builtin_amdgcn_sqrt(v); \
        else if (OP == 5) v = __builtin_amdgcn_rsq(v); \
        else if (OP == 6) v = 1.0 / v; \
        else if (OP == 7) { float f = (float)v; f = __builtin_amdgcn_rcpf(f); v = (double)f; }
        APPLY(a0) APPLY(a1) APPLY(a2) APPLY(a3) APPLY(a4) APPLY(a5) APPLY(a6) APPLY(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_rcp_acc(const double *x, double *r0, double *r1, double *r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = x[i];
    double r = __builtin_amdgcn_rcp(s);
    r0[i] = r;
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    r1[i] = r;
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    r2[i] = r;
}

template <int OP> double run(const char *name, int blocks, int iters, double *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_thr<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_thr<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 8;
    double rate = ops / (ms * 1e-3);
    printf("%-10s %8.3f ms  %.3e lane-ops/s  = %.2f cycles per wave-instr per SIMD (at 2.4 GHz, 1024 SIMDs)\n", name, ms, rate,
           1024.0 * 2.4e9 * 64.0 / rate);
    return rate;
}

int main()
{
    int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
    double *d;
    CHK(hipMalloc(&d, (size_t)blocks * 256 * 8));
    int iters = 20000;
    run<0>("fma", blocks, iters, d);
    run<1>("mul", blocks, iters, d);
    run<2>("add", blocks, iters, d);
    run<3>("rcp", blocks, iters / 4, d);
    run<4>("sqrt", blocks, iters / 4, d);
    run<5>("rsq", blocks, iters / 4, d);
    run<6>("div", blocks, iters / 8, d);
    run<7>("cvt+rcpf", blocks, iters / 4, d);
    // accuracy
    int n = 1 << 20;
    std::vector<double> x(n), r0(n), r1(n), r2(n);
    for (int i = 0; i < n; i++) x[i] = pow(10.0, -3.0 + 20.0 * (double)rand() / RAND_MAX) * (1.0 + (double)rand() / RAND_MAX);
    double *dx, *d0, *d1, *d2;
    CHK(hipMalloc(&dx, n * 8)); CHK(hipMalloc(&d0, n * 8)); CHK(hipMalloc(&d1, n * 8)); CHK(hipMalloc(&d2, n * 8));
    CHK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rcp_acc, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    CHK(hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost));
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)x[i];
        m0 = fmax(m0, (double)fabsl((r0[i] - t) / t));
        m1 = fmax(m1, (double)fabsl((r1[i] - t) / t));
        m2 = fmax(m2, (double)fabsl((r2[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n", m0, m1, m2);
    return 0;
}
