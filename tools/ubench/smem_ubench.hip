// How fast can a CU stream wave-uniform records through scalar loads?  Each wave walks `n` records of BYTES bytes (uniform
// index -> s_load_dwordxN) and does VALU_OPS dependent fp64 FMAs per record with the record as SGPR operand.
// Build: hipcc --offload-arch=gfx950 -O3 -o smem_ubench smem_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DW, int OPS>
__global__ __launch_bounds__(256) void k(const double *__restrict__ rec, int n, int stride_waves, double *out)
{
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    const double *p = rec + (size_t)(wave % stride_waves) * 64;   // different waves start at different records
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int j = 0; j < n; j++) {
        const double *r = p + (size_t)j * (DW / 2);
        double s = 0;
#pragma unroll
        for (int e = 0; e < DW / 2; e++) s += r[e];   // forces the whole record to be loaded (scalar adds are VALU here: uniform)
        // OPS independent-ish FMAs using the record
#pragma unroll
        for (int e = 0; e < OPS / 4; e++) {
            a0 = __builtin_fma(a0, r[0], s); a1 = __builtin_fma(a1, r[1 % (DW / 2)], s);
            a2 = __builtin_fma(a2, r[2 % (DW / 2)], s); a3 = __builtin_fma(a3, r[3 % (DW / 2)], s);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int DW, int OPS> void run(const double *rec, double *out, int n)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<DW, OPS>), dim3(blocks), dim3(256), 0, 0, rec, 100, 4096, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<DW, OPS>), dim3(blocks), dim3(256), 0, 0, rec, n, 4096, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double recs = (double)blocks * 4 * n;          // wave-records
    const double cyc_per_rec_per_simd = ms * 1e-3 * 2.4e9 * 1024 / recs;
    printf("record %3d B, %2d FMA+%d adds: %.3f ms, %.1f cycles(2.4GHz) per wave-record per SIMD, %.2f B/clk/CU\n", DW * 4, OPS, DW / 2, ms,
           cyc_per_rec_per_simd, DW * 4.0 * 4 / cyc_per_rec_per_simd);
}

int main()
{
    const size_t nd = (size_t)1 << 24;
    double *rec, *out;
    hipMalloc(&rec, nd * 8); hipMalloc(&out, 256 * 8 * 256 * 8);
    std::vector<double> h(nd, 1.0000001);
    hipMemcpy(rec, h.data(), nd * 8, hipMemcpyHostToDevice);
    const int n = 2000;
    run<8, 4>(rec, out, n); run<8, 8>(rec, out, n); run<8, 12>(rec, out, n); run<8, 16>(rec, out, n);
    run<16, 4>(rec, out, n); run<16, 8>(rec, out, n); run<16, 16>(rec, out, n); run<16, 24>(rec, out, n);
    run<4, 4>(rec, out, n); run<4, 8>(rec, out, n);
    return 0;
}
