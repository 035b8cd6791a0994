// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment:  hipcc --offload-arch=gfx950 -O2 mfma_f64_layout.hip
// D(16x16) = A(16x4) * B(4x16).  Checked: lane l holds A[l%16][l/16], B[l/16][l%16] and, in register r, D[4*r + l/16][l%16]  (found with one-hot probes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    v4f64 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * r + l / 16) * 16 + l % 16] = c[r];
}
int main()
{
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; i++) { hA[i] = 1.0 + 0.37 * i + 0.01 * i * i; hB[i] = 2.0 - 0.11 * i + 0.003 * i * i; }
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 256; i++) { double e = fabs(hD[i] - ref[i]) / fabs(ref[i]); if (e > worst) worst = e; }
    printf("max rel deviation from A*B under the hypothesised layout: %.3e  (%s)\n", worst, worst < 1e-14 ? "layout confirmed" : "layout WRONG");
    return worst < 1e-14 ? 0 : 1;
}
