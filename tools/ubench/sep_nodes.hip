// Micro-benchmark: node sums of far lines as a matrix product on v_mfma_f64_16x16x4 (state-separable far wings).
//   F[state][node] = sum_line sum_{n=1..4} C_n[state][line] * w(node, line)^n,   w = 1 / (nu_node - nu_line)^2
// One wave = 64 nodes (4 sub-tiles of 16) x 16 states; per step 4 lines: every lane owns ONE (node, line) pair per sub-tile,
// forms w, w^2, w^3, w^4 (10 VALU instructions) and feeds four matrix instructions (one per term, K = the 4 lines).
// Compared with the scalar-load VALU loop of k_cheb_nodes (13 instructions per (node, line, state)).
//   hipcc --offload-arch=gfx950 -O3 sep_nodes.hip -o sep_nodes && ./sep_nodes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double rcp_fast(double s)
{
    double r = (double)__builtin_amdgcn_rcpf((float)s);
    double e = __builtin_fma(-s, r, 1.0);
    return __builtin_fma(r, e, r);
}

// Csep: [4 terms][L][64 states]   nul: [L]   nodes: [nI][64]   F: [nI][64 nodes][64 states]
__global__ __launch_bounds__(256) void k_sep(const double *__restrict__ nodes, const double *__restrict__ nul, const double *__restrict__ Csep,
                                             int L, int lines_per_interval, double *__restrict__ F)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int T = blockIdx.x, g = wv;                 // 4 waves = 4 state groups of one interval
    const int lr = lane & 15, lq = lane >> 4;
    double vn[4];
    for (int st = 0; st < 4; st++) vn[st] = nodes[(size_t)T * 64 + st * 16 + lr];
    v4f64 acc[4];
    for (int st = 0; st < 4; st++) acc[st] = v4f64{0, 0, 0, 0};
    const int j0 = (T * 37) % (L - lines_per_interval);            // this interval's line range
    const size_t LS = (size_t)L * 64;
#pragma unroll 2
    for (int j = j0; j < j0 + lines_per_interval; j += 4) {
        const double nl = nul[j + lq];                               // lane group lq owns line j + lq
        double a[4];
        for (int n = 0; n < 4; n++) a[n] = Csep[n * LS + (size_t)(j + lq) * 64 + g * 16 + lr];   // A[i = state lr][k = line lq]
        for (int st = 0; st < 4; st++) {
            const double dv = vn[st] - nl;
            const double w = rcp_fast(dv * dv);
            const double w2 = w * w, w3 = w2 * w, w4 = w2 * w2;
            acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], w, acc[st], 0, 0, 0);
            acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], w2, acc[st], 0, 0, 0);
            acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], w3, acc[st], 0, 0, 0);
            acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], w4, acc[st], 0, 0, 0);
        }
    }
    for (int st = 0; st < 4; st++)
        for (int r = 0; r < 4; r++)   // D[state 4r + lq][node lr]
            F[((size_t)T * 64 + st * 16 + lr) * 64 + g * 16 + 4 * r + lq] = acc[st][r];
}

// the classic form: one wave = 64 nodes x 1 state, line record by scalar loads, 13-instruction 2-term body
struct Hot { double nul, d, y2, p3; };
__global__ __launch_bounds__(256) void k_classic(const double *__restrict__ nodes, const Hot *__restrict__ hot, int L, int lines_per_interval,
                                                 int K, double *__restrict__ F)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nsb = K / 4;
    const int T = blockIdx.x / nsb, k = (blockIdx.x % nsb) * 4 + wv;
    const double v = nodes[(size_t)T * 64 + lane];
    const Hot *__restrict__ hk = hot + (size_t)k * L;
    const int j0 = (T * 37) % (L - lines_per_interval);
    double acc = 0.0;
#pragma unroll 4
    for (int j = j0; j < j0 + lines_per_interval; j++) {
        const Hot h = hk[j];
        const double dv = v - h.nul, x = dv * h.d, s = __builtin_fma(x, x, h.y2), u = rcp_fast(s);
        const double q = __builtin_fma(h.y2, -2.0, 3.75);
        acc += (h.p3 * u) * __builtin_fma(__builtin_fma(q, u, 1.5), u, 1.0);
    }
    F[((size_t)T * 64 + lane) * 64 + k] = acc;
}

int main()
{
    const int L = 50000, K = 64, nI = 1369, lpi = 300;   // ~ C3: 1369 intervals, ~300 own lines each
    std::vector<double> nul(L), nodes((size_t)nI * 64), C((size_t)4 * L * 64);
    std::vector<Hot> hot((size_t)K * L);
    for (int j = 0; j < L; j++) nul[j] = 1.0 + 2500.0 * j / L;
    for (int T = 0; T < nI; T++) for (int m = 0; m < 64; m++) nodes[(size_t)T * 64 + m] = 3000.0 + T * 0.01 + m * 1e-4;
    for (size_t i = 0; i < C.size(); i++) C[i] = 1e-3 * ((i * 2654435761u) % 1000) / 1000.0;
    for (size_t i = 0; i < hot.size(); i++) hot[i] = Hot{nul[i % L], 800.0, 0.5, 1e-20};
    double *dn, *dl, *dC, *dF; Hot *dh;
    (void)hipMalloc(&dn, nodes.size() * 8); (void)hipMalloc(&dl, nul.size() * 8); (void)hipMalloc(&dC, C.size() * 8);
    (void)hipMalloc(&dF, (size_t)nI * 64 * 64 * 8); (void)hipMalloc(&dh, hot.size() * sizeof(Hot));
    (void)hipMemcpy(dn, nodes.data(), nodes.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dl, nul.data(), nul.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dh, hot.data(), hot.size() * sizeof(Hot), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_sep, dim3(nI), dim3(256), 0, 0, dn, dl, dC, L, lpi, dF);
            else hipLaunchKernelGGL(k_classic, dim3(nI * (K / 4)), dim3(256), 0, 0, dn, dh, L, lpi, K, dF);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double triples = (double)nI * 64 * lpi * 64;   // (node, line, state)
            if (rep == 2) printf("%s: %.3f ms, %.3e (node,line,state) per s%s\n", which == 0 ? "separable / MFMA " : "classic / VALU   ", ms, triples / (ms * 1e-3),
                                 which == 0 ? "" : "");
        }
    }
    // do the two kernels overlap when launched on two streams?  (matrix pipe vs fp64 vector unit)
    {
        hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        double *dF2; (void)hipMalloc(&dF2, (size_t)nI * 64 * 64 * 8);
        hipEvent_t a0, a1, b1; (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventCreate(&b1);
        for (int mode = 0; mode < 2; mode++)
            for (int rep = 0; rep < 3; rep++) {
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(a0, s1);
                (void)hipStreamWaitEvent(s2, a0, 0);
                hipStream_t sb = mode == 0 ? s1 : s2;
                hipLaunchKernelGGL(k_sep, dim3(nI), dim3(256), 0, s1, dn, dl, dC, L, lpi, dF);
                hipLaunchKernelGGL(k_classic, dim3(nI * (K / 4)), dim3(256), 0, sb, dn, dh, L, lpi, K, dF2);
                (void)hipEventRecord(b1, sb);
                (void)hipStreamWaitEvent(s1, b1, 0);
                (void)hipEventRecord(a1, s1); (void)hipEventSynchronize(a1);
                float ms; (void)hipEventElapsedTime(&ms, a0, a1);
                if (rep == 2) printf("%s: %.3f ms for both\n", mode == 0 ? "one stream (back to back)" : "two streams             ", ms);
            }
    }
    const double mfma = (double)nI * 4 * (lpi / 4) * 16;   // matrix instructions
    printf("matrix instructions %.3e; at 47 TFLOP/s (2048 flop each): %.3f ms\n", mfma, mfma * 2048 / 47e12 * 1e3);
    return 0;
}
