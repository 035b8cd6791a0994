// cs_kernels.h -- gfx950 kernels of the line-by-line hot path (fp64).
//   K1 k_prep      per-(node state, line) parameters            line_shapes.jl:107-132,144,255-257
//   K2 k_linesum   windowed line-shape sum over sorted lines    line_shapes.jl:53-87 (surf!), :366-392
//   K3 k_rt        tau (Lobatto), Planck, multi-stream sweeps,  discretized.jl:76-87,136-177,249-326; radiation.jl:48-54
//                  per-block nu-trapezoid partial sums          shared.jl:125-137, util.jl:26-33
//   K4 k_freduce   fixed-order reduction of the block partials
//   K5 k_flux      K3 + K4 with the last additions to the cross-sections on chip (interpolated wings, CIA, near-line plane)
// Paths are relative to the reference root.  Near a line the sum is elementwise on the vector unit; far from it the Voigt term is a
// short power series in 1/dnu^2 whose coefficients carry the state, so the sum over lines for 16 states is a matrix product on
// v_mfma_f64_16x16x4_f64 (k_cheb_nodes_mx, k_voigt_edge_mx), as is the carry of the node sums to the grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocprim/warp/warp_reduce.hpp>
#include <rocprim/warp/warp_scan.hpp>
#include "cs_faddeeva.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "cs_kernels.h is written for gfx950 (CDNA4: 64-wide waves, v_mfma_f64_16x16x4_f64, 160 KB of LDS per CU, 8 XCDs) -- build with --offload-arch=gfx950"
#endif

namespace csdev {

// src/constants.jl:1-26, verbatim (k is the CODATA-2014 value on purpose)
constexpr double kC = 299792458.0;
constexpr double kHp = 6.62607015e-34;
constexpr double kKb = 1.38064852e-23;
constexpr double kRgas = 8.31446262;
constexpr double kAtm = 101325.0;
constexpr double kNa = 6.02214076e23;
constexpr double kTref = 296.0;
constexpr double kTmin = 25.0, kTmax = 1000.0;
constexpr double kSqLn2 = 0.8325546111576977;     // sqrt(ln 2)            line_shapes.jl:4
constexpr double kOSqPiLn2 = 0.46971863934982566;  // 1/sqrt(pi/ln 2)       line_shapes.jl:3
constexpr double kC2 = 100.0 * kHp * kC / kKb;     // 100 h c / k           line_shapes.jl:5

enum { SH_VOIGT = 0, SH_LORENTZ = 1, SH_DOPPLER = 2, SH_PHCO2 = 3 };

// per-(state, line) parameters.  "hot" is what the far-wing loops read through scalar loads -- 32 bytes per line is
// the budget at which those loops stay VALU-bound (64-byte records made them SMEM-bound: profiles/r01_notes.md);
// "cold" is read only by the near-line code.
//   Voigt/PHCO2: hot = {nul, d = sqrt(ln2)/alpha, y^2, A*y/sqrt(pi)},  cold = {y = gamma*d, A = C*S(T)/sqrt(pi/ln2)/alpha}
//   Lorentz    : hot = {nul, gamma^2, C*S(T)*gamma/pi, 0}
//   Doppler    : hot = {nul, 1/alpha^2, C*S(T)/(alpha*sqrt(pi)), 0}
struct __attribute__((aligned(32))) LineHot { double nul, p1, p2, p3; };
struct __attribute__((aligned(16))) LineCold { double y, A; };
// mixed-precision variant (BASELINE configs[4]): far-wing record in fp32 -- the line position stays fp64 so that
// nu - nul is formed exactly; ay carries A*y/sqrt(pi) scaled by 2^100 (line strengths span 1e-38..1e-18)
struct __attribute__((aligned(16))) LineF32 { float d, y2, ay, c2; };   // 16 B per (node, line); nul comes from the fp64 line table
constexpr double kMixScale = 1.2676506002282294e30;      // 2^100
constexpr double kMixUnscale = 7.888609052210118e-31;    // 2^-100

struct GasDev {
    int64_t L;
    const double *nu, *S, *ga, *gs, *Epp, *na, *mu;
    const double *sref;  // S / [exp(-c2 E''/Tref) (1 - exp(-c2 nul/Tref))]: the state-independent part of scaleintensity
    const int16_t *iso;
    const int32_t *ncheb;
    const double *cheb;  // [niso][16]
    const uint8_t *gid;  // merged table of several gases of a column: member index of each line (NULL: one gas) -- selects the
                         // member's partial pressure and concentration in k_gas_setup; everything else of a record is per line
};

// line_shapes.jl:27-48
__host__ __device__ __forceinline__ double cheby_qrefq(double T, int n, const double *__restrict__ a)
{
    double tau = 2.0 * (T - kTmin) / (kTmax - kTmin) - 1.0;
    double c1 = 1.0, c2 = tau;
    double y = a[0] + a[1] * c2;
    for (int k = 2; k < n; k++) {
        double c3 = 2.0 * tau * c2 - c1;
        y += a[k] * c3;
        c1 = c2;
        c2 = c3;
    }
    return 1.0 / y;
}

// K1: one thread per (line j, chunk of CS_PREP_KC states), j in [jlo, jhi) = the lines some window of this wavenumber grid can
// reach (a nu-shard of a multi-GPU run needs only its part of the table); j fastest so the SoA line table is read coalesced -- once
// per chunk instead of once per state (round 2: one thread per (state, line), 184 MB of line-table fetches per launch at C3 for a
// 6 MB table) -- and the records of a state are written in 2 KB runs.  What depends on the state alone comes from the host:
// ln(Tref/T) and Qref/Q(T) of every isotopologue (line_shapes.jl:27-48), K x niso values.
#define CS_PREP_KC 8
struct PrepArgs {
    int shape, K;
    GasDev g;
    int64_t jlo, jhi;
    const double *Tk, *Pk, *Ppk, *scale;   // Ppk, scale: [members][mstride], element (m, k) at m * mstride + k
    int mstride;
    const double *lrt;     // [K] ln(Tref / T_k)
    const double *qrefq;   // [K][niso] Qref/Q(T_k) per isotopologue of the table (0 where there is no fit: refused on the host)
    int niso;
    LineHot *hot;
    LineCold *cold;
    LineF32 *hot32;
    double *phfac;    // PHCO2 fast path: [6][K][L] line factors exp(+-b_r(T) (nul - nu_c)), r = 1..3 (NULL: not needed)
    double nu_c;      // reference wavenumber of those factors (centre of the grid)
};
__device__ __forceinline__ void prep_body(unsigned bid, const PrepArgs &pa)
{
    const int shape = pa.shape, K = pa.K;
    const GasDev &g = pa.g;
    const int64_t jlo = pa.jlo, jhi = pa.jhi;
    const double *__restrict__ Tk = pa.Tk, *__restrict__ Pk = pa.Pk, *__restrict__ Ppk = pa.Ppk, *__restrict__ scale = pa.scale;
    LineHot *__restrict__ hot = pa.hot;
    LineCold *__restrict__ cold = pa.cold;
    LineF32 *__restrict__ hot32 = pa.hot32;
    const unsigned njb = (unsigned)((jhi - jlo + 255) / 256);   // blocks per chunk of states
    const int chunk = (int)(bid / njb);
    const int64_t j = jlo + (int64_t)(bid % njb) * 256 + threadIdx.x;
    if (j >= jhi) return;
    const int k0 = chunk * CS_PREP_KC, k1 = min(k0 + CS_PREP_KC, K);
    const double nul = g.nu[j];
    // scaleintensity, line_shapes.jl:107-123
    const double a = -kC2 * g.Epp[j];
    const double b = -kC2 * nul;
    const int I = g.iso[j];
    const double sref = g.sref[j], mu = g.mu[j], na = g.na[j], ga = g.ga[j], gs = g.gs[j];
    const size_t mo = g.gid ? (size_t)g.gid[j] * pa.mstride : 0;
    for (int k = k0; k < k1; k++) {
        const size_t idx = (size_t)k * g.L + j;   // records are addressed by the line's index in the full table
        const double T = Tk[k], P = Pk[k], Pp = Ppk[mo + k], C = scale ? scale[mo + k] : 1.0;
        const double n = exp(a / T) * (1.0 - exp(b / T));
        const double QrefQ = pa.qrefq[(size_t)k * pa.niso + (I - 1)];
        const double S = sref * QrefQ * n;   // the factor at Tref is folded into sref at upload (two exp and a divide less per record)
        // alphadoppler :144, gammalorentz :255-257; (Tref/T)^na as exp(na ln(Tref/T)): a third of the instructions of pow()
        const double alpha = (nul / kC) * sqrt(2.0 * kRgas * T / mu);
        const double gamma = exp(na * pa.lrt[k]) * (ga * (P - Pp) + gs * Pp) / kAtm;
        LineHot h;
        LineCold c;
        h.nul = nul;
        if (shape == SH_LORENTZ) {
            h.p1 = gamma * gamma; h.p2 = C * S * gamma / kPi; h.p3 = 0.0;
            c.y = gamma; c.A = C * S;
        } else if (shape == SH_DOPPLER) {
            h.p1 = 1.0 / (alpha * alpha); h.p2 = C * S / (alpha * 1.7724538509055159); h.p3 = 0.0;
            c.y = alpha; c.A = C * S;
        } else {
            const double beta = 1.0 / alpha;
            const double dd = kSqLn2 * beta;
            const double y = gamma * dd;
            const double A = C * (S * (kOSqPiLn2 * beta));
            const double y2 = y * y;
            h.p1 = dd; h.p2 = y2; h.p3 = A * y * kIsqPi;
            c.y = y; c.A = A;
        }
        hot[idx] = h;
        cold[idx] = c;
        if (pa.phfac) {   // chi(dnu) = exp(a_r - b_r |nu - nul|) in region r (line_shapes.jl:467-481) factorises into a per-lane and a per-line part
            const double B1 = 0.0888 - 0.16 * exp(-0.0041 * T), B2 = 0.0526 * exp(-0.00152 * T);
            const double dl = nul - pa.nu_c;
            const size_t KL = (size_t)K * g.L;
            pa.phfac[0 * KL + idx] = exp(B1 * dl);      pa.phfac[3 * KL + idx] = exp(-B1 * dl);
            pa.phfac[1 * KL + idx] = exp(B2 * dl);      pa.phfac[4 * KL + idx] = exp(-B2 * dl);
            pa.phfac[2 * KL + idx] = exp(0.0232 * dl);  pa.phfac[5 * KL + idx] = exp(-0.0232 * dl);
        }
        if (hot32) {
            LineF32 f;
            f.d = (float)h.p1; f.y2 = (float)h.p2; f.ay = (float)(h.p3 * kMixScale); f.c2 = (float)(3.75 - 2.0 * h.p2);
            hot32[idx] = f;
        }
    }
}

// line_shapes.jl:467-481 with the two temperature-only factors hoisted
__device__ __forceinline__ double chi_phco2(double dn, double B1, double B2)
{
    if (dn < 3.0) return 1.0;
    if (dn < 30.0) return exp(-B1 * (dn - 3.0));
    if (dn < 120.0) return exp(-B1 * 27.0 - B2 * (dn - 30.0));
    return exp(-B1 * 27.0 - B2 * 90.0 - 0.0232 * (dn - 120.0));
}

// issue priority of a kernel's waves against the waves of the other kernels on the same SIMD (s_setprio).  A step is three streams of
// kernels side by side; the stream that ends last on a long grid is the near-line one (k_voigt_sub, then k_voigt_near<0,1> once
// k_voigt_far has handed over its ranges), whose waves are chains of dependent gathers that lose every issue slot to the streaming
// kernels beside them.  With priority 3 for k_voigt_sub and k_voigt_near the bench column's step goes 2.00 -> 1.95 ms and its half
// 1.16 -> 1.14; on a quarter it is a tie and on an eighth (196 tiles) a loss (0.378 -> 0.387): the host asks for it from 512 tiles on
// (cs_set_tuning key 16).  Priorities for the other kernels (far, edge_mx, the node sums) changed nothing.
__device__ __forceinline__ void wave_prio(int p)
{
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
}


// K2 (generic form): block = 256 consecutive wavenumbers x one node state; every lane walks the block's union
// window of lines in ascending order (the order surf! sums in) with wave-uniform parameter loads.
//   sigma[k][i] (nu-fastest) is overwritten when accumulate == 0, else added to (second and later gases).
template <int SHAPE>
__global__ __launch_bounds__(256) void k_linesum(const double *__restrict__ nu, int64_t nnu, int64_t L,
                                                  const LineHot *__restrict__ hot, const LineCold *__restrict__ cold,
                                                  const int32_t *__restrict__ tileJ0, const int32_t *__restrict__ tileJ1,
                                                  double cut, const double *__restrict__ Tk, double base,
                                                  const double *__restrict__ extra, double *__restrict__ sigma,
                                                  int accumulate)
{
    const int tile = blockIdx.x, k = blockIdx.y;
    const int64_t i = (int64_t)tile * 256 + threadIdx.x;
    const int J0 = tileJ0[tile], J1 = tileJ1[tile];
    const LineHot *__restrict__ hk = hot + (size_t)k * L;
    const LineCold *__restrict__ ck = cold + (size_t)k * L;
    const double v = nu[i < nnu ? i : nnu - 1];
    double B1 = 0.0, B2 = 0.0;
    if (SHAPE == SH_PHCO2) {
        double T = Tk[k];
        B1 = 0.0888 - 0.16 * exp(-0.0041 * T);
        B2 = 0.0526 * exp(-0.00152 * T);
    }
    double acc = 0.0;
    for (int j = J0; j < J1; j++) {
        const LineHot h = hk[j];
        const double dv = v - h.nul;
        if (!(fabs(dv) > cut)) {  // cutline is a strict >, line_shapes.jl:10
            if (SHAPE == SH_LORENTZ) {
                acc += h.p2 / __builtin_fma(dv, dv, h.p1);
            } else if (SHAPE == SH_DOPPLER) {
                const double a2 = (dv * dv) * h.p1;
                if (a2 < 750.0) acc += h.p2 * exp(-a2);   // (beyond: an exact zero in fp64, as the reference's exp gives)
            } else if (SHAPE == SH_VOIGT) {
                const double x = dv * h.p1;
                const double s = __builtin_fma(x, x, h.p2);
                if (s >= kFarS) {
                    acc = __builtin_fma(h.p3, fad_far_core(h.p2, 1.0 / s), acc);
                } else {
                    const LineCold c = ck[j];
                    acc = __builtin_fma(c.A, s >= kMidS ? fad_mid(fabs(x), c.y) : fad_near(fabs(x), c.y), acc);
                }
            } else {
                const LineCold c = ck[j];
                const double chi = chi_phco2(fabs(dv), B1, B2);
                acc = __builtin_fma(c.A, fad_re(dv * h.p1, chi * c.y), acc);
            }
        }
    }
    if (i < nnu) {
        const size_t o = (size_t)k * nnu + i;
        double prev = accumulate ? sigma[o] : (base + (extra ? extra[o] : 0.0));
        sigma[o] = prev + acc;
    }
}


// ---- K2, Voigt fast path ---------------------------------------------------------------------------------------------
// One wave = 64 consecutive wavenumbers x one node state.  The wave's window of lines [W0,W1) (sorted by nul) is cut
// into five wave-uniform segments so that 95 % of the (nu, line) pairs run a branch-free far-wing body whose line
// parameters arrive through scalar loads:
//   [W0,a)  left edge  : far wing + cut-off predicate      (lines not within the cut-off of every lane)
//   [a,N0)  far left   : far wing, no predicate
//   [N0,N1) near zone  : lines within  d_A = 100*alpha_max/sqrt(ln2)  of the wave's span (x^2 < 1e4 possible)
//   [N1,b)  far right, [b,W1) right edge
// In the near zone every lane still takes the far-wing body where s >= 1e4 and records the index range of its own
// s < 1e4 (and s < 100) lines; two short per-lane loops then evaluate the continued-fraction and the near-centre
// forms only for those, so the expensive bodies run ~20 and ~4 times per (wave, state) instead of once per line.
struct WaveWin { int32_t W0, W1, E0, E1; };  // per 64-point tile: window, first line inside every lane's cut-off, one past the last
// per (state, tile): [M0,N0) and [N1,M1) mid-far lines (1e4 <= x^2 possible < 1e6), [N0,N1) near zone (x^2 < 1e4 possible);
// [Q0,M0) and [M1,Q1): lines far enough for the 3-term body (s >= 1e6) but not for dropping its y-dependent u^2 terms, which
// needs 15 y^2/s^3 < 1e-15 (y^2 bounded per window from the largest Lorentz and smallest Doppler width); outside [Q0,Q1): 2 terms + c2
struct __attribute__((aligned(16))) Zone { int32_t M0, N0, N1, M1, Q0, Q1, pad1, pad2; };

// N binary searches over the same sorted array side by side (one thread: their dependent loads overlap instead of queueing up --
// a zone thread is a chain of six to eight searches of ~10 loads each otherwise, and those chains are what k_gas_setup waits for).
// Query q looks in [lo[q], hi[q]) for the first index whose value is >= val[q] (bit q of upmask clear: lower bound) resp.
// > val[q] (set: upper bound); the answer comes back in lo[q].  lo[q] >= hi[q] on entry: query not wanted, lo[q] is returned.
template <int N>
__device__ __forceinline__ void search_many(const double *__restrict__ a, const double (&val)[N], unsigned upmask, int (&lo)[N], int (&hi)[N])
{
    for (;;) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < N; q++) any = any || lo[q] < hi[q];
        if (!any) break;
        double x[N];
        int m[N];
#pragma unroll
        for (int q = 0; q < N; q++) { m[q] = (lo[q] + hi[q]) >> 1; x[q] = lo[q] < hi[q] ? a[m[q]] : 0.0; }
#pragma unroll
        for (int q = 0; q < N; q++)
            if (lo[q] < hi[q]) {
                const bool right = ((upmask >> q) & 1u) ? x[q] <= val[q] : x[q] < val[q];
                if (right) lo[q] = m[q] + 1; else hi[q] = m[q];
            }
    }
}

// node-state dependent zone bounds, one thread per (state, tile)
struct ZoneArgs {
    const double *nu, *nul, *Tk, *gbound;
    const WaveWin *win;
    Zone *zones;
    int64_t nnu;
    int ntile, K, lorentz;   // lorentz: pure Lorentz profile -- one body everywhere, no Doppler core, no near zone
    double mu_min, mu_max, cut, far_s;
    double margin;           // an interval's interpolated set stays max(dA, margin x its half-width) away from it (kChebMargin)
};
// the zones of (state k, tile t)
__device__ __forceinline__ Zone zone_compute(const ZoneArgs &a, int k, int t)
{
    const double *__restrict__ nu = a.nu, *__restrict__ nul = a.nul, *__restrict__ Tk = a.Tk, *__restrict__ gbound = a.gbound;
    const WaveWin *__restrict__ win = a.win;
    const int64_t nnu = a.nnu;
    const double mu_min = a.mu_min, mu_max = a.mu_max, cut = a.cut, far_s = a.far_s;
    const int64_t i0 = (int64_t)t * 64, i1 = (i0 + 63 < nnu ? i0 + 63 : nnu - 1);
    const double vlo = nu[i0], vhi = nu[i1];
    const WaveWin w = win[t];
    if (a.lorentz) {
        // every line takes the (exact) Lorentz body, so the zones only say where the cut-off predicate is needed: the window is
        // split at the first line at or above the tile (an empty "near zone" there keeps the interpolated sets, which are clamped
        // against it, on their own sides); [Q0, split) and [split, Q1) run with the predicate, [E0, Q0) and [Q1, E1) -- lines
        // inside the cut-off of every lane -- without
        int lo = w.W0, hi = w.W1;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (nul[m] < vlo) lo = m + 1; else hi = m; }
        Zone z;
        z.N0 = z.N1 = lo;
        z.M0 = z.Q0 = min(lo, w.E1);
        z.M1 = z.Q1 = max(lo, w.E0);
        z.pad1 = z.pad2 = 0;
        return z;
    }
    // largest Doppler width any line of the window can have at this temperature (alphadoppler, line_shapes.jl:144)
    const double vth = sqrt(2.0 * kRgas * Tk[k]);
    const double amax = ((vhi + cut) / kC) * vth / sqrt(mu_min);
    const double dA = 100.0 * amax / kSqLn2 * (1.0 + 1e-6);  // |dnu| >= dA  =>  x^2 >= 1e4 (4-term series good to 1e-14)
    const double dAA = dA * sqrt(far_s * 1e-4);               // |dnu| >= dAA =>  x^2 >= far_s (>= 1e6: u^3 terms < 1e-16)
    // (inside [N0,N1) the far kernel uses the 6-term series down to s = 1e3, the near kernel takes over below)
    // y = gamma*sqrt(ln2)/alpha <= gbound[k]*sqrt(ln2)/alpha_min(window); gbound = host-side bound on gammalorentz
    const double vmin = vlo - cut;
    double y2b = 1e300;
    if (vmin > 0.0 && w.W1 > w.W0) {
        const double amin = (fmax(vmin, nul[w.W0]) / kC) * vth / sqrt(mu_max);
        const double yb = gbound[k] * kSqLn2 / amin;
        y2b = yb * yb;
    }
    // mode-0 body is exact to 1e-15 where s >= s0 = max(1e6, (1.5e16 y2b)^(1/3))  <=>  |dnu| >= dAA * sqrt(s0/1e6)
    const bool wantq = y2b > 60.0 && y2b < 1e290;
    const double dQ = wantq ? dA * sqrt(fmax(cbrt(1.5e16 * y2b), far_s) * 1e-4) : dAA;
    // M0 <= N0 <= N1 <= M1 and Q0 <= M0, Q1 >= M1 follow from dQ >= dAA >= dA: all six over the whole window, side by side
    const double sv[6] = {vlo - dAA, vlo - dA, vlo - dQ, vhi + dA, vhi + dAA, vhi + dQ};
    int slo[6] = {w.W0, w.W0, w.W0, w.W0, w.W0, w.W0}, shi[6] = {w.W1, w.W1, wantq ? w.W1 : w.W0, w.W1, w.W1, wantq ? w.W1 : w.W0};
    search_many<6>(nul, sv, 0x38u, slo, shi);
    Zone z;
    z.M0 = slo[0];
    z.N0 = max(slo[1], z.M0);
    z.N1 = max(slo[3], z.N0);
    z.M1 = max(slo[4], z.N1);
    if (y2b <= 60.0) {
        z.Q0 = z.M0; z.Q1 = z.M1;
    } else if (wantq) {
        z.Q0 = min(slo[2], z.M0);
        z.Q1 = max(slo[5], z.M1);
    } else {
        z.Q0 = w.W0; z.Q1 = w.W1;
    }
    z.pad1 = z.pad2 = 0;
    return z;
}
__device__ __forceinline__ void zones_body(unsigned bid, const ZoneArgs &a)
{
    const int idx = bid * blockDim.x + threadIdx.x;
    if (idx >= a.ntile * a.K) return;
    const int k = idx / a.ntile, t = idx - k * a.ntile;
    a.zones[idx] = zone_compute(a, k, t);
}

// constants of the series kept in VGPRs for the whole kernel (gfx950 VALU instructions take one constant-bus operand:
// the line parameter; a second literal would cost a v_mov per use).
struct FarK { double k1p5, k3p75, k12, km15, km105, k13p125, k210, km120; };
// (coefficients of the higher terms, a4 U8 and a5 U10 as polynomials in t: 59.0625 -787.5 2835 -3780 1680 | 324.84375 -6496.875
//  36382.5 -83160 83160 -30240)
// an opaque constant in a VGPR pair: the empty asm hides the value from the optimiser, which would otherwise rematerialise it as
// a literal (a v_mov per use) -- and costs no memory access (the first version loaded the table with eight volatile loads, i.e.
// eight serialised round trips at the start of every wave: most of a short wave's life on a sparse line table)
__device__ __forceinline__ double vgpr_const(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}
// the same in a scalar register pair, for constants that meet only vector operands (a gfx950 VALU instruction reads one scalar
// source): the higher coefficients of the near-zone series, which cost the hot kernel 22 VGPRs as vector constants and every wave
// 11 serialised loads as volatile memory operands
__device__ __forceinline__ double sgpr_const(double x)
{
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ FarK load_fark()
{
    FarK c;
    c.k1p5 = vgpr_const(1.5); c.k3p75 = vgpr_const(3.75); c.k12 = vgpr_const(12.0); c.km15 = vgpr_const(-15.0);
    c.km105 = vgpr_const(-105.0); c.k13p125 = vgpr_const(13.125); c.k210 = vgpr_const(210.0); c.km120 = vgpr_const(-120.0);
    return c;
}

// Far-wing term  A y/sqrt(pi) * u (1 + u p1(t) + u^2 p2(t) + u^3 p3(t)),  u = 1/s, t = y^2 u.  MODE:
//   0  s >= 1e6 and y^2 <= 60 : 1 + u (1.5 + u (3.75 - 2 y^2))                          13 VALU instructions
//   1  s >= 1e6               : 1 + u (1.5 + u (3.75 - 2 y^2 + t (12 t - 15)))           16
//   2  s >= 1e4               : all four terms                                           20
//   LOR: the Lorentz profile itself, (C S gamma/pi) / (dnu^2 + gamma^2) with hot = {nul, gamma^2, C S gamma/pi} -- exact at any
//        distance, 8 VALU instructions (lorentz!, line_shapes.jl:273,313-324)
template <bool PRED, int MODE, bool LOR = false>
__device__ __forceinline__ double far_term(const LineHot &h, double v, double cut, const FarK &c)
{
    const double dv = v - h.nul;
    if (LOR) {
        double r = h.p2 * rcp_nr1(__builtin_fma(dv, dv, h.p1));
        if (PRED) r = (fabs(dv) > cut) ? 0.0 : r;
        return r;
    }
    const double x = dv * h.p1;
    const double s = __builtin_fma(x, x, h.p2);
    const double u = rcp_fast(s);
    double P;
    if (MODE == 2) {
        const double t = h.p2 * u;
        const double p3 = __builtin_fma(__builtin_fma(__builtin_fma(c.km120, t, c.k210), t, c.km105), t, c.k13p125);
        const double p2 = __builtin_fma(__builtin_fma(c.k12, t, c.km15), t, c.k3p75);
        const double p1 = __builtin_fma(-2.0, t, c.k1p5);
        P = __builtin_fma(u, __builtin_fma(u, __builtin_fma(u, p3, p2), p1), 1.0);
    } else {
        double q = __builtin_fma(h.p2, -2.0, c.k3p75);  // 3.75 - 2 y^2 (wave-uniform)
        if (MODE == 1) {
            const double t = h.p2 * u;
            q = __builtin_fma(t, __builtin_fma(c.k12, t, c.km15), q);
        }
        P = __builtin_fma(__builtin_fma(q, u, c.k1p5), u, 1.0);
    }
    double r = (h.p3 * u) * P;
    if (PRED) r = (fabs(dv) > cut) ? 0.0 : r;
    return r;
}

template <bool PRED, int MODE, bool LOR = false>
__device__ __forceinline__ double far_segment(double acc, double v, const LineHot *__restrict__ hk, int j0, int j1, double cut,
                                              const FarK &c)
{
#pragma unroll 4
    for (int j = j0; j < j1; j++) acc += far_term<PRED, MODE, LOR>(hk[j], v, cut, c);
    return acc;
}
// the same lines from j1-1 down to j0 (lines right of nu: farthest = smallest terms first)
template <bool PRED, int MODE, bool LOR = false>
__device__ __forceinline__ double far_segment_rev(double acc, double v, const LineHot *__restrict__ hk, int j0, int j1, double cut,
                                                  const FarK &c)
{
#pragma unroll 4
    for (int j = j1 - 1; j >= j0; j--) acc += far_term<PRED, MODE, LOR>(hk[j], v, cut, c);
    return acc;
}

// fp32 far-wing body of the mixed-precision variant: 2 fp64 + 8 fp32 VALU instructions per line; partial sums of 4 terms
// are formed in fp32 and added to the fp64 accumulator.  Relative error of a term ~2e-7.
template <bool PRED, int MODE>
__device__ __forceinline__ float far_term32(double nul, const LineF32 h, double v, double cut)
{
    const double dvd = v - nul;
    const float dv = (float)dvd;
    const float x = dv * h.d;
    const float s = __builtin_fmaf(x, x, h.y2);
    const float u = __builtin_amdgcn_rcpf(s);
    float q = h.c2;
    if (MODE == 1) {
        const float t = h.y2 * u;
        q = __builtin_fmaf(t, __builtin_fmaf(12.0f, t, -15.0f), q);
    }
    const float P = __builtin_fmaf(__builtin_fmaf(q, u, 1.5f), u, 1.0f);
    float r = (h.ay * u) * P;
    if (PRED) r = (fabs(dvd) > cut) ? 0.0f : r;
    return r;
}

template <bool PRED, int MODE>
__device__ __forceinline__ double far_segment32(double acc, double v, const double *__restrict__ nul, const LineF32 *__restrict__ hk,
                                                int j0, int j1, double cut)
{
    int j = j0;
    for (; j + 3 < j1; j += 4) {   // 4 lines: one s_load_dwordx8 (positions) + one s_load_dwordx16 (parameters)
        float part = far_term32<PRED, MODE>(nul[j], hk[j], v, cut);
        part += far_term32<PRED, MODE>(nul[j + 1], hk[j + 1], v, cut);
        part += far_term32<PRED, MODE>(nul[j + 2], hk[j + 2], v, cut);
        part += far_term32<PRED, MODE>(nul[j + 3], hk[j + 3], v, cut);
        acc = __builtin_fma((double)part, kMixUnscale, acc);
    }
    float part = 0.0f;
    for (; j < j1; j++) part += far_term32<PRED, MODE>(nul[j], hk[j], v, cut);
    return __builtin_fma((double)part, kMixUnscale, acc);
}
template <bool PRED, int MODE>
__device__ __forceinline__ double far_segment32_rev(double acc, double v, const double *__restrict__ nul, const LineF32 *__restrict__ hk,
                                                    int j0, int j1, double cut)
{
    int j = j1 - 4;
    for (; j >= j0; j -= 4) {
        float part = far_term32<PRED, MODE>(nul[j + 3], hk[j + 3], v, cut);
        part += far_term32<PRED, MODE>(nul[j + 2], hk[j + 2], v, cut);
        part += far_term32<PRED, MODE>(nul[j + 1], hk[j + 1], v, cut);
        part += far_term32<PRED, MODE>(nul[j], hk[j], v, cut);
        acc = __builtin_fma((double)part, kMixUnscale, acc);
    }
    float part = 0.0f;
    for (j += 3; j >= j0; j--) part += far_term32<PRED, MODE>(nul[j], hk[j], v, cut);
    return __builtin_fma((double)part, kMixUnscale, acc);
}

// blockIdx.x -> block of 4/S wave tiles.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each), so blocks b and
// b+8 share an L2.  With a dense line table the records are a third of this kernel's bytes and plain block order makes every XCD
// fetch most of them (C3: FETCH 237 MiB per launch): give each XCD one contiguous eighth of the spectrum, whose overlapping line
// windows then stay in that L2 and leave HBM once (123 MiB = the algorithmic bytes).  That order costs time -- +3 % at C3, +14 %
// on the sparse tables of C5, and neither equal-cost stretches nor finer interleaving (2-8 blocks per XCD at a time) removed it
// (profiles/r02_notes.md) -- so the host chooses it only where the records matter (wave_windows: xc[0] >= 0).
// Speed/traffic only: any placement is correct.
__device__ __forceinline__ int tile_block(const int32_t *__restrict__ xc, int tiles_per_block, bool &valid)
{
    valid = true;
    if (xc[0] < 0) return blockIdx.x;   // plain order
    const int x = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int b0 = xc[x] / tiles_per_block, b1 = xc[x + 1] / tiles_per_block;
    valid = b0 + r < b1;
    return b0 + r;
}

// ---- K2c: far wings by spectral interpolation ---------------------------------------------------------------------------
// A line whose centre lies D >= 0.3 h beyond an interval of half-width h contributes a function of nu that is analytic inside
// a Bernstein ellipse of parameter rho >= 2.1 around the interval (its poles sit at nul +- i*gamma): the Chebyshev interpolant
// through CS_NC = 64 extrema reproduces it to rounding (error ~ rho^-63 < 1e-18; 1.2e-15 measured, tools/cheb_proto.py).
// So the sum over all such lines of an interval of N = 128 .. 2048 wavenumbers is evaluated at 64 nodes instead of N points
// and interpolated (an N x 64 matrix per interval) -- N/64 times fewer line evaluations.  Only lines inside the cut-off of
// EVERY point of the interval qualify (the cut-off makes the others discontinuous in nu); lines nearer than
// max(dA, 0.3 h) and the cut-off edges are left to the next smaller interval size, and finally to the per-point kernels.
#define CS_NC 64
#define CS_MAX_LEVEL 5
#define CS_MAX_ALEVEL 16  // levels one apply launch can carry (PHCO2 with its 500 cm^-1 cut-off: 8192 .. 64 points, some sizes with two node counts)
constexpr double kChebMargin = 0.3;   // default of ZoneArgs::margin
// per (state, interval): own set = [E0,Z0) U [Z1,E1), cut into the 2-/3-/4-term zones of the far body; [P0,P1) and [P2,P3)
// is the part of it the parent interval (next level up) has already summed.
// S0, S1: the lines below S0 and from S1 on are at least R4(state) = 133.6 sqrt(gamma_max^2 + 4.33 alpha_max^2) from the interval -- where
// the 4-term series in 1/dnu^2 holds for THIS state (k_cheb_nodes_mx sums them for the states of a group it holds for, k_cheb_nodes
// the rest: a group of 16 states spans a factor 3-7 in pressure, and one radius for all of it -- its widest line's -- left the
// lower-pressure states' lines to the vector unit)
struct __attribute__((aligned(16))) IZone { int32_t E0, Q0, M0, Z0, Z1, M1, Q1, E1, P0, P1, P2, P3, S0, S1, pad0, pad1; };

// nodes[T][m] = centre + h cos(pi m/63) and C[T][m][i] = l_m(nu_i): Lagrange basis of the extrema, barycentric form
// nc <= CS_NC nodes per interval (64 everywhere on the Voigt path; 16 or 32 where the lines of a set are many half-widths away:
// k_phco2_nodes)
__global__ __launch_bounds__(256) void k_cheb_setup(const double *__restrict__ nu, int64_t nnu, int itv, int nI, int nc,
                                                     double *__restrict__ nodes, double *__restrict__ Cm)
{
    const int T = blockIdx.x;
    const int64_t i0 = (int64_t)T * itv, i1 = (i0 + itv - 1 < nnu ? i0 + itv - 1 : nnu - 1);
    const double vlo = nu[i0], vhi = nu[i1];
    const double cen = 0.5 * (vlo + vhi), h = 0.5 * (vhi - vlo);
    __shared__ double xm[CS_NC], wm[CS_NC];
    if ((int)threadIdx.x < nc) {
        const double x = cen + h * cos(kPi * threadIdx.x / (nc - 1));
        xm[threadIdx.x] = x;
        nodes[(size_t)T * nc + threadIdx.x] = x;
    }
    __syncthreads();
    // barycentric weights of the nodes AS ROUNDED: w_m = 1 / prod_{j != m} (x_m - x_j).  The closed form (-1)^m {1/2,1,..,1,1/2}
    // belongs to the exact extrema; nodes near nu ~ 1e3 are rounded by ~1e-13 / h of the interval, and the mismatch shows up
    // as a 1e-13..1e-12 error of the interpolant (tools/cheb_proto.py).  Differences of nodes are exact in fp64.
    if ((int)threadIdx.x < nc) {
        double prod = 1.0;
        const double x = xm[threadIdx.x], ih = h > 0.0 ? 1.0 / h : 1.0;
        for (int j = 0; j < nc; j++)
            if (j != threadIdx.x) prod *= 2.0 * (x - xm[j]) * ih;   // factor 2: keeps the product near 1e2 instead of 1e-17
        wm[threadIdx.x] = 1.0 / prod;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < itv; p += blockDim.x) {
        const int64_t i = i0 + p;
        const double v = nu[i < nnu ? i : nnu - 1];
        double den = 0.0;
        int hit = -1;
        for (int m = 0; m < nc; m++) {
            const double d = v - xm[m];
            if (d == 0.0) hit = m;
            den += (d == 0.0) ? 0.0 : wm[m] / d;
        }
        for (int m = 0; m < nc; m++) {
            const double d = v - xm[m];
            const double c = (hit >= 0) ? (m == hit ? 1.0 : 0.0) : (wm[m] / d) / den;   // (one-point interval: all nodes coincide, hit = the last)
            Cm[((size_t)T * nc + m) * itv + p] = c;
        }
    }
}

// zones of the interpolation intervals, all levels in one launch.  A thread owns one (state, interval); the four bounds it
// needs from its parent interval (next larger size) are recomputed rather than read, so there is no order between levels.
struct IzParams {
    int nlev, nItot, l0;   // levels l0 .. nlev-1 are in use for this gas (a sparse line table skips the largest intervals)
    int itv[CS_MAX_LEVEL], nI[CS_MAX_LEVEL], ioff[CS_MAX_LEVEL];
    const WaveWin *iwin[CS_MAX_LEVEL];
};
// span, window and margin of (level l, interval T) for a state with thermal speed vth: E0..E1 = lines inside the cut-off of every
// point, dZ = how far the interpolated set stays from the interval; false if the set is empty
__device__ __forceinline__ bool izone_frame(const IzParams &P, int l, int T, const double *__restrict__ nu, int64_t nnu, double vth,
                                            double mu_min, double cut, double &vlo, double &vhi, double &dA, double &dZ, int &E0, int &E1,
                                            bool lorentz, double margin)
{
    const int itv = P.itv[l];
    const int64_t i0 = (int64_t)T * itv, i1 = (i0 + itv - 1 < nnu ? i0 + itv - 1 : nnu - 1);
    vlo = nu[i0];
    vhi = nu[i1];
    const WaveWin w = P.iwin[l][T];   // E0..E1: lines inside the cut-off of every point of the interval
    E0 = w.E0; E1 = w.E1;
    if (w.E1 <= w.E0) { E0 = E1 = w.E0; return false; }   // (then the parent has nothing either: the sets are nested)
    const double h = 0.5 * (vhi - vlo);
    const double amax = ((vhi + cut) / kC) * vth / sqrt(mu_min);
    dA = lorentz ? 0.0 : 100.0 * amax / kSqLn2 * (1.0 + 1e-6);   // (a Lorentz profile has no Doppler core to stay clear of)
    dZ = fmax(dA, margin * h);
    return true;
}
// the zones of (state k, interval q of the concatenated list; q >= ioff[l0])
__device__ __forceinline__ IZone izone_compute(const IzParams &P, const ZoneArgs &a, int k, int q)
{
    const double *__restrict__ nu = a.nu, *__restrict__ nul = a.nul, *__restrict__ Tk = a.Tk, *__restrict__ gbound = a.gbound;
    const int64_t nnu = a.nnu;
    const double mu_min = a.mu_min, mu_max = a.mu_max, cut = a.cut, far_s = a.far_s;
    int l = P.l0;
    while (l + 1 < P.nlev && q >= P.ioff[l + 1]) l++;
    const int T = q - P.ioff[l];
    const double vth = sqrt(2.0 * kRgas * Tk[k]);
    double vlo, vhi, dA, dZ;
    IZone z;
    if (!izone_frame(P, l, T, nu, nnu, vth, mu_min, cut, vlo, vhi, dA, dZ, z.E0, z.E1, a.lorentz, a.margin)) {
        z.Z0 = z.Z1 = z.Q0 = z.M0 = z.M1 = z.Q1 = z.P0 = z.P1 = z.P2 = z.P3 = z.S0 = z.S1 = z.E0;
        z.pad0 = z.pad1 = 0;
        return z;
    }
    const double dAA = dA * sqrt(far_s * 1e-4);
    const double vmin = vlo - cut;
    double y2b = 1e300;
    if (vmin > 0.0) {
        const double amin = (fmax(vmin, nul[z.E0]) / kC) * vth / sqrt(mu_max);
        const double yb = gbound[k] * kSqLn2 / amin;
        y2b = yb * yb;
    }
    const bool wantq = y2b > 60.0 && y2b < 1e290 && !a.lorentz;
    const double dQ = wantq ? dA * sqrt(fmax(cbrt(1.5e16 * y2b), far_s) * 1e-4) : dAA;
    // the parent interval (next level up): its own set is [qE0, qZ0) U [qZ1, qE1)
    bool par = false;
    double pvlo = 0.0, pvhi = 0.0, pdA, pdZ = 0.0;
    int qE0 = 0, qE1 = 0;
    if (l > P.l0) {
        int pshift = 0;
        for (int r = P.itv[l - 1] / P.itv[l]; r > 1; r >>= 1) pshift++;
        par = izone_frame(P, l - 1, T >> pshift, nu, nnu, vth, mu_min, cut, pvlo, pvhi, pdA, pdZ, qE0, qE1, a.lorentz, a.margin);
    }
    // this state's radius of the 4-term series in 1/dnu^2 (sepzones_body takes the group's pieces from these)
    const double R4 = [&] {
        const double amax = ((vhi + cut) / kC) * vth / sqrt(mu_min), gb = gbound[k];
        return 133.6 * sqrt(gb * gb + 4.33 * amax * amax) * (1.0 + 1e-6);
    }();
    // own Z0, Z1 (set stays dZ away), M0, M1 (4-term zone), Q0, Q1 (3-term zone), parent's Z0, Z1, series radius: ten searches side by side
    const double sv[10] = {vlo - dZ, vlo - dAA, vlo - dQ, pvlo - pdZ, vlo - R4, vhi + dZ, vhi + dAA, vhi + dQ, pvhi + pdZ, vhi + R4};
    int slo[10] = {z.E0, z.E0, z.E0, qE0, z.E0, z.E0, z.E0, z.E0, qE0, z.E0};
    int shi[10] = {z.E1, z.E1, wantq ? z.E1 : z.E0, par ? qE1 : qE0, z.E1, z.E1, z.E1, wantq ? z.E1 : z.E0, par ? qE1 : qE0, z.E1};
    search_many<10>(nul, sv, 0x3e0u, slo, shi);
    z.S0 = slo[4];                 // first line with nul >= vlo - R4: the lines below it are far enough on the left
    z.S1 = max(slo[9], z.S0);      // first line with nul >  vhi + R4: from here on far enough on the right
    z.pad0 = z.pad1 = 0;
    z.Z0 = slo[0];
    z.Z1 = max(slo[5], z.Z0);
    z.M0 = min(slo[1], z.Z0);
    z.M1 = max(slo[6], z.Z1);
    if (y2b <= 60.0) {
        z.Q0 = z.M0; z.Q1 = z.M1;
    } else if (wantq) {
        z.Q0 = min(slo[2], z.M0);
        z.Q1 = max(slo[7], z.M1);
    } else {
        z.Q0 = z.E0; z.Q1 = z.E1;
    }
    if (a.lorentz) { z.Q0 = z.M0 = z.Z0; z.Q1 = z.M1 = z.Z1; }   // one body: the whole set runs as "mode 0"
    // the parent's own set is nested in this one ([E0,E1) grows and [Z0,Z1) shrinks with the interval); clamp it anyway
    z.P0 = z.P1 = z.E0;
    z.P2 = z.P3 = z.E1;
    if (par) {
        const int qZ0 = slo[3], qZ1 = max(slo[8], qZ0);
        if (qZ0 > qE0) { z.P0 = min(max(qE0, z.E0), z.Z0); z.P1 = min(max(qZ0, z.P0), z.Z0); }
        if (qE1 > qZ1) { z.P2 = min(max(qZ1, z.Z1), z.E1); z.P3 = min(max(qE1, z.P2), z.E1); }
    }
    return z;
}
__device__ __forceinline__ void izones_body(unsigned bid, const IzParams &P, const ZoneArgs &a, IZone *__restrict__ iz)
{
    // zones of all levels live in one array [K][nItot]; level l starts at ioff[l]
    const int idx0 = bid * blockDim.x + threadIdx.x;
    const int q0 = P.ioff[P.l0], nq = P.nItot - q0;
    if (idx0 >= nq * a.K) return;
    const int k = idx0 / nq, q = q0 + (idx0 - k * nq);
    iz[(size_t)k * P.nItot + q] = izone_compute(P, a, k, q);
}

// K1 and the zone bounds of one gas in ONE launch: the three jobs are independent of each other (the first nb_zones blocks find the
// per-tile zones, the next ones the interval zones, the last nb_prep prepare the line records), so fusing them only removes two
// dependent kernel boundaries per gas -- which is what a nu-shard of a multi-GPU run, a few hundred tiles, spends its time on.
// The zone blocks go first: they are chains of dependent loads (binary searches) that the record blocks, a stream of stores,
// then run beside instead of before.
__global__ __launch_bounds__(256) void k_gas_setup(unsigned nb_prep, unsigned nb_zones, PrepArgs pa, ZoneArgs za, IzParams ip, IZone *__restrict__ iz)
{
    const unsigned nb_iz = gridDim.x - nb_prep - nb_zones;
    if (blockIdx.x < nb_zones) zones_body(blockIdx.x, za);
    else if (blockIdx.x < nb_zones + nb_iz) izones_body(blockIdx.x - nb_zones, ip, za, iz);
    else prep_body(blockIdx.x - nb_zones - nb_iz, pa);
}

// one wave = the 64 Chebyshev nodes of one interval x one node state: far-wing sums at the nodes -> F[interval][node][state].
// All levels run in one launch over the concatenated interval list (largest intervals, i.e. longest waves, first).
#define CS_KPAD 16   // F rows are padded to a multiple of 16 states (k_cheb_apply reads 16 at a time with scalar loads)
// Far lines whose 4-term series in 1/dnu^2 is exact for all 16 states of a state group (k_cheb_nodes_mx below) are summed
// on the matrix cores; per (state group, interval) their four pieces -- sub-ranges of [E0,P0), [P1,Z0), [Z1,P2), [P3,E1) common to
// the group's states -- are what this kernel then skips.
struct __attribute__((aligned(16))) SepZone { int32_t a[4], b[4], m[4]; };   // piece p = [a[p], b[p]); empty: a = b.  m[p] splits it: the lines
                                                                              // farther than it (below m on the left, from m on the right) need 3 terms
// the same for the per-point sum (k_voigt_edge_mx), per (state group, 64-point tile): the window ends [W0, eL) and [eR, W1), and the
// pieces [mL0, mL1), [mR0, mR1) between the interpolated sets and the near zone (empty: m.1 <= m.0)
// mL3, mR3 split the middle pieces like SepZone::m; far3 bit 0 / 1: the left / right window end needs 3 terms only; bit 2: the
// core takes 8 terms (and R is their radius, not that of 4)
// [cL, cR) (empty: cR <= cL): the core of the window -- everything between the matrix-core pieces, near zone included -- when the
// series radius R of the group is shorter than the tile: there k_voigt_sub sums the pairs with |dnu| < R on 16-point sub-tiles
// and k_voigt_edge_mx the pairs with |dnu| >= R (a second mask), and k_voigt_far leaves the core alone
struct __attribute__((aligned(16))) EdgeZone { int32_t eL, mL0, mL1, mR0, mR1, eR, mL3, mR3, far3, cL, cR, pad0; double R, pad1; };

// vector-unit node sum of one (interval, state) at the lane's node v: own set minus the parent's -- [E0,P0) U [P1,Z0) left of the
// interval, [Z1,P2) U [P3,E1) right of it -- each minus the piece [sa[p], sb[p]) the matrix cores take.  Both sides are summed from
// the far end towards the interval (increasing terms): the rounding error of a node sum then stays a few ulp of the sum itself,
// which the interpolation amplifies by up to (2.3/0.3)^2.
template <bool MIXED, bool LOR>
__device__ __forceinline__ double node_sum_valu(double v, const LineHot *__restrict__ hk, const LineF32 *__restrict__ hf,
                                                const double *__restrict__ gnul, const IZone &z, const int (&sa)[4], const int (&sb)[4],
                                                double cut, const FarK &c, int part = 0, int nparts = 1)
{
    // part / nparts: this wave's share of every window (short grids: the waves of a block split the lines of one (interval, state))
    double accL = 0.0, accR = 0.0;
    {
        const int wl[4] = {z.E0, sb[0], z.P1, sb[1]}, wh[4] = {sa[0], z.P0, sa[1], z.Z0};
        for (int cw = 0; cw < 4; cw++) {
            const int w0 = wl[cw], w1 = wh[cw];
            if (w0 >= w1) continue;
            const int p0 = w0 + (int)((int64_t)(w1 - w0) * part / nparts), p1 = w0 + (int)((int64_t)(w1 - w0) * (part + 1) / nparts);
            if (p0 >= p1) continue;
#define LO(x) max((x), p0)
#define HI(x) min((x), p1)
            if (MIXED) {
                accL = far_segment32<false, 0>(accL, v, gnul, hf, LO(z.E0), HI(z.Q0), cut);
                accL = far_segment32<false, 1>(accL, v, gnul, hf, LO(z.Q0), HI(z.M0), cut);
            } else {
                accL = far_segment<false, 0, LOR>(accL, v, hk, LO(z.E0), HI(z.Q0), cut, c);
                accL = far_segment<false, 1, LOR>(accL, v, hk, LO(z.Q0), HI(z.M0), cut, c);
            }
            accL = far_segment<false, 2, LOR>(accL, v, hk, LO(z.M0), HI(z.Z0), cut, c);
        }
    }
    {
        const int wl[4] = {sb[3], z.P3, sb[2], z.Z1}, wh[4] = {z.E1, sa[3], z.P2, sa[2]};   // far end first
        for (int cw = 0; cw < 4; cw++) {
            const int w0 = wl[cw], w1 = wh[cw];
            if (w0 >= w1) continue;
            const int p0 = w0 + (int)((int64_t)(w1 - w0) * part / nparts), p1 = w0 + (int)((int64_t)(w1 - w0) * (part + 1) / nparts);
            if (p0 >= p1) continue;
            if (MIXED) {
                accR = far_segment32_rev<false, 0>(accR, v, gnul, hf, LO(z.Q1), HI(z.E1), cut);
                accR = far_segment32_rev<false, 1>(accR, v, gnul, hf, LO(z.M1), HI(z.Q1), cut);
            } else {
                accR = far_segment_rev<false, 0, LOR>(accR, v, hk, LO(z.Q1), HI(z.E1), cut, c);
                accR = far_segment_rev<false, 1, LOR>(accR, v, hk, LO(z.M1), HI(z.Q1), cut, c);
            }
            accR = far_segment_rev<false, 2, LOR>(accR, v, hk, LO(z.Z1), HI(z.M1), cut, c);
#undef LO
#undef HI
        }
    }
    return accL + accR;
}

template <bool MIXED, bool LOR, int S = 1>   // S = 4 (short grids): the four waves of a block share ONE (interval, state), a quarter of every window each
__global__ __launch_bounds__(256) void k_cheb_nodes(const double *__restrict__ nodes, int64_t L, const LineHot *__restrict__ hot,
                                                     const LineF32 *__restrict__ hot32, const double *__restrict__ gnul,
                                                     const IZone *__restrict__ iz, int nItot, int q0, int q_acc, int K, int Kpad, double cut,
                                                     double *__restrict__ F, const SepZone *__restrict__ sep)
{
    // q_acc: intervals >= q_acc already hold the node sums of earlier gases of the column -- add to them.  The interpolation
    // is linear, so k_cheb_apply then carries the SUM over gases to the grid in one pass per level instead of one per (gas, level).
    // 1-D grid (an interval list can exceed the 65535 limit of gridDim.y): block = interval * nsb + state block, state fastest;
    // intervals q0 .. nItot-1 (the levels this gas uses)
    __shared__ double red[S > 1 ? 3 : 1][64];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nsb = S > 1 ? K : (K + 3) >> 2;
    const int T = q0 + (int)(blockIdx.x / nsb);
    // state block rotated by the interval: workgroups go round-robin to the 8 XCDs, and with the matrix cores taking most far lines
    // of the low-pressure states the work left here sits in the last state blocks -- unrotated, on half of the XCDs.
    // (Tried in round 3: one contiguous stretch of every level's intervals per XCD, cut at equal sums of window sizes, so that the ~5
    // neighbouring intervals that read a record share an L2 -- FETCH_SIZE 336 -> 291 MB here, 470 -> 448 MB in k_cheb_nodes_mx, but
    // 0.37 -> 0.43 ms: the in-flight footprint of an XCD's blocks is several times its 4 MB L2 either way, profiles/r03_notes.md.)
    const int k = S > 1 ? (int)((blockIdx.x % nsb + (sep ? T : 0)) % nsb)
                        : (int)((blockIdx.x % nsb + (sep ? T : 0)) % nsb) * 4 + wv;   // (unrotated, a state block stays on one XCD: its records stay in that L2)
    if (S == 1 && k >= K) return;
    const LineHot *__restrict__ hk = hot + (size_t)k * L;
    const LineF32 *__restrict__ hf = MIXED ? hot32 + (size_t)k * L : nullptr;
    const double v = nodes[(size_t)T * CS_NC + lane];
    const IZone z = iz[(size_t)k * nItot + T];
    const FarK c = load_fark();
    int sa[4] = {z.P0, z.Z0, z.Z1, z.P3}, sb[4] = {z.P0, z.Z0, z.Z1, z.P3};   // (no matrix-core pieces: empty ones at the window ends)
    if (!LOR && sep) {   // (also in the mixed-precision variant: what the matrix cores take stays fp64 -- they beat the fp32 vector bodies)
        const SepZone sz = sep[(size_t)(k >> 4) * nItot + T];
        // of the group's piece the matrix cores sum, for THIS state, only the lines beyond its own series radius (IZone::S0, S1): the
        // rest of the piece stays here
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int pa = p < 2 ? sz.a[p] : max(sz.a[p], z.S1), pb = p < 2 ? min(sz.b[p], z.S0) : sz.b[p];
            if (pb > pa) { sa[p] = pa; sb[p] = pb; }
        }
    }
    double acc = node_sum_valu<MIXED, LOR>(v, hk, hf, gnul, z, sa, sb, cut, c, S > 1 ? wv : 0, S);
    if (S > 1) {   // partial sums added in wave order (bitwise repeatable)
        if (wv > 0) red[wv - 1][lane] = acc;
        __syncthreads();
        if (wv > 0) return;
        acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
    }
    double *__restrict__ Fo = F + ((size_t)T * CS_NC + lane) * Kpad + k;
    *Fo = T >= q_acc ? *Fo + acc : acc;
}

// ---- K2d: state-separable far wings on the matrix cores ------------------------------------------------------------------------
// Far from the line the Voigt term is a power series in w = 1/dnu^2 whose coefficients carry ALL the state dependence
// (tools/voigt_series.py):  A K(x,y) = sum_{n=1..4} C_n w^n,  C_n = (A y/sqrt(pi)) c_n(y^2) / d^(2n),  c_1 = 1, c_2 = 3/2 - y^2,
// c_3 = 15/4 - 5 y^2 + y^4, c_4 = 105/8 - 105/4 y^2 + 21/2 y^4 - y^6; truncation below 1e-17 where
// |dnu| >= 133.6 sqrt(gamma^2 + 4.33 alpha^2).  The node sums of 16 states are then a matrix product
//     F[state][node] += sum_line sum_n C_n[state][line] * w[line][node]^n
// which v_mfma_f64_16x16x4 does with K = four LINES per instruction and one instruction per term: every lane owns one
// (node, line) pair of a 16-node sub-tile, forms w .. w^4 (10 VALU instructions) and, as (state, line), the four coefficients
// of its own record; 16 matrix instructions per 4 lines x 64 nodes x 16 states.  The loop runs at the matrix pipe's rate (0.92
// of it in tools/ubench/sep_nodes.hip: 2.5x the scalar-load VALU loop for the same triples) and leaves the vector unit to the
// kernels beside it.  Pieces from sepzones_body (k_mxzones).
// four binary searches over the same sorted array side by side (one thread: their dependent loads overlap instead of queueing up):
// r[q] = first index in [p, e) whose value is >= val[q] (q < 2: lower bound) resp. > val[q] (q >= 2: upper bound)
__device__ __forceinline__ void search4(const double *__restrict__ a, const double (&val)[4], int p, int e, int (&r)[4])
{
    int lo[4] = {p, p, p, p}, hi[4] = {e, e, e, e};
    while (lo[0] < hi[0] || lo[1] < hi[1] || lo[2] < hi[2] || lo[3] < hi[3]) {
        double x[4];
        int m[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { m[q] = (lo[q] + hi[q]) >> 1; x[q] = lo[q] < hi[q] ? a[m[q]] : 0.0; }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (lo[q] < hi[q]) {
                const bool right = q < 2 ? x[q] < val[q] : x[q] <= val[q];
                if (right) lo[q] = m[q] + 1; else hi[q] = m[q];
            }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = lo[q];
}

// validity of the truncated series (tools/voigt_series.py): relative truncation error <= 1e-17 where
//   4 terms: (y^2 + 3.0) / x^2 <= 5.6e-5   <=>  |dnu| >= 133.6 sqrt(gamma^2 + 4.33 alpha^2)
//   3 terms: (y^2 + 3.5) / x^2 <= 2.15e-6  <=>  |dnu| >= 682.0 sqrt(gamma^2 + 5.05 alpha^2)
//   8 terms: (y^2 + 4.4) / x^2 <= 7.5e-3   <=>  |dnu| >= 11.55 sqrt(gamma^2 + 6.35 alpha^2)
constexpr double kSep4 = 133.6, kSep3 = 682.0, kSep8 = 11.55;
typedef double v4f64_sep __attribute__((ext_vector_type(4)));
// one step of the matrix-core sums: 4 lines x 64 columns (nodes or points) x 16 states.  The lane's record as (state lr, line lq)
// gives the NT coefficients (A operands) and, as (column lr, line lq), the line position; vn[st] = the lane's column of sub-tile
// st.  MASK: w = 0 beyond the cut-off (line_shapes.jl:10).  4 NT matrix instructions.
template <int NT, int MASK, int NST = 4, int ST0 = 0, int NA = 4>   // MASK 1: w = 0 beyond the cut-off; 2: also inside the radius rin (those pairs are k_voigt_sub's)
__device__ __forceinline__ void sep_step(v4f64_sep (&acc)[NA], const double (&vn)[NA], const LineHot &h, bool valid, double cut, double rin = 0.0)
{   // the NST sub-tiles of 16 columns from ST0 on (4: a 64-point tile or the 64 nodes of an interval; 1, 2: a far piece on 16 or 32
    // nodes; 1 .. 3 from either end: a cut-off edge whose far lines reach only the first or last columns of the tile)
    static_assert(ST0 + NST <= NA, "sub-tile range");
    const double id2 = rcp_nr1(h.p1 * h.p1);
    const double y2 = h.p2;
    // a_n = (A y / sqrt(pi)) c_n(y^2) / d^(2n), c_n from tools/voigt_series.py (exact rationals, all representable)
    double a[NT];
    double Cn = valid ? h.p3 * id2 : 0.0;
    a[0] = Cn;
    Cn *= id2; a[1] = Cn * (1.5 - y2);
    Cn *= id2; a[2] = Cn * __builtin_fma(y2, y2 - 5.0, 3.75);
    if (NT >= 4) { Cn *= id2; a[3] = Cn * __builtin_fma(y2, __builtin_fma(y2, 10.5 - y2, -26.25), 13.125); }
    if (NT == 8) {
        Cn *= id2; a[4] = Cn * __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, y2 - 18.0, 94.5), -157.5), 59.0625);
        Cn *= id2; a[5] = Cn * __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, 27.5 - y2, -247.5), 866.25), -1082.8125), 324.84375);
        Cn *= id2; a[6] = Cn * __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, y2 - 39.0, 536.25), -3217.5), 8445.9375),
                                                               -8445.9375), 2111.484375);
        Cn *= id2; a[7] = Cn * __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, __builtin_fma(y2, 52.5 - y2, -1023.75), 9384.375),
                                                                                          -42229.6875), 88682.34375), -73901.953125), 15836.1328125);
    }
    // w = 1 / dnu^2 of the NST sub-tiles.  Where no dnu can vanish (MASK 0, 1: the lines of these pieces are at least a series radius
    // from every column) the reciprocals of two or four sub-tiles come from ONE reciprocal of their product -- 1/a = b/(ab) -- 10 resp.
    // 16 instructions instead of 7 per sub-tile (f32 seed + two Newton steps each).  The cores (MASK 2) hold lines INSIDE the tile
    // (a dnu may vanish): one reciprocal per sub-tile, as before.
    double dvv[NA], wv[NA];
#pragma unroll
    for (int st = ST0; st < ST0 + NST; st++) dvv[st] = vn[st] - h.nul;
    if (MASK != 2 && NST >= 2) {   // (A/B of two builds, round 5: bench column 1.919 -> 1.907 ms, a 1/8 shard 0.353 -> 0.343)
        double s2[NA];
#pragma unroll
        for (int st = ST0; st < ST0 + NST; st++) s2[st] = dvv[st] * dvv[st];
        auto rcp2 = [](double p) {   // v_rcp_f64 seed (the product of four dnu^2 can leave the f32 range of rcp_fast's) + two Newton steps
            double r = __builtin_amdgcn_rcp(p);
            r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
            return __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
        };
        if (NST == 4) {
            const double p01 = s2[ST0] * s2[ST0 + 1], p23 = s2[ST0 + 2] * s2[ST0 + 3];
            const double r = rcp2(p01 * p23);
            const double t01 = r * p23, t23 = r * p01;      // 1 / (s0 s1), 1 / (s2 s3)
            wv[ST0] = t01 * s2[ST0 + 1]; wv[ST0 + 1] = t01 * s2[ST0];
            wv[ST0 + 2] = t23 * s2[ST0 + 3]; wv[ST0 + 3] = t23 * s2[ST0 + 2];
        } else {
            const double r = rcp2(s2[ST0] * s2[ST0 + 1]);
            wv[ST0] = r * s2[ST0 + 1]; wv[ST0 + 1] = r * s2[ST0];
            if (NST == 3) wv[ST0 + 2] = rcp2(s2[ST0 + 2]);
        }
    } else {
#pragma unroll
        for (int st = ST0; st < ST0 + NST; st++) {
            const double s2 = dvv[st] * dvv[st];
            const double r = rcp_fast(s2);
            wv[st] = __builtin_fma(r, __builtin_fma(-s2, r, 1.0), r);   // second Newton step (powers of it are taken)
        }
    }
#pragma unroll
    for (int st = ST0; st < ST0 + NST; st++) {
        const double dv = dvv[st];
        double w = wv[st];
        if (MASK == 1) w = fabs(dv) > cut ? 0.0 : w;
        if (MASK == 2) w = (fabs(dv) > cut || fabs(dv) < rin) ? 0.0 : w;
        // (tried: skip the matrix instructions of a sub-tile whose (point, line) pairs are all masked, one ballot per sub-tile -- half of
        //  the columns at a cut-off edge add exact zeros -- 0.428 -> 0.429 ms: the branch costs what the instructions did)
        double wn = w;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[n], wn, acc[st], 0, 0, 0);
            if (n + 1 < NT) wn *= w;
        }
        if (NT == 8) __builtin_amdgcn_sched_barrier(0);   // (one sub-tile at a time: interleaving the four costs 60 more registers)
    }
}
// the lines [ja, jb) of one (state-row pointer hk) in steps of 4, ascending or descending; the load of step t + 1 is issued before the
// matrix instructions of step t and waited for after them
// [jlo_ok, jhi_ok): the lines of the run this LANE's state takes part in (its coefficients are zero for the others: a state whose own
// series radius excludes a line of the group's piece leaves it to the vector-unit kernel)
template <int NT, int MASK, int NST = 4, int ST0 = 0, int NA = 4>
__device__ __forceinline__ void sep_run(v4f64_sep (&acc)[NA], const double (&vn)[NA], const LineHot *__restrict__ hk, int ja, int jb, bool asc,
                                        int lq, double cut, double rin = 0.0, int jlo_ok = -0x7fffffff, int jhi_ok = 0x7fffffff)
{
    if (ja >= jb) return;
    const int nst = (jb - ja + 3) >> 2;
    const int b0 = asc ? ja + lq : jb - 4 + lq, db = asc ? 4 : -4;   // line of this lane group at step t: b0 + t db
    auto rec = [&](int t) { return hk[min(max(b0 + t * db, ja), jb - 1)]; };
    auto ok = [&](int t) { const int j = b0 + t * db; return j >= max(ja, jlo_ok) && j < min(jb, jhi_ok); };
    LineHot cur = rec(0);
    for (int t = 0; t < nst; t++) {
        const LineHot nxt = rec(t + 1);      // (past the end: a harmless re-read of an end record, never used)
        __builtin_amdgcn_sched_barrier(0);   // keep the load here: the scheduler would sink it behind the matrix instructions
        sep_step<NT, MASK, NST, ST0, NA>(acc, vn, cur, ok(t), cut, rin);
        __builtin_amdgcn_sched_barrier(0);   // ... and the wait for it there
        cur = nxt;
    }
}

struct SepArgs {
    const double *nodes, *nul, *gbound, *Tk;
    const IZone *iz;
    SepZone *out;
    int nItot, q0, K, ngrp;
    int min_states;   // a line joins a group's matrix-core piece when at least this many of its states are beyond their series radius
    double mu_min, cut;
};
// per (state group, interval): the four pieces common to the group's states, clipped to the distance at which the 4-term series
// holds for the widest line of the group (gbound: Lorentz width bound per state; Doppler width at the upper end of the window)
__device__ __forceinline__ void sepzones_body(unsigned bid, const SepArgs &a)
{
    const int idx = bid * blockDim.x + threadIdx.x;
    const int nq = a.nItot - a.q0;
    if (idx >= nq * a.ngrp) return;
    const int g = idx / nq, T = a.q0 + (idx - g * nq);
    const double vhi = a.nodes[(size_t)T * CS_NC], vlo = a.nodes[(size_t)T * CS_NC + CS_NC - 1];   // nodes run from the upper end down
    int lo[4] = {0, 0, 0, 0}, hi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    double R3 = 0.0;
    int E0 = 0, E1 = 0;
    int s0v[16], s1v[16], ns = 0;
#pragma unroll
    for (int kk = 0; kk < 16; kk++) { s0v[kk] = -0x7fffffff; s1v[kk] = 0x7fffffff; }   // (a group's missing tail states never count)
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        const int k = g * 16 + kk;
        if (k >= a.K) continue;
        const IZone z = a.iz[(size_t)k * a.nItot + T];
        lo[0] = max(lo[0], z.E0); hi[0] = min(hi[0], z.P0);
        lo[1] = max(lo[1], z.P1); hi[1] = min(hi[1], z.Z0);
        lo[2] = max(lo[2], z.Z1); hi[2] = min(hi[2], z.P2);
        lo[3] = max(lo[3], z.P3); hi[3] = min(hi[3], z.E1);
        E0 = z.E0; E1 = z.E1;
        s0v[kk] = z.S0; s1v[kk] = z.S1; ns = kk + 1;
        const double amax = ((vhi + a.cut) / kC) * sqrt(2.0 * kRgas * a.Tk[k]) / sqrt(a.mu_min);
        const double gb = a.gbound[k];
        R3 = fmax(R3, kSep3 * sqrt(gb * gb + 5.05 * amax * amax) * (1.0 + 1e-6));
    }
    // A matrix step costs the same whether one or sixteen states of the group can use a line (the others' coefficients are zero:
    // IZone::S0, S1), the vector unit per (state, line): a line joins the piece when at least min_states states are beyond their own
    // series radius -- ~500 cycles per line and 64 nodes for the group against ~80 per state on the vector unit.  The states' bounds are
    // ordered (sort them: 16 values): left pieces end at the min_states-th largest S0, right pieces start at the min_states-th smallest S1
    const int rk = min(max(a.min_states, 1), ns) - 1;
    int S0 = 0, S1 = 0;
    // rank selection with every index static (the arrays stay in registers; a sort with run-time indices puts them in scratch memory,
    // 0.027 -> 0.045 ms for this kernel): element i is the rk-th largest S0 when rk others precede it in descending order
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            c0 += (s0v[j] > s0v[i] || (s0v[j] == s0v[i] && j < i)) ? 1 : 0;
            c1 += (s1v[j] < s1v[i] || (s1v[j] == s1v[i] && j < i)) ? 1 : 0;
        }
        if (c0 == rk) S0 = s0v[i];
        if (c1 == rk) S1 = s1v[i];
    }
    S1 = max(S1, S0);
    int sr[4];
    {
        const double sv[4] = {vlo - R3, vlo - R3, vhi + R3, vhi + R3};
        search4(a.nul, sv, E0, E1, sr);
    }
    const int T0 = min(sr[1], S0), T1 = max(sr[3], S1);   // three terms do in [E0, T0) and [T1, E1): beyond the 3-term radius of the group's WIDEST state
    SepZone z;
    for (int p = 0; p < 4; p++) {
        int pa = lo[p], pb = hi[p];
        if (p < 2) pb = min(pb, S0); else pa = max(pa, S1);
        if (pb - pa < 8) { pa = 0; pb = 0; }   // (too short to be worth a wave's trip)
        z.a[p] = pa; z.b[p] = pb;
        int pm = p < 2 ? min(max(T0, pa), pb) : min(max(T1, pa), pb);
        if (p < 2 && pm - pa < 8) pm = pa;     // (a 3-term part that short is not worth its own steps: four terms for it too)
        if (p >= 2 && pb - pm < 8) pm = pb;
        z.m[p] = pm;
    }
    a.out[(size_t)g * a.nItot + T] = z;
}

#define CS_MX_PITCH 66   // LDS row pitch (doubles) of the partial sums: rows of the four lane groups land on different banks
// One block = one interval x one group of 16 states: every piece is cut into four runs of lines (multiples of 4), wave w takes
// run w -- far end first on both sides -- with the next record in flight while the 16 matrix instructions of a step issue; the
// four partial sums meet in LDS, are added in wave order and added to F (k_cheb_nodes, which runs first, has written the rest).
// (Tried: the vector part of the same (interval, group) in the same block, four states per wave, so that both pipes work side by
// side -- the matrix and vector phases do overlap, but the vector loop lives on eight waves per SIMD hiding its scalar loads, and
// this kernel's registers and LDS allow four: 0.94 ms for a quarter of the vector work, profiles/r02_notes.md.)
// Far pieces on fewer nodes.  The pieces p = 0 and p = 3 of an interval -- the lines beyond its parent's set -- are at least
// cut-off minus the parent's width away: 3.8 half-widths for the 256-point intervals of the bench grid, 11.6 for the 128-point ones,
// against 0.3 for the sets next to an interval.  Their sum converges like rho^-n (rho = x0 + sqrt(x0^2 - 1), x0 = 1 + distance in
// half-widths), so 32 resp. 16 nodes carry it to rounding where the near pieces need 64: two resp. one 16-node sub-tile per matrix
// step instead of four.  The values at those nodes are then carried to the interval's 64 nodes -- polynomial interpolation again, a
// fixed 64 x n matrix in the interval's own coordinate (R: [64][32] then [64][16], built by the host in extended precision), one more
// small matrix product per (interval, state group) -- and F, the apply kernel and everything after them never know.
struct MxFar { int nlev, ioff[CS_MAX_LEVEL + 1], nfar[CS_MAX_LEVEL]; const double *R; };   // nfar[l]: 16, 32 or 64 (= as before); R = NULL: 64 everywhere
template <int NST>
__device__ __forceinline__ void mx_far_pieces(v4f64_sep (&acc)[4], const SepZone &z, const LineHot *__restrict__ hk, double vlo, double vhi,
                                              int lr, int lq, int S0k, int S1k, const double *__restrict__ R, double (*__restrict__ tr)[CS_MX_PITCH],
                                              int wq = 0, int nq = 1)
{
    // wq / nq: this wave's run of every part when the four waves of a block share the item (nq = 4: short grids, where every interval
    // size is shared -- round 5; the carry below is linear, so each wave carries its own partial sum into its own accumulators)
    constexpr int n = 16 * NST;
    auto mine = [&](int pa, int pb, bool asc, int &ja, int &jb) {   // (as `quarter` in k_cheb_nodes_mx: runs of a multiple of 4 lines, wave 0 at the far end)
        if (nq == 1) { ja = pa; jb = pb; return; }
        const int run = ((pb - pa + 15) >> 4) << 2;
        ja = asc ? pa + wq * run : max(pb - (wq + 1) * run, pa);
        jb = asc ? min(ja + run, pb) : pb - wq * run;
    };
    int ja, jb;
    const double cen = 0.5 * (vlo + vhi), h = 0.5 * (vhi - vlo);
    double vf[NST];
    v4f64_sep af[NST];
#pragma unroll
    for (int st = 0; st < NST; st++) {
        vf[st] = cen + h * cospi((double)(st * 16 + lr) / (double)(n - 1));   // extrema of T_(n-1), from the upper end down like the 64
        af[st] = v4f64_sep{0.0, 0.0, 0.0, 0.0};
    }
    if (z.b[0] > z.a[0]) {   // left of the interval, ascending: the 3-term part (the far end) first
        if (z.m[0] > z.a[0]) { mine(z.a[0], z.m[0], true, ja, jb); sep_run<3, 0, NST, 0, NST>(af, vf, hk, ja, jb, true, lq, 0.0, 0.0, -0x7fffffff, S0k); }
        if (z.b[0] > z.m[0]) { mine(z.m[0], z.b[0], true, ja, jb); sep_run<4, 0, NST, 0, NST>(af, vf, hk, ja, jb, true, lq, 0.0, 0.0, -0x7fffffff, S0k); }
    }
    if (z.b[3] > z.a[3]) {   // right of it, descending
        if (z.b[3] > z.m[3]) { mine(z.m[3], z.b[3], false, ja, jb); sep_run<3, 0, NST, 0, NST>(af, vf, hk, ja, jb, false, lq, 0.0, 0.0, S1k); }
        if (z.m[3] > z.a[3]) { mine(z.a[3], z.m[3], false, ja, jb); sep_run<4, 0, NST, 0, NST>(af, vf, hk, ja, jb, false, lq, 0.0, 0.0, S1k); }
    }
    // acc[state][m] += sum_j af[state][j] R[m][j]: af goes through LDS from the D layout (state 4r + lq, node lr) to the A layout
    // (state lr, node lq of a group of four)
#pragma unroll
    for (int st = 0; st < NST; st++)
#pragma unroll
        for (int r = 0; r < 4; r++) tr[4 * r + lq][st * 16 + lr] = af[st][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const double *__restrict__ Rn = R + (NST == 2 ? 0 : CS_NC * 32);
#pragma unroll 2
    for (int kk = 0; kk < 4 * NST; kk++) {
        const double a = tr[lr][4 * kk + lq];
#pragma unroll
        for (int st2 = 0; st2 < 4; st2++)
            acc[st2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Rn[(size_t)(16 * st2 + lr) * n + 4 * kk + lq], acc[st2], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// (waves_per_eu: left alone the allocator puts the 32 accumulator registers into AGPRs but carries them around the loop's back edge in VGPRs
//  -- 32 v_accvgpr_write + 32 v_accvgpr_read per 4-line step, 64 of the step's 144 vector instructions, round 5's reading of the ISA; with
//  the register budget of three waves per SIMD it keeps them in VGPRs, as k_voigt_edge_mx has had since round 3)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_cheb_nodes_mx(const double *__restrict__ nodes, int64_t L, const LineHot *__restrict__ hot,
                                                       const SepZone *__restrict__ sep, int nItot, int q0, int nsplit, int K, int Kpad,
                                                       int ngrp, double *__restrict__ F, const IZone *__restrict__ iz, MxFar far)
{
    // the first nsplit intervals (the largest interval size in use: several hundred lines per piece) are shared by the four waves
    // of a block as described; the rest (a few dozen lines, ~10 steps) go one (interval, group) per wave -- no LDS, no barrier
    __shared__ double part[4][16][CS_MX_PITCH];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nb_split = nsplit * ngrp;
    const bool split = (int)blockIdx.x < nb_split;   // (block-uniform)
    int T, g;
    if (split) {
        T = q0 + (int)(blockIdx.x / ngrp);
        g = (int)((blockIdx.x % ngrp + T) % ngrp);   // (rotated: the groups differ in work and would alias with the XCD round-robin)
    } else {
        const int item = nb_split + ((int)blockIdx.x - nb_split) * 4 + wv;
        if (item >= (nItot - q0) * ngrp) return;
        T = q0 + item / ngrp;
        g = item % ngrp;
    }
    const SepZone z = sep[(size_t)g * nItot + T];
    if (!(z.b[0] > z.a[0] || z.b[1] > z.a[1] || z.b[2] > z.a[2] || z.b[3] > z.a[3])) return;   // (uniform per block when split, else per wave)
    {
        const int lr = lane & 15, lq = lane >> 4;
        const int kk = min(g * 16 + lr, K - 1);                       // (a group's tail states re-read the last one: never stored)
        const LineHot *__restrict__ hk = hot + (size_t)kk * L;
        // this lane's state: the lines below S0k (left pieces) and from S1k on (right pieces) are beyond ITS 4-term radius
        const int S0k = iz[(size_t)kk * nItot + T].S0, S1k = iz[(size_t)kk * nItot + T].S1;
        double vn[4];
#pragma unroll
        for (int st = 0; st < 4; st++) vn[st] = nodes[(size_t)T * CS_NC + st * 16 + lr];
        v4f64_sep acc[4];
#pragma unroll
        for (int st = 0; st < 4; st++) acc[st] = v4f64_sep{0.0, 0.0, 0.0, 0.0};
        // left pieces ascending, right pieces descending, the 3-term part of a piece (its far end) first: wave 0 owns the far end of
        // every part
        auto quarter = [&](int pa, int pb, bool asc, int &ja, int &jb) {   // this wave's run of [pa, pb): multiples of 4 lines
            if (!split) { ja = pa; jb = pb; return; }
            const int run = ((pb - pa + 15) >> 4) << 2;
            ja = asc ? pa + wv * run : max(pb - (wv + 1) * run, pa);
            jb = asc ? min(ja + run, pb) : pb - wv * run;
        };
        bool far_done = false;
        if (far.R) {   // (wave-uniform) the far pieces on 16 or 32 nodes where the interval's size allows; in a shared item every wave its own
                       // quarter of them (the largest size has no far pieces on fewer nodes: only short grids, where every size is shared, get here so)
            int l = 0;
            while (l + 1 < far.nlev && T >= far.ioff[l + 1]) l++;
            const int nf = far.nfar[l];
            if (nf < CS_NC && (z.b[0] > z.a[0] || z.b[3] > z.a[3])) {
                const double vhi = nodes[(size_t)T * CS_NC], vlo = nodes[(size_t)T * CS_NC + CS_NC - 1];
                if (nf == 16) mx_far_pieces<1>(acc, z, hk, vlo, vhi, lr, lq, S0k, S1k, far.R, part[wv], split ? wv : 0, split ? 4 : 1);
                else mx_far_pieces<2>(acc, z, hk, vlo, vhi, lr, lq, S0k, S1k, far.R, part[wv], split ? wv : 0, split ? 4 : 1);
                far_done = true;
            }
        }
        for (int pp = 0; pp < 4; pp++) {
            const int p = pp < 2 ? pp : 5 - pp;          // 0, 1, 3, 2
            const bool asc = pp < 2;
            if (z.b[p] <= z.a[p] || (far_done && (p == 0 || p == 3))) continue;
            int ja, jb;
            if (asc) {
                if (z.m[p] > z.a[p]) { quarter(z.a[p], z.m[p], true, ja, jb); sep_run<3, 0>(acc, vn, hk, ja, jb, true, lq, 0.0, 0.0, -0x7fffffff, S0k); }
                if (z.b[p] > z.m[p]) { quarter(z.m[p], z.b[p], true, ja, jb); sep_run<4, 0>(acc, vn, hk, ja, jb, true, lq, 0.0, 0.0, -0x7fffffff, S0k); }
            } else {
                if (z.b[p] > z.m[p]) { quarter(z.m[p], z.b[p], false, ja, jb); sep_run<3, 0>(acc, vn, hk, ja, jb, false, lq, 0.0, 0.0, S1k); }
                if (z.m[p] > z.a[p]) { quarter(z.a[p], z.m[p], false, ja, jb); sep_run<4, 0>(acc, vn, hk, ja, jb, false, lq, 0.0, 0.0, S1k); }
            }
        }
        if (!split) {   // D[state 4r + lq][node 16 st + lr] straight into F
#pragma unroll
            for (int st = 0; st < 4; st++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = g * 16 + 4 * r + lq;
                    if (k < K) F[((size_t)T * CS_NC + st * 16 + lr) * Kpad + k] += acc[st][r];
                }
            return;
        }
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[wv][4 * r + lq][st * 16 + lr] = acc[st][r];
    }
    __syncthreads();
    for (int s4 = 0; s4 < 4; s4++) {
        const int ks = 4 * wv + s4, k = g * 16 + ks;
        if (k >= K) break;
        double *__restrict__ Fo = F + ((size_t)T * CS_NC + lane) * Kpad + k;
        *Fo += ((part[0][ks][lane] + part[1][ks][lane]) + part[2][ks][lane]) + part[3][ks][lane];
    }
}

// sigma[k][i] (+)= sum over levels of  C_l[T_l][:, i] . F[T_l][:, k]  -- the interpolation as a small matrix product.
// One wave = one 64-point tile x 16 states: a column of C is loaded once and used for 16 states whose F values arrive as
// wave-uniform scalar operands.
struct ChebApply {
    int nlev, ngas;
    int shift[CS_MAX_ALEVEL];   // log2(interval size / 64)
    int ioff[CS_MAX_ALEVEL];    // offset of the level in the concatenated interval list
    int nc[CS_MAX_ALEVEL];      // nodes per interval of the level (CS_NC on the Voigt path)
    int noff[CS_MAX_ALEVEL];    // offset of the level's first node in F (ioff x CS_NC there)
    const double *Cm[CS_MAX_ALEVEL];
    const double *F[16];       // node sums of up to CS_MAX_GAS gases: C is read once for all of them
    int l0[16];                // first level each gas uses
};

// ---- cascade: the node sums of a level carried to the nodes of the next smaller one ----------------------------------------------
// The interpolant of an interval is a polynomial of degree 63; its values at the 64 nodes of a child interval define the same
// polynomial there.  So instead of carrying every level to the grid (levels x 64 x 64 multiply-adds per point and 16 states, and one
// pass over every level's [64][points] matrix), each level is added into the next smaller one -- F_child += Rc F_parent, a 64 x 64
// matrix per child interval, built like Cm from the nodes as rounded -- and only the smallest interval size is carried to the grid:
// 64 x 64 x (1 + 1/2 + 1/4 + ...) per point instead of 64 x 64 x levels.  Exact up to rounding (same polynomials).
// Rc[T][j][m] = l_j^parent(x_m^child)
__global__ __launch_bounds__(256) void k_cascade_setup(const double *__restrict__ nodes, int ioff_p, int ioff_c, int pshift, double *__restrict__ Rc)
{
    const int T = blockIdx.x;
    const double *__restrict__ xp = nodes + (size_t)(ioff_p + (T >> pshift)) * CS_NC, *__restrict__ xc = nodes + (size_t)(ioff_c + T) * CS_NC;
    __shared__ double xm[CS_NC], wm[CS_NC];
    if (threadIdx.x < CS_NC) xm[threadIdx.x] = xp[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < CS_NC) {   // barycentric weights of the parent's nodes as rounded (k_cheb_setup)
        const double h = 0.5 * (xm[0] - xm[CS_NC - 1]), ih = h > 0.0 ? 1.0 / h : 1.0, x = xm[threadIdx.x];
        double prod = 1.0;
        for (int j = 0; j < CS_NC; j++)
            if (j != (int)threadIdx.x) prod *= 2.0 * (x - xm[j]) * ih;
        wm[threadIdx.x] = 1.0 / prod;
    }
    __syncthreads();
    const int m = threadIdx.x & (CS_NC - 1), jq = threadIdx.x >> 6;   // child node m, a quarter of the parent's nodes
    const double v = xc[m];
    double den = 0.0;
    int hit = -1;
    for (int j = 0; j < CS_NC; j++) {
        const double d = v - xm[j];
        if (d == 0.0) hit = j;
        den += (d == 0.0) ? 0.0 : wm[j] / d;
    }
    for (int j = jq * 16; j < jq * 16 + 16; j++) {
        const double d = v - xm[j];
        Rc[((size_t)T * CS_NC + j) * CS_NC + m] = (hit >= 0) ? (j == hit ? 1.0 : 0.0) : (wm[j] / d) / den;
    }
}
// one wave = one child interval x 16 states: D(16 child nodes x 16 states) += A(16 child nodes x 4 parent nodes) B(4 parent nodes x
// 16 states), operands as in k_cheb_apply_mfma -- both loads and the read-modify-write of F are 128-byte runs (F is state-fastest)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_cheb_cascade(const double *__restrict__ Rc, double *__restrict__ F, int ioff_p, int ioff_c, int pshift,
                                                      int nIc, int Kpad, int nst)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int item = (int)blockIdx.x * 4 + wv;
    const int T = item / nst, sidx = item - T * nst;
    if (T >= nIc) return;
    const int lr = lane & 15, lq = lane >> 4;
    const double *__restrict__ Fp = F + (size_t)(ioff_p + (T >> pshift)) * CS_NC * Kpad + (size_t)sidx * 16 + lr;
    double *__restrict__ Fc = F + (size_t)(ioff_c + T) * CS_NC * Kpad + (size_t)sidx * 16 + lr;
    const double *__restrict__ R = Rc + (size_t)T * CS_NC * CS_NC + lr;
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 acc[4];
#pragma unroll
    for (int st = 0; st < 4; st++) acc[st] = v4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < CS_NC / 4; kk++) {
        const int j = 4 * kk + lq;
        const double b = Fp[(size_t)j * Kpad];
#pragma unroll
        for (int st = 0; st < 4; st++) acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(R[(size_t)j * CS_NC + 16 * st], b, acc[st], 0, 0, 0);
    }
#pragma unroll
    for (int st = 0; st < 4; st++)
#pragma unroll
        for (int r = 0; r < 4; r++) Fc[(size_t)(16 * st + 4 * r + lq) * Kpad] += acc[st][r];
}

// The same contraction on the matrix cores: v_mfma_f64_16x16x4_f64 computes D(16 states x 16 nu) += A(16 states x 4 nodes) *
// B(4 nodes x 16 nu).  Operand layout on gfx950 (tools/ubench/mfma_f64_layout.hip): lane l holds A[l%16][l/16], B[l/16][l%16] and,
// in register r, D[4r + l/16][l%16] -- with states as the rows of D a store of one register is four 128-byte runs of consecutive
// wavenumbers, and both operand loads are 128-byte runs as well (F is state-fastest, C is nu-fastest).
// One wave = one 64-point tile x NSUB*16 states: NSUB A-operands (F) and 4 B-operands (C) feed 4*NSUB matrix instructions per
// 4-node step, so the loop is bound by the matrix pipe (~64 cycles per instruction), not by operand delivery as the vector
// version is (64 dependent FMA steps per (level, gas), one scalar F stream per 16 states).  NSUB = 4 for full grids, 1 for small
// ones (a nu-shard), where 4x more, 4x shorter waves fill the chip.
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NSUB, bool VARNC = false>   // VARNC: node counts per level from A.nc (k_phco2_nodes' levels); else CS_NC everywhere
__global__ __launch_bounds__(256, NSUB >= 4 ? 2 : 4) void k_cheb_apply_mfma(ChebApply A, int Kpad, int64_t nnu, int ntile, int K, double base,
                                                          const double *__restrict__ extra, double *__restrict__ sigma, int accumulate)
{
    // 1-D grid, XCD-aware as k_cheb_apply: all state chunks of a tile block go to the same XCD, back to back
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nst = (K + 15) >> 4;                   // 16-state sub-tiles in use (Kpad, the row pitch of F, may be larger)
    const int nsg = (nst + NSUB - 1) / NSUB;         // state chunks of NSUB sub-tiles
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tile = ((q / nsg) * 8 + xcd) * 4 + wv;
    if (tile >= ntile) return;
    const int s0 = (q % nsg) * NSUB;
    const int nsub = min(NSUB, nst - s0);            // wave-uniform
    const int lr = lane & 15, lq = lane >> 4;
    v4f64 acc[NSUB][4];
#pragma unroll
    for (int si = 0; si < NSUB; si++)
#pragma unroll
        for (int jt = 0; jt < 4; jt++) acc[si][jt] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < A.ngas; g++) {
        const double *__restrict__ Fg = A.F[g];
        for (int l = A.l0[g]; l < A.nlev; l++) {
            const int sh = A.shift[l];
            const int T = tile >> sh, sub = tile & ((1 << sh) - 1);
            const size_t itv = (size_t)64 << sh;
            const int nc = VARNC ? A.nc[l] : CS_NC;
            const double *__restrict__ Cp = A.Cm[l] + ((size_t)T * nc + lq) * itv + (size_t)sub * 64 + lr;                         // node lq, point lr
            const double *__restrict__ Fp = Fg + ((size_t)A.noff[l] + (size_t)T * nc + lq) * Kpad + (size_t)s0 * 16 + lr;        // node lq, state lr
#pragma unroll 2
            for (int m = 0; m < nc; m += 4) {
                double b[4], a[NSUB];
#pragma unroll
                for (int jt = 0; jt < 4; jt++) b[jt] = Cp[(size_t)m * itv + jt * 16];
#pragma unroll
                for (int si = 0; si < NSUB; si++) a[si] = Fp[(size_t)m * Kpad + min(si, nsub - 1) * 16];   // (a sub-tile past the last one re-reads it: never stored)
#pragma unroll
                for (int si = 0; si < NSUB; si++)
#pragma unroll
                    for (int jt = 0; jt < 4; jt++) acc[si][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[si], b[jt], acc[si][jt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int si = 0; si < NSUB; si++) {
        if (si >= nsub) break;
#pragma unroll
        for (int jt = 0; jt < 4; jt++) {
            const int64_t i = (int64_t)tile * 64 + jt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = (s0 + si) * 16 + 4 * r + lq;
                if (k < K && i < nnu) {
                    const size_t o = (size_t)k * nnu + i;
                    const double prev = accumulate ? sigma[o] : (base + (extra ? extra[o] : 0.0));
                    sigma[o] = prev + acc[si][jt][r];
                }
            }
        }
    }
}

// K2a: far wings.  One wave = 64 consecutive wavenumbers x one node state; its window of lines [W0,W1) (sorted by nul)
// is cut into wave-uniform segments so that ~90 % of the (nu, line) pairs run a 13-16 instruction branch-free body whose
// line parameters arrive through scalar loads:
//   [W0,a) left edge (cut-off predicate) | [a,M0) far | [M0,N0) mid-far | [N0,N1) near zone | [N1,M1) | [M1,b) | [b,W1)
// In the near zone only the pairs with s >= 1e4 are summed here; the others belong to k_voigt_near.
#ifndef CS_FAR_ATTR
#define CS_FAR_ATTR
#endif
template <bool MIXED, int S, bool LOR, bool EDGE = false>
__global__ __launch_bounds__(256) CS_FAR_ATTR void k_voigt_far(const double *__restrict__ nu, int64_t nnu, int64_t L,
                                                    const LineHot *__restrict__ hot, const LineF32 *__restrict__ hot32,
                                                    const double *__restrict__ gnul, const WaveWin *__restrict__ win,
                                                    const Zone *__restrict__ zones, int ntile, int nblk, double cut,
                                                    double base, const double *__restrict__ extra,
                                                    double *__restrict__ sigma, int accumulate, int2 *__restrict__ ranges,
                                                    const IZone *__restrict__ iz, int nI, int ishift, const EdgeZone *__restrict__ edge,
                                                    double *__restrict__ zero2 = nullptr)
{
    // zero2 != NULL: the plane k_voigt_near will add this step's near-line pairs into (its first writer of the step clears it)
    // edge != NULL: the window ends [W0, eL) and [eR, W1) of the tile -- the cut-off edges and the far lines no interval could
    // take -- and the pieces [mL0, mL1), [mR0, mR1) between the interpolated sets and the near zone where the series in 1/dnu^2 holds
    // are summed for 16 states at a time on the matrix cores (k_voigt_edge_mx): skip them here.
    // iz != NULL: the lines [E0,Z0) U [Z1,E1) of the tile's parent interval (tile >> ishift, smallest interval size) were
    // summed by k_voigt_cheb -- skip them here.
    // S = 1: one wave per tile.  S = 2, 4: the S waves of a tile split its window of lines into S parts of equal estimated
    // cost and add their partial sums through LDS -- S times more, S times shorter waves, for grids too small to fill the
    // chip otherwise (a nu-shard of a multi-GPU run, bake on a short grid).
    __shared__ double acc_sh[S > 1 ? 256 : 1];
    __shared__ int4 rng_sh[S > 1 ? 256 : 1];
    bool in_stretch;
    const int tb = tile_block(reinterpret_cast<const int32_t *>(win + ntile), 4 / S, in_stretch);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform by construction:
    const int tile = tb * (4 / S) + wv / S;                                                      // tell the compiler, so that the
    const int part = wv % S;                                                                     // line records stay scalar loads
    const bool work = in_stretch && tb < nblk && tile < ntile;
    if (S == 1 && !work) return;
    // states from the last (the surface: highest pressure, widest lines, by far the longest waves of this kernel) to the first: the
    // grid is dealt out y-slowest, and the long waves must not be the tail
    const int k = (int)(gridDim.y - 1 - blockIdx.y);
    const int64_t i = (int64_t)tile * 64 + lane;
    double acc = 0.0;
    int bl = 0x3fffffff, bh = -1, cl = 0x3fffffff, ch = -1;
    Zone z = {};
    bool core = false;   // (wave-uniform) the core of this (tile, state group) belongs to k_voigt_sub, the near-line hand-off with it
    if (work) {
        const LineHot *__restrict__ hk = hot + (size_t)k * L;
        const LineF32 *__restrict__ hf = MIXED ? hot32 + (size_t)k * L : nullptr;
        const double v = nu[i < nnu ? i : nnu - 1];
        const WaveWin w = win[tile];
        z = zones[(size_t)k * ntile + tile];
        const FarK c = load_fark();
        // lines already covered by the interpolated sum: [sa0,sa1) and [sb0,sb1) (empty when interpolation is off)
        int sa0 = z.M0, sa1 = z.M0, sb0 = z.M1, sb1 = z.M1;
        if (iz) {
            const IZone zi = iz[(size_t)k * nI + (tile >> ishift)];
            sa0 = min(max(zi.E0, w.W0), z.N0); sa1 = min(max(zi.Z0, sa0), z.N0);
            sb0 = max(min(zi.Z1, w.W1), z.N1); sb1 = max(min(zi.E1, w.W1), sb0);
        }
        // what is left of the window after the matrix-core pieces, [wl, wr), and this wave's share [q0, q1) of it
        int wl = w.W0, wr = w.W1, pL0 = sa1, pL1 = sa1, pR0 = sb0, pR1 = sb0, cL = 0, cR = 0;
        if (EDGE) {   // (its own instantiation: on sparse tables the extra clip windows cost more than they save)
            const EdgeZone e = edge[(size_t)(k >> 4) * ntile + tile];
            wl = e.eL; wr = e.eR;
            if (e.mL1 > e.mL0) { pL0 = e.mL0; pL1 = e.mL1; }
            if (e.mR1 > e.mR0) { pR0 = e.mR0; pR1 = e.mR1; }
            if (e.cR > e.cL) { cL = e.cL; cR = e.cR; core = true; }
        }
        if (!core) cL = cR = pR0;   // (no core: the middle window [pL1, pR0) stays whole)
        int q0 = wl, q1 = wr;
        if (S > 1) {
            // piecewise-constant cost per line: 14 (far), 0 (skipped), 20 (4-term), 36 (near zone)
            const int cc = core ? 0 : 1;   // (the core of the window is not this kernel's)
            const int b12[12] = {wl, max(sa0, wl), sa1, pL0, pL1, core ? pL1 : z.N0, core ? pR0 : z.N1, pR0, pR1, sb0, min(sb1, wr), wr};
            const int c11[11] = {14, 0, 20, 0, 20 * cc, 36 * cc, 20 * cc, 0, 20, 0, 14};
            int total = 0;
            for (int q = 0; q < 11; q++) total += (b12[q + 1] - b12[q]) * c11[q];
            auto cut_at = [&](int target) {
                int accu = 0;
                for (int q = 0; q < 11; q++) {
                    const int seg = (b12[q + 1] - b12[q]) * c11[q];
                    if (c11[q] > 0 && accu + seg >= target) return b12[q] + (target - accu) / c11[q];
                    accu += seg;
                }
                return wr;
            };
            q0 = part == 0 ? wl : cut_at((int)((long long)total * part / S));
            q1 = part == S - 1 ? wr : cut_at((int)((long long)total * (part + 1) / S));
        }
        const int cw_lo[6] = {wl, sa1, EDGE ? pL1 : sb1, EDGE ? cR : sb1, pR1, sb1};
        const int cw_hi[6] = {sa0, EDGE ? pL0 : sb0, EDGE ? cL : wr, pR0, sb0, wr};
        for (int cw = 0; cw < (EDGE ? 6 : 3); cw++) {
        const int p0 = max(q0, cw_lo[cw]), p1 = min(q1, cw_hi[cw]);
        if (p0 >= p1) continue;
#define LO(x) max((x), p0)
#define HI(x) min((x), p1)
        {   // left of the wave: [W0,a) edge (cut-off predicate) | [a,Q0) 2-term | [Q0,a1) | [a1,M0) 3-term   (a <= Q0 <= a1 <= M0)
            const int a = min(max(w.E0, w.W0), z.Q0), a1 = min(max(w.E0, z.Q0), z.M0);
            if (MIXED) {
                acc = far_segment32<true, 0>(acc, v, gnul, hf, LO(w.W0), HI(a), cut);
                acc = far_segment32<false, 0>(acc, v, gnul, hf, LO(a), HI(z.Q0), cut);
                acc = far_segment32<true, 1>(acc, v, gnul, hf, LO(z.Q0), HI(a1), cut);
                acc = far_segment32<false, 1>(acc, v, gnul, hf, LO(a1), HI(z.M0), cut);
            } else {
                acc = far_segment<true, 0, LOR>(acc, v, hk, LO(w.W0), HI(a), cut, c);
                acc = far_segment<false, 0, LOR>(acc, v, hk, LO(a), HI(z.Q0), cut, c);
                acc = far_segment<true, 1, LOR>(acc, v, hk, LO(z.Q0), HI(a1), cut, c);
                acc = far_segment<false, 1, LOR>(acc, v, hk, LO(a1), HI(z.M0), cut, c);
            }
        }
        acc = far_segment<true, 2, LOR>(acc, v, hk, LO(z.M0), HI(z.N0), cut, c);
        // near zone: six-term series where s >= 1e3; the index ranges of this lane's s < 1e3 and s < 100 lines go to
        // k_voigt_near through `ranges` (relative to N0; empty = {0,0})
        if (HI(z.N1) > LO(z.N0)) {
            const double q40 = sgpr_const(59.0625), q41 = sgpr_const(-787.5), q42 = sgpr_const(2835.0), q43 = sgpr_const(-3780.0),
                         q44 = vgpr_const(1680.0);                                                            // a4 U8 in t
            const double q50 = sgpr_const(324.84375), q51 = sgpr_const(-6496.875), q52 = sgpr_const(36382.5), q53 = sgpr_const(-83160.0),
                         q54 = sgpr_const(83160.0), q55 = vgpr_const(-30240.0);                                // a5 U10 in t
#pragma unroll 4
            for (int j = LO(z.N0); j < HI(z.N1); j++) {
                const LineHot h = hk[j];
                const double dv = v - h.nul;
                const double x = dv * h.p1;
                const double s = __builtin_fma(x, x, h.p2);
                const bool in = !(fabs(dv) > cut);
                const double u = rcp_fast(s);
                const double t = h.p2 * u;
                const double p5 = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(q55, t, q54), t, q53), t, q52), t, q51), t, q50);
                const double p4 = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(q44, t, q43), t, q42), t, q41), t, q40);
                const double p3 = __builtin_fma(__builtin_fma(__builtin_fma(c.km120, t, c.k210), t, c.km105), t, c.k13p125);
                const double p2 = __builtin_fma(__builtin_fma(c.k12, t, c.km15), t, c.k3p75);
                const double p1v = __builtin_fma(-2.0, t, c.k1p5);
                double P = __builtin_fma(u, p5, p4);
                P = __builtin_fma(u, P, p3);
                P = __builtin_fma(u, P, p2);
                P = __builtin_fma(u, P, p1v);
                P = __builtin_fma(u, P, 1.0);
                // (the term is made opaque before the select: otherwise the compiler sinks the whole series into a branch under the
                // lanes' exec mask -- which saves nothing on a wave that always has such lanes -- and with it the load of p3, so that
                // every line waits twice for scalar memory instead of four lines waiting once)
                double term = (h.p3 * u) * P;
                asm volatile("" : "+v"(term));
                acc += (in && s >= kSerS) ? term : 0.0;
                if (in && s < kSerS) { bl = min(bl, j); bh = j; }
                if (in && s < kMidS) { cl = min(cl, j); ch = j; }
                // (tried, round 5: four lines per hand-unrolled step with the nine instructions of this range bookkeeping skipped -- a
                //  scalar test of the records' y^2 >= 1e3, true for every line of the high-pressure states -- the kernel alone 0.275 ->
                //  0.283 ms, the step +0.1 %: the compiler's own unrolling schedules the four scalar loads and series better)
            }
        }
        acc = far_segment<true, 2, LOR>(acc, v, hk, LO(z.N1), HI(z.M1), cut, c);
        {   // right of the wave: [M1,b1) | [b1,Q1) 3-term | [Q1,b) | [b,W1) 2-term, beyond E1 with the cut-off predicate
            const int b1 = max(min(w.E1, z.Q1), z.M1), bq = max(min(w.E1, w.W1), z.Q1);
            if (MIXED) {
                acc = far_segment32<false, 1>(acc, v, gnul, hf, LO(z.M1), HI(b1), cut);
                acc = far_segment32<true, 1>(acc, v, gnul, hf, LO(b1), HI(z.Q1), cut);
                acc = far_segment32<false, 0>(acc, v, gnul, hf, LO(z.Q1), HI(bq), cut);
                acc = far_segment32<true, 0>(acc, v, gnul, hf, LO(bq), HI(w.W1), cut);
            } else {
                acc = far_segment<false, 1, LOR>(acc, v, hk, LO(z.M1), HI(b1), cut, c);
                acc = far_segment<true, 1, LOR>(acc, v, hk, LO(b1), HI(z.Q1), cut, c);
                acc = far_segment<false, 0, LOR>(acc, v, hk, LO(z.Q1), HI(bq), cut, c);
                acc = far_segment<true, 0, LOR>(acc, v, hk, LO(bq), HI(w.W1), cut, c);
            }
        }
#undef LO
#undef HI
        }  // clip windows
    }
    if (S > 1) {   // add the parts in part order (deterministic) and merge the index ranges
        acc_sh[threadIdx.x] = acc;
        rng_sh[threadIdx.x] = make_int4(bl, bh, cl, ch);
        __syncthreads();
        if (part != 0 || !work) return;
        for (int q = 1; q < S; q++) {
            const int o = threadIdx.x + 64 * q;
            acc += acc_sh[o];
            const int4 r = rng_sh[o];
            bl = min(bl, r.x); bh = max(bh, r.y); cl = min(cl, r.z); ch = max(ch, r.w);
        }
    }
    const bool live = i < nnu;
    if (live) {
        const size_t o = (size_t)k * nnu + i;
        if (!accumulate) sigma[o] = (base + (extra ? extra[o] : 0.0)) + acc;
        else if (acc != 0.0) sigma[o] += acc;   // (nothing to add -- most tiles of a sparse table: no trip to sigma at all; x + 0.0 = x bit for bit)
        if (zero2) zero2[o] = 0.0;
    }
    // hand-off to k_voigt_near<0>, <1>: per (nu, node) and tier one word, (first line - N0) << 12 | count -- 8 bytes per
    // spectral point and node in all (cs_api.hip refuses tables dense enough to overflow 20 + 12 bits: check_near_density) --
    // and per (tile, node) and tier a flag "some lane has candidates": tiles without any (most tiles of a sparse table, every
    // tile of a high-pressure state, where y^2 alone exceeds 1e3) skip the store here and the whole wave there
    if (!LOR && !core) {   // (a Lorentz profile has no near-line kernels to hand anything to; a core's hand-off is k_voigt_sub's)
        const unsigned r0 = (live && bh >= bl) ? ((unsigned)(bl - z.N0) << 12) | (unsigned)(bh + 1 - bl) : 0u;
        const unsigned r1 = (live && ch >= cl) ? ((unsigned)(cl - z.N0) << 12) | (unsigned)(ch + 1 - cl) : 0u;
        const bool any0 = __any(r0 != 0u), any1 = __any(r1 != 0u);
        unsigned *__restrict__ rp = reinterpret_cast<unsigned *>(ranges);
        const size_t plane = (size_t)gridDim.y * nnu;
        if (any0 && live) rp[(size_t)k * nnu + i] = r0;
        if (any1 && live) rp[plane + (size_t)k * nnu + i] = r1;
        if (lane == 0) {
            unsigned *__restrict__ fl = rp + 2 * plane;
            fl[(size_t)k * ntile + tile] = any0 ? 1u : 0u;
            fl[((size_t)gridDim.y + k) * ntile + tile] = any1 ? 1u : 0u;
        }
    }
}

// ---- K2e: window ends of the per-point sum on the matrix cores -----------------------------------------------------------------
// What no interval can take at the far ends of a tile's window -- the lines inside the cut-off of only some points of the
// smallest interval, a third of k_voigt_far's instructions at 17 per pair -- is as far from the tile as lines get, so the same
// state-separable series as K2d applies (sum_n C_n[state][line] w^n, w = 1/dnu^2), here with the points of the tile as columns
// and the cut-off as a mask on w:  one wave = one 64-point tile x 16 states, 16 matrix instructions per 4 lines.
struct EdgeArgs {
    const double *nu, *nul, *gbound, *Tk;
    const WaveWin *win;
    const Zone *zones;
    const IZone *iz;      // lowest interpolation level, or NULL
    EdgeZone *out;        // [ngrp][ntile]
    int64_t nnu;
    int ntile, K, ngrp, nI, ishift, core;   // core: sub-tile treatment of the window core where it pays (k_voigt_sub)
    double mu_min, cut, core4;              // core4: the core takes the 4-term series where its radius is below core4 x the tile's span
};
// per (state group, tile): the pieces common to the group's states -- the window ends left of every state's first interpolated or
// near-zone line and right of the last; between every state's interpolated sets and near zone -- clipped to the distance at which
// the 4-term series holds for the widest line of the group
__device__ __forceinline__ void edgezones_body(unsigned bid, const EdgeArgs &a)
{
    const int idx = bid * blockDim.x + threadIdx.x;
    if (idx >= a.ntile * a.ngrp) return;
    const int g = idx / a.ntile, t = idx - g * a.ntile;
    const int64_t i0 = (int64_t)t * 64, i1 = (i0 + 63 < a.nnu ? i0 + 63 : a.nnu - 1);
    const double vlo = a.nu[i0], vhi = a.nu[i1];
    const WaveWin w = a.win[t];
    int eL = w.W1, eR = w.W0, mL0 = w.W0, mL1 = w.W1, mR0 = w.W0, mR1 = w.W1;
    double R = 0.0, R3 = 0.0, R8 = 0.0;
    for (int k = g * 16; k < min(g * 16 + 16, a.K); k++) {
        const Zone z = a.zones[(size_t)k * a.ntile + t];
        int sa0 = z.M0, sa1 = z.M0, sb0 = z.M1, sb1 = z.M1;   // (as k_voigt_far)
        if (a.iz) {
            const IZone zi = a.iz[(size_t)k * a.nI + (t >> a.ishift)];
            sa0 = min(max(zi.E0, w.W0), z.N0); sa1 = min(max(zi.Z0, sa0), z.N0);
            sb0 = max(min(zi.Z1, w.W1), z.N1); sb1 = max(min(zi.E1, w.W1), sb0);
        }
        eL = min(eL, sa0); eR = max(eR, sb1);
        mL0 = max(mL0, sa1); mL1 = min(mL1, z.N0);
        mR0 = max(mR0, z.N1); mR1 = min(mR1, sb0);
        const double amax = ((vhi + a.cut) / kC) * sqrt(2.0 * kRgas * a.Tk[k]) / sqrt(a.mu_min);
        const double gb = a.gbound[k];
        R = fmax(R, kSep4 * sqrt(gb * gb + 4.33 * amax * amax) * (1.0 + 1e-6));
        R3 = fmax(R3, kSep3 * sqrt(gb * gb + 5.05 * amax * amax) * (1.0 + 1e-6));
        R8 = fmax(R8, kSep8 * sqrt(gb * gb + 6.35 * amax * amax) * (1.0 + 1e-6));
    }
    int sr[4];
    {
        const double sv[4] = {vlo - R, vlo - R3, vhi + R, vhi + R3};
        search4(a.nul, sv, w.W0, w.W1, sr);
    }
    const int S0 = sr[0], S1 = max(sr[2], S0);            // the series holds in [W0, S0) and [S1, W1)
    const int T0 = min(sr[1], S0), T1 = max(sr[3], S1);   // ... with three terms in [W0, T0) and [T1, W1)
    EdgeZone e;
    e.eL = max(min(eL, S0), w.W0);
    e.eR = min(max(eR, S1), w.W1);
    if (e.eL - w.W0 < 8) e.eL = w.W0;   // (too short to be worth a wave's trip)
    if (w.W1 - e.eR < 8) e.eR = w.W1;
    e.mL0 = mL0; e.mL1 = min(mL1, S0);
    e.mR0 = max(mR0, S1); e.mR1 = mR1;
    if (e.mL1 - e.mL0 < 8 || e.mL0 < e.eL) e.mL0 = e.mL1 = 0;
    if (e.mR1 - e.mR0 < 8 || e.mR1 > e.eR) e.mR0 = e.mR1 = 0;
    e.far3 = (e.eL <= T0 ? 1 : 0) | (e.eR >= T1 ? 2 : 0);    // (a window end is 20 cm^-1 away: all of it or none)
    e.mL3 = min(max(T0, e.mL0), e.mL1);
    if (e.mL3 - e.mL0 < 8) e.mL3 = e.mL0;
    e.mR3 = min(max(T1, e.mR0), e.mR1);
    if (e.mR1 - e.mR3 < 8) e.mR3 = e.mR1;
    // core: from the end of the left middle piece (or of every state's interpolated set) to the start of the right one.  Worth
    // it where a sub-tile's neighbourhood (16 points + 2 R) is well below the tile's (64 points + 2 R), and only with all 16 points
    // of every sub-tile present (a ragged last tile keeps the tile-wide pass)
    e.cL = e.mL1 > e.mL0 ? e.mL1 : mL0;
    e.cR = e.mR1 > e.mR0 ? e.mR0 : mR1;
    // the pairs of the core inside the radius are k_voigt_sub's: the radius of the 4-term series where it is short against the tile
    // (the low-pressure groups), else that of the 8-term series -- twice the matrix instructions for the core's lines, a radius
    // 12 times shorter (the groups whose Lorentz widths set it)
    const bool core8 = !(R < a.core4 * (vhi - vlo));
    const double Rc = core8 ? R8 : R;
    const bool core_ok = a.core && a.iz && i0 + 63 < a.nnu && mL0 <= mL1 && mR0 <= mR1 && e.cL <= mL1 && e.cR >= mR0 && e.cR > e.cL &&
                         Rc < (core8 ? 0.3 : 0.75) * (vhi - vlo);   // (an 8-term core costs twice the matrix instructions: only where the sub-tiles then see few lines)
    if (!core_ok) e.cL = e.cR = 0;
    else if (core8) e.far3 |= 4;
    e.R = Rc;
    e.pad0 = 0;
    e.pad1 = 0.0;
    a.out[idx] = e;
}
// k_sepzones and the edge zones in one launch (both need the zones of k_gas_setup)
__global__ __launch_bounds__(256) void k_mxzones(unsigned nb_sep, SepArgs sa, EdgeArgs ea)
{
    if (blockIdx.x < nb_sep) sepzones_body(blockIdx.x, sa);
    else edgezones_body(blockIdx.x - nb_sep, ea);
}
// The same two tables with SIXTEEN lanes per item, lane = state of the group (k_mxzones16): the bodies above are one thread per item --
// sixteen zone records read one after the other and a 16 x 16 counting rank, ~3000 instructions in a chain -- and their launch sits at the
// head of every step, where a nu-shard of a few hundred tiles has six blocks of it to run (21 us of 0.39 ms).  Here every lane reads its
// own state's records, the group's bounds are 16-lane reductions, the rank of a state's series bound is sixteen shuffled comparisons,
// and lane 0 stores: the same integers, a chain ten times shorter.
template <class T, class Op> __device__ __forceinline__ T red16(T v, Op op)
{
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) v = op(v, __shfl_xor(v, m, 16));
    return v;
}
// RC: the per-state zones are computed here (izone_compute / zone_compute: the searches of k_gas_setup's zone blocks again, every lane its
// own state's) instead of read -- then the piece tables need nothing k_gas_setup writes and are blocks of ITS launch (k_gas_setup_mx):
// one launch and one dependent kernel boundary less at the head of every step (14 + 3 us on a 1/8 shard of the bench column)
template <bool RC>
__device__ __forceinline__ void sepzones_body16(unsigned bid, const SepArgs &a, const IzParams *ip = nullptr, const ZoneArgs *za = nullptr)
{
    const int item = (int)(bid * (blockDim.x >> 4) + (threadIdx.x >> 4)), kk = threadIdx.x & 15;
    const int nq = a.nItot - a.q0;
    if (item >= nq * a.ngrp) return;            // (uniform over the 16 lanes of an item)
    const int g = item / nq, T = a.q0 + (item - g * nq);
    const double vhi = a.nodes[(size_t)T * CS_NC], vlo = a.nodes[(size_t)T * CS_NC + CS_NC - 1];   // nodes run from the upper end down
    const int k = g * 16 + kk;
    const bool have = k < a.K;
    IZone z = {};
    if (have) z = RC ? izone_compute(*ip, *za, k, T) : a.iz[(size_t)k * a.nItot + T];
    const auto imax = [](int x, int y) { return max(x, y); };
    const auto imin = [](int x, int y) { return min(x, y); };
    int lo[4], hi[4];
    lo[0] = red16(have ? z.E0 : 0, imax); hi[0] = red16(have ? z.P0 : 0x7fffffff, imin);
    lo[1] = red16(have ? z.P1 : 0, imax); hi[1] = red16(have ? z.Z0 : 0x7fffffff, imin);
    lo[2] = red16(have ? z.Z1 : 0, imax); hi[2] = red16(have ? z.P2 : 0x7fffffff, imin);
    lo[3] = red16(have ? z.P3 : 0, imax); hi[3] = red16(have ? z.E1 : 0x7fffffff, imin);
    const int ns = min(16, a.K - g * 16);
    const int E0 = __shfl(z.E0, ns - 1, 16), E1 = __shfl(z.E1, ns - 1, 16);      // (the last state's, as the one-thread body leaves them)
    double R3 = 0.0;
    if (have) {
        const double amax = ((vhi + a.cut) / kC) * sqrt(2.0 * kRgas * a.Tk[k]) / sqrt(a.mu_min);
        const double gb = a.gbound[k];
        R3 = kSep3 * sqrt(gb * gb + 5.05 * amax * amax) * (1.0 + 1e-6);
    }
    R3 = red16(R3, [](double x, double y) { return fmax(x, y); });
    // rank of this state's series bounds among the group's (ties: the lower state first), as in the one-thread body
    const int s0 = have ? z.S0 : -0x7fffffff, s1 = have ? z.S1 : 0x7fffffff;
    int c0 = 0, c1 = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int t0 = __shfl(s0, j, 16), t1 = __shfl(s1, j, 16);
        c0 += (t0 > s0 || (t0 == s0 && j < kk)) ? 1 : 0;
        c1 += (t1 < s1 || (t1 == s1 && j < kk)) ? 1 : 0;
    }
    const int rk = min(max(a.min_states, 1), ns) - 1;
    const int S0 = red16(c0 == rk ? s0 : -0x7fffffff - 1, imax);
    int S1 = red16(c1 == rk ? s1 : -0x7fffffff - 1, imax);
    S1 = max(S1, S0);
    int sr[4];
    {
        const double sv[4] = {vlo - R3, vlo - R3, vhi + R3, vhi + R3};
        search4(a.nul, sv, E0, E1, sr);          // (every lane of the item: same loads, no divergence)
    }
    if (kk != 0) return;
    const int T0 = min(sr[1], S0), T1 = max(sr[3], S1);
    SepZone o;
    for (int p = 0; p < 4; p++) {
        int pa = lo[p], pb = hi[p];
        if (p < 2) pb = min(pb, S0); else pa = max(pa, S1);
        if (pb - pa < 8) { pa = 0; pb = 0; }
        o.a[p] = pa; o.b[p] = pb;
        int pm = p < 2 ? min(max(T0, pa), pb) : min(max(T1, pa), pb);
        if (p < 2 && pm - pa < 8) pm = pa;
        if (p >= 2 && pb - pm < 8) pm = pb;
        o.m[p] = pm;
    }
    a.out[(size_t)g * a.nItot + T] = o;
}
template <bool RC>
__device__ __forceinline__ void edgezones_body16(unsigned bid, const EdgeArgs &a, const IzParams *ip = nullptr, const ZoneArgs *za = nullptr)
{
    const int item = (int)(bid * (blockDim.x >> 4) + (threadIdx.x >> 4)), kk = threadIdx.x & 15;
    if (item >= a.ntile * a.ngrp) return;
    const int g = item / a.ntile, t = item - g * a.ntile;
    const int64_t i0 = (int64_t)t * 64, i1 = (i0 + 63 < a.nnu ? i0 + 63 : a.nnu - 1);
    const double vlo = a.nu[i0], vhi = a.nu[i1];
    const WaveWin w = a.win[t];
    const int k = g * 16 + kk;
    const bool have = k < a.K;
    int eL = w.W1, eR = w.W0, mL0 = w.W0, mL1 = w.W1, mR0 = w.W0, mR1 = w.W1;
    double R = 0.0, R3 = 0.0, R8 = 0.0;
    if (have) {
        const Zone z = RC ? zone_compute(*za, k, t) : a.zones[(size_t)k * a.ntile + t];
        int sa0 = z.M0, sa1 = z.M0, sb0 = z.M1, sb1 = z.M1;   // (as k_voigt_far)
        if (a.iz) {
            const IZone zi = RC ? izone_compute(*ip, *za, k, ip->ioff[ip->nlev - 1] + (t >> a.ishift)) : a.iz[(size_t)k * a.nI + (t >> a.ishift)];
            sa0 = min(max(zi.E0, w.W0), z.N0); sa1 = min(max(zi.Z0, sa0), z.N0);
            sb0 = max(min(zi.Z1, w.W1), z.N1); sb1 = max(min(zi.E1, w.W1), sb0);
        }
        eL = min(eL, sa0); eR = max(eR, sb1);
        mL0 = max(mL0, sa1); mL1 = min(mL1, z.N0);
        mR0 = max(mR0, z.N1); mR1 = min(mR1, sb0);
        const double amax = ((vhi + a.cut) / kC) * sqrt(2.0 * kRgas * a.Tk[k]) / sqrt(a.mu_min);
        const double gb = a.gbound[k];
        R = kSep4 * sqrt(gb * gb + 4.33 * amax * amax) * (1.0 + 1e-6);
        R3 = kSep3 * sqrt(gb * gb + 5.05 * amax * amax) * (1.0 + 1e-6);
        R8 = kSep8 * sqrt(gb * gb + 6.35 * amax * amax) * (1.0 + 1e-6);
    }
    const auto imax = [](int x, int y) { return max(x, y); };
    const auto imin = [](int x, int y) { return min(x, y); };
    const auto dmax = [](double x, double y) { return fmax(x, y); };
    eL = red16(eL, imin); eR = red16(eR, imax);
    mL0 = red16(mL0, imax); mL1 = red16(mL1, imin);
    mR0 = red16(mR0, imax); mR1 = red16(mR1, imin);
    R = red16(R, dmax); R3 = red16(R3, dmax); R8 = red16(R8, dmax);
    int sr[4];
    {
        const double sv[4] = {vlo - R, vlo - R3, vhi + R, vhi + R3};
        search4(a.nul, sv, w.W0, w.W1, sr);
    }
    if (kk != 0) return;
    const int S0 = sr[0], S1 = max(sr[2], S0);            // the series holds in [W0, S0) and [S1, W1)
    const int T0 = min(sr[1], S0), T1 = max(sr[3], S1);   // ... with three terms in [W0, T0) and [T1, W1)
    EdgeZone e;
    e.eL = max(min(eL, S0), w.W0);
    e.eR = min(max(eR, S1), w.W1);
    if (e.eL - w.W0 < 8) e.eL = w.W0;   // (too short to be worth a wave's trip)
    if (w.W1 - e.eR < 8) e.eR = w.W1;
    e.mL0 = mL0; e.mL1 = min(mL1, S0);
    e.mR0 = max(mR0, S1); e.mR1 = mR1;
    if (e.mL1 - e.mL0 < 8 || e.mL0 < e.eL) e.mL0 = e.mL1 = 0;
    if (e.mR1 - e.mR0 < 8 || e.mR1 > e.eR) e.mR0 = e.mR1 = 0;
    e.far3 = (e.eL <= T0 ? 1 : 0) | (e.eR >= T1 ? 2 : 0);
    e.mL3 = min(max(T0, e.mL0), e.mL1);
    if (e.mL3 - e.mL0 < 8) e.mL3 = e.mL0;
    e.mR3 = min(max(T1, e.mR0), e.mR1);
    if (e.mR1 - e.mR3 < 8) e.mR3 = e.mR1;
    e.cL = e.mL1 > e.mL0 ? e.mL1 : mL0;
    e.cR = e.mR1 > e.mR0 ? e.mR0 : mR1;
    const bool core8 = !(R < a.core4 * (vhi - vlo));
    const double Rc = core8 ? R8 : R;
    const bool core_ok = a.core && a.iz && i0 + 63 < a.nnu && mL0 <= mL1 && mR0 <= mR1 && e.cL <= mL1 && e.cR >= mR0 && e.cR > e.cL &&
                         Rc < (core8 ? 0.3 : 0.75) * (vhi - vlo);
    if (!core_ok) e.cL = e.cR = 0;
    else if (core8) e.far3 |= 4;
    e.R = Rc;
    e.pad0 = 0;
    e.pad1 = 0.0;
    a.out[item] = e;
}
__global__ __launch_bounds__(256) void k_mxzones16(unsigned nb_sep, SepArgs sa, EdgeArgs ea)
{
    if (blockIdx.x < nb_sep) sepzones_body16<false>(blockIdx.x, sa);
    else edgezones_body16<false>(blockIdx.x - nb_sep, ea);
}
// k_gas_setup and k_mxzones16 in ONE launch: zone blocks, interval-zone blocks, the piece tables of the matrix-core kernels (which
// compute the zones they need themselves), record blocks
__global__ __launch_bounds__(256) void k_gas_setup_mx(unsigned nb_prep, unsigned nb_zones, unsigned nb_iz, unsigned nb_sep, PrepArgs pa, ZoneArgs za, IzParams ip,
                                                      IZone *__restrict__ iz, SepArgs sa, EdgeArgs ea)
{
    unsigned b = blockIdx.x;
    if (b < nb_zones) { zones_body(b, za); return; }
    b -= nb_zones;
    if (b < nb_iz) { izones_body(b, ip, za, iz); return; }
    b -= nb_iz;
    const unsigned nb_mx = gridDim.x - nb_zones - nb_iz - nb_prep;
    if (b < nb_mx) {
        if (b < nb_sep) sepzones_body16<true>(b, sa, &ip, &za);
        else edgezones_body16<true>(b - nb_sep, ea, &ip, &za);
        return;
    }
    prep_body(b - nb_mx, pa);
}

// (three waves per SIMD, with the 164 registers that allows: the 8-term step of the cores wants them -- left alone the allocator
//  takes 148 + 32 accumulators, i.e. two waves: 0.44 vs 0.41 ms)
// fuse != 0: the interpolated far wings of the tile come along -- sigma += sum_level C_l[interval][:, point] . F[interval][:, state]
// (k_cheb_apply_mfma's product, same operand and accumulator layout: states are the rows of D here too), added into the same
// accumulators, so that sigma makes one round trip less and the step one launch less.  The host fuses when the column has ONE
// launch group with interpolated wings and that group runs this kernel.
// SPLIT = 4 (short grids: a nu-shard of a multi-GPU run has a few hundred tiles, i.e. fewer (tile, group) waves than the chip has
// SIMDs, each a chain of a hundred dependent load -> matrix steps): the four waves of a block share ONE (tile, group), every piece
// cut into four runs of lines as in k_cheb_nodes_mx, partial sums added through LDS in wave order (bitwise repeatable).
template <int SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_voigt_edge_mx(const double *__restrict__ nu, int64_t nnu, int64_t L, const LineHot *__restrict__ hot,
                                                       const WaveWin *__restrict__ win, const EdgeZone *__restrict__ edge, int ntile, int K,
                                                       double cut, double *__restrict__ sigma, int fuse, ChebApply A, int Kpad,
                                                       const double *__restrict__ gnul, int phases, const double *__restrict__ tnodes,
                                                       const double *__restrict__ tC)
{
    // phases (SPLIT = 1): the far lines of a cut-off edge reach only the first (left end) or last (right end) columns of the tile, and
    // the reach grows with the line index -- so the window end is cut where the next 16-column sub-tile comes into reach, and each
    // part multiplies only the sub-tiles it can reach (a per-step test of "sub-tile all masked" cost what it saved; these are four
    // loops with the sub-tile count fixed at compile time).  gnul: the table's line positions (state-independent).
    __shared__ double part[SPLIT > 1 ? 4 : 1][SPLIT > 1 ? 16 : 1][SPLIT > 1 ? CS_MX_PITCH : 1];
    __shared__ double trn[SPLIT == 1 ? 4 : 1][SPLIT == 1 ? 16 : 1][SPLIT == 1 ? 18 : 1];   // (the node path's transpose, per wave)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // (2-D grid, tile blocks fastest: consecutive blocks are neighbouring tiles of ONE state group, whose pieces overlap by half --
    // the XCD-aware 1-D order of k_cheb_apply_mfma, all groups of a tile block back to back on one XCD, costs 0.43 -> 0.57 ms at C3)
    const int tile = SPLIT > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wv, g = blockIdx.y;
    if (tile >= ntile) return;   // (block-uniform when SPLIT > 1)
    const WaveWin w = win[tile];
    const EdgeZone e = edge[(size_t)g * ntile + tile];
    if (!fuse && e.eL <= w.W0 && e.eR >= w.W1 && e.mL1 <= e.mL0 && e.mR1 <= e.mR0 && e.cR <= e.cL) return;   // (block-uniform when SPLIT > 1)
    const int lr = lane & 15, lq = lane >> 4;
    const int kk = min(g * 16 + lr, K - 1);                       // (a group's tail states re-read the last one: never stored)
    const LineHot *__restrict__ hk = hot + (size_t)kk * L;
    double vn[4];
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const int64_t i = (int64_t)tile * 64 + st * 16 + lr;
        vn[st] = nu[i < nnu ? i : nnu - 1];
    }
    v4f64_sep acc[4];
#pragma unroll
    for (int st = 0; st < 4; st++) acc[st] = v4f64_sep{0.0, 0.0, 0.0, 0.0};
    if (fuse) {
        const double *__restrict__ Fg = A.F[0];
        const int m0 = SPLIT > 1 ? wv * (CS_NC / 4) : 0, m1 = SPLIT > 1 ? m0 + CS_NC / 4 : CS_NC;   // (split: a quarter of the nodes per wave)
        for (int l = A.l0[0]; l < A.nlev; l++) {
            const int sh = A.shift[l];
            const int T = tile >> sh, sub = tile & ((1 << sh) - 1);
            const size_t itv = (size_t)64 << sh;
            const double *__restrict__ Cp = A.Cm[l] + ((size_t)T * CS_NC + lq) * itv + (size_t)sub * 64 + lr;        // node lq, point lr
            const double *__restrict__ Fp = Fg + ((size_t)(A.ioff[l] + T) * CS_NC + lq) * Kpad + (size_t)g * 16 + lr;  // node lq, state lr
#pragma unroll 2
            for (int m = m0; m < m1; m += 4) {
                double b[4];
#pragma unroll
                for (int st = 0; st < 4; st++) b[st] = Cp[(size_t)m * itv + st * 16];
                const double a = Fp[(size_t)m * Kpad];
#pragma unroll
                for (int st = 0; st < 4; st++) acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[st], acc[st], 0, 0, 0);
            }
        }
    }
    // this wave's run of the piece [pa, pb): all of it, or (SPLIT) one of four runs of a multiple of 4 lines, wave 0 at the far end
    auto run = [&](int pa, int pb, bool asc, int &ja, int &jb) {
        if (SPLIT == 1) { ja = pa; jb = pb; return; }
        const int len = ((max(pb - pa, 0) + 15) >> 4) << 2;
        ja = asc ? min(pa + wv * len, pb) : max(pb - (wv + 1) * len, pa);
        jb = asc ? min(ja + len, pb) : max(pb - wv * len, pa);
    };
    int ja, jb;
    // left pieces ascending, right pieces descending: far lines first; three terms where they do
    const bool ph = SPLIT == 1 && phases && gnul;
    const double tolc = 1e-9 * (fabs(__shfl(vn[0], 0)) + cut + 1.0);
    // (shorter ends: the four partial loops and their step fill cost more than the skipped sub-tiles -- C5: 0.350 -> 0.386 ms)
    const bool phL = ph && e.eL - w.W0 >= 48, phR = ph && w.W1 - e.eR >= 48;
    // Round 5: the lines of a window end that are inside the cut-off of EVERY point of the tile (they are in this end only because
    // the tile's 128-point interval has points they do not reach) need no mask and are 23+ cm^-1 = 29 half-widths of the tile away:
    // their sum is a polynomial of low degree across the tile, formed on 16 Chebyshev nodes of the tile (one sub-tile per matrix step
    // instead of four) and carried to the 64 points by the tile's 64 x 16 matrix tC (k_cheb_setup with 16 nodes; tnodes = NULL: off).
    const bool nd = tnodes != nullptr;
    int J[4] = {e.eL, e.eL, e.eL, e.eL}, U[4] = {w.W1, w.W1, w.W1, e.eR};   // J[3] .. eL, eR .. U[3]: the lines that go to the nodes
    if (phL) {
        // J[q]: first line of [W0, eL) whose cut-off reaches sub-tile q + 1 (its first column); J[3]: ... the tile's last point.  One
        // vector load per 64 lines and ballots
        const int p0 = w.W0, p1 = e.eL;
        const double a0 = __shfl(vn[1], 0) - cut - tolc, a1 = __shfl(vn[2], 0) - cut - tolc, a2 = __shfl(vn[3], 0) - cut - tolc,
                     a3 = __shfl(vn[3], 15) - cut + tolc;
        bool f0 = false, f1 = false, f2 = false, f3 = false;
        for (int base = p0; base < p1; base += 64) {
            const int j = base + lane;
            const double x = gnul[j < p1 ? j : p1 - 1];
            const uint64_t m0 = __builtin_amdgcn_ballot_w64(j < p1 && x >= a0), m1 = __builtin_amdgcn_ballot_w64(j < p1 && x >= a1),
                           m2 = __builtin_amdgcn_ballot_w64(j < p1 && x >= a2), m3 = __builtin_amdgcn_ballot_w64(j < p1 && x >= a3);
            if (m0 != 0 && !f0) { J[0] = base + __builtin_ctzll(m0); f0 = true; }
            if (m1 != 0 && !f1) { J[1] = base + __builtin_ctzll(m1); f1 = true; }
            if (m2 != 0 && !f2) { J[2] = base + __builtin_ctzll(m2); f2 = true; }
            if (m3 != 0 && !f3) { J[3] = base + __builtin_ctzll(m3); f3 = true; }
        }
        J[1] = max(J[1], J[0]); J[2] = max(J[2], J[1]); J[3] = max(J[3], J[2]);
        if (!nd || p1 - J[3] < 16) J[3] = p1;     // (too few for a trip of their own)
    }
    if (phR) {
        // U[q]: first line of [eR, W1) whose cut-off no longer reaches sub-tile q (its last column); lines from U[q] on need the
        // sub-tiles q + 1 .. 3 only.  U[3]: first line that no longer reaches the tile's FIRST point: [eR, U[3]) reach every point
        const int p0 = e.eR, p1 = w.W1;
        const double b0 = __shfl(vn[0], 15) + cut + tolc, b1 = __shfl(vn[1], 15) + cut + tolc, b2 = __shfl(vn[2], 15) + cut + tolc,
                     b3 = __shfl(vn[0], 0) + cut - tolc;
        bool f0 = false, f1 = false, f2 = false, f3 = false;
        U[3] = p1;
        for (int base = p0; base < p1; base += 64) {
            const int j = base + lane;
            const double x = gnul[j < p1 ? j : p1 - 1];
            const uint64_t m0 = __builtin_amdgcn_ballot_w64(j < p1 && x > b0), m1 = __builtin_amdgcn_ballot_w64(j < p1 && x > b1),
                           m2 = __builtin_amdgcn_ballot_w64(j < p1 && x > b2), m3 = __builtin_amdgcn_ballot_w64(j < p1 && x > b3);
            if (m0 != 0 && !f0) { U[0] = base + __builtin_ctzll(m0); f0 = true; }
            if (m1 != 0 && !f1) { U[1] = base + __builtin_ctzll(m1); f1 = true; }
            if (m2 != 0 && !f2) { U[2] = base + __builtin_ctzll(m2); f2 = true; }
            if (m3 != 0 && !f3) { U[3] = base + __builtin_ctzll(m3); f3 = true; }
        }
        U[1] = max(U[1], U[0]); U[2] = max(U[2], U[1]); U[3] = min(U[3], U[0]);
        if (!nd || U[3] - p0 < 16) U[3] = p0;
    }
    const bool ndL = phL && J[3] < e.eL, ndR = phR && U[3] > e.eR;
    if (SPLIT == 1 && nd && (ndL || ndR)) {   // (wave-uniform)
        v4f64_sep af[1] = {v4f64_sep{0.0, 0.0, 0.0, 0.0}};
        const double vf[1] = {tnodes[(size_t)tile * 16 + lr]};
        if (ndL) {
            if (e.far3 & 1) sep_run<3, 0, 1, 0, 1>(af, vf, hk, J[3], e.eL, true, lq, 0.0); else sep_run<4, 0, 1, 0, 1>(af, vf, hk, J[3], e.eL, true, lq, 0.0);
        }
        if (ndR) {
            if (e.far3 & 2) sep_run<3, 0, 1, 0, 1>(af, vf, hk, e.eR, U[3], false, lq, 0.0); else sep_run<4, 0, 1, 0, 1>(af, vf, hk, e.eR, U[3], false, lq, 0.0);
        }
        // acc[state][point] += sum_m af[state][m] tC[tile][m][point]: af goes through LDS from the D layout (state 4r + lq, node lr) to the A
        // layout (state lr, node lq of a group of four), as in mx_far_pieces
#pragma unroll
        for (int r = 0; r < 4; r++) trn[wv][4 * r + lq][lr] = af[0][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const double *__restrict__ Cp = tC + (size_t)tile * 16 * 64;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const double a = trn[wv][lr][4 * kk + lq];
#pragma unroll
            for (int st2 = 0; st2 < 4; st2++)
                acc[st2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Cp[(size_t)(4 * kk + lq) * 64 + 16 * st2 + lr], acc[st2], 0, 0, 0);
        }
    }
    if (phL) {
        const int p0 = w.W0;
        if (e.far3 & 1) {
            sep_run<3, 1, 1, 0>(acc, vn, hk, p0, J[0], true, lq, cut);
            sep_run<3, 1, 2, 0>(acc, vn, hk, J[0], J[1], true, lq, cut);
            sep_run<3, 1, 3, 0>(acc, vn, hk, J[1], J[2], true, lq, cut);
            sep_run<3, 1, 4, 0>(acc, vn, hk, J[2], J[3], true, lq, cut);
        } else {
            sep_run<4, 1, 1, 0>(acc, vn, hk, p0, J[0], true, lq, cut);
            sep_run<4, 1, 2, 0>(acc, vn, hk, J[0], J[1], true, lq, cut);
            sep_run<4, 1, 3, 0>(acc, vn, hk, J[1], J[2], true, lq, cut);
            sep_run<4, 1, 4, 0>(acc, vn, hk, J[2], J[3], true, lq, cut);
        }
    } else {
        run(w.W0, e.eL, true, ja, jb);
        if (e.far3 & 1) sep_run<3, 1>(acc, vn, hk, ja, jb, true, lq, cut); else sep_run<4, 1>(acc, vn, hk, ja, jb, true, lq, cut);
    }
    if (e.mL1 > e.mL0) {
        run(e.mL0, e.mL3, true, ja, jb); sep_run<3, 1>(acc, vn, hk, ja, jb, true, lq, cut);
        run(e.mL3, e.mL1, true, ja, jb); sep_run<4, 1>(acc, vn, hk, ja, jb, true, lq, cut);
    }
    if (phR) {
        const int p1 = w.W1;
        if (e.far3 & 2) {
            sep_run<3, 1, 1, 3>(acc, vn, hk, U[2], p1, false, lq, cut);
            sep_run<3, 1, 2, 2>(acc, vn, hk, U[1], U[2], false, lq, cut);
            sep_run<3, 1, 3, 1>(acc, vn, hk, U[0], U[1], false, lq, cut);
            sep_run<3, 1, 4, 0>(acc, vn, hk, U[3], U[0], false, lq, cut);
        } else {
            sep_run<4, 1, 1, 3>(acc, vn, hk, U[2], p1, false, lq, cut);
            sep_run<4, 1, 2, 2>(acc, vn, hk, U[1], U[2], false, lq, cut);
            sep_run<4, 1, 3, 1>(acc, vn, hk, U[0], U[1], false, lq, cut);
            sep_run<4, 1, 4, 0>(acc, vn, hk, U[3], U[0], false, lq, cut);
        }
    } else {
        run(e.eR, w.W1, false, ja, jb);
        if (e.far3 & 2) sep_run<3, 1>(acc, vn, hk, ja, jb, false, lq, cut); else sep_run<4, 1>(acc, vn, hk, ja, jb, false, lq, cut);
    }
    if (e.mR1 > e.mR0) {
        run(e.mR3, e.mR1, false, ja, jb); sep_run<3, 1>(acc, vn, hk, ja, jb, false, lq, cut);
        run(e.mR0, e.mR3, false, ja, jb); sep_run<4, 1>(acc, vn, hk, ja, jb, false, lq, cut);
    }
    if (e.cR > e.cL) {   // the core: pairs at least R apart
        run(e.cL, e.cR, true, ja, jb);
        if (e.far3 & 4) sep_run<8, 2>(acc, vn, hk, ja, jb, true, lq, cut, e.R); else sep_run<4, 2>(acc, vn, hk, ja, jb, true, lq, cut, e.R);
    }
    if (SPLIT == 1) {
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
            for (int r = 0; r < 4; r++) {   // D[state 4r + lq][point 16 st + lr]
                const int k = g * 16 + 4 * r + lq;
                const int64_t i = (int64_t)tile * 64 + st * 16 + lr;
                if (k < K && i < nnu) sigma[(size_t)k * nnu + i] += acc[st][r];
            }
        return;
    }
#pragma unroll
    for (int st = 0; st < 4; st++)
#pragma unroll
        for (int r = 0; r < 4; r++) part[wv][4 * r + lq][st * 16 + lr] = acc[st][r];
    __syncthreads();
    const int64_t i = (int64_t)tile * 64 + lane;
    for (int s4 = 0; s4 < 4; s4++) {
        const int ks = 4 * wv + s4, k = g * 16 + ks;
        if (k >= K || i >= nnu) break;
        sigma[(size_t)k * nnu + i] += ((part[0][ks][lane] + part[1][ks][lane]) + part[2][ks][lane]) + part[3][ks][lane];
    }
}

// K2f: the core of the window on 16-point sub-tiles.  The tile-wide near-zone pass of k_voigt_far runs its 37 instructions for every
// line within the TILE's reach on all 64 lanes, of which the few within 100 Doppler widths of the line need them.  Here one wave =
// one sub-tile of 16 points x 4 states (lane = 16 s + point; the record of (state, line) through a vector load, 16 lanes per
// address), so a line is visited by the sub-tiles within R of it only -- a third as many (lane, line) evaluations -- and only its
// pairs with |dnu| < R are summed; the others are k_voigt_edge_mx's (same comparison, complementary mask).  Same series and
// hand-off as the near-zone pass (six terms where s >= 1e3, index ranges of the s < 1e3 and s < 100 lines for k_voigt_near).
// One block = the four sub-tiles of a tile x the same four states; the per-(tile, state) hand-off flags are OR-ed through LDS.
// SW = points per sub-tile (16: 4 states per wave, 4 waves per block; 8: 8 states per wave, 8 waves per block = the 8 sub-tiles of
// the tile: a quarter fewer (lane, line) evaluations again, 0.38 -> 0.30 ms at C3)
template <int SW>
__global__ __launch_bounds__(4096 / SW) void k_voigt_sub(const double *__restrict__ nu, int64_t nnu, int64_t L, const LineHot *__restrict__ hot,
                                                         const double *__restrict__ gnul, const Zone *__restrict__ zones,
                                                         const EdgeZone *__restrict__ edge, int ntile, int K, double cut,
                                                         double *__restrict__ sigma, unsigned *__restrict__ rp, int prio, int assign)
{
    // assign != 0: `sigma` is the near-line plane of the step (its first writer on the side stream): every (state, point) of it is
    // WRITTEN here -- the sum where the tile has a core, a zero where it has not (and for the points of a ragged last tile) -- instead
    // of a memset of the whole plane in front of this kernel (49 MB at 0.7 TB/s = 72 us at the head of the near-line stream of the bench
    // column, 24 us on a 1/8 shard: profiles/r04_trace_c3.txt); assign == 0: added to the cross-sections where non-zero, as before
    wave_prio(prio);
    constexpr int NSW = 64 / SW;     // states per wave = sub-tiles per tile = waves per block
    __shared__ unsigned fl_sh[NSW][2];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = blockIdx.x, kq = blockIdx.y;   // states NSW kq .. NSW kq + NSW - 1 (all in one group of 16)
    const EdgeZone e = edge[(size_t)((kq * NSW) >> 4) * ntile + tile];
    const int s4 = lane / SW, pt = lane % SW;
    const int k = NSW * kq + s4;
    const bool kin = k < K;
    if (e.cR <= e.cL) {   // (block-uniform) no core here: nothing to sum
        if (assign && kin) {
            const int64_t i0 = (int64_t)tile * 64 + wv * SW + pt;
            if (i0 < nnu) sigma[(size_t)k * nnu + i0] = 0.0;
        }
        return;
    }
    const LineHot *__restrict__ hk = hot + (size_t)(kin ? k : K - 1) * L;
    const int64_t i = (int64_t)tile * 64 + wv * SW + pt;   // (a core exists on complete tiles only)
    const double v = nu[i];
    const double Rg = e.R;
    // the lines within Rg of the sub-tile (wave-uniform: first and last point of the sub-tile sit in lanes 0 and SW - 1)
    int ja = e.cL, jb = e.cR;
    {
        const double vlo = __builtin_bit_cast(double, ((uint64_t)__builtin_amdgcn_readlane((int)(__builtin_bit_cast(uint64_t, v) >> 32), 0) << 32) |
                                                          (uint32_t)__builtin_amdgcn_readlane((int)__builtin_bit_cast(uint64_t, v), 0));
        const double vhi = __builtin_bit_cast(double, ((uint64_t)__builtin_amdgcn_readlane((int)(__builtin_bit_cast(uint64_t, v) >> 32), SW - 1) << 32) |
                                                          (uint32_t)__builtin_amdgcn_readlane((int)__builtin_bit_cast(uint64_t, v), SW - 1));
        // one vector load per 64 lines of the core and two ballots instead of two binary searches (chains of dependent scalar
        // loads: most of what a wave with half a dozen lines to sum spent its time on)
        const double a0 = vlo - Rg, a1 = vhi + Rg;
        const int c0 = ja, c1 = jb;
        int first = c1, last = c0;   // first line >= a0, one past the last line <= a1
        for (int base = c0; base < c1; base += 64) {
            const int j = base + lane;
            const double x = gnul[j < c1 ? j : c1 - 1];
            const uint64_t mlo = __builtin_amdgcn_ballot_w64(j < c1 && x >= a0), mhi = __builtin_amdgcn_ballot_w64(j < c1 && x <= a1);
            if (mlo != 0 && first == c1) first = base + __builtin_ctzll(mlo);
            if (mhi != 0) last = base + 64 - __builtin_clzll(mhi);
        }
        ja = first;
        jb = max(last, first);
    }
    // (every operand of the series is a vector register here -- the record is per lane -- so all its constants can be scalar)
    const double k1p5 = sgpr_const(1.5), k3p75 = sgpr_const(3.75), k12 = vgpr_const(12.0), km15 = sgpr_const(-15.0), km105 = sgpr_const(-105.0),
                 k13p125 = sgpr_const(13.125), k210 = sgpr_const(210.0), km120 = vgpr_const(-120.0);
    const double q40 = sgpr_const(59.0625), q41 = sgpr_const(-787.5), q42 = sgpr_const(2835.0), q43 = sgpr_const(-3780.0), q44 = vgpr_const(1680.0);
    const double q50 = sgpr_const(324.84375), q51 = sgpr_const(-6496.875), q52 = sgpr_const(36382.5), q53 = sgpr_const(-83160.0),
                 q54 = sgpr_const(83160.0), q55 = vgpr_const(-30240.0);
    struct { double k1p5, k3p75, k12, km15, km105, k13p125, k210, km120; } c = {k1p5, k3p75, k12, km15, km105, k13p125, k210, km120};
    double acc = 0.0;
    int bl = 0x3fffffff, bh = -1, cl = 0x3fffffff, ch = -1;
    // one line: the near-zone pass's series for this lane's (state, point) pair, hand-off ranges of the pairs the series does not reach
    auto eval = [&](const LineHot &h, int j) {
        const double dv = v - h.nul;
        const double x = dv * h.p1;
        const double s = __builtin_fma(x, x, h.p2);
        const bool in = fabs(dv) < Rg && !(fabs(dv) > cut);
        const double u = rcp_fast(s);
        const double t = h.p2 * u;
        const double p5 = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(q55, t, q54), t, q53), t, q52), t, q51), t, q50);
        const double p4 = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(q44, t, q43), t, q42), t, q41), t, q40);
        const double p3 = __builtin_fma(__builtin_fma(__builtin_fma(c.km120, t, c.k210), t, c.km105), t, c.k13p125);
        const double p2 = __builtin_fma(__builtin_fma(c.k12, t, c.km15), t, c.k3p75);
        const double p1v = __builtin_fma(-2.0, t, c.k1p5);
        double P = __builtin_fma(u, p5, p4);
        P = __builtin_fma(u, P, p3);
        P = __builtin_fma(u, P, p2);
        P = __builtin_fma(u, P, p1v);
        P = __builtin_fma(u, P, 1.0);
        double term = (h.p3 * u) * P;
        asm volatile("" : "+v"(term));
        acc += (in && s >= kSerS) ? term : 0.0;
        if (in && s < kSerS) { bl = min(bl, j); bh = j; }
        if (in && s < kMidS) { cl = min(cl, j); ch = j; }
    };
    // (the records of the next TWO lines in flight, three buffers in rotation: k_voigt_sub alone 0.196 -> 0.184 ms, the step unchanged: round 5)
    LineHot cur = hk[ja < jb ? ja : 0];
    for (int j = ja; j < jb; j++) {
        const LineHot nxt = hk[min(j + 1, jb - 1)];   // in flight while this line is evaluated
        __builtin_amdgcn_sched_barrier(0);
        eval(cur, j);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    const int N0 = zones[(size_t)(kin ? k : K - 1) * ntile + tile].N0;
    const unsigned r0 = (kin && bh >= bl) ? ((unsigned)(bl - N0) << 12) | (unsigned)(bh + 1 - bl) : 0u;
    const unsigned r1 = (kin && ch >= cl) ? ((unsigned)(cl - N0) << 12) | (unsigned)(ch + 1 - cl) : 0u;
    const size_t plane = (size_t)K * nnu;
    if (kin) {
        const size_t o = (size_t)k * nnu + i;
        if (assign) sigma[o] = acc;
        else if (acc != 0.0) sigma[o] += acc;
        rp[o] = r0;            // (always written: whether the tile has candidates at all is known after the block's OR)
        rp[plane + o] = r1;
    }
    // flags "some lane of (tile, state) has candidates", one word per tier: bit s = state NSW kq + s
    const uint64_t b0 = __builtin_amdgcn_ballot_w64(r0 != 0u), b1 = __builtin_amdgcn_ballot_w64(r1 != 0u);
    if (lane == 0) {
        unsigned f0 = 0u, f1 = 0u;
        const uint64_t m = (1ull << SW) - 1ull;
        for (int q = 0; q < NSW; q++) {
            f0 |= ((b0 >> (SW * q)) & m) ? (1u << q) : 0u;
            f1 |= ((b1 >> (SW * q)) & m) ? (1u << q) : 0u;
        }
        fl_sh[wv][0] = f0; fl_sh[wv][1] = f1;
    }
    __syncthreads();
    if (threadIdx.x < 2 * NSW) {
        const int q = threadIdx.x % NSW, tier = threadIdx.x / NSW;
        unsigned f = 0u;
        for (int x = 0; x < NSW; x++) f |= fl_sh[x][tier];
        const int kk = NSW * kq + q;
        if (kk < K) (rp + 2 * plane)[((size_t)tier * K + kk) * ntile + tile] = (f >> q) & 1u;
    }
}

// ---- PHCO2 (Perrin & Hartmann sub-Lorentzian CO2 wings, line_shapes.jl:467-540) ------------------------------------------------
// PHCO2 = voigt with the Lorentz width scaled by chi(|dnu|): 1 below 3 cm^-1, then exp(a_r - b_r |dnu|) on [3,30), [30,120),
// [120, cut] with temperature-dependent a_r, b_r.  The default cut-off is 500 cm^-1, i.e. ~2e4 lines per point at 20 lines per
// cm^-1 -- all but a few hundred of them far enough from the whole 64-point tile to (i) sit in ONE region r and on one side for
// every lane and (ii) take the asymptotic far-wing body.  For those, chi = [exp(a_r -+ b_r (nu - nu_c))] x [exp(+- b_r (nul - nu_c))]:
// a per-lane factor formed once per wave and a per-(state, line) factor k_prep tabulates, so a pair costs the far body + 4
// multiplies, with wave-uniform scalar loads -- no exponential per pair.  What is left: the lines whose region boundary falls inside
// the tile (the lane picks the factorised chi of its own side: "boundary sets"), and the lines within 3 cm^-1 of the tile (chi = 1:
// handed to the Voigt kernels with that cut-off, or k_phco2's own core loop).  With the far wings interpolated (k_phco2_nodes,
// further down) the uniform sets shrink to what the smallest interval leaves of them.
// Per 64-point tile (state-independent): line indices of the region boundaries, computed by k_phwin.
struct PhWin {
    int32_t W0, L3, L2a, L2, L1a, L1, C0, C1, R1, R1b, R2, R2b, R3, W1;
    // uniform sets: [W0,L3) r=3 left | [L2a,L2) r=2 left | [L1a,L1) r=1 left | [R1,R1b) r=1 right | [R2,R2b) r=2 right | [R3,W1) r=3 right
    // boundary sets: [L3,L2a), [L2,L1a), [R1b,R2), [R2b,R3); core [L1,R1) (contains [C0,C1): every lane within 3 cm^-1)
    int32_t E0, E1;   // lines inside the cut-off of every lane (as WaveWin)
};
struct PhArgs { const double *nu, *nul; int64_t nnu; int ntile; int32_t J0, J1; double cut, tol; PhWin *out; };   // tol: one value per grid
__device__ __forceinline__ void phwin_body(unsigned bid, const PhArgs &a)
{
    const int t = bid * blockDim.x + threadIdx.x;
    if (t >= a.ntile) return;
    const int64_t i0 = (int64_t)t * 64, i1 = (i0 + 63 < a.nnu ? i0 + 63 : a.nnu - 1);
    const double vlo = a.nu[i0], vhi = a.nu[i1];
    const double *__restrict__ nul = a.nul;
    const double tol = a.tol;   // boundary lines go to the generic sets (chi is continuous across them); ONE value for the whole grid, so that
                                // the sets of a tile, of its interval and of that one's parents are nested exactly (phiwin_body)
    auto lower = [&](double val) { int lo = a.J0, hi = a.J1; while (lo < hi) { const int m = (lo + hi) >> 1; if (nul[m] < val) lo = m + 1; else hi = m; } return lo; };
    PhWin w;
    w.W0 = lower(vlo - a.cut - tol);
    w.W1 = lower(vhi + a.cut + tol);
    w.L3 = lower(vlo - 120.0 - tol);     // left of it: dnu >= 120 for every lane
    w.L2a = lower(vhi - 120.0 + tol);    // from here: dnu < 120 for every lane
    w.L2 = lower(vlo - 30.0 - tol);
    w.L1a = lower(vhi - 30.0 + tol);
    w.L1 = lower(vlo - 3.0 - tol);
    w.C0 = lower(vhi - 3.0 + tol);
    w.C1 = lower(vlo + 3.0 - tol);
    w.R1 = lower(vhi + 3.0 + tol);       // from here (right side): dnu >= 3 for every lane
    w.R1b = lower(vlo + 30.0 - tol);
    w.R2 = lower(vhi + 30.0 + tol);
    w.R2b = lower(vlo + 120.0 - tol);
    w.R3 = lower(vhi + 120.0 + tol);
    // a tile wider than a region: empty uniform sets (everything generic)
    w.L3 = max(w.L3, w.W0); w.L2a = max(w.L2a, w.L3); w.L2 = max(w.L2, w.L2a); w.L1a = max(w.L1a, w.L2); w.L1 = max(w.L1, w.L1a);
    w.R1 = max(w.R1, w.L1); w.R1b = max(w.R1b, w.R1); w.R2 = max(w.R2, w.R1b); w.R2b = max(w.R2b, w.R2); w.R3 = max(w.R3, w.R2b);
    w.W1 = max(w.W1, w.R3);
    w.C0 = min(max(w.C0, w.L1), w.R1); w.C1 = min(max(w.C1, w.C0), w.R1);
    w.E0 = min(max(lower(vhi - a.cut + tol), w.W0), w.W1);
    w.E1 = min(max(lower(vlo + a.cut - tol), w.E0), w.W1);
    a.out[t] = w;
}

// ---- PHCO2 far wings by interpolation ------------------------------------------------------------------------------------------
// Inside one chi-region and on one side, a far line's term A K(x, chi y) with chi = exp(a_r -+ b_r (nu - nul)) is analytic in nu, so
// the sums over the lines that are region-uniform for a whole INTERVAL of the grid go through the Chebyshev machinery of the Voigt
// path (k_cheb_setup's nodes and matrices, k_cheb_apply_mfma): 64 node evaluations instead of 128 .. 2048 point evaluations per
// line.  What keeps a line out of an interval's set is state-independent here: a region boundary (3, 30, 120 cm^-1 or the
// cut-off) crossing the interval, or the line being nearer than max(3 cm^-1, margin x half-width).  Per interval and (region,
// side) -- s = 0..5 in line order: r3 left, r2 left, r1 left, r1 right, r2 right, r3 right -- the uniform range [a[s], b[s]).
// The ranges of an interval contain those of its parent (next size up) and are contained in those of its tiles (PhWin): a level
// sums its ranges minus its parent's, and k_phco2 the tile's minus the smallest interval's.
struct PhIWin { int32_t a[6], b[6]; };
struct PhLevels { int nlev, nItot; int itv[CS_MAX_ALEVEL], nI[CS_MAX_ALEVEL], ioff[CS_MAX_ALEVEL]; };
// How many nodes an interval needs depends on how far its lines are in half-widths h: a set that stays D away converges like rho^-n,
// rho = x0 + sqrt(x0^2 - 1), x0 = 1 + D/h -- 64 nodes at the margin (rho = 2.1), but 16 or 32 for the far regions of the smaller
// intervals (region 3 from a 1024-point interval: D/h >= 9).  A "virtual level" is (interval size, node count, regions summed with
// it); a region that could hold nothing at a size is not carried there, and par[] names the next larger size that does carry it
// (the level whose ranges this one's are reduced by).
struct PhVLevels {
    int nv;
    int rl[CS_MAX_ALEVEL], nc[CS_MAX_ALEVEL], rmask[CS_MAX_ALEVEL];   // interval size (index into PhLevels), nodes per interval, regions (bit r = region r + 1)
    int noff[CS_MAX_ALEVEL], boff[CS_MAX_ALEVEL + 1];                // first node in F / in the node array, first interval slot in the launch
    int par[CS_MAX_ALEVEL][3];                                      // per region: the size above that carries it (-1: none)
};
// what k_phco2 leaves out of a tile's uniform sets, per region: the ranges of the smallest interval size that carries it
struct PhFine { int off[3], shift[3]; };                           // first PhIWin of that size (-1: none), log2(size / 64)
struct PhIArgs { const double *nu, *nul; int64_t nnu; int32_t J0, J1; double cut, tol, margin; PhLevels lv; PhIWin *out; };
__device__ __forceinline__ void phiwin_body(unsigned bid, const PhIArgs &A)
{
    const int q = bid * blockDim.x + threadIdx.x;
    if (q >= A.lv.nItot) return;
    int l = 0;
    while (l + 1 < A.lv.nlev && q >= A.lv.ioff[l + 1]) l++;
    const int T = q - A.lv.ioff[l], itv = A.lv.itv[l];
    const int64_t i0 = (int64_t)T * itv, i1 = (i0 + itv - 1 < A.nnu ? i0 + itv - 1 : A.nnu - 1);
    const double vlo = A.nu[i0], vhi = A.nu[i1];
    const double dZ = A.margin * 0.5 * (vhi - vlo), tol = A.tol, cut = A.cut;
    const double d1 = fmax(3.0, dZ), d2 = fmax(30.0, dZ), d3 = fmax(120.0, dZ);
    // lower bounds (first line with nul >= value), twelve searches side by side; every value is monotone in vlo, vhi and dZ with the
    // same tol as the tiles' (phwin_body), which is what nests the ranges
    const double sv[12] = {vhi - cut + tol, vlo - d3 - tol, vhi - 120.0 + tol, vlo - d2 - tol, vhi - 30.0 + tol, vlo - d1 - tol,
                           vhi + d1 + tol, vlo + 30.0 - tol, vhi + d2 + tol, vlo + 120.0 - tol, vhi + d3 + tol, vlo + cut - tol};
    int lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = A.J0; hi[i] = A.J1; }
    search_many<12>(A.nul, sv, 0u, lo, hi);
    PhIWin w;
#pragma unroll
    for (int s = 0; s < 6; s++) { w.a[s] = lo[2 * s]; w.b[s] = max(lo[2 * s + 1], lo[2 * s]); }
    // r3 left starts inside the cut-off of every point, r3 right ends there; sets in line order never overlap
#pragma unroll
    for (int s = 1; s < 6; s++) { w.a[s] = max(w.a[s], w.b[s - 1]); w.b[s] = max(w.b[s], w.a[s]); }
    A.out[q] = w;
}
// Within 3 cm^-1 of a line chi = 1: PHCO2 is the Voigt profile there, and those pairs -- every near-line pair among them -- go
// through the Voigt kernels with a 3 cm^-1 cut-off (k_voigt_far, k_voigt_near) after k_phco2 has summed the pairs beyond.  Their
// per-tile windows (WaveWin as wave_windows() builds them on the host) for that cut-off, XCD stretch table behind them:
struct WwArgs { const double *nu, *nul; int64_t nnu; int ntile; int32_t J0, J1; double cut; int sparse; WaveWin *out; };
__device__ __forceinline__ void wavewin_body(unsigned bid, const WwArgs &a)
{
    const int t = bid * blockDim.x + threadIdx.x;
    if (t >= a.ntile) return;
    const int64_t i0 = (int64_t)t * 64, i1 = (i0 + 63 < a.nnu ? i0 + 63 : a.nnu - 1);
    const double vlo = a.nu[i0], vhi = a.nu[i1], cut = a.cut;
    const double tol = 1e-9 * (fabs(vhi) + cut + 1.0);
    const double sv[4] = {vlo - cut - tol, vhi - cut + tol, vhi + cut + tol, vlo + cut - tol};
    int lo[4] = {a.J0, a.J0, a.J0, a.J0}, hi[4] = {a.J1, a.J1, a.J1, a.J1};
    search_many<4>(a.nul, sv, 0xcu, lo, hi);
    WaveWin w;
    w.W0 = lo[0]; w.W1 = max(lo[2], lo[0]);
    w.E0 = min(max(lo[1], w.W0), w.W1);
    w.E1 = min(max(lo[3], w.E0), w.W1);
    a.out[t] = w;
    if (t == 0) {   // tile_block()'s table: eight stretches of `per` tiles, or plain order for a table sparse against the grid
        int32_t *xc = reinterpret_cast<int32_t *>(a.out + a.ntile);
        const int nt4 = (a.ntile + 3) / 4 * 4, per = ((nt4 / 4 + 7) / 8) * 4;
        for (int x = 0; x <= 8; x++) xc[x] = min(x * per, nt4);
        for (int x = 9; x < 12; x++) xc[x] = 0;
        if (a.sparse) xc[0] = -1;
    }
}
__global__ __launch_bounds__(256) void k_phwin(unsigned nb_tiles, unsigned nb_itv, PhArgs a, PhIArgs ia, WwArgs wa)
{
    if (blockIdx.x < nb_tiles) phwin_body(blockIdx.x, a);
    else if (blockIdx.x < nb_tiles + nb_itv) phiwin_body(blockIdx.x - nb_tiles, ia);
    else wavewin_body(blockIdx.x - nb_tiles - nb_itv, wa);
}

// far-wing term with a per-lane Lorentz-width factor chi: y -> chi y.  FOUR: 4-term series (s >= 1e4), else 2-term + y-dependent u^2
// terms (exact wherever k_zones' Q bounds put a line, which it derived for the unscaled, i.e. larger, y)
template <bool PRED, bool FOUR>
__device__ __forceinline__ double ph_term(const LineHot &h, double chi, double v, double cut, const FarK &c)
{
    const double dv = v - h.nul;
    const double x = dv * h.p1;
    const double y2 = (h.p2 * chi) * chi;
    const double s = __builtin_fma(x, x, y2);
    const double u = rcp_fast(s);
    double P;
    if (FOUR) {
        const double t = y2 * u;
        const double p3 = __builtin_fma(__builtin_fma(__builtin_fma(c.km120, t, c.k210), t, c.km105), t, c.k13p125);
        const double p2 = __builtin_fma(__builtin_fma(c.k12, t, c.km15), t, c.k3p75);
        const double p1 = __builtin_fma(-2.0, t, c.k1p5);
        P = __builtin_fma(u, __builtin_fma(u, __builtin_fma(u, p3, p2), p1), 1.0);
    } else {
        const double t = y2 * u;
        const double q = __builtin_fma(t, __builtin_fma(c.k12, t, c.km15), __builtin_fma(y2, -2.0, c.k3p75));
        P = __builtin_fma(__builtin_fma(q, u, c.k1p5), u, 1.0);
    }
    double r = ((h.p3 * chi) * u) * P;
    if (PRED) r = (fabs(dv) > cut) ? 0.0 : r;
    return r;
}
template <bool PRED, bool FOUR>
__device__ __forceinline__ double ph_segment(double acc, double v, double glane, const LineHot *__restrict__ hk, const double *__restrict__ fk,
                                             int j0, int j1, double cut, const FarK &c)
{
#pragma unroll 4
    for (int j = j0; j < j1; j++) acc += ph_term<PRED, FOUR>(hk[j], glane * fk[j], v, cut, c);
    return acc;
}

template <bool PRED, bool FOUR>
__device__ __forceinline__ double ph_segment_rev(double acc, double v, double glane, const LineHot *__restrict__ hk, const double *__restrict__ fk,
                                                 int j0, int j1, double cut, const FarK &c)
{
#pragma unroll 4
    for (int j = j1 - 1; j >= j0; j--) acc += ph_term<PRED, FOUR>(hk[j], glane * fk[j], v, cut, c);
    return acc;
}
// chi = exp(a_r - b_r |dnu|) on [3,30), [30,120), [120, cut] (line_shapes.jl:467-481): a_r, b_r for r = 1..3
struct PhCoef { double a[3], b[3]; };
__device__ __forceinline__ PhCoef ph_coef(double T)
{
    PhCoef p;
    const double B1 = 0.0888 - 0.16 * exp(-0.0041 * T), B2 = 0.0526 * exp(-0.00152 * T), B3 = 0.0232;
    p.b[0] = B1; p.b[1] = B2; p.b[2] = B3;
    p.a[0] = 3.0 * B1; p.a[1] = -27.0 * B1 + 30.0 * B2; p.a[2] = -27.0 * B1 - 90.0 * B2 + 120.0 * B3;
    return p;
}

// one wave = one interval (of one virtual level) x one state: F[node][state] = sum over the interval's own ranges (its
// region-uniform ranges minus those of the size above), every line with the far body (all of them are >= 3 cm^-1 >= 100 Doppler
// widths away: phco2_fast_ok) and chi = (node factor) x (tabulated line factor).  Both sides are summed towards the interval.
// With nc = 64 nodes a lane is a node and the line records arrive as scalar loads; with 32 (16) the wave holds 2 (4) copies of the
// nodes, each taking a contiguous half (quarter) of every piece through vector loads, and the copies are added at the end.
#define CS_PH_PITCH 68   // doubles per record field in a wave's staging area: 4 parts x (16 + 1) or 2 parts x (32 + 1)
// the lines [j0, j1) for a wave that holds nparts = 2 or 4 copies of its nodes: 64 records at a time are fetched one per lane
// (coalesced) and laid out in LDS by (part, position) -- line i of the batch belongs to part i % nparts -- then every copy walks its
// own lines with broadcast reads (the parts' addresses fall on different banks).  Per-lane global loads of the records instead cost
// more than they save: four waves per CU x (2 x 16 B + 8 B) x 64 lanes per evaluation keep the CU's one address path busy longer
// than the arithmetic takes (measured: 32 / 16 nodes no faster than 64).
template <bool FOUR>
__device__ __forceinline__ double ph_segment_lds(double acc, double v, double glane, const LineHot *__restrict__ hk, const double *__restrict__ fk,
                                                 int j0, int j1, int part, int nparts, bool rev, double cut, const FarK &c, double *__restrict__ st)
{
    if (j1 <= j0) return acc;   // (wave-uniform)
    const int lane = threadIdx.x & 63;
    const int per = 64 / nparts, pitch = per + 1;
    const int wslot = (lane % nparts) * pitch + lane / nparts;   // where this lane's record goes
    for (int base = j0; base < j1; base += 64) {
        const int n = min(64, j1 - base);
        const int j = min(base + lane, j1 - 1);
        const LineHot h = hk[j];
        const double fj = fk[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();     // (the reads of the previous batch are done)
        st[0 * CS_PH_PITCH + wslot] = h.nul;
        st[1 * CS_PH_PITCH + wslot] = h.p1;
        st[2 * CS_PH_PITCH + wslot] = h.p2;
        st[3 * CS_PH_PITCH + wslot] = h.p3;
        st[4 * CS_PH_PITCH + wslot] = fj;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int tmax = (n + nparts - 1) / nparts;          // positions in use (part 0 has them all)
        const int cnt = (n - part + nparts - 1) / nparts;    // ... and this part: lines part, part + nparts, ...
#pragma unroll 2
        for (int t = 0; t < tmax; t++) {
            const int pos = rev ? tmax - 1 - t : t;
            const int idx = part * pitch + pos;
            LineHot hh;
            hh.nul = st[0 * CS_PH_PITCH + idx]; hh.p1 = st[1 * CS_PH_PITCH + idx]; hh.p2 = st[2 * CS_PH_PITCH + idx]; hh.p3 = st[3 * CS_PH_PITCH + idx];
            const double term = ph_term<false, FOUR>(hh, glane * st[4 * CS_PH_PITCH + idx], v, cut, c);
            acc += pos < cnt ? term : 0.0;
        }
    }
    return acc;
}
__global__ __launch_bounds__(256) void k_phco2_nodes(const double *__restrict__ nodes, int64_t L, const LineHot *__restrict__ hot,
                                                      const double *__restrict__ phfac, double nu_c, const PhIWin *__restrict__ piw,
                                                      PhLevels lv, PhVLevels vl, int K, int Kpad, const double *__restrict__ Tk, double cut,
                                                      const double *__restrict__ gbound, double mu_min, double mu_max, double far_s,
                                                      double *__restrict__ F)
{
    __shared__ double stage[4][5 * CS_PH_PITCH];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nsb = (K + 3) >> 2;
    const int slot = (int)(blockIdx.x / nsb);
    int vi = 0;
    while (vi + 1 < vl.nv && slot >= vl.boff[vi + 1]) vi++;
    const int T = slot - vl.boff[vi], rl = vl.rl[vi], nc = vl.nc[vi], rmask = vl.rmask[vi];
    const int k = (int)((blockIdx.x % nsb + T) % nsb) * 4 + wv;   // (rotated: see k_cheb_nodes)
    if (k >= K) return;
    const int node = lane & (nc - 1), nparts = 64 / nc, part = lane / nc;
    const PhIWin w = piw[lv.ioff[rl] + T];
    const LineHot *__restrict__ hk = hot + (size_t)k * L;
    const size_t KL = (size_t)K * L;
    const size_t n0 = (size_t)vl.noff[vi] + (size_t)T * nc;
    const double v = nodes[n0 + node];
    const FarK c = load_fark();
    const PhCoef pc = ph_coef(Tk[k]);
    const double dc = v - nu_c;
    // the two-term body does for a region whose every line has s >= max(cbrt(1.5e16 y^2), far_s) -- the bound behind Zone::Q0, Q1
    // (zones_body), with the widest Doppler width of the window for x and the narrowest for y (chi <= 1 only shrinks y)
    bool two[3];
    {
        const double vhi = nodes[n0], vlo = nodes[n0 + nc - 1];   // nodes run from the upper end down
        const double vth = sqrt(2.0 * kRgas * Tk[k]);
        const double amax = ((vhi + cut) / kC) * vth / sqrt(mu_min);
        const double vmin = w.b[5] > w.a[0] ? fmax(vlo - cut, hk[w.a[0]].nul) : vlo - cut;   // (the lowest line any range holds)
        double need = 1e300;
        if (vmin > 0.0) {
            const double amin = (vmin / kC) * vth / sqrt(mu_max), yb = gbound[k] * kSqLn2 / amin;
            need = fmax(cbrt(1.5e16 * yb * yb), far_s) * (1.0 + 1e-6);
        }
        const double D[3] = {3.0, 30.0, 120.0};
#pragma unroll
        for (int r = 0; r < 3; r++) { const double x = D[r] * kSqLn2 / amax; two[r] = x * x >= need; }
    }
    double accL = 0.0, accR = 0.0;
#pragma unroll
    for (int s = 0; s < 6; s++) {   // left of the interval: r = 3, 2, 1 with ascending lines; then right of it: r = 3, 2, 1, descending
        const bool left = s < 3;
        const int q = left ? s : 8 - s;           // set in line order: 0, 1, 2, 5, 4, 3
        const int r = left ? 2 - s : q - 3;
        if (!((rmask >> r) & 1)) continue;       // (wave-uniform)
        const double *__restrict__ f = phfac + (size_t)(left ? r : 3 + r) * KL + (size_t)k * L;   // exp(+-b_r (nul - nu_c))
        const double g = exp(left ? pc.a[r] - pc.b[r] * dc : pc.a[r] + pc.b[r] * dc);
        const int lo = w.a[q], hi = w.b[q];
        int pa = hi, pb = hi;
        const int prl = vl.par[vi][r];
        if (prl >= 0) {
            int pshift = 0;
            for (int x = lv.itv[prl] / lv.itv[rl]; x > 1; x >>= 1) pshift++;
            const PhIWin p = piw[lv.ioff[prl] + (T >> pshift)];
            pa = min(max(p.a[q], lo), hi);
            pb = min(max(p.b[q], pa), hi);
        }
        // far end first: [lo, pa) then [pb, hi) on the left, [pb, hi) then [lo, pa) on the right
        for (int piece = 0; piece < 2; piece++) {
            const bool first = (piece == 0) == left;
            const int j0 = first ? lo : pb, j1 = first ? pa : hi;
            double &acc = left ? accL : accR;
            if (nc == CS_NC) {
                if (left) acc = two[r] ? ph_segment<false, false>(acc, v, g, hk, f, j0, j1, cut, c) : ph_segment<false, true>(acc, v, g, hk, f, j0, j1, cut, c);
                else      acc = two[r] ? ph_segment_rev<false, false>(acc, v, g, hk, f, j0, j1, cut, c) : ph_segment_rev<false, true>(acc, v, g, hk, f, j0, j1, cut, c);
            } else {
                acc = two[r] ? ph_segment_lds<false>(acc, v, g, hk, f, j0, j1, part, nparts, !left, cut, c, stage[wv])
                             : ph_segment_lds<true>(acc, v, g, hk, f, j0, j1, part, nparts, !left, cut, c, stage[wv]);
            }
        }
    }
    double acc = accL + accR;
    if (nc <= 32) acc += __shfl_xor(acc, 32);
    if (nc <= 16) acc += __shfl_xor(acc, 16);
    if (lane < nc) F[(n0 + lane) * Kpad + k] = acc;
}

__global__ __launch_bounds__(256) void k_phco2(const double *__restrict__ nu, int64_t nnu, int64_t L, const LineHot *__restrict__ hot,
                                               const LineCold *__restrict__ cold, const double *__restrict__ phfac, double nu_c,
                                               const PhWin *__restrict__ pw, const Zone *__restrict__ zones, int ntile, double cut,
                                               const double *__restrict__ Tk, int K, double base, const double *__restrict__ extra,
                                               double *__restrict__ sigma, int accumulate, const PhIWin *__restrict__ piw, PhFine fine, int inner_voigt)
{
    // inner_voigt: the pairs within 3 cm^-1 (chi = 1) are left to the Voigt kernels that follow with that cut-off (wavewin_body)
    // piw != NULL: the region-uniform ranges of the tile's interval (at the smallest interval size that carries the region: fine) are
    // in sigma already, carried there from the node sums of k_phco2_nodes -- the tile's uniform sets shrink to what lies outside them
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * 4 + wv;
    if (tile >= ntile) return;
    const int k = blockIdx.y;
    const int64_t i = (int64_t)tile * 64 + lane;
    const double v = nu[i < nnu ? i : nnu - 1];
    const LineHot *__restrict__ hk = hot + (size_t)k * L;
    const LineCold *__restrict__ ck = cold + (size_t)k * L;
    const size_t KL = (size_t)K * L;
    const PhWin w = pw[tile];
    const Zone z = zones[(size_t)k * ntile + tile];
    const FarK c = load_fark();
    const PhCoef pc = ph_coef(Tk[k]);
    const double dc = v - nu_c;
    // per-lane factors of chi, left (line below the point) and right, r = 1..3
    double gL[3], gR[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { gL[r] = exp(pc.a[r] - pc.b[r] * dc); gR[r] = exp(pc.a[r] + pc.b[r] * dc); }
    const double *__restrict__ fL[3] = {phfac + 0 * KL + (size_t)k * L, phfac + 1 * KL + (size_t)k * L, phfac + 2 * KL + (size_t)k * L};
    const double *__restrict__ fR[3] = {phfac + 3 * KL + (size_t)k * L, phfac + 4 * KL + (size_t)k * L, phfac + 5 * KL + (size_t)k * L};
    PhIWin x[3];   // what the intervals took, per region
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (piw && fine.off[r] >= 0) x[r] = piw[fine.off[r] + (tile >> fine.shift[r])];
    const int slo[6] = {w.W0, w.L2a, w.L1a, w.R1, w.R2, w.R3}, shi[6] = {w.L3, w.L2, w.L1, w.R1b, w.R2b, w.W1};
    double acc = 0.0;
    // uniform sets: region r, side; inside [Q0,Q1) (k_zones) the 4-term series, outside the 2-/3-term body; cut-off edges with the
    // predicate (they sit in region 3: the launcher only takes this kernel for cut-offs beyond 130 cm^-1)
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const bool left = s < 3;
        const int r = left ? 2 - s : s - 3;
        const double *__restrict__ f = left ? fL[r] : fR[r];
        const double g = left ? gL[r] : gR[r];
        const int lo = slo[s], hi = shi[s];
        const bool took = piw && fine.off[r] >= 0;
        const int xa = took ? min(max(x[r].a[s], lo), hi) : hi, xb = took ? min(max(x[r].b[s], xa), hi) : hi;
        for (int piece = 0; piece < 2; piece++) {
            const int p0 = piece ? xb : lo, p1 = piece ? hi : xa;
            if (p0 >= p1) continue;
#define LO(x) max((x), p0)
#define HI(x) min((x), p1)
            if (left) {
                // [lo, e) cut-off edge (r = 3 only) | [e, q) far | [q, hi) 4-term
                const int e = s == 0 ? min(max(w.E0, lo), hi) : lo, q = min(max(z.Q0, e), hi);
                if (s == 0) acc = ph_segment<true, false>(acc, v, g, hk, f, LO(lo), HI(e), cut, c);
                acc = ph_segment<false, false>(acc, v, g, hk, f, LO(e), HI(q), cut, c);
                acc = ph_segment<false, true>(acc, v, g, hk, f, LO(q), HI(hi), cut, c);
            } else {
                const int e = s == 5 ? max(min(w.E1, hi), lo) : hi, q = min(max(z.Q1, lo), e);
                acc = ph_segment<false, true>(acc, v, g, hk, f, LO(lo), HI(q), cut, c);
                acc = ph_segment<false, false>(acc, v, g, hk, f, LO(q), HI(e), cut, c);
                if (s == 5) acc = ph_segment<true, false>(acc, v, g, hk, f, LO(e), HI(hi), cut, c);
            }
#undef LO
#undef HI
        }
    }
    // boundary sets: a region boundary D = 120 or 30 cm^-1 crosses the tile -- the lane picks the side's factorised chi of its own
    // region; all of these lines are far (4-term body)
    {
        const int bl[4] = {w.L3, w.L2, w.R1b, w.R2b}, bh[4] = {w.L2a, w.L1a, w.R2, w.R3};
        const double D[4] = {120.0, 30.0, 30.0, 120.0};
        const int rn[4] = {1, 0, 0, 1};   // region (0-based) on the near side of the boundary; the far side is rn + 1
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const bool left = b < 2;
            const double gn = left ? gL[rn[b]] : gR[rn[b]], gf = left ? gL[rn[b] + 1] : gR[rn[b] + 1];
            const double *__restrict__ fn = left ? fL[rn[b]] : fR[rn[b]], *__restrict__ ff = left ? fL[rn[b] + 1] : fR[rn[b] + 1];
#pragma unroll 2
            for (int j = bl[b]; j < bh[b]; j++) {
                const LineHot h = hk[j];
                const double chi = fabs(v - h.nul) < D[b] ? gn * fn[j] : gf * ff[j];
                acc += ph_term<true, true>(h, chi, v, cut, c);
            }
        }
    }
    // core [L1, R1): lines within 3 cm^-1 of some lane -- chi = 1 there, region 1 beyond; the full Faddeeva (line_shapes.jl:496-499)
    // only for the lines that have a lane inside s < 1e4 (wave-uniform test), the 4-term body for the others
    if (inner_voigt) {
        // [L1, C0) and [C1, R1): lines with lanes on both sides of 3 cm^-1 -- the lanes beyond it here ([C0, C1): every lane inside)
        // (the side is the lane's own: on a tile wider than 3 cm^-1 such a line can lie inside it)
        for (int half = 0; half < 2; half++) {
            const int j0 = half ? max(w.C1, w.C0) : w.L1, j1 = half ? w.R1 : w.C0;
#pragma unroll 2
            for (int j = j0; j < j1; j++) {
                const LineHot h = hk[j];
                const double dv = v - h.nul;
                const double t = ph_term<false, true>(h, dv > 0.0 ? gL[0] * fL[0][j] : gR[0] * fR[0][j], v, cut, c);
                acc += fabs(dv) > 3.0 ? t : 0.0;
            }
        }
    } else
    for (int j = w.L1; j < w.R1; j++) {
        const LineHot h = hk[j];
        const double dv = v - h.nul;
        const double chi = fabs(dv) < 3.0 ? 1.0 : (dv > 0.0 ? gL[0] * fL[0][j] : gR[0] * fR[0][j]);
        const double xx = dv * h.p1;
        const bool nearl = __builtin_fma(xx, xx, (h.p2 * chi) * chi) < 1.0e4;
        if (__builtin_amdgcn_ballot_w64(nearl) != 0ull) {
            const LineCold cc = ck[j];
            const double t = cc.A * fad_re(xx, chi * cc.y);
            acc += (fabs(dv) > cut) ? 0.0 : t;
        } else {
            acc += ph_term<true, true>(h, chi, v, cut, c);
        }
    }
    if (i < nnu) {
        const size_t o = (size_t)k * nnu + i;
        const double prev = accumulate ? sigma[o] : (base + (extra ? extra[o] : 0.0));
        sigma[o] = prev + acc;
    }
}

// K2b: the pairs with s < 1e3.  k_voigt_far left every lane the index ranges of its own such lines.  A wave compacts
// the (lane, line) candidates of its 64 lanes into an LDS queue and evaluates them 64 at a time -- so the expensive forms
// (continued fraction for 100 <= s < 1e3, trapezoid + pole correction for s < 100) run with full lanes instead of
// once per "deepest" lane -- then every lane sums its own results in ascending line order (deterministic).
#define CS_NEAR_Q 256  // queue entries per wave; longer candidate lists go through the queue in windows
#define CS_NEAR_R 1    // consecutive 64-point tiles per wave (2 and 4 fill the 64-wide trips better but measured slower: 0.47-0.58 vs 0.45 ms;
static_assert(CS_NEAR_R == 1, "queue entries carry no sub-tile field");   //  a sub-tile field would have to come out of the 26 line-index bits)
// candidates of lane l: for r = 0..R-1 the lines [lo[r], hi[r]) (absolute indices) against wavenumber (tile0 + r) * 64 + l
template <int TIER>  // 0: 100 <= s < 1e3 (fad_mid), 1: s < 100 (fad_near)
__device__ __forceinline__ void near_pass(const double *__restrict__ nu, int64_t nnu, int tile0, const int (&lo)[CS_NEAR_R],
                                          const int (&hi)[CS_NEAR_R], double (&acc)[CS_NEAR_R], const LineHot *__restrict__ hk,
                                          const LineCold *__restrict__ ck, double cut, unsigned *qidx, double *qres)
{
    const int lane = threadIdx.x & 63;
    int cnt = 0, pre[CS_NEAR_R + 1];
#pragma unroll
    for (int r = 0; r < CS_NEAR_R; r++) { pre[r] = cnt; cnt += hi[r] - lo[r]; }
    pre[CS_NEAR_R] = cnt;
    // exclusive prefix sum of the candidate counts over the wave (rocPRIM: DPP row shifts + broadcasts, no LDS round trips)
    int incl;
    {
        using scan_t = rocprim::warp_scan<int, 64>;
        typename scan_t::storage_type st;
        scan_t().inclusive_scan(cnt, incl, st);
    }
    const int off = incl - cnt;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    for (int base = 0; base < total; base += CS_NEAR_Q) {   // wave-uniform
        // this lane's candidates that fall into the window [base, base + Q): entry = lane | sub-tile | line
#pragma unroll
        for (int r = 0; r < CS_NEAR_R; r++) {
            const int o = off + pre[r] - base;   // queue position of this sub-tile's first candidate
            const int c0 = max(-o, 0), c1 = min(CS_NEAR_Q - o, hi[r] - lo[r]);
            for (int c = c0; c < c1; c++) qidx[o + c] = ((unsigned)lane << 26) | (unsigned)(lo[r] + c);   // 6 + 26 bits: cs_gas_upload rejects L >= 2^26
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int nwin = min(total - base, CS_NEAR_Q);
        const int ntrip = (nwin + 63) >> 6;
        // the line records of trip it+1 are fetched while trip it is evaluated (a wave is a chain of dependent gathers otherwise)
        LineHot hn = {};
        LineCold cn = {};
        unsigned en = 0;
        double vn = 0.0;
        auto fetch = [&](int p) {
            en = qidx[p];
            const int j = (int)(en & 0x3ffffffu);
            hn = hk[j];
            cn = ck[j];
            const int64_t i = (int64_t)tile0 * 64 + (int)(en >> 26);
            vn = nu[i < nnu ? i : nnu - 1];
        };
        if (lane < nwin) fetch(lane);
        for (int it = 0; it < ntrip; it++) {
            const int p = it * 64 + lane;
            const bool live = p < nwin;
            const LineHot h = hn;
            const LineCold c = cn;
            const double vo = vn;
            if (p + 64 < nwin) fetch(p + 64);
            if (live) {
                double r = 0.0;
                const double dv = vo - h.nul;
                const double x = dv * h.p1;
                const double s = __builtin_fma(x, x, h.p2);
                const bool mine = TIER == 0 ? (s < kSerS && s >= kMidS) : (s < kMidS);
                if (!(fabs(dv) > cut) && mine) r = c.A * (TIER == 0 ? fad_mid(fabs(x), c.y) : fad_near(fabs(x), c.y));
                qres[p] = r;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < CS_NEAR_R; r++) {   // ascending line order per (lane, sub-tile): deterministic
            const int o = off + pre[r] - base;
            const int c0 = max(-o, 0), c1 = min(CS_NEAR_Q - o, hi[r] - lo[r]);
            for (int c = c0; c < c1; c++) acc[r] += qres[o + c];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// one launch per tier: the tier-0 kernel is light (continued fraction, few registers, many waves in flight), the tier-1 kernel
// carries the trapezoid + pole correction (exp, sincospi).  One wave = CS_NEAR_R consecutive tiles x one state.  The register
// allocator would give tier 1 161 VGPRs (3 waves per SIMD); capped at 96 (5 waves) it still does not spill and runs 12 % faster.
template <int TIER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_voigt_near(const double *__restrict__ nu, int64_t nnu, int64_t L,
                                                     const LineHot *__restrict__ hot, const LineCold *__restrict__ cold,
                                                     const Zone *__restrict__ zones, int ntile, int ngrp, int nrep, double cut,
                                                     double *__restrict__ sigma, const int2 *__restrict__ ranges, int prio)
{
    wave_prio(prio);
    // nrep consecutive tiles per wave, one after the other: a wave that only reads its flag and exits still costs its launch -- 7.9e5 of
    // them per kernel at BASELINE configs[4], 0.2 ms each for flags and ranges alone (measured with the kernel cut short), whatever the
    // gas.  Eight tiles per wave there: near-line kernels 1.27 -> 1.06 ms
    __shared__ unsigned qidx_s[4][CS_NEAR_Q];
    __shared__ double qres_s[4][CS_NEAR_Q];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * 4 + wv) * nrep);   // first group of CS_NEAR_R tiles
    const int k = blockIdx.y;
    const LineHot *__restrict__ hk = hot + (size_t)k * L;
    const LineCold *__restrict__ ck = cold + (size_t)k * L;
    // two planes of packed words: tier 0 ranges, then tier 1 ranges, [gridDim.y][nnu] each; behind them the per-(tile, node) flags
    const unsigned *__restrict__ fl = reinterpret_cast<const unsigned *>(ranges) + 2 * (size_t)gridDim.y * nnu + ((size_t)TIER * gridDim.y + k) * ntile;
    const unsigned *__restrict__ rp = reinterpret_cast<const unsigned *>(ranges) + (size_t)TIER * gridDim.y * nnu + (size_t)k * nnu;
    // (the flags of the wave's tiles -- nrep <= 8 -- are requested together: one round trip, not one per tile)
    unsigned livemask = 0u;
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (q < nrep && grp0 + q < ngrp && fl[grp0 + q] != 0u) livemask |= 1u << q;
    livemask = __builtin_amdgcn_readfirstlane(livemask);
    for (int grp = grp0; grp < min(grp0 + nrep, ngrp); grp++) {
        if (!((livemask >> (grp - grp0)) & 1u)) continue;   // (wave-uniform: CS_NEAR_R = 1, group = tile)
        const int tile0 = grp * CS_NEAR_R;
        int lo[CS_NEAR_R], hi[CS_NEAR_R];
        double acc[CS_NEAR_R];
#pragma unroll
        for (int r = 0; r < CS_NEAR_R; r++) {
            const int tile = tile0 + r;
            const int64_t i = (int64_t)tile * 64 + lane;
            lo[r] = hi[r] = 0;
            acc[r] = 0.0;
            if (tile < ntile && i < nnu) {
                const unsigned q = rp[i];
                const int N0 = zones[(size_t)k * ntile + tile].N0;
                lo[r] = N0 + (int)(q >> 12);
                hi[r] = lo[r] + (int)(q & 0xfffu);
            }
        }
        near_pass<TIER>(nu, nnu, tile0, lo, hi, acc, hk, ck, cut, qidx_s[wv], qres_s[wv]);
#pragma unroll
        for (int r = 0; r < CS_NEAR_R; r++) {
            const int64_t i = (int64_t)(tile0 + r) * 64 + lane;
            if (tile0 + r < ntile && i < nnu && acc[r] != 0.0) sigma[(size_t)k * nnu + i] += acc[r];
        }
    }
}

// both tiers in one launch (one tile per wave): flags, ranges, zone and the cross-section the lane adds to are requested once and
// together, tier 0 then tier 1 go through the wave's queue, one store -- (sigma + tier 0) + tier 1, what the two launches leave.  For
// every grid on which a wave takes one tile (all but the half-million-wave ones): one launch and its gap less, the zone and the flags
// read once -- a 1/8 shard of the bench column 0.367 -> 0.360 ms, BASELINE configs[1] 0.136 -> 0.132, the bench column 1.947 -> 1.940
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_voigt_near_both(const double *__restrict__ nu, int64_t nnu, int64_t L,
                                                     const LineHot *__restrict__ hot, const LineCold *__restrict__ cold,
                                                     const Zone *__restrict__ zones, int ntile, double cut,
                                                     double *__restrict__ sigma, const int2 *__restrict__ ranges, int prio)
{
    wave_prio(prio);
    __shared__ unsigned qidx_s[4][CS_NEAR_Q];
    __shared__ double qres_s[4][CS_NEAR_Q];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv);
    const int k = blockIdx.y;
    if (tile >= ntile) return;
    const unsigned *__restrict__ words = reinterpret_cast<const unsigned *>(ranges);
    const size_t plane = (size_t)gridDim.y * nnu;
    const unsigned *__restrict__ fl = words + 2 * plane;
    const unsigned f0 = fl[(size_t)k * ntile + tile], f1 = fl[((size_t)gridDim.y + k) * ntile + tile];
    if (f0 == 0u && f1 == 0u) return;            // (wave-uniform)
    const int64_t i = (int64_t)tile * 64 + lane;
    const bool in = i < nnu;
    const size_t o = (size_t)k * nnu + (in ? i : nnu - 1);
    const unsigned q0 = (f0 != 0u && in) ? words[o] : 0u, q1 = (f1 != 0u && in) ? words[plane + o] : 0u;
    const int N0 = zones[(size_t)k * ntile + tile].N0;
    const double prev = sigma[o];
    const LineHot *__restrict__ hk = hot + (size_t)k * L;
    const LineCold *__restrict__ ck = cold + (size_t)k * L;
    double a0[1] = {0.0}, a1[1] = {0.0};
    if (f0 != 0u) {
        const int lo[1] = {N0 + (int)(q0 >> 12)}, hi[1] = {lo[0] + (int)(q0 & 0xfffu)};
        near_pass<0>(nu, nnu, tile, lo, hi, a0, hk, ck, cut, qidx_s[wv], qres_s[wv]);
    }
    if (f1 != 0u) {
        const int lo[1] = {N0 + (int)(q1 >> 12)}, hi[1] = {lo[0] + (int)(q1 & 0xfffu)};
        near_pass<1>(nu, nnu, tile, lo, hi, a1, hk, ck, cut, qidx_s[wv], qres_s[wv]);
    }
    if (in && (a0[0] != 0.0 || a1[0] != 0.0)) sigma[o] = (a0[0] != 0.0 ? prev + a0[0] : prev) + a1[0];
}

// exp for the flux kernel, where it is half the instructions: Cody-Waite reduction x = n ln2 + r, |r| <= ln2 / 2, Taylor polynomial
// of degree 13 (truncation 4e-18), 2^n by v_ldexp_f64 -- 20 instructions against the library's ~28 (no special cases occur here:
// arguments are finite; far below -745 the result is the 0 it should be)
__device__ __forceinline__ double exp_rt(double x)
{
    x = fmax(x, -1000.0);
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(n, -0.693147180369123816490, x);     // ln 2, high part (trailing zeros: n ln2_hi is exact)
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);         // ... low part
    double p = 1.6059043836821613e-10;                            // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);                // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);               // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);               // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);              // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);                // 1/8!
    p = __builtin_fma(p, r, 0.0001984126984126984);               // 1/7!
    p = __builtin_fma(p, r, 0.001388888888888889);                // 1/6!
    p = __builtin_fma(p, r, 0.008333333333333333);                // 1/5!
    p = __builtin_fma(p, r, 0.041666666666666664);                // 1/4!
    p = __builtin_fma(p, r, 0.16666666666666666);                 // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// radiation.jl:48-54
__device__ __forceinline__ double planck(double nu, double T)
{
    double num = 100.0 * nu;
    double x = kHp * kC * num / (kKb * T);
    double p = 2.0 * kHp * (kC * kC) * (num * num * num);
    return 100.0 * p / (exp_rt(x) - 1.0);
}

// discretized.jl:85-87
__device__ __forceinline__ double layerplanck(double B1, double B2, double tau, double t)
{
    return B2 * (1.0 - t) - (B1 - B2) * t + (1.0 - t) * (B1 - B2) / tau;
}
// the same with 1/tau handed in: k_rt forms 1/(t m_k) as (1/t)(1/m_k), one division per layer instead of one per stream
__device__ __forceinline__ double layerplanck_inv(double B1, double B2, double itau, double t)
{
    return B2 * (1.0 - t) - (B1 - B2) * t + ((1.0 - t) * (B1 - B2)) * itau;
}

struct RtParams {
    int np, nlobatto, K, nstream;
    double C;        // 1e-4*Na/g, fluxes.jl:259
    double cos_ts;   // cos(theta_s)
    double ws[16];   // Lobatto weights on [0,1]
    double m[16];    // 1/cos(theta_k)
    double W[16];    // stream weights
    double im[16];   // cos(theta_k) = 1/m
};

// sum over the wave, valid in lane 0 (rocPRIM: DPP row shifts instead of six LDS-routed shuffles -- k_rt does this twice per
// level inside a chain of dependent work)
__device__ __forceinline__ double wave_sum(double v)
{
    using red_t = rocprim::warp_reduce<double, 64>;
    typename red_t::storage_type st;
    double r;
    red_t().reduce(v, r, st);
    return r;
}

// K3: one lane per wavenumber.  Pass 1 walks TOA -> surface: layer optical depths from the node cross-sections
// (dDepth!) and the downward sweep of all NS streams at once; pass 2 walks surface -> TOA (upward sweep).
// All NS intensities advance together, so M-[i+1] = sum_k W_k I_k + stellar beam is complete when layer i is done --
// same summation order as the reference's stream-outer loops, no [np] scratch per lane.
// UD = false: one wave runs both passes; tau is kept nu-fastest in HBM between them when the caller wants it (tau != NULL),
//             otherwise the upward pass recomputes it from the cross-sections and no optical depth is ever stored.
// UD = true : the two passes of a 64-point tile run in TWO waves of the block side by side (waves [0,nw) go down, [nw,2nw) go
//             up and recompute the layer optical depths they need on the way) -- each pass is a chain of ~300 dependent fp64
//             operations per layer that no amount of lanes shortens, so with few waves per SIMD (a 1e5-point grid is 1.5, a
//             nu-shard less) halving the chain is what halves the time.  The upward sweep starts from the surface intensity,
//             which depends on the downward flux only through the albedo: with an albedo array the up-waves wait for it at a
//             barrier (no overlap, same result), without one they start at once.
// Mup/Mdn are optional outputs.  red : per-block partial sums of w_j*M[i][j], layout [block][2*np] (Fup then Fdn).
template <int NS, bool UD>
__global__ __launch_bounds__(256) void k_rt(RtParams p, const double *__restrict__ nu, const double *__restrict__ wts,
                                             int64_t nnu, const double *__restrict__ sigma,
                                             const double *__restrict__ muk, const double *__restrict__ P,
                                             const double *__restrict__ Tlev, const double *__restrict__ S_toa,
                                             const double *__restrict__ albedo, double *__restrict__ tau,
                                             double *__restrict__ Mup, double *__restrict__ Mdn,
                                             double *__restrict__ partial, size_t sig_bstride, const double *__restrict__ sigma2)
{
    // sigma2 != NULL: the near-line pairs of the step were summed into a plane of their own (k_voigt_near on a side stream beside the
    // matrix-core kernels): the cross-section is sigma + sigma2
    extern __shared__ double red[];  // [2*np][nw] (+ UD: [nw][64] surface downward flux)
    const int nwv = blockDim.x >> 6;
    const int nw = UD ? nwv >> 1 : nwv;   // 64-point tiles per block: 4 (2 with UD) for big grids, 1 for small ones (more blocks than CUs)
    {   // blockIdx.y = column of a batch (cs_column_batch): same grid, pressures and stream rule, its own node states
        const size_t b = blockIdx.y;
        sigma += b * sig_bstride;     // K*nnu, or 0 when the columns of a batch share their cross-sections (AcceleratedAbsorber)
        muk += b * p.K;
        Tlev += b * p.np;
        if (tau) tau += b * (size_t)(p.np - 1) * nnu;
        partial += b * (size_t)gridDim.x * 2 * p.np;
    }
    const int lane = threadIdx.x & 63;
    const int wva = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool up_role = UD && wva >= nw, down_role = !UD || wva < nw;
    const int wv = up_role ? wva - nw : wva;
    const int64_t j = ((int64_t)blockIdx.x * nw + wv) * 64 + lane;
    const bool live = j < nnu;
    const int64_t jj = live ? j : nnu - 1;
    const int np = p.np, nl = np - 1, nlob = p.nlobatto;
    double *xch = red + (size_t)2 * np * nw;
    const double v = nu[jj];
    const double w = live ? wts[jj] : 0.0;
    const double fS = S_toa ? S_toa[jj] : 0.0;
    const double fa = albedo ? albedo[jj] : 0.0;
    const double c = p.cos_ts;
    auto sg = [&](size_t idx) { return sigma2 ? sigma[idx] + sigma2[idx] : sigma[idx]; };

    double I[NS];
    double Md = 0.0, Bprev = 0.0;
    if (down_role) {
#pragma unroll
        for (int k = 0; k < NS; k++) I[k] = 0.0;
        double b1 = p.C * (sg((size_t)jj) / muk[0]);  // beta at node 0, discretized.jl:150
        double Ms = c * fS;                      // M-[1] = c*fS(nu), discretized.jl:299
        Md = Ms;
        Bprev = planck(v, Tlev[0]);
        {
            double r = wave_sum(w * Md);
            if (lane == 0) red[(np + 0) * nw + wv] = r;
            if (Mdn && live) Mdn[j] = Md;
        }
        double sg_next = sg((size_t)(nlob - 1) * nnu + jj);   // end node of layer 0; later layers are fetched one layer ahead
        for (int i = 0; i < nl; i++) {
            const double dP = P[i + 1] - P[i];
            double ti = (dP * p.ws[0]) * b1;
            for (int n = 1; n < nlob - 1; n++) {
                const int k = i * (nlob - 1) + n;
                ti += (dP * p.ws[n]) * (p.C * (sg((size_t)k * nnu + jj) / muk[k]));
            }
            const int ke = (i + 1) * (nlob - 1);
            const double sgv = sg_next;
            if (i + 1 < nl) sg_next = sg((size_t)(ke + nlob - 1) * nnu + jj);   // in flight during this layer's exp/divide chain
            const double bn = p.C * (sgv / muk[ke]);
            ti += (dP * p.ws[nlob - 1]) * bn;
            b1 = bn;
            const double t = ti > 1e-6 ? ti : 1e-6;  // floor, discretized.jl:147,174
            if (tau && live) tau[(size_t)i * nnu + j] = t;
            const double Bnext = planck(v, Tlev[i + 1]);
            Md = 0.0;
            const double it = 1.0 / t;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const double tk = t * p.m[k];
                const double tr = exp_rt(-tk);
                const double Be = layerplanck_inv(Bprev, Bnext, it * p.im[k], tr);
                I[k] = I[k] * tr + Be;
                Md += p.W[k] * I[k];
            }
            if (S_toa) Ms *= exp(-t / c);   // (wave-uniform; without a stellar beam Ms stays 0)
            Md += Ms;
            Bprev = Bnext;
            double r = wave_sum(w * Md);
            if (lane == 0) red[(np + i + 1) * nw + wv] = r;
            if (Mdn && live) Mdn[(size_t)(i + 1) * nnu + j] = Md;
        }
        if (UD && albedo) xch[wv * 64 + lane] = Md;
    }
    if (UD && albedo) __syncthreads();   // (block-uniform condition) the up-waves need the surface downward flux of their tile
    if (!down_role || !UD) {
        if (UD) {
            Bprev = planck(v, Tlev[np - 1]);
            Md = albedo ? xch[wv * 64 + lane] : 0.0;
        }
        // surface: Lambertian reflection + Planck emission, discretized.jl:309-310
        const double Is = Md * fa / kPi + Bprev;
        double Mu = Is * kPi;
        {
            double r = wave_sum(w * Mu);
            if (lane == 0) red[(np - 1) * nw + wv] = r;
            if (Mup && live) Mup[(size_t)(np - 1) * nnu + j] = Mu;
        }
#pragma unroll
        for (int k = 0; k < NS; k++) I[k] = Is;
        double Bhi = Bprev;  // B at level i+1
        // layer optical depths on the way up: re-read from HBM (one wave ran both passes) or recomputed exactly as the
        // downward pass forms them (UD: that pass runs beside this one and has not written them yet)
        const bool recompute = UD || !tau;   // no tau output requested ("OLR-only"): optical depths never touch HBM
        double t_next = 1.0, b_hi = 0.0, sg_lo = 0.0;
        if (recompute) {
            b_hi = p.C * (sg((size_t)(p.K - 1) * nnu + jj) / muk[p.K - 1]);
            sg_lo = sg((size_t)(nl - 1) * (nlob - 1) * nnu + jj);
        } else {
            t_next = live ? tau[(size_t)(nl - 1) * nnu + j] : 1.0;
        }
        for (int i = nl - 1; i >= 0; i--) {
            double t;
            if (recompute) {
                const double dP = P[i + 1] - P[i];
                const int kl = i * (nlob - 1);
                const double b_lo = p.C * (sg_lo / muk[kl]);
                if (i > 0) sg_lo = sg((size_t)(kl - (nlob - 1)) * nnu + jj);   // one layer ahead
                double ti = (dP * p.ws[0]) * b_lo;
                for (int n = 1; n < nlob - 1; n++) ti += (dP * p.ws[n]) * (p.C * (sg((size_t)(kl + n) * nnu + jj) / muk[kl + n]));
                ti += (dP * p.ws[nlob - 1]) * b_hi;
                b_hi = b_lo;
                t = ti > 1e-6 ? ti : 1e-6;
            } else {
                t = t_next;
                if (i > 0) t_next = live ? tau[(size_t)(i - 1) * nnu + j] : 1.0;   // one layer ahead
            }
            const double Blo = planck(v, Tlev[i]);
            Mu = 0.0;
            const double it = 1.0 / t;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const double tk = t * p.m[k];
                const double tr = exp_rt(-tk);
                const double Be = layerplanck_inv(Bhi, Blo, it * p.im[k], tr);
                I[k] = I[k] * tr + Be;
                Mu += p.W[k] * I[k];
            }
            Bhi = Blo;
            double r = wave_sum(w * Mu);
            if (lane == 0) red[i * nw + wv] = r;
            if (Mup && live) Mup[(size_t)i * nnu + j] = Mu;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * np; e += blockDim.x) {
        const double *q = red + e * nw;
        double t = q[0];
        for (int x = 1; x < nw; x++) t += q[x];
        partial[(size_t)blockIdx.x * 2 * np + e] = t;
    }
}

// K3 for short grids (a nu-shard, a small column): there a sweep is one chain per lane -- per layer a global load of the layer's
// cross-sections (its latency is what k_rt spends most of a short grid's time on) and ~200 dependent fp64 instructions that no
// second wave shortens.  But the layer optical depths and the Planck values do not depend on the sweep, and the NS stream
// intensities of a sweep are independent recurrences of which only the weighted sum is needed per level.  One block = ONE 64-point
// tile, 2 NS waves:
//   phase 0  all waves together: Planck at the levels and the layer optical depths (dDepth!, 1e-6 floor) into LDS, level / layer i
//            by wave i mod 2 NS -- every wave's loads are in flight at once;
//   phase 1  wave (role, k) carries stream k of the downward / upward sweep and leaves W_k I_k in LDS every layer; after the layer's
//            barrier the role's wave 0 adds them in stream order (as k_rt does: same rounding), adds the stellar beam, reduces over
//            the wave and stores.  Chain per layer: one exp + layerplanck, ~40 instructions, + one barrier.
// With a non-zero albedo the upward sweep needs the surface downward flux: the two roles then run one after the other.
template <int NS>
__global__ __launch_bounds__(2 * NS * 64) void k_rt_streams(RtParams p, const double *__restrict__ nu, const double *__restrict__ wts,
                                                            int64_t nnu, const double *__restrict__ sigma, const double *__restrict__ muk,
                                                            const double *__restrict__ P, const double *__restrict__ Tlev,
                                                            const double *__restrict__ S_toa, const double *__restrict__ albedo,
                                                            double *__restrict__ tau, double *__restrict__ Mup, double *__restrict__ Mdn,
                                                            double *__restrict__ partial, size_t sig_bstride, const double *__restrict__ sigma2)
{
    extern __shared__ double sh[];   // Blev[np][64] | tl[nl][64] | xch[2 buffers][2 roles][NS][64] | red[2 np] | msurf[64]
    const int np = p.np, nl = np - 1, nlob = p.nlobatto;
    double *Blev = sh, *tl = sh + (size_t)np * 64, *xch = tl + (size_t)nl * 64, *red = xch + (size_t)2 * 2 * NS * 64, *msurf = red + 2 * np;
    {   // blockIdx.y = column of a batch (cs_column_batch)
        const size_t b = blockIdx.y;
        sigma += b * sig_bstride;
        muk += b * p.K;
        Tlev += b * p.np;
        if (tau) tau += b * (size_t)(p.np - 1) * nnu;
        partial += b * (size_t)gridDim.x * 2 * p.np;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool up = wave >= NS;
    const int k = up ? wave - NS : wave;
    const int64_t j = (int64_t)blockIdx.x * 64 + lane;
    const bool live = j < nnu;
    const int64_t jj = live ? j : nnu - 1;
    const double v = nu[jj];
    const double w = live ? wts[jj] : 0.0;
    const double fS = S_toa ? S_toa[jj] : 0.0;
    const double fa = albedo ? albedo[jj] : 0.0;
    const double c = p.cos_ts;
    for (int i = wave; i < np; i += 2 * NS) {
        Blev[(size_t)i * 64 + lane] = planck(v, Tlev[i]);
        if (i < nl) {   // optical depth of layer i exactly as k_rt forms it (dDepth!, discretized.jl:136-177): beta at the layer's nodes, 1e-6 floor
            const double dP = P[i + 1] - P[i];
            const int kl = i * (nlob - 1);
            auto sg = [&](size_t idx) { return sigma2 ? sigma[idx] + sigma2[idx] : sigma[idx]; };   // (near-line plane, see k_rt)
            double ti = (dP * p.ws[0]) * (p.C * (sg((size_t)kl * nnu + jj) / muk[kl]));
            for (int n = 1; n < nlob - 1; n++) ti += (dP * p.ws[n]) * (p.C * (sg((size_t)(kl + n) * nnu + jj) / muk[kl + n]));
            ti += (dP * p.ws[nlob - 1]) * (p.C * (sg((size_t)(kl + nlob - 1) * nnu + jj) / muk[kl + nlob - 1]));
            const double t = ti > 1e-6 ? ti : 1e-6;
            tl[(size_t)i * 64 + lane] = t;
            if (tau && live) tau[(size_t)i * nnu + j] = t;
        }
    }
    __syncthreads();
    const bool serial = albedo != nullptr;           // (block-uniform) the upward sweep waits for the surface downward flux
    const double mk = p.m[k], imk = p.im[k], Wk = p.W[k];
    auto slot = [&](int buf, int role, int kk) { return xch + (((size_t)buf * 2 + role) * NS + kk) * 64 + lane; };
    double I = 0.0, Ms = c * fS;
    // ---- downward sweep (the up-waves run their own sweep in the same loop when nothing ties them to this one)
    if (!up && k == 0) {   // level 0: M-[1] = c fS(nu), discretized.jl:299
        const double r = wave_sum(w * Ms);
        if (lane == 0) red[np + 0] = r;
        if (Mdn && live) Mdn[j] = Ms;
    }
    double Iu = 0.0;
    if (up && !serial) {   // surface: Planck emission only (no reflected part without an albedo), discretized.jl:309-310
        Iu = Blev[(size_t)(np - 1) * 64 + lane];
        if (k == 0) {
            const double Mu = Iu * kPi;
            const double r = wave_sum(w * Mu);
            if (lane == 0) red[np - 1] = r;
            if (Mup && live) Mup[(size_t)(np - 1) * nnu + j] = Mu;
        }
    }
    for (int s = 0; s < nl; s++) {
        const int buf = s & 1;
        if (!up) {
            const double t = tl[(size_t)s * 64 + lane];
            const double tr = exp_rt(-(t * mk));
            const double Be = layerplanck_inv(Blev[(size_t)s * 64 + lane], Blev[(size_t)(s + 1) * 64 + lane], (1.0 / t) * imk, tr);
            I = I * tr + Be;
            *slot(buf, 0, k) = Wk * I;
            if (k == 0 && S_toa) Ms *= exp(-t / c);
        } else if (!serial) {
            const int i = nl - 1 - s;
            const double t = tl[(size_t)i * 64 + lane];
            const double tr = exp_rt(-(t * mk));
            const double Be = layerplanck_inv(Blev[(size_t)(i + 1) * 64 + lane], Blev[(size_t)i * 64 + lane], (1.0 / t) * imk, tr);
            Iu = Iu * tr + Be;
            *slot(buf, 1, k) = Wk * Iu;
        }
        __syncthreads();
        if (k == 0 && (!up || !serial)) {   // the role's sum over the streams, in stream order
            const int role = up ? 1 : 0;
            double M = 0.0;
#pragma unroll
            for (int kk = 0; kk < NS; kk++) M += *slot(buf, role, kk);
            if (!up) {
                M += Ms;
                const double r = wave_sum(w * M);
                if (lane == 0) red[np + s + 1] = r;
                if (Mdn && live) Mdn[(size_t)(s + 1) * nnu + j] = M;
                if (serial && s == nl - 1) msurf[lane] = M;
            } else {
                const int i = nl - 1 - s;
                const double r = wave_sum(w * M);
                if (lane == 0) red[i] = r;
                if (Mup && live) Mup[(size_t)i * nnu + j] = M;
            }
        }
    }
    if (serial) {   // ---- upward sweep after the downward one: Lambertian reflection + Planck emission at the surface
        __syncthreads();
        if (up) {
            Iu = msurf[lane] * fa / kPi + Blev[(size_t)(np - 1) * 64 + lane];
            if (k == 0) {
                const double Mu = Iu * kPi;
                const double r = wave_sum(w * Mu);
                if (lane == 0) red[np - 1] = r;
                if (Mup && live) Mup[(size_t)(np - 1) * nnu + j] = Mu;
            }
        }
        for (int s = 0; s < nl; s++) {
            const int buf = s & 1, i = nl - 1 - s;
            if (up) {
                const double t = tl[(size_t)i * 64 + lane];
                const double tr = exp_rt(-(t * mk));
                const double Be = layerplanck_inv(Blev[(size_t)(i + 1) * 64 + lane], Blev[(size_t)i * 64 + lane], (1.0 / t) * imk, tr);
                Iu = Iu * tr + Be;
                *slot(buf, 1, k) = Wk * Iu;
            }
            __syncthreads();
            if (up && k == 0) {
                double M = 0.0;
#pragma unroll
                for (int kk = 0; kk < NS; kk++) M += *slot(buf, 1, kk);
                const double r = wave_sum(w * M);
                if (lane == 0) red[i] = r;
                if (Mup && live) Mup[(size_t)i * nnu + j] = M;
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * np; e += blockDim.x) partial[(size_t)blockIdx.x * 2 * np + e] = red[e];
}

// K4: F[e] = sum over blocks of partial[b][e], fixed order (bitwise reproducible run to run)
__global__ __launch_bounds__(256) void k_freduce(const double *__restrict__ partial, int nblk, int n2, double *__restrict__ F)
{
    __shared__ double sh[256];
    const int e = blockIdx.x;
    partial += (size_t)blockIdx.y * nblk * n2;   // column of a batch
    F += (size_t)blockIdx.y * n2;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(size_t)b * n2 + e];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) F[e] = sh[0];
}

// out[c*R + r] = in[r*C + c]   (nu-fastest [R][C] -> the reference's level-fastest [R, C] column-major)
__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ in, int R, int64_t Cn, double *__restrict__ out)
{
    __shared__ double t[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int yy = ty; yy < 32; yy += 8) {
        int r = r0 + yy;
        int64_t cc = c0 + tx;
        if (r < R && cc < Cn) t[yy][tx] = in[(size_t)r * Cn + cc];
    }
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8) {
        int64_t cc = c0 + yy;
        int r = r0 + tx;
        if (r < R && cc < Cn) out[(size_t)cc * R + r] = t[tx][yy];
    }
}

// sigma += sigma2 (the near-line plane folded in where the cross-sections themselves are the result: cs_column_sigma_run)
__global__ __launch_bounds__(256) void k_fold(int64_t n, double *__restrict__ sigma, const double *__restrict__ sigma2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sigma[i] += sigma2[i];
}

__global__ __launch_bounds__(256) void k_fill(int64_t n, double base, const double *__restrict__ extra, double *__restrict__ sigma)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sigma[i] = base + (extra ? extra[i] : 0.0);
}


// ---- opacity tables ("Mode T": the reference's baked Gas objects, gases.jl:68-145) --------------------------------------
// Z[m][nu], m = iT + nT*iP, holds sigma after cs_bake's line sums.  Reproduce the tail of bake + OpacityTable:
//   * a wavenumber whose states mix exact zeros with non-zeros is zeroed for every state (gases.jl:132-142);
//   * ln(sigma), or ln(floatmin) everywhere when no state exceeds floatmin (gases.jl:76-79).
__global__ __launch_bounds__(256) void k_table_log(double *__restrict__ Z, int M, int64_t nnu)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnu) return;
    const double tiny = 2.2250738585072014e-308;
    double mn = 1e300, mx = 0.0;
    for (int m = 0; m < M; m++) { const double v = Z[(size_t)m * nnu + i]; mn = fmin(mn, v); mx = fmax(mx, v); }
    const bool zero = (mn == 0.0 && mx > 0.0) || !(mx > tiny);
    const double lt = log(tiny);
    for (int m = 0; m < M; m++) { const size_t o = (size_t)m * nnu + i; Z[o] = zero ? lt : log(Z[o]); }
}

// sigma[k][nu] += conc[k] * exp( sum_m Z[m][nu] * W[m][k] ):  the Gas functor fC(T,P)*exp(Phi(T, ln P)) (gases.jl:85,278)
// with the 2-D Chebyshev interpolant written as a contraction against the Lagrange-basis weights W = a(T_k) (x) b(ln P_k).
// Block = 256 wavenumbers x 16 node states; the 16 weight columns are staged in LDS and read as broadcasts.
#define CS_TAB_KC 16

// The same on the matrix cores (the contraction over the nT*nP table nodes is the dense part of Mode T): D(16 states x 16 nu) +=
// A(16 states x 4 table nodes) * B(4 table nodes x 16 nu) with v_mfma_f64_16x16x4, operands as in k_cheb_apply_mfma.  One wave =
// one 64-point tile x NSUB*16 states, so the ln sigma table is read from HBM once per NSUB*16 states instead of once per 16
// (k_table_eval: 4 passes over 230 MB per gas at K = 61 -- HBM-bound at 0.42 ms per gas; this: profiles/r02_notes.md).
template <int NSUB>
__global__ __launch_bounds__(256, NSUB >= 4 ? 2 : 4) void k_table_eval_mfma(const double *__restrict__ Z, int M, int64_t nnu, int ntile,
                                                                            const double *__restrict__ W, int K, const double *__restrict__ conc,
                                                                            double *__restrict__ sigma)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nst = (K + 15) >> 4, nsg = (nst + NSUB - 1) / NSUB;
    const int tile = (int)(blockIdx.x / nsg) * 4 + wv;
    if (tile >= ntile) return;
    const int s0 = (int)(blockIdx.x % nsg) * NSUB;
    const int lr = lane & 15, lq = lane >> 4;
    v4f64 acc[NSUB][4];
#pragma unroll
    for (int si = 0; si < NSUB; si++)
#pragma unroll
        for (int jt = 0; jt < 4; jt++) acc[si][jt] = v4f64{0.0, 0.0, 0.0, 0.0};
    // B: Z[m0 + lq][nu]; the last tile may run past nnu: clamp the column (never stored)
    int64_t col[4];
#pragma unroll
    for (int jt = 0; jt < 4; jt++) { const int64_t i = (int64_t)tile * 64 + jt * 16 + lr; col[jt] = i < nnu ? i : nnu - 1; }
    int kcol[NSUB];
#pragma unroll
    for (int si = 0; si < NSUB; si++) { const int k = (s0 + si) * 16 + lr; kcol[si] = k < K ? k : -1; }
#pragma unroll 2
    for (int m = 0; m < M; m += 4) {
        const int mm = m + lq;
        const bool ok = mm < M;
        const double *__restrict__ zr = Z + (size_t)(ok ? mm : 0) * nnu;
        const double *__restrict__ wr = W + (size_t)(ok ? mm : 0) * K;
        double b[4], a[NSUB];
#pragma unroll
        for (int jt = 0; jt < 4; jt++) b[jt] = ok ? zr[col[jt]] : 0.0;
#pragma unroll
        for (int si = 0; si < NSUB; si++) a[si] = (ok && kcol[si] >= 0) ? wr[kcol[si]] : 0.0;
#pragma unroll
        for (int si = 0; si < NSUB; si++)
#pragma unroll
            for (int jt = 0; jt < 4; jt++) acc[si][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[si], b[jt], acc[si][jt], 0, 0, 0);
    }
#pragma unroll
    for (int si = 0; si < NSUB; si++)
#pragma unroll
        for (int jt = 0; jt < 4; jt++) {
            const int64_t i = (int64_t)tile * 64 + jt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = (s0 + si) * 16 + 4 * r + lq;
                if (k < K && i < nnu) sigma[(size_t)k * nnu + i] += conc[k] * exp(acc[si][jt][r]);
            }
        }
}

// ---- collision-induced absorption (collision_induced_absorption.jl:145-303) ----------------------------------------------
// One CIA object = a few bands; a band is ln k on a (nu, T) grid evaluated bilinearly (BilinearInterpolator of ln k with
// NoBoundaries, :207) or a single-temperature range evaluated linearly in nu (:188).  Per node state the host prepares the
// temperature cell (wave-uniform); each lane finds its own wavenumber cell once per band.
struct CiaBand {
    const double *nu;   // [nb] ascending
    const double *lnk;  // [nb][nt], nu fastest
    int nb, nt;
};
struct CiaState {       // per (band, node): temperature cell of the band's T grid, or "skip"
    int use, jT;        // use = 0: T outside the band and no extrapolation (or a single range while singles = false)
    double fT;          // fractional position in the T cell (0 for single ranges)
};
#define CS_MAX_CIA_BAND 24

// sigma[k][nu] += (ktot*Lo^2)*rho1*rho2/rhoa   (cia(k,T,Pa,P1,P2), :295-303)
__global__ __launch_bounds__(256) void k_cia(int nband, const CiaBand *__restrict__ bands, const CiaState *__restrict__ st,
                                              const double *__restrict__ nu, int64_t nnu, int K,
                                              const double *__restrict__ rho1, const double *__restrict__ rho2,
                                              const double *__restrict__ rhoa, double *__restrict__ sigma)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnu) return;
    const double v = nu[i];
    int cell[CS_MAX_CIA_BAND];
    double fx[CS_MAX_CIA_BAND];
    for (int b = 0; b < nband; b++) {
        const CiaBand B = bands[b];
        cell[b] = -1;
        fx[b] = 0.0;
        if (B.nu[0] <= v && v <= B.nu[B.nb - 1]) {  // Phi.G.xa <= nu <= Phi.G.xb, :255
            int lo = 0, hi = B.nb - 1;              // cell index c with nu[c] <= v < nu[c+1] (last cell when v == nu[nb-1])
            while (hi - lo > 1) { int m = (lo + hi) >> 1; if (B.nu[m] <= v) lo = m; else hi = m; }
            cell[b] = lo;
            fx[b] = (v - B.nu[lo]) / (B.nu[lo + 1] - B.nu[lo]);
        }
    }
    const double Lo2 = 7.21879268e38;  // constants.jl:18
    for (int k = 0; k < K; k++) {
        double ktot = 0.0;
        for (int b = 0; b < nband; b++) {
            const CiaState s = st[(size_t)b * K + k];
            if (cell[b] < 0 || !s.use) continue;
            const CiaBand B = bands[b];
            const int c = cell[b];
            const double x = fx[b];
            double lnk;
            if (B.nt == 1) {
                const double y0 = B.lnk[c], y1 = B.lnk[c + 1];
                lnk = (v - B.nu[c]) * (y1 - y0) / (B.nu[c + 1] - B.nu[c]) + y0;
            } else {
                const double *z0 = B.lnk + (size_t)s.jT * B.nb, *z1 = z0 + B.nb;
                const double y = s.fT;
                lnk = (1.0 - x) * (1.0 - y) * z0[c] + x * (1.0 - y) * z0[c + 1] + (1.0 - x) * y * z1[c] + x * y * z1[c + 1];
            }
            ktot += exp(lnk);
        }
        if (ktot != 0.0) sigma[(size_t)k * nnu + i] += (ktot * Lo2) * rho1[k] * rho2[k] / rhoa[k];
    }
}

// ---- K5: the flux kernel with the cross-sections finished on chip -------------------------------------------------------------
// fluxes.jl:270-277 does depth and flux of a wavenumber in one loop body.  Here the line kernels leave their sums in sigma (far,
// matrix-core pieces) and sigma2 (sub-tile cores, near-line pairs); what is still missing of Sigma(U, i, T_k, P_k) (absorbers.jl:84-95)
// -- the interpolated far wings sum_level C_l F (k_cheb_apply_mfma's product), the CIA pairs (k_cia) and the sum of the two planes
// (k_fold) -- used to be three read-modify-write passes over the [K][nnu] plane plus k_rt's read of both planes.  k_flux does all of
// it for ONE 64-point tile per block: phase A builds sigma_total[K][64] in LDS (matrix cores for the wings, every wave of the block),
// phase B is k_rt's / k_rt_streams' arithmetic, instruction for instruction, reading the cross-sections from LDS; the block partials
// of the nu-trapezoid are added by the last block to finish (a ticket), in k_freduce's order.  Nothing of sigma_total, of the
// optical depths or of the Planck values touches HBM unless the caller asks for tau.
struct CiaPairDev {
    int nband;
    const CiaBand *bands;      // [nband]
    const CiaState *st;        // [nband][K]
    const double *tab;         // ln k of band b interpolated to the temperature of state k: tab[toff[b] + k * nb + c]   (k_cia_tab)
    const int64_t *toff;       // [nband]
    const double *rho1, *rho2, *rhoa;   // [K]
    double *fac;               // [K] Lo^2 rho1 rho2 / rhoa of cia(k, T, Pa, P1, P2), :295-303 (k_cia_tab)
    // per grid (built once at cs_column_set_cia): the bands reaching each 64-point tile, [ntile][CS_CIA_ACT] (-1: none); the sample cell
    // of wavenumber i in the band of slot q, cell[q * nnu + i] (-1: outside the band), and its position in the cell, fx[q * nnu + i]
    const int32_t *tband, *cell;
    const double *fx;
    int nslot;
};
#define CS_CIA_ACT 4           // bands of one CIA object that may overlap one 64-point tile (checked on the host)
struct FluxFuse {
    int apply;                 // node sums to carry to the grid (A, Kpad valid)
    int Kpad;
    int ncia;
    ChebApply A;
    CiaPairDev cia[8];
    const double *sigma2;      // near-line plane (NULL: none)
    unsigned *ticket;          // NULL: k_freduce adds the block partials; else [1 + groups] counters, zero between launches
    double *gpartial;          // [groups][2 np] sums of the partials of 16 consecutive blocks
    double *F;                 // [2 np] band fluxes
    unsigned long long *dbg;   // NULL, or [8] time stamps (100 MHz wall clock) of the phases of block 0 (measurement hook)
};
__device__ __forceinline__ void flux_stamp(const FluxFuse &f, int slot)
{
    if (f.dbg && blockIdx.x == 0 && threadIdx.x == 0) f.dbg[slot] = wall_clock64();
}
// fine stamps of block 0's first thread behind the per-block ones (dbg[8 + 2 nblk + idx]): where inside a phase the time goes
__device__ __forceinline__ void flux_stamp_fine(const FluxFuse &f, int idx)
{
    if (f.dbg && blockIdx.x == 0 && threadIdx.x == 0) f.dbg[8 + 2 * gridDim.x + idx] = wall_clock64();
}
// ... and every block's first and last instruction behind them: dbg[8 + 2 b], dbg[9 + 2 b] (start skew and tail of the launch)
__device__ __forceinline__ void flux_stamp_block(const FluxFuse &f, int end)
{
    if (f.dbg && threadIdx.x == 0) f.dbg[8 + 2 * blockIdx.x + end] = wall_clock64();
}
#define CS_FLUX_GROUP 16

// tab[toff[b] + k*nb + c] = (1 - y) z0[c] + y z1[c]: the temperature half of the bilinear interpolation of ln k (BilinearInterpolator,
// collision_induced_absorption.jl:207,259) for state k -- wave-uniform in k_cia, so done once per (band, state, sample) instead of
// once per (wavenumber, state).  Single-temperature ranges (nt = 1) are copied.
__global__ __launch_bounds__(256) void k_cia_tab(CiaPairDev p, int K)
{
    const int b = blockIdx.z, k = blockIdx.y;
    const CiaBand B = p.bands[b];
    const int c = (int)(blockIdx.x * 256 + threadIdx.x);
    if (b == 0 && c == 0) p.fac[k] = 7.21879268e38 * p.rho1[k] * p.rho2[k] / p.rhoa[k];      // Lo^2 (constants.jl:18) x number densities
    if (c >= B.nb) return;
    const CiaState s = p.st[(size_t)b * K + k];
    double v;
    if (B.nt == 1) v = B.lnk[c];
    else if (!s.use) v = 0.0;
    else {
        const double *z0 = B.lnk + (size_t)s.jT * B.nb, *z1 = z0 + B.nb;
        v = (1.0 - s.fT) * z0[c] + s.fT * z1[c];
    }
    const_cast<double *>(p.tab)[p.toff[b] + (int64_t)k * B.nb + c] = v;
}

// the CIA terms of one object for the lane's wavenumber at the states k = k0, k0 + kstep, ... (n of them, n <= 16), added into
// dst[(k % R) * 64]: for every band that reaches the tile, exp(ln k) x Lo^2 rho1 rho2 / rhoa (cia(k, T, Pa, P1, P2), :295-303) with ln k
// the linear interpolation in nu of the band's samples at the state's temperature (tab).  Written for memory-level parallelism: what
// depends on the band alone is loaded once, the two samples of eight states are requested together, nothing in the state loop waits
// for a load that depends on another load.  (Where two bands overlap their terms are added one after the other.)
__device__ __forceinline__ void cia_add(const CiaPairDev &p, int tile, int64_t ii, int64_t nnu, double v, int K, int k0, int kstep, int n,
                                        double *__restrict__ dst, int R)
{
    const int4 bands4 = *reinterpret_cast<const int4 *>(p.tband + (size_t)tile * CS_CIA_ACT);      // wave-uniform
    const int bq[CS_CIA_ACT] = {bands4.x, bands4.y, bands4.z, bands4.w};
#pragma unroll
    for (int q = 0; q < CS_CIA_ACT; q++) {
        const int b = bq[q];
        if (q >= p.nslot || b < 0) continue;                      // (wave-uniform)
        const int cc = p.cell[(size_t)q * nnu + ii];              // -1: the lane's wavenumber is outside the band
        const double x = p.fx[(size_t)q * nnu + ii];
        const CiaBand B = p.bands[b];
        const double *__restrict__ tb = p.tab + p.toff[b] + max(cc, 0);
        const CiaState *__restrict__ stb = p.st + (size_t)b * K;
        for (int h0 = 0; h0 < n; h0 += 8) {
            double t0[8], t1[8];
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int k = min(k0 + (h0 + s) * kstep, K - 1);
                t0[s] = tb[(int64_t)k * B.nb];
                t1[s] = tb[(int64_t)k * B.nb + 1];
            }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int k = k0 + (h0 + s) * kstep;
                if (h0 + s >= n || k >= K) break;
                if (!stb[k].use) continue;                        // (wave-uniform)
                double e;
                if (B.nt == 1) {     // single-temperature range (`singles`): samples may be ln 0 = -inf, the library exponential keeps the reference's inf / NaN rules
                    const int c1 = max(cc, 0);
                    e = exp((v - B.nu[c1]) * (t1[s] - t0[s]) / (B.nu[c1 + 1] - B.nu[c1]) + t0[s]);
                } else {             // finite samples (floatmin-clamped, :205): the flux kernel's own exponential (<= 2 ulp, tests/test_gpu_kat.py)
                    e = exp_rt((1.0 - x) * t0[s] + x * t1[s]);
                }
                if (cc >= 0) dst[(size_t)(k % R) * 64] += e * p.fac[k];
            }
        }
    }
}

// phase A: sig[k][lane] = Sigma(U, i, T_k, P_k) of the block's tile for every node state k, in the order the separate kernels add
// it up: ((sigma + interpolated wings) + CIA pairs, one after the other) + sigma2
__device__ __forceinline__ void flux_sigma_tile(const FluxFuse &f, int K, int64_t nnu, int tile, const double *__restrict__ sigma,
                                                const double *__restrict__ nu, double *__restrict__ sig, int wave, int nwaves, int lane)
{
    const int64_t i = (int64_t)tile * 64 + lane;
    const int64_t ii = i < nnu ? i : nnu - 1;      // lanes past a ragged end repeat the last column (their trapezoid weight is 0)
    const bool fold2 = f.sigma2 != nullptr && f.ncia == 0;      // the near-line plane goes in with the store (no CIA terms come between)
    if (f.apply) {
        // items = (16-state group, half of the tile's points): D(16 states x 16 points) += A(16 states x 4 nodes: F) B(4 nodes x 16
        // points: C), operands as in k_cheb_apply_mfma; an item is one wave's, so no sums cross waves.  A short grid is a chain of
        // latencies: all operands of a level are requested before its first matrix instruction, and the item's cross-sections before that
        const int nst = (K + 15) >> 4;
        const int lr = lane & 15, lq = lane >> 4;
        for (int item = wave; item < 2 * nst; item += nwaves) {
            const int sgrp = item >> 1, h = item & 1;
            double s1[2][4], s2[2][4];
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                const int64_t ic = (int64_t)tile * 64 + h * 32 + jt * 16 + lr;
                const int64_t icc = ic < nnu ? ic : nnu - 1;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = min(sgrp * 16 + 4 * r + lq, K - 1);
                    s1[jt][r] = sigma[(size_t)k * nnu + icc];
                    s2[jt][r] = fold2 ? f.sigma2[(size_t)k * nnu + icc] : 0.0;
                }
            }
            v4f64 acc[2] = {v4f64{0.0, 0.0, 0.0, 0.0}, v4f64{0.0, 0.0, 0.0, 0.0}};
            for (int g = 0; g < f.A.ngas; g++) {
                const double *__restrict__ Fg = f.A.F[g];
                for (int l = f.A.l0[g]; l < f.A.nlev; l++) {
                    const int sh = f.A.shift[l];
                    const int T = tile >> sh, sub = tile & ((1 << sh) - 1);
                    const size_t itv = (size_t)64 << sh;
                    const double *__restrict__ Cp = f.A.Cm[l] + ((size_t)T * CS_NC + lq) * itv + (size_t)sub * 64 + (size_t)h * 32 + lr;
                    const double *__restrict__ Fp = Fg + ((size_t)f.A.noff[l] + (size_t)T * CS_NC + lq) * f.Kpad + (size_t)sgrp * 16 + lr;
                    double a[CS_NC / 4], b0[CS_NC / 4], b1[CS_NC / 4];
#pragma unroll
                    for (int st = 0; st < CS_NC / 4; st++) {
                        a[st] = Fp[(size_t)(4 * st) * f.Kpad];
                        b0[st] = Cp[(size_t)(4 * st) * itv];
                        b1[st] = Cp[(size_t)(4 * st) * itv + 16];
                    }
#pragma unroll
                    for (int st = 0; st < CS_NC / 4; st++) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st], b0[st], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st], b1[st], acc[1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                const int col = h * 32 + jt * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = sgrp * 16 + 4 * r + lq;
                    if (k < K) sig[(size_t)k * 64 + col] = fold2 ? (s1[jt][r] + acc[jt][r]) + s2[jt][r] : s1[jt][r] + acc[jt][r];
                }
            }
        }
    } else {
        for (int k = wave; k < K; k += nwaves)
            sig[(size_t)k * 64 + lane] = fold2 ? sigma[(size_t)k * nnu + ii] + f.sigma2[(size_t)k * nnu + ii] : sigma[(size_t)k * nnu + ii];
    }
    if (f.ncia == 0) return;
    __syncthreads();
    const double v = nu[ii];
    for (int pq = 0; pq < f.ncia; pq++)
        for (int k0 = wave; k0 < K; k0 += 16 * nwaves)        // this wave's states k0, k0 + nwaves, ..., sixteen at a time
            cia_add(f.cia[pq], tile, ii, nnu, v, K, k0, nwaves, 16, sig + lane, K);
    if (f.sigma2)
        for (int k = wave; k < K; k += nwaves) sig[(size_t)k * 64 + lane] += f.sigma2[(size_t)k * nnu + ii];
}

// the band fluxes from the block partials without another launch (f.ticket != NULL: grids of up to a few hundred blocks).  Blocks form
// groups of CS_FLUX_GROUP consecutive ones: the last block of a group to arrive adds the group's partials in block order into
// gpartial[group]; the last GROUP to finish adds the group sums in group order into F.  Who arrives last varies, what is added in
// which order does not: the band fluxes are bitwise repeatable like k_freduce's.  Every thread's loads of a stage are independent
// (requested together), so the tail of the kernel is two memory latencies, not one per block.
// (The partials cross XCDs, whose L2s are not coherent with each other.  A device-scope fence per block would write back and
//  invalidate the whole L2 of the block's XCD -- measured: +40 us on a 157-block grid -- so the partials and group sums travel as
//  device-scope relaxed atomic stores and loads instead (write-through / L2-bypassing accesses to just those words), ordered by
//  waiting for the stores before the ticket and by the ticket's own device-scope atomicity.)
// The band-flux partials of k_flux_* cross from the blocks that form them to the last block(s) to finish, which add them -- in other
// workgroups, usually on other XCDs (eight L2s, not coherent with each other).  The textbook hand-over is release / acquire at agent
// scope around the ticket; on this part an agent-scope release is a write-back of the XCD's whole L2 (measured: +40 us on a 157-block
// grid, profiles/r04_notes.md), for a few hundred doubles.  What is used instead, and what it rests on:
//   * the partials are written and read with AGENT-scope relaxed ATOMIC stores / loads.  On gfx942 / gfx950 these are global_store /
//     global_load with sc1 set: the store writes through the XCD's L2 to memory, the load misses L2 on purpose -- each access is
//     coherent at the agent by itself, no cache maintenance needed (AMDGPU backend memory model, "gfx942" column: atomic store /
//     load monotonic, agent scope);
//   * s_waitcnt vmcnt(0) after the stores: a write-through store is counted in vmcnt until it has been acknowledged by the memory
//     side, so when the wait returns this thread's partials are visible at the agent; the workgroup barrier behind it extends that to
//     every thread of the block BEFORE thread 0 takes the ticket.  The ticket is an agent-scope atomic RMW (performed at memory on
//     this part): a block that draws the last number therefore finds every other block's partials already there;
//   * the compiler must not move the partial stores below, or the partial loads above, the ticket: the workgroup-scope release /
//     acquire fences around it say so in the language of the memory model (no L2 action at that scope), beside the barriers.
// Formally the ticket would have to be release / acquire at AGENT scope for the hand-over to be a happens-before edge of the HIP
// memory model; the relaxed form is correct for the hardware named in the #error above, which is why the host hands out a ticket only
// on a device that reports gfx950 (cs_create; otherwise k_freduce, a second launch, adds the partials) -- tests/test_gpu_flux_fused.py
// compares the two bit for bit.
__device__ __forceinline__ void flux_store_dev(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double flux_load_dev(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flux_last_block_reduce(const FluxFuse &f, const double *__restrict__ partial, int nblk, int n2)
{
    __shared__ unsigned last;
    const int grp = (int)blockIdx.x / CS_FLUX_GROUP, ngrp = (nblk + CS_FLUX_GROUP - 1) / CS_FLUX_GROUP;
    const int b0 = grp * CS_FLUX_GROUP, nb = min(CS_FLUX_GROUP, nblk - b0);
    __builtin_amdgcn_s_waitcnt(0);      // this thread's partials (device-scope stores) have left before the block takes its ticket
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(f.ticket + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nb - 1u) ? 1u : 0u;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!last) return;
    for (int e = threadIdx.x; e < n2; e += blockDim.x) {
        double v[CS_FLUX_GROUP];
#pragma unroll
        for (int q = 0; q < CS_FLUX_GROUP; q++) v[q] = q < nb ? flux_load_dev(&partial[(size_t)(b0 + q) * n2 + e]) : 0.0;
        double t = v[0];
#pragma unroll
        for (int q = 1; q < CS_FLUX_GROUP; q++) t += v[q];       // (absent blocks add +0.0: no change)
        flux_store_dev(&f.gpartial[(size_t)grp * n2 + e], t);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(f.ticket + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch (stream order)
        last = (__hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ngrp - 1u) ? 1u : 0u;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!last) return;
    for (int e = threadIdx.x; e < n2; e += blockDim.x) {
        double t = 0.0;
        for (int g0 = 0; g0 < ngrp; g0 += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = g0 + q < ngrp ? flux_load_dev(&f.gpartial[(size_t)(g0 + q) * n2 + e]) : 0.0;
#pragma unroll
            for (int q = 0; q < 8; q++) t += v[q];
        }
        f.F[e] = t;
    }
    if (threadIdx.x == 0) __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f.dbg && threadIdx.x == 0) f.dbg[7] = wall_clock64();   // (measurement hook: the last block's last word)
}

// K5 on short grids, second form: the sweeps as a SCAN over layer chunks.  In k_flux_streams a wave walks all layers of one stream: 60
// dependent steps of one exponential each, one barrier per layer -- 45 us that no amount of idle chip shortens.  But only the recurrence
// I <- I e^(-tau m) + B_eff is sequential, and it is linear: a run of layers acts on the incoming intensity as I -> A I + B.  One block =
// one tile, NW waves, wave w = a chunk of consecutive layers, all NS streams, both sweeps:
//   phase A, 0  as k_flux_streams: cross-sections, then optical depths and Planck values of every layer into LDS;
//   phase 1     every wave runs its chunk from zero incoming intensity -- NS independent chains per lane, no barrier -- and keeps the
//               chunk's (A, B) per stream and sweep (and the stellar beam's attenuation) in registers;
//   phase 2     the incoming intensities travel from chunk to chunk: NW steps of one fused multiply-add per stream, downward sweep in
//               ascending chunk order, upward in descending (after the surface term, which needs the downward flux only with an albedo);
//   phase 3     every wave runs its chunk again from its true incoming intensity and forms what radiate! returns at its levels.
// Inside a chunk the operations and their order are discretized.jl:282-322's; across chunk boundaries the incoming intensity was
// formed as A I + B instead of layer by layer: the same numbers to a few units in the last place (tests: 1e-13 against k_rt_streams).
template <int NS, int PER>
__global__ __launch_bounds__(768) void k_flux_scan(RtParams p, const double *__restrict__ nu, const double *__restrict__ wts, int64_t nnu,
                                                    const double *__restrict__ sigma, const double *__restrict__ muk, const double *__restrict__ P,
                                                    const double *__restrict__ Tlev, const double *__restrict__ S_toa,
                                                    const double *__restrict__ albedo, double *__restrict__ tau, double *__restrict__ Mup,
                                                    double *__restrict__ Mdn, double *__restrict__ partial, FluxFuse f)
{
    extern __shared__ double sh[];   // sig[K][64] | Blev[np][64] | tl[nl][64] | xin[2 sweeps][NS + 1][64] | red[2 np]
    const int np = p.np, nl = np - 1, nlob = p.nlobatto, K = p.K;
    double *sig = sh, *Blev = sig + (size_t)K * 64, *tl = Blev + (size_t)np * 64, *xin = tl + (size_t)nl * 64, *red = xin + (size_t)2 * (NS + 1) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = (int)(blockDim.x >> 6);
    const int64_t j = (int64_t)blockIdx.x * 64 + lane;
    const bool live = j < nnu;
    const int64_t jj = live ? j : nnu - 1;
    flux_stamp(f, 0);
    flux_stamp_block(f, 0);
    flux_sigma_tile(f, K, nnu, (int)blockIdx.x, sigma, nu, sig, wave, NW, lane);
    __syncthreads();
    flux_stamp(f, 1);
    const double v = nu[jj];
    const double w = live ? wts[jj] : 0.0;
    const double fS = S_toa ? S_toa[jj] : 0.0;
    const double fa = albedo ? albedo[jj] : 0.0;
    const double c = p.cos_ts;
    for (int i = wave; i < np; i += NW) {
        Blev[(size_t)i * 64 + lane] = planck(v, Tlev[i]);
        if (i < nl) {   // optical depth of layer i exactly as k_rt forms it (dDepth!, discretized.jl:136-177): beta at the layer's nodes, 1e-6 floor
            const double dP = P[i + 1] - P[i];
            const int kl = i * (nlob - 1);
            auto sg = [&](int kk) { return sig[(size_t)kk * 64 + lane]; };
            double ti = (dP * p.ws[0]) * (p.C * (sg(kl) / muk[kl]));
            for (int n = 1; n < nlob - 1; n++) ti += (dP * p.ws[n]) * (p.C * (sg(kl + n) / muk[kl + n]));
            ti += (dP * p.ws[nlob - 1]) * (p.C * (sg(kl + nlob - 1) / muk[kl + nlob - 1]));
            const double t = ti > 1e-6 ? ti : 1e-6;
            tl[(size_t)i * 64 + lane] = t;
            if (tau && live) tau[(size_t)i * nnu + j] = t;
        }
    }
    __syncthreads();
    flux_stamp(f, 2);
    // this wave's layers [l0, l1)
    const int per = (nl + NW - 1) / NW;
    const int l0 = min(wave * per, nl), l1 = min(l0 + per, nl);
    // ---- phase 1: the chunk from zero incoming intensity.  PER > 0 (a chunk has at most PER layers): the transmissivities of the chunk
    // -- one exponential per (layer, stream) -- stay in registers for the upward sweep and for phase 3, which would form the same
    // numbers again (PER = 0: they do; same results either way)
    double Ad[NS], Bd[NS], Au[NS], Bu[NS], As = 1.0;
    constexpr int PQ = PER > 0 ? PER : 1;
    double trv[PQ][NS], itv[PQ], trs[PQ];
#pragma unroll
    for (int k = 0; k < NS; k++) { Ad[k] = 1.0; Bd[k] = 0.0; Au[k] = 1.0; Bu[k] = 0.0; }
    if constexpr (PER > 0) {
#pragma unroll
        for (int q = 0; q < PER; q++) {      // downward: layers ascending
            const int i = l0 + q;
            itv[q] = 0.0; trs[q] = 1.0;
#pragma unroll
            for (int k = 0; k < NS; k++) trv[q][k] = 1.0;
            if (i < l1) {                    // (wave-uniform)
                const double t = tl[(size_t)i * 64 + lane], it = 1.0 / t;
                const double B1 = Blev[(size_t)i * 64 + lane], B2 = Blev[(size_t)(i + 1) * 64 + lane];
                itv[q] = it;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const double tr = exp_rt(-(t * p.m[k]));
                    const double Be = layerplanck_inv(B1, B2, it * p.im[k], tr);
                    trv[q][k] = tr;
                    Bd[k] = Bd[k] * tr + Be;
                    Ad[k] *= tr;
                }
                if (S_toa) { trs[q] = exp(-t / c); As *= trs[q]; }
            }
        }
#pragma unroll
        for (int q = PER - 1; q >= 0; q--) { // upward: layers descending
            const int i = l0 + q;
            if (i < l1) {
                const double it = itv[q];
                const double Bhi = Blev[(size_t)(i + 1) * 64 + lane], Blo = Blev[(size_t)i * 64 + lane];
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const double tr = trv[q][k];
                    const double Be = layerplanck_inv(Bhi, Blo, it * p.im[k], tr);
                    Bu[k] = Bu[k] * tr + Be;
                    Au[k] *= tr;
                }
            }
        }
    } else {
    for (int i = l0; i < l1; i++) {          // downward: layers ascending
        const double t = tl[(size_t)i * 64 + lane], it = 1.0 / t;
        const double B1 = Blev[(size_t)i * 64 + lane], B2 = Blev[(size_t)(i + 1) * 64 + lane];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const double tr = exp_rt(-(t * p.m[k]));
            const double Be = layerplanck_inv(B1, B2, it * p.im[k], tr);
            Bd[k] = Bd[k] * tr + Be;
            Ad[k] *= tr;
        }
        if (S_toa) As *= exp(-t / c);
    }
    for (int i = l1 - 1; i >= l0; i--) {     // upward: layers descending
        const double t = tl[(size_t)i * 64 + lane], it = 1.0 / t;
        const double Bhi = Blev[(size_t)(i + 1) * 64 + lane], Blo = Blev[(size_t)i * 64 + lane];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const double tr = exp_rt(-(t * p.m[k]));
            const double Be = layerplanck_inv(Bhi, Blo, it * p.im[k], tr);
            Bu[k] = Bu[k] * tr + Be;
            Au[k] *= tr;
        }
    }
    }
    flux_stamp(f, 3);
    // ---- phase 2: incoming intensities from chunk to chunk.  xin[0][k] = what enters the current chunk going down, xin[1][k] going up;
    // xin[0][NS] = the stellar beam entering the chunk
    auto X = [&](int sweep, int k) { return xin + ((size_t)sweep * (NS + 1) + k) * 64 + lane; };
    double Id[NS], Iu[NS], Ms_in = 0.0;
    const bool serial = albedo != nullptr;   // (block-uniform) the surface term of the upward sweep needs the downward flux only with an albedo
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) *X(0, k) = 0.0;
        *X(0, NS) = c * fS;                  // M-[1] = c fS(nu), discretized.jl:299
    }
    auto surface = [&](double Md) {          // Lambertian reflection + Planck emission, discretized.jl:309-310
        const double Is = Md * fa / kPi + Blev[(size_t)(np - 1) * 64 + lane];
#pragma unroll
        for (int k = 0; k < NS; k++) *X(1, k) = Is;
        const double Mu = Is * kPi;
        const double r = wave_sum(w * Mu);
        if (lane == 0) red[np - 1] = r;
        if (Mup && live) Mup[(size_t)(np - 1) * nnu + j] = Mu;
    };
    if (!serial && wave == NW - 1) surface(0.0);   // (no reflected part: both sweeps travel in the same steps below)
    for (int s = 0; s < NW; s++) {
        __syncthreads();
        flux_stamp_fine(f, s);
        if (wave == s) {
#pragma unroll
            for (int k = 0; k < NS; k++) { Id[k] = *X(0, k); *X(0, k) = Id[k] * Ad[k] + Bd[k]; }
            Ms_in = *X(0, NS);
            *X(0, NS) = Ms_in * As;
        }
        if (!serial && wave == NW - 1 - s) {
#pragma unroll
            for (int k = 0; k < NS; k++) { Iu[k] = *X(1, k); *X(1, k) = Iu[k] * Au[k] + Bu[k]; }
        }
    }
    if (serial) {
        __syncthreads();
        if (wave == NW - 1) {   // the downward flux at the surface is in xin[0]
            double Md = 0.0;
#pragma unroll
            for (int k = 0; k < NS; k++) Md += p.W[k] * *X(0, k);
            Md += *X(0, NS);
            surface(Md);
        }
        for (int s = NW - 1; s >= 0; s--) {
            __syncthreads();
            if (wave == s) {
#pragma unroll
                for (int k = 0; k < NS; k++) { Iu[k] = *X(1, k); *X(1, k) = Iu[k] * Au[k] + Bu[k]; }
            }
        }
    }
    flux_stamp(f, 4);
    // ---- phase 3: the chunk from its true incoming intensities; what radiate! returns at its levels
    if (wave == 0) {   // level 0 of the downward flux
        const double r = wave_sum(w * Ms_in);
        if (lane == 0) red[np + 0] = r;
        if (Mdn && live) Mdn[j] = Ms_in;
    }
    double Ms = Ms_in;
    if constexpr (PER > 0) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int i = l0 + q;
            if (i < l1) {
                const double it = itv[q];
                const double B1 = Blev[(size_t)i * 64 + lane], B2 = Blev[(size_t)(i + 1) * 64 + lane];
                double Md = 0.0;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const double tr = trv[q][k];
                    const double Be = layerplanck_inv(B1, B2, it * p.im[k], tr);
                    Id[k] = Id[k] * tr + Be;
                    Md += p.W[k] * Id[k];
                }
                if (S_toa) Ms *= trs[q];
                Md += Ms;
                const double r = wave_sum(w * Md);
                if (lane == 0) red[np + i + 1] = r;
                if (Mdn && live) Mdn[(size_t)(i + 1) * nnu + j] = Md;
            }
            flux_stamp_fine(f, 16 + q);
        }
#pragma unroll
        for (int q = PER - 1; q >= 0; q--) {
            const int i = l0 + q;
            if (i < l1) {
                const double it = itv[q];
                const double Bhi = Blev[(size_t)(i + 1) * 64 + lane], Blo = Blev[(size_t)i * 64 + lane];
                double Mu = 0.0;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const double tr = trv[q][k];
                    const double Be = layerplanck_inv(Bhi, Blo, it * p.im[k], tr);
                    Iu[k] = Iu[k] * tr + Be;
                    Mu += p.W[k] * Iu[k];
                }
                const double r = wave_sum(w * Mu);
                if (lane == 0) red[i] = r;
                if (Mup && live) Mup[(size_t)i * nnu + j] = Mu;
            }
            flux_stamp_fine(f, 24 + q);
        }
    } else {
    for (int i = l0; i < l1; i++) {
        const double t = tl[(size_t)i * 64 + lane], it = 1.0 / t;
        const double B1 = Blev[(size_t)i * 64 + lane], B2 = Blev[(size_t)(i + 1) * 64 + lane];
        double Md = 0.0;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const double tr = exp_rt(-(t * p.m[k]));
            const double Be = layerplanck_inv(B1, B2, it * p.im[k], tr);
            Id[k] = Id[k] * tr + Be;
            Md += p.W[k] * Id[k];
        }
        if (S_toa) Ms *= exp(-t / c);
        Md += Ms;
        const double r = wave_sum(w * Md);
        if (lane == 0) red[np + i + 1] = r;
        if (Mdn && live) Mdn[(size_t)(i + 1) * nnu + j] = Md;
    }
    for (int i = l1 - 1; i >= l0; i--) {
        const double t = tl[(size_t)i * 64 + lane], it = 1.0 / t;
        const double Bhi = Blev[(size_t)(i + 1) * 64 + lane], Blo = Blev[(size_t)i * 64 + lane];
        double Mu = 0.0;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const double tr = exp_rt(-(t * p.m[k]));
            const double Be = layerplanck_inv(Bhi, Blo, it * p.im[k], tr);
            Iu[k] = Iu[k] * tr + Be;
            Mu += p.W[k] * Iu[k];
        }
        const double r = wave_sum(w * Mu);
        if (lane == 0) red[i] = r;
        if (Mup && live) Mup[(size_t)i * nnu + j] = Mu;
    }
    }
    __syncthreads();
    flux_stamp(f, 5);
    for (int e = threadIdx.x; e < 2 * np; e += blockDim.x) {
        if (f.ticket) flux_store_dev(&partial[(size_t)blockIdx.x * 2 * np + e], red[e]);
        else partial[(size_t)blockIdx.x * 2 * np + e] = red[e];
    }
    flux_stamp_block(f, 1);
    if (f.ticket) flux_last_block_reduce(f, partial, (int)gridDim.x, 2 * np);
    flux_stamp(f, 6);
}

// K5 on long grids (thousands of tiles): one WAVE = one 64-point tile, both sweeps (k_rt<NS, false>'s arithmetic), four tiles per
// block.  The cross-sections are finished 16 node states at a time -- one matrix-core state group -- into a ring of 16 + nlobatto - 1
// rows of LDS per wave, just ahead of the downward sweep that consumes them: the interpolation product of the chunk (4 matrix
// instructions per 4-node step: one F operand, four C operands), sigma and sigma2 of the chunk, its CIA terms.  The layer optical
// depths go to `tau` (the caller's output, or scratch) for the upward sweep, as in k_rt.  LDS: 8.7 KB per wave with nlobatto = 2.
template <int NS>
__device__ __forceinline__ void flux_chunk_body(const RtParams &p, const double *__restrict__ nu, const double *__restrict__ wts, int64_t nnu,
                                                const double *__restrict__ sigma, const double *__restrict__ muk, const double *__restrict__ P,
                                                const double *__restrict__ Tlev, const double *__restrict__ S_toa,
                                                const double *__restrict__ albedo, double *__restrict__ tau, double *__restrict__ Mup,
                                                double *__restrict__ Mdn, double *__restrict__ partial, const FluxFuse &f, int ntile)
{
    extern __shared__ double sh[];   // red[2 np][nw] | ring[nw][R][64]
    const int nw = blockDim.x >> 6;
    const int np = p.np, nl = np - 1, nlob = p.nlobatto, K = p.K;
    const int R = 16 + nlob - 1;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *red = sh, *ring = sh + (size_t)2 * np * nw + (size_t)wv * R * 64;
    const int tile = (int)blockIdx.x * nw + wv;
    const bool wave_live = tile < ntile;
    const int tl_ = wave_live ? tile : ntile - 1;     // (a wave past the last tile repeats it with zero weights: barriers stay uniform)
    const int64_t j = (int64_t)tl_ * 64 + lane;
    const bool live = wave_live && j < nnu;
    const int64_t jj = j < nnu ? j : nnu - 1;
    const double v = nu[jj];
    const double w = live ? wts[jj] : 0.0;
    const double fS = S_toa ? S_toa[jj] : 0.0;
    const double fa = albedo ? albedo[jj] : 0.0;
    const double c = p.cos_ts;
    const int lr = lane & 15, lq = lane >> 4;
    int chunk_next = 0;                // first node state not yet in the ring
    auto load_chunk = [&]() {          // node states [chunk_next, chunk_next + 16) -> ring rows k % R
        const int k0 = chunk_next, sgrp = k0 >> 4;
        if (f.apply) {
            v4f64 acc[4];
#pragma unroll
            for (int jt = 0; jt < 4; jt++) acc[jt] = v4f64{0.0, 0.0, 0.0, 0.0};
            for (int g = 0; g < f.A.ngas; g++) {
                const double *__restrict__ Fg = f.A.F[g];
                for (int l = f.A.l0[g]; l < f.A.nlev; l++) {
                    const int sh_ = f.A.shift[l];
                    const int T = tl_ >> sh_, sub = tl_ & ((1 << sh_) - 1);
                    const size_t itv = (size_t)64 << sh_;
                    const double *__restrict__ Cp = f.A.Cm[l] + ((size_t)T * CS_NC + lq) * itv + (size_t)sub * 64 + lr;
                    const double *__restrict__ Fp = Fg + ((size_t)f.A.noff[l] + (size_t)T * CS_NC + lq) * f.Kpad + (size_t)sgrp * 16 + lr;
#pragma unroll 4
                    for (int m = 0; m < CS_NC; m += 4) {
                        const double a = Fp[(size_t)m * f.Kpad];
                        double b[4];
#pragma unroll
                        for (int jt = 0; jt < 4; jt++) b[jt] = Cp[(size_t)m * itv + jt * 16];
#pragma unroll
                        for (int jt = 0; jt < 4; jt++) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[jt], acc[jt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < 4; jt++) {
                const int col = jt * 16 + lr;
                const int64_t ic = (int64_t)tl_ * 64 + col;
                const int64_t icc = ic < nnu ? ic : nnu - 1;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = k0 + 4 * r + lq;
                    if (k < K) ring[(size_t)(k % R) * 64 + col] = sigma[(size_t)k * nnu + icc] + acc[jt][r];
                }
            }
        } else {
            for (int k = k0; k < min(k0 + 16, K); k++) ring[(size_t)(k % R) * 64 + lane] = sigma[(size_t)k * nnu + jj];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int pq = 0; pq < f.ncia; pq++) cia_add(f.cia[pq], tl_, jj, nnu, v, K, k0, 1, 16, ring + lane, R);
        if (f.sigma2)
            for (int k = k0; k < min(k0 + 16, K); k++) ring[(size_t)(k % R) * 64 + lane] += f.sigma2[(size_t)k * nnu + jj];
        chunk_next = k0 + 16;
    };
    auto sg = [&](int k) { return ring[(size_t)(k % R) * 64 + lane]; };

    double I[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) I[k] = 0.0;
    load_chunk();
    double b1 = p.C * (sg(0) / muk[0]);  // beta at node 0, discretized.jl:150
    double Ms = c * fS;                   // M-[1] = c*fS(nu), discretized.jl:299
    double Md = Ms;
    double Bprev = planck(v, Tlev[0]);
    {
        const double r = wave_sum(w * Md);
        if (lane == 0) red[(np + 0) * nw + wv] = r;
        if (Mdn && live) Mdn[j] = Md;
    }
    for (int i = 0; i < nl; i++) {
        const int ke = (i + 1) * (nlob - 1);
        while (ke >= chunk_next) load_chunk();      // (wave-uniform)
        const double dP = P[i + 1] - P[i];
        double ti = (dP * p.ws[0]) * b1;
        for (int n = 1; n < nlob - 1; n++) {
            const int k = i * (nlob - 1) + n;
            ti += (dP * p.ws[n]) * (p.C * (sg(k) / muk[k]));
        }
        const double bn = p.C * (sg(ke) / muk[ke]);
        ti += (dP * p.ws[nlob - 1]) * bn;
        b1 = bn;
        const double t = ti > 1e-6 ? ti : 1e-6;  // floor, discretized.jl:147,174
        if (live) tau[(size_t)i * nnu + j] = t;
        const double Bnext = planck(v, Tlev[i + 1]);
        Md = 0.0;
        const double it = 1.0 / t;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const double tk = t * p.m[k];
            const double tr = exp_rt(-tk);
            const double Be = layerplanck_inv(Bprev, Bnext, it * p.im[k], tr);
            I[k] = I[k] * tr + Be;
            Md += p.W[k] * I[k];
        }
        if (S_toa) Ms *= exp(-t / c);   // (wave-uniform; without a stellar beam Ms stays 0)
        Md += Ms;
        Bprev = Bnext;
        const double r = wave_sum(w * Md);
        if (lane == 0) red[(np + i + 1) * nw + wv] = r;
        if (Mdn && live) Mdn[(size_t)(i + 1) * nnu + j] = Md;
    }
    {
        // surface: Lambertian reflection + Planck emission, discretized.jl:309-310
        const double Is = Md * fa / kPi + Bprev;
        double Mu = Is * kPi;
        {
            const double r = wave_sum(w * Mu);
            if (lane == 0) red[(np - 1) * nw + wv] = r;
            if (Mup && live) Mup[(size_t)(np - 1) * nnu + j] = Mu;
        }
#pragma unroll
        for (int k = 0; k < NS; k++) I[k] = Is;
        double Bhi = Bprev;  // B at level i+1
        double t_next = live ? tau[(size_t)(nl - 1) * nnu + j] : 1.0;      // (this lane's own store, same address: program order)
        for (int i = nl - 1; i >= 0; i--) {
            const double t = t_next;
            if (i > 0) t_next = live ? tau[(size_t)(i - 1) * nnu + j] : 1.0;   // one layer ahead
            const double Blo = planck(v, Tlev[i]);
            Mu = 0.0;
            const double it = 1.0 / t;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const double tk = t * p.m[k];
                const double tr = exp_rt(-tk);
                const double Be = layerplanck_inv(Bhi, Blo, it * p.im[k], tr);
                I[k] = I[k] * tr + Be;
                Mu += p.W[k] * I[k];
            }
            Bhi = Blo;
            const double r = wave_sum(w * Mu);
            if (lane == 0) red[i * nw + wv] = r;
            if (Mup && live) Mup[(size_t)i * nnu + j] = Mu;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * np; e += blockDim.x) {
        const double *q = red + e * nw;
        double t = q[0];
        for (int x = 1; x < nw; x++) t += q[x];
        if (f.ticket) flux_store_dev(&partial[(size_t)blockIdx.x * 2 * np + e], t);
        else partial[(size_t)blockIdx.x * 2 * np + e] = t;
    }
    if (f.ticket) flux_last_block_reduce(f, partial, (int)gridDim.x, 2 * np);
}

// (the register budget decides how many waves a SIMD holds: 168 registers = 3 waves, nothing spilled -- the default; 128 = 4 waves with a
//  few values in scratch: BASELINE configs[4] 1.59 vs 2.58 ms (profiles/r04_notes.md).  Both are kept for A/B, cs_set_tuning key 15 | 8)
#define CS_FLUX_CHUNK_KERNEL(NAME, WAVES)                                                                                                       \
    template <int NS>                                                                                                                           \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void NAME(                                              \
        RtParams p, const double *__restrict__ nu, const double *__restrict__ wts, int64_t nnu, const double *__restrict__ sigma,               \
        const double *__restrict__ muk, const double *__restrict__ P, const double *__restrict__ Tlev, const double *__restrict__ S_toa,         \
        const double *__restrict__ albedo, double *__restrict__ tau, double *__restrict__ Mup, double *__restrict__ Mdn,                         \
        double *__restrict__ partial, FluxFuse f, int ntile)                                                                                    \
    {                                                                                                                                           \
        flux_chunk_body<NS>(p, nu, wts, nnu, sigma, muk, P, Tlev, S_toa, albedo, tau, Mup, Mdn, partial, f, ntile);                             \
    }
CS_FLUX_CHUNK_KERNEL(k_flux_chunk, 4)
CS_FLUX_CHUNK_KERNEL(k_flux_chunk3, 3)
#undef CS_FLUX_CHUNK_KERNEL

// ---- AcceleratedAbsorber (absorbers.jl:114-203): per-wavenumber ln sigma on pressure knots, linear in ln P -----------------
// update!: L[i][nu] = max(ln sigma[i][nu], ln floatmin)  (absorbers.jl:185-196)
__global__ __launch_bounds__(256) void k_accel_store(int64_t n, const double *__restrict__ sigma, double *__restrict__ L)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double lt = log(2.2250738585072014e-308);
    const double l = log(sigma[i]);
    L[i] = (l < lt) ? lt : l;     // (ln 0 = -inf and NaN-free inputs: the comparison is the reference's)
}
// Sigma(A, i, T, P) = exp(phi_i(ln P)) at every node state k (absorbers.jl:203): LinearInterpolator without boundaries,
// cell[k] = knot interval, x[k] = ln P_k, xa/xb = its ends -- (x - xa)*(yb - ya)/(xb - xa) + ya.   sigma[k][nu] = base + extra + that
__global__ __launch_bounds__(256) void k_accel_eval(const double *__restrict__ L, int64_t nnu, int K, const int32_t *__restrict__ cell,
                                                     const double *__restrict__ x, const double *__restrict__ xa,
                                                     const double *__restrict__ xb, double base, const double *__restrict__ extra,
                                                     double *__restrict__ sigma)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= nnu) return;
    const int c = cell[k];
    const double ya = L[(size_t)c * nnu + i], yb = L[(size_t)(c + 1) * nnu + i];
    const size_t o = (size_t)k * nnu + i;
    sigma[o] = (base + (extra ? extra[o] : 0.0)) + exp((x[k] - xa[k]) * (yb - ya) / (xb[k] - xa[k]) + ya);
}

// test hook: the device functions the flux kernel is built from, evaluated point by point (0: exp_rt(x), 1: planck(x = nu, y = T),
// 2: layerplanck_inv(B1 = x, B2 = y, 1/tau = 1/z, t = exp_rt(-z)) with z > 0)
__global__ void k_devfn(int which, int64_t n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                        double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) out[i] = exp_rt(x[i]);
    else if (which == 1) out[i] = planck(x[i], y[i]);
    else out[i] = layerplanck_inv(x[i], y[i], 1.0 / z[i], exp_rt(-z[i]));
}

__global__ void k_faddeeva(int64_t n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fad_re(x[i], y[i]);
}

}  // namespace csdev
