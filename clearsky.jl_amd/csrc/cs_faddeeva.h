// cs_faddeeva.h -- Re w(x+iy) for gfx950, fp64.  Replaces Faddeyeva985.faddeyeva(x,y) at its only call site,
// fvoigt (reference src/absorption/line_shapes.jl:366-378, call :375).  Three regions in s = x^2+y^2:
//   s >= 1e4 : 4-term real asymptotic series (the branch 99% of (nu,line) pairs take; k_voigt_far runs 2-, 3- and 6-term
//              cuts of the same series where they are exact to 1e-15)
//   s >= 100 : 10-term Laplace continued fraction as the rational  z*PA(z^2)/PB(z^2)
//   s <  100 : trapezoid rule, h = 1/2, on the integer or half-shifted grid + pole correction when y < 2*pi
// Max relative error ~1e-14 against 40-digit mpmath (tools/faddeeva_proto.py); the reference's own Faddeeva
// (ACM TOMS Algorithm 985) is only ~4e-5 accurate, so this is strictly closer to the exact profile.
#pragma once
#include <hip/hip_runtime.h>

namespace csdev {

constexpr double kIsqPi = 0.5641895835477563;   // 1/sqrt(pi)
constexpr double kPi = 3.14159265358979323846;
constexpr double kFarS = 1.0e4;
constexpr double kMidS = 100.0;
constexpr double kSerS = 1.0e3;   // six-term asymptotic series is good to 2e-15 down to here (k_voigt_far near-zone pass)

// 1/s for s > 0 well inside the normal range: v_rcp_f64 seed + two Newton steps
__device__ __forceinline__ double rcp_nr(double s)
{
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// 1/s to ~2e-15 over the whole fp64 range: v_rcp_f64 seed (4.5e-8) + one Newton step
__device__ __forceinline__ double rcp_nr1(double s)
{
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    return __builtin_fma(r, e, r);
}

// 1/s to ~3e-15 for the far-wing series (s >= 1e4): single-precision reciprocal seed (v_cvt_f32_f64, v_rcp_f32,
// v_cvt_f64_f32 = 1.7 FMA-equivalents, against 3.2 for v_rcp_f64 and 12 for an IEEE division; tools/ubench) + one Newton
// step.  s beyond the f32 range gives a zero seed and a zero term (the true term is < 1e-38 of the line strength).
__device__ __forceinline__ double rcp_fast(double s)
{
    double r = (double)__builtin_amdgcn_rcpf((float)s);
    double e = __builtin_fma(-s, r, 1.0);
    return __builtin_fma(r, e, r);
}

// sqrt(pi)*K/y for s >= 1e4 given y^2 and 1/s:  inv*(1 + inv*(p1 + inv*(p2 + inv*p3))),  p_k polynomials in t = y^2/s
__device__ __forceinline__ double fad_far_core(double y2, double inv)
{
    double t = y2 * inv;
    double p3 = __builtin_fma(__builtin_fma(__builtin_fma(-120.0, t, 210.0), t, -105.0), t, 13.125);
    double p2 = __builtin_fma(__builtin_fma(12.0, t, -15.0), t, 3.75);
    double p1 = __builtin_fma(-2.0, t, 1.5);
    double inner = __builtin_fma(inv, __builtin_fma(inv, p3, p2), p1);
    return inv * __builtin_fma(inv, inner, 1.0);
}

__device__ __forceinline__ double fad_mid(double x, double y)
{
    double ur = x * x - y * y, ui = 2.0 * x * y;
    // PA = {180.9375, -330, 147, -22, 1}, PB = {-29.53125, 295.3125, -393.75, 157.5, -22.5, 1} (ascending in u)
    double ar = 1.0, ai = 0.0, br = 1.0, bi = 0.0, tr;
#define CS_H(cr, ci, c) tr = __builtin_fma(cr, ur, __builtin_fma(-ci, ui, c)); ci = __builtin_fma(cr, ui, ci * ur); cr = tr;
    CS_H(ar, ai, -22.0) CS_H(ar, ai, 147.0) CS_H(ar, ai, -330.0) CS_H(ar, ai, 180.9375)
    CS_H(br, bi, -22.5) CS_H(br, bi, 157.5) CS_H(br, bi, -393.75) CS_H(br, bi, 295.3125) CS_H(br, bi, -29.53125)
#undef CS_H
    double nr = x * ar - y * ai, ni = x * ai + y * ar;
    return kIsqPi * (nr * bi - ni * br) * rcp_nr(__builtin_fma(br, br, bi * bi));
}

__device__ __forceinline__ double fad_near(double x, double y)
{
    constexpr double c0[13] = {1.0, 0.7788007830714049, 0.36787944117144233, 0.10539922456186433, 0.01831563888873418,
                               0.0019304541362277093, 0.00012340980408667956, 4.785117392129009e-06,
                               1.1253517471925912e-07, 1.6052280551856116e-09, 1.3887943864964021e-11,
                               7.287724095819692e-14, 2.3195228302435696e-16};
    constexpr double c1[12] = {0.9394130628134758, 0.569782824730923, 0.2096113871510978, 0.04677062238395898,
                               0.006329715427485747, 0.0005195746821548384, 2.586810022265412e-05,
                               7.811489408304491e-07, 1.4307241918567688e-08, 1.5893910094516368e-10,
                               1.0709232382508077e-12, 4.37661850287085e-15};
    double y2 = y * y;
    double u = 2.0 * x;
    double fr = u - floor(u);
    bool shift = fabs(fr - 0.5) > 0.25;  // x near an integer node -> half-shifted grid
    double acc = shift ? 0.0 : rcp_nr(__builtin_fma(x, x, y2));
#pragma unroll
    for (int k = 0; k < 12; k++) {
        double t = 0.5 * (k + 1) - (shift ? 0.25 : 0.0);  // shifted: (k+1/2)h ; integer: (k+1)h
        double c = shift ? c1[k] : c0[k + 1];
        double a = x - t, b = x + t;
        double da = __builtin_fma(a, a, y2), db = __builtin_fma(b, b, y2);
        acc = __builtin_fma(c * (da + db), rcp_fast(da * db), acc);   // (|x -+ t| >= 1/8 by the choice of grid: 2e-4 <= da db <= 2e5, inside the
                                                                       // f32 range of the seed; 3.6e-15 per term against 1e-16 -- 24 terms of
                                                                       // one sign -- for 3.5 of the 12 FMA-equivalents a term costs)
    }
    double res = (0.5 / kPi) * y * acc;
    if (y < 2.0 * kPi) {
        double sgn = shift ? 1.0 : -1.0;
        double g = exp(-4.0 * kPi * y);
        double q = 4.0 * x;
        q -= 2.0 * rint(0.5 * q);
        double sph, cph;
        sincospi(q, &sph, &cph);
        double dr = __builtin_fma(sgn, cph, g), di = -sgn * sph;
        double em = exp(y2 - x * x);
        double sa, ca;
        sincospi((2.0 / kPi) * x * y, &sa, &ca);   // |2xy| < 126: the phase error of the scaled argument stays below 1e-14
        // Re[(ca - i sa)/(dr + i di)] = (ca*dr - sa*di)/|d|^2
        res += 2.0 * g * em * (ca * dr - sa * di) * rcp_nr(__builtin_fma(dr, dr, di * di));
    }
    return res;
}

// generic entry: Re w(|x| + i y)
__device__ __forceinline__ double fad_re(double x, double y)
{
    x = fabs(x);
    double y2 = y * y;
    double s = __builtin_fma(x, x, y2);
    if (s >= kFarS) return kIsqPi * y * fad_far_core(y2, 1.0 / s);
    if (s >= kMidS) return fad_mid(x, y);
    return fad_near(x, y);
}

}  // namespace csdev
