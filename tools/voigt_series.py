"""Far-wing expansion of the Voigt function used by k_cheb_sep (CPU only; needs mpmath).

sqrt(pi) K(x,y)/y = sum_{n>=1} c_n(y^2) / x^(2n)   from  K = Re w(x+iy),  w(z) ~ (i/(sqrt(pi) z)) sum_k (2k-1)!!/(2 z^2)^k,
c_n polynomials of degree n-1 in y^2 with rational coefficients (printed; kSepPoly in cs_kernels.h holds n = 1..8).
Truncation after n terms is below ((y^2 + a_n)/x^2)^n, a_4 ~ 3, a_8 ~ 4.4: 1e-17 needs (y^2+3)/x^2 <= 5.6e-5 (n = 4) or
(y^2+4.4)/x^2 <= 7.5e-3 (n = 8); with x = sqrt(ln2) dnu/alpha, y = sqrt(ln2) gamma/alpha that is
|dnu| >= 133.6 sqrt(gamma^2 + 4.33 alpha^2)  resp.  |dnu| >= 11.55 sqrt(gamma^2 + 6.35 alpha^2)   (k_izones: S4, S8).
"""
from fractions import Fraction as Fr
from math import comb


def coefficients(N=9):
    def dfact(k):
        r = 1
        for q in range(1, 2 * k, 2):
            r *= q
        return r
    coef = {}
    for k in range(0, N + 2):
        a, m = Fr(dfact(k), 2 ** k), 2 * k + 1
        for j in range(1, 2 * N + 3, 2):           # Re[i z^-m] picks the odd powers of (i y/x)
            n, p = (m + j) // 2, (j - 1) // 2
            if n <= N:
                coef[(n, p)] = coef.get((n, p), 0) + a * (-1) ** j * comb(m + j - 1, j) * (-1) ** ((j + 1) // 2)
    return coef


if __name__ == "__main__":
    from mpmath import mp, mpf, erfc, exp, sqrt, pi, re
    mp.dps = 40
    c = coefficients()
    for n in range(1, 9):
        print(n, [str(c[(n, p)]) for p in range(n)], [float(c[(n, p)]) for p in range(n)])
    K = lambda x, y: re(exp(-(mpf(x) + 1j * mpf(y)) ** 2) * erfc(-1j * (mpf(x) + 1j * mpf(y))))
    print("relative truncation error at the edge of the validity regions")
    for nt, eps, an in ((4, 5.6e-5, 3.0), (8, 7.5e-3, 4.4)):
        for y in (1e-3, 1.0, 8.0, 60.0):
            x = ((y * y + an) / eps) ** 0.5
            exact = K(x, y) * sqrt(pi) / y
            s = sum(mpf(c[(n, p)].numerator) / c[(n, p)].denominator * mpf(y) ** (2 * p) / mpf(x) ** (2 * n)
                    for n in range(1, nt + 1) for p in range(n))
            print(f"  n = {nt}, y = {y:g}, x = {x:.1f}: {float(abs(s / exact - 1)):.1e}")
