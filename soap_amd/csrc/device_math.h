// Device arithmetic helpers whose results must match NumPy bit-for-bit (or, where the
// host libm is only faithfully rounded, to the last ulp in all but rare cases).
// The whole translation unit is compiled with -ffp-contract=off: NumPy never fuses a
// multiply and an add, so neither may we; every fma below is deliberate.
#pragma once
#include <hip/hip_runtime.h>

#define SOAP_DEV __device__ __forceinline__

// NumPy's `a % b` for doubles (npy_divmod: fmod, then move the result to the sign of b).
// Used for the periodic wrap of SOAP/core/halo_tasks.py:113-117 and the centre-of-mass
// re-wrap of SO_properties.py:562-564.
SOAP_DEV double np_mod(double a, double b) {
    // fast paths, exact: for b > 0 and -b < a < 2b the IEEE remainder is a, a - b (Sterbenz) or a
    if (b > 0.0) {
        if (a >= 0.0) {
            if (a < b) return a;
            if (a < 2.0 * b) return a - b;
        } else if (a > -b) {
            return a + b;   // fmod(a, b) = a (negative, nonzero) and NumPy then adds b
        }
    }
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// x**3 as libm pow(x, 3.0) returns it: the correctly rounded cube.  (x*x)*x has two
// roundings; this evaluates the product in double-double and rounds once.
SOAP_DEV double cube(double x) {
    double x2 = x * x;
    double e2 = __builtin_fma(x, x, -x2);   // x^2  = x2 + e2 exactly
    double p = x2 * x;
    double ep = __builtin_fma(x2, x, -p);   // x2*x = p + ep exactly
    return p + (ep + e2 * x);
}

// position relative to the halo centre after the nearest-image wrap, exactly as
//   pos[:, :] = ((pos - offset) % boxsize) + offset ;  pos - centre
// (halo_tasks.py:116-117 followed by SO_properties.py:336 / aperture_properties.py:290-293)
SOAP_DEV double wrap_rel(double x, double c, double L) {
    double off = c - 0.5 * L;
    double w = np_mod(x - off, L) + off;
    return w - c;
}

// minimum-image separation of the mesh query (shared_mesh.py:138-142)
SOAP_DEV double min_image(double x, double c, double L) {
    double dr = x - c;
    if (dr > 0.5 * L) dr -= L;
    if (dr < -0.5 * L) dr += L;
    return dr;
}

// scipy.optimize.brentq (scipy/optimize/Zeros/brentq.c), defaults xtol=2e-12,
// rtol=4*eps, maxiter=100, applied to cumulative_mass_intersection
// (SO_properties.py:50-77):  f(u) = 4*pi/3*rho_d*u^3 - S_d*u + S_d - 1
struct SOIntersection {
    double c3;   // (4.0*pi/3.0)*rho_dim
    double s;    // slope_dim
    SOAP_DEV double operator()(double u) const { return ((c3 * cube(u) - s * u) + s) - 1.0; }
};

SOAP_DEV double brentq(const SOIntersection &f, double xa, double xb, int *err) {
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    const int iter = 100;
    double xpre = xa, xcur = xb;
    double xblk = 0., fpre, fcur, fblk = 0., spre = 0., scur = 0., sbis;
    double delta, stry, dpre, dblk;
    *err = 0;
    fpre = f(xpre);
    fcur = f(xcur);
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    if (signbit(fpre) == signbit(fcur)) { *err = -1; return 0.; }
    for (int i = 0; i < iter; i++) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double lim = fmin(fabs(spre), 3 * fabs(sbis) - delta);
            if (2 * fabs(stry) < lim) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = f(xcur);
    }
    *err = -2;
    return xcur;
}
