/* oracle/det_math.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product.
 *
 * Self-contained, platform-independent replacements for the libm calls that the
 * reference reaches through Rust `std` (f32::atan2 / sin / cos / exp / cbrt, used by
 * palette 0.7.6's Lab conversion and CIEDE2000: /root/reference/src/lib.rs:101-103,
 * 344-346, 1091-1099).  Rust forwards those to the platform libm, so the reference's
 * own results are platform-defined at the last ulp.  The oracle pins them instead:
 * every function below is evaluated in IEEE binary64 with only + - * / sqrt and
 * explicit operation order (no FMA contraction: compile with -ffp-contract=off),
 * then rounded once to binary32.  The HIP product carries an independent
 * transcription of the same algorithms (snesimage_amd/csrc/dmath.hpp); tests compare
 * the two bit-for-bit.
 *
 * Accuracy: each function's binary64 result is within ~1e-15 relative of the true
 * value, so the binary32 result equals the correctly rounded one except when the true
 * value lies within ~1e-15 of a rounding boundary.
 */
#ifndef SNES_ORACLE_DET_MATH_H
#define SNES_ORACLE_DET_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline double det_from_bits64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint32_t det_bits32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float det_from_bits32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* exp(x) for x <= 0 (the only use: CIEDE2000's delta-theta term).  k = rint(x/ln2),
 * r = x - k*ln2 (two-part ln2), Taylor to r^13, scale by 2^k through the exponent. */
static inline double det_exp_neg(double x)
{
    if (x < -700.0) return 0.0;
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01; /* 33 significant bits */
    const double ln2_lo = 1.90821492927058770002e-10;
    double kd = rint(x * inv_ln2);
    double r = (x - kd * ln2_hi) - kd * ln2_lo;
    double p = 1.0 / 6227020800.0; /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int64_t k = (int64_t)kd; /* in [-1010, 0] */
    return p * det_from_bits64((uint64_t)(k + 1023) << 52);
}

/* sin/cos core on |r| <= pi/4 (Taylor, error < 1e-16). */
static inline double det_sin_core(double r)
{
    double z = r * r;
    double p = -1.0 / 355687428096000.0; /* -1/17! */
    p = p * z + 1.0 / 1307674368000.0;    /* 1/15! */
    p = p * z - 1.0 / 6227020800.0;
    p = p * z + 1.0 / 39916800.0;
    p = p * z - 1.0 / 362880.0;
    p = p * z + 1.0 / 5040.0;
    p = p * z - 1.0 / 120.0;
    p = p * z + 1.0 / 6.0;
    return r - (r * z) * p;
}
static inline double det_cos_core(double r)
{
    double z = r * r;
    double p = 1.0 / 6402373705728000.0; /* 1/18! */
    p = p * z - 1.0 / 20922789888000.0;  /* 1/16! */
    p = p * z + 1.0 / 87178291200.0;
    p = p * z - 1.0 / 479001600.0;
    p = p * z + 1.0 / 3628800.0;
    p = p * z - 1.0 / 40320.0;
    p = p * z + 1.0 / 720.0;
    p = p * z - 1.0 / 24.0;
    p = p * z + 0.5;
    return 1.0 - z * p;
}
/* quadrant reduction, valid for |x| < 1e5 */
static inline double det_reduce_pio2(double x, int *q)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00; /* 33 bits */
    const double pio2_lo = 6.07710050650619224932e-11;
    double kd = rint(x * two_over_pi);
    *q = (int)((int64_t)kd & 3);
    return (x - kd * pio2_hi) - kd * pio2_lo;
}
static inline double det_sin(double x)
{
    int q; double r = det_reduce_pio2(x, &q);
    switch (q) { case 0: return det_sin_core(r); case 1: return det_cos_core(r);
                 case 2: return -det_sin_core(r); default: return -det_cos_core(r); }
}
static inline double det_cos(double x)
{
    int q; double r = det_reduce_pio2(x, &q);
    switch (q) { case 0: return det_cos_core(r); case 1: return -det_sin_core(r);
                 case 2: return -det_cos_core(r); default: return det_sin_core(r); }
}

/* atan on [0,1]: two half-angle reductions, then the Maclaurin series to t^21. */
static inline double det_atan01(double t)
{
    double t1 = t / (1.0 + sqrt(1.0 + t * t));
    double t2 = t1 / (1.0 + sqrt(1.0 + t1 * t1));
    double z = t2 * t2;
    double p = 1.0 / 21.0;
    p = 1.0 / 19.0 - z * p;
    p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p;
    p = 1.0 / 13.0 - z * p;
    p = 1.0 / 11.0 - z * p;
    p = 1.0 / 9.0 - z * p;
    p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p;
    p = 1.0 / 3.0 - z * p;
    p = 1.0 - z * p;
    return 4.0 * (t2 * p);
}
static inline double det_atan2(double y, double x)
{
    const double pi = 3.14159265358979323846;
    const double pio2 = 1.57079632679489661923;
    double ay = fabs(y), ax = fabs(x);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax) a = det_atan01(ay / ax);
    else a = pio2 - det_atan01(ax / ay);
    if (x < 0.0 || (x == 0.0 && signbit(x))) a = pi - a;
    return signbit(y) ? -a : a;
}

/* binary32 wrappers */
static inline float det_sinf(float x) { return (float)det_sin((double)x); }
static inline float det_cosf(float x) { return (float)det_cos((double)x); }
static inline float det_atan2f(float y, float x) { return (float)det_atan2((double)y, (double)x); }
static inline float det_expf_neg(float x) { return (float)det_exp_neg((double)x); }

/* cbrtf: the musl/FreeBSD algorithm (bit-hack seed + two Newton steps in binary64),
 * which is also what yuvxyb-math 0.1.1's `cbrtf` is believed to be.
 * Argument must be finite; zero returns zero; subnormals are not special-cased
 * (never produced by this path). */
static inline float det_cbrtf(float x)
{
    uint32_t ui = det_bits32(x);
    uint32_t hx = ui & 0x7fffffffu;
    if (hx == 0) return x;
    hx = hx / 3u + 709958130u; /* B1 */
    ui = (ui & 0x80000000u) | hx;
    double t = (double)det_from_bits32(ui);
    double xd = (double)x;
    double r = t * t * t;
    t = t * (xd + xd + r) / (xd + r + r);
    r = t * t * t;
    t = t * (xd + xd + r) / (xd + r + r);
    return (float)t;
}

#endif
