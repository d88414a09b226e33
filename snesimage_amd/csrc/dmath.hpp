// snesimage_amd/csrc/dmath.hpp — deterministic math shared by the HIP kernels and the library's
// host code.  Every function is evaluated in IEEE binary64 with + - * / sqrt only (no FMA
// contraction: the translation unit is compiled with -ffp-contract=off) and rounded once to
// binary32, so host and gfx950 produce identical bits.  These stand in for the platform libm that
// the reference reaches through Rust std (f32::{sin,cos,atan2,exp,cbrt} inside palette 0.7.6 and
// yuvxyb-math 0.1.1; call sites /root/reference/src/lib.rs:101-103, 344-346, 1091-1099, 547).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SNES_HD __host__ __device__ __forceinline__

namespace snes {

SNES_HD double f64_from_bits(uint64_t u) { union { uint64_t u; double d; } v; v.u = u; return v.d; }
SNES_HD uint32_t f32_bits(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
SNES_HD float f32_from_bits(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
SNES_HD bool f64_signbit(double d) { union { double d; uint64_t u; } v; v.d = d; return (v.u >> 63) != 0; }

// cube root: the musl / yuvxyb-math cbrtf — bit-hack seed, two Halley steps in binary64 (each with a division), the
// binary64 result rounded to float.  That binary64 value t is within 6.3e-15 (2^-47) of the true root, so ANY binary64
// approximation t' within 5e-16 rounds to the same float unless t' lies within ~7e-15 (relative) of a float rounding
// boundary.  d_cbrtf computes such a t' without divisions — x^(-1/3) by three cubically convergent steps
// y <- y + y*e*(1/3 + 2e/9), e = 1 - x*y^3, then x*y*y — and accepts it when both ends of a +-3e-14 band round to the
// same float (all but ~5e-7 of the arguments); otherwise it runs the reference sequence itself.  Same floats, a third
// of the cost: two binary64 divisions were most of linear_to_positive_xyb.
SNES_HD float d_cbrtf_ref(float x) {
    uint32_t ui = f32_bits(x);
    uint32_t hx = ui & 0x7fffffffu;
    if (hx == 0u) return x;
    hx = hx / 3u + 709958130u;
    double t = (double)f32_from_bits((ui & 0x80000000u) | hx);
    double xd = (double)x;
    double r = t * t * t;
    t = t * (xd + xd + r) / (xd + r + r);
    r = t * t * t;
    t = t * (xd + xd + r) / (xd + r + r);
    return (float)t;
}
SNES_HD float d_cbrtf(float x) {
    const uint32_t ui = f32_bits(x), hx = ui & 0x7fffffffu;
    if (hx - 0x00800000u >= 0x7f000000u) return d_cbrtf_ref(x); // zero, subnormal, inf, nan: not the fast path's business
    const double ax = (double)f32_from_bits(hx);
    double y = (double)f32_from_bits(0x54a21d2au - hx / 3u); // |x|^(-1/3) to ~3.5 %
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double e = fma(-(ax * y), y * y, 1.0);
        y = fma(y * e, fma(e, 2.0 / 9.0, 1.0 / 3.0), y);
    }
    const double t = (ax * y) * y;
    const float lo = (float)(t * (1.0 - 3e-14)), hi = (float)(t * (1.0 + 3e-14));
    if (lo != hi) return d_cbrtf_ref(x);
    return f32_from_bits((ui & 0x80000000u) | f32_bits(lo));
}

// exp(x), x <= 0.
SNES_HD double d_exp_neg(double x) {
    if (x < -700.0) return 0.0;
    double kd = rint(x * 1.44269504088896338700e+00);
    double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
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
    long long k = (long long)kd;
    return p * f64_from_bits((uint64_t)(k + 1023) << 52);
}

SNES_HD double d_sin_core(double r) {
    double z = r * r;
    double p = -1.0 / 355687428096000.0;
    p = p * z + 1.0 / 1307674368000.0;
    p = p * z - 1.0 / 6227020800.0;
    p = p * z + 1.0 / 39916800.0;
    p = p * z - 1.0 / 362880.0;
    p = p * z + 1.0 / 5040.0;
    p = p * z - 1.0 / 120.0;
    p = p * z + 1.0 / 6.0;
    return r - (r * z) * p;
}
SNES_HD double d_cos_core(double r) {
    double z = r * r;
    double p = 1.0 / 6402373705728000.0;
    p = p * z - 1.0 / 20922789888000.0;
    p = p * z + 1.0 / 87178291200.0;
    p = p * z - 1.0 / 479001600.0;
    p = p * z + 1.0 / 3628800.0;
    p = p * z - 1.0 / 40320.0;
    p = p * z + 1.0 / 720.0;
    p = p * z - 1.0 / 24.0;
    p = p * z + 0.5;
    return 1.0 - z * p;
}
SNES_HD double d_reduce_pio2(double x, int &q) {
    double kd = rint(x * 6.36619772367581382433e-01);
    q = (int)((long long)kd & 3);
    return (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
}
SNES_HD double d_sin(double x) {
    int q; double r = d_reduce_pio2(x, q);
    double s = d_sin_core(r), c = d_cos_core(r);
    return q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
}
SNES_HD double d_cos(double x) {
    int q; double r = d_reduce_pio2(x, q);
    double s = d_sin_core(r), c = d_cos_core(r);
    return q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}
SNES_HD double d_atan01(double t) {
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
SNES_HD double d_atan2(double y, double x) {
    const double pi = 3.14159265358979323846, pio2 = 1.57079632679489661923;
    double ay = fabs(y), ax = fabs(x), a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax) a = d_atan01(ay / ax);
    else a = pio2 - d_atan01(ax / ay);
    if (x < 0.0 || (x == 0.0 && f64_signbit(x))) a = pi - a;
    return f64_signbit(y) ? -a : a;
}
SNES_HD float d_sinf(float x) { return (float)d_sin((double)x); }
SNES_HD float d_cosf(float x) { return (float)d_cos((double)x); }
SNES_HD float d_atan2f(float y, float x) { return (float)d_atan2((double)y, (double)x); }
SNES_HD float d_expf_neg(float x) { return (float)d_exp_neg((double)x); }

} // namespace snes
