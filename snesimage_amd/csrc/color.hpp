// snesimage_amd/csrc/color.hpp — colour primitives of the hot path, host + device.
// Reference: /root/reference/src/lib.rs:628-795 (SnesColor, NES table, nearest-entry argmin) and
// :1080-1100 (distances); third-party arithmetic per SURVEY App. A (palette 0.7.6, yuvxyb 0.4.2).
#pragma once
#include "dmath.hpp"
#include "../../include/ssimulacra2_constants.h" // the one copy of the restated crates' constants (shared with the oracle)

namespace snes {

// SnesColor::as_rgba, lib.rs:662-669: `v*8 + v/4` in u8 arithmetic (wraps for v >= 32, quirk Q3).
SNES_HD uint32_t expand5(uint32_t v) { return ((v * 8u) + (v / 4u)) & 0xffu; }
// raw 5-bit triple -> packed 0x00BBGGRR of the 8-bit expansion
SNES_HD uint32_t rgb5_to_rgb8(uint32_t r, uint32_t g, uint32_t b) { return expand5(r) | (expand5(g) << 8) | (expand5(b) << 16); }
// SnesColor::as_u16, lib.rs:679-681
SNES_HD uint16_t rgb5_as_u16(uint32_t r, uint32_t g, uint32_t b) { return (uint16_t)(r + (g << 5) + (b << 10)); }

// color_distance_red_mean (lib.rs:1080-1088) without the sqrt, times 512: every term is an exact
// integer (max 299,505,150 < 2^31), and sqrt is strictly monotone, so ordering by this key equals
// ordering by the reference's f64 distance, ties included.
SNES_HD uint32_t red_mean_key(uint32_t c1, uint32_t c2) {
    int r1 = c1 & 0xff, r2 = c2 & 0xff;
    int dr = r1 - r2, dg = (int)((c1 >> 8) & 0xff) - (int)((c2 >> 8) & 0xff), db = (int)((c1 >> 16) & 0xff) - (int)((c2 >> 16) & 0xff);
    int rs = r1 + r2;
#ifdef __HIP_DEVICE_COMPILE__
    // every factor fits 24 bits (|d| <= 255, d*d <= 65,025, weights <= 1,534): full-rate v_mul/v_mad_*24 instead of
    // the quarter-rate 32-bit multiply
    uint32_t k = (uint32_t)(dg * dg) << 11;
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(k) : "v"(1024 + rs), "v"(dr * dr));
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(k) : "v"(1534 - rs), "v"(db * db));
    return k;
#else
    return (uint32_t)((1024 + rs) * dr * dr + 2048 * dg * dg + (1534 - rs) * db * db);
#endif
}

// NES table, lib.rs:685-745
__device__ __constant__ const uint8_t kNesTableDev[56][3] = {
    {13, 13, 13}, {0, 2, 16},   {3, 0, 17},   {7, 0, 15},   {10, 0, 10},  {11, 0, 3},   {9, 2, 0},
    {7, 3, 0},    {4, 6, 0},    {0, 7, 0},    {0, 8, 0},    {0, 7, 4},    {0, 5, 10},   {0, 0, 0},
    {23, 23, 23}, {3, 10, 24},  {9, 6, 28},   {14, 4, 26},  {18, 3, 21},  {19, 5, 11},  {19, 6, 0},
    {15, 9, 0},   {11, 12, 0},  {4, 14, 0},   {0, 15, 0},   {0, 14, 8},   {0, 13, 17},  {0, 0, 0},
    {31, 31, 31}, {13, 20, 31}, {17, 19, 31}, {22, 16, 31}, {27, 14, 31}, {28, 14, 23}, {28, 17, 13},
    {26, 19, 5},  {22, 21, 1},  {15, 24, 2},  {10, 25, 8},  {8, 25, 16},  {8, 24, 24},  {9, 9, 9},
    {31, 31, 31}, {25, 29, 31}, {27, 27, 31}, {29, 27, 31}, {31, 26, 31}, {31, 26, 30}, {31, 27, 25},
    {31, 28, 22}, {30, 30, 21}, {27, 31, 21}, {25, 31, 23}, {24, 31, 26}, {24, 30, 30}, {23, 24, 23}};
static const uint8_t kNesTableHost[56][3] = {
    {13, 13, 13}, {0, 2, 16},   {3, 0, 17},   {7, 0, 15},   {10, 0, 10},  {11, 0, 3},   {9, 2, 0},
    {7, 3, 0},    {4, 6, 0},    {0, 7, 0},    {0, 8, 0},    {0, 7, 4},    {0, 5, 10},   {0, 0, 0},
    {23, 23, 23}, {3, 10, 24},  {9, 6, 28},   {14, 4, 26},  {18, 3, 21},  {19, 5, 11},  {19, 6, 0},
    {15, 9, 0},   {11, 12, 0},  {4, 14, 0},   {0, 15, 0},   {0, 14, 8},   {0, 13, 17},  {0, 0, 0},
    {31, 31, 31}, {13, 20, 31}, {17, 19, 31}, {22, 16, 31}, {27, 14, 31}, {28, 14, 23}, {28, 17, 13},
    {26, 19, 5},  {22, 21, 1},  {15, 24, 2},  {10, 25, 8},  {8, 25, 16},  {8, 24, 24},  {9, 9, 9},
    {31, 31, 31}, {25, 29, 31}, {27, 27, 31}, {29, 27, 31}, {31, 26, 31}, {31, 26, 30}, {31, 27, 25},
    {31, 28, 22}, {30, 30, 21}, {27, 31, 21}, {25, 31, 23}, {24, 31, 26}, {24, 30, 30}, {23, 24, 23}};
constexpr uint32_t kNesColorCount = 56;

// ---- palette 0.7.6 Lab pipeline, f32 ----------------------------------------------------------
struct Lab { float l, a, b; };

SNES_HD float lab_f(float c) {
    const float epsilon = (float)PALETTE_LAB_EPS_ROOT_D * (float)PALETTE_LAB_EPS_ROOT_D * (float)PALETTE_LAB_EPS_ROOT_D; // include/ssimulacra2_constants.h
    const float kappa = (float)PALETTE_LAB_KAPPA_D;
    const float delta = (float)PALETTE_LAB_DELTA_D;
    return c > epsilon ? d_cbrtf(c) : (kappa * c) + delta;
}
// lin_* : linear-light components from the 256-entry sRGB table (built on the host, see capi)
SNES_HD Lab linear_to_lab(float r, float g, float b) {
    float x = (r * PALETTE_XYZ_XR) + (g * PALETTE_XYZ_XG) + (b * PALETTE_XYZ_XB);
    float y = (r * PALETTE_XYZ_YR) + (g * PALETTE_XYZ_YG) + (b * PALETTE_XYZ_YB);
    float z = (r * PALETTE_XYZ_ZR) + (g * PALETTE_XYZ_ZG) + (b * PALETTE_XYZ_ZB);
    x = x / PALETTE_D65_X; y = y / PALETTE_D65_Y; z = z / PALETTE_D65_Z;
    float fx = lab_f(x), fy = lab_f(y), fz = lab_f(z);
    Lab o; o.l = (fy * PALETTE_LAB_L_SCALE) - PALETTE_LAB_L_OFFSET; o.a = (fx - fy) * PALETTE_LAB_A_SCALE; o.b = (fy - fz) * PALETTE_LAB_B_SCALE;
    return o;
}

SNES_HD float ciede_hprime(float b, float ap) {
    if (b == 0.0f && ap == 0.0f) return 0.0f;
    float r = d_atan2f(b, ap) * (float)(180.0 / 3.14159265358979323846);
    return r < 0.0f ? r + 360.0f : r;
}
// A sure "no" for the win test `ciede2000(c1, c2) < bound` (or <= on a tie) without evaluating the formula:
// dE00^2 = (dL/S_L)^2 + (dC/S_C)^2 + (dH/S_H)^2 + R_T (dC/S_C)(dH/S_H) with |R_T| <= 2, so the last three terms are at
// least (|dC/S_C| - |dH/S_H|)^2 >= 0 and dE00 >= |dL| / S_L; S_L = 1 + 0.015 (Lm-50)^2 / sqrt(20 + (Lm-50)^2) <= 1.7471
// for Lm in [0,100].  The margin (1.752 against 1.7471: 0.28 %) dwarfs the rounding of the f32 evaluation (~1e-6), so
// whenever this returns true the computed distance is strictly above the bound.  Random candidate colours differ from a
// pixel in lightness by far more than the pixel's current error most of the time: this spares ~85 % of the evaluations.
SNES_HD bool ciede2000_cannot_beat(const Lab &c1, const Lab &c2, float bound) { return fabsf(c1.l - c2.l) > 1.752f * bound; }

// A second sure "no", on lightness and the a-b plane together.  With t_C = dC'/S_C and t_H = dH'/S_H,
//   t_C^2 + t_H^2 + R_T t_C t_H >= (1 - |R_T| / 2) (t_C^2 + t_H^2)                    (|t_C t_H| <= half the sum of squares)
//   t_C^2 + t_H^2 >= (dC'^2 + dH'^2) / max(S_C, S_H)^2,   dC'^2 + dH'^2 = da'^2 + db^2 >= da^2 + db^2
// (the law of cosines with dH' = 2 sqrt(C1' C2') sin(dh'/2); a' = (1 + G) a with the same G in [0, 0.5] on both sides),
//   S_C = 1 + 0.045 Cm', S_H = 1 + 0.015 Cm' T <= 1 + 0.029 Cm' (T <= 1.93), Cm' <= 1.5 (C1 + C2) / 2,
//   |R_T| = R_C |sin(2 dtheta)| <= 0.8661 R_C (2 dtheta <= 60 degrees), R_C increasing in Cm',
// so dE00^2 >= (dL / 1.7471)^2 + (1 - 0.8661 R_C(0.75 s) / 2) (da^2 + db^2) / (1 + 0.03375 s)^2, s = C1 + C2.  Every
// constant below is rounded to the safe side and the bound is raised by 0.5 % and 1e-3 before the comparison (the f32
// evaluation of the formula itself is good to ~1e-4 absolute at worst, where C' or h' cancel): whenever this returns true
// the computed distance is strictly above `bound`.  Of the pixels the lightness test lets through, this rules out
// most: a random candidate colour is far from a pixel in chroma more often than it is near.
SNES_HD bool ciede2000_cannot_beat_ab(const Lab &c1, float ch1, const Lab &c2, float bound) {
    const float ch2 = sqrtf(c2.a * c2.a + c2.b * c2.b);
    const float s = ch1 + ch2, u = 0.7501f * s;
    const float u2 = u * u, u4 = u2 * u2, u7 = (u * u2) * u4;
    const float dl = c1.l - c2.l, da = c1.a - c2.a, db = c1.b - c2.b;
    const float sm = 1.0f + 0.03376f * s;
#ifdef __HIP_DEVICE_COMPILE__
    const float rcq = sqrtf(__fdividef(u7, u7 + 6103515625.0f)) * 1.0001f; // R_C / 2 (fast division: within the margins)
    const float lb2 = 0.3257f * dl * dl + __fdividef((1.0f - 0.8661f * rcq) * (da * da + db * db), sm * sm * 1.0001f);
#else
    const float rcq = sqrtf(u7 / (u7 + 6103515625.0f)) * 1.0001f;
    const float lb2 = 0.3257f * dl * dl + (1.0f - 0.8661f * rcq) * (da * da + db * db) / (sm * sm * 1.0001f);
#endif
    const float b = 1.005f * bound + 1e-3f;
    return lb2 > b * b;
}

// palette::color_difference::Ciede2000 for Lab<_, f32>, kL = kC = kH = 1
SNES_HD float ciede2000(Lab c1, Lab c2) {
    const float pi_over_180 = (float)(3.14159265358979323846 / 180.0);
    const float p25_7 = 6103515625.0f;
    float ch1 = sqrtf(c1.a * c1.a + c1.b * c1.b), ch2 = sqrtf(c2.a * c2.a + c2.b * c2.b);
    float c_bar = (ch1 + ch2) / 2.0f;
    float cb2 = c_bar * c_bar, cb4 = cb2 * cb2;
    float cb7 = (c_bar * cb2) * cb4;
    float g = 0.5f * (1.0f - sqrtf(cb7 / (cb7 + p25_7)));
    float a1p = c1.a * (1.0f + g), a2p = c2.a * (1.0f + g);
    float c1p = sqrtf(a1p * a1p + c1.b * c1.b), c2p = sqrtf(a2p * a2p + c2.b * c2.b);
    float h1p = ciede_hprime(c1.b, a1p), h2p = ciede_hprime(c2.b, a2p);
    float hd = h2p - h1p, had = fabsf(hd);
    bool zc = (c1p == 0.0f) || (c2p == 0.0f);
    float dh;
    if (zc) dh = 0.0f;
    else if (had <= 180.0f) dh = hd;
    else if (h2p <= h1p) dh = hd + 360.0f;
    else dh = hd - 360.0f;
    float dH = 2.0f * sqrtf(c1p * c2p) * d_sinf(dh / 2.0f * pi_over_180);
    float hs = h1p + h2p;
    float hbar;
    if (zc) hbar = hs;
    else if (had > 180.0f) hbar = (hs + 360.0f) / 2.0f;
    else hbar = hs / 2.0f;
    float lbar = (c1.l + c2.l) / 2.0f;
    float cbp = (c1p + c2p) / 2.0f;
    float t = 1.0f - 0.17f * d_cosf((hbar - 30.0f) * pi_over_180) + 0.24f * d_cosf((hbar * 2.0f) * pi_over_180)
              + 0.32f * d_cosf((hbar * 3.0f + 6.0f) * pi_over_180) - 0.20f * d_cosf((hbar * 4.0f - 63.0f) * pi_over_180);
    float lm = lbar - 50.0f;
    float sl = 1.0f + ((0.015f * lm * lm) / sqrtf(lm * lm + 20.0f));
    float sc = 1.0f + 0.045f * cbp;
    float sh = 1.0f + 0.015f * cbp * t;
    float hb = (hbar - 275.0f) / 25.0f;
    float dtheta = 30.0f * d_expf_neg(-(hb * hb));
    float cp2 = cbp * cbp, cp4 = cp2 * cp2;
    float cp7 = (cbp * cp2) * cp4;
    float rc = 2.0f * sqrtf(cp7 / (cp7 + p25_7));
    float rt = -rc * d_sinf(2.0f * dtheta * pi_over_180);
    float dl = c2.l - c1.l, dc = c2p - c1p;
    float tl = dl / sl, tc = dc / sc, th = dH / sh;
    return sqrtf(tl * tl + tc * tc + th * th + (rt * dc * dH) / (sc * sh));
}

// ---- yuvxyb 0.4.2 linear RGB -> XYB, then ssimulacra2's make_positive_xyb ----------------------
SNES_HD void linear_to_positive_xyb(float r, float g, float b, float &X, float &Y, float &B) {
    const float m02 = SSIM2_OPSIN_M02, m00 = SSIM2_OPSIN_M00, m01 = 1.0f - m02 - m00; // include/ssimulacra2_constants.h
    const float m12 = SSIM2_OPSIN_M12, m10 = SSIM2_OPSIN_M10, m11 = 1.0f - m12 - m10;
    const float m20 = SSIM2_OPSIN_M20, m21 = SSIM2_OPSIN_M21, m22 = 1.0f - m20 - m21;
    const float b0 = SSIM2_OPSIN_BIAS;
    const float b0_root = SSIM2_OPSIN_BIAS_CBRT;
    float a0 = fmaf(m00, r, fmaf(m01, g, fmaf(m02, b, b0)));
    float a1 = fmaf(m10, r, fmaf(m11, g, fmaf(m12, b, b0)));
    float a2 = fmaf(m20, r, fmaf(m21, g, fmaf(m22, b, b0)));
    if (a0 < 0.0f) a0 = 0.0f;
    if (a1 < 0.0f) a1 = 0.0f;
    if (a2 < 0.0f) a2 = 0.0f;
    a0 = d_cbrtf(a0) - b0_root; a1 = d_cbrtf(a1) - b0_root; a2 = d_cbrtf(a2) - b0_root;
    float x = 0.5f * (a0 - a1), y = 0.5f * (a0 + a1);
    B = (a2 - y) + SSIM2_POS_B_OFFSET;
    X = fmaf(x, SSIM2_POS_X_SCALE, SSIM2_POS_X_OFFSET);
    Y = y + SSIM2_POS_Y_OFFSET;
}

} // namespace snes
