// oracle/snes_oracle.cpp — TEST INFRASTRUCTURE (see snes_oracle.h for the parity status).
//
// CPU restatement of the snesimage optimizer hot path.  Every function cites the reference
// lines (/root/reference/src/lib.rs unless noted) or the third-party crate it restates.
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (oracle/Makefile).  All fused
// multiply-adds that the restated Rust code spells `mul_add` are explicit fmaf()/fma() calls;
// nothing else may be contracted.
#include "snes_oracle.h"
#include "../include/ssimulacra2_constants.h" // the one copy of the restated crates' constants (shared with the product)
#include "det_math.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;
int fail(const char *msg) { g_last_error = msg; return -1; }

// ---------------------------------------------------------------------------------------------
// Colour primitives (lib.rs:628-795, 1080-1100)
// ---------------------------------------------------------------------------------------------
struct Rgb8 { uint8_t r, g, b; };

// lib.rs:662-669 — `v * 8 + v / 4` in u8 arithmetic.  The reference overflows for v >= 32
// (quirk Q3: debug panics, release wraps); the oracle keeps release-mode wrapping.
inline uint8_t expand5(uint8_t v) { return (uint8_t)((uint8_t)(v * 8u) + (uint8_t)(v / 4u)); }
inline Rgb8 snes_as_rgba(const uint8_t *c) { return Rgb8{expand5(c[0]), expand5(c[1]), expand5(c[2])}; }
// lib.rs:679-681
inline uint16_t snes_as_u16(const uint8_t *c) {
    return (uint16_t)((uint16_t)c[0] + (uint16_t)((uint16_t)c[1] << 5) + (uint16_t)((uint16_t)c[2] << 10));
}

// lib.rs:685-745
const uint8_t NES_TABLE[56][3] = {
    {13, 13, 13}, {0, 2, 16},   {3, 0, 17},   {7, 0, 15},   {10, 0, 10},  {11, 0, 3},   {9, 2, 0},
    {7, 3, 0},    {4, 6, 0},    {0, 7, 0},    {0, 8, 0},    {0, 7, 4},    {0, 5, 10},   {0, 0, 0},
    {23, 23, 23}, {3, 10, 24},  {9, 6, 28},   {14, 4, 26},  {18, 3, 21},  {19, 5, 11},  {19, 6, 0},
    {15, 9, 0},   {11, 12, 0},  {4, 14, 0},   {0, 15, 0},   {0, 14, 8},   {0, 13, 17},  {0, 0, 0},
    {31, 31, 31}, {13, 20, 31}, {17, 19, 31}, {22, 16, 31}, {27, 14, 31}, {28, 14, 23}, {28, 17, 13},
    {26, 19, 5},  {22, 21, 1},  {15, 24, 2},  {10, 25, 8},  {8, 25, 16},  {8, 24, 24},  {9, 9, 9},
    {31, 31, 31}, {25, 29, 31}, {27, 27, 31}, {29, 27, 31}, {31, 26, 31}, {31, 26, 30}, {31, 27, 25},
    {31, 28, 22}, {30, 30, 21}, {27, 31, 21}, {25, 31, 23}, {24, 31, 26}, {24, 30, 30}, {23, 24, 23}};
const uint32_t NES_COLOR_COUNT = 56; // lib.rs:31
inline void nes_color(uint32_t index, uint8_t *out) {
    if (index < NES_COLOR_COUNT) { out[0] = NES_TABLE[index][0]; out[1] = NES_TABLE[index][1]; out[2] = NES_TABLE[index][2]; }
    else { out[0] = out[1] = out[2] = 0; } // lib.rs:743
}

// lib.rs:1080-1088
inline double distance_red_mean(Rgb8 c1, Rgb8 c2) {
    double red_mean = ((double)c1.r + (double)c2.r) / 2.0; // f64::midpoint of two small integers is exact
    double r = (double)c1.r - (double)c2.r;
    double g = (double)c1.g - (double)c2.g;
    double b = (double)c1.b - (double)c2.b;
    return std::sqrt((((512.0 + red_mean) * r * r) / 256.0) + 4.0 * g * g + (((767.0 - red_mean) * b * b) / 256.0));
}
// 512 x the pre-sqrt value, exact in u32: the integer key the HIP remap orders by.
inline uint32_t red_mean_key(Rgb8 c1, Rgb8 c2) {
    int32_t rs = (int32_t)c1.r + (int32_t)c2.r;
    int32_t r = (int32_t)c1.r - (int32_t)c2.r, g = (int32_t)c1.g - (int32_t)c2.g, b = (int32_t)c1.b - (int32_t)c2.b;
    return (uint32_t)((1024 + rs) * r * r + 2048 * g * g + (1534 - rs) * b * b);
}

// ---- palette 0.7.6: Srgb<u8> -> Srgb<f32> -> LinSrgb -> Xyz(D65) -> Lab, all f32 ------------
struct SrgbLut {
    float lin[256];
    SrgbLut() {
        for (int v = 0; v < 256; v++) {
            float x = (float)v / 255.0f; // into_format: u8 -> f32 stimulus
            // Srgb::into_linear (constants: include/ssimulacra2_constants.h, PALETTE_SRGB_*), f32; powf is
            // evaluated through binary64 pow and rounded once (correctly rounded powf).
            if (x <= PALETTE_SRGB_THRESHOLD) lin[v] = (float)(1.0 / PALETTE_SRGB_LINEAR_DIV_D) * x;
            else {
                float t = fmaf(x, (float)(1.0 / PALETTE_SRGB_SCALE_D), (float)(PALETTE_SRGB_OFFSET_D / PALETTE_SRGB_SCALE_D));
                lin[v] = (float)std::pow((double)t, (double)PALETTE_SRGB_GAMMA);
            }
        }
    }
};
const SrgbLut &srgb_lut() { static SrgbLut l; return l; }

struct Lab32 { float l, a, b; };

inline float lab_f(float c) {
    const float epsilon = (float)PALETTE_LAB_EPS_ROOT_D * (float)PALETTE_LAB_EPS_ROOT_D * (float)PALETTE_LAB_EPS_ROOT_D; // powi(3)
    const float kappa = (float)PALETTE_LAB_KAPPA_D;
    const float delta = (float)PALETTE_LAB_DELTA_D;
    return c > epsilon ? det_cbrtf(c) : (kappa * c) + delta;
}
inline Lab32 srgb8_to_lab(Rgb8 c) {
    const SrgbLut &L = srgb_lut();
    float r = L.lin[c.r], g = L.lin[c.g], b = L.lin[c.b];
    // sRGB -> XYZ (D65), Lindbloom's matrix as carried by palette's Srgb space.
    float x = (r * PALETTE_XYZ_XR) + (g * PALETTE_XYZ_XG) + (b * PALETTE_XYZ_XB);
    float y = (r * PALETTE_XYZ_YR) + (g * PALETTE_XYZ_YG) + (b * PALETTE_XYZ_YB);
    float z = (r * PALETTE_XYZ_ZR) + (g * PALETTE_XYZ_ZG) + (b * PALETTE_XYZ_ZB);
    // Xyz -> Lab: divide by the D65 white point, f(), then the affine map.
    x = x / PALETTE_D65_X; y = y / PALETTE_D65_Y; z = z / PALETTE_D65_Z;
    float fx = lab_f(x), fy = lab_f(y), fz = lab_f(z);
    Lab32 o; o.l = (fy * PALETTE_LAB_L_SCALE) - PALETTE_LAB_L_OFFSET; o.a = (fx - fy) * PALETTE_LAB_A_SCALE; o.b = (fy - fz) * PALETTE_LAB_B_SCALE;
    return o;
}

// palette 0.7.6 color_difference::get_ciede2000_difference, T = f32 (k_L = k_C = k_H = 1).
inline float ciede2000(Lab32 c1, Lab32 c2) {
    const float pi_over_180 = (float)(3.14159265358979323846 / 180.0);
    const float twenty_five_pow_seven = 6103515625.0f;
    float chroma1 = std::sqrt(c1.a * c1.a + c1.b * c1.b);
    float chroma2 = std::sqrt(c2.a * c2.a + c2.b * c2.b);
    float c_bar = (chroma1 + chroma2) / 2.0f;
    // powi(7): LLVM's square-and-multiply expansion, x^7 = (x * x^2) * x^4
    float cb2 = c_bar * c_bar, cb4 = cb2 * cb2;
    float c_bar_pow_seven = (c_bar * cb2) * cb4;
    float g = 0.5f * (1.0f - std::sqrt(c_bar_pow_seven / (c_bar_pow_seven + twenty_five_pow_seven)));
    float a_one_prime = c1.a * (1.0f + g);
    float a_two_prime = c2.a * (1.0f + g);
    float c_one_prime = std::sqrt(a_one_prime * a_one_prime + c1.b * c1.b);
    float c_two_prime = std::sqrt(a_two_prime * a_two_prime + c2.b * c2.b);
    auto calc_h_prime = [](float b, float a_prime) -> float {
        if (b == 0.0f && a_prime == 0.0f) return 0.0f;
        float result = det_atan2f(b, a_prime) * (float)(180.0 / 3.14159265358979323846);
        return result < 0.0f ? result + 360.0f : result;
    };
    float h_one_prime = calc_h_prime(c1.b, a_one_prime);
    float h_two_prime = calc_h_prime(c2.b, a_two_prime);
    float h_prime_diff = h_two_prime - h_one_prime;
    float h_prime_abs_diff = std::fabs(h_prime_diff);
    bool zero_chroma = (c_one_prime == 0.0f) || (c_two_prime == 0.0f);
    float delta_h_prime;
    if (zero_chroma) delta_h_prime = 0.0f;
    else if (h_prime_abs_diff <= 180.0f) delta_h_prime = h_prime_diff;
    else if (h_two_prime <= h_one_prime) delta_h_prime = h_prime_diff + 360.0f;
    else delta_h_prime = h_prime_diff - 360.0f;
    float delta_big_h_prime = 2.0f * std::sqrt(c_one_prime * c_two_prime) * det_sinf(delta_h_prime / 2.0f * pi_over_180);
    float h_prime_sum = h_one_prime + h_two_prime;
    float h_bar_prime;
    if (zero_chroma) h_bar_prime = h_prime_sum;
    else if (h_prime_abs_diff > 180.0f) h_bar_prime = (h_prime_sum + 360.0f) / 2.0f;
    else h_bar_prime = h_prime_sum / 2.0f;
    float l_bar = (c1.l + c2.l) / 2.0f;
    float c_bar_prime = (c_one_prime + c_two_prime) / 2.0f;
    float t = 1.0f - 0.17f * det_cosf((h_bar_prime - 30.0f) * pi_over_180)
              + 0.24f * det_cosf((h_bar_prime * 2.0f) * pi_over_180)
              + 0.32f * det_cosf((h_bar_prime * 3.0f + 6.0f) * pi_over_180)
              - 0.20f * det_cosf((h_bar_prime * 4.0f - 63.0f) * pi_over_180);
    float lm50 = l_bar - 50.0f;
    float s_l = 1.0f + ((0.015f * lm50 * lm50) / std::sqrt(lm50 * lm50 + 20.0f));
    float s_c = 1.0f + 0.045f * c_bar_prime;
    float s_h = 1.0f + 0.015f * c_bar_prime * t;
    float hb = (h_bar_prime - 275.0f) / 25.0f;
    float delta_theta = 30.0f * det_expf_neg(-(hb * hb));
    float cp2 = c_bar_prime * c_bar_prime, cp4 = cp2 * cp2;
    float c_bar_prime_pow_seven = (c_bar_prime * cp2) * cp4;
    float r_c = 2.0f * std::sqrt(c_bar_prime_pow_seven / (c_bar_prime_pow_seven + twenty_five_pow_seven));
    float r_t = -r_c * det_sinf(2.0f * delta_theta * pi_over_180);
    float delta_l_prime = c2.l - c1.l;
    float delta_c_prime = c_two_prime - c_one_prime;
    float tl = delta_l_prime / s_l, tc = delta_c_prime / s_c, th = delta_big_h_prime / s_h;
    return std::sqrt(tl * tl + tc * tc + th * th + (r_t * delta_c_prime * delta_big_h_prime) / (s_c * s_h));
}

// lib.rs:1090-1100 (the #[cached] memo only changes speed, not values)
inline double distance_cielab(Rgb8 c1, Rgb8 c2) { return (double)ciede2000(srgb8_to_lab(c1), srgb8_to_lab(c2)); }

// palette 0.7.6: Lab<D65,f64> -> Xyz -> LinSrgb -> Srgb<f64> (clamped) -> Srgb<u8>  (lib.rs:141-142, 369-371)
inline Rgb8 lab_to_srgb8(const double *lab) {
    double y = (lab[0] + PALETTE_LAB_L_OFFSET_D) / PALETTE_LAB_L_SCALE_D;
    double x = y + (lab[1] / PALETTE_LAB_A_SCALE_D);
    double z = y - (lab[2] / PALETTE_LAB_B_SCALE_D);
    const double epsilon = PALETTE_LAB_EPS_ROOT_D, kappa = PALETTE_LAB_KAPPA_INV_D, delta = PALETTE_LAB_DELTA_D;
    auto conv = [&](double c) { return c > epsilon ? c * c * c : (c - delta) * kappa; };
    double X = conv(x) * PALETTE_D65_X_D, Y = conv(y) * PALETTE_D65_Y_D, Z = conv(z) * PALETTE_D65_Z_D;
    double r = (X * PALETTE_RGB_RX) + (Y * PALETTE_RGB_RY) + (Z * PALETTE_RGB_RZ);
    double g = (X * PALETTE_RGB_GX) + (Y * PALETTE_RGB_GY) + (Z * PALETTE_RGB_GZ);
    double b = (X * PALETTE_RGB_BX) + (Y * PALETTE_RGB_BY) + (Z * PALETTE_RGB_BZ);
    auto enc = [](double v) -> uint8_t {
        double e = v <= PALETTE_SRGB_ENCODE_THRESHOLD_D ? PALETTE_SRGB_LINEAR_DIV_D * v : PALETTE_SRGB_SCALE_D * std::pow(v, 1.0 / PALETTE_SRGB_GAMMA_D) - PALETTE_SRGB_OFFSET_D;
        if (!(e >= 0.0)) e = 0.0; // clamp (NaN -> 0)
        if (e > 1.0) e = 1.0;
        return (uint8_t)std::nearbyint(e * 255.0); // from_format f64 -> u8: scale and round-to-nearest
    };
    return Rgb8{enc(r), enc(g), enc(b)};
}

// lib.rs:762-795 on `n` entries starting at `entries` (raw 5-bit triples)
inline uint32_t closest_color_index(const uint8_t *entries, uint32_t n, const double *target, bool cielab) {
    uint32_t best_index = 0;
    double best_error = 1.7976931348623157e308; // f64::MAX
    auto q = [](double v) -> uint8_t {
        double c = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v); // clamp(0,255); NaN cannot occur here
        return (uint8_t)std::round(c);                       // Rust round(): half away from zero
    };
    Rgb8 tc{q(target[0]), q(target[1]), q(target[2])};
    for (uint32_t index = 0; index < n; index++) {
        Rgb8 color = snes_as_rgba(entries + 3 * index);
        double error = cielab ? distance_cielab(color, tc) : distance_red_mean(color, tc);
        if (error < best_error) { best_error = error; best_index = index; }
    }
    return best_index;
}

// lib.rs:640-660
inline void new_nes_only(const uint8_t *rgb5, bool cielab, uint8_t *out) {
    Rgb8 color = snes_as_rgba(rgb5);
    uint8_t best[3]; nes_color(0, best);
    double best_error = 1.7976931348623157e308;
    for (uint32_t index = 0; index < NES_COLOR_COUNT; index++) {
        uint8_t nc[3]; nes_color(index, nc);
        double error = cielab ? distance_cielab(color, snes_as_rgba(nc)) : distance_red_mean(color, snes_as_rgba(nc));
        if (error < best_error) { best[0] = nc[0]; best[1] = nc[1]; best[2] = nc[2]; best_error = error; }
    }
    out[0] = best[0]; out[1] = best[1]; out[2] = best[2];
}

// Rust `f64 as u8`: saturating, NaN -> 0 (lib.rs:158-169, 388-399)
inline uint8_t rust_f64_as_u8(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v; // truncation toward zero
}

// ---------------------------------------------------------------------------------------------
// cogset 0.2.0 Kmeans::new(data, k) restated: Lloyd's algorithm on Euclid<[f64;3]>,
// initial centres = first k points, squared-Euclidean assignment with first-min ties, centre = sum
// scaled by 1/count, stop when |delta objective| < 1e-6 or after 100 iterations.
// ---------------------------------------------------------------------------------------------
const double KMEANS_TOL = 1e-6;
const uint32_t KMEANS_MAX_ITER = 100;

struct KmeansResult { std::vector<double> centres; std::vector<uint32_t> assign; uint32_t iterations; bool ok; };

void kmeans_update_assignments(const double *pts, uint32_t n, uint32_t k, const std::vector<double> &centres,
                               std::vector<uint32_t> &assign, std::vector<uint32_t> &counts, std::vector<double> &costs) {
    std::fill(counts.begin(), counts.end(), 0u);
    for (uint32_t p = 0; p < n; p++) {
        double min_dist = INFINITY; uint32_t index = 0;
        for (uint32_t i = 0; i < k; i++) {
            double d0 = pts[3 * p] - centres[3 * i], d1 = pts[3 * p + 1] - centres[3 * i + 1], d2 = pts[3 * p + 2] - centres[3 * i + 2];
            double dist = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
            if (dist < min_dist) { min_dist = dist; index = i; }
        }
        costs[p] = min_dist; assign[p] = index; counts[index] += 1;
    }
}
KmeansResult kmeans(const double *pts, uint32_t n, uint32_t k) {
    KmeansResult R; R.ok = false; R.iterations = 0;
    if (!(2 <= k && k < n)) return R; // cogset asserts 2 <= k < n (panics in the reference: quirk Q4)
    R.ok = true;
    R.centres.assign(pts, pts + 3 * k);
    R.assign.assign(n, 0xffffffffu);
    std::vector<uint32_t> counts(k, 0);
    std::vector<double> costs(n, 0.0);
    kmeans_update_assignments(pts, n, k, R.centres, R.assign, counts, costs);
    double objective = 0.0; for (uint32_t p = 0; p < n; p++) objective = objective + costs[p];
    uint32_t iterations = 0;
    while (iterations < KMEANS_MAX_ITER) {
        // update_centres: zero, add members in data order, scale by 1/count
        std::fill(R.centres.begin(), R.centres.end(), 0.0);
        for (uint32_t p = 0; p < n; p++) {
            uint32_t a = R.assign[p];
            R.centres[3 * a] += pts[3 * p]; R.centres[3 * a + 1] += pts[3 * p + 1]; R.centres[3 * a + 2] += pts[3 * p + 2];
        }
        for (uint32_t i = 0; i < k; i++) {
            double s = 1.0 / (double)counts[i]; // count 0 -> inf -> NaN centre (never re-assigned)
            R.centres[3 * i] *= s; R.centres[3 * i + 1] *= s; R.centres[3 * i + 2] *= s;
        }
        kmeans_update_assignments(pts, n, k, R.centres, R.assign, counts, costs);
        double new_objective = 0.0; for (uint32_t p = 0; p < n; p++) new_objective = new_objective + costs[p];
        if (std::fabs(new_objective - objective) < KMEANS_TOL) break;
        objective = new_objective;
        iterations++;
    }
    R.iterations = iterations;
    return R;
}

// ---------------------------------------------------------------------------------------------
// ssimulacra2 0.5.1 (+ yuvxyb 0.4.2, yuvxyb-math 0.1.1) restated
// ---------------------------------------------------------------------------------------------
// Recursive Gaussian, sigma = 1.5 (ssimulacra2 build.rs; libjxl's Charalampidis truncated-cosine filter)
struct BlurConsts {
    int radius;
    float mul_in[3], mul_prev[3], mul_prev2[3]; // horizontal: n2, -d1, -1
    float vert_mul_in[3], vert_mul_prev[3];     // vertical: n2, d1
    double fir[9];                              // equivalent zero-padded FIR taps h[-4..4] (binary64)
    BlurConsts() {
        const double SIGMA = SSIM2_BLUR_SIGMA, PI = 3.14159265358979323846;
        double radius_d = std::round(std::fma(SSIM2_BLUR_RADIUS_A, SIGMA, SSIM2_BLUR_RADIUS_B)); // (57), N = 5
        radius = (int)radius_d;
        double pi_div_2r = PI / (2.0 * radius_d);
        double omega[3] = {pi_div_2r, 3.0 * pi_div_2r, 5.0 * pi_div_2r};
        double p_1 = 1.0 / std::tan(0.5 * omega[0]);
        double p_3 = -1.0 / std::tan(0.5 * omega[1]);
        double p_5 = 1.0 / std::tan(0.5 * omega[2]);
        double r_1 = p_1 * p_1 / std::sin(omega[0]);
        double r_3 = -p_3 * p_3 / std::sin(omega[1]);
        double r_5 = p_5 * p_5 / std::sin(omega[2]);
        double neg_half_sigma2 = -0.5 * SIGMA * SIGMA;
        double recip_radius = 1.0 / radius_d;
        double rho[3];
        for (int i = 0; i < 3; i++) rho[i] = std::exp(neg_half_sigma2 * omega[i] * omega[i]) * recip_radius;
        double d_13 = std::fma(p_1, r_3, -r_1 * p_3);
        double d_35 = std::fma(p_3, r_5, -r_3 * p_5);
        double d_51 = std::fma(p_5, r_1, -r_5 * p_1);
        double recip_d13 = 1.0 / d_13;
        double zeta_15 = d_35 * recip_d13;
        double zeta_35 = d_51 * recip_d13;
        // (56): A = [[p1,p3,p5],[r1,r3,r5],[z15,z35,1]]; beta = A^-1 * gamma
        double A[3][3] = {{p_1, p_3, p_5}, {r_1, r_3, r_5}, {zeta_15, zeta_35, 1.0}};
        double gamma[3] = {1.0, std::fma(radius_d, radius_d, -SIGMA * SIGMA), std::fma(zeta_15, rho[0], zeta_35 * rho[1]) + rho[2]};
        double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        double inv[3][3];
        inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
        inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det; inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
        inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
        inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
        inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
        double beta[3];
        for (int i = 0; i < 3; i++) beta[i] = inv[i][0] * gamma[0] + inv[i][1] * gamma[1] + inv[i][2] * gamma[2];
        double n2[3], d1[3];
        for (int i = 0; i < 3; i++) {
            n2[i] = -beta[i] * std::cos(omega[i] * (radius_d + 1.0)); // (33)
            d1[i] = -2.0 * std::cos(omega[i]);
            mul_in[i] = (float)n2[i]; mul_prev[i] = (float)(-d1[i]); mul_prev2[i] = -1.0f;
            vert_mul_in[i] = (float)n2[i]; vert_mul_prev[i] = (float)d1[i];
        }
        // Impulse response of the (binary64) recurrence = the FIR it implements: support [-4,4].
        for (int j = 0; j < 9; j++) fir[j] = 0.0;
        double prev[3] = {0, 0, 0}, prev2[3] = {0, 0, 0};
        const int m = 16; // impulse position
        for (int n = m - 12; n <= m + 12; n++) {
            int left = n - radius - 1, right = n + radius - 1;
            double sum = (left == m ? 1.0 : 0.0) + (right == m ? 1.0 : 0.0);
            double tot = 0.0;
            for (int i = 0; i < 3; i++) {
                double o = sum * (double)mul_in[i] - (double)vert_mul_prev[i] * prev[i] - prev2[i];
                prev2[i] = prev[i]; prev[i] = o; tot += o;
            }
            if (n - m >= -4 && n - m <= 4) fir[n - m + 4] = tot;
        }
    }
};
const BlurConsts &blur_consts() { static BlurConsts c; return c; }

// ssimulacra2 blur/gaussian.rs horizontal_row
void blur_horizontal_row(const float *input, float *output, int width) {
    const BlurConsts &C = blur_consts();
    const int big_n = C.radius;
    float prev_1 = 0, prev_3 = 0, prev_5 = 0, prev2_1 = 0, prev2_3 = 0, prev2_5 = 0;
    for (int n = -big_n + 1; n < width; n++) {
        int left = n - big_n - 1, right = n + big_n - 1;
        float left_val = left >= 0 ? input[left] : 0.0f;
        float right_val = right < width ? input[right] : 0.0f;
        float sum = left_val + right_val;
        float out_1 = sum * C.mul_in[0], out_3 = sum * C.mul_in[1], out_5 = sum * C.mul_in[2];
        out_1 = fmaf(C.mul_prev2[0], prev2_1, out_1);
        out_3 = fmaf(C.mul_prev2[1], prev2_3, out_3);
        out_5 = fmaf(C.mul_prev2[2], prev2_5, out_5);
        prev2_1 = prev_1; prev2_3 = prev_3; prev2_5 = prev_5;
        out_1 = fmaf(C.mul_prev[0], prev_1, out_1);
        out_3 = fmaf(C.mul_prev[1], prev_3, out_3);
        out_5 = fmaf(C.mul_prev[2], prev_5, out_5);
        prev_1 = out_1; prev_3 = out_3; prev_5 = out_5;
        if (n >= 0) output[n] = out_1 + out_3 + out_5;
    }
}
// ssimulacra2 blur/gaussian.rs vertical_pass (column-chunking does not change per-column arithmetic)
void blur_vertical(const float *input, float *output, int width, int height) {
    const BlurConsts &C = blur_consts();
    const int big_n = C.radius;
    std::vector<float> prev(3 * width, 0.0f), prev2(3 * width, 0.0f), out(3 * width, 0.0f);
    for (int n = -big_n + 1; n < height; n++) {
        int top = n - big_n - 1, bottom = n + big_n - 1;
        const float *top_row = top >= 0 ? input + (size_t)top * width : nullptr;
        const float *bottom_row = bottom < height ? input + (size_t)bottom * width : nullptr;
        for (int i = 0; i < width; i++) {
            float sum = (top_row ? top_row[i] : 0.0f) + (bottom_row ? bottom_row[i] : 0.0f);
            int i1 = i, i3 = i + width, i5 = i + 2 * width;
            float o1 = fmaf(prev[i1], C.vert_mul_prev[0], prev2[i1]);
            float o3 = fmaf(prev[i3], C.vert_mul_prev[1], prev2[i3]);
            float o5 = fmaf(prev[i5], C.vert_mul_prev[2], prev2[i5]);
            o1 = fmaf(sum, C.vert_mul_in[0], -o1);
            o3 = fmaf(sum, C.vert_mul_in[1], -o3);
            o5 = fmaf(sum, C.vert_mul_in[2], -o5);
            out[i1] = o1; out[i3] = o3; out[i5] = o5;
            if (n >= 0) output[(size_t)n * width + i] = o1 + o3 + o5;
        }
        prev2.swap(prev); // prev2 <- prev
        prev.swap(out);   // prev  <- out (old prev2 storage becomes scratch)
    }
}
// Test-only equivalent: zero-padded separable 9-tap FIR (taps rounded to f32, plain mul/add)
void blur_fir(const float *in, float *out, int width, int height) {
    const BlurConsts &C = blur_consts();
    float h[9]; for (int j = 0; j < 9; j++) h[j] = (float)C.fir[j];
    std::vector<float> tmp((size_t)width * height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float s = 0.0f;
            for (int j = -4; j <= 4; j++) { int xx = x + j; if (xx >= 0 && xx < width) s = fmaf(h[j + 4], in[(size_t)y * width + xx], s); }
            tmp[(size_t)y * width + x] = s;
        }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float s = 0.0f;
            for (int j = -4; j <= 4; j++) { int yy = y + j; if (yy >= 0 && yy < height) s = fmaf(h[j + 4], tmp[(size_t)yy * width + x], s); }
            out[(size_t)y * width + x] = s;
        }
}
enum { MODE_FIR = 1, MODE_ZIMG_SRGB = 2, MODE_FAST_MATH = 4, MODE_PERTURB = 8 }; // bits of the oracle's `mode` (see the what-if variants below)
void blur_plane(const float *in, float *out, int width, int height, int mode) {
    if (mode & MODE_FIR) { blur_fir(in, out, width, height); return; }
    std::vector<float> temp((size_t)width * height);
    for (int y = 0; y < height; y++) blur_horizontal_row(in + (size_t)y * width, temp.data() + (size_t)y * width, width);
    blur_vertical(temp.data(), out, width, height);
}

// ---- what-if variants of the unpinned third-party arithmetic (oracle only: DESIGN.md section 2, profiles/r4_exposure.py) ----
// The `mode` integer that selects the blur (bit 0: the FIR stand-in of blur_fir) also carries, in its upper bits, variants
// of yuvxyb's transfer function and of its powf / cbrtf, so that the distance between this restatement and what upstream
// may compute can be MEASURED on the error() the optimizer sees.  Mode 0 is the restatement the product is held to.
// MODE_FAST_MATH: powf and cbrtf evaluated in binary32 the way a fast-math crate would (yuvxyb-math is said to approximate
// both): powf(x, y) = exp2f(y * log2f(x)) — the rounding of log2f is multiplied by y |log2 x|: ~1e-6 relative for x ~ 0.01 —
// and cbrtf by a bit-hack seed with two Halley steps in binary32 (~2e-7 relative).
inline float fast_powf(float x, float y) { return x > 0.0f ? exp2f(y * log2f(x)) : 0.0f; }
inline float fast_cbrtf(float x) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t b; std::memcpy(&b, &x, 4);
    b = b / 3u + 709958130u;
    float t; std::memcpy(&t, &b, 4);
    for (int i = 0; i < 2; i++) { const float t3 = t * t * t; t = t * ((t3 + 2.0f * x) / (2.0f * t3 + x)); }
    return t;
}
// MODE_PERTURB: the exact result moved by a pseudo-random relative amount in [-1e-6, 1e-6] (a hash of its bits): "an
// approximation good to 1e-6 whose error has no structure"
inline float perturb_1e6(float v) {
    uint32_t b; std::memcpy(&b, &v, 4);
    uint64_t z = (uint64_t)b * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    const double u = (double)(z & 0xfffffu) / 524287.5 - 1.0; // [-1, 1]
    return (float)((double)v * (1.0 + 1e-6 * u));
}
inline float mode_powf(float x, float y, int mode) {
    float r = (mode & MODE_FAST_MATH) ? fast_powf(x, y) : (float)std::pow((double)x, (double)y); // (exact: binary64 pow, rounded once)
    return (mode & MODE_PERTURB) ? perturb_1e6(r) : r;
}
inline float mode_cbrtf(float x, int mode) {
    float r = (mode & MODE_FAST_MATH) ? fast_cbrtf(x) : det_cbrtf(x);
    return (mode & MODE_PERTURB) ? perturb_1e6(r) : r;
}

// yuvxyb 0.4.2: sRGB transfer (TransferCharacteristic::SRGB) on f32 in [0,1]; BT.709 primaries -> no-op.
// Constants: include/ssimulacra2_constants.h (SSIM2_SRGB_*; MODE_ZIMG_SRGB: the zimg-style pair alpha / beta).
struct EotfLut {
    float lin[256];
    explicit EotfLut(int mode) {
        for (int v = 0; v < 256; v++) {
            float x = (float)v / 255.0f; // lib.rs:511-513, 531-533
            if (mode & MODE_ZIMG_SRGB) {
                x = x < 0.0f ? 0.0f : x;
                if (x < SSIM2_SRGB_LINEAR_DIV * SSIM2_ZIMG_SRGB_BETA) lin[v] = x / SSIM2_SRGB_LINEAR_DIV;
                else lin[v] = mode_powf((x + (SSIM2_ZIMG_SRGB_ALPHA - 1.0f)) / SSIM2_ZIMG_SRGB_ALPHA, SSIM2_SRGB_GAMMA, mode);
            } else if (x < SSIM2_SRGB_THRESHOLD) lin[v] = x / SSIM2_SRGB_LINEAR_DIV;
            else lin[v] = mode_powf((x + SSIM2_SRGB_OFFSET) / SSIM2_SRGB_SCALE, SSIM2_SRGB_GAMMA, mode);
        }
    }
};
const EotfLut &eotf_lut(int mode) { // one table per variant (bits 1..3 of the mode)
    static const EotfLut l[8] = {EotfLut(0), EotfLut(2), EotfLut(4), EotfLut(6), EotfLut(8), EotfLut(10), EotfLut(12), EotfLut(14)};
    return l[(mode >> 1) & 7];
}

struct Img3 { int w = 0, h = 0; std::vector<float> p[3]; void resize(int W, int H) { w = W; h = H; for (auto &v : p) v.assign((size_t)W * H, 0.0f); } };

// ssimulacra2 downscale_by_2 (linear RGB, edge-clamped 2x2 box)
void downscale_by_2(const Img3 &in, Img3 &out) {
    int ow = (in.w + 1) / 2, oh = (in.h + 1) / 2;
    out.resize(ow, oh);
    for (int c = 0; c < 3; c++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++) {
                float sum = 0.0f;
                for (int iy = 0; iy < 2; iy++)
                    for (int ix = 0; ix < 2; ix++) {
                        int x = std::min(ox * 2 + ix, in.w - 1), y = std::min(oy * 2 + iy, in.h - 1);
                        sum += in.p[c][(size_t)y * in.w + x];
                    }
                out.p[c][(size_t)oy * ow + ox] = sum * 0.25f;
            }
}
// yuvxyb linear_rgb_to_xyb + ssimulacra2 make_positive_xyb, planar output
inline void linear_rgb_to_positive_xyb(float r, float g, float b, float *X, float *Y, float *B, int mode = 0) {
    const float K_M02 = SSIM2_OPSIN_M02, K_M00 = SSIM2_OPSIN_M00, K_M01 = 1.0f - K_M02 - K_M00;
    const float K_M12 = SSIM2_OPSIN_M12, K_M10 = SSIM2_OPSIN_M10, K_M11 = 1.0f - K_M12 - K_M10;
    const float K_M20 = SSIM2_OPSIN_M20, K_M21 = SSIM2_OPSIN_M21, K_M22 = 1.0f - K_M20 - K_M21;
    const float K_B0 = SSIM2_OPSIN_BIAS;
    const float K_B0_ROOT = SSIM2_OPSIN_BIAS_CBRT;
    float m0 = fmaf(K_M00, r, fmaf(K_M01, g, fmaf(K_M02, b, K_B0)));
    float m1 = fmaf(K_M10, r, fmaf(K_M11, g, fmaf(K_M12, b, K_B0)));
    float m2 = fmaf(K_M20, r, fmaf(K_M21, g, fmaf(K_M22, b, K_B0)));
    if (m0 < 0.0f) m0 = 0.0f;
    if (m1 < 0.0f) m1 = 0.0f;
    if (m2 < 0.0f) m2 = 0.0f;
    if (mode & (MODE_FAST_MATH | MODE_PERTURB)) { m0 = mode_cbrtf(m0, mode) - K_B0_ROOT; m1 = mode_cbrtf(m1, mode) - K_B0_ROOT; m2 = mode_cbrtf(m2, mode) - K_B0_ROOT; }
    else { m0 = det_cbrtf(m0) - K_B0_ROOT; m1 = det_cbrtf(m1) - K_B0_ROOT; m2 = det_cbrtf(m2) - K_B0_ROOT; }
    float x = 0.5f * (m0 - m1), y = 0.5f * (m0 + m1), bb = m2;
    // make_positive_xyb
    *B = (bb - y) + SSIM2_POS_B_OFFSET;
    *X = fmaf(x, SSIM2_POS_X_SCALE, SSIM2_POS_X_OFFSET);
    *Y = y + SSIM2_POS_Y_OFFSET;
}
void to_positive_xyb(const Img3 &lin, Img3 &xyb, int mode = 0) {
    xyb.resize(lin.w, lin.h);
    size_t n = (size_t)lin.w * lin.h;
    for (size_t i = 0; i < n; i++) linear_rgb_to_positive_xyb(lin.p[0][i], lin.p[1][i], lin.p[2][i], &xyb.p[0][i], &xyb.p[1][i], &xyb.p[2][i], mode);
}

const double SSIM2_WEIGHT[108] = SSIM2_WEIGHTS; // include/ssimulacra2_constants.h
static_assert(sizeof(SSIM2_WEIGHT) / sizeof(double) == 108, "weight table must hold 108 entries");

struct ScaleStats { double avg_ssim[6]; double avg_edgediff[12]; };

// Source-side terms of one scale (depend only on `original`)
struct SrcScale { int w, h; Img3 img1; Img3 mu1; Img3 sigma1_sq; };
struct SrcPyramid { std::vector<SrcScale> scales; };

void build_src_pyramid(const Img3 &lin0, int blur_mode, SrcPyramid &P) {
    P.scales.clear();
    Img3 lin = lin0;
    int width = lin.w, height = lin.h;
    std::vector<float> mul;
    for (int scale = 0; scale < 6; scale++) {
        if (width < 8 || height < 8) break;
        if (scale > 0) { Img3 d; downscale_by_2(lin, d); lin = d; width = lin.w; height = lin.h; }
        SrcScale S; S.w = width; S.h = height;
        to_positive_xyb(lin, S.img1, blur_mode);
        S.mu1.resize(width, height); S.sigma1_sq.resize(width, height);
        mul.resize((size_t)width * height);
        for (int c = 0; c < 3; c++) {
            for (size_t i = 0; i < mul.size(); i++) mul[i] = S.img1.p[c][i] * S.img1.p[c][i];
            blur_plane(mul.data(), S.sigma1_sq.p[c].data(), width, height, blur_mode);
            blur_plane(S.img1.p[c].data(), S.mu1.p[c].data(), width, height, blur_mode);
        }
        P.scales.push_back(std::move(S));
    }
}

// ssim_map + edge_diff_map of one scale
void scale_maps(const SrcScale &S, const Img3 &img2, const Img3 &mu2, const Img3 &s22, const Img3 &s12, ScaleStats &out) {
    const float C2 = SSIM2_C2;
    const int width = S.w, height = S.h;
    const double one_per_pixels = 1.0 / (double)((size_t)width * height);
    for (int c = 0; c < 3; c++) {
        double sum1[2] = {0.0, 0.0};
        double sum2[4] = {0.0, 0.0, 0.0, 0.0};
        const float *m1 = S.mu1.p[c].data(), *m2 = mu2.p[c].data(), *s11 = S.sigma1_sq.p[c].data(), *p22 = s22.p[c].data(), *p12 = s12.p[c].data();
        const float *i1 = S.img1.p[c].data(), *i2 = img2.p[c].data();
        size_t n = (size_t)width * height;
        for (size_t x = 0; x < n; x++) {
            float mu1 = m1[x], mu2v = m2[x];
            float mu11 = mu1 * mu1, mu22 = mu2v * mu2v, mu12 = mu1 * mu2v;
            float mu_diff = mu1 - mu2v;
            float num_m = fmaf(mu_diff, -mu_diff, 1.0f);
            float num_s = fmaf(2.0f, p12[x] - mu12, C2);
            float denom_s = (s11[x] - mu11) + (p22[x] - mu22) + C2;
            double d = 1.0 - (double)((num_m * num_s) / denom_s);
            d = d > 0.0 ? d : 0.0; // f64::max(0.0): NaN -> 0.0
            sum1[0] += d;
            double d2 = d * d;
            sum1[1] += d2 * d2; // powi(4)
        }
        for (size_t x = 0; x < n; x++) {
            double d1 = (1.0 + (double)std::fabs(i2[x] - m2[x])) / (1.0 + (double)std::fabs(i1[x] - m1[x])) - 1.0;
            double artifact = d1 > 0.0 ? d1 : 0.0;
            sum2[0] += artifact;
            double a2 = artifact * artifact; sum2[1] += a2 * a2;
            double detail_lost = (-d1) > 0.0 ? (-d1) : 0.0;
            sum2[2] += detail_lost;
            double l2 = detail_lost * detail_lost; sum2[3] += l2 * l2;
        }
        out.avg_ssim[c * 2] = one_per_pixels * sum1[0];
        out.avg_ssim[c * 2 + 1] = std::sqrt(std::sqrt(one_per_pixels * sum1[1]));
        out.avg_edgediff[c * 4] = one_per_pixels * sum2[0];
        out.avg_edgediff[c * 4 + 1] = std::sqrt(std::sqrt(one_per_pixels * sum2[1]));
        out.avg_edgediff[c * 4 + 2] = one_per_pixels * sum2[2];
        out.avg_edgediff[c * 4 + 3] = std::sqrt(std::sqrt(one_per_pixels * sum2[3]));
    }
}

// Msssim::score
double msssim_score(const std::vector<ScaleStats> &scales) {
    double ssim = 0.0;
    size_t i = 0;
    for (int c = 0; c < 3; c++)
        for (const ScaleStats &scale : scales)
            for (int n = 0; n < 2; n++) {
                ssim = std::fma(SSIM2_WEIGHT[i], std::fabs(scale.avg_ssim[c * 2 + n]), ssim); i++;
                ssim = std::fma(SSIM2_WEIGHT[i], std::fabs(scale.avg_edgediff[c * 4 + n]), ssim); i++;
                ssim = std::fma(SSIM2_WEIGHT[i], std::fabs(scale.avg_edgediff[c * 4 + n + 2]), ssim); i++;
            }
    ssim *= SSIM2_SCORE_SCALE;
    ssim = std::fma(SSIM2_SCORE_C3 * ssim * ssim, ssim, std::fma(SSIM2_SCORE_C1, ssim, SSIM2_SCORE_C2 * ssim * ssim));
    if (ssim > 0.0) ssim = std::fma(std::pow(ssim, SSIM2_SCORE_EXP), SSIM2_SCORE_GAIN, SSIM2_SCORE_MAX);
    else ssim = SSIM2_SCORE_MAX;
    return ssim;
}

// compute_frame_ssimulacra2 with the source side supplied (identical arithmetic to recomputing it)
double ssimulacra2_against(const SrcPyramid &P, const Img3 &dst_lin0, int blur_mode) {
    std::vector<ScaleStats> stats;
    Img3 lin = dst_lin0;
    Img3 img2, mu2, s22, s12;
    std::vector<float> mul;
    for (size_t scale = 0; scale < P.scales.size(); scale++) {
        const SrcScale &S = P.scales[scale];
        if (scale > 0) { Img3 d; downscale_by_2(lin, d); lin = d; }
        to_positive_xyb(lin, img2, blur_mode);
        int width = S.w, height = S.h;
        mu2.resize(width, height); s22.resize(width, height); s12.resize(width, height);
        mul.resize((size_t)width * height);
        for (int c = 0; c < 3; c++) {
            for (size_t i = 0; i < mul.size(); i++) mul[i] = img2.p[c][i] * img2.p[c][i];
            blur_plane(mul.data(), s22.p[c].data(), width, height, blur_mode);
            for (size_t i = 0; i < mul.size(); i++) mul[i] = S.img1.p[c][i] * img2.p[c][i];
            blur_plane(mul.data(), s12.p[c].data(), width, height, blur_mode);
            blur_plane(img2.p[c].data(), mu2.p[c].data(), width, height, blur_mode);
        }
        ScaleStats st; scale_maps(S, img2, mu2, s22, s12, st);
        stats.push_back(st);
    }
    return msssim_score(stats);
}

void rgba_to_linear(const uint8_t *rgba, int w, int h, Img3 &out, int mode = 0) {
    const EotfLut &L = eotf_lut(mode);
    out.resize(w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) { out.p[0][i] = L.lin[rgba[4 * i]]; out.p[1][i] = L.lin[rgba[4 * i + 1]]; out.p[2][i] = L.lin[rgba[4 * i + 2]]; }
}

// splitmix64
inline uint64_t splitmix64_next(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

} // namespace

// ---------------------------------------------------------------------------------------------
// OptimizedImage (lib.rs:33-626)
// ---------------------------------------------------------------------------------------------
struct oracle_ctx {
    uint32_t width, height, sub_count, sub_size;
    bool dither, perceptual, nes;
    std::vector<uint8_t> original;      // RGBA8 row-major (lib.rs:36,57)
    std::vector<uint8_t> tile_palettes; // 32*32 (lib.rs:58)
    std::vector<uint8_t> colors;        // sub_count*sub_size raw 5-bit triples (lib.rs:59,756)
    std::vector<uint8_t> palette_map;   // w*h (lib.rs:60)
    int cache_source = 0, blur_mode = 0;
    bool src_valid = false; int src_mode = -1;
    SrcPyramid src;
    uint32_t width_in_tiles() const { return width / 8; }
    uint32_t height_in_tiles() const { return height / 8; }
    const uint8_t *px(uint32_t x, uint32_t y) const { return &original[4 * ((size_t)y * width + x)]; }
};

namespace {

// lib.rs:417-423
inline uint32_t get_palette_index(const oracle_ctx *c, uint32_t x, uint32_t y) { return c->tile_palettes[(x / 8) + (y / 8) * c->width_in_tiles()]; }

// lib.rs:425-501
void optimize(oracle_ctx *c) {
    const double w0 = c->dither ? 7.0 / 16.0 : 0.0, w1 = c->dither ? 3.0 / 16.0 : 0.0, w2 = c->dither ? 5.0 / 16.0 : 0.0, w3 = c->dither ? 1.0 / 16.0 : 0.0;
    const double error_multiplier = 0.8;
    const uint32_t W = c->width, H = c->height;
    std::vector<double> error((size_t)W * H * 3, 0.0);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            size_t pixel_index = (size_t)y * W + x;
            const uint8_t *oc = c->px(x, y);
            uint32_t palette = get_palette_index(c, x, y);
            double target[3] = {(double)oc[0] + error[3 * pixel_index], (double)oc[1] + error[3 * pixel_index + 1], (double)oc[2] + error[3 * pixel_index + 2]};
            uint32_t color_index = closest_color_index(&c->colors[3 * (size_t)palette * c->sub_size], c->sub_size, target, c->perceptual);
            c->palette_map[pixel_index] = oc[3] > 0 ? (uint8_t)color_index : 0;
            Rgb8 nc = snes_as_rgba(&c->colors[3 * ((size_t)palette * c->sub_size + color_index)]);
            double pixel_error[3];
            if (oc[3] > 0) { pixel_error[0] = target[0] - (double)nc.r; pixel_error[1] = target[1] - (double)nc.g; pixel_error[2] = target[2] - (double)nc.b; }
            else { pixel_error[0] = error[3 * pixel_index]; pixel_error[1] = error[3 * pixel_index + 1]; pixel_error[2] = error[3 * pixel_index + 2]; }
            for (int i = 0; i < 3; i++) {
                double value = pixel_error[i];
                if (x + 1 < W) error[3 * (pixel_index + 1) + i] += value * error_multiplier * w0;
                if (y + 1 < H) {
                    if (x > 0) error[3 * (pixel_index + W - 1) + i] += value * error_multiplier * w1;
                    error[3 * (pixel_index + W) + i] += value * error_multiplier * w2;
                    if (x + 1 < W) error[3 * (pixel_index + W + 1) + i] += value * error_multiplier * w3;
                }
            }
        }
}

// lib.rs:550-577
void as_rgba(const oracle_ctx *c, uint8_t *image) {
    std::memset(image, 0, (size_t)c->width * c->height * 4);
    for (uint32_t y = 0; y < c->height; y++)
        for (uint32_t x = 0; x < c->width; x++) {
            uint32_t palette_index = c->tile_palettes[(y / 8) * 32 + (x / 8)];
            size_t color_index = (size_t)palette_index * c->sub_size + c->palette_map[(size_t)y * c->width + x];
            if (c->px(x, y)[3] > 0) {
                Rgb8 v = snes_as_rgba(&c->colors[3 * color_index]);
                uint8_t *o = image + 4 * ((size_t)y * c->width + x);
                o[0] = v.r; o[1] = v.g; o[2] = v.b; o[3] = 255;
            }
        }
}

// lib.rs:503-548
double error_of(oracle_ctx *c) {
    std::vector<uint8_t> rgba((size_t)c->width * c->height * 4);
    as_rgba(c, rgba.data());
    if (!(c->cache_source && c->src_valid && c->src_mode == c->blur_mode)) {
        Img3 src_lin; rgba_to_linear(c->original.data(), (int)c->width, (int)c->height, src_lin, c->blur_mode);
        build_src_pyramid(src_lin, c->blur_mode, c->src);
        c->src_valid = true; c->src_mode = c->blur_mode;
    }
    Img3 dst_lin; rgba_to_linear(rgba.data(), (int)c->width, (int)c->height, dst_lin, c->blur_mode);
    return 100.0 - ssimulacra2_against(c->src, dst_lin, c->blur_mode);
}

// centre -> SnesColor (lib.rs:140-171 and 369-401, same expressions)
void centre_to_color(const oracle_ctx *c, const double *centre, uint8_t *out) {
    if (c->perceptual) {
        Rgb8 rgb = lab_to_srgb8(centre);
        uint8_t raw[3] = {(uint8_t)(rgb.r / 8), (uint8_t)(rgb.g / 8), (uint8_t)(rgb.b / 8)};
        if (c->nes) new_nes_only(raw, true, out); else { out[0] = raw[0]; out[1] = raw[1]; out[2] = raw[2]; }
    } else {
        uint8_t raw[3] = {rust_f64_as_u8(std::round(centre[0] / 8.0)), rust_f64_as_u8(std::round(centre[1] / 8.0)), rust_f64_as_u8(std::round(centre[2] / 8.0))};
        if (c->nes) new_nes_only(raw, false, out); else { out[0] = raw[0]; out[1] = raw[1]; out[2] = raw[2]; }
    }
}

// lib.rs:330-405.  Returns false when cogset's precondition fails (the reference would panic: Q4).
bool recalculate_palette(oracle_ctx *c, uint32_t palette) {
    std::vector<double> pixels;
    for (uint32_t tile = 0; tile < c->tile_palettes.size(); tile++) {
        if (c->tile_palettes[tile] != palette) continue;
        uint32_t tile_x = tile % c->width_in_tiles(), tile_y = tile / c->width_in_tiles();
        if (tile_y >= c->height_in_tiles()) continue; // tiles beyond a 256xN image hold no pixels (Q1 fence)
        for (uint32_t x = 0; x < 8; x++)
            for (uint32_t y = 0; y < 8; y++) {
                const uint8_t *col = c->px(tile_x * 8 + x, tile_y * 8 + y);
                if (col[3] > 0) {
                    if (c->perceptual) { Lab32 lab = srgb8_to_lab(Rgb8{col[0], col[1], col[2]}); pixels.push_back((double)lab.l); pixels.push_back((double)lab.a); pixels.push_back((double)lab.b); }
                    else { pixels.push_back((double)col[0]); pixels.push_back((double)col[1]); pixels.push_back((double)col[2]); }
                }
            }
    }
    KmeansResult km = kmeans(pixels.data(), (uint32_t)(pixels.size() / 3), c->sub_size);
    if (!km.ok) return false;
    for (uint32_t index = 0; index < c->sub_size; index++) centre_to_color(c, &km.centres[3 * index], &c->colors[3 * ((size_t)palette * c->sub_size + index)]);
    return true;
}

} // namespace

extern "C" {

const char *oracle_last_error(void) { return g_last_error.c_str(); }

oracle_ctx *oracle_create(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t sub_count, uint32_t sub_size, uint32_t flags) {
    if (!rgba) { fail("rgba is null"); return nullptr; }
    if (w != 256 || h < 8 || h > 256 || (h % 8) != 0) { fail("image must be 256 wide and 8..256 (multiple of 8) high"); return nullptr; }
    if (sub_count < 1 || sub_size < 1 || sub_count > 255 || sub_size > 255 || sub_count * sub_size > 4096) { fail("bad subpalette count/size"); return nullptr; }
    oracle_ctx *c = new oracle_ctx();
    c->width = w; c->height = h; c->sub_count = sub_count; c->sub_size = sub_size;
    c->dither = flags & ORACLE_DITHER; c->perceptual = flags & ORACLE_PERCEPTUAL; c->nes = flags & ORACLE_NES;
    c->original.assign(rgba, rgba + (size_t)w * h * 4);
    c->tile_palettes.assign(32 * 32, 0);
    c->colors.assign((size_t)sub_count * sub_size * 3, 0);
    c->palette_map.assign((size_t)w * h, 0);
    return c;
}
void oracle_destroy(oracle_ctx *c) { delete c; }
void oracle_set_cache_source(oracle_ctx *c, int on) { c->cache_source = on; }
void oracle_set_blur_mode(oracle_ctx *c, int mode) { c->blur_mode = (c->blur_mode & ~1) | (mode & 1); }
void oracle_set_variant(oracle_ctx *c, int bits) { c->blur_mode = (c->blur_mode & 1) | ((bits & 7) << 1); }

int oracle_optimize(oracle_ctx *c) { optimize(c); return 0; }

// Dynamic tile -> subpalette reassignment: NOT in the reference — /root/reference/TODO.md:36-37 lists it as missing ("no
// attempt is made to reassign tiles dynamically ... The initial guess is probably not optimal").  Definition used by the
// oracle and the product alike: for every tile with at least one opaque pixel, cost(p) = sum over its opaque pixels, in
// raster order inside the tile (y outer, x inner), of the distance optimize() minimises (lib.rs:1080-1100) between the pixel
// and the nearest entry of subpalette p (first minimum wins, lib.rs:788-791), accumulated in binary64.  The tile moves
// to the subpalette with the strictly smallest cost, scanning p upwards from its current one's cost (ties keep the current
// subpalette, then the lower index).  Palettes are kept; optimize() re-runs if any tile moved.  Returns the tiles moved.
int oracle_reassign_tiles(oracle_ctx *c, uint32_t *moved_out) {
    uint32_t moved = 0;
    for (uint32_t ty = 0; ty < c->height_in_tiles(); ty++)
        for (uint32_t tx = 0; tx < c->width_in_tiles(); tx++) {
            std::vector<double> cost(c->sub_count, 0.0);
            bool any = false;
            for (uint32_t y = 0; y < 8; y++)
                for (uint32_t x = 0; x < 8; x++) {
                    const uint8_t *oc = c->px(tx * 8 + x, ty * 8 + y);
                    if (oc[3] == 0) continue;
                    any = true;
                    const double target[3] = {(double)oc[0], (double)oc[1], (double)oc[2]};
                    const Rgb8 tc{oc[0], oc[1], oc[2]};
                    for (uint32_t p = 0; p < c->sub_count; p++) {
                        const uint8_t *entries = &c->colors[3 * (size_t)p * c->sub_size];
                        const uint32_t j = closest_color_index(entries, c->sub_size, target, c->perceptual);
                        const Rgb8 color = snes_as_rgba(entries + 3 * j);
                        cost[p] = cost[p] + (c->perceptual ? distance_cielab(color, tc) : distance_red_mean(color, tc));
                    }
                }
            if (!any) continue;
            uint8_t &cur = c->tile_palettes[ty * c->width_in_tiles() + tx];
            uint32_t best = cur;
            for (uint32_t p = 0; p < c->sub_count; p++) if (cost[p] < cost[best]) best = p;
            if (best != cur) { cur = (uint8_t)best; moved++; }
        }
    if (moved) optimize(c);
    if (moved_out) *moved_out = moved;
    return 0;
}
int oracle_error(oracle_ctx *c, double *out) { *out = error_of(c); return 0; }

// lib.rs:79-189
int oracle_initialize_tiles(oracle_ctx *c) {
    if (c->sub_count == 1) {
        if (!recalculate_palette(c, 0)) return fail("k-means precondition 2 <= k < n violated (reference panics)");
        optimize(c);
        return 0;
    }
    std::vector<double> means; std::vector<uint32_t> map;
    for (uint32_t tile_x = 0; tile_x < c->width_in_tiles(); tile_x++)
        for (uint32_t tile_y = 0; tile_y < c->height_in_tiles(); tile_y++) {
            float sum[3] = {0.0f, 0.0f, 0.0f}; int32_t count = 0;
            uint32_t index = tile_y * c->width_in_tiles() + tile_x;
            for (uint32_t x = 0; x < 8; x++)
                for (uint32_t y = 0; y < 8; y++) {
                    const uint8_t *col = c->px(tile_x * 8 + x, tile_y * 8 + y);
                    if (col[3] > 0) {
                        if (c->perceptual) { Lab32 lab = srgb8_to_lab(Rgb8{col[0], col[1], col[2]}); sum[0] += lab.l; sum[1] += lab.a; sum[2] += lab.b; }
                        else { sum[0] += (float)col[0]; sum[1] += (float)col[1]; sum[2] += (float)col[2]; }
                        count += 1;
                    }
                }
            if (sum[0] + sum[1] + sum[2] > 0.0f) {
                means.push_back((double)sum[0] / (double)count); means.push_back((double)sum[1] / (double)count); means.push_back((double)sum[2] / (double)count);
                map.push_back(index);
            }
        }
    KmeansResult km = kmeans(means.data(), (uint32_t)map.size(), c->sub_count);
    if (!km.ok) return fail("k-means precondition 2 <= k < n violated (reference panics)");
    for (uint32_t t = 0; t < map.size(); t++) c->tile_palettes[map[t]] = (uint8_t)km.assign[t]; // lib.rs:133-138
    for (uint32_t index = 0; index < c->sub_count; index++) {
        uint8_t color[3]; centre_to_color(c, &km.centres[3 * index], color);
        for (uint32_t i = 0; i < c->sub_size; i++) std::memcpy(&c->colors[3 * ((size_t)index * c->sub_size + i)], color, 3); // lib.rs:181-183
    }
    optimize(c);
    return 0;
}

// lib.rs:407-415
int oracle_recalculate_palettes(oracle_ctx *c) {
    for (uint32_t p = 0; p < c->sub_count; p++)
        if (!recalculate_palette(c, p)) return fail("k-means precondition 2 <= k < n violated (reference panics)");
    optimize(c);
    return 0;
}

int oracle_score_candidates(oracle_ctx *c, uint32_t palette, uint32_t index, const uint8_t *rgb5, uint32_t n, double *errors, uint8_t *maps_out) {
    if (palette >= c->sub_count || index >= c->sub_size) return fail("slot out of range");
    uint8_t *slot = &c->colors[3 * ((size_t)palette * c->sub_size + index)];
    uint8_t saved[3] = {slot[0], slot[1], slot[2]};
    std::vector<uint8_t> saved_map = c->palette_map;
    for (uint32_t k = 0; k < n; k++) {
        slot[0] = rgb5[3 * k]; slot[1] = rgb5[3 * k + 1]; slot[2] = rgb5[3 * k + 2]; // lib.rs:210-211
        optimize(c);                                                                 // lib.rs:212
        errors[k] = error_of(c);                                                     // lib.rs:214
        if (maps_out) std::memcpy(maps_out + (size_t)k * c->palette_map.size(), c->palette_map.data(), c->palette_map.size());
    }
    slot[0] = saved[0]; slot[1] = saved[1]; slot[2] = saved[2];
    c->palette_map = saved_map;
    return 0;
}

void oracle_random_candidates(uint64_t seed, uint64_t step_id, uint32_t n, uint8_t *rgb5) {
    uint64_t key = mix64(seed ^ (step_id * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull);
    for (uint32_t k = 0; k < n; k++) {
        uint64_t z = mix64(key + ((uint64_t)k + 1) * 0x9E3779B97F4A7C15ull);
        rgb5[3 * k] = (uint8_t)(z & 31); rgb5[3 * k + 1] = (uint8_t)((z >> 5) & 31); rgb5[3 * k + 2] = (uint8_t)((z >> 10) & 31);
    }
}

int oracle_step(oracle_ctx *c, uint32_t method, uint32_t palette, uint32_t index, uint32_t channel, uint64_t seed, uint64_t step_id,
                uint32_t n_random, double *best_error_out, uint8_t *best_rgb5) {
    if (palette >= c->sub_count || index >= c->sub_size || channel > 2 || method > 2) return fail("bad step arguments");
    uint8_t *slot = &c->colors[3 * ((size_t)palette * c->sub_size + index)];
    const uint8_t original_color[3] = {slot[0], slot[1], slot[2]};
    std::vector<uint8_t> cand;
    uint32_t n;
    double best_error;
    if (method == 0) { // lib.rs:191-240
        n = n_random ? n_random : 64;
        cand.resize(3 * (size_t)n);
        oracle_random_candidates(seed, step_id, n, cand.data());
        best_error = error_of(c); // lib.rs:199
    } else if (method == 1) { // lib.rs:286-328
        n = 32; cand.resize(3 * n);
        for (uint32_t v = 0; v < 32; v++) { cand[3 * v] = original_color[0]; cand[3 * v + 1] = original_color[1]; cand[3 * v + 2] = original_color[2]; cand[3 * v + channel] = (uint8_t)v; }
        best_error = error_of(c); // lib.rs:294
    } else { // lib.rs:242-284
        n = NES_COLOR_COUNT; cand.resize(3 * n);
        for (uint32_t v = 0; v < n; v++) nes_color(v, &cand[3 * v]);
        best_error = 1.7976931348623157e308; // lib.rs:250
    }
    uint8_t best_color[3] = {original_color[0], original_color[1], original_color[2]};
    if (method == 2) nes_color(0, best_color); // best_index = 0 (lib.rs:249)
    for (uint32_t k = 0; k < n; k++) {
        slot[0] = cand[3 * k]; slot[1] = cand[3 * k + 1]; slot[2] = cand[3 * k + 2];
        optimize(c);
        double error = error_of(c);
        if (error < best_error) { best_error = error; best_color[0] = cand[3 * k]; best_color[1] = cand[3 * k + 1]; best_color[2] = cand[3 * k + 2]; }
    }
    slot[0] = best_color[0]; slot[1] = best_color[1]; slot[2] = best_color[2]; // lib.rs:236, 280, 324
    optimize(c);                                                                  // lib.rs:237 (and :906)
    if (best_error_out) *best_error_out = error_of(c);                            // lib.rs:910
    if (best_rgb5) { best_rgb5[0] = best_color[0]; best_rgb5[1] = best_color[1]; best_rgb5[2] = best_color[2]; }
    return 0;
}

// lib.rs:890 and 917-932
void oracle_schedule_next(uint32_t sub_count, uint32_t sub_size, uint32_t *palette, uint32_t *index, uint32_t *channel, uint32_t *step,
                          uint32_t *method_out, int nes) {
    bool random = (*step % 5) < 4;
    if (method_out) *method_out = nes ? 2u : (random ? 0u : 1u);
    *channel += 1;
    if (*channel == 3 || random) {
        *channel = 0; *index += 1;
        if (*index == sub_size) { *index = 0; *palette += 1; if (*palette == sub_count) { *palette = 0; *step += 1; } }
    }
}

int oracle_get_tile_palettes(oracle_ctx *c, uint8_t *out) { std::memcpy(out, c->tile_palettes.data(), 1024); return 0; }
int oracle_set_tile_palettes(oracle_ctx *c, const uint8_t *in) { std::memcpy(c->tile_palettes.data(), in, 1024); return 0; }
int oracle_get_palette_rgb5(oracle_ctx *c, uint8_t *out) { std::memcpy(out, c->colors.data(), c->colors.size()); return 0; }
int oracle_set_palette_rgb5(oracle_ctx *c, const uint8_t *in) { std::memcpy(c->colors.data(), in, c->colors.size()); return 0; }
int oracle_get_palette_u16(oracle_ctx *c, uint16_t *out) { for (size_t i = 0; i < c->colors.size() / 3; i++) out[i] = snes_as_u16(&c->colors[3 * i]); return 0; }
int oracle_get_palette_map(oracle_ctx *c, uint8_t *out) { std::memcpy(out, c->palette_map.data(), c->palette_map.size()); return 0; }
int oracle_set_palette_map(oracle_ctx *c, const uint8_t *in) { std::memcpy(c->palette_map.data(), in, c->palette_map.size()); return 0; }
int oracle_as_rgba(oracle_ctx *c, uint8_t *out) { as_rgba(c, out); return 0; }

// lib.rs:579-625; serde_json without preserve_order sorts keys: palette, tile_palettes, tiles
int64_t oracle_as_json(oracle_ctx *c, char *out, int64_t cap) {
    std::string s = "{\"palette\":[";
    bool first = true;
    for (uint32_t p = 0; p < c->sub_count; p++)
        for (uint32_t i = 0; i < 16; i++) {
            unsigned v = 0;
            if (i != 0 && i <= c->sub_size) v = snes_as_u16(&c->colors[3 * ((size_t)p * c->sub_size + i - 1)]);
            if (!first) s += ",";
            first = false;
            s += std::to_string(v);
        }
    s += "],\"tile_palettes\":[";
    first = true;
    for (uint32_t ty = 0; ty < c->height_in_tiles(); ty++)
        for (uint32_t tx = 0; tx < c->width_in_tiles(); tx++) {
            if (!first) s += ",";
            first = false;
            s += std::to_string((unsigned)c->tile_palettes[ty * c->width_in_tiles() + tx]);
        }
    s += "],\"tiles\":[";
    first = true;
    for (uint32_t ty = 0; ty < c->height_in_tiles(); ty++)
        for (uint32_t tx = 0; tx < c->width_in_tiles(); tx++) {
            if (!first) s += ",";
            first = false;
            s += "[";
            for (uint32_t y = 0; y < 8; y++)
                for (uint32_t x = 0; x < 8; x++) {
                    size_t idx = (size_t)(ty * 8 + y) * c->width + (tx * 8 + x);
                    unsigned v = c->px(tx * 8 + x, ty * 8 + y)[3] == 0 ? 0u : (unsigned)(uint8_t)(c->palette_map[idx] + 1);
                    if (x || y) s += ",";
                    s += std::to_string(v);
                }
            s += "]";
        }
    s += "]}";
    int64_t need = (int64_t)s.size() + 1;
    if (out && cap > 0) { int64_t m = std::min<int64_t>(cap - 1, (int64_t)s.size()); std::memcpy(out, s.data(), (size_t)m); out[m] = 0; }
    return need;
}

// ---- primitives ----
double oracle_distance_red_mean(const uint8_t *a, const uint8_t *b) { return distance_red_mean(Rgb8{a[0], a[1], a[2]}, Rgb8{b[0], b[1], b[2]}); }
double oracle_distance_cielab(const uint8_t *a, const uint8_t *b) { return distance_cielab(Rgb8{a[0], a[1], a[2]}, Rgb8{b[0], b[1], b[2]}); }
uint32_t oracle_red_mean_key(const uint8_t *a, const uint8_t *b) { return red_mean_key(Rgb8{a[0], a[1], a[2]}, Rgb8{b[0], b[1], b[2]}); }
void oracle_srgb8_to_lab(const uint8_t *rgb, float *lab) { Lab32 l = srgb8_to_lab(Rgb8{rgb[0], rgb[1], rgb[2]}); lab[0] = l.l; lab[1] = l.a; lab[2] = l.b; }
float oracle_ciede2000(const float *a, const float *b) { return ciede2000(Lab32{a[0], a[1], a[2]}, Lab32{b[0], b[1], b[2]}); }
void oracle_lab_to_srgb8(const double *lab, uint8_t *rgb) { Rgb8 v = lab_to_srgb8(lab); rgb[0] = v.r; rgb[1] = v.g; rgb[2] = v.b; }
void oracle_snes_as_rgba(const uint8_t *rgb5, uint8_t *rgba) { Rgb8 v = snes_as_rgba(rgb5); rgba[0] = v.r; rgba[1] = v.g; rgba[2] = v.b; rgba[3] = 255; }
uint16_t oracle_snes_as_u16(const uint8_t *rgb5) { return snes_as_u16(rgb5); }
void oracle_nes_color(uint32_t index, uint8_t *rgb5) { nes_color(index, rgb5); }
void oracle_new_nes_only(const uint8_t *rgb5, int cielab, uint8_t *out) { new_nes_only(rgb5, cielab != 0, out); }
uint32_t oracle_closest_color_index(const uint8_t *entries, uint32_t n, const double *target, int cielab) { return closest_color_index(entries, n, target, cielab != 0); }

int oracle_kmeans(const double *points, uint32_t n, uint32_t k, double *centres_out, uint32_t *assign_out, uint32_t *iterations_out) {
    KmeansResult km = kmeans(points, n, k);
    if (!km.ok) return fail("k-means precondition 2 <= k < n violated");
    std::memcpy(centres_out, km.centres.data(), sizeof(double) * 3 * k);
    if (assign_out) std::memcpy(assign_out, km.assign.data(), sizeof(uint32_t) * n);
    if (iterations_out) *iterations_out = km.iterations;
    return 0;
}

int oracle_ssimulacra2_rgba(const uint8_t *src, const uint8_t *dst, uint32_t w, uint32_t h, int blur_mode, double *score) {
    if (w < 8 || h < 8) return fail("image smaller than 8x8");
    Img3 a, b; rgba_to_linear(src, (int)w, (int)h, a, blur_mode); rgba_to_linear(dst, (int)w, (int)h, b, blur_mode);
    SrcPyramid P; build_src_pyramid(a, blur_mode, P);
    *score = ssimulacra2_against(P, b, blur_mode);
    return 0;
}
void oracle_blur_plane(const float *in, float *out, uint32_t w, uint32_t h, int mode) { blur_plane(in, out, (int)w, (int)h, mode); }
void oracle_blur_constants(float *n2, float *d1, float *fir9) {
    const BlurConsts &C = blur_consts();
    for (int i = 0; i < 3; i++) { n2[i] = C.vert_mul_in[i]; d1[i] = C.vert_mul_prev[i]; }
    for (int j = 0; j < 9; j++) fir9[j] = (float)C.fir[j];
}
void oracle_det_math(int op, const float *x, const float *y, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; i++) {
        switch (op) {
        case 0: out[i] = det_sinf(x[i]); break;
        case 1: out[i] = det_cosf(x[i]); break;
        case 2: out[i] = det_expf_neg(x[i]); break;
        case 3: out[i] = det_cbrtf(x[i]); break;
        default: out[i] = det_atan2f(y[i], x[i]); break;
        }
    }
}

// SURVEY §8d synthetic image
void oracle_synth_image(uint64_t seed, uint32_t w, uint32_t h, int variant, uint8_t *rgba) {
    uint64_t s = seed;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint64_t z = splitmix64_next(s);
            uint32_t n0 = (uint32_t)(z & 63), n1 = (uint32_t)((z >> 8) & 63), n2 = (uint32_t)((z >> 16) & 63);
            uint8_t *o = rgba + 4 * ((size_t)y * w + x);
            o[0] = (uint8_t)((x + n0) & 255); o[1] = (uint8_t)((y + n1) & 255); o[2] = (uint8_t)(((x + y) / 2 + n2) & 255); o[3] = 255;
            if (variant == 1 && x >= 96 && x < 160 && y >= 96 && y < 160 && y < h) o[3] = 0;
        }
}

} // extern "C"
