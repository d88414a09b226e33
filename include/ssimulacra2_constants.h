/* include/ssimulacra2_constants.h — every numeric constant of the restated third-party arithmetic behind `error()`
 * (lib.rs:503-548): ssimulacra2 0.5.1 (+ yuvxyb 0.4.2), and of palette 0.7.6's sRGB / XYZ / Lab conversions behind the
 * CIEDE2000 distance and the k-means initialisers (lib.rs:101-103, 141-142, 1090-1100), as restated in SURVEY.md Appendix A.
 *
 * The reference pulls these from crates that are not vendored under /root/reference and cannot be fetched or built in this
 * environment, so the values below are a transcription from the published algorithm (libjxl tools/ssimulacra2.cc,
 * SSIMULACRA 2.1) and are NOT pinned to upstream ("parity unpinned").  They live in this ONE header, included by
 * both the CPU oracle (oracle/snes_oracle.cpp, test infrastructure) and the product (snesimage_amd/csrc), so that a
 * correction against an authoritative copy is a one-file change that moves both sides together.  Plain C: macros only.
 */
#ifndef SSIMULACRA2_CONSTANTS_H
#define SSIMULACRA2_CONSTANTS_H

/* linear sRGB -> XYB (yuvxyb linear_rgb_to_xyb): opsin absorbance matrix rows, bias, and the cube root of the bias */
#define SSIM2_OPSIN_M00 0.30f
#define SSIM2_OPSIN_M02 0.078f
#define SSIM2_OPSIN_M10 0.23f
#define SSIM2_OPSIN_M12 0.078f
#define SSIM2_OPSIN_M20 0.24342268924547819f
#define SSIM2_OPSIN_M21 0.20476744424496821f
#define SSIM2_OPSIN_BIAS 0.0037930732552754493f
#define SSIM2_OPSIN_BIAS_CBRT 0.1559542025327239180319220163705f
/* (M01 = 1 - M02 - M00, M11 = 1 - M12 - M10, M22 = 1 - M20 - M21, formed in binary32) */
/* make_positive_xyb: B = (B - Y) + 0.55, X = 14 X + 0.42, Y += 0.01 */
#define SSIM2_POS_B_OFFSET 0.55f
#define SSIM2_POS_X_SCALE 14.0f
#define SSIM2_POS_X_OFFSET 0.42f
#define SSIM2_POS_Y_OFFSET 0.01f

/* recursive Gaussian (libjxl FastGaussian): sigma, and radius = round(3.2795 sigma + 0.2546) */
#define SSIM2_BLUR_SIGMA 1.5
#define SSIM2_BLUR_RADIUS_A 3.2795
#define SSIM2_BLUR_RADIUS_B 0.2546

/* ssim_map */
#define SSIM2_C2 0.0009f

/* number of scales and the size test of the scale loop (`if width < 8 || height < 8 { break }`) */
#define SSIM2_NUM_SCALES 6
#define SSIM2_MIN_SIDE 8

/* Msssim::score: 108 weights (channel -> scale -> norm; {ssim, artifact, detail_lost} per step), then the polynomial */
#define SSIM2_WEIGHTS { \
    0.0, 0.0007376606707406586, 0.0, 0.0, 0.0007793481682867309, 0.0, 0.0, 0.0004371155730107379, 0.0, \
    1.1041726426657346, 0.00066284834129271, 0.00015231632783718752, 0.0, 0.0016406437456599754, 0.0, \
    1.8422455520539298, 11.441172603757666, 0.0, 0.0007989109436015163, 0.000176816438078653, 0.0, \
    1.8787594979546387, 10.94906990605142, 0.0, 0.0007289346991508072, 0.9677937080626833, 0.0, \
    0.00014003424285435884, 0.9981766977854967, 0.00031949755934435053, 0.0004550992113792063, 0.0, 0.0, \
    0.0013648766163243398, 0.0, 0.0, 0.0, 0.0, 0.0, 7.466890328078848, 0.0, 17.445833984131262, \
    0.0006235601634041466, 0.0, 0.0, 6.683678146179332, 0.00037724407979611296, 1.027889937768264, \
    225.20515300849274, 0.0, 0.0, 19.213238186143016, 0.0011401524586618361, 0.001237755635509985, \
    176.39317598450694, 0.0, 0.0, 24.43300999870476, 0.28520802612117757, 0.0004485436923833408, \
    0.0, 0.0, 0.0, 34.77906344483772, 44.835625328877896, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, \
    0.0, 0.0008680556573291698, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0005313191874358747, 0.0, \
    0.00016533814161379112, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0004179171803251336, 0.0017290828234722833, \
    0.0, 0.0020827005846636437, 0.0, 0.0, 8.826982764996862, 23.19243343998926, 0.0, \
    95.1080498811086, 0.9863978034400682, 0.9834382792465353, 0.0012286405048278493, \
    171.2667255897307, 0.9807858872435379, 0.0, 0.0, 0.0, 0.0005130064588990679, 0.0, \
    0.00010854057858411537 }
#define SSIM2_SCORE_SCALE 0.9562382616834844
#define SSIM2_SCORE_C1 2.326765642916932
#define SSIM2_SCORE_C2 (-0.020884521182843837)
#define SSIM2_SCORE_C3 6.248496625763138e-5
#define SSIM2_SCORE_EXP 0.6276336467831387
#define SSIM2_SCORE_GAIN (-10.0)
#define SSIM2_SCORE_MAX 100.0

/* ---- yuvxyb 0.4.2: sRGB transfer (TransferCharacteristic::SRGB) on binary32 values in [0,1] (lib.rs:518-525, 538-545) ----
 * x < THRESHOLD ? x / LINEAR_DIV : ((x + OFFSET) / SCALE) ^ GAMMA.  (To the builder's recollection — unverifiable offline —
 * upstream may port zimg's constants instead: SCALE 1.0550107, beta 0.0030412825, threshold 12.92 beta, and evaluate powf /
 * cbrtf through yuvxyb-math's approximations; oracle_set_variant() measures what either would move: DESIGN.md section 2.) */
#define SSIM2_SRGB_THRESHOLD 0.04045f
#define SSIM2_SRGB_LINEAR_DIV 12.92f
#define SSIM2_SRGB_OFFSET 0.055f
#define SSIM2_SRGB_SCALE 1.055f
#define SSIM2_SRGB_GAMMA 2.4f
/* the zimg-style variant (oracle only, never the product): alpha, beta; linear below 12.92 beta, negative inputs clamp to 0 */
#define SSIM2_ZIMG_SRGB_ALPHA 1.0550107f
#define SSIM2_ZIMG_SRGB_BETA 0.0030412825f

/* ---- palette 0.7.6: Srgb<u8> -> Lab<D65, f32> (lib.rs:101-103, 344-346, 1092-1097) and back (lib.rs:141-142, 369-371) ----
 * Srgb::into_linear: x <= THRESHOLD ? x * (1 / 12.92) : (x * (1 / 1.055) + 0.055 / 1.055) ^ 2.4, constants formed in binary64 and
 * rounded once (PALETTE_SRGB_*_D are the binary64 values the binary32 ones are rounded from) */
#define PALETTE_SRGB_THRESHOLD 0.04045f
#define PALETTE_SRGB_LINEAR_DIV_D 12.92
#define PALETTE_SRGB_SCALE_D 1.055
#define PALETTE_SRGB_OFFSET_D 0.055
#define PALETTE_SRGB_GAMMA 2.4f
#define PALETTE_SRGB_GAMMA_D 2.4
#define PALETTE_SRGB_ENCODE_THRESHOLD_D 0.0031308
/* linear sRGB -> XYZ (D65), rows X, Y, Z */
#define PALETTE_XYZ_XR 0.4124564f
#define PALETTE_XYZ_XG 0.3575761f
#define PALETTE_XYZ_XB 0.1804375f
#define PALETTE_XYZ_YR 0.2126729f
#define PALETTE_XYZ_YG 0.7151522f
#define PALETTE_XYZ_YB 0.0721750f
#define PALETTE_XYZ_ZR 0.0193339f
#define PALETTE_XYZ_ZG 0.1191920f
#define PALETTE_XYZ_ZB 0.9503041f
/* XYZ -> linear sRGB (binary64: the centres of the k-means initialisers are Lab<D65, f64>), rows R, G, B */
#define PALETTE_RGB_RX 3.2404542
#define PALETTE_RGB_RY (-1.5371385)
#define PALETTE_RGB_RZ (-0.4985314)
#define PALETTE_RGB_GX (-0.9692660)
#define PALETTE_RGB_GY 1.8760108
#define PALETTE_RGB_GZ 0.0415560
#define PALETTE_RGB_BX 0.0556434
#define PALETTE_RGB_BY (-0.2040259)
#define PALETTE_RGB_BZ 1.0572252
/* D65 white point */
#define PALETTE_D65_X 0.95047f
#define PALETTE_D65_Y 1.0f
#define PALETTE_D65_Z 1.08883f
#define PALETTE_D65_X_D 0.95047
#define PALETTE_D65_Y_D 1.0
#define PALETTE_D65_Z_D 1.08883
/* Lab: f(t) = t > (6/29)^3 ? cbrt(t) : (841/108) t + 4/29; L = 116 f(y) - 16, a = 500 (f(x) - f(y)), b = 200 (f(y) - f(z)) */
#define PALETTE_LAB_EPS_ROOT_D (6.0 / 29.0)
#define PALETTE_LAB_KAPPA_D (841.0 / 108.0)
#define PALETTE_LAB_KAPPA_INV_D (108.0 / 841.0)
#define PALETTE_LAB_DELTA_D (4.0 / 29.0)
#define PALETTE_LAB_L_SCALE 116.0f
#define PALETTE_LAB_L_OFFSET 16.0f
#define PALETTE_LAB_A_SCALE 500.0f
#define PALETTE_LAB_B_SCALE 200.0f
#define PALETTE_LAB_L_SCALE_D 116.0
#define PALETTE_LAB_L_OFFSET_D 16.0
#define PALETTE_LAB_A_SCALE_D 500.0
#define PALETTE_LAB_B_SCALE_D 200.0

#endif
