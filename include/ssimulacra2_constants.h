/* include/ssimulacra2_constants.h — every numeric constant of the restated third-party arithmetic behind `error()`
 * (lib.rs:503-548): ssimulacra2 0.5.1 (+ yuvxyb 0.4.2) as restated in SURVEY.md Appendix A.
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

#endif
