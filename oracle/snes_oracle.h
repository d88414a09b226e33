/* oracle/snes_oracle.h — TEST INFRASTRUCTURE. CPU restatement of the snesimage palette-optimizer
 * hot path (reference: aexoden/snesimage, /root/reference/src/lib.rs).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the shipped
 * product (libsnesimage_hip.so) never links, loads or calls it.
 *
 * PARITY STATUS: "parity unpinned" for every third-party-crate boundary (cogset 0.2.0,
 * palette 0.7.6, ssimulacra2 0.5.1 / yuvxyb 0.4.2 / yuvxyb-math 0.1.1): the reference holds no
 * tests, fixtures or golden vectors, its crates are not vendored and there is
 * no Rust toolchain here, so those parts restate the published algorithms and
 * are pinned only by (a) the first-party known-answer tests derivable from src/lib.rs
 *  and (b) external published vectors (Sharma's CIEDE2000 table, the
 * canonical sRGB->Lab primaries).  First-party arithmetic (src/lib.rs) is restated exactly.
 */
#ifndef SNES_ORACLE_H
#define SNES_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

enum { ORACLE_DITHER = 1, ORACLE_PERCEPTUAL = 2, ORACLE_NES = 4 };

/* lib.rs:46-65. rgba is row-major RGBA8, w must be 256 (tile stride is hard-coded to 32 at
 * lib.rs:58,565), h a multiple of 8 in [8,256]. Returns NULL on bad arguments. */
oracle_ctx *oracle_create(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t sub_count,
                          uint32_t sub_size, uint32_t flags);
void oracle_destroy(oracle_ctx *);

/* 1: reuse the source-side SSIMULACRA2 terms between error() calls (same results, faster tests).
 * 0 (default): recompute them on every call like lib.rs:506-525 does (the CPU baseline). */
void oracle_set_cache_source(oracle_ctx *, int on);
/* 0 (default): recursive-Gaussian blur exactly as restated from ssimulacra2 0.5.1.
 * 1: mathematically equivalent zero-padded 9-tap FIR (an experiment knob for tests only). */
void oracle_set_blur_mode(oracle_ctx *, int mode);
/* What-if variants of the UNPINNED third-party arithmetic behind error() (never the product's arithmetic):
 * bit 0: zimg-style sRGB transfer constants in yuvxyb's EOTF (alpha 1.0550107, beta 0.0030412825, linear below 12.92 beta),
 * bit 1: powf / cbrtf evaluated in binary32 as a fast-math crate would (exp2f(y log2f(x)); bit-hack seed + two Halley steps),
 * bit 2: the exact powf / cbrtf results moved by a pseudo-random relative amount of at most 1e-6.  0 (default): the restatement. */
void oracle_set_variant(oracle_ctx *, int bits);

int oracle_initialize_tiles(oracle_ctx *);       /* lib.rs:79-189  */
int oracle_recalculate_palettes(oracle_ctx *);   /* lib.rs:407-415 */
int oracle_optimize(oracle_ctx *);               /* lib.rs:425-501 */
int oracle_error(oracle_ctx *, double *out);     /* lib.rs:503-548 */
int oracle_reassign_tiles(oracle_ctx *, uint32_t *moved); /* not in the reference (TODO.md:36-37); definition in snes_oracle.cpp */

/* Body of lib.rs:205-220 for an explicit candidate list: for each k, entry (palette,index) :=
 * rgb5[3k..3k+2] (raw 5-bit r,g,b), optimize(), error().  The palette entry and palette_map are
 * restored afterwards.  maps_out (optional, n*w*h bytes) receives each candidate's palette_map. */
int oracle_score_candidates(oracle_ctx *, uint32_t palette, uint32_t index, const uint8_t *rgb5,
                            uint32_t n, double *errors, uint8_t *maps_out);

/* One optimizer call: method 0 = random (lib.rs:191-240, 64 candidates from the counter RNG),
 * 1 = channel (lib.rs:286-328), 2 = nes (lib.rs:242-284); then lib.rs:906-910.
 * n_random overrides the 64 of lib.rs:205 when non-zero. */
int oracle_step(oracle_ctx *, uint32_t method, uint32_t palette, uint32_t index, uint32_t channel,
                uint64_t seed, uint64_t step_id, uint32_t n_random, double *best_error,
                uint8_t *best_rgb5);

/* Counter-based candidate generator shared by definition with the product:
 * candidate k of (seed, step_id) -> r,g,b in 0..31, sampled in that order (lib.rs:206-208). */
void oracle_random_candidates(uint64_t seed, uint64_t step_id, uint32_t n, uint8_t *rgb5);

/* Slot scheduler of lib.rs:881-933: advances (palette, index, channel, step) after one call. */
void oracle_schedule_next(uint32_t sub_count, uint32_t sub_size, uint32_t *palette,
                          uint32_t *index, uint32_t *channel, uint32_t *step, uint32_t *method_out,
                          int nes);

int oracle_get_tile_palettes(oracle_ctx *, uint8_t *out /*1024*/);
int oracle_set_tile_palettes(oracle_ctx *, const uint8_t *in /*1024*/);
int oracle_get_palette_rgb5(oracle_ctx *, uint8_t *out /*count*size*3 raw components*/);
int oracle_set_palette_rgb5(oracle_ctx *, const uint8_t *in);
int oracle_get_palette_u16(oracle_ctx *, uint16_t *out /*count*size, as_u16 lib.rs:679-681*/);
int oracle_get_palette_map(oracle_ctx *, uint8_t *out /*w*h*/);
int oracle_set_palette_map(oracle_ctx *, const uint8_t *in);
int oracle_as_rgba(oracle_ctx *, uint8_t *out /*w*h*4, lib.rs:550-577*/);
/* lib.rs:579-625 + 1002: compact JSON, keys sorted. Returns bytes needed (incl. NUL); writes at
 * most cap bytes. */
int64_t oracle_as_json(oracle_ctx *, char *out, int64_t cap);

/* ---- primitives exposed for known-answer tests ---- */
double oracle_distance_red_mean(const uint8_t *rgb1, const uint8_t *rgb2);       /* lib.rs:1080-1088 */
double oracle_distance_cielab(const uint8_t *rgb1, const uint8_t *rgb2);         /* lib.rs:1090-1100 */
uint32_t oracle_red_mean_key(const uint8_t *rgb1, const uint8_t *rgb2);          /* 512 * pre-sqrt value */
void oracle_srgb8_to_lab(const uint8_t *rgb, float *lab);                        /* palette 0.7.6 */
float oracle_ciede2000(const float *lab1, const float *lab2);                    /* palette 0.7.6 */
void oracle_lab_to_srgb8(const double *lab, uint8_t *rgb);                       /* lib.rs:141-142 */
void oracle_snes_as_rgba(const uint8_t *rgb5, uint8_t *rgba);                    /* lib.rs:662-669 */
uint16_t oracle_snes_as_u16(const uint8_t *rgb5);                                /* lib.rs:679-681 */
void oracle_nes_color(uint32_t index, uint8_t *rgb5);                            /* lib.rs:685-745 */
void oracle_new_nes_only(const uint8_t *rgb5, int cielab, uint8_t *out_rgb5);    /* lib.rs:640-660 */
uint32_t oracle_closest_color_index(const uint8_t *entries_rgb5, uint32_t n, const double *target,
                                    int cielab);                                 /* lib.rs:762-795 */
/* cogset 0.2.0 Kmeans::new restated: points are n x 3 f64; centres_out k x 3; assign_out n. */
int oracle_kmeans(const double *points, uint32_t n, uint32_t k, double *centres_out,
                  uint32_t *assign_out, uint32_t *iterations_out);
/* ssimulacra2 0.5.1 compute_frame_ssimulacra2 on two RGBA8 images (alpha ignored): returns score. */
int oracle_ssimulacra2_rgba(const uint8_t *src, const uint8_t *dst, uint32_t w, uint32_t h,
                            int blur_mode, double *score);
/* Blur one plane (mode as oracle_set_blur_mode). */
void oracle_blur_plane(const float *in, float *out, uint32_t w, uint32_t h, int mode);
/* Recursive-Gaussian constants (n2[3], d1[3] as f32) and the equivalent FIR taps h[-4..4]. */
void oracle_blur_constants(float *n2, float *d1, float *fir9);
/* det_math sweeps for device-vs-host bit comparison: op 0 sin, 1 cos, 2 exp_neg, 3 cbrt, 4 atan2(x,y) */
void oracle_det_math(int op, const float *x, const float *y, uint32_t n, float *out);
/* Synthetic image of SURVEY §8d: variant 0 opaque, 1 with a transparent 64x64 square. */
void oracle_synth_image(uint64_t seed, uint32_t w, uint32_t h, int variant, uint8_t *rgba);

const char *oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
