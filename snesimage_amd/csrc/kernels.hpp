// snesimage_amd/csrc/kernels.hpp — gfx950 kernels of the candidate-scoring hot path.
//
// One candidate scored = replace one palette entry, remap the image (optimize(), lib.rs:425-501)
// and evaluate 100 - SSIMULACRA2 (error(), lib.rs:503-548).  The kernels keep every binary32
// operation of the restated ssimulacra2 0.5.1 pipeline in the reference's order (explicit fmaf
// where the crate uses mul_add; the TU is built with -ffp-contract=off), because the recursive
// Gaussian's rounding noise is amplified by the SSIM map's cancellations: a reordered blur moves
// the error by up to ~4e-5 relative (measured against the oracle), above the 1e-5 budget.
//
// Data layout in HBM (per context unless noted):
//   pack  [H][W] u64  : lo = r | g<<8 | b<<16 | ci<<24 (ci = colour index the pixel takes when the
//                        candidate does NOT win; NCOL+1 = transparent), hi = threshold: the
//                        candidate wins the pixel iff its distance key < hi
//   packT [W][H] u64  : the same, transposed (rows become contiguous along y) for the H pass
//   src scale s       : img1[3][H][W], img1T[3][W][H], mu1[3][H][W], s11[3][H][W] (f32)
//   per candidate     : xyb_s[3][H][W], xybT_s[3][W][H] (s >= 1), hout_s[9][H][W] (H-pass output),
//                        part[S][3][6] f64 partial sums
// Scale s has W_s = W >> s, H_s = H >> s (W = 256, H a power of two).
#pragma once
#include "color.hpp"

namespace snes {

constexpr int kMaxScales = SSIM2_NUM_SCALES;
constexpr int kBlurRadius = 5; // N of ssimulacra2's recursive Gaussian at sigma = 1.5

// Blocked layouts used by the wide-access kernels of kernels_fast.hpp
SNES_HD long long idx_c4(int x, int y, int H) { return ((long long)(x >> 2) * H + y) * 4 + (x & 3); }   // [x/4][y][x%4]
SNES_HD long long idx_r4(int x, int y, int W) { return ((long long)(y >> 2) * W + x) * 4 + (y & 3); }   // [y/4][x][y%4]
SNES_HD long long idx_xt4(int x, int y, int H) { return ((((long long)(x >> 6) * (H >> 2)) + (y >> 2)) << 8) + ((x & 63) << 2) + (y & 3); } // [x/64][y/4][x%64][y%4]

struct BlurK { // recursive-Gaussian constants (host-computed in binary64, rounded to f32)
    float n2[3];   // MUL_IN_k = VERT_MUL_IN_k
    float d1[3];   // VERT_MUL_PREV_k; MUL_PREV_k = -d1
};

struct Geom {
    int W, H, nscales;
    int sw[kMaxScales], sh[kMaxScales];
    // float offsets inside one candidate's workspace
    long long off_xyb[kMaxScales], off_xybT[kMaxScales], off_hout[kMaxScales];
    long long cand_stride; // floats per candidate
    // float offsets inside the source arrays
    long long src_off[kMaxScales]; // same offset used in img1, img1T, mu1, s11 (each 3*N_s at scale s)
};

__device__ __constant__ double kSsim2Weight[108] = SSIM2_WEIGHTS; // include/ssimulacra2_constants.h

// ------------------------------------------------------------------------------------------------
// Palette tables: per colour index the linear RGB and positive-XYB triple of its 8-bit expansion.
// Entry ncol is reserved for the candidate, entry ncol+1 is the transparent pixel (0,0,0).
// ------------------------------------------------------------------------------------------------
__global__ void k_palette_tables(const uint8_t *__restrict__ colors /*ncol*3 raw*/, int ncol, const float *__restrict__ eotf /*256*/,
                                 uint32_t *__restrict__ pal_rgb8, float *__restrict__ pal_lin, float *__restrict__ pal_xyb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncol + 2) return;
    uint32_t rgb8 = 0;
    if (i < ncol) rgb8 = rgb5_to_rgb8(colors[3 * i], colors[3 * i + 1], colors[3 * i + 2]);
    float r = eotf[rgb8 & 0xff], g = eotf[(rgb8 >> 8) & 0xff], b = eotf[(rgb8 >> 16) & 0xff];
    float X, Y, B;
    linear_to_positive_xyb(r, g, b, X, Y, B);
    pal_rgb8[i] = rgb8;
    pal_lin[3 * i] = r; pal_lin[3 * i + 1] = g; pal_lin[3 * i + 2] = b;
    pal_xyb[3 * i] = X; pal_xyb[3 * i + 1] = Y; pal_xyb[3 * i + 2] = B;
}

// Per candidate: 8-bit expansion, linear RGB, positive XYB.  cand_tab[k] = {lin r,g,b, X,Y,B, rgb8 bits, 0}
__device__ __forceinline__ void candidate_tables_body(const uint8_t *__restrict__ rgb5, int n, const float *__restrict__ eotf, float *__restrict__ cand_tab) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t rgb8 = rgb5_to_rgb8(rgb5[3 * k], rgb5[3 * k + 1], rgb5[3 * k + 2]);
    float r = eotf[rgb8 & 0xff], g = eotf[(rgb8 >> 8) & 0xff], b = eotf[(rgb8 >> 16) & 0xff];
    float X, Y, B;
    linear_to_positive_xyb(r, g, b, X, Y, B);
    float *o = cand_tab + 8 * (size_t)k;
    o[0] = r; o[1] = g; o[2] = b; o[3] = X; o[4] = Y; o[5] = B; o[6] = __uint_as_float(rgb8); o[7] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Remap preparation (no dither): nearest-entry argmin of lib.rs:762-795 per pixel.
//   mode 0: argmin over the whole subpalette -> palette_map (this IS optimize() without dither), thr = 0
//   mode 1: ci from the stored palette_map (error() of the current state), thr = 0
//   mode 2: argmin over the subpalette excluding slot (sp, si); thr = key the candidate must beat
// Distance: exact integer redmean key (lib.rs:1080-1088) or, with lab != null, CIEDE2000 in f32
// (lib.rs:1090-1100) on precomputed Lab of the pixel (labpx) and of the entries (pal_lab).
// Ties keep the lowest index (strict <, lib.rs:788-791): the candidate (index si) wins a tie
// against base index j iff si < j, folded into thr (integer keys) or the tie flag bit (float keys).
// ------------------------------------------------------------------------------------------------
struct PrepParams {
    const uint8_t *orig; const uint8_t *tile_pal; const uint32_t *pal_rgb8; uint8_t *map;
    unsigned long long *pack, *packT, *packC4, *packR4;
    uint8_t *subC4, *subR4; // per pixel: subpalette base (sub * sub_size) of its tile, 255 if transparent (dither path)
    const float *labpx; const float *pal_lab; // perceptual only
    int W, H, sub_size, ncol, mode, sp, si, perceptual;
    int *zero; int nzero; // counters of the group-sparse path that belong to this pack: cleared here instead of by memset launches
};

__device__ __forceinline__ void prep_body(const PrepParams &P) {
    if (P.zero && blockIdx.x == 0 && (int)threadIdx.x < P.nzero) P.zero[threadIdx.x] = 0;
    __shared__ uint32_t s_rgb8[256];
    __shared__ float s_lab[256 * 3];
    for (int i = threadIdx.x; i < P.ncol; i += blockDim.x) {
        s_rgb8[i] = P.pal_rgb8[i];
        if (P.perceptual) { s_lab[3 * i] = P.pal_lab[3 * i]; s_lab[3 * i + 1] = P.pal_lab[3 * i + 1]; s_lab[3 * i + 2] = P.pal_lab[3 * i + 2]; }
    }
    __syncthreads();
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= P.W * P.H) return;
    int x = px % P.W, y = px / P.W;
    uint32_t o = reinterpret_cast<const uint32_t *>(P.orig)[px];
    uint32_t rgb = o & 0x00ffffffu;
    bool opaque = (o >> 24) != 0;
    int sub = P.tile_pal[(x >> 3) + (y >> 3) * (P.W >> 3)];
    int base = sub * P.sub_size;
    uint32_t ci, thr = 0;
    if (!opaque) {
        ci = (uint32_t)P.ncol + 1u;
        if (P.mode == 0) P.map[px] = 0;
    } else if (P.mode == 1) {
        ci = (uint32_t)(base + P.map[px]);
    } else {
        bool excl = (P.mode == 2 && sub == P.sp);
        int best = -1;
        if (!P.perceptual) {
            uint32_t bk = 0xffffffffu;
            for (int j = 0; j < P.sub_size; j++) {
                if (excl && j == P.si) continue;
                uint32_t k = red_mean_key(s_rgb8[base + j], rgb);
                if (best < 0 || k < bk) { bk = k; best = j; }
            }
            if (excl) thr = (best < 0) ? 0xffffffffu : bk + (P.si < best ? 1u : 0u);
        } else {
            Lab t; t.l = P.labpx[3 * px]; t.a = P.labpx[3 * px + 1]; t.b = P.labpx[3 * px + 2];
            float bd = 0.0f;
            for (int j = 0; j < P.sub_size; j++) {
                if (excl && j == P.si) continue;
                Lab e; e.l = s_lab[3 * (base + j)]; e.a = s_lab[3 * (base + j) + 1]; e.b = s_lab[3 * (base + j) + 2];
                float d = ciede2000(e, t);
                if (best < 0 || d < bd) { bd = d; best = j; }
            }
            // float keys: bit 31 = "candidate also wins on equality"; distances are >= 0 so bit 31 is free
            if (excl) thr = (best < 0) ? 0xffffffffu : (__float_as_uint(bd) | (P.si < best ? 0x80000000u : 0u));
        }
        if (best < 0) best = 0; // sub_size == 1 and the only entry is the slot
        ci = (uint32_t)(base + best);
        if (P.mode == 0) P.map[px] = (uint8_t)best;
    }
    unsigned long long w = (unsigned long long)(rgb | (ci << 24)) | ((unsigned long long)thr << 32);
    P.pack[px] = w;
    if (P.packT) P.packT[(size_t)x * P.H + y] = w; // (the transposed copy feeds the general H pass only: slot contexts have none)
    P.packC4[idx_c4(x, y, P.H)] = w;
    P.packR4[idx_r4(x, y, P.W)] = w;
    if (P.subC4) { const uint8_t sbv = opaque ? (uint8_t)base : (uint8_t)255; P.subC4[idx_c4(x, y, P.H)] = sbv; P.subR4[idx_r4(x, y, P.W)] = sbv; }
}

// Which colour index does pixel `pk` take for a candidate with 8-bit colour crgb / Lab clab?
template <bool PERCEPTUAL>
__device__ __forceinline__ uint32_t resolve_ci(unsigned long long pk, uint32_t crgb, const Lab &clab, const float *labpx3, uint32_t ncol) {
    uint32_t lo = (uint32_t)pk, thr = (uint32_t)(pk >> 32);
    uint32_t ci0 = lo >> 24;
    if (thr == 0u) return ci0;
    if (!PERCEPTUAL) {
        return red_mean_key(crgb, lo & 0x00ffffffu) < thr ? ncol : ci0;
    } else {
        if (thr == 0xffffffffu) return ncol;
        Lab t; t.l = labpx3[0]; t.a = labpx3[1]; t.b = labpx3[2];
        float bd = __uint_as_float(thr & 0x7fffffffu);
        if (ciede2000_cannot_beat(clab, t, bd)) return ci0;
        float d = ciede2000(clab, t);
        bool win = (d < bd) || ((thr & 0x80000000u) && d == bd);
        return win ? ncol : ci0;
    }
}

// Optional: materialise each candidate's palette_map (for parity tests / maps_out)
struct MapsParams {
    const unsigned long long *pack; const float *cand_tab; const float *cand_lab; const float *labpx;
    uint8_t *maps; int npx, ncol, sub_size, si, ncand, perceptual;
};
__global__ __launch_bounds__(256) void k_candidate_maps(MapsParams P) {
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    int cand = blockIdx.y;
    if (px >= P.npx) return;
    uint32_t crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
    Lab cl; cl.l = cl.a = cl.b = 0.0f;
    uint32_t ci;
    unsigned long long pk = P.pack[px];
    if (P.perceptual) {
        cl.l = P.cand_lab[3 * cand]; cl.a = P.cand_lab[3 * cand + 1]; cl.b = P.cand_lab[3 * cand + 2];
        ci = resolve_ci<true>(pk, crgb, cl, P.labpx + 3 * (size_t)px, (uint32_t)P.ncol);
    } else ci = resolve_ci<false>(pk, crgb, cl, nullptr, (uint32_t)P.ncol);
    uint8_t m;
    if (ci == (uint32_t)P.ncol) m = (uint8_t)P.si;
    else if (ci == (uint32_t)P.ncol + 1u) m = 0;
    else m = (uint8_t)(ci % (uint32_t)P.sub_size);
    P.maps[(size_t)cand * P.npx + px] = m;
}

// The remap on its own (optimize() of lib.rs:425-501 for every candidate, no scoring): thread = four consecutive pixels
// whose pack words stay in registers while it walks kRemapCands candidates; one 32-bit store per candidate.  The pack
// (512 KiB) is served from L2 after its first touch, so the HBM traffic of the kernel is the 64 KiB map per candidate.
constexpr int kRemapCands = 8;
template <bool PERC>
__global__ __launch_bounds__(256) void k_remap4(MapsParams P) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= P.npx) return;
    const uint4 a = reinterpret_cast<const uint4 *>(P.pack)[2 * (size_t)q], b = reinterpret_cast<const uint4 *>(P.pack)[2 * (size_t)q + 1];
    const unsigned long long pk[4] = {((unsigned long long)a.y << 32) | a.x, ((unsigned long long)a.w << 32) | a.z, ((unsigned long long)b.y << 32) | b.x,
                                      ((unsigned long long)b.w << 32) | b.z};
    uint32_t m0[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t ci0 = (uint32_t)pk[i] >> 24;
        m0[i] = ci0 == (uint32_t)P.ncol ? (uint32_t)P.si : (ci0 == (uint32_t)P.ncol + 1u ? 0u : ci0 % (uint32_t)P.sub_size);
    }
    const int c0 = blockIdx.y * kRemapCands;
    for (int cc = 0; cc < kRemapCands && c0 + cc < P.ncand; cc++) {
        const int cand = c0 + cc;
        const uint32_t crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        Lab cl; cl.l = cl.a = cl.b = 0.0f;
        if (PERC) { cl.l = P.cand_lab[3 * cand]; cl.a = P.cand_lab[3 * cand + 1]; cl.b = P.cand_lab[3 * cand + 2]; }
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t ci = resolve_ci<PERC>(pk[i], crgb, cl, PERC ? P.labpx + 3 * ((size_t)q * 4 + i) : nullptr, (uint32_t)P.ncol);
            word |= (ci == (uint32_t)P.ncol ? (uint32_t)P.si : m0[i]) << (8 * i);
        }
        reinterpret_cast<uint32_t *>(P.maps + (size_t)cand * P.npx)[q] = word;
    }
}

// ------------------------------------------------------------------------------------------------
// Downscale chain + XYB for scales 1..S-1 (ssimulacra2 downscale_by_2 in linear RGB, then
// linear_rgb_to_xyb + make_positive_xyb).  One block = one 32x32 block of scale-0 pixels of one
// candidate: 256 threads produce its 16x16 scale-1 pixels, then 8x8, 4x4, 2x2, 1x1.
// Source of scale-0 linear RGB:
//   CAND = true : colour index from pack (+ candidate test) -> table lookup
//   CAND = false: lin0 planes (the source image), written by k_source_scale0
// ------------------------------------------------------------------------------------------------
struct DownParams {
    Geom G;
    const unsigned long long *pack; const float *pal_lin; const float *cand_tab; const float *cand_lab; const float *labpx;
    const float *lin0; // CAND=false
    float *work;       // candidate workspace base (CAND) or source img1 base (!CAND)
    float *workT;      // !CAND: source img1T base
    int ncol, perceptual, use_maps, fast_mask; // fast_mask bit s: scale s planes are stored R4 (off_xyb) and C4 (off_xybT)
    const uint8_t *maps; const uint8_t *tile_pal; int sub_size; // use_maps: ci from per-candidate maps (dither path)
};

template <bool CAND>
__global__ __launch_bounds__(256) void k_downscale_chain(DownParams P) {
    __shared__ float s_lin[256 * 3];
    __shared__ float l1[3][16][17], l2[3][8][9], l3[3][4][5], l4[3][2][3];
    __shared__ float tr[3][16][17];
    const Geom &G = P.G;
    const int t = threadIdx.x;
    const int cand = blockIdx.y;
    const int bx = blockIdx.x % (G.W / 32), by = blockIdx.x / (G.W / 32);
    uint32_t crgb = 0; Lab cl; cl.l = cl.a = cl.b = 0.0f;
    if (CAND) {
        for (int i = t; i < (P.ncol + 2) * 3; i += 256) s_lin[i] = P.pal_lin[i];
        __syncthreads();
        if (t < 3) s_lin[3 * P.ncol + t] = P.cand_tab[8 * (size_t)cand + t];
        crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        if (P.perceptual) { cl.l = P.cand_lab[3 * cand]; cl.a = P.cand_lab[3 * cand + 1]; cl.b = P.cand_lab[3 * cand + 2]; }
        __syncthreads();
    }
    float *wbase = CAND ? P.work + (size_t)cand * G.cand_stride : P.work;
    // ---- scale 1: one pixel per thread
    {
        const int lx = t & 15, ly = t >> 4;
        const int X1 = bx * 16 + lx, Y1 = by * 16 + ly;
        const int W1 = G.sw[1], H1 = G.sh[1];
        float sum[3] = {0.0f, 0.0f, 0.0f};
        const bool in = (Y1 < H1);
        if (in) {
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) {
                    int x0 = X1 * 2 + ix, y0 = Y1 * 2 + iy; // never clamps: W, H are even
                    int px = y0 * G.W + x0;
                    if (CAND) {
                        uint32_t ci;
                        if (P.use_maps) {
                            uint32_t lo = (uint32_t)P.pack[px];
                            ci = lo >> 24; // ncol+1 when transparent
                            if (ci != (uint32_t)P.ncol + 1u) ci = (uint32_t)P.tile_pal[(x0 >> 3) + (y0 >> 3) * (G.W >> 3)] * P.sub_size + P.maps[(size_t)cand * G.W * G.H + px];
                            // candidate colour sits at its real slot in use_maps mode (pal_lin patched per candidate is not possible), so
                            // the caller passes si via cand_tab[7]; see k_dither for the map producer
                            if (ci == __float_as_uint(P.cand_tab[8 * (size_t)cand + 7])) ci = (uint32_t)P.ncol;
                        } else if (P.perceptual) ci = resolve_ci<true>(P.pack[px], crgb, cl, P.labpx + 3 * (size_t)px, (uint32_t)P.ncol);
                        else ci = resolve_ci<false>(P.pack[px], crgb, cl, nullptr, (uint32_t)P.ncol);
                        sum[0] += s_lin[3 * ci]; sum[1] += s_lin[3 * ci + 1]; sum[2] += s_lin[3 * ci + 2];
                    } else {
                        size_t n0 = (size_t)G.W * G.H;
                        sum[0] += P.lin0[px]; sum[1] += P.lin0[n0 + px]; sum[2] += P.lin0[2 * n0 + px];
                    }
                }
        }
        float X, Y, B;
        float r = sum[0] * 0.25f, g = sum[1] * 0.25f, b = sum[2] * 0.25f;
        linear_to_positive_xyb(r, g, b, X, Y, B);
        l1[0][ly][lx] = r; l1[1][ly][lx] = g; l1[2][ly][lx] = b;
        tr[0][ly][lx] = X; tr[1][ly][lx] = Y; tr[2][ly][lx] = B;
        if (in) {
            size_t n1 = (size_t)W1 * H1;
            float *o = wbase + (CAND ? G.off_xyb[1] : G.src_off[1]);
            const size_t oi = (CAND && (P.fast_mask & 2)) ? (size_t)idx_r4(X1, Y1, W1) : (size_t)Y1 * W1 + X1;
            o[oi] = X; o[n1 + oi] = Y; o[2 * n1 + oi] = B;
        }
        __syncthreads();
        // transposed copy: thread (tx = t>>4 -> x, ty = t&15 -> y) so y is the fast index
        {
            const int tx = t >> 4, ty = t & 15;
            const int XT = bx * 16 + tx, YT = by * 16 + ty;
            if (YT < H1) {
                size_t n1 = (size_t)W1 * H1;
                float *oT = CAND ? (wbase + G.off_xybT[1]) : (P.workT + G.src_off[1]);
                const size_t ti = (CAND && (P.fast_mask & 2)) ? (size_t)idx_c4(XT, YT, H1) : (size_t)XT * H1 + YT;
                oT[ti] = tr[0][ty][tx]; oT[n1 + ti] = tr[1][ty][tx]; oT[2 * n1 + ti] = tr[2][ty][tx];
            }
        }
    }
    // ---- scales 2..5 from the LDS copy of the previous scale's linear RGB
#define SNES_DOWN_LEVEL(S, SRC, DST, DIM)                                                                                     \
    if (G.nscales > S) {                                                                                                      \
        if (t < DIM * DIM) {                                                                                                  \
            const int lx = t % DIM, ly = t / DIM;                                                                             \
            const int XS = bx * DIM + lx, YS = by * DIM + ly;                                                                 \
            const int WS = G.sw[S], HS = G.sh[S];                                                                             \
            float v[3];                                                                                                       \
            for (int c = 0; c < 3; c++) {                                                                                     \
                float sum = 0.0f;                                                                                             \
                sum += SRC[c][2 * ly][2 * lx]; sum += SRC[c][2 * ly][2 * lx + 1];                                             \
                sum += SRC[c][2 * ly + 1][2 * lx]; sum += SRC[c][2 * ly + 1][2 * lx + 1];                                     \
                v[c] = sum * 0.25f; DST[c][ly][lx] = v[c];                                                                    \
            }                                                                                                                 \
            if (YS < HS) {                                                                                                    \
                float X, Y, B;                                                                                                \
                linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);                                                            \
                size_t ns = (size_t)WS * HS;                                                                                  \
                float *o = wbase + (CAND ? G.off_xyb[S] : G.src_off[S]);                                                      \
                float *oT = CAND ? (wbase + G.off_xybT[S]) : (P.workT + G.src_off[S]);                                        \
                const bool fst = CAND && ((P.fast_mask >> S) & 1);                                                            \
                const size_t oi = fst ? (size_t)idx_r4(XS, YS, WS) : (size_t)YS * WS + XS;                                    \
                const size_t ti = fst ? (size_t)idx_c4(XS, YS, HS) : (size_t)XS * HS + YS;                                    \
                o[oi] = X; o[ns + oi] = Y; o[2 * ns + oi] = B;                                                                \
                oT[ti] = X; oT[ns + ti] = Y; oT[2 * ns + ti] = B;                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
        __syncthreads();                                                                                                      \
    }
    __syncthreads();
    SNES_DOWN_LEVEL(2, l1, l2, 8)
    SNES_DOWN_LEVEL(3, l2, l3, 4)
    SNES_DOWN_LEVEL(4, l3, l4, 2)
    if (G.nscales > 5 && t == 0) {
        const int WS = G.sw[5], HS = G.sh[5];
        if (by < HS) {
            float v[3];
            for (int c = 0; c < 3; c++) { float sum = 0.0f; sum += l4[c][0][0]; sum += l4[c][0][1]; sum += l4[c][1][0]; sum += l4[c][1][1]; v[c] = sum * 0.25f; }
            float X, Y, B;
            linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);
            size_t ns = (size_t)WS * HS;
            float *o = wbase + (CAND ? G.off_xyb[5] : G.src_off[5]);
            float *oT = CAND ? (wbase + G.off_xybT[5]) : (P.workT + G.src_off[5]);
            o[(size_t)by * WS + bx] = X; o[ns + (size_t)by * WS + bx] = Y; o[2 * ns + (size_t)by * WS + bx] = B;
            oT[(size_t)bx * HS + by] = X; oT[ns + (size_t)bx * HS + by] = Y; oT[2 * ns + (size_t)bx * HS + by] = B;
        }
    }
#undef SNES_DOWN_LEVEL
}

// Source image, scale 0: RGBA8 -> linear RGB planes (lin0) and positive XYB (img1 row-major + transposed)
__global__ __launch_bounds__(256) void k_source_scale0(const uint8_t *__restrict__ orig, const float *__restrict__ eotf, int W, int H,
                                                       float *__restrict__ lin0, float *__restrict__ img1, float *__restrict__ img1T) {
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= W * H) return;
    int x = px % W, y = px / W;
    uint32_t o = reinterpret_cast<const uint32_t *>(orig)[px];
    float r = eotf[o & 0xff], g = eotf[(o >> 8) & 0xff], b = eotf[(o >> 16) & 0xff]; // RGB of transparent pixels included (lib.rs:506-516)
    size_t n = (size_t)W * H;
    lin0[px] = r; lin0[n + px] = g; lin0[2 * n + px] = b;
    float X, Y, B;
    linear_to_positive_xyb(r, g, b, X, Y, B);
    img1[px] = X; img1[n + px] = Y; img1[2 * n + px] = B;
    size_t pt = (size_t)x * H + y;
    img1T[pt] = X; img1T[n + pt] = Y; img1T[2 * n + pt] = B;
}

// ------------------------------------------------------------------------------------------------
// ssim_map + edge_diff_map of one pixel (ssimulacra2 0.5.1), added to the six pooling sums of its column
// {d, d^4, artifact, artifact^4, detail_lost, detail_lost^4}.  Shared by every V-pass flavour, so that all of them (and
// B's checkpoints) hold bit-identical sums.
// The source side is precomputed once per image (k_vpass<.., SRC = true>): sd1 = sigma11 - mu1*mu1 and a1 = |img1 - mu1|,
// the binary32 values the crate forms per call, and r1 = 1 / (1 + a1) in binary64.  The crate's
//     d1 = (1 + |img2 - mu2|) / (1 + |img1 - mu1|) - 1        (binary64)
// is evaluated as (|img2 - mu2| - a1) * r1: equal in exact arithmetic, within 3.3e-16 RELATIVE in binary64 (the
// difference of two binary32 values widened to binary64, one rounding of r1, one of the product) and exactly 0 where the
// quotient form is (a reconstruction identical to the source scores exactly 100) — ~1e-15 relative on the pooled
// sums, ten orders below the 1e-5 the error may move.  It replaces a binary64 division per pixel and channel (a quarter
// of the V pass's issue slots) by a subtraction and a product.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void maps_accumulate(double (&acc)[6], float m1, float sd1, float a1, double r1, float m2, float v22, float v12, float i2) {
    const float mu22 = m2 * m2, mu12 = m1 * m2;
    const float mu_diff = m1 - m2;
    const float num_m = fmaf(mu_diff, -mu_diff, 1.0f);
    const float num_s = fmaf(2.0f, v12 - mu12, SSIM2_C2);
    const float denom_s = sd1 + (v22 - mu22) + SSIM2_C2;
    double d = 1.0 - (double)((num_m * num_s) / denom_s);
    d = d > 0.0 ? d : 0.0;
    acc[0] += d;
    const double dd = d * d;
    acc[1] += dd * dd;
    const double d1 = ((double)fabsf(i2 - m2) - (double)a1) * r1;
    const double art = d1 > 0.0 ? d1 : 0.0;
    const double det = (-d1) > 0.0 ? (-d1) : 0.0;
    acc[2] += art;
    const double a2 = art * art;
    acc[3] += a2 * a2;
    acc[4] += det;
    const double l2 = det * det;
    acc[5] += l2 * l2;
}
__device__ __forceinline__ float source_sd1(float mu1, float sigma11) { const float mu11 = mu1 * mu1; return sigma11 - mu11; }
__device__ __forceinline__ float source_a1(float img1, float mu1) { return fabsf(img1 - mu1); }
__device__ __forceinline__ double source_r1(float a1) { return 1.0 / (1.0 + (double)a1); }

// ------------------------------------------------------------------------------------------------
// Horizontal pass of the recursive Gaussian (ssimulacra2 blur/gaussian.rs horizontal_row) for the
// three planes of one (candidate, channel) pair: mu2 <- img2, s22 <- img2^2, s12 <- img1*img2.
// thread = one image row (the recurrence runs along x and cannot be split without changing its
// rounding), lanes run along y, so inputs are read from the transposed layouts (coalesced) and
// the outputs are staged through LDS in 16-column chunks and written row-major for the V pass.
// A block holds 256 rows: 256/H_s pairs at scale s.
// ------------------------------------------------------------------------------------------------
struct HParams {
    Geom G; BlurK K;
    int s, npairs, ncol, perceptual, use_maps, sub_size;
    const unsigned long long *packT; const float *pal_xyb; const float *cand_tab; const float *cand_lab; const float *labpxT;
    const float *in1T;  // source img1T + src_off[s]: [3][W][H]
    const float *in2T;  // SRC mode: == in1T; else null (candidate workspace is used)
    float *work;        // candidate workspace base (or source scratch hout in SRC mode)
    const uint8_t *mapsT; const uint8_t *tile_pal; // use_maps (dither): per-candidate transposed maps [cand][W][H]
};

template <bool S0, bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_hpass(HParams P) {
    __shared__ __attribute__((aligned(16))) float buf[3][256][20];
    __shared__ float s_lut[3][256];
    const Geom &G = P.G;
    const int s = P.s;
    const int W = G.sw[s], H = G.sh[s];
    const int t = threadIdx.x;
    const int ppw = 256 / H;
    const int ql = t / H, y = t - ql * H;
    const int pair_raw = blockIdx.x * ppw + ql;
    const bool active = pair_raw < P.npairs;
    const int pair = active ? pair_raw : 0;
    const int cand = pair / 3, ch = pair - cand * 3;
    const size_t ns = (size_t)W * H;

    float cand_v = 0.0f; uint32_t crgb = 0; Lab cl; cl.l = cl.a = cl.b = 0.0f; uint32_t cand_slot_ci = 0xffffffffu;
    if (S0) {
        for (int i = t; i < 3 * 256; i += 256) { int c = i >> 8, j = i & 255; s_lut[c][j] = (j < P.ncol + 2) ? P.pal_xyb[3 * j + c] : 0.0f; }
        cand_v = P.cand_tab[8 * (size_t)cand + 3 + ch];
        crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        cand_slot_ci = __float_as_uint(P.cand_tab[8 * (size_t)cand + 7]);
        if (PERCEPTUAL) { cl.l = P.cand_lab[3 * cand]; cl.a = P.cand_lab[3 * cand + 1]; cl.b = P.cand_lab[3 * cand + 2]; }
        __syncthreads();
    }
    const float *in1 = P.in1T + (size_t)ch * ns;
    const float *in2 = S0 ? nullptr : (P.in2T ? P.in2T + (size_t)ch * ns : P.work + (size_t)cand * G.cand_stride + G.off_xybT[s] + (size_t)ch * ns);
    float *hout = P.work + (size_t)cand * G.cand_stride + G.off_hout[s] + (size_t)(ch * 3) * ns;
    (void)hout;

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];

    float pv[3][3], pv2[3][3]; // [plane][term]
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) { pv[p][k] = 0.0f; pv2[p][k] = 0.0f; }
    float dl1[10], dl2[10];
#pragma unroll
    for (int j = 0; j < 10; j++) { dl1[j] = 0.0f; dl2[j] = 0.0f; }

    const int nchunk_cols = W < 16 ? W : 16;
    for (int n0 = -kBlurRadius + 1; n0 < W; n0 += 10) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int n = n0 + j;
            const int xr = n + kBlurRadius - 1;
            float v1 = 0.0f, v2 = 0.0f;
            if (xr < W && n < W) {
                const size_t idx = (size_t)xr * H + y;
                v1 = in1[idx];
                if (S0) {
                    uint32_t ci;
                    if (P.use_maps) {
                        uint32_t lo = (uint32_t)P.packT[idx];
                        ci = lo >> 24;
                        if (ci != (uint32_t)P.ncol + 1u) {
                            ci = (uint32_t)P.tile_pal[(xr >> 3) + (y >> 3) * (G.W >> 3)] * P.sub_size + P.mapsT[(size_t)cand * ns + idx];
                            if (ci == cand_slot_ci) ci = (uint32_t)P.ncol;
                        }
                    } else ci = resolve_ci<PERCEPTUAL>(P.packT[idx], crgb, cl, PERCEPTUAL ? P.labpxT + 3 * idx : nullptr, (uint32_t)P.ncol);
                    v2 = (ci == (uint32_t)P.ncol) ? cand_v : s_lut[ch][ci];
                } else v2 = in2[idx];
            }
            const float l1v = dl1[j], l2v = dl2[j];
            dl1[j] = v1; dl2[j] = v2;
            // plane inputs: left + right
            const float sum0 = l2v + v2;
            const float sum1 = (l2v * l2v) + (v2 * v2);
            const float sum2 = (l1v * l2v) + (v1 * v2);
            const float sums[3] = {sum0, sum1, sum2};
            float outp[3];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                float o1 = sums[p] * n2_0, o3 = sums[p] * n2_1, o5 = sums[p] * n2_2;
                o1 = fmaf(-1.0f, pv2[p][0], o1); o3 = fmaf(-1.0f, pv2[p][1], o3); o5 = fmaf(-1.0f, pv2[p][2], o5);
                pv2[p][0] = pv[p][0]; pv2[p][1] = pv[p][1]; pv2[p][2] = pv[p][2];
                o1 = fmaf(mp_0, pv[p][0], o1); o3 = fmaf(mp_1, pv[p][1], o3); o5 = fmaf(mp_2, pv[p][2], o5);
                pv[p][0] = o1; pv[p][1] = o3; pv[p][2] = o5;
                outp[p] = o1 + o3 + o5;
            }
            if (n >= 0 && n < W) {
                const int xi = n & 15;
                buf[0][t][xi] = outp[0]; buf[1][t][xi] = outp[1]; buf[2][t][xi] = outp[2];
            }
            // flush: uniform condition (n is the same for every thread)
            if (n >= 0 && n < W && ((n & 15) == 15 || n == W - 1)) {
                __syncthreads();
                const int x0 = n & ~15;
                const int q4 = nchunk_cols >> 2; // float4 per row
                const int items = 3 * 256 * q4;
                for (int it = t; it < items; it += 256) {
                    const int c4 = it % q4;
                    const int row = (it / q4) & 255;
                    const int plane = it / (q4 * 256);
                    const int rq = row / H, ry = row - rq * H;
                    const int rpair = blockIdx.x * ppw + rq;
                    if (rpair < P.npairs) {
                        const int rc = rpair / 3, rch = rpair - rc * 3;
                        float *dst = P.work + (size_t)rc * G.cand_stride + G.off_hout[s] + (size_t)(rch * 3 + plane) * ns + (size_t)ry * W + x0 + c4 * 4;
                        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(&buf[plane][row][c4 * 4]);
                    }
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Vertical pass (vertical_pass of blur/gaussian.rs) fused with ssim_map + edge_diff_map and the
// per-(candidate, channel) sums.  thread = one image column; a block holds 256 columns = 256/W_s
// pairs.  SRC = true instead stores mu1 / sigma11 of the source image.
// ------------------------------------------------------------------------------------------------
struct VParams {
    Geom G; BlurK K;
    int s, npairs, ncol, perceptual, use_maps, sub_size, exact_edge;
    const unsigned long long *pack; const float *pal_xyb; const float *cand_tab; const float *cand_lab; const float *labpx;
    const float *img1, *mu1, *sd1, *a1; const double *r1; // source arrays + src_off[s] (img1: SRC only); see maps_accumulate
    float *mu1_out, *sd1_out, *a1_out; double *r1_out;   // SRC: what the pass leaves behind
    const float *work; // candidate workspace (hout, xyb)
    double *part;      // [cand][S][3][6]
    const uint8_t *maps; const uint8_t *tile_pal;
};

template <bool S0, bool SRC, bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_vpass(VParams P) {
    __shared__ float s_lut[3][256];
    __shared__ double red[256][6];
    const Geom &G = P.G;
    const int s = P.s;
    const int W = G.sw[s], H = G.sh[s];
    const int t = threadIdx.x;
    const int ppw = 256 / W;
    const int ql = t / W, x = t - ql * W;
    const int pair_raw = blockIdx.x * ppw + ql;
    const bool active = pair_raw < P.npairs;
    const int pair = active ? pair_raw : 0;
    const int cand = pair / 3, ch = pair - cand * 3;
    const size_t ns = (size_t)W * H;

    float cand_v = 0.0f; uint32_t crgb = 0; Lab cl; cl.l = cl.a = cl.b = 0.0f; uint32_t cand_slot_ci = 0xffffffffu;
    if (S0 && !SRC) {
        for (int i = t; i < 3 * 256; i += 256) { int c = i >> 8, j = i & 255; s_lut[c][j] = (j < P.ncol + 2) ? P.pal_xyb[3 * j + c] : 0.0f; }
        cand_v = P.cand_tab[8 * (size_t)cand + 3 + ch];
        crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        cand_slot_ci = __float_as_uint(P.cand_tab[8 * (size_t)cand + 7]);
        if (PERCEPTUAL) { cl.l = P.cand_lab[3 * cand]; cl.a = P.cand_lab[3 * cand + 1]; cl.b = P.cand_lab[3 * cand + 2]; }
        __syncthreads();
    }
    const float *hout = P.work + (size_t)cand * G.cand_stride + G.off_hout[s] + (size_t)(ch * 3) * ns;
    const float *xyb = (S0 || SRC) ? nullptr : P.work + (size_t)cand * G.cand_stride + G.off_xyb[s] + (size_t)ch * ns;
    const float *img1 = SRC ? P.img1 + (size_t)ch * ns : nullptr;
    const float *mu1 = SRC ? nullptr : P.mu1 + (size_t)ch * ns;
    const float *sd1 = SRC ? nullptr : P.sd1 + (size_t)ch * ns;
    const float *a1 = SRC ? nullptr : P.a1 + (size_t)ch * ns;
    const double *r1 = SRC ? nullptr : P.r1 + (size_t)ch * ns;

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];

    float pv[3][3], pv2[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) { pv[p][k] = 0.0f; pv2[p][k] = 0.0f; }
    float dl[3][10];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int j = 0; j < 10; j++) dl[p][j] = 0.0f;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    for (int n0 = -kBlurRadius + 1; n0 < H; n0 += 10) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int n = n0 + j;
            const int yb = n + kBlurRadius - 1;
            float in[3] = {0.0f, 0.0f, 0.0f};
            if (yb < H && n < H) {
                const size_t idx = (size_t)yb * W + x;
                in[0] = hout[idx]; in[1] = hout[ns + idx]; in[2] = hout[2 * ns + idx];
            }
            float outp[3];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const float sum = dl[p][j] + in[p];
                dl[p][j] = in[p];
                float o1 = fmaf(pv[p][0], d1_0, pv2[p][0]);
                float o3 = fmaf(pv[p][1], d1_1, pv2[p][1]);
                float o5 = fmaf(pv[p][2], d1_2, pv2[p][2]);
                o1 = fmaf(sum, n2_0, -o1); o3 = fmaf(sum, n2_1, -o3); o5 = fmaf(sum, n2_2, -o5);
                pv2[p][0] = pv[p][0]; pv2[p][1] = pv[p][1]; pv2[p][2] = pv[p][2];
                pv[p][0] = o1; pv[p][1] = o3; pv[p][2] = o5;
                outp[p] = o1 + o3 + o5;
            }
            if (n >= 0 && n < H && active) {
                const size_t idx = (size_t)n * W + x;
                if (SRC) {
                    P.mu1_out[(size_t)ch * ns + idx] = outp[0];
                    P.sd1_out[(size_t)ch * ns + idx] = source_sd1(outp[0], outp[1]);
                    const float a1v = source_a1(img1[idx], outp[0]);
                    P.a1_out[(size_t)ch * ns + idx] = a1v;
                    P.r1_out[(size_t)ch * ns + idx] = source_r1(a1v);
                } else {
                    float i2;
                    if (S0) {
                        uint32_t ci;
                        if (P.use_maps) {
                            uint32_t lo = (uint32_t)P.pack[idx];
                            ci = lo >> 24;
                            if (ci != (uint32_t)P.ncol + 1u) {
                                ci = (uint32_t)P.tile_pal[(x >> 3) + (n >> 3) * (G.W >> 3)] * P.sub_size + P.maps[(size_t)cand * ns + idx];
                                if (ci == cand_slot_ci) ci = (uint32_t)P.ncol;
                            }
                        } else ci = resolve_ci<PERCEPTUAL>(P.pack[idx], crgb, cl, PERCEPTUAL ? P.labpx + 3 * idx : nullptr, (uint32_t)P.ncol);
                        i2 = (ci == (uint32_t)P.ncol) ? cand_v : s_lut[ch][ci];
                    } else i2 = xyb[idx];
                    maps_accumulate(acc, mu1[idx], sd1[idx], a1[idx], r1[idx], outp[0], outp[1], outp[2], i2);
                }
            }
        }
    }
    if (!SRC) {
#pragma unroll
        for (int k = 0; k < 6; k++) red[t][k] = active ? acc[k] : 0.0;
        __syncthreads();
        for (int stride = W >> 1; stride > 0; stride >>= 1) {
            if (x < stride) {
#pragma unroll
                for (int k = 0; k < 6; k++) red[t][k] += red[t + stride][k];
            }
            __syncthreads();
        }
        if (x == 0 && active) {
            double *o = P.part + (((size_t)cand * G.nscales + s) * 3 + ch) * 6;
#pragma unroll
            for (int k = 0; k < 6; k++) o[k] = red[t][k];
        }
    }
}

// Msssim::score + `100 - score` (lib.rs:547).  One thread per candidate.
__device__ __forceinline__ void final_score_body(const double *__restrict__ part, int ncand, const Geom &G, double *__restrict__ errors, int err_stride, int err_offset, int *__restrict__ zero, int nzero) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero && c < nzero) zero[c] = 0; // the launch group's work-item counters, ready for the lane's next chunk
    if (c >= ncand) return;
    double ssim = 0.0;
    int i = 0;
    for (int ch = 0; ch < 3; ch++)
        for (int s = 0; s < G.nscales; s++) {
            const double *p = part + (((size_t)c * G.nscales + s) * 3 + ch) * 6;
            const double opp = 1.0 / (double)((size_t)G.sw[s] * G.sh[s]);
            double avg_ssim[2] = {opp * p[0], sqrt(sqrt(opp * p[1]))};
            double avg_edge[4] = {opp * p[2], sqrt(sqrt(opp * p[3])), opp * p[4], sqrt(sqrt(opp * p[5]))};
            for (int n = 0; n < 2; n++) {
                ssim = fma(kSsim2Weight[i], fabs(avg_ssim[n]), ssim); i++;
                ssim = fma(kSsim2Weight[i], fabs(avg_edge[n]), ssim); i++;
                ssim = fma(kSsim2Weight[i], fabs(avg_edge[n + 2]), ssim); i++;
            }
        }
    ssim *= SSIM2_SCORE_SCALE;
    ssim = fma(SSIM2_SCORE_C3 * ssim * ssim, ssim, fma(SSIM2_SCORE_C1, ssim, SSIM2_SCORE_C2 * ssim * ssim));
    if (ssim > 0.0) ssim = fma(pow(ssim, SSIM2_SCORE_EXP), SSIM2_SCORE_GAIN, SSIM2_SCORE_MAX);
    else ssim = SSIM2_SCORE_MAX;
    errors[(size_t)err_offset + (size_t)c * err_stride] = 100.0 - ssim;
}

// The same with a wave per candidate (round 4).  One thread per candidate walks 18 (channel, scale) records of six sums one
// after the other — ~110 dependent loads and square roots, 11-14 us whatever the launch holds: a fixed piece of every call
// and every slot window.  Here lane l < 3 * nscales forms the six terms of record l (channel l / nscales, scale l % nscales:
// the order the chain consumes them in), and the chain itself — the SAME 108 fused multiply-adds in the same order — runs
// on values fetched lane by lane with v_readlane.
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void final_score_wave_body(const double *__restrict__ part, int ncand, const Geom &G, double *__restrict__ errors, int err_stride, int err_offset, int *__restrict__ zero, int nzero) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero && gt < nzero) zero[gt] = 0; // the launch group's work-item counters, ready for the lane's next chunk
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane(gt >> 6);
    if (c >= ncand) return;
    const int ns = G.nscales, npair = 3 * ns;
    double term[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (lane < npair) {
        const int ch = lane / ns, s = lane - ch * ns;
        const double *p = part + (((size_t)c * ns + s) * 3 + ch) * 6;
        const double opp = 1.0 / (double)((size_t)G.sw[s] * G.sh[s]);
        term[0] = opp * p[0]; term[1] = opp * p[2]; term[2] = opp * p[4];                                  // avg_ssim[0], avg_edge[0], avg_edge[2]
        term[3] = sqrt(sqrt(opp * p[1])); term[4] = sqrt(sqrt(opp * p[3])); term[5] = sqrt(sqrt(opp * p[5])); // avg_ssim[1], avg_edge[1], avg_edge[3]
    }
    double ssim = 0.0;
    int i = 0;
    for (int l = 0; l < npair; l++) {
#pragma unroll
        for (int t = 0; t < 6; t++) { ssim = fma(kSsim2Weight[i], fabs(readlane_f64(term[t], l)), ssim); i++; }
    }
    if (lane != 0) return;
    ssim *= SSIM2_SCORE_SCALE;
    ssim = fma(SSIM2_SCORE_C3 * ssim * ssim, ssim, fma(SSIM2_SCORE_C1, ssim, SSIM2_SCORE_C2 * ssim * ssim));
    if (ssim > 0.0) ssim = fma(pow(ssim, SSIM2_SCORE_EXP), SSIM2_SCORE_GAIN, SSIM2_SCORE_MAX);
    else ssim = SSIM2_SCORE_MAX;
    errors[(size_t)err_offset + (size_t)c * err_stride] = 100.0 - ssim;
}

// ---- kernel entry points of the bodies above (kernels_batch.hpp holds the many-images flavours) ----
__global__ void k_candidate_tables(const uint8_t *__restrict__ rgb5, int n, const float *__restrict__ eotf, float *__restrict__ cand_tab) { candidate_tables_body(rgb5, n, eotf, cand_tab); }
__global__ __launch_bounds__(256) void k_prep(PrepParams P) { prep_body(P); }
__global__ void k_final_score(const double *__restrict__ part, int ncand, Geom G, double *__restrict__ errors, int err_stride, int err_offset, int *__restrict__ zero = nullptr, int nzero = 0) {
    final_score_body(part, ncand, G, errors, err_stride, err_offset, zero, nzero);
}
// a wave per candidate, four candidates per block: grid (ncand + 3) / 4, 256 threads
__global__ __launch_bounds__(256) void k_final_score_wave(const double *__restrict__ part, int ncand, Geom G, double *__restrict__ errors, int err_stride, int err_offset, int *__restrict__ zero = nullptr, int nzero = 0) {
    final_score_wave_body(part, ncand, G, errors, err_stride, err_offset, zero, nzero);
}

} // namespace snes
