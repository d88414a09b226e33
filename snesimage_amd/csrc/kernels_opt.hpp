// snesimage_amd/csrc/kernels_opt.hpp — the kernels around the scoring pipeline: Lab tables for
// --perceptual-palettes, the Floyd-Steinberg remap (--dither), candidate generation and the
// commit rule of the optimizer step, and the k-means initialisers.
#pragma once
#include "kernels.hpp"

namespace snes {

// ---- Lab tables (palette 0.7.6 pipeline; lab_eotf = Srgb::into_linear on v/255) -----------------
__global__ void k_palette_lab(const uint32_t *__restrict__ pal_rgb8, int ncol, const float *__restrict__ lab_eotf, float *__restrict__ pal_lab) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncol) return;
    uint32_t c = pal_rgb8[i];
    Lab l = linear_to_lab(lab_eotf[c & 0xff], lab_eotf[(c >> 8) & 0xff], lab_eotf[(c >> 16) & 0xff]);
    pal_lab[3 * i] = l.l; pal_lab[3 * i + 1] = l.a; pal_lab[3 * i + 2] = l.b;
}
__device__ __forceinline__ void candidate_lab_body(const float *__restrict__ cand_tab, int n, const float *__restrict__ lab_eotf, float *__restrict__ cand_lab) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t c = __float_as_uint(cand_tab[8 * (size_t)k + 6]);
    Lab l = linear_to_lab(lab_eotf[c & 0xff], lab_eotf[(c >> 8) & 0xff], lab_eotf[(c >> 16) & 0xff]);
    cand_lab[3 * k] = l.l; cand_lab[3 * k + 1] = l.a; cand_lab[3 * k + 2] = l.b;
}
__global__ void k_candidate_lab(const float *__restrict__ cand_tab, int n, const float *__restrict__ lab_eotf, float *__restrict__ cand_lab) { candidate_lab_body(cand_tab, n, lab_eotf, cand_lab); }
__global__ void k_candidate_slot(float *__restrict__ cand_tab, int n, uint32_t slot_ci) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) cand_tab[8 * (size_t)k + 7] = __uint_as_float(slot_ci);
}
__global__ __launch_bounds__(256) void k_pixel_lab(const uint8_t *__restrict__ orig, const float *__restrict__ lab_eotf, int W, int H, float *__restrict__ labpx, float *__restrict__ labpxT) {
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= W * H) return;
    int x = px % W, y = px / W;
    uint32_t o = reinterpret_cast<const uint32_t *>(orig)[px];
    Lab l = linear_to_lab(lab_eotf[o & 0xff], lab_eotf[(o >> 8) & 0xff], lab_eotf[(o >> 16) & 0xff]);
    labpx[3 * (size_t)px] = l.l; labpx[3 * (size_t)px + 1] = l.a; labpx[3 * (size_t)px + 2] = l.b;
    size_t pt = (size_t)x * H + y;
    labpxT[3 * pt] = l.l; labpxT[3 * pt + 1] = l.a; labpxT[3 * pt + 2] = l.b;
}

// ------------------------------------------------------------------------------------------------
// optimize() with Floyd-Steinberg error diffusion (lib.rs:425-501), one block per candidate.
// The raster scan's dependency (x-1,y), (x-1..x+1,y-1) leaves the anti-diagonals t = x + 2y free:
// thread j owns rows j and j+128, so it works through 512 consecutive pixels starting at step 2j.
// Diffused error is gathered, not scattered: e(x,y) = (((0 + v(x-1,y-1)*0.8*w3) + v(x,y-1)*0.8*w2)
// + v(x+1,y-1)*0.8*w1) + v(x-1,y)*0.8*w0 — the order in which the reference's `+=` reach the
// pixel — in binary64, so the rounding is the reference's.  v of the row above comes from a
// 4-deep LDS ring written by the neighbouring thread 1..3 steps earlier.
// ------------------------------------------------------------------------------------------------
struct DitherParams {
    const uint8_t *orig; const uint8_t *tile_pal; const uint32_t *pal_rgb8; const float *pal_lab; const float *cand_tab; const float *cand_lab; const float *lab_eotf;
    uint8_t *maps, *mapsC4; // row-major and the column-blocked layout of kernels_fast.hpp (k_maps_relayout derives the others)
    int W, H, sub_size, ncol; uint32_t slot_ci; int perceptual;
    const int *skip; // optional device flag: nonzero = the map is already in place, leave at once
    // MODE 1 (the base image B of a slot, dithered once with the slot's entry out of play) additionally leaves behind:
    unsigned long long *rec_pack; // per pixel, in the pack's format: lo = dithered target (8-bit, clamped and rounded as lib.rs:773-778) | B's colour index << 24,
                                  // hi = the key a candidate must beat to take the pixel (0 outside the slot's subpalette): the win test of k_dither_first
                                  // (PERC: the bits of the CIEDE2000 distance to beat — non-negative binary32 values order as their bits — plus one where a tie goes to the slot)
    double *ck_out;               // [H/4 + 1][W][3]: diffused-error state entering row 4g (the v values of row 4g-1), g >= 1
    float *rec_lab;               // PERC: the dithered target's Lab per pixel (what the search compared the entries with): k_dither_first_lab tests candidates against it
    int excl_sub, excl_si, excl_j0; // slot (subpalette, index); j0 = the other entry whose colour stands in for the slot's (see k_dither)
    // MODE 2 (a candidate resumed where it first differs from B):
    const int *first_group;       // per candidate: 4-row group holding its first won pixel in raster order (H/4 if it wins nothing); P.k0 offsets the index
    int first_k0;
    const double *ck_in; const uint8_t *bmap, *bmapC4; // B's checkpoints and map (rows above the first group are B's)
    const unsigned long long *rec_in; // B's per-pixel record (rec_pack of its MODE 1 run): where a resumed run meets B's dithered target again, B's search result stands
    const int *order;             // block b takes candidate order[b] (k_sparse_order: longest resumes first); nullptr = as listed
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// RGB nearest entry among n (<= 8) consecutive table rows: 8*red_mean_key + u for row u, so the smallest value is the
// first minimal key.  Per entry the table holds {r | b<<16, 8*(1024+r) | 8*(1534-r)<<16, g<<7, rgb8}; with the target's
// {r | b<<16, 8r | (-8r)<<16, g<<7} one key is three packed 16-bit ops (differences, their squares <= 65,025, the two
// red-mean weights <= 12,272), the green term 2048 * 8 * dg^2 = (dg<<7)^2 as one 24-bit multiply-add that also brings
// the row's number (|dg<<7| < 2^15; the 32-bit v_mul_lo_u32 runs at a quarter of the rate) and one v_dot2_u32_u16 —
// every intermediate exact.
template <int N>
__device__ __forceinline__ uint32_t dither_group_min(const uint4 *__restrict__ ent, uint32_t t1, uint32_t tw, int tg) {
    uint32_t k[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (u < N) {
            const uint4 e = ent[u];
            const u16x2 d = as_u16x2(e.x) - as_u16x2(t1);
            const u16x2 sq = d * d;
            const u16x2 wg = as_u16x2(e.y) + as_u16x2(tw);
            const int dg = (int)e.z - tg; // (times 128)
            k[u] = __builtin_amdgcn_udot2(sq, wg, (uint32_t)(__mul24(dg, dg) + u), false);
        } else k[u] = 0xffffffffu;
    }
    return min(min(min(k[0], k[1]), min(k[2], k[3])), min(min(k[4], k[5]), min(k[6], k[7])));
}

// PERC: CIEDE2000 in Lab (the --perceptual-palettes remap); SUB: subpalette size known at compile time (0 = read it from P)
// MODE 0: the whole image.
// MODE 1: the same for the base image B of a slot — the candidate colour handed in is the colour of another entry j0 of the
//         slot's subpalette, so the slot's entry duplicates j0: the nearest-entry search then returns the best OTHER entry
//         (a duplicate never changes the minimum key; where the duplicate itself is returned, the lowest-index minimum
//         among the others is j0), its key is what a candidate has to beat, and the diffused error is that of the image
//         without the slot's entry.  Records the win-test word per pixel and the row state at every 4-row boundary.
// MODE 2: a candidate, resumed: Floyd-Steinberg is causal in raster order, so up to its first won pixel the candidate's
//         run IS B's run.  Rows above the 4-row group of that pixel are copied from B's map, the state entering the group
//         comes from B's checkpoint, and only the rows from there on are dithered (all of them: the error spreads).
// NT: threads per block = rows in flight.  A thread that finishes a row goes on with row y + NT, whose upper neighbour is the
//     block's last thread, 2 (NT - 1) steps behind: its ring still holds the columns needed only if 2 (NT - 1) >= W - 2, i.e.
//     NT = 128 at W = 256 (a 64-thread variant is not merely slower, it is wrong).
template <bool PERC, int SUB, int MODE = 0, int NT = 128>
__device__ __forceinline__ void dither_body(const DitherParams &P, const int blk) { // blk: the launch's block index (blockIdx.x unless the caller batches over z)
    static_assert(2 * (NT - 1) >= 256 - 2, "the wavefront's ring is four columns deep: see NT above");
    __shared__ uint4 s_ent[256];
    __shared__ float s_lab[PERC ? 256 * 3 : 1];
    __shared__ float s_eotf[PERC ? 256 : 1];
    __shared__ double ring[NT][4][3];
    __shared__ uint8_t s_tile[1024];
    if (P.skip && *P.skip) return;
    const int j = threadIdx.x;
    const int cand = (MODE == 2 && P.order) ? P.order[blk] : blk;
    constexpr int W = 256; // snesimage_create admits no other width; a constant keeps the per-step index arithmetic to shifts
    const int H = P.H;
    const int sub_size = SUB ? SUB : P.sub_size;
    for (int i = j; i < P.ncol; i += NT) {
        uint32_t c = P.pal_rgb8[i];
        if ((uint32_t)i == P.slot_ci) c = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        const uint32_t r = c & 0xff, g = (c >> 8) & 0xff, b = (c >> 16) & 0xff;
        s_ent[i] = make_uint4(r | (b << 16), (8u * (1024u + r)) | ((8u * (1534u - r)) << 16), g << 7, c);
        if (PERC) {
            const float *src = ((uint32_t)i == P.slot_ci) ? P.cand_lab + 3 * (size_t)cand : P.pal_lab + 3 * (size_t)i;
            s_lab[3 * i] = src[0]; s_lab[3 * i + 1] = src[1]; s_lab[3 * i + 2] = src[2];
        }
    }
    if (PERC) for (int i = j; i < 256; i += NT) s_eotf[i] = P.lab_eotf[i];
    for (int i = j; i < 1024; i += NT) s_tile[i] = P.tile_pal[i];
    for (int q = 0; q < 4; q++) { ring[j][q][0] = 0.0; ring[j][q][1] = 0.0; ring[j][q][2] = 0.0; }
    __syncthreads();
    const double w0 = 7.0 / 16.0, w1 = 3.0 / 16.0, w2 = 5.0 / 16.0, w3 = 1.0 / 16.0, mult = 0.8;
    uint8_t *map = P.maps + (size_t)cand * W * H;
    uint32_t *map4 = reinterpret_cast<uint32_t *>(map);
    uint32_t *mapC4 = P.mapsC4 ? reinterpret_cast<uint32_t *>(P.mapsC4 + (size_t)cand * W * H) : nullptr;
    int y0 = 0; // first row to dither
    if (MODE == 2) {
        const int g0 = P.first_group[P.first_k0 + cand];
        y0 = min(4 * g0, H);
        // rows above are B's
        const uint32_t *b4 = reinterpret_cast<const uint32_t *>(P.bmap), *bC4 = reinterpret_cast<const uint32_t *>(P.bmapC4);
        for (int i = j; i < y0 * (W >> 2); i += NT) map4[i] = b4[i];
        if (mapC4) for (int i = j; i < y0 * (W >> 2); i += NT) { const int q = i / y0, yy = i - q * y0; mapC4[q * H + yy] = bC4[q * H + yy]; }
        if (y0 >= H) return;
    }
    // MODE 2: the row above the first row is B's.  Its state (the checkpoint of the group) reaches thread 0 the way any row
    // above does, through the ring of the thread "above" it — the block's last thread, whose own rows start 2 (NT - 1) = 254
    // steps in: until then it plays row y0 - 1, two columns ahead of thread 0 like every upper row, writing B's checkpoint
    // instead of dithering.  No special case in the step, nothing extra in registers.
    const bool ck_feed = MODE == 2 && j == NT - 1 && y0 > 0;
    const double *ckp = MODE == 2 ? P.ck_in + (size_t)(y0 >> 2) * W * 3 : nullptr;
    double ck_n[3] = {0.0, 0.0, 0.0}; // column t + 2 of the checkpoint, fetched a step ahead
    if (ck_feed) {
        for (int c = 0; c < 3; c++) { ring[j][0][c] = ckp[c] * mult; ring[j][1][c] = ckp[3 + c] * mult; ck_n[c] = ckp[6 + c] * mult; }
    }
    __syncthreads();
    const int nrows = H - y0;
    const int rows_per_thread = (nrows + NT - 1) / NT; // at most H / NT
    const int nth = nrows < NT ? nrows : NT;
    const int total_steps = 2 * (nth - 1) + rows_per_thread * W;
    double left[3] = {0.0, 0.0, 0.0}; // v(x-1, y) of this thread's current row
    uint32_t macc = 0; // the four map bytes of the current x quad: one word store instead of scattered byte stores
    const int up = (j + NT - 1) & (NT - 1); // thread owning row y-1
    // the source pixels of the NEXT x quad are fetched while the current quad is processed (the step is a dependent chain)
    const uint4 *orig4 = reinterpret_cast<const uint4 *>(P.orig);
    uint4 o_cur = make_uint4(0, 0, 0, 0), o_nxt = (y0 + j < H) ? orig4[(size_t)(y0 + j) * (W >> 2)] : make_uint4(0, 0, 0, 0);
    // MODE 2: B's records of the same pixels, fetched with the source pixels (two 16-byte words per x quad).  The
    // perturbation a won pixel injects fades by 0.8 per row, so away from it the run's dithered target rounds to what B's
    // rounded to — and for the same rounded target the nearest-entry search over the fourteen other entries has B's result:
    // only the candidate's own colour has to be tested against the key B recorded (one key instead of fifteen, exact).
    const bool use_rec = MODE == 2 && P.rec_in != nullptr;
    const uint4 *rec4 = reinterpret_cast<const uint4 *>(P.rec_in);
    uint4 ra_cur = make_uint4(0, 0, 0, 0), rb_cur = ra_cur, ra_nxt = ra_cur, rb_nxt = ra_cur;
    if (use_rec && y0 + j < H) { ra_nxt = rec4[(size_t)(y0 + j) * (W >> 1)]; rb_nxt = rec4[(size_t)(y0 + j) * (W >> 1) + 1]; }
    for (int t = 0; t < total_steps; t++) {
        const int local = t - 2 * j; // position in this thread's 512-pixel stream
        const int x = local & (W - 1), y = y0 + j + NT * (local >> 8);
        const bool act = local >= 0 && local < rows_per_thread * W && y < H;
        if (act) {
            // every operand is fetched unconditionally and the border cases are selects, so the step has no divergent
            // branches and its LDS reads are all in flight together
            const double *ru0 = ring[up][(x - 1) & 3], *ru1 = ring[up][x & 3], *ru2 = ring[up][(x + 1) & 3];
            const int base = (int)s_tile[(x >> 3) + (y >> 3) * (W >> 3)] * sub_size;
            double r0[3], r1[3], r2[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { r0[c] = ru0[c]; r1[c] = ru1[c]; r2[c] = ru2[c]; }
            if ((x & 3) == 0) { // this thread's stream continues with (x+4, y), then row y+128
                o_cur = o_nxt;
                const int ln = local + 4;
                const int xn = ln & (W - 1), yn = y0 + j + NT * (ln >> 8);
                if (ln < rows_per_thread * W && yn < H) o_nxt = orig4[((size_t)yn * W + xn) >> 2];
                if (use_rec) {
                    ra_cur = ra_nxt; rb_cur = rb_nxt;
                    if (ln < rows_per_thread * W && yn < H) { const size_t q2 = ((size_t)yn * W + xn) >> 1; ra_nxt = rec4[q2]; rb_nxt = rec4[q2 + 1]; }
                }
            }
            const uint32_t o = (x & 2) ? ((x & 1) ? o_cur.w : o_cur.z) : ((x & 1) ? o_cur.y : o_cur.x);
            const bool opaque = (o >> 24) != 0;
            const bool hasU = y > 0, hasL = x > 0, hasUL = hasU && hasL, hasUR = hasU && (x + 1 < W);
            double e[3], target[3];
            uint32_t tq[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double acc = 0.0;
                // (the ring and `left` hold value * 0.8 — lib.rs:477-496 multiplies every diffused value by the error multiplier
                // first, then by the neighbour's weight: the first product is the same for the four neighbours it goes to)
                const double a0 = acc + r0[c] * w3; acc = hasUL ? a0 : acc;
                const double a1 = acc + r1[c] * w2; acc = hasU ? a1 : acc;
                const double a2 = acc + r2[c] * w1; acc = hasUR ? a2 : acc;
                const double a3 = acc + left[c] * w0; acc = hasL ? a3 : acc;
                e[c] = acc;
                target[c] = (double)((o >> (8 * c)) & 0xff) + acc;
                const double cl = fmin(fmax(target[c], 0.0), 255.0);
                const double tr = trunc(cl); // Rust f64::round: half away from zero (cl >= 0)
                tq[c] = (uint32_t)(tr + ((cl - tr >= 0.5) ? 1.0 : 0.0));
            }
            int best = 0;
            uint32_t key_min = 0; // MODE 1: key of the nearest entry
            bool searched = false;
            if (use_rec) {
                const uint32_t rlo = (x & 2) ? ((x & 1) ? rb_cur.z : rb_cur.x) : ((x & 1) ? ra_cur.z : ra_cur.x);
                const uint32_t rhi = (x & 2) ? ((x & 1) ? rb_cur.w : rb_cur.y) : ((x & 1) ? ra_cur.w : ra_cur.y);
                const bool same = (rlo & 0x00ffffffu) == (tq[0] | (tq[1] << 8) | (tq[2] << 16));
                if (!__any(!same)) { // every row of the wave is back on B's targets at its pixel: B's choice, or the candidate's colour where it beats B's key
                    uint32_t kc;
                    if (PERC) { // (the record holds the bits of the distance to beat: one CIEDE2000 evaluation instead of sub_size)
                        const Lab tl = linear_to_lab(s_eotf[tq[0]], s_eotf[tq[1]], s_eotf[tq[2]]);
                        Lab el; el.l = s_lab[3 * P.slot_ci]; el.a = s_lab[3 * P.slot_ci + 1]; el.b = s_lab[3 * P.slot_ci + 2];
                        kc = __float_as_uint(ciede2000(el, tl));
                    } else {
                        const uint32_t t1 = tq[0] | (tq[2] << 16), tw = (8u * tq[0]) | ((0u - 8u * tq[0]) << 16);
                        kc = dither_group_min<1>(s_ent + P.slot_ci, t1, tw, (int)(tq[1] << 7)) >> 3;
                    }
                    const int bb = (int)(rlo >> 24) - base; // (a transparent pixel's record holds ncol + 1: its index is not used)
                    best = (kc < rhi) ? (int)P.slot_ci - base : (opaque ? bb : 0); // rhi = 0 outside the slot's subpalette: never beaten
                    searched = true;
                }
            }
            if (searched) {
            } else if (!PERC) {
                const uint32_t t1 = tq[0] | (tq[2] << 16), tw = (8u * tq[0]) | ((0u - 8u * tq[0]) << 16);
                const int tg = (int)(tq[1] << 7);
                uint32_t bk = 0xffffffffu; int bbase = 0;
                if (SUB) {
#pragma unroll
                    for (int i0 = 0; i0 < SUB; i0 += 8) {
                        const uint32_t g = (SUB - i0 >= 8) ? dither_group_min<8>(s_ent + base + i0, t1, tw, tg)
                                                           : dither_group_min<(SUB & 7) ? (SUB & 7) : 8>(s_ent + base + i0, t1, tw, tg);
                        if (i0 == 0 || (g >> 3) < (bk >> 3)) { bk = g; bbase = i0; }
                    }
                    best = bbase + (int)(bk & 7);
                    key_min = bk >> 3;
                } else {
                    int i0 = 0;
                    for (; i0 + 8 <= sub_size; i0 += 8) {
                        const uint32_t g = dither_group_min<8>(s_ent + base + i0, t1, tw, tg);
                        if (i0 == 0 || (g >> 3) < (bk >> 3)) { bk = g; bbase = i0; }
                    }
                    best = bbase + (int)(bk & 7);
                    uint32_t bkey = bk >> 3;
                    for (; i0 < sub_size; i0++) { // ragged tail, one entry at a time
                        const uint32_t g = dither_group_min<1>(s_ent + base + i0, t1, tw, tg) >> 3;
                        if (i0 == 0 || g < bkey) { bkey = g; best = i0; }
                    }
                    key_min = bkey;
                }
            } else {
                Lab tl = linear_to_lab(s_eotf[tq[0]], s_eotf[tq[1]], s_eotf[tq[2]]);
                float bd = 0.0f;
                for (int i = 0; i < sub_size; i++) {
                    Lab el; el.l = s_lab[3 * (base + i)]; el.a = s_lab[3 * (base + i) + 1]; el.b = s_lab[3 * (base + i) + 2];
                    float d = ciede2000(el, tl);
                    if (i == 0 || d < bd) { bd = d; best = i; }
                }
                if (MODE == 1) {
                    key_min = __float_as_uint(bd); // (a distance is never negative: its bits order as the values do, and bits + 1 is the next value up)
                    float *rl = P.rec_lab + 3 * ((size_t)y * W + x);
                    rl[0] = tl.l; rl[1] = tl.a; rl[2] = tl.b;
                }
            }
            if (MODE == 1) { // B: the slot's entry is a stand-in for j0 (same colour, hence same key and same diffused error)
                const bool in_sub = base == P.excl_sub * sub_size;
                if (in_sub && best == P.excl_si) best = P.excl_j0;
                const uint32_t thr = (in_sub && opaque) ? (sub_size == 1 ? 0xffffffffu : key_min + (P.excl_si < best ? 1u : 0u)) : 0u;
                const uint32_t ci = opaque ? (uint32_t)(base + best) : (uint32_t)P.ncol + 1u;
                P.rec_pack[(size_t)y * W + x] = (unsigned long long)(tq[0] | (tq[1] << 8) | (tq[2] << 16) | (ci << 24)) | ((unsigned long long)thr << 32);
            }
            const uint8_t m = opaque ? (uint8_t)best : 0;
            macc = (macc >> 8) | ((uint32_t)m << 24);
            if ((x & 3) == 3) {
                const size_t px = (size_t)y * W + x;
                map4[px >> 2] = macc;
                if (mapC4) mapC4[idx_c4(x & ~3, y, H) >> 2] = macc;
            }
            const uint32_t nc = s_ent[base + best].w;
            double v[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double d = target[c] - (double)((nc >> (8 * c)) & 0xff);
                v[c] = opaque ? d : e[c]; // transparent pixels forward their incoming error (lib.rs:469-474)
                left[c] = v[c] * mult;
            }
            // Row y-1 is two columns ahead (t = x + 2y): at this step it writes slot (x+2)&3 while this thread read slots
            // (x-1), x, (x+1) & 3 — never the same slot, so one barrier per step orders everything.
            ring[j][x & 3][0] = left[0]; ring[j][x & 3][1] = left[1]; ring[j][x & 3][2] = left[2];
            if (MODE == 1 && (y & 3) == 3 && y + 1 < H) { double *co = P.ck_out + ((size_t)((y + 1) >> 2) * W + x) * 3; co[0] = v[0]; co[1] = v[1]; co[2] = v[2]; }
        }
        if (MODE == 2 && ck_feed && t + 2 < W) { // row y0 - 1, column t + 2 (this thread's own stream starts at step 2 (NT - 1) = W - 2)
            const int xc = t + 2;
            ring[j][xc & 3][0] = ck_n[0]; ring[j][xc & 3][1] = ck_n[1]; ring[j][xc & 3][2] = ck_n[2];
            if (xc + 1 < W) { ck_n[0] = ckp[3 * (xc + 1)] * mult; ck_n[1] = ckp[3 * (xc + 1) + 1] * mult; ck_n[2] = ckp[3 * (xc + 1) + 2] * mult; }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// k_dither with a quad of lanes per row (RGB distance only).  The step of k_dither is one dependent chain of ~270
// instructions per row — three channels' diffused error and target one after the other, then a 15-entry search — and both
// the base image's run (766 steps of a single block) and the resumed runs are bound by its length.  Here lane q < 3 of a
// quad carries channel q (its ring column, its diffusion sum, its target and its residue), the four lanes split the
// entry search (entries q, q + 4, ...) and exchange their results over DPP quad permutes: ~80 instructions per step.
// Same arithmetic per channel and the same choice (lowest key, then lowest index), hence the same maps and records.
// 512 threads = 128 rows in flight, as in k_dither (and for the same reason: see NT there).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, int lane_in_quad) { // lane_in_quad: compile-time 0..2
    return lane_in_quad == 0 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, false)
         : lane_in_quad == 1 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, false)
                             : (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, false);
}
template <int SUB, int MODE, bool PERC = false>
__device__ __forceinline__ void dither4_body(const DitherParams &P, const int blk) {
    constexpr int NT = 128; // rows in flight
    __shared__ uint4 s_ent[256];
    __shared__ float s_lab[PERC ? 256 * 3 : 1]; // PERC: CIEDE2000 in Lab — the quad's lanes take entries q, q + 4, ... as below, ~2,000 instructions each
    __shared__ float s_eotf[PERC ? 256 : 1];
    __shared__ double ring[NT][4][3];
    __shared__ uint8_t s_tile[1024];
    if (P.skip && *P.skip) return;
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    const int c = q < 3 ? q : 2; // lane 3 shadows channel 2 (it only takes part in the entry search)
    const int cand = (MODE == 2 && P.order) ? P.order[blk] : blk;
    constexpr int W = 256;
    const int H = P.H;
    const int sub_size = SUB ? SUB : P.sub_size;
    for (int i = tid; i < P.ncol; i += 512) {
        uint32_t col = P.pal_rgb8[i];
        if ((uint32_t)i == P.slot_ci) col = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        const uint32_t r = col & 0xff, g = (col >> 8) & 0xff, b = (col >> 16) & 0xff;
        s_ent[i] = make_uint4(r | (b << 16), (8u * (1024u + r)) | ((8u * (1534u - r)) << 16), g << 7, col);
        if (PERC) {
            const float *src = ((uint32_t)i == P.slot_ci) ? P.cand_lab + 3 * (size_t)cand : P.pal_lab + 3 * (size_t)i;
            s_lab[3 * i] = src[0]; s_lab[3 * i + 1] = src[1]; s_lab[3 * i + 2] = src[2];
        }
    }
    if (PERC) for (int i = tid; i < 256; i += 512) s_eotf[i] = P.lab_eotf[i];
    for (int i = tid; i < 1024; i += 512) s_tile[i] = P.tile_pal[i];
    if (q < 3) for (int s4 = 0; s4 < 4; s4++) ring[j][s4][q] = 0.0;
    __syncthreads();
    const double w0 = 7.0 / 16.0, w1 = 3.0 / 16.0, w2 = 5.0 / 16.0, w3 = 1.0 / 16.0, mult = 0.8;
    uint8_t *map = P.maps + (size_t)cand * W * H;
    uint32_t *map4 = reinterpret_cast<uint32_t *>(map);
    uint32_t *mapC4 = P.mapsC4 ? reinterpret_cast<uint32_t *>(P.mapsC4 + (size_t)cand * W * H) : nullptr;
    int y0 = 0;
    if (MODE == 2) {
        const int g0 = P.first_group[P.first_k0 + cand];
        y0 = min(4 * g0, H);
        const uint32_t *b4 = reinterpret_cast<const uint32_t *>(P.bmap), *bC4 = reinterpret_cast<const uint32_t *>(P.bmapC4);
        for (int i = tid; i < y0 * (W >> 2); i += 512) map4[i] = b4[i];
        if (mapC4) for (int i = tid; i < y0 * (W >> 2); i += 512) { const int qq = i / y0, yy = i - qq * y0; mapC4[qq * H + yy] = bC4[qq * H + yy]; }
        if (y0 >= H) return;
    }
    // MODE 2: B's checkpoint row reaches row-thread 0 through the ring of the last row-thread (see k_dither)
    const bool ck_feed = MODE == 2 && j == NT - 1 && y0 > 0 && q < 3;
    const double *ckp = MODE == 2 ? P.ck_in + (size_t)(y0 >> 2) * W * 3 : nullptr;
    double ck_n = 0.0;
    if (ck_feed) { ring[j][0][q] = ckp[q] * mult; ring[j][1][q] = ckp[3 + q] * mult; ck_n = ckp[6 + q] * mult; }
    __syncthreads();
    const int nrows = H - y0;
    const int rows_per_thread = (nrows + NT - 1) / NT;
    const int nth = nrows < NT ? nrows : NT;
    const int total_steps = 2 * (nth - 1) + rows_per_thread * W;
    double left = 0.0; // v(x-1, y), this lane's channel
    uint32_t macc = 0;
    const int up = (j + NT - 1) & (NT - 1);
    const uint4 *orig4 = reinterpret_cast<const uint4 *>(P.orig);
    uint4 o_cur = make_uint4(0, 0, 0, 0), o_nxt = (y0 + j < H) ? orig4[(size_t)(y0 + j) * (W >> 2)] : make_uint4(0, 0, 0, 0);
    for (int t = 0; t < total_steps; t++) {
        const int local = t - 2 * j;
        const int x = local & (W - 1), y = y0 + j + NT * (local >> 8);
        const bool act = local >= 0 && local < rows_per_thread * W && y < H; // the same for the four lanes of a quad
        if (act) {
            const double r0 = ring[up][(x - 1) & 3][c], r1 = ring[up][x & 3][c], r2 = ring[up][(x + 1) & 3][c];
            const int base = (int)s_tile[(x >> 3) + (y >> 3) * (W >> 3)] * sub_size;
            if ((x & 3) == 0) {
                o_cur = o_nxt;
                const int ln = local + 4;
                const int xn = ln & (W - 1), yn = y0 + j + NT * (ln >> 8);
                if (ln < rows_per_thread * W && yn < H) o_nxt = orig4[((size_t)yn * W + xn) >> 2];
            }
            const uint32_t o = (x & 2) ? ((x & 1) ? o_cur.w : o_cur.z) : ((x & 1) ? o_cur.y : o_cur.x);
            const bool opaque = (o >> 24) != 0;
            const bool hasU = y > 0, hasL = x > 0, hasUL = hasU && hasL, hasUR = hasU && (x + 1 < W);
            double acc = 0.0;
            const double a0 = acc + r0 * w3; acc = hasUL ? a0 : acc; // (ring and `left` hold value * 0.8: see dither_body)
            const double a1 = acc + r1 * w2; acc = hasU ? a1 : acc;
            const double a2 = acc + r2 * w1; acc = hasUR ? a2 : acc;
            const double a3 = acc + left * w0; acc = hasL ? a3 : acc;
            const double e = acc;
            const double target = (double)((o >> (8 * c)) & 0xff) + acc;
            const double cl = fmin(fmax(target, 0.0), 255.0);
            const double tr = trunc(cl); // Rust f64::round: half away from zero (cl >= 0)
            const uint32_t tqc = (uint32_t)(tr + ((cl - tr >= 0.5) ? 1.0 : 0.0));
            const uint32_t tq0 = quad_bcast(tqc, 0), tq1 = quad_bcast(tqc, 1), tq2 = quad_bcast(tqc, 2);
            // the quad's lanes take entries q, q + 4, ...: 8 * key each (dither_group_min's arithmetic), first minimum kept
            const uint32_t t1 = tq0 | (tq2 << 16), tw = (8u * tq0) | ((0u - 8u * tq0) << 16);
            const int tg = (int)(tq1 << 7);
            uint32_t bk = 0xffffffffu, bi = 0xffu;
            Lab tl{0.0f, 0.0f, 0.0f};
            if (PERC) { // the bits of a distance (never negative) order as the distances do: the quad's reduction below takes them as keys
                tl = linear_to_lab(s_eotf[tq0], s_eotf[tq1], s_eotf[tq2]);
                for (int idx = q; idx < sub_size; idx += 4) {
                    Lab el; el.l = s_lab[3 * (base + idx)]; el.a = s_lab[3 * (base + idx) + 1]; el.b = s_lab[3 * (base + idx) + 2];
                    const uint32_t kd = __float_as_uint(ciede2000(el, tl));
                    if (kd < bk) { bk = kd; bi = (uint32_t)idx; }
                }
            } else
#pragma unroll 4
            for (int i = 0; i < (SUB ? (SUB + 3) / 4 : 64); i++) {
                const int idx = q + 4 * i;
                if (idx >= sub_size) break;
                const uint4 en = s_ent[base + idx];
                const u16x2 d = as_u16x2(en.x) - as_u16x2(t1);
                const u16x2 sq = d * d;
                const u16x2 wg = as_u16x2(en.y) + as_u16x2(tw);
                const int dg = (int)en.z - tg; // (times 128)
                const uint32_t k8 = __builtin_amdgcn_udot2(sq, wg, (uint32_t)__mul24(dg, dg), false);
                if (k8 < bk) { bk = k8; bi = (uint32_t)idx; }
            }
            { // lowest key, then lowest index, over the quad
                uint32_t pk = (uint32_t)__builtin_amdgcn_mov_dpp((int)bk, 0xB1, 0xf, 0xf, false), pi = (uint32_t)__builtin_amdgcn_mov_dpp((int)bi, 0xB1, 0xf, 0xf, false);
                if (pk < bk || (pk == bk && pi < bi)) { bk = pk; bi = pi; }
                pk = (uint32_t)__builtin_amdgcn_mov_dpp((int)bk, 0x4E, 0xf, 0xf, false); pi = (uint32_t)__builtin_amdgcn_mov_dpp((int)bi, 0x4E, 0xf, 0xf, false);
                if (pk < bk || (pk == bk && pi < bi)) { bk = pk; bi = pi; }
            }
            int best = (int)bi;
            const uint32_t key_min = PERC ? bk : bk >> 3;
            if (MODE == 1) { // B: the slot's entry is a stand-in for j0 (same colour, hence same key and same diffused error)
                const bool in_sub = base == P.excl_sub * sub_size;
                if (in_sub && best == P.excl_si) best = P.excl_j0;
                if (PERC && q == 0) { float *rl = P.rec_lab + 3 * ((size_t)y * W + x); rl[0] = tl.l; rl[1] = tl.a; rl[2] = tl.b; }
                if (q == 0) {
                    const uint32_t thr = (in_sub && opaque) ? (sub_size == 1 ? 0xffffffffu : key_min + (P.excl_si < best ? 1u : 0u)) : 0u;
                    const uint32_t ci = opaque ? (uint32_t)(base + best) : (uint32_t)P.ncol + 1u;
                    P.rec_pack[(size_t)y * W + x] = (unsigned long long)(tq0 | (tq1 << 8) | (tq2 << 16) | (ci << 24)) | ((unsigned long long)thr << 32);
                }
            }
            if (q == 0) {
                const uint8_t m = opaque ? (uint8_t)best : 0;
                macc = (macc >> 8) | ((uint32_t)m << 24);
                if ((x & 3) == 3) {
                    const size_t px = (size_t)y * W + x;
                    map4[px >> 2] = macc;
                    if (mapC4) mapC4[idx_c4(x & ~3, y, H) >> 2] = macc;
                }
            }
            const uint32_t nc = s_ent[base + best].w;
            const double dres = target - (double)((nc >> (8 * c)) & 0xff);
            const double v = opaque ? dres : e; // transparent pixels forward their incoming error (lib.rs:469-474)
            left = v * mult;
            if (q < 3) {
                ring[j][x & 3][q] = left;
                if (MODE == 1 && (y & 3) == 3 && y + 1 < H) P.ck_out[((size_t)((y + 1) >> 2) * W + x) * 3 + q] = v;
            }
        }
        if (MODE == 2 && ck_feed && t + 2 < W) {
            const int xc = t + 2;
            ring[j][xc & 3][q] = ck_n;
            if (xc + 1 < W) ck_n = ckp[3 * (xc + 1) + q] * mult;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Resumed runs (MODE 2, RGB distance), one WAVE per run: lane j owns rows y0 + j, y0 + j + 64, ...  With thousands of runs
// in a launch the chip is short of issue slots and of waves to hide a step's latency behind, not of parallelism inside a
// run, and k_dither's 128 rows in flight cost it two waves that meet at a barrier every step, a 12 KB ring in LDS per run
// and — for a run of r rows — 2 (min(r,128) - 1) + 256 ceil(r / 128) steps of both waves.  Here
//  * the row above is the lane below: v(x+1, y-1) * 0.8 arrives over DPP (wave_shr:1) one step after lane j-1 formed it,
//    and the lane keeps the three columns it needs in registers; no ring, no barrier after the table is loaded;
//  * lane 0's row above is lane 63's row of the previous pass, 130 steps earlier: a 256-column buffer in LDS per wave,
//    written by lane 63 alone (LDS operations of one wave execute in order) and prefilled with B's checkpoint row;
//  * the four runs of a block share one entry table: the slot's entry holds the colour of its stand-in j0 as in B's
//    own run (MODE 1 above), the search returns the best OTHER entry and its key, and the run's own colour is one more
//    key from registers, compared under the same rule (lowest key, then lowest index): the same choice as a search of a
//    table holding the run's colour.  B's record replaces the search where the run's dithered target rounds to B's.
// 2 * 63 + 256 ceil(r / 64) steps of one wave for a run of r rows: 30 % fewer wave-steps over resume rows drawn evenly,
// 29 KB of LDS per four runs.  The border cases multiply by a weight of zero instead of skipping the addition: every
// value is finite, the sums start at +0.0 and (+0.0) + (-0.0) = +0.0, so the diffused error is bit for bit the one above.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp_up(double wrap, double mine) { // lane j: lane j-1's `mine`; lane 0: `wrap`
    const uint64_t w = (uint64_t)__double_as_longlong(wrap), m = (uint64_t)__double_as_longlong(mine);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)w, (int)(uint32_t)m, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(w >> 32), (int)(uint32_t)(m >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
template <int SUB>
__device__ __forceinline__ void ditherw_body(const DitherParams &P, const int blk, const int nrun) {
    constexpr int W = 256, RW = 4; // runs (waves) per block
    __shared__ uint4 s_ent[256];
    __shared__ uint8_t s_tile[1024];
    __shared__ double s_wrap[RW][W][3];
    if (P.skip && *P.skip) return;
    const int tid = threadIdx.x, wv = tid >> 6, j = tid & 63;
    const int H = P.H;
    const int sub_size = SUB ? SUB : P.sub_size; // >= 2: a one-entry subpalette has no stand-in (the caller keeps k_dither for it)
    const int slot_sub = (int)P.slot_ci / sub_size, slot_si = (int)P.slot_ci - slot_sub * sub_size, slot_j0 = (slot_si + 1) % sub_size;
    for (int i = tid; i < P.ncol; i += 64 * RW) {
        const uint32_t c = P.pal_rgb8[(uint32_t)i == P.slot_ci ? slot_sub * sub_size + slot_j0 : i];
        const uint32_t r = c & 0xff, g = (c >> 8) & 0xff, b = (c >> 16) & 0xff;
        s_ent[i] = make_uint4(r | (b << 16), (8u * (1024u + r)) | ((8u * (1534u - r)) << 16), g << 7, c);
    }
    for (int i = tid; i < 1024; i += 64 * RW) s_tile[i] = P.tile_pal[i];
    __syncthreads(); // the only one: from here on the waves go their own ways
    const int run = blk * RW + wv;
    if (run >= nrun) return;
    const int cand = P.order ? P.order[run] : run;
    uint4 own; // the run's colour, as a table entry
    { const uint32_t c = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]); const uint32_t r = c & 0xff, g = (c >> 8) & 0xff, b = (c >> 16) & 0xff;
      own = make_uint4(r | (b << 16), (8u * (1024u + r)) | ((8u * (1534u - r)) << 16), g << 7, c); }
    const double w0 = 7.0 / 16.0, w1 = 3.0 / 16.0, w2 = 5.0 / 16.0, w3 = 1.0 / 16.0, mult = 0.8;
    uint8_t *map = P.maps + (size_t)cand * W * H;
    uint32_t *map4 = reinterpret_cast<uint32_t *>(map);
    uint32_t *mapC4 = P.mapsC4 ? reinterpret_cast<uint32_t *>(P.mapsC4 + (size_t)cand * W * H) : nullptr;
    const int g0 = P.first_group[P.first_k0 + cand];
    const int y0 = min(4 * g0, H);
    { // rows above are B's
        const uint32_t *b4 = reinterpret_cast<const uint32_t *>(P.bmap), *bC4 = reinterpret_cast<const uint32_t *>(P.bmapC4);
        for (int i = j; i < y0 * (W >> 2); i += 64) map4[i] = b4[i];
        if (mapC4) for (int i = j; i < y0 * (W >> 2); i += 64) { const int q = i / y0, yy = i - q * y0; mapC4[q * H + yy] = bC4[q * H + yy]; }
        if (y0 >= H) return;
    }
    double (*wrap)[3] = s_wrap[wv];
    { // row y0 - 1: B's checkpoint of the group (times 0.8, as every value on its way down), or nothing above row 0
        const double *ckp = P.ck_in + (size_t)(y0 >> 2) * W * 3;
        for (int i = j; i < W * 3; i += 64) (&wrap[0][0])[i] = y0 > 0 ? ckp[i] * mult : 0.0;
    }
    const int nrows = H - y0;
    const int passes = (nrows + 63) >> 6;
    const int nth = nrows < 64 ? nrows : 64;
    const int total_steps = 2 * (nth - 1) + passes * W;
    double left[3] = {0.0, 0.0, 0.0}; // v(x-1, y) * 0.8: this lane's left neighbour, and what lane j+1 fetches as its row above in the next step
    double u[3][3];                   // [age][channel]: the row above at columns x-1, x, x+1 (rotating: see the loop)
#pragma unroll
    for (int a = 0; a < 3; a++) { u[a][0] = 0.0; u[a][1] = 0.0; u[a][2] = 0.0; }
    double wnx[3] = {wrap[0][0], wrap[0][1], wrap[0][2]}; // lane 0's row above, column (t + 1) & 255, read a step ahead: column 0 for the step before the first
    uint32_t macc = 0;
    const uint4 *orig4 = reinterpret_cast<const uint4 *>(P.orig);
    uint4 o_cur = make_uint4(0, 0, 0, 0), o_nxt = (y0 + j < H) ? orig4[(size_t)(y0 + j) * (W >> 2)] : make_uint4(0, 0, 0, 0);
    const bool use_rec = P.rec_in != nullptr;
    const uint4 *rec4 = reinterpret_cast<const uint4 *>(P.rec_in);
    uint4 ra_cur = make_uint4(0, 0, 0, 0), rb_cur = ra_cur, ra_nxt = ra_cur, rb_nxt = ra_cur;
    if (use_rec && y0 + j < H) { ra_nxt = rec4[(size_t)(y0 + j) * (W >> 1)]; rb_nxt = rec4[(size_t)(y0 + j) * (W >> 1) + 1]; }
    // Before step t lane j holds columns x-1, x of the row above and fetches x+1 = what lane j-1 formed in step t-1.  The
    // lane's first step is t = 2j; it has shifted through the steps before (the row above started two steps earlier).
    // One "step before the first" hands column 0 to the lanes' registers: lane j-1's `left` is still zero then, which is
    // right for every lane but lane 0 — whose row above is in `wrap` — so the loop starts at t = -1 with nothing active.
    for (int t3 = -1; t3 < total_steps; t3 += 3) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int t = t3 + k;
            // u[(k+2)%3] <- column x+1 of the row above; u[k%3] is x-1, u[(k+1)%3] is x   (k = t + 1 mod 3)
            double *um = u[k % 3], *uc = u[(k + 1) % 3], *up = u[(k + 2) % 3];
#pragma unroll
            for (int c = 0; c < 3; c++) up[c] = dpp_up(wnx[c], left[c]);
            { const double *wn = wrap[(t + 2) & (W - 1)]; wnx[0] = wn[0]; wnx[1] = wn[1]; wnx[2] = wn[2]; } // for the next step
            const int local = t - 2 * j;
            const int x = local & (W - 1), pass = local >> 8, y = y0 + j + 64 * pass;
            const bool act = t >= 0 && t < total_steps && local >= 0 && pass < passes && y < H;
            if (act) {
                const int base = (int)s_tile[(x >> 3) + (y >> 3) * (W >> 3)] * sub_size;
                if ((x & 3) == 0) {
                    o_cur = o_nxt;
                    const int ln = local + 4;
                    const int xn = ln & (W - 1), yn = y0 + j + 64 * (ln >> 8);
                    const bool more = (ln >> 8) < passes && yn < H;
                    if (more) o_nxt = orig4[((size_t)yn * W + xn) >> 2];
                    if (use_rec) {
                        ra_cur = ra_nxt; rb_cur = rb_nxt;
                        if (more) { const size_t q2 = ((size_t)yn * W + xn) >> 1; ra_nxt = rec4[q2]; rb_nxt = rec4[q2 + 1]; }
                    }
                }
                const uint32_t o = (x & 2) ? ((x & 1) ? o_cur.w : o_cur.z) : ((x & 1) ? o_cur.y : o_cur.x);
                const bool opaque = (o >> 24) != 0;
                // (no row above row 0: `wrap` and the registers hold zeros there)
                const double w3x = x > 0 ? w3 : 0.0, w1x = x + 1 < W ? w1 : 0.0, w0x = x > 0 ? w0 : 0.0;
                double e[3], target[3];
                uint32_t tq[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double acc = 0.0;
                    acc = acc + um[c] * w3x;
                    acc = acc + uc[c] * w2;
                    acc = acc + up[c] * w1x;
                    acc = acc + left[c] * w0x;
                    e[c] = acc;
                    target[c] = (double)((o >> (8 * c)) & 0xff) + acc;
                    const double cl = fmin(fmax(target[c], 0.0), 255.0);
                    const double tr = trunc(cl); // Rust f64::round: half away from zero (cl >= 0)
                    tq[c] = (uint32_t)(tr + ((cl - tr >= 0.5) ? 1.0 : 0.0));
                }
                const uint32_t t1 = tq[0] | (tq[2] << 16), tw = (8u * tq[0]) | ((0u - 8u * tq[0]) << 16);
                const int tg = (int)(tq[1] << 7);
                const bool in_sub = base == slot_sub * sub_size;
                int best = 0;        // the best entry other than the slot's
                uint32_t thr = 0;    // the run's colour takes the pixel iff its key is below
                bool searched = false;
                if (use_rec) {
                    const uint32_t rlo = (x & 2) ? ((x & 1) ? rb_cur.z : rb_cur.x) : ((x & 1) ? ra_cur.z : ra_cur.x);
                    const uint32_t rhi = (x & 2) ? ((x & 1) ? rb_cur.w : rb_cur.y) : ((x & 1) ? ra_cur.w : ra_cur.y);
                    const bool same = (rlo & 0x00ffffffu) == (tq[0] | (tq[1] << 8) | (tq[2] << 16));
                    if (!__any(!same)) { best = opaque ? (int)(rlo >> 24) - base : 0; thr = rhi; searched = true; }
                }
                if (!searched) {
                    uint32_t bk = 0xffffffffu; int bbase = 0;
                    if (SUB) {
#pragma unroll
                        for (int i0 = 0; i0 < SUB; i0 += 8) {
                            const uint32_t g = (SUB - i0 >= 8) ? dither_group_min<8>(s_ent + base + i0, t1, tw, tg)
                                                               : dither_group_min<(SUB & 7) ? (SUB & 7) : 8>(s_ent + base + i0, t1, tw, tg);
                            if (i0 == 0 || (g >> 3) < (bk >> 3)) { bk = g; bbase = i0; }
                        }
                        best = bbase + (int)(bk & 7);
                        bk >>= 3;
                    } else {
                        int i0 = 0;
                        for (; i0 + 8 <= sub_size; i0 += 8) {
                            const uint32_t g = dither_group_min<8>(s_ent + base + i0, t1, tw, tg);
                            if (i0 == 0 || (g >> 3) < (bk >> 3)) { bk = g; bbase = i0; }
                        }
                        best = bbase + (int)(bk & 7);
                        bk >>= 3;
                        for (; i0 < sub_size; i0++) { // ragged tail, one entry at a time
                            const uint32_t g = dither_group_min<1>(s_ent + base + i0, t1, tw, tg) >> 3;
                            if (i0 == 0 || g < bk) { bk = g; best = i0; }
                        }
                    }
                    if (in_sub && best == slot_si) best = slot_j0; // the stand-in was found under the slot's index
                    thr = (in_sub && opaque) ? bk + (slot_si < best ? 1u : 0u) : 0u;
                }
                const uint32_t kc = dither_group_min<1>(&own, t1, tw, tg) >> 3;
                const bool mine = kc < thr;
                const uint32_t nc = mine ? own.w : s_ent[base + best].w;
                const uint8_t m = opaque ? (uint8_t)(mine ? slot_si : best) : 0;
                macc = (macc >> 8) | ((uint32_t)m << 24);
                if ((x & 3) == 3) {
                    const size_t px = (size_t)y * W + x;
                    map4[px >> 2] = macc;
                    if (mapC4) mapC4[idx_c4(x & ~3, y, H) >> 2] = macc;
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double d = target[c] - (double)((nc >> (8 * c)) & 0xff);
                    left[c] = (opaque ? d : e[c]) * mult; // transparent pixels forward their incoming error (lib.rs:469-474)
                }
                if (j == 63) { wrap[x][0] = left[0]; wrap[x][1] = left[1]; wrap[x][2] = left[2]; } // the row above lane 0's next pass
            }
        }
    }
}

// ---- the winner's map -----------------------------------------------------------------------------------
// Each lane remembers the first-lowest error it has scored in the current candidate list and that candidate's map.
struct BestRec { double err; int k; int pad; };
struct StepResult { double error; int32_t best_k; uint8_t rgb5[3]; uint8_t changed; };
__global__ void k_reset_best(BestRec *__restrict__ recs, int n) {
    if ((int)threadIdx.x < n) { recs[threadIdx.x].err = __longlong_as_double(0x7ff0000000000000ll); recs[threadIdx.x].k = -1; }
}
// errors of this chunk sit at errors[off + i*stride], i < nc; that position is the candidate's index in the step's list
__global__ __launch_bounds__(1024) void k_keep_best(const double *__restrict__ errors, int stride, int off, int nc, const uint8_t *__restrict__ maps, int npx, BestRec *__restrict__ rec,
                                                   uint8_t *__restrict__ bestmap) {
    __shared__ double s_e[1024];
    __shared__ int s_i[1024];
    const int t = threadIdx.x;
    double be = __longlong_as_double(0x7ff0000000000000ll); int bi = 0x7fffffff;
    for (int i = t; i < nc; i += 1024) { const double e = errors[off + (size_t)i * stride]; if (e < be) { be = e; bi = i; } }
    s_e[t] = be; s_i[t] = bi;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (t < st) {
            const double e2 = s_e[t + st]; const int i2 = s_i[t + st];
            if (e2 < s_e[t] || (e2 == s_e[t] && i2 < s_i[t])) { s_e[t] = e2; s_i[t] = i2; }
        }
        __syncthreads();
    }
    be = s_e[0]; bi = s_i[0];
    if (bi == 0x7fffffff || !(be < rec->err)) return; // chunks reach a lane in ascending order: strict < keeps the first index
    const uint4 *src = reinterpret_cast<const uint4 *>(maps + (size_t)bi * npx);
    uint4 *dst = reinterpret_cast<uint4 *>(bestmap);
    for (int i = t; i < npx / 16; i += 1024) dst[i] = src[i];
    __syncthreads(); // every thread has compared against the old record
    if (t == 0) { rec->err = be; rec->k = off + bi * stride; }
}
// sp.ahead (capi.hip): B's Floyd-Steinberg run for this call was made during the previous one; it stands iff that call's commit kept the palette
__global__ void k_ahead_ok(const StepResult *__restrict__ last, int *__restrict__ ok) { *ok = (last && last->changed) ? 0 : 1; } // (last = nullptr: no commit in between)
// After k_commit: if the winner was scored here, its map becomes the image's map and *skip = 1 (the re-dither is void);
// likewise when nothing was accepted and the stored map already belongs to the palette.
__global__ __launch_bounds__(1024) void k_take_best_map(const StepResult *__restrict__ last, const BestRec *__restrict__ recs, int nrec, const uint8_t *__restrict__ bestmaps, int npx,
                                                       int map_synced, uint8_t *__restrict__ map, int *__restrict__ skip) {
    const int best_k = last->best_k;
    int lane = -1;
    if (best_k >= 0) for (int r = 0; r < nrec; r++) if (recs[r].k == best_k) lane = r;
    if (lane >= 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bestmaps + (size_t)lane * npx);
        uint4 *dst = reinterpret_cast<uint4 *>(map);
        for (int i = threadIdx.x; i < npx / 16; i += 1024) dst[i] = src[i];
    }
    if (threadIdx.x == 0) *skip = (lane >= 0 || (best_k < 0 && map_synced)) ? 1 : 0;
}

// Row-blocked (R4: [y/4][x][y%4]) and, on request, transposed ([x][y]) copies of the row-major per-candidate maps.
// Both hold the same word — rows 4q..4q+3 of column x — at different places.
__global__ __launch_bounds__(256) void k_maps_relayout(const uint8_t *__restrict__ maps, int W, int H, uint32_t *__restrict__ r4, uint32_t *__restrict__ mT) {
    // block = one group of four rows (W = 256): thread t fetches the word of columns 4(t%64).. of row t/64, LDS turns the
    // 4x256 bytes around, thread t stores the word of column t (rows 4q..4q+3)
    __shared__ uint8_t s_b[4][256 + 4];
    const int q = blockIdx.x, t = threadIdx.x;
    const size_t cb = (size_t)blockIdx.y * W * H;
    const uint32_t w4 = reinterpret_cast<const uint32_t *>(maps + cb + (size_t)(4 * q + (t >> 6)) * W)[t & 63];
    *reinterpret_cast<uint32_t *>(&s_b[t >> 6][4 * (t & 63)]) = w4;
    __syncthreads();
    const uint32_t w = (uint32_t)s_b[0][t] | ((uint32_t)s_b[1][t] << 8) | ((uint32_t)s_b[2][t] << 16) | ((uint32_t)s_b[3][t] << 24);
    r4[(cb >> 2) + (size_t)q * W + t] = w;
    if (mT) mT[(cb >> 2) + (size_t)t * (H >> 2) + q] = w;
}

// ---- optimizer step: candidates and commit -------------------------------------------------------

__device__ __forceinline__ unsigned long long dev_mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// method 0: r,g,b sampled in that order from the counter RNG (lib.rs:206-208); 1: 32 values of one
// channel of the current colour (lib.rs:296-297); 2: the NES table (lib.rs:252-253)
// With `sel`: also the shard's own list (candidate k belongs to rank k % count, position k / count) and the error vector
// preset to +inf — what k_shard_select does for an explicit list.
__device__ __forceinline__ void gen_candidates_body(int method, int n, unsigned long long key, const uint8_t *__restrict__ colors, int slot, int channel, uint8_t *__restrict__ cand,
                                                    int rank, int count, uint8_t *__restrict__ sel, double *__restrict__ errors) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint8_t c[3];
    if (method == 0) {
        unsigned long long z = dev_mix64(key + ((unsigned long long)k + 1ull) * 0x9E3779B97F4A7C15ull);
        c[0] = (uint8_t)(z & 31); c[1] = (uint8_t)((z >> 5) & 31); c[2] = (uint8_t)((z >> 10) & 31);
    } else if (method == 1) {
        c[0] = colors[3 * slot]; c[1] = colors[3 * slot + 1]; c[2] = colors[3 * slot + 2];
        c[channel] = (uint8_t)k;
    } else {
        if (k < (int)kNesColorCount) { c[0] = kNesTableDev[k][0]; c[1] = kNesTableDev[k][1]; c[2] = kNesTableDev[k][2]; } else { c[0] = c[1] = c[2] = 0; }
    }
    cand[3 * k] = c[0]; cand[3 * k + 1] = c[1]; cand[3 * k + 2] = c[2];
    if (sel) {
        errors[k] = __longlong_as_double(0x7ff0000000000000ll);
        if (k % count == rank) { const int j = k / count; sel[3 * j] = c[0]; sel[3 * j + 1] = c[1]; sel[3 * j + 2] = c[2]; }
    }
}

// Keep candidates k with k % count == rank (compacted, in order); errors[] <- +inf everywhere.
__global__ void k_shard_select(const uint8_t *__restrict__ cand, int n, int rank, int count, uint8_t *__restrict__ sel, double *__restrict__ errors) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    errors[k] = __longlong_as_double(0x7ff0000000000000ll);
    if (k % count == rank) { int j = k / count; sel[3 * j] = cand[3 * k]; sel[3 * j + 1] = cand[3 * k + 1]; sel[3 * j + 2] = cand[3 * k + 2]; }
}

// Acceptance rule: ascending k, strict `<` against the running best starting from the incumbent
// (lib.rs:216-219, 302-305) or from f64::MAX for the NES method (lib.rs:250, 258-261).
struct PaletteTables { const float *eotf, *lab_eotf; uint32_t *rgb8; float *lin, *xyb, *lab; }; // lab == nullptr without --perceptual-palettes
// The decision of one optimizer call given the first-lowest error (min_e at index min_k; min_k = 0x7fffffff: no finite
// error) of its candidate list: accept it iff it is strictly below the incumbent (always, for the NES method), write
// the slot's colour and its rows of the palette tables, leave the committed state's error in *inc_err.  One thread.
__device__ __forceinline__ void commit_decide(const double min_e, const int min_k, const uint8_t *__restrict__ cand, uint8_t *__restrict__ colors, int slot, int nes, double *__restrict__ inc_err,
                                              StepResult *__restrict__ last, const PaletteTables &T) {
    const double start = nes ? 1.7976931348623157e308 : *inc_err;
    double best = start; int best_k = -1;
    if (min_k != 0x7fffffff && min_e < start) { best = min_e; best_k = min_k; }
    uint8_t c[3] = {colors[3 * slot], colors[3 * slot + 1], colors[3 * slot + 2]};
    uint8_t changed = 0;
    if (nes && best_k < 0) best_k = 0; // best_index = 0 (lib.rs:249)
    if (best_k >= 0) {
        uint8_t nc[3] = {cand[3 * best_k], cand[3 * best_k + 1], cand[3 * best_k + 2]};
        changed = (nc[0] != c[0] || nc[1] != c[1] || nc[2] != c[2]) ? 1 : 0;
        c[0] = nc[0]; c[1] = nc[1]; c[2] = nc[2];
        colors[3 * slot] = c[0]; colors[3 * slot + 1] = c[1]; colors[3 * slot + 2] = c[2];
        if (T.rgb8) { // the one changed row of the palette tables (k_palette_tables / k_palette_lab for entry `slot`)
            const uint32_t rgb8 = rgb5_to_rgb8(c[0], c[1], c[2]);
            const float r = T.eotf[rgb8 & 0xff], g = T.eotf[(rgb8 >> 8) & 0xff], b = T.eotf[(rgb8 >> 16) & 0xff];
            float X, Y, B;
            linear_to_positive_xyb(r, g, b, X, Y, B);
            T.rgb8[slot] = rgb8;
            T.lin[3 * slot] = r; T.lin[3 * slot + 1] = g; T.lin[3 * slot + 2] = b;
            T.xyb[3 * slot] = X; T.xyb[3 * slot + 1] = Y; T.xyb[3 * slot + 2] = B;
            if (T.lab) {
                const Lab l = linear_to_lab(T.lab_eotf[rgb8 & 0xff], T.lab_eotf[(rgb8 >> 8) & 0xff], T.lab_eotf[(rgb8 >> 16) & 0xff]);
                T.lab[3 * slot] = l.l; T.lab[3 * slot + 1] = l.a; T.lab[3 * slot + 2] = l.b;
            }
        }
        if (best < 1.7976931348623157e308) *inc_err = best; // error() of the committed state (lib.rs:910)
    }
    last->error = *inc_err; last->best_k = best_k; last->rgb5[0] = c[0]; last->rgb5[1] = c[1]; last->rgb5[2] = c[2]; last->changed = changed;
}

__device__ __forceinline__ void commit_body(const double *__restrict__ errors, int n, const uint8_t *__restrict__ cand, uint8_t *__restrict__ colors, int slot, int nes, double *__restrict__ inc_err,
                                            StepResult *__restrict__ last, const PaletteTables &T) {
    // The sequential scan "for k ascending: if e_k < best" ends on the FIRST index attaining the minimum, provided that
    // minimum is < the starting value; a parallel (error, index) lexicographic minimum gives the same answer.
    __shared__ double s_e[256];
    __shared__ int s_k[256];
    const int t = threadIdx.x;
    double be = __longlong_as_double(0x7ff0000000000000ll); int bk = 0x7fffffff;
    for (int k = t; k < n; k += 256) { const double e = errors[k]; if (e < be) { be = e; bk = k; } } // NaN never wins, as in the reference
    s_e[t] = be; s_k[t] = bk;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (t < st) {
            const double e2 = s_e[t + st]; const int k2 = s_k[t + st];
            if (e2 < s_e[t] || (e2 == s_e[t] && k2 < s_k[t])) { s_e[t] = e2; s_k[t] = k2; }
        }
        __syncthreads();
    }
    if (t != 0) return;
    commit_decide(s_e[0], s_k[0], cand, colors, slot, nes, inc_err, last, T);
}

// ---- deterministic-math probes for the bit-parity tests -------------------------------------------
__global__ void k_debug_math(int op, const float *__restrict__ x, const float *__restrict__ y, int n, const float *__restrict__ lab_eotf, float *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
    case 0: out[i] = d_sinf(x[i]); break;
    case 1: out[i] = d_cosf(x[i]); break;
    case 2: out[i] = d_expf_neg(x[i]); break;
    case 3: out[i] = d_cbrtf(x[i]); break;
    case 4: out[i] = d_atan2f(y[i], x[i]); break;
    case 5: { Lab a{x[3 * i], x[3 * i + 1], x[3 * i + 2]}, b{y[3 * i], y[3 * i + 1], y[3 * i + 2]}; out[i] = ciede2000(a, b); break; }
    default: {
        Lab l = linear_to_lab(lab_eotf[(int)x[3 * i] & 255], lab_eotf[(int)x[3 * i + 1] & 255], lab_eotf[(int)x[3 * i + 2] & 255]);
        out[3 * i] = l.l; out[3 * i + 1] = l.a; out[3 * i + 2] = l.b; break;
    }
    }
}

// ------------------------------------------------------------------------------------------------
// k-means (cogset 0.2.0 Kmeans::new restated): Lloyd on points [n][3] f64, one block per problem
// (a subpalette, or the tile means), every round inside ONE launch: no per-round launches, no host polling.
//
// What must not change is the order of three families of binary64 sums, because they decide the convergence test
// |objective - previous| < 1e-6 and the centres bit for bit: the objective adds the per-point costs in point order, and
// each centre adds its members' coordinates in point order.  What can be parallel is everything else:
//   * points are processed in chunks of 1,024 (one per thread): nearest centre (first minimum wins) and cost;
//   * a STABLE partition of the chunk by cluster (ranks from wave ballots, wave and cluster prefix sums in LDS) puts every
//     cluster's members of the chunk side by side, still in point order, so that
//   * thread c adds cluster c's members only — chains of n/k additions instead of the n masked additions per cluster of
//     the round-1 kernel (k_km_update: 271 us per round, 92 % of the GPU time of a 128-image initialisation) — while the
//     last thread of the block adds the chunk's costs.
// The ordered cost chain (1,024 dependent additions per chunk) is what a round now costs: ~45 us for 8,192 points.
// ------------------------------------------------------------------------------------------------
struct KmeansWork {
    double *pts = nullptr; uint32_t *assign = nullptr; double *centres = nullptr; int *rounds = nullptr;
    size_t cap_pts = 0; int cap_prob = 0, cap_k = 0;
    // small scratch the initialisers need on every call, kept with the context: hipMalloc / hipFree per call synchronise the
    // whole device, which serialises the host threads that initialise the images of a throughput batch side by side
    int *d_n = nullptr; long long *d_off = nullptr; float *d_sums = nullptr; int *d_counts = nullptr; uint32_t *d_index = nullptr;
    int cap_meta = 0, cap_tile = 0; size_t cap_index = 0;
};
inline void kmeans_free(KmeansWork &w) {
    if (w.pts) (void)hipFree(w.pts); if (w.assign) (void)hipFree(w.assign); if (w.centres) (void)hipFree(w.centres); if (w.rounds) (void)hipFree(w.rounds);
    if (w.d_n) (void)hipFree(w.d_n); if (w.d_off) (void)hipFree(w.d_off); if (w.d_sums) (void)hipFree(w.d_sums); if (w.d_counts) (void)hipFree(w.d_counts);
    if (w.d_index) (void)hipFree(w.d_index);
    w = KmeansWork{};
}

struct KmParams {
    const double *pts; uint32_t *assign; double *centres; int *rounds;
    const int *n; const long long *off; // per problem: point count, offset (in points) into pts/assign
    int k;
};
constexpr int kKmChunk = 1024;
__global__ __launch_bounds__(1024) void k_kmeans(KmParams P) {
    __shared__ double s_c[256 * 3];                 // centres of the round
    __shared__ double s_p[kKmChunk * 3];            // the chunk's points, cluster-major, point order inside a cluster
    __shared__ double s_cost[kKmChunk];
    __shared__ unsigned short s_wcnt[16][256];      // members of cluster c in wave w of the chunk, then their exclusive prefix over waves
    __shared__ int s_tot[256], s_start[256], s_wsum[4];
    __shared__ int s_stop;
    const int prob = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int n = P.n[prob], k = P.k;
    const double *pts = P.pts + 3 * P.off[prob];
    uint32_t *assign = P.assign + P.off[prob];
    double *centres = P.centres + (size_t)prob * k * 3;
    for (int i = t; i < k * 3; i += 1024) s_c[i] = centres[i];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool is_obj = t == 1023, is_clu = t < k; // the chains: costs on the block's last thread, cluster c on thread c
    double prev = 0.0;
    int round = 0;
    for (;; round++) {
        double o = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0; uint32_t cnt = 0;
        for (int c0 = 0; c0 < n; c0 += kKmChunk) {
            for (int i = t; i < 16 * 256; i += 1024) (&s_wcnt[0][0])[i] = 0;
            __syncthreads(); // also: s_c is in place, the previous chunk's chains are done
            const int p = c0 + t;
            const bool valid = p < n;
            double x = 0.0, y = 0.0, z = 0.0, min_dist = 0.0;
            uint32_t a = 0xffffffffu;
            if (valid) {
                x = pts[3 * (size_t)p]; y = pts[3 * (size_t)p + 1]; z = pts[3 * (size_t)p + 2];
                min_dist = __longlong_as_double(0x7ff0000000000000ll);
                a = 0;
                for (int i = 0; i < k; i++) {
                    const double d0 = x - s_c[3 * i], d1 = y - s_c[3 * i + 1], d2 = z - s_c[3 * i + 2];
                    const double dist = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
                    if (dist < min_dist) { min_dist = dist; a = (uint32_t)i; }
                }
                assign[p] = a;
            }
            s_cost[t] = min_dist; // +0.0 behind the last point: the running sum is never -0.0, so it stays bit-identical
            // rank of the point among the wave's earlier members of its cluster: one ballot per cluster present in the wave
            int rank = 0;
            unsigned long long todo = __ballot(valid);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)a, leader);
                const unsigned long long m = __ballot(valid && a == c);
                if (valid && a == c) rank = __popcll(m & lt_mask);
                if (lane == leader) s_wcnt[w][c] = (unsigned short)__popcll(m);
                todo &= ~m;
            }
            __syncthreads();
            if (t < 256) { // exclusive prefix over the waves, per cluster; then over the clusters
                int run = 0;
                if (t < k) for (int w2 = 0; w2 < 16; w2++) { const int tmp = s_wcnt[w2][t]; s_wcnt[w2][t] = (unsigned short)run; run += tmp; }
                s_tot[t] = run;
                int inc = run;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(inc, d); if (lane >= d) inc += up; }
                if (lane == 63) s_wsum[w] = inc;
                s_start[t] = inc - run; // exclusive inside the wave; the earlier waves' totals are added below
            }
            __syncthreads();
            if (t < 256) { int add = 0; for (int w2 = 0; w2 < w; w2++) add += s_wsum[w2]; s_start[t] += add; }
            __syncthreads();
            if (valid) {
                const int pos = s_start[a] + (int)s_wcnt[w][a] + rank;
                s_p[3 * pos] = x; s_p[3 * pos + 1] = y; s_p[3 * pos + 2] = z;
            }
            __syncthreads();
            if (is_obj) {
                for (int j0 = 0; j0 < kKmChunk; j0 += 8) { // eight LDS reads in flight, then the ordered adds
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = s_cost[j0 + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) o = o + v[u];
                }
            } else if (is_clu) {
                const int b = s_start[t], e = b + s_tot[t];
#pragma unroll 4
                for (int j = b; j < e; j++) { s0 += s_p[3 * j]; s1 += s_p[3 * j + 1]; s2 += s_p[3 * j + 2]; }
                cnt += (uint32_t)(e - b);
            }
        }
        __syncthreads(); // every chain is complete, nobody reads s_c any more this round
        if (is_obj) {
            int stop = 0;
            if (round > 0 && fabs(o - prev) < 1e-6) stop = 1; // converged: keep the centres of this assignment
            if (round == 100) stop = 1;                        // iteration cap (100): no further update_centres
            s_stop = stop;
            prev = o;
        }
        __syncthreads();
        if (s_stop) break;
        if (is_clu) {
            const double sc = 1.0 / (double)cnt; // empty cluster -> inf -> NaN centre, as in the reference
            s_c[3 * t] = s0 * sc; s_c[3 * t + 1] = s1 * sc; s_c[3 * t + 2] = s2 * sc;
        }
        // (the barrier at the top of the next chunk loop publishes the new centres)
    }
    for (int i = t; i < k * 3; i += 1024) centres[i] = s_c[i];
    if (t == 0) P.rounds[prob] = round;
}

// per-tile f32 sums in the reference's order (tile pixels x outer, y inner; lib.rs:91-116)
__global__ void k_tile_sums(const uint8_t *__restrict__ orig, const float *__restrict__ labpx, int W, int H, int perceptual, float *__restrict__ sums /*[tiles][3]*/, int *__restrict__ counts) {
    int tile = blockIdx.x * blockDim.x + threadIdx.x;
    int wt = W / 8, ht = H / 8;
    if (tile >= wt * ht) return;
    int tx = tile % wt, ty = tile / wt;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f; int cnt = 0;
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            size_t px = (size_t)(ty * 8 + y) * W + tx * 8 + x;
            uint32_t o = reinterpret_cast<const uint32_t *>(orig)[px];
            if ((o >> 24) != 0) {
                if (perceptual) { s0 += labpx[3 * px]; s1 += labpx[3 * px + 1]; s2 += labpx[3 * px + 2]; }
                else { s0 += (float)(o & 0xff); s1 += (float)((o >> 8) & 0xff); s2 += (float)((o >> 16) & 0xff); }
                cnt++;
            }
        }
    sums[3 * tile] = s0; sums[3 * tile + 1] = s1; sums[3 * tile + 2] = s2; counts[tile] = cnt;
}
// gather k-means points from a pixel index list (RGB as f64, or Lab f32 widened to f64; lib.rs:344-358)
__global__ void k_gather_points(const uint8_t *__restrict__ orig, const float *__restrict__ labpx, const uint32_t *__restrict__ index, long long n, int perceptual, double *__restrict__ pts) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t px = index[i];
    if (perceptual) { pts[3 * i] = (double)labpx[3 * (size_t)px]; pts[3 * i + 1] = (double)labpx[3 * (size_t)px + 1]; pts[3 * i + 2] = (double)labpx[3 * (size_t)px + 2]; }
    else { uint32_t o = reinterpret_cast<const uint32_t *>(orig)[px]; pts[3 * i] = (double)(o & 0xff); pts[3 * i + 1] = (double)((o >> 8) & 0xff); pts[3 * i + 2] = (double)((o >> 16) & 0xff); }
}

// ---- dynamic tile -> subpalette reassignment (snesimage_reassign_tiles; not in the reference, TODO.md:36-37) ----------
// cost[tile][p] = sum over the tile's opaque pixels, raster order inside the tile, of the distance optimize() minimises
// (lib.rs:1080-1100) to the nearest entry of subpalette p, in binary64.  The redmean distance is sqrt(key / 512) with the
// exact integer key of red_mean_key: bit-identical to the reference's expression, whose every intermediate is exact.
// thread = (tile, subpalette): the sum is one ordered chain of at most 64 terms.
__global__ void k_tile_costs(const uint8_t *__restrict__ orig, const uint32_t *__restrict__ pal_rgb8, const float *__restrict__ pal_lab, const float *__restrict__ labpx, int W, int H,
                             int sub_count, int sub_size, int perceptual, double *__restrict__ cost, int *__restrict__ any) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int wt = W / 8, ntile = wt * (H / 8);
    if (i >= ntile * sub_count) return;
    const int tile = i / sub_count, p = i - tile * sub_count;
    const int tx = tile % wt, ty = tile / wt;
    double c = 0.0; int a = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            const size_t px = (size_t)(ty * 8 + y) * W + tx * 8 + x;
            const uint32_t o = reinterpret_cast<const uint32_t *>(orig)[px];
            if ((o >> 24) == 0) continue;
            a = 1;
            if (!perceptual) {
                uint32_t bk = 0xffffffffu;
                for (int j = 0; j < sub_size; j++) { const uint32_t k = red_mean_key(pal_rgb8[p * sub_size + j], o & 0x00ffffffu); if (k < bk) bk = k; }
                c = c + sqrt((double)bk / 512.0);
            } else {
                Lab t; t.l = labpx[3 * px]; t.a = labpx[3 * px + 1]; t.b = labpx[3 * px + 2];
                float bd = 0.0f;
                for (int j = 0; j < sub_size; j++) {
                    Lab e; e.l = pal_lab[3 * (p * sub_size + j)]; e.a = pal_lab[3 * (p * sub_size + j) + 1]; e.b = pal_lab[3 * (p * sub_size + j) + 2];
                    const float d = ciede2000(e, t);
                    if (j == 0 || d < bd) bd = d;
                }
                c = c + (double)bd;
            }
        }
    cost[i] = c;
    if (p == 0) any[tile] = a;
}
// the move: strictly smallest cost, scanning upwards from the current subpalette's (ties keep the current one, then the lower index)
__global__ void k_tile_move(const double *__restrict__ cost, const int *__restrict__ any, int ntile, int sub_count, uint8_t *__restrict__ tile_pal, unsigned int *__restrict__ moved) {
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= ntile || !any[tile]) return;
    const int cur = tile_pal[tile];
    int best = cur;
    for (int p = 0; p < sub_count; p++) if (cost[tile * sub_count + p] < cost[tile * sub_count + best]) best = p;
    if (best != cur) { tile_pal[tile] = (uint8_t)best; atomicAdd(moved, 1u); }
}

// ---- kernel entry points of the bodies above ----
template <bool PERC, int SUB, int MODE = 0, int NT = 128>
__global__ __launch_bounds__(NT) void k_dither(DitherParams P) { dither_body<PERC, SUB, MODE, NT>(P, (int)blockIdx.x); }
template <int SUB, int MODE>
__global__ __launch_bounds__(512) void k_dither4(DitherParams P) { dither4_body<SUB, MODE>(P, (int)blockIdx.x); }
template <int MODE>
__global__ __launch_bounds__(512) void k_dither4_lab(DitherParams P) { dither4_body<0, MODE, true>(P, (int)blockIdx.x); }
template <int SUB>
__global__ __launch_bounds__(256) void k_ditherw(DitherParams P, int nrun) { ditherw_body<SUB>(P, (int)blockIdx.x, nrun); }
__global__ void k_gen_candidates(int method, int n, unsigned long long key, const uint8_t *__restrict__ colors, int slot, int channel, uint8_t *__restrict__ cand,
                                 int rank = 0, int count = 1, uint8_t *__restrict__ sel = nullptr, double *__restrict__ errors = nullptr) {
    gen_candidates_body(method, n, key, colors, slot, channel, cand, rank, count, sel, errors);
}
__global__ __launch_bounds__(256) void k_commit(const double *__restrict__ errors, int n, const uint8_t *__restrict__ cand, uint8_t *__restrict__ colors, int slot, int nes, double *__restrict__ inc_err,
                                               StepResult *__restrict__ last, PaletteTables T) {
    commit_body(errors, n, cand, colors, slot, nes, inc_err, last, T);
}

} // namespace snes
