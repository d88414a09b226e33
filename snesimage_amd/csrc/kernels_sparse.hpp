// snesimage_amd/csrc/kernels_sparse.hpp — group-sparse ("delta") scoring of candidate palettes.
//
// Observation (measured on the BASELINE workload): replacing one palette entry by a random colour
// changes few pixels.  Let B be the image in which every pixel of the slot's subpalette takes its
// best *other* entry (the pack's fixed colour index); a candidate differs from B only at the pixels
// it wins — a median of ~40 pixels in ~20 of the 256 rows, clustered in the subpalette's 8x8 tiles.
//
// Every stage of ssimulacra2 is causal along its sweep, so everything computed before the first
// changed input is bit-identical to what the same stage computes for B:
//   * H pass: an output row depends only on its input row  -> only changed rows are recomputed;
//   * V pass + maps + pooling sums: a column's recurrence state and its running sums at step n
//     depend only on rows < n+5  -> resume from B's checkpoint at the first changed row; columns left
//     of the first changed column (minus the filter's reach) are B's outright;
//   * downscale/XYB: a pixel of scale s depends on its 2^s x 2^s block -> only changed rows.
// The unit of change is a *group* of four consecutive rows (4g..4g+3): it matches the 16-byte,
// four-rows-per-lane accesses of the V pass and the four-lane quads of the H pass, and changed rows
// come in runs anyway (tiles are 8 rows tall).  B itself is scored once per slot — it is the
// pseudo-candidate "every group changed, wins nothing" — and leaves, in its own storage, all planes
// plus one V-pass checkpoint per group.  Results are bit-identical to the dense kernels (same
// operations in the same order; tests compare the two paths); the work per candidate shrinks to the
// groups and columns it actually touches.
//
// Per-candidate storage, slot j = index of the group in the candidate's ascending changed-group list
// of that scale (for B: j = g):
//   lin [s][j][3][4][W]        linear RGB rows (input of the next downscale)
//   xybC[s][j][3][W/4][4][4]   XYB, "C4" inside the group: [x/4][row][x%4]   (H-pass input, lane = row)
//   xybR[s][j][3][W][4]        XYB, "R4" inside the group: [x][row]          (V-pass maps, lane = column)
//   hout[s][j][9][W/64][64][4] H-pass output, "XT4" inside the group: [x/64][x%64][row]
#pragma once
#include "kernels.hpp"

namespace snes {

constexpr int kGroupsTotal = 126; // 64 + 32 + 16 + 8 + 4 + 2

struct CandMeta {
    int ngroups[kMaxScales];
    int xmin;                          // smallest x of a won pixel (W if none)
    int won;                           // number of won pixels
    unsigned char glist[kGroupsTotal + 2]; // per scale: ascending changed groups
    short gslot[kGroupsTotal];             // per scale: group -> slot, -1 if unchanged
    unsigned char gcb[kGroupsTotal];       // per scale: group -> first 64-column block holding a won pixel (0 at scales narrower than 64)
};
// a work item: candidate (storage index), 4-row group of the scale, the group's slot in the candidate's storage, channel
__device__ __forceinline__ int item_k(unsigned int it) { return (int)(it >> 14); }
__device__ __forceinline__ int item_g(unsigned int it) { return (int)((it >> 8) & 63u); }
__device__ __forceinline__ int item_j(unsigned int it) { return (int)((it >> 2) & 63u); }
__device__ __forceinline__ int item_ch(unsigned int it) { return (int)(it & 3u); }
constexpr int kColBuckets = 4; // 64-column blocks of the widest scale: work items are listed per (scale, first block)
constexpr int kItemLists = kMaxScales * kColBuckets;

struct SparseGeom {
    long long off_lin[kMaxScales], off_xybC[kMaxScales], off_xybR[kMaxScales], off_hout[kMaxScales]; // floats inside one candidate's storage
    long long cand_stride;
    long long off_ckf[kMaxScales], off_cka[kMaxScales]; // checkpoint arrays of B (V pass)
    long long off_ckh[kMaxScales];                      // B's H-pass state on entering column blocks 1.. of the wide scales: [3 ch][3][18][H]
    int goff[kMaxScales];                               // offset of scale s inside CandMeta::glist / gslot
};

struct SparseParams {
    Geom G; SparseGeom S; BlurK K;
    int ncand, k0, base, ncol, is_base; // candidates of this launch occupy storage indices [k0, k0+ncand); base = index of B
    const unsigned long long *pack, *packC4, *packR4;
    const uint4 *plist; const int *plist_count; // contested pixels of the slot: {px, rgb, thr, 0}
    const float *pal_lin, *pal_xyb, *cand_tab;
    // --perceptual-palettes: the win test is CIEDE2000 (f32), too dear to repeat per stage: k_sparse_scan_lab evaluates it once
    // per contested pixel and records the won pixels in a per-candidate bitmap (W*H bits) that the other stages consult
    int perceptual; const float *labpx, *cand_lab; uint32_t *bitmap;
    const float *img1C4, *mu1R4, *sd1R4, *a1R4; const double *r1R4; // source arrays in the blocked layouts, + G.src_off[s] (maps_accumulate)
    float *store; CandMeta *meta;
    unsigned int *items; int *item_count; long long item_stride; // per (scale, first column block b): items[(s*4+b)*item_stride + i] = cand << 14 | group << 8 | slot << 2 | ch (item_k / item_g / item_j / item_ch)
    float *ckf; double *cka; double *part; float *ckh;
    const float *zeros; // 3 * 4 * W floats of 0.0f: what sparse_v2_body prefetches for the group below the image
    float *trash;       // 256 floats nobody reads: where the H pass's flush sends the stores of lanes that have nothing to store (its store instructions carry no predicate)
    // --dither: a candidate's pixels come from its own palette_map (k_dither, MODE 2) instead of the pack's win test; B's from bmap
    int use_maps, sub_size; uint32_t slot_ci; const uint8_t *maps, *mapsC4, *bmap, *bmapC4, *subC4, *tile_pal; // maps: + (k - k0) * W * H
    int s_first; // the general H and V bodies skip scales below this one (the wide scales run the bodies of kernels_sparse2.hpp)
    int *first;       // per candidate: first changed group of scale 0 (H/4 if none), written by the scan for k_sparse_order
    const int *order; // k_sparse_v: candidates of the launch, longest column sweeps first (k_sparse_order); nullptr = as listed
};

__device__ __forceinline__ uint32_t sparse_ci(uint32_t lo, uint32_t thr, uint32_t crgb, uint32_t ncol) {
    return red_mean_key(crgb, lo & 0x00ffffffu) < thr ? ncol : (lo >> 24); // B passes crgb with thr ignored: see callers
}

__device__ __forceinline__ uint32_t won_bit(const uint32_t *bm, int px) { return (bm[px >> 5] >> (px & 31)) & 1u; }
// --dither: colour index of pixel (x0, y0) from a palette_map (ncol = the candidate's colour, ncol + 1 = transparent)
__device__ __forceinline__ uint32_t maps_ci(const SparseParams &P, const uint8_t *map, bool is_base, int x0, int y0, uint32_t pack_lo) {
    uint32_t ci = pack_lo >> 24; // ncol + 1 when transparent (pack mode 1)
    if (ci != (uint32_t)P.ncol + 1u) {
        ci = (uint32_t)P.tile_pal[(x0 >> 3) + (y0 >> 3) * (P.G.W >> 3)] * (uint32_t)P.sub_size + map[y0 * P.G.W + x0];
        if (!is_base && ci == P.slot_ci) ci = (uint32_t)P.ncol;
    }
    return ci;
}

__device__ __forceinline__ unsigned long long pair_or_compress(unsigned long long m) { // bit i of result = bit 2i | bit 2i+1 of m
    m = (m | (m >> 1)) & 0x5555555555555555ull;
    m = (m | (m >> 1)) & 0x3333333333333333ull;
    m = (m | (m >> 2)) & 0x0f0f0f0f0f0f0f0full;
    m = (m | (m >> 4)) & 0x00ff00ff00ff00ffull;
    m = (m | (m >> 8)) & 0x0000ffff0000ffffull;
    m = (m | (m >> 16)) & 0x00000000ffffffffull;
    return m;
}

// ---- what a scan leaves behind for one candidate (shared by the RGB and the CIEDE2000 scans) ------------------------
// Inputs, per wave = candidate: `mask` = changed 4-row groups of scale 0, `xmin`/`won` as in CandMeta, and in lane g the
// smallest x of a won pixel inside group g of scale 0 (`xg`, >= W if none).  A scale has at most 64 groups, so its
// changed set is one 64-bit word: scale s+1 is the pairwise OR of scale s, a group's slot is a popcount, its first
// changed column the pairwise minimum.  Work items (candidate, slot, channel) are listed per scale and per first
// 64-column block (the H pass of a changed row starts at that block, see sparse_h2_body); list space is claimed with one
// returning atomic per block, scale and column block (a single counter word serialises at ~90 atomics/us, so one per
// candidate would cost more than the scan itself); the block's NW waves take consecutive ranges.
template <int NW>
__device__ __forceinline__ void scan_publish(const SparseParams &P, const int k, const bool live, const unsigned long long mask, const int xmin, const int won, const int xg) {
    constexpr int NL = kItemLists;
    __shared__ int s_tot[NW][NL], s_off[NW][NL];
    const Geom &G = P.G;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below_mask = (1ull << lane) - 1ull;
    unsigned long long ms[kMaxScales]; int cbs[kMaxScales];
    {
        unsigned long long m = live ? mask : 0ull;
        int xgs = xg;
#pragma unroll
        for (int s = 0; s < kMaxScales; s++) {
            ms[s] = 0ull; cbs[s] = 0;
            if (s < G.nscales) {
                const bool flag = ((m >> lane) & 1ull) != 0ull;
                const int nb = G.sw[s] >= 64 ? (G.sw[s] >> 6) : 1;
                // a changed input at column x moves the H outputs from column x - 4 on (the filter's right taps reach n + 4)
                const int cb = (flag && nb > 1) ? min(max((xgs >> s) - 4, 0) >> 6, nb - 1) : 0;
                ms[s] = m; cbs[s] = cb;
#pragma unroll
                for (int b = 0; b < kColBuckets; b++) {
                    const unsigned long long mb = __ballot(flag && cb == b);
                    if (lane == 0) s_tot[w][s * kColBuckets + b] = 3 * __popcll(mb);
                }
                m = pair_or_compress(m);
                const int xa = __shfl(xgs, (2 * lane) & 63), xb = __shfl(xgs, (2 * lane + 1) & 63);
                xgs = min(xa, xb); // lanes >= half the group count hold junk: their groups do not exist at the next scale
            } else if (lane == 0) {
#pragma unroll
                for (int b = 0; b < kColBuckets; b++) s_tot[w][s * kColBuckets + b] = 0;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < NL) {
        const int l = threadIdx.x;
        int sum = 0;
        for (int i = 0; i < NW; i++) { s_off[i][l] = sum; sum += s_tot[i][l]; }
        const int base = sum ? atomicAdd(&P.item_count[l], sum) : 0;
        for (int i = 0; i < NW; i++) s_off[i][l] += base;
    }
    __syncthreads();
    if (!live) return;
    CandMeta *M = P.meta + k;
    if (lane == 0) { M->xmin = xmin; M->won = won; }
#pragma unroll
    for (int s = 0; s < kMaxScales; s++) {
        if (s >= G.nscales) break;
        const int NG = G.sh[s] >> 2;
        const unsigned long long m = ms[s];
        const int cb = cbs[s];
        const bool flag = ((m >> lane) & 1ull) != 0ull;
        const int below = __popcll(m & below_mask);
        if (lane < NG) {
            M->gslot[P.S.goff[s] + lane] = flag ? (short)below : (short)-1;
            M->gcb[P.S.goff[s] + lane] = (unsigned char)cb;
            if (flag) M->glist[P.S.goff[s] + below] = (unsigned char)lane;
        }
        if (lane == 0) M->ngroups[s] = __popcll(m);
        if (lane == 0 && s == 0 && P.first) P.first[k] = m ? __ffsll((long long)m) - 1 : NG;
#pragma unroll
        for (int b = 0; b < kColBuckets; b++) {
            const bool mine = flag && cb == b;
            const unsigned long long mb = __ballot(mine);
            if (mine) {
                unsigned int *dst = P.items + (size_t)(s * kColBuckets + b) * P.item_stride + s_off[w][s * kColBuckets + b] + 3 * __popcll(mb & below_mask);
                const unsigned int v = ((unsigned int)k << 14) | ((unsigned int)lane << 8) | ((unsigned int)below << 2); // (lane = the group: its number rides along, so that the consumers need no second look-up)
                dst[0] = v; dst[1] = v + 1u; dst[2] = v + 2u;
            }
        }
    }
}

// ---- perceptual scan: CIEDE2000 win test per contested pixel, one block (four waves) per candidate ---------------------
// The evaluation is a long dependent chain (transcendentals): the kernel lives on resident waves.  One wave per candidate
// leaves two waves per SIMD at 2,048 candidates per launch; the four waves of a block share a candidate instead, each
// taking every fourth round of 64 contested pixels (won pixels go to the bitmap with atomics, the groups' leftmost won
// columns to one LDS array; masks and counts are combined at the end).
__device__ __forceinline__ void sparse_scan_lab_body(const SparseParams &P) {
    __shared__ uint32_t s_queue[4][128];
    __shared__ int s_gx[64]; // smallest won x of every scale-0 group
    __shared__ unsigned long long s_mask[4];
    __shared__ int s_xmin[4], s_won[4];
    const Geom &G = P.G;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    if (w == 0) s_gx[lane] = 0x7fff;
    __syncthreads();
    const int wi = (int)blockIdx.x;
    const bool live = wi < P.ncand;
    const int k = P.k0 + (live ? wi : 0);
    unsigned long long mask = 0ull; int xmin = G.W, won = 0;
    if (live) {
        Lab cl; cl.l = P.cand_lab[3 * (size_t)k]; cl.a = P.cand_lab[3 * (size_t)k + 1]; cl.b = P.cand_lab[3 * (size_t)k + 2];
        uint32_t *bm = P.bitmap + (size_t)k * (G.W * G.H / 32);
        const int n = *P.plist_count;
        auto take = [&](uint32_t px) { // the candidate wins pixel px
            const int x = (int)(px & (unsigned)(G.W - 1)), y = (int)(px / (unsigned)G.W);
            atomicOr(&bm[px >> 5], 1u << (px & 31));
            atomicMin(&s_gx[y >> 2], x);
            mask |= 1ull << (y >> 2);
            xmin = min(xmin, x);
            won++;
        };
        auto full_test = [&](int i) { // the CIEDE2000 evaluation proper
            const uint4 e = P.plist[i];
            Lab t; t.l = P.labpx[3 * (size_t)e.x]; t.a = P.labpx[3 * (size_t)e.x + 1]; t.b = P.labpx[3 * (size_t)e.x + 2];
            const float d = ciede2000(cl, t), bd = __uint_as_float(e.z & 0x7fffffffu);
            if ((d < bd) || ((e.z & 0x80000000u) && d == bd)) take(e.x); // strict <, ties to the lower index (lib.rs:788-791)
        };
        // Most contested pixels are out of reach on lightness alone (ciede2000_cannot_beat).  A lane that cannot rule its
        // pixel out queues it; the ~2,000-instruction evaluation then runs on full waves of queued pixels instead of on
        // the few surviving lanes of every iteration.
        uint32_t *q = s_queue[w];
        int queued = 0; // wave-uniform
        uint4 e_n = make_uint4(0, 0, 0, 0); Lab t_n{0.0f, 0.0f, 0.0f}; // entry and colour of the wave's NEXT round, fetched a round ahead
        if (64 * w + lane < n) { e_n = P.plist[64 * w + lane]; t_n.l = P.labpx[3 * (size_t)e_n.x]; t_n.a = P.labpx[3 * (size_t)e_n.x + 1]; t_n.b = P.labpx[3 * (size_t)e_n.x + 2]; }
        const float cch = sqrtf(cl.a * cl.a + cl.b * cl.b);
        for (int i0 = 64 * w; i0 < n; i0 += 256) {
            const int i = i0 + lane;
            const uint4 e = e_n; const Lab t = t_n;
            if (i + 256 < n) { e_n = P.plist[i + 256]; t_n.l = P.labpx[3 * (size_t)e_n.x]; t_n.a = P.labpx[3 * (size_t)e_n.x + 1]; t_n.b = P.labpx[3 * (size_t)e_n.x + 2]; }
            bool maybe = false;
            if (i < n) {
                if (e.z == 0xffffffffu) take(e.x);
                else { const float bd = __uint_as_float(e.z & 0x7fffffffu); maybe = !ciede2000_cannot_beat(cl, t, bd) && !ciede2000_cannot_beat_ab(cl, cch, t, bd); } // the two sure "no"s of color.hpp
            }
            const unsigned long long mm = __ballot(maybe);
            if (maybe) q[queued + __popcll(mm & ((1ull << lane) - 1ull))] = (uint32_t)i;
            queued += __popcll(mm);
            if (queued >= 64) { // (the queue holds at most 127 entries)
                full_test((int)q[lane]);
                queued -= 64;
                if (lane < queued) { const uint32_t v = q[64 + lane]; q[lane] = v; } // same-wave LDS traffic is ordered
            }
        }
        if (lane < queued) full_test((int)q[lane]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mask |= __shfl_xor(mask, o);
            xmin = min(xmin, __shfl_xor(xmin, o));
            won += __shfl_xor(won, o);
        }
    }
    if (lane == 0) { s_mask[w] = mask; s_xmin[w] = xmin; s_won[w] = won; }
    __syncthreads();
    mask = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
    xmin = min(min(s_xmin[0], s_xmin[1]), min(s_xmin[2], s_xmin[3]));
    won = s_won[0] + s_won[1] + s_won[2] + s_won[3];
    scan_publish<4>(P, k, live && w == 0, mask, xmin, won, s_gx[lane]); // wave 0 publishes for the candidate
}

// ---- which groups does each candidate change? ---------------------------------------------------------
// One wavefront per candidate (four candidates per block).  Each lane tests its share of the slot's
// contested pixels and keeps a 64-bit mask of the 4-row groups it saw change; the wave ORs the masks
// (and min-reduces x) with shuffles.  A scale has at most 64 groups, so its changed set is one 64-bit
// word: scale s+1 is the pairwise OR of scale s, a group's slot is a popcount.  No LDS, no barriers.
constexpr int kScanTile = 6144; // contested pixels staged in LDS per pass (the BASELINE slot has ~6.8 k)
// WPC: waves per candidate.  A wave walks the list in steps of 64: ~107 dependent iterations for the BASELINE slot, whatever
// the launch holds — 35-50 us, a tenth of a short call or a short slot window.  With WPC = 4 a candidate's list is dealt to
// four waves (a block = four candidates; masks, counts and the groups' leftmost columns combined through LDS, as
// dither_diff_body does): a quarter of the chain for four times the blocks, which is what lists that do not fill the chip want.
template <int WPC>
__device__ __forceinline__ void sparse_scan_body(const SparseParams &P) {
    constexpr int CPB = 16 / WPC; // candidates per block: sixteen waves share one LDS copy of the slot's contested-pixel list
    __shared__ uint32_t s_rgb[kScanTile], s_thr[kScanTile];
    __shared__ unsigned short s_px[kScanTile]; // x | (y>>2) << 8 would lose x precision: keep x (8 bit) and group (6 bit)
    __shared__ int s_gx[CPB][64]; // per candidate: smallest won x of every scale-0 group
    __shared__ unsigned long long s_mask[16];
    __shared__ int s_xmin[16], s_won[16];
    const Geom &G = P.G;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cw = w / WPC, part = w % WPC;
    if (part == 0) s_gx[cw][lane] = 0x7fff;
    const int wi = (int)blockIdx.x * CPB + cw;
    const bool live = P.is_base ? (wi == 0) : (wi < P.ncand);
    const int k = P.is_base ? P.base : P.k0 + (live ? wi : 0);
    unsigned long long mask = 0ull; int xmin = G.W, won = 0;
    if (P.is_base) { const int ng0 = G.sh[0] >> 2; mask = ng0 >= 64 ? ~0ull : ((1ull << ng0) - 1ull); xmin = 0; } // every group of the image (64 at 256 rows)
    else {
        const uint32_t crgb = __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
        const int n = *P.plist_count;
        for (int t0 = 0; t0 < n; t0 += kScanTile) {
            const int nt = min(kScanTile, n - t0);
            __syncthreads();
            for (int i = threadIdx.x; i < nt; i += 1024) {
                const uint4 e = P.plist[t0 + i];
                s_rgb[i] = e.y; s_thr[i] = e.z;
                s_px[i] = (unsigned short)((e.x & (unsigned)(G.W - 1)) | ((e.x / (unsigned)G.W) >> 2) << 8);
            }
            __syncthreads();
            if (live) {
                for (int i = part * 64 + lane; i < nt; i += 64 * WPC) {
                    if (red_mean_key(crgb, s_rgb[i]) < s_thr[i]) {
                        const int px = s_px[i];
                        atomicMin(&s_gx[cw][px >> 8], px & 255);
                        mask |= 1ull << (px >> 8);
                        xmin = min(xmin, px & 255);
                        won++;
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mask |= __shfl_xor(mask, o);
            xmin = min(xmin, __shfl_xor(xmin, o));
            won += __shfl_xor(won, o);
        }
    }
    if (WPC > 1) { // the candidate's waves combine what they saw
        if (lane == 0) { s_mask[w] = mask; s_xmin[w] = xmin; s_won[w] = won; }
        __syncthreads();
        if (!P.is_base) {
            mask = 0ull; xmin = G.W; won = 0;
#pragma unroll
            for (int q = 0; q < WPC; q++) { mask |= s_mask[cw * WPC + q]; xmin = min(xmin, s_xmin[cw * WPC + q]); won += s_won[cw * WPC + q]; }
        }
    }
    scan_publish<16>(P, k, live && part == 0, mask, xmin, won, P.is_base ? 0 : s_gx[cw][lane]); // the candidate's first wave publishes
}

// ---- --dither: where does a candidate first differ from B? -----------------------------------------------------
// Floyd-Steinberg is causal in raster order: the candidate's run is B's run (k_dither MODE 1: the slot's entry out of play)
// up to the first pixel whose dithered target is nearer to the candidate's colour than to B's choice.  The contested-pixel
// list holds B's targets and the keys to beat (built from k_dither's record by k_build_plist); one wave per candidate
// min-reduces the index of the pixels it wins.  From the 4-row group of that pixel on every row differs (the error
// may spread right and down): k_dither (MODE 2) re-runs those rows, k_dither_diff then finds what actually changed.
__device__ __forceinline__ void dither_first_body(const SparseParams &P) {
    __shared__ uint32_t s_rgb[kScanTile], s_thr[kScanTile];
    __shared__ unsigned short s_px[kScanTile]; // W * H <= 65,536
    const Geom &G = P.G;
    const int lane = threadIdx.x & 63;
    const int wi = (int)blockIdx.x * 16 + (threadIdx.x >> 6);
    const bool live = wi < P.ncand;
    const int k = P.k0 + (live ? wi : 0);
    const uint32_t crgb = __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    const int n = *P.plist_count;
    int first = 0x7fffffff;
    for (int t0 = 0; t0 < n; t0 += kScanTile) {
        const int nt = min(kScanTile, n - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += 1024) {
            const uint4 e = P.plist[t0 + i];
            s_rgb[i] = e.y; s_thr[i] = e.z; s_px[i] = (unsigned short)e.x;
        }
        __syncthreads();
        if (live)
            for (int i = lane; i < nt; i += 64)
                if (red_mean_key(crgb, s_rgb[i]) < s_thr[i]) first = min(first, (int)s_px[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (live && lane == 0) P.first[k] = first == 0x7fffffff ? (G.H >> 2) : (first / G.W) >> 2; // k_dither (MODE 2) resumes there
}

// The same with --perceptual-palettes: B's record holds the bits of the CIEDE2000 distance to beat (plus one where a tie goes
// to the slot: see DitherParams::rec_pack) and the dithered targets' Lab values lie in P.labpx (B's run left them behind);
// the candidate takes a pixel iff the bits of ciede2000(candidate, target) — the call the resumed run will make, argument
// for argument — are below.  One block (four waves) per candidate, the two sure "no"s of color.hpp and full waves of
// queued pixels as in k_sparse_scan_lab.
__device__ __forceinline__ void dither_first_lab_body(const SparseParams &P) {
    __shared__ uint32_t s_queue[4][128];
    __shared__ int s_first;
    const Geom &G = P.G;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = (int)blockIdx.x;
    if (wi >= P.ncand) return;
    const int k = P.k0 + wi;
    if (threadIdx.x == 0) s_first = 0x7fffffff;
    __syncthreads();
    Lab cl; cl.l = P.cand_lab[3 * (size_t)k]; cl.a = P.cand_lab[3 * (size_t)k + 1]; cl.b = P.cand_lab[3 * (size_t)k + 2];
    const float cch = sqrtf(cl.a * cl.a + cl.b * cl.b);
    const int n = *P.plist_count;
    int first = 0x7fffffff;
    auto full_test = [&](int i) {
        const uint4 e = P.plist[i];
        Lab t; t.l = P.labpx[3 * (size_t)e.x]; t.a = P.labpx[3 * (size_t)e.x + 1]; t.b = P.labpx[3 * (size_t)e.x + 2];
        if (__float_as_uint(ciede2000(cl, t)) < e.z) first = min(first, (int)e.x);
    };
    uint32_t *q = s_queue[w];
    int queued = 0; // wave-uniform
    for (int i0 = 64 * w; i0 < n; i0 += 256) {
        const int i = i0 + lane;
        bool maybe = false;
        if (i < n) {
            const uint4 e = P.plist[i];
            if (e.z == 0xffffffffu) first = min(first, (int)e.x);
            else {
                Lab t; t.l = P.labpx[3 * (size_t)e.x]; t.a = P.labpx[3 * (size_t)e.x + 1]; t.b = P.labpx[3 * (size_t)e.x + 2];
                const float bd = __uint_as_float(e.z); // the distance to beat, or the next value up
                maybe = !ciede2000_cannot_beat(cl, t, bd) && !ciede2000_cannot_beat_ab(cl, cch, t, bd);
            }
        }
        const unsigned long long mm = __ballot(maybe);
        if (maybe) q[queued + __popcll(mm & ((1ull << lane) - 1ull))] = (uint32_t)i;
        queued += __popcll(mm);
        if (queued >= 64) {
            full_test((int)q[lane]);
            queued -= 64;
            if (lane < queued) { const uint32_t v = q[64 + lane]; q[lane] = v; }
        }
    }
    if (lane < queued) full_test((int)q[lane]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (lane == 0) atomicMin(&s_first, first);
    __syncthreads();
    if (threadIdx.x == 0) P.first[k] = s_first == 0x7fffffff ? (G.H >> 2) : (s_first / G.W) >> 2;
}

// ---- --dither: what did the resumed run change? -------------------------------------------------------------------
// The error a changed pixel injects is diffused with a total weight of 0.8 per row, so it fades: away from the pixels the
// candidate takes, its resumed run soon chooses what B chose.  The score depends on the picture only, so the changed set
// of the group-sparse scorer is simply where the two palette_maps differ (a pixel on the slot's index always does: B never
// uses it).  Four waves per candidate (a block = four candidates), each taking every fourth chunk of eight rows (lane = four
// pixels; the loop is a chain of loads, and one wave per candidate left two waves per SIMD); publishes like the scans.
__device__ __forceinline__ void dither_diff_body(const SparseParams &P) {
    __shared__ int s_gx[4][64];
    __shared__ unsigned long long s_mask[16];
    __shared__ int s_xmin[16], s_won[16];
    const Geom &G = P.G;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = w >> 2, q = w & 3;
    const int wi = (int)blockIdx.x * 4 + c;
    const bool live = wi < P.ncand;
    const int k = P.k0 + (live ? wi : 0);
    if (q == 0) s_gx[c][lane] = 0x7fff;
    __syncthreads();
    unsigned long long mask = 0ull; int xmin = G.W, won = 0;
    if (live) {
        const uint32_t *mc = reinterpret_cast<const uint32_t *>(P.maps + (size_t)(k - P.k0) * G.W * G.H), *mb = reinterpret_cast<const uint32_t *>(P.bmap);
        const int y0 = min(4 * P.first[k], G.H);
        for (int yb = y0 + 8 * q; yb < G.H; yb += 32) { // W = 256: 64 words per row; a chunk = two whole groups, one wave's alone
            uint32_t dd[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int yy = min(yb + u, G.H - 1); dd[u] = mc[yy * 64 + lane] ^ mb[yy * 64 + lane]; } // (rows past the image are clamped here and dropped below)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int y = yb + u;
                const uint32_t d = y < G.H ? dd[u] : 0u;
                const unsigned long long m = __ballot(d != 0u);
                if (m) {
                    const int l0 = __ffsll((long long)m) - 1;
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, l0);
                    const int x = 4 * l0 + ((__ffs((int)d0) - 1) >> 3);
                    if (lane == 0) s_gx[c][y >> 2] = min(s_gx[c][y >> 2], x);
                    mask |= 1ull << (y >> 2);
                    xmin = min(xmin, x);
                    won += __popcll(m); // (words, not pixels: only a statistic)
                }
            }
        }
    }
    if (lane == 0) { s_mask[w] = mask; s_xmin[w] = xmin; s_won[w] = won; }
    __syncthreads();
    mask = s_mask[4 * c] | s_mask[4 * c + 1] | s_mask[4 * c + 2] | s_mask[4 * c + 3];
    xmin = min(min(s_xmin[4 * c], s_xmin[4 * c + 1]), min(s_xmin[4 * c + 2], s_xmin[4 * c + 3]));
    won = s_won[4 * c] + s_won[4 * c + 1] + s_won[4 * c + 2] + s_won[4 * c + 3];
    scan_publish<16>(P, k, live && q == 0, mask, xmin, won, s_gx[c][lane]); // the candidate's first wave publishes
}

// ---- downscale chain + XYB on changed groups only -------------------------------------------------------
// base: grid.x blocks share the rows of each scale (launched once per scale, P.ncand = scale to do);
// candidates: one block per candidate walks the scales itself.
#ifndef SNES_DOWN_TILES_U
#define SNES_DOWN_TILES_U 2
#endif
__device__ __forceinline__ void sparse_down_body(const SparseParams &P, int only_scale, const int bx) { // bx: blockIdx.x unless the caller remaps blocks
    __shared__ float s_lin[256 * 3];
    const Geom &G = P.G;
    const int t = threadIdx.x;
    const bool is_base = P.is_base != 0;
    if (!is_base && bx >= P.ncand) return; // (a batched launch is sized for its longest member)
    const int k = is_base ? P.base : P.k0 + bx;
    for (int i = t; i < (P.ncol + 2) * 3; i += 256) s_lin[i] = P.pal_lin[i];
    __syncthreads();
    if (t < 3 && !is_base) s_lin[3 * P.ncol + t] = P.cand_tab[8 * (size_t)k + t];
    const uint32_t crgb = is_base ? 0u : __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    __syncthreads();
    const CandMeta *M = P.meta + k;
    float *mine = P.store + (size_t)k * P.S.cand_stride;
    const float *basep = P.store + (size_t)P.base * P.S.cand_stride;
    const int s_lo = only_scale > 0 ? only_scale : (only_scale < 0 ? -only_scale : 1), s_hi = only_scale > 0 ? only_scale + 1 : G.nscales; // only_scale < 0: from scale -only_scale on
    const int part0 = is_base ? bx : 0, nparts = is_base ? (int)gridDim.x : 1;
    for (int s = s_lo; s < s_hi; s++) {
        const int Ws = G.sw[s], Wp = G.sw[s - 1];
        const int n = M->ngroups[s];
        const int lw4 = 31 - __clz(4 * Ws); // the scales' widths are powers of two (W = 256)
        for (int i = part0 * 256 + t; i < n * 4 * Ws; i += 256 * nparts) {
            // consecutive lanes walk a group the way its planes are laid out (x & 3, then the row, then the column quad): every
            // store instruction below writes whole lines
            const int j = i >> lw4, rem = i & (4 * Ws - 1), r = (rem >> 2) & 3, x = ((rem >> 4) << 2) | (rem & 3);
            const int g = M->glist[P.S.goff[s] + j];
            const int y = 4 * g + r;
            float v[3];
            if (s == 1) {
                float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int iy = 0; iy < 2; iy++)
#pragma unroll
                    for (int ix = 0; ix < 2; ix++) {
                        const unsigned long long w = P.pack[(size_t)(2 * y + iy) * G.W + 2 * x + ix];
                        const int px0 = (2 * y + iy) * G.W + 2 * x + ix;
                        const uint32_t ci = P.use_maps ? maps_ci(P, is_base ? P.bmap : P.maps + (size_t)(k - P.k0) * G.W * G.H, is_base, 2 * x + ix, 2 * y + iy, (uint32_t)w)
                                          : is_base ? ((uint32_t)w >> 24)
                                          : (P.perceptual ? (won_bit(P.bitmap + (size_t)k * (G.W * G.H / 32), px0) ? (uint32_t)P.ncol : ((uint32_t)w >> 24))
                                                          : sparse_ci((uint32_t)w, (uint32_t)(w >> 32), crgb, (uint32_t)P.ncol));
                        sum[0] += s_lin[3 * ci]; sum[1] += s_lin[3 * ci + 1]; sum[2] += s_lin[3 * ci + 2];
                    }
                v[0] = sum[0] * 0.25f; v[1] = sum[1] * 0.25f; v[2] = sum[2] * 0.25f;
            } else {
                // rows 2y, 2y+1 of scale s-1 live in one group: the candidate's own if it changed, else B's
                const int gp = (2 * y) >> 2, rp = (2 * y) & 3;
                const short sl = M->gslot[P.S.goff[s - 1] + gp];
                const float *grp = sl >= 0 ? mine + P.S.off_lin[s - 1] + (size_t)sl * 12 * Wp : basep + P.S.off_lin[s - 1] + (size_t)gp * 12 * Wp;
#pragma unroll
                for (int c = 0; c < 3; c++) { // linear planes in the C4 order: [column quad][row][column & 3]
                    const float *q0 = grp + (size_t)c * 4 * Wp + (size_t)((2 * x) >> 2) * 16 + rp * 4 + ((2 * x) & 3);
                    const float2 a = *reinterpret_cast<const float2 *>(q0), b = *reinterpret_cast<const float2 *>(q0 + 4);
                    float sum = 0.0f;
                    sum += a.x; sum += a.y; sum += b.x; sum += b.y;
                    v[c] = sum * 0.25f;
                }
            }
            float X, Y, B;
            linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);
            float *ol = mine + P.S.off_lin[s] + (size_t)j * 12 * Ws;
            float *oc = mine + P.S.off_xybC[s] + (size_t)j * 12 * Ws, *orr = mine + P.S.off_xybR[s] + (size_t)j * 12 * Ws;
            const float xyb[3] = {X, Y, B};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                ol[(size_t)c * 4 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3)] = v[c];
                oc[(size_t)c * 4 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3)] = xyb[c];
                if (is_base || Ws < 64) orr[(size_t)c * 4 * Ws + (size_t)x * 4 + r] = xyb[c]; // (wide scales: the candidates' H pass writes the R4 copy)
            }
        }
        __syncthreads(); // the rows written above are read by this block at the next scale
    }
}

// ---- scale 1 of the candidates' downscale, one block per changed group ---------------------------------------------------
// Scale 1 holds three quarters of the pixels sparse_down_body computes and reads nothing that body writes (its inputs are
// the pack and the win test), so it need not wait its turn in a block that walks a candidate's scales one after the
// other: the scan's item lists of scale 1 name every (candidate, changed group) — three items per group, one per channel —
// and a block of 256 threads takes one group (4 rows x W/2 pixels).  Same arithmetic, same stores as the s == 1 branch there.
__device__ __forceinline__ void sparse_down1_body(const SparseParams &P, const int bx, const int gx) {
    __shared__ float s_lin[256 * 3];
    const Geom &G = P.G;
    const int t = threadIdx.x;
    const int Ws = G.sw[1];
    for (int i = t; i < (P.ncol + 2) * 3; i += 256) s_lin[i] = P.pal_lin[i];
    const int nb = Ws >= 64 ? (Ws >> 6) : 1; // item lists of scale 1: one per first column block
    int total = 0, cnt[kColBuckets];
#pragma unroll
    for (int b = 0; b < kColBuckets; b++) { cnt[b] = b < nb ? P.item_count[kColBuckets + b] / 3 : 0; total += cnt[b]; }
    __syncthreads();
    // A group is two pixels per thread behind a chain of look-ups (item -> candidate's colour): the block fetches the NEXT
    // group's item and colour while it works on this one (the item carries the group's number: no look-up in the candidate's lists)
    auto fetch_item = [&](int gi) -> unsigned int {
        int b = 0, li = gi;
        while (li >= cnt[b]) { li -= cnt[b]; b++; }
        return P.items[(size_t)(kColBuckets + b) * P.item_stride + 3 * (size_t)li]; // channel 0's item of the group
    };
    unsigned int it_n = bx < total ? fetch_item(bx) : 0u;
    uint32_t crgb_n = 0u; float cl_n[3] = {0.0f, 0.0f, 0.0f};
    if (bx < total) { const float *ct = P.cand_tab + 8 * (size_t)item_k(it_n); crgb_n = __float_as_uint(ct[6]); cl_n[0] = ct[0]; cl_n[1] = ct[1]; cl_n[2] = ct[2]; }
    for (int gi = bx; gi < total; gi += gx) {
        const unsigned int it = it_n;
        const uint32_t crgb = crgb_n; const float cl0 = cl_n[0], cl1 = cl_n[1], cl2 = cl_n[2];
        if (gi + gx < total) {
            it_n = fetch_item(gi + gx);
            const float *ct = P.cand_tab + 8 * (size_t)item_k(it_n); crgb_n = __float_as_uint(ct[6]); cl_n[0] = ct[0]; cl_n[1] = ct[1]; cl_n[2] = ct[2];
        }
        const int k = item_k(it), j = item_j(it), g = item_g(it);
        float *mine = P.store + (size_t)k * P.S.cand_stride;
        float *ol = mine + P.S.off_lin[1] + (size_t)j * 12 * Ws;
        float *oc = mine + P.S.off_xybC[1] + (size_t)j * 12 * Ws, *orr = mine + P.S.off_xybR[1] + (size_t)j * 12 * Ws;
        for (int rem = t; rem < 4 * Ws; rem += 256) {
            const int r = (rem >> 2) & 3, x = ((rem >> 4) << 2) | (rem & 3);
            const int y = 4 * g + r;
            float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) {
                    const int px0 = (2 * y + iy) * G.W + 2 * x + ix;
                    const unsigned long long w = P.pack[px0];
                    const uint32_t ci = P.use_maps ? maps_ci(P, P.maps + (size_t)(k - P.k0) * G.W * G.H, false, 2 * x + ix, 2 * y + iy, (uint32_t)w)
                                      : (P.perceptual ? (won_bit(P.bitmap + (size_t)k * (G.W * G.H / 32), px0) ? (uint32_t)P.ncol : ((uint32_t)w >> 24))
                                                      : sparse_ci((uint32_t)w, (uint32_t)(w >> 32), crgb, (uint32_t)P.ncol));
                    const bool cw = ci == (uint32_t)P.ncol; // the candidate's own colour stands at table row ncol
                    sum[0] += cw ? cl0 : s_lin[3 * ci]; sum[1] += cw ? cl1 : s_lin[3 * ci + 1]; sum[2] += cw ? cl2 : s_lin[3 * ci + 2];
                }
            const float v[3] = {sum[0] * 0.25f, sum[1] * 0.25f, sum[2] * 0.25f};
            float X, Y, B;
            linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);
            const float xyb[3] = {X, Y, B};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                ol[(size_t)c * 4 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3)] = v[c];
                oc[(size_t)c * 4 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3)] = xyb[c];
                if (Ws < 64) orr[(size_t)c * 4 * Ws + (size_t)x * 4 + r] = xyb[c];
            }
        }
    }
}

// ---- scales 2.. of the candidates' downscale, one block per changed group of scale 3 --------------------------------------
// sparse_down_body walks a candidate's scales one after the other: every scale reads the linear planes the one before stored,
// behind a block barrier — four dependent trips to memory for a few thousand pixels.  Here a block of four waves takes
// one (candidate, changed group of scale 3): the group's four rows are eight rows of scale 2 — 8 x 64 pixels, two 8 x 8 tiles
// per wave, one pixel per lane and tile, averaged from scale 1 (the candidate's own linear planes where that group changed, B's where
// not; an unchanged group of scale 2 recomputed from B's scale 1 has the very bits B stored) — and scale 3 follows by
// shuffles inside the tile, scales 4 and 5 from the 4 x 32 pixels of scale 3 in LDS: one trip to memory, one barrier.  The
// sums run in sparse_down_body's order ((a + b) + c) + d over rows 2y, 2y+1 and columns 2x, 2x+1.
// A changed group of scale 4 (5) spans two (four) groups of scale 3, not all of them changed: the rows under an unchanged
// one are B's, copied into the candidate's slot by the block of the group's changed neighbour (scale 5: the first changed one).
// The candidates' linear planes of scales 2.. are read by nobody and not stored.
// Needs G.nscales >= 4 (a scale 3) and W = 256 (scale 2 is 64 wide: eight tiles; scales 3.. are narrow: both layouts stored).
// A group is three dependent trips to memory (item -> the candidate's group tables -> the planes of scale 1) for half a
// microsecond of arithmetic, with the CU's wave slots full: a block takes U groups at a time, every trip made for all of
// them before the first is used.
template <int U>
__device__ __forceinline__ void sparse_down_tiles_body(const SparseParams &P, const int bx, const int gx) {
    // Four waves per block, two tiles per wave (until late in round 4: eight waves, a tile each — a block that needs eight free wave
    // slots of one CU at once is never placed while a kernel of one- or four-wave blocks keeps the chip full: beside the H pass of
    // scale 0 this kernel took 211 us instead of 93, beside the V pass 627)
    constexpr int TPW = 2;
    __shared__ float s_v3[U][3][4][32];
    const Geom &G = P.G;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ly = lane >> 3, lx = lane & 7;
    const int total = P.item_count[3 * kColBuckets] / 3;
    const int NG3 = G.sh[3] >> 2;
    const float *basep = P.store + (size_t)P.base * P.S.cand_stride;
    const unsigned int *items = P.items + (size_t)(3 * kColBuckets) * P.item_stride;
    const int Wp = G.sw[1], W2 = G.sw[2];
    for (int g0 = bx * U; g0 < total; g0 += gx * U) {
        unsigned int it[U];
#pragma unroll
        for (int u = 0; u < U; u++) it[u] = items[3 * (size_t)min(g0 + u, total - 1)]; // (past the list: the last group once more, nothing stored)
        short sl[U], sl2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const CandMeta *M = P.meta + item_k(it[u]);
            const int y2 = 8 * item_g(it[u]) + ly;
            sl[u] = M->gslot[P.S.goff[1] + (y2 >> 1)];
            sl2[u] = M->gslot[P.S.goff[2] + (y2 >> 2)];
        }
        // ---- scale 2: pixel (8 ty + ly, 8 tx + lx) from rows 2y, 2y+1 of scale 1 (one group: 4 ty + ly / 2) ----
        float2 a[U][TPW][3], b[U][TPW][3];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float *mine = P.store + (size_t)item_k(it[u]) * P.S.cand_stride;
            const int gp = (8 * item_g(it[u]) + ly) >> 1, rp = (ly & 1) * 2;
            const float *grp = sl[u] >= 0 ? mine + P.S.off_lin[1] + (size_t)sl[u] * 12 * Wp : basep + P.S.off_lin[1] + (size_t)gp * 12 * Wp;
#pragma unroll
            for (int i = 0; i < TPW; i++) {
                const int x2 = 8 * (TPW * w + i) + lx;
                const float *q0 = grp + (size_t)((2 * x2) >> 2) * 16 + rp * 4 + ((2 * x2) & 3);
#pragma unroll
                for (int c = 0; c < 3; c++) { a[u][i][c] = *reinterpret_cast<const float2 *>(q0 + (size_t)c * 4 * Wp); b[u][i][c] = *reinterpret_cast<const float2 *>(q0 + (size_t)c * 4 * Wp + 4); }
            }
        }
        __syncthreads(); // (the block's last read of s_v3 for the groups before)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool live = g0 + u < total;
            float *mine = P.store + (size_t)item_k(it[u]) * P.S.cand_stride;
            const int y2 = 8 * item_g(it[u]) + ly;
#pragma unroll
            for (int i = 0; i < TPW; i++) {
            const int tx = TPW * w + i, x2 = 8 * tx + lx;
            float v2[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float sum = 0.0f;
                sum += a[u][i][c].x; sum += a[u][i][c].y; sum += b[u][i][c].x; sum += b[u][i][c].y;
                v2[c] = sum * 0.25f;
            }
            if (live && sl2[u] >= 0) { // (uniform over each half of the wave: rows 0-3 and 4-7 of the tile are the two groups of scale 2)
                float X, Y, B;
                linear_to_positive_xyb(v2[0], v2[1], v2[2], X, Y, B);
                float *oc = mine + P.S.off_xybC[2] + (size_t)sl2[u] * 12 * W2 + (size_t)(x2 >> 2) * 16 + (y2 & 3) * 4 + (x2 & 3);
                oc[0] = X; oc[(size_t)4 * W2] = Y; oc[(size_t)8 * W2] = B;
            }
            // scale 3 inside the tile: the lanes of even row and column hold pixel (4 ty + ly / 2, 4 tx + lx / 2)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float b1 = __shfl(v2[c], (lane + 1) & 63), c1 = __shfl(v2[c], (lane + 8) & 63), d1 = __shfl(v2[c], (lane + 9) & 63);
                float sum = 0.0f;
                sum += v2[c]; sum += b1; sum += c1; sum += d1;
                if (((ly | lx) & 1) == 0) s_v3[u][c][ly >> 1][4 * tx + (lx >> 1)] = sum * 0.25f;
            }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (g0 + u >= total) break;
            const int k = item_k(it[u]), ty = item_g(it[u]), j3 = item_j(it[u]);
            const CandMeta *M = P.meta + k;
            float *mine = P.store + (size_t)k * P.S.cand_stride;
            const float (*v3)[4][32] = s_v3[u];
            if (t < 128) { // scale 3: 4 rows x 32 pixels
                const int r = t >> 5, x = t & 31, Ws = G.sw[3];
                float X, Y, B;
                linear_to_positive_xyb(v3[0][r][x], v3[1][r][x], v3[2][r][x], X, Y, B);
                float *oc = mine + P.S.off_xybC[3] + (size_t)j3 * 12 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3);
                float *orr = mine + P.S.off_xybR[3] + (size_t)j3 * 12 * Ws + (size_t)x * 4 + r;
                oc[0] = X; oc[(size_t)4 * Ws] = Y; oc[(size_t)8 * Ws] = B;
                orr[0] = X; orr[(size_t)4 * Ws] = Y; orr[(size_t)8 * Ws] = B;
            } else if (t < 192) { // scale 4: rows 2 ty, 2 ty + 1 x 16 pixels, then (below) the rows under an unchanged neighbour
                const int i = t - 128;
                if (G.nscales > 4 && i < 32) {
                    const int a4 = i >> 4, x = i & 15, Ws = G.sw[4];
                    const int y4 = 2 * ty + a4, r = y4 & 3;
                    const int j4 = M->gslot[P.S.goff[4] + (y4 >> 2)];
                    float v[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        float sum = 0.0f;
                        sum += v3[c][2 * a4][2 * x]; sum += v3[c][2 * a4][2 * x + 1]; sum += v3[c][2 * a4 + 1][2 * x]; sum += v3[c][2 * a4 + 1][2 * x + 1];
                        v[c] = sum * 0.25f;
                    }
                    float X, Y, B;
                    linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);
                    float *oc = mine + P.S.off_xybC[4] + (size_t)j4 * 12 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3);
                    float *orr = mine + P.S.off_xybR[4] + (size_t)j4 * 12 * Ws + (size_t)x * 4 + r;
                    oc[0] = X; oc[(size_t)4 * Ws] = Y; oc[(size_t)8 * Ws] = B;
                    orr[0] = X; orr[(size_t)4 * Ws] = Y; orr[(size_t)8 * Ws] = B;
                }
            } else if (t < 256) { // scale 5: row ty x 8 pixels
                const int x = t - 192;
                if (G.nscales > 5 && x < 8) {
                    const int Ws = G.sw[5], r = ty & 3;
                    const int j5 = M->gslot[P.S.goff[5] + (ty >> 2)];
                    float v[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        float v4[2][2];
#pragma unroll
                        for (int a4 = 0; a4 < 2; a4++)
#pragma unroll
                            for (int e = 0; e < 2; e++) {
                                const int x4 = 2 * x + e;
                                float sum = 0.0f;
                                sum += v3[c][2 * a4][2 * x4]; sum += v3[c][2 * a4][2 * x4 + 1]; sum += v3[c][2 * a4 + 1][2 * x4]; sum += v3[c][2 * a4 + 1][2 * x4 + 1];
                                v4[a4][e] = sum * 0.25f;
                            }
                        float sum = 0.0f;
                        sum += v4[0][0]; sum += v4[0][1]; sum += v4[1][0]; sum += v4[1][1];
                        v[c] = sum * 0.25f;
                    }
                    float X, Y, B;
                    linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);
                    float *oc = mine + P.S.off_xybC[5] + (size_t)j5 * 12 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3);
                    float *orr = mine + P.S.off_xybR[5] + (size_t)j5 * 12 * Ws + (size_t)x * 4 + r;
                    oc[0] = X; oc[(size_t)4 * Ws] = Y; oc[(size_t)8 * Ws] = B;
                    orr[0] = X; orr[(size_t)4 * Ws] = Y; orr[(size_t)8 * Ws] = B;
                }
            }
            if (t >= 128 && t < 192) { // scale 4 (the wave that took its pixels): the two rows under the unchanged other half of the group are B's
                const int sib = ty ^ 1;
                if (G.nscales > 4 && sib < NG3 && M->gslot[P.S.goff[3] + sib] < 0) {
                    const int Ws = G.sw[4], g4 = ty >> 1;
                    const int j4 = M->gslot[P.S.goff[4] + g4];
                    const float *bc = basep + P.S.off_xybC[4] + (size_t)g4 * 12 * Ws, *br = basep + P.S.off_xybR[4] + (size_t)g4 * 12 * Ws;
                    float *oc = mine + P.S.off_xybC[4] + (size_t)j4 * 12 * Ws, *orr = mine + P.S.off_xybR[4] + (size_t)j4 * 12 * Ws;
                    for (int i = t - 128; i < 3 * 2 * Ws; i += 64) { // (channel, row of the pair, column)
                        const int c = i / (2 * Ws), rem = i % (2 * Ws), r = ((2 * sib) & 3) + rem / Ws, x = rem % Ws;
                        const size_t ic = (size_t)c * 4 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3), ir = (size_t)c * 4 * Ws + (size_t)x * 4 + r;
                        oc[ic] = bc[ic]; orr[ir] = br[ir];
                    }
                }
            } else if (t >= 192) { // scale 5 (the wave that took its pixels): the group's first changed group of scale 3 brings B's rows under the unchanged ones
                if (G.nscales > 5) {
                    const int g5 = ty >> 2, lo = 4 * g5, hi = min(lo + 4, NG3);
                    bool first = true;
                    for (int q = lo; q < ty; q++) first = first && M->gslot[P.S.goff[3] + q] < 0;
                    if (first) {
                        const int Ws = G.sw[5];
                        const int j5 = M->gslot[P.S.goff[5] + g5];
                        const float *bc = basep + P.S.off_xybC[5] + (size_t)g5 * 12 * Ws, *br = basep + P.S.off_xybR[5] + (size_t)g5 * 12 * Ws;
                        float *oc = mine + P.S.off_xybC[5] + (size_t)j5 * 12 * Ws, *orr = mine + P.S.off_xybR[5] + (size_t)j5 * 12 * Ws;
                        for (int q = lo; q < hi; q++) {
                            if (q == ty || M->gslot[P.S.goff[3] + q] >= 0) continue;
                            const int r = q & 3;
                            for (int i = t - 192; i < 3 * Ws; i += 64) {
                                const int c = i / Ws, x = i % Ws;
                                const size_t ic = (size_t)c * 4 * Ws + (size_t)(x >> 2) * 16 + r * 4 + (x & 3), ir = (size_t)c * 4 * Ws + (size_t)x * 4 + r;
                                oc[ic] = bc[ic]; orr[ir] = br[ir];
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- B's downscale chain in one launch: one block = one 32x32 block of scale-0 pixels -> 16x16, 8x8, ... 1x1 ----
// (same arithmetic as k_sparse_down / k_downscale_chain; only the store addresses follow B's group-major layouts)
__device__ __forceinline__ void base_store(const SparseParams &P, float *mine, int s, int X, int Y, const float *lin, bool keep_lin) {
    const int Ws = P.G.sw[s];
    const int g = Y >> 2, r = Y & 3;
    float Xc, Yc, Bc;
    linear_to_positive_xyb(lin[0], lin[1], lin[2], Xc, Yc, Bc);
    const float xyb[3] = {Xc, Yc, Bc};
    float *ol = mine + P.S.off_lin[s] + (size_t)g * 12 * Ws;
    float *oc = mine + P.S.off_xybC[s] + (size_t)g * 12 * Ws, *orr = mine + P.S.off_xybR[s] + (size_t)g * 12 * Ws;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (keep_lin) ol[(size_t)c * 4 * Ws + (size_t)(X >> 2) * 16 + r * 4 + (X & 3)] = lin[c];
        oc[(size_t)c * 4 * Ws + (size_t)(X >> 2) * 16 + r * 4 + (X & 3)] = xyb[c];
        orr[(size_t)c * 4 * Ws + (size_t)X * 4 + r] = xyb[c];
    }
}
__device__ __forceinline__ void base_down_body(const SparseParams &P) {
    __shared__ float s_lin[256 * 3];
    __shared__ float l1[3][16][17], l2[3][8][9], l3[3][4][5], l4[3][2][3];
    const Geom &G = P.G;
    const int t = threadIdx.x;
    const int bx = blockIdx.x % (G.W / 32), by = blockIdx.x / (G.W / 32);
    for (int i = t; i < (P.ncol + 2) * 3; i += 256) s_lin[i] = P.pal_lin[i];
    __syncthreads();
    float *mine = P.store + (size_t)P.base * P.S.cand_stride;
    {
        const int lx = t & 15, ly = t >> 4;
        const int X1 = bx * 16 + lx, Y1 = by * 16 + ly;
        const bool in1 = Y1 < G.sh[1]; // (images of 8 and 16 rows: a block of 32 x 32 pixels reaches below the image)
        float sum[3] = {0.0f, 0.0f, 0.0f};
        if (in1) {
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) {
                    const uint32_t lo = (uint32_t)P.pack[(size_t)(Y1 * 2 + iy) * G.W + X1 * 2 + ix];
                    const uint32_t ci = P.use_maps ? maps_ci(P, P.bmap, true, X1 * 2 + ix, Y1 * 2 + iy, lo) : lo >> 24;
                    sum[0] += s_lin[3 * ci]; sum[1] += s_lin[3 * ci + 1]; sum[2] += s_lin[3 * ci + 2];
                }
        }
        const float v[3] = {sum[0] * 0.25f, sum[1] * 0.25f, sum[2] * 0.25f};
        l1[0][ly][lx] = v[0]; l1[1][ly][lx] = v[1]; l1[2][ly][lx] = v[2];
        if (in1) base_store(P, mine, 1, X1, Y1, v, G.nscales > 2);
    }
    __syncthreads();
#define SNES_BASE_LEVEL(S, SRC, DST, DIM)                                                              \
    if (G.nscales > S) {                                                                               \
        if (t < DIM * DIM) {                                                                           \
            const int lx = t % DIM, ly = t / DIM;                                                      \
            float v[3];                                                                                \
            for (int c = 0; c < 3; c++) {                                                              \
                float sum = 0.0f;                                                                      \
                sum += SRC[c][2 * ly][2 * lx]; sum += SRC[c][2 * ly][2 * lx + 1];                      \
                sum += SRC[c][2 * ly + 1][2 * lx]; sum += SRC[c][2 * ly + 1][2 * lx + 1];              \
                v[c] = sum * 0.25f; DST[c][ly][lx] = v[c];                                             \
            }                                                                                          \
            if (by * DIM + ly < G.sh[S]) base_store(P, mine, S, bx * DIM + lx, by * DIM + ly, v, G.nscales > S + 1); \
        }                                                                                              \
        __syncthreads();                                                                               \
    }
    SNES_BASE_LEVEL(2, l1, l2, 8)
    SNES_BASE_LEVEL(3, l2, l3, 4)
    SNES_BASE_LEVEL(4, l3, l4, 2)
    if (G.nscales > 5 && t == 0) {
        float v[3];
        for (int c = 0; c < 3; c++) { float sum = 0.0f; sum += l4[c][0][0]; sum += l4[c][0][1]; sum += l4[c][1][0]; sum += l4[c][1][1]; v[c] = sum * 0.25f; }
        if (by < G.sh[5]) base_store(P, mine, 5, bx, by, v, false);
    }
#undef SNES_BASE_LEVEL
}

// recurrence steps, identical to kernels_fast.hpp
#define SNES_HSTEP(SUM, A, B, OUT)                                                   \
    {                                                                                \
        float o1_ = (SUM) * n2_0, o3_ = (SUM) * n2_1, o5_ = (SUM) * n2_2;            \
        o1_ = fmaf(-1.0f, B[0], o1_); o3_ = fmaf(-1.0f, B[1], o3_); o5_ = fmaf(-1.0f, B[2], o5_); \
        o1_ = fmaf(mp_0, A[0], o1_); o3_ = fmaf(mp_1, A[1], o3_); o5_ = fmaf(mp_2, A[2], o5_);    \
        B[0] = o1_; B[1] = o3_; B[2] = o5_;                                          \
        OUT = o1_ + o3_ + o5_;                                                       \
    }
#define SNES_VSTEP(SUM, A, B, OUT)                                                   \
    {                                                                                \
        float o1_ = fmaf(A[0], d1_0, B[0]), o3_ = fmaf(A[1], d1_1, B[1]), o5_ = fmaf(A[2], d1_2, B[2]); \
        o1_ = fmaf((SUM), n2_0, -o1_); o3_ = fmaf((SUM), n2_1, -o3_); o5_ = fmaf((SUM), n2_2, -o5_);    \
        B[0] = o1_; B[1] = o3_; B[2] = o5_;                                          \
        OUT = o1_ + o3_ + o5_;                                                       \
    }

// ---- H pass of changed groups: one lane quad = the four rows of one (candidate, group slot, channel) ----
__device__ __forceinline__ void sparse_h_body(const SparseParams &P) {
    __shared__ float s_lut[3][256];
    __shared__ float s_tr[3][64 * 5];
    const Geom &G = P.G;
    const int list = P.s_first * kColBuckets + (int)blockIdx.y, s = list / kColBuckets; // one item list per (scale, first column block); grid.y = the narrow scales' lists
    if (s >= G.nscales) return;
    const int lane = threadIdx.x;
    const int count = P.item_count[list];
    if ((int)blockIdx.x * 16 >= count) return;
    const bool S0 = (s == 0);
    if (S0) {
        for (int i = lane; i < 3 * 256; i += 64) { int c = i >> 8, j = i & 255; s_lut[c][j] = (j < P.ncol + 2) ? P.pal_xyb[3 * j + c] : 0.0f; }
        __syncthreads();
    }
    for (int i0 = blockIdx.x * 16; i0 < count; i0 += gridDim.x * 16) { // grid-stride over item quads: no empty waves
    const int qi = i0 + (lane >> 2);
    const bool valid = qi < count;
    const unsigned int it = P.items[(size_t)list * P.item_stride + (valid ? qi : i0)];
    const int k = item_k(it), j = item_j(it), ch = item_ch(it);
    const bool is_base = (k == P.base);
    const int W = G.sw[s], H = G.sh[s];
    const int r = lane & 3;
    const int y = 4 * item_g(it) + r;
    const size_t ns = (size_t)W * H;
    const float cand_v = is_base ? 0.0f : P.cand_tab[8 * (size_t)k + 3 + ch];
    const uint32_t crgb = is_base ? 0u : __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    const uint32_t never = is_base ? 0u : 0xffffffffu; // thr & never == 0 for B: it wins nothing
    const float4 *in1 = reinterpret_cast<const float4 *>(P.img1C4 + G.src_off[s] + (size_t)ch * ns) + y;                   // C4: + g*H
    const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.packC4) + 2 * (size_t)y : nullptr;                            // + g*2H
    const float4 *in2 = S0 ? nullptr : reinterpret_cast<const float4 *>(P.store + (size_t)k * P.S.cand_stride + P.S.off_xybC[s] + (size_t)j * 12 * W + (size_t)ch * 4 * W) + r; // + g*4
    float *out = P.store + (size_t)k * P.S.cand_stride + P.S.off_hout[s] + (size_t)j * 36 * W + (size_t)(ch * 3) * 4 * W;

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
    float sa[3][3], sb[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) { sa[p][q] = 0.0f; sb[p][q] = 0.0f; }
    float4 r1[5], r2[5];
#pragma unroll
    for (int a = 0; a < 5; a++) { r1[a] = make_float4(0.f, 0.f, 0.f, 0.f); r2[a] = r1[a]; }
    const int G4 = W >> 2;
    uint4 n_pa = make_uint4(0, 0, 0, 0), n_pb = n_pa;
    r1[0] = in1[0];
    if (S0) { n_pa = pk[0]; n_pb = pk[1]; } else r2[0] = in2[0];
    const int li = lane & 3, k4 = lane & ~3;
    for (int g0 = 0; g0 <= G4; g0 += 5) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int g = g0 + u;
            if (g > G4) break;
            const int un = (u + 1) % 5, ua = (u + 2) % 5, ub = (u + 3) % 5;
            const uint4 c_pa = n_pa, c_pb = n_pb;
            if (g + 1 < G4) {
                r1[un] = in1[(size_t)(g + 1) * H];
                if (S0) { n_pa = pk[(size_t)(g + 1) * H * 2]; n_pb = pk[(size_t)(g + 1) * H * 2 + 1]; } else r2[un] = in2[(size_t)(g + 1) * 4];
            } else { r1[un] = make_float4(0.f, 0.f, 0.f, 0.f); r2[un] = r1[un]; n_pa = make_uint4(0, 0, 0, 0); n_pb = n_pa; }
            if (S0 && g < G4) {
                uint32_t c0, c1, c2, c3;
                if (P.perceptual && !is_base) { // four consecutive pixels of row y: four consecutive bits of one bitmap word
                    const int px0 = y * W + (g << 2);
                    const uint32_t b4 = (P.bitmap[(size_t)k * (G.W * G.H / 32) + (px0 >> 5)] >> (px0 & 31)) & 0xfu;
                    c0 = (b4 & 1u) ? (uint32_t)P.ncol : (c_pa.x >> 24); c1 = (b4 & 2u) ? (uint32_t)P.ncol : (c_pa.z >> 24);
                    c2 = (b4 & 4u) ? (uint32_t)P.ncol : (c_pb.x >> 24); c3 = (b4 & 8u) ? (uint32_t)P.ncol : (c_pb.z >> 24);
                } else {
                    c0 = sparse_ci(c_pa.x, c_pa.y & never, crgb, (uint32_t)P.ncol); c1 = sparse_ci(c_pa.z, c_pa.w & never, crgb, (uint32_t)P.ncol);
                    c2 = sparse_ci(c_pb.x, c_pb.y & never, crgb, (uint32_t)P.ncol); c3 = sparse_ci(c_pb.z, c_pb.w & never, crgb, (uint32_t)P.ncol);
                }
                r2[u].x = c0 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c0]; r2[u].y = c1 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c1];
                r2[u].z = c2 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c2]; r2[u].w = c3 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c3];
            }
            const float v1[4] = {r1[u].x, r1[u].y, r1[u].z, r1[u].w}, v2[4] = {r2[u].x, r2[u].y, r2[u].z, r2[u].w};
            const float l1[4] = {r1[ua].z, r1[ua].w, r1[ub].x, r1[ub].y}, l2[4] = {r2[ua].z, r2[ua].w, r2[ub].x, r2[ub].y};
            float outp[3][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float s0 = l2[q] + v2[q];
                const float s1 = (l2[q] * l2[q]) + (v2[q] * v2[q]);
                const float s2 = (l1[q] * l2[q]) + (v1[q] * v2[q]);
                if ((q & 1) == 0) { SNES_HSTEP(s0, sa[0], sb[0], outp[0][q]) SNES_HSTEP(s1, sa[1], sb[1], outp[1][q]) SNES_HSTEP(s2, sa[2], sb[2], outp[2][q]) }
                else { SNES_HSTEP(s0, sb[0], sa[0], outp[0][q]) SNES_HSTEP(s1, sb[1], sa[1], outp[1][q]) SNES_HSTEP(s2, sb[2], sa[2], outp[2][q]) }
            }
            if (g >= 1) { // 4x4 transpose inside the quad: lane li then holds column 4(g-1)+li of the group's four rows
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    s_tr[p][lane * 5 + 0] = outp[p][0]; s_tr[p][lane * 5 + 1] = outp[p][1]; s_tr[p][lane * 5 + 2] = outp[p][2]; s_tr[p][lane * 5 + 3] = outp[p][3];
                }
                __syncthreads();
                const int x = ((g - 1) << 2) + li;
                const size_t o = ((size_t)(x >> 6) << 8) + ((size_t)(x & 63) << 2);
                if (valid) {
#pragma unroll
                    for (int p = 0; p < 3; p++) {
                        float4 v;
                        v.x = s_tr[p][(k4 + 0) * 5 + li]; v.y = s_tr[p][(k4 + 1) * 5 + li]; v.z = s_tr[p][(k4 + 2) * 5 + li]; v.w = s_tr[p][(k4 + 3) * 5 + li];
                        *reinterpret_cast<float4 *>(out + (size_t)p * 4 * W + o) = v;
                    }
                }
                __syncthreads();
            }
        }
    }
    }
}

// ---- V pass + maps, resumed from B's checkpoint at the first changed group --------------------------------
// block = 256 threads = 256/W_s (candidate, channel) pairs; thread = one image column.
// Checkpoint record g (0 <= g <= H/4 + 1) of B: recurrence state before group iteration g (steps 4g-4..4g-1)
// and pooling sums of rows < 4g-4; record H/4+1 holds the final sums.
// Register budget: 128 VGPRs, i.e. four waves per SIMD (the kernel is bound by VALU issue and latency, and went
// 1.12 -> 0.82 ms per 1,024 candidates from two to three waves).  Only the current and the next input group live in
// registers; the trailing filter taps — rows 4g-10..4g-7, i.e. the second half of group g-3 and the first half of
// group g-2 — come back from a per-thread LDS ring (read, then overwritten by the same thread: no barrier), and the
// map inputs are loaded at the top of the iteration that consumes them, behind the twelve recurrence steps.
// UNI: scales at least 64 wide — a wave's 64 columns belong to one (candidate, channel) pair, so the first changed
// group, the loop counter, every border test and the group -> slot choice are wave-uniform and live in SGPRs (scalar
// branches instead of exec-mask regions with zero-filled defaults).  A wave skips only if all its columns do; a skipped
// column inside a working wave recomputes B's values from the same checkpoint, bit for bit.
// S0M / BM: whether the launch is scale 0 alone (pixels from the pack, not from XYB planes) / scores the image B —
// 0 no, 1 yes, 2 decided at run time.  The candidates' launch keeps both at 2: the register allocator lands on 13 spills
// at 128 VGPRs there and on 30-70 with either specialised; B's launches are specialised and take 2 waves' worth of registers.
template <bool UNI, int S0M, int BM>
__device__ __forceinline__ void sparse_v_body(const SparseParams &P, const int s) {
    __shared__ float s_lut[256];
    __shared__ short s_gslot[256];
    // tails: xy halves two deep, zw halves three deep, [slot][plane][thread]; the pooling reduction reuses the space
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[(2 + 3) * 3 * 256 * sizeof(float2)];
    float2 (*s_xy)[3][256] = reinterpret_cast<float2 (*)[3][256]>(s_raw);
    float2 (*s_zw)[3][256] = reinterpret_cast<float2 (*)[3][256]>(s_raw + 2 * 3 * 256 * sizeof(float2));
    double (*red)[6] = reinterpret_cast<double (*)[6]>(s_raw);
    static_assert(sizeof(s_raw) >= 256 * 6 * sizeof(double), "reduction scratch must fit the tail ring");
    const Geom &G = P.G;
    if (s >= G.nscales) return;
    const int W = G.sw[s], H = G.sh[s];
    const bool S0 = S0M == 2 ? s == 0 : S0M == 1;
    const int t = threadIdx.x;
    const int ppw = 256 / W;
    const bool is_base = BM == 2 ? P.is_base != 0 : BM == 1;
    const int npairs = is_base ? 3 : P.ncand * 3;
    if ((int)blockIdx.x * ppw >= npairs) return;
    const int ql = t / W, x = t - ql * W;
    const int pair_raw = blockIdx.x * ppw + ql;
    const bool active = pair_raw < npairs;
    const int pair = active ? pair_raw : 0;
    const int k = is_base ? P.base : P.k0 + (P.order ? P.order[pair / 3] : pair / 3), ch = pair % 3;
    if (S0) { // W = 256: the block is one (candidate, channel) pair, so one channel of the XYB table is enough
        s_lut[t] = (t < P.ncol + 2) ? P.pal_xyb[3 * t + ch] : 0.0f;
    }
    const size_t ns = (size_t)W * H;
    const int H4 = H >> 2;
    const CandMeta *M = P.meta + k;
    // group -> slot table of this block's pairs, staged in LDS so the per-group pointer choice costs no global load
    for (int i = t; i < ppw * H4; i += 256) {
        const int pr = blockIdx.x * ppw + i / H4;
        const int pq = (pr < npairs ? pr : 0) / 3;
        const int kk = is_base ? P.base : P.k0 + (P.order ? P.order[pq] : pq);
        s_gslot[i] = (P.meta + kk)->gslot[P.S.goff[s] + i % H4];
    }
    __syncthreads();
    const short *gslot = s_gslot + ql * H4;
    const int ng = M->ngroups[s];
    const int cmin = (M->xmin >> s) - 5; // columns <= cmin see only unchanged inputs
    const bool skip = !is_base && (ng == 0 || x <= cmin);
    int gs = skip ? H4 + 1 : (is_base ? 0 : (int)M->glist[P.S.goff[s]]);
    if (UNI) { // the smallest first-group of the wave: the pair's, unless every column skips
        const int gs_pair = is_base ? 0 : (ng == 0 ? H4 + 1 : (int)M->glist[P.S.goff[s]]);
        gs = __builtin_amdgcn_readfirstlane(__all(skip) ? H4 + 1 : gs_pair);
    }
#define SNES_GSLOT(I) (UNI ? __builtin_amdgcn_readfirstlane((int)gslot[I]) : (int)gslot[I])
    const float cand_v = is_base ? 0.0f : P.cand_tab[8 * (size_t)k + 3 + ch];
    const uint32_t crgb = is_base ? 0u : __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    const uint32_t never = is_base ? 0u : 0xffffffffu;
    const float *mine = P.store + (size_t)k * P.S.cand_stride;
    const float *basep = P.store + (size_t)P.base * P.S.cand_stride;
    // hout of group g, plane p: float4 index (p*W) + ((x>>6)<<6) + (x&63) inside the group's 9*W float4s
    const size_t hx = ((size_t)(x >> 6) << 6) + (size_t)(x & 63);
    const float4 *hb = reinterpret_cast<const float4 *>(basep + P.S.off_hout[s]) + (size_t)(ch * 3) * W + hx;  // + g*9W
    const float4 *hm = reinterpret_cast<const float4 *>(mine + P.S.off_hout[s]) + (size_t)(ch * 3) * W + hx;   // + slot*9W
    const float4 *xb = S0 ? nullptr : reinterpret_cast<const float4 *>(basep + P.S.off_xybR[s]) + (size_t)ch * W + x; // + g*3W
    const float4 *xm = S0 ? nullptr : reinterpret_cast<const float4 *>(mine + P.S.off_xybR[s]) + (size_t)ch * W + x;
    const float4 *mu1 = reinterpret_cast<const float4 *>(P.mu1R4 + G.src_off[s] + (size_t)ch * ns) + x; // + g*W
    const float4 *sd1 = reinterpret_cast<const float4 *>(P.sd1R4 + G.src_off[s] + (size_t)ch * ns) + x;
    const float4 *a1 = reinterpret_cast<const float4 *>(P.a1R4 + G.src_off[s] + (size_t)ch * ns) + x;
    const double2 *r1 = reinterpret_cast<const double2 *>(P.r1R4 + G.src_off[s] + (size_t)ch * ns) + 2 * (size_t)x; // + g*2W
    const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.packR4) + 2 * (size_t)x : nullptr; // + g*2W
    // checkpoints: ckf[s][ch][g][18][W], cka[s][ch][g][6][W]
    float *ckf = P.ckf + P.S.off_ckf[s] + (size_t)ch * (H4 + 2) * 18 * W + x;
    double *cka = P.cka + P.S.off_cka[s] + (size_t)ch * (H4 + 2) * 6 * W + x;

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
    float sa[3][3], sb[3][3];
    double acc[6];
    if (is_base) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = 0.0f; sb[p][q] = 0.0f; }
#pragma unroll
        for (int q = 0; q < 6; q++) acc[q] = 0.0;
    } else {
        const float *cf = ckf + (size_t)gs * 18 * W;
        const double *ca = cka + (size_t)gs * 6 * W;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = cf[(size_t)(p * 6 + q) * W]; sb[p][q] = cf[(size_t)(p * 6 + 3 + q) * W]; }
#pragma unroll
        for (int q = 0; q < 6; q++) acc[q] = ca[(size_t)q * W];
    }
    // Offsets stay 32-bit and come from 24-bit multiplies (groups <= 64, 9W <= 2,304 float4s): the 64-bit
    // multiplies of plain size_t indexing were a fifth of the loop's issue slots.
    const int W9 = 9 * W, W3 = 3 * W;
    auto hgroup = [&](int g) -> const float4 * { // plane 0 of group g: the candidate's if it changed, else B's
        const int sl = SNES_GSLOT(g);
        return sl >= 0 ? hm + (uint32_t)__mul24(sl, W9) : hb + (uint32_t)__mul24(g, W9);
    };
    auto hload = [&](int g, int p) -> float4 { return hgroup(g)[(uint32_t)__mul24(p, W)]; };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // Tail ring slots are relative to gs: group gs+r keeps its xy half in slot r mod 2 and its zw half in slot r mod 3.
    // Groups gs-3 .. gs-1 precede the first changed group (they are B's, or zero above the image).
    float4 bufA[3], bufB[3]; // current / next input group, roles alternate per iteration
#pragma unroll
    for (int p = 0; p < 3; p++) {
        float4 g1 = zero4, g2 = zero4, g3 = zero4;
        bufA[p] = zero4; bufB[p] = zero4;
        if (UNI ? gs <= H4 : !skip) { // (a skipping wave has gs = H4+1: nothing to fetch)
            if (gs < H4) bufA[p] = hload(gs, p);
            if (gs - 1 >= 0) g1 = hload(gs - 1, p);
            if (gs - 2 >= 0) g2 = hload(gs - 2, p);
            if (gs - 3 >= 0) g3 = hload(gs - 3, p);
        }
        s_xy[1][p][t] = make_float2(g1.x, g1.y); s_zw[2][p][t] = make_float2(g1.z, g1.w); // r = -1
        s_xy[0][p][t] = make_float2(g2.x, g2.y); s_zw[1][p][t] = make_float2(g2.z, g2.w); // r = -2
        s_zw[0][p][t] = make_float2(g3.z, g3.w);                                          // r = -3
    }
    // running pointers of the map inputs of row group g-1 (valid from g = 1 on)
    const int gm0 = gs >= 1 ? gs - 1 : 0;
    const float4 *p_m1 = mu1 + (uint32_t)__mul24(gm0, W), *p_sd1 = sd1 + (uint32_t)__mul24(gm0, W), *p_a1 = a1 + (uint32_t)__mul24(gm0, W);
    const double2 *p_r1 = r1 + 2u * (uint32_t)__mul24(gm0, W);
    const uint4 *p_pk = S0 ? pk + 2u * (uint32_t)__mul24(gm0, W) : nullptr;
    const uint32_t *bm = (S0 && P.perceptual && !is_base) ? P.bitmap + (size_t)k * (G.W * G.H / 32) : nullptr;
    float4 n_m1 = zero4, n_sd1 = zero4, n_a1 = zero4, n_x = zero4; // BM == 1 only
    double2 n_ra = make_double2(1.0, 1.0), n_rb = n_ra;
    uint4 n_pa = make_uint4(0, 0, 0, 0), n_pb = n_pa;
    float *ck_f = BM == 1 ? ckf + (size_t)gs * 18 * W : nullptr; // B only: running pointers of the record being written
    double *ck_a = BM == 1 ? cka + (size_t)gs * 6 * W : nullptr;
#define SNES_VGROUP(U, CUR, NXT)                                                                                              \
    {                                                                                                                         \
        const int g = g0 + (U);                                                                                               \
        if (g > H4) break;                                                                                                    \
        if (BM == 2 && is_base && active) { /* record g (run-time flavour) */                                                 \
            float *cf = ckf + (size_t)g * 18 * W;                                                                             \
            double *ca = cka + (size_t)g * 6 * W;                                                                             \
            _Pragma("unroll") for (int p = 0; p < 3; p++)                                                                     \
                _Pragma("unroll") for (int q = 0; q < 3; q++) { cf[(size_t)(p * 6 + q) * W] = sa[p][q]; cf[(size_t)(p * 6 + 3 + q) * W] = sb[p][q]; } \
            _Pragma("unroll") for (int q = 0; q < 6; q++) ca[(size_t)q * W] = acc[q];                                         \
        }                                                                                                                     \
        if (BM == 1 && active) { /* record g */                                                                               \
            _Pragma("unroll") for (int p = 0; p < 3; p++)                                                                     \
                _Pragma("unroll") for (int q = 0; q < 3; q++) {                                                               \
                    ck_f[(uint32_t)__mul24(p * 6 + q, W)] = sa[p][q]; ck_f[(uint32_t)__mul24(p * 6 + 3 + q, W)] = sb[p][q];   \
                }                                                                                                             \
            _Pragma("unroll") for (int q = 0; q < 6; q++) ck_a[(uint32_t)__mul24(q, W)] = acc[q];                             \
            ck_f += 18 * W; ck_a += 6 * W;                                                                                    \
        }                                                                                                                     \
        if (g + 1 < H4) { const float4 *hp = hgroup(g + 1); NXT[0] = hp[0]; NXT[1] = hp[W]; NXT[2] = hp[2 * W]; }             \
        else { NXT[0] = zero4; NXT[1] = zero4; NXT[2] = zero4; }                                                              \
        /* inputs of the maps of row group g-1, consumed after the recurrence steps below */                                  \
        float4 c_m1 = zero4, c_sd1 = zero4, c_a1 = zero4, c_x = zero4;                                                        \
        double2 c_ra = make_double2(1.0, 1.0), c_rb = c_ra;                                                                   \
        uint4 c_pa = make_uint4(0, 0, 0, 0), c_pb = c_pa;                                                                     \
        if (BM == 1) { /* B (single image, registers to spare): fetched one iteration ahead, group g now for iteration g+1 */  \
            c_m1 = n_m1; c_sd1 = n_sd1; c_a1 = n_a1; c_ra = n_ra; c_rb = n_rb; c_x = n_x; c_pa = n_pa; c_pb = n_pb;            \
            if (g < H4) {                                                                                                     \
                n_m1 = *p_m1; n_sd1 = *p_sd1; n_a1 = *p_a1; n_ra = p_r1[0]; n_rb = p_r1[1];                                   \
                p_m1 += W; p_sd1 += W; p_a1 += W; p_r1 += 2 * W;                                                              \
                if (S0) { n_pa = p_pk[0]; n_pb = p_pk[1]; p_pk += 2 * W; }                                                    \
                else n_x = xb[(uint32_t)__mul24(g, W3)];                                                                      \
            }                                                                                                                 \
        } else if (g >= 1) {                                                                                                  \
            c_m1 = *p_m1; c_sd1 = *p_sd1; c_a1 = *p_a1; c_ra = p_r1[0]; c_rb = p_r1[1];                                       \
            p_m1 += W; p_sd1 += W; p_a1 += W; p_r1 += 2 * W;                                                                  \
            if (S0) { c_pa = p_pk[0]; c_pb = p_pk[1]; p_pk += 2 * W; }                                                        \
            else { const int sl = SNES_GSLOT(g - 1); c_x = sl >= 0 ? xm[(uint32_t)__mul24(sl, W3)] : xb[(uint32_t)__mul24(g - 1, W3)]; } \
        }                                                                                                                     \
        float outp[3][4];                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 3; p++) {                                                                       \
            const float2 tz = s_zw[(U) % 3][p][t], tx = s_xy[(U) % 2][p][t]; /* group g-3 second half, g-2 first half */      \
            s_zw[(U) % 3][p][t] = make_float2(CUR[p].z, CUR[p].w);                                                            \
            s_xy[(U) % 2][p][t] = make_float2(CUR[p].x, CUR[p].y);                                                            \
            SNES_VSTEP(tz.x + CUR[p].x, sa[p], sb[p], outp[p][0])                                                             \
            SNES_VSTEP(tz.y + CUR[p].y, sb[p], sa[p], outp[p][1])                                                             \
            SNES_VSTEP(tx.x + CUR[p].z, sa[p], sb[p], outp[p][2])                                                             \
            SNES_VSTEP(tx.y + CUR[p].w, sb[p], sa[p], outp[p][3])                                                             \
        }                                                                                                                     \
        if (g >= 1) {                                                                                                         \
            float i2v[4] = {c_x.x, c_x.y, c_x.z, c_x.w};                                                                      \
            if (S0) {                                                                                                         \
                uint32_t c0, c1, c2, c3;                                                                                      \
                if (is_base || SNES_GSLOT(g - 1) < 0) { /* no pixel of an unchanged group is won: B's colour indices as they are */ \
                    c0 = c_pa.x >> 24; c1 = c_pa.z >> 24; c2 = c_pb.x >> 24; c3 = c_pb.z >> 24;                               \
                } else if (bm) { /* rows 4(g-1)..4(g-1)+3 of column x */                                                      \
                    const int px0 = ((g - 1) << 2) * W + x;                                                                   \
                    c0 = won_bit(bm, px0) ? (uint32_t)P.ncol : (c_pa.x >> 24); c1 = won_bit(bm, px0 + W) ? (uint32_t)P.ncol : (c_pa.z >> 24);               \
                    c2 = won_bit(bm, px0 + 2 * W) ? (uint32_t)P.ncol : (c_pb.x >> 24); c3 = won_bit(bm, px0 + 3 * W) ? (uint32_t)P.ncol : (c_pb.z >> 24);   \
                } else {                                                                                                      \
                    c0 = sparse_ci(c_pa.x, c_pa.y & never, crgb, (uint32_t)P.ncol); c1 = sparse_ci(c_pa.z, c_pa.w & never, crgb, (uint32_t)P.ncol);         \
                    c2 = sparse_ci(c_pb.x, c_pb.y & never, crgb, (uint32_t)P.ncol); c3 = sparse_ci(c_pb.z, c_pb.w & never, crgb, (uint32_t)P.ncol);         \
                }                                                                                                             \
                i2v[0] = c0 == (uint32_t)P.ncol ? cand_v : s_lut[c0]; i2v[1] = c1 == (uint32_t)P.ncol ? cand_v : s_lut[c1];   \
                i2v[2] = c2 == (uint32_t)P.ncol ? cand_v : s_lut[c2]; i2v[3] = c3 == (uint32_t)P.ncol ? cand_v : s_lut[c3];   \
            }                                                                                                                 \
            const float m1v[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w}, sd1v[4] = {c_sd1.x, c_sd1.y, c_sd1.z, c_sd1.w}, a1v[4] = {c_a1.x, c_a1.y, c_a1.z, c_a1.w}; \
            const double r1v[4] = {c_ra.x, c_ra.y, c_rb.x, c_rb.y};                                                           \
            _Pragma("unroll") for (int q = 0; q < 4; q++)                                                                     \
                maps_accumulate(acc, m1v[q], sd1v[q], a1v[q], r1v[q], outp[0][q], outp[1][q], outp[2][q], i2v[q]);            \
        }                                                                                                                     \
    }
    for (int g0 = gs; g0 <= H4; g0 += 6) {
        SNES_VGROUP(0, bufA, bufB) SNES_VGROUP(1, bufB, bufA) SNES_VGROUP(2, bufA, bufB)
        SNES_VGROUP(3, bufB, bufA) SNES_VGROUP(4, bufA, bufB) SNES_VGROUP(5, bufB, bufA)
    }
#undef SNES_VGROUP
#undef SNES_GSLOT
    if (is_base && active) { // final record
        double *ca = cka + (size_t)(H4 + 1) * 6 * W;
#pragma unroll
        for (int q = 0; q < 6; q++) ca[(size_t)q * W] = acc[q];
    }
    __syncthreads(); // the tail ring is dead: its space becomes the reduction scratch
#pragma unroll
    for (int q = 0; q < 6; q++) red[t][q] = active ? acc[q] : 0.0;
    __syncthreads();
    for (int stride = W >> 1; stride > 0; stride >>= 1) {
        if (x < stride) {
#pragma unroll
            for (int q = 0; q < 6; q++) red[t][q] += red[t + stride][q];
        }
        __syncthreads();
    }
    if (x == 0 && active) {
        double *o = P.part + (((size_t)k * G.nscales + s) * 3 + ch) * 6;
#pragma unroll
        for (int q = 0; q < 6; q++) o[q] = red[t][q];
    }
}
// candidates: every scale in one launch, flags decided at run time (see sparse_v_body)
__global__ __launch_bounds__(256, 2) void k_sparse_v(SparseParams P) { sparse_v_body<false, 2, 2>(P, (int)blockIdx.y + P.s_first); } // grid.y = narrow scales
// B: one launch as well, each scale in its specialised flavour
__device__ __forceinline__ void sparse_v_base_body(const SparseParams &P) {
    const int s = (int)blockIdx.y;
    if (s >= P.G.nscales) return;
    // (scale 0 too reads its map input from the XYB plane B's H pass leaves behind, not from the pack: same values, and with
    // --dither the pack does not describe B)
    if (P.G.sw[s] >= 64) sparse_v_body<true, 0, 1>(P, s);
    else sparse_v_body<false, 0, 1>(P, s);
}
#undef SNES_HSTEP
#undef SNES_VSTEP

// contested pixels of a slot (thr != 0) -> compact list, built once per slot
__device__ __forceinline__ void build_plist_body(const unsigned long long *__restrict__ pack, int npx, uint4 *__restrict__ plist, int *__restrict__ count) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long w = 0;
    if (px < npx) w = pack[px];
    const uint32_t thr = (uint32_t)(w >> 32);
    const bool hit = thr != 0u;
    // one atomic per wave: the leader reserves the wave's entries, each contested lane takes its rank among them
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    if (hit) plist[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint4((uint32_t)px, (uint32_t)w & 0x00ffffffu, thr, 0u);
}

// ---- the remap alone with --perceptual-palettes (snesimage_remap_candidates_device) -----------------------------
// CIEDE2000 costs ~250 instructions, and only the contested pixels of the slot (a tenth of the image) need it: every
// candidate's map starts as B's (k_remap_fill4), then one wave per candidate walks the compact list and overwrites
// the pixels its colour wins with the slot's index.
__global__ __launch_bounds__(256) void k_remap_fill4(MapsParams P) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= P.npx) return;
    const uint4 a = reinterpret_cast<const uint4 *>(P.pack)[2 * (size_t)q], b = reinterpret_cast<const uint4 *>(P.pack)[2 * (size_t)q + 1];
    const uint32_t lo[4] = {a.x, a.z, b.x, b.z};
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t ci0 = lo[i] >> 24;
        word |= (ci0 == (uint32_t)P.ncol ? (uint32_t)P.si : (ci0 == (uint32_t)P.ncol + 1u ? 0u : ci0 % (uint32_t)P.sub_size)) << (8 * i);
    }
    const int c0 = blockIdx.y * kRemapCands;
    for (int cc = 0; cc < kRemapCands && c0 + cc < P.ncand; cc++) reinterpret_cast<uint32_t *>(P.maps + (size_t)(c0 + cc) * P.npx)[q] = word;
}
// One block (four waves) per candidate, as in k_sparse_scan_lab: a lane that cannot rule its pixel out with the two sure
// "no"s of color.hpp queues it, and the CIEDE2000 evaluation runs on full waves of queued pixels.
__global__ __launch_bounds__(256) void k_remap_won_lab(MapsParams P, const uint4 *__restrict__ plist, const int *__restrict__ plist_count) {
    __shared__ uint32_t s_queue[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cand = (int)blockIdx.x;
    if (cand >= P.ncand) return;
    Lab cl; cl.l = P.cand_lab[3 * cand]; cl.a = P.cand_lab[3 * cand + 1]; cl.b = P.cand_lab[3 * cand + 2];
    const float cch = sqrtf(cl.a * cl.a + cl.b * cl.b);
    uint8_t *map = P.maps + (size_t)cand * P.npx;
    const int n = *plist_count;
    auto full_test = [&](int i) {
        const uint4 e = plist[i];
        Lab t; t.l = P.labpx[3 * (size_t)e.x]; t.a = P.labpx[3 * (size_t)e.x + 1]; t.b = P.labpx[3 * (size_t)e.x + 2];
        const float d = ciede2000(cl, t), bd = __uint_as_float(e.z & 0x7fffffffu);
        if ((d < bd) || ((e.z & 0x80000000u) && d == bd)) map[e.x] = (uint8_t)P.si; // strict <, ties to the lower index (lib.rs:788-791)
    };
    uint32_t *q = s_queue[w];
    int queued = 0; // wave-uniform
    for (int i0 = 64 * w; i0 < n; i0 += 256) {
        const int i = i0 + lane;
        bool maybe = false;
        if (i < n) {
            const uint4 e = plist[i];
            if (e.z == 0xffffffffu) map[e.x] = (uint8_t)P.si;
            else {
                Lab t; t.l = P.labpx[3 * (size_t)e.x]; t.a = P.labpx[3 * (size_t)e.x + 1]; t.b = P.labpx[3 * (size_t)e.x + 2];
                const float bd = __uint_as_float(e.z & 0x7fffffffu);
                maybe = !ciede2000_cannot_beat(cl, t, bd) && !ciede2000_cannot_beat_ab(cl, cch, t, bd);
            }
        }
        const unsigned long long mm = __ballot(maybe);
        if (maybe) q[queued + __popcll(mm & ((1ull << lane) - 1ull))] = (uint32_t)i;
        queued += __popcll(mm);
        if (queued >= 64) { // (the queue holds at most 127 entries)
            full_test((int)q[lane]);
            queued -= 64;
            if (lane < queued) { const uint32_t v = q[64 + lane]; q[lane] = v; } // same-wave LDS traffic is ordered
        }
    }
    if (lane < queued) full_test((int)q[lane]);
}

// Longest first: the V pass of a candidate sweeps every column from its first changed group to the bottom, so blocks
// differ several-fold in length; handing them out in descending length keeps the tail of the launch short.
// Counting sort by the first changed group of scale 0 (H/4 + 1 bins), one block per launch.
__device__ __forceinline__ void sparse_order_body(const SparseParams &P, int *__restrict__ order) {
    __shared__ int s_cnt[66];
    const int t = threadIdx.x, H4 = P.G.sh[0] >> 2;
    const int *first = P.first + P.k0;
    if (t < 66) s_cnt[t] = 0;
    __syncthreads();
    for (int i = t; i < P.ncand; i += blockDim.x) atomicAdd(&s_cnt[first[i]], 1);
    __syncthreads();
    if (t == 0) { int run = 0; for (int b = 0; b <= H4; b++) { const int c = s_cnt[b]; s_cnt[b] = run; run += c; } }
    __syncthreads();
    for (int i = t; i < P.ncand; i += blockDim.x) order[atomicAdd(&s_cnt[first[i]], 1)] = i; // arbitrary inside a bin: results do not depend on it
}
__global__ __launch_bounds__(1024) void k_sparse_order(SparseParams P, int *__restrict__ order) { sparse_order_body(P, order); }

// ---- kernel entry points of the bodies above ----
__global__ __launch_bounds__(256) void k_sparse_scan_lab(SparseParams P) { sparse_scan_lab_body(P); }
__global__ __launch_bounds__(1024) void k_sparse_scan(SparseParams P) { sparse_scan_body<1>(P); }
__global__ __launch_bounds__(1024) void k_sparse_scan4(SparseParams P) { sparse_scan_body<4>(P); } // short lists: four waves per candidate
__global__ __launch_bounds__(1024) void k_dither_first(SparseParams P) { dither_first_body(P); }
__global__ __launch_bounds__(256) void k_dither_first_lab(SparseParams P) { dither_first_lab_body(P); }
__global__ __launch_bounds__(1024) void k_dither_diff(SparseParams P) { dither_diff_body(P); }
__global__ __launch_bounds__(256) void k_sparse_down(SparseParams P, int only_scale) { sparse_down_body(P, only_scale, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void k_sparse_down1(SparseParams P) { sparse_down1_body(P, (int)blockIdx.x, (int)gridDim.x); }
__global__ __launch_bounds__(256) void k_sparse_down_tiles(SparseParams P) { sparse_down_tiles_body<SNES_DOWN_TILES_U>(P, (int)blockIdx.x, (int)gridDim.x); }
__global__ __launch_bounds__(256) void k_base_down(SparseParams P) { base_down_body(P); }
__global__ __launch_bounds__(64) void k_sparse_h(SparseParams P) { sparse_h_body(P); }
__global__ __launch_bounds__(256, 1) void k_sparse_v_base(SparseParams P) { sparse_v_base_body(P); }
__global__ __launch_bounds__(256) void k_build_plist(const unsigned long long *__restrict__ pack, int npx, uint4 *__restrict__ plist, int *__restrict__ count) { build_plist_body(pack, npx, plist, count); }

} // namespace snes
