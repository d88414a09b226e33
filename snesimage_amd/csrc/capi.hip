// snesimage_amd/csrc/capi.hip — libsnesimage_hip.so: context, launch sequencing and the C ABI of
// include/snesimage_hip.h.  gfx950 only; there is no CPU path in this library.
#include "../../include/snesimage_hip.h"
#include "kernels.hpp"
#include "kernels_opt.hpp"
#include "kernels_fast.hpp"
#include "kernels_sparse.hpp"
#include "kernels_sparse2.hpp"
#include "kernels_batch.hpp"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <memory>
#include <vector>

using namespace snes;

namespace {

thread_local std::string g_err;
int32_t fail(int32_t code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return fail(SNES_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)
#define CHECK(expr)                                                                                               \
    do {                                                                                                          \
        int32_t rc_ = (expr);                                                                                     \
        if (rc_ != SNES_OK) return rc_;                                                                           \
    } while (0)

// ssimulacra2 build.rs: recursive-Gaussian constants for sigma = 1.5 (binary64, then f32)
BlurK make_blur_constants() {
    const double SIGMA = SSIM2_BLUR_SIGMA, PI = 3.14159265358979323846;
    const double radius = std::round(std::fma(SSIM2_BLUR_RADIUS_A, SIGMA, SSIM2_BLUR_RADIUS_B));
    const double w0 = PI / (2.0 * radius);
    const double omega[3] = {w0, 3.0 * w0, 5.0 * w0};
    const double p1 = 1.0 / std::tan(0.5 * omega[0]), p3 = -1.0 / std::tan(0.5 * omega[1]), p5 = 1.0 / std::tan(0.5 * omega[2]);
    const double r1 = p1 * p1 / std::sin(omega[0]), r3 = -p3 * p3 / std::sin(omega[1]), r5 = p5 * p5 / std::sin(omega[2]);
    const double nhs2 = -0.5 * SIGMA * SIGMA, rr = 1.0 / radius;
    double rho[3];
    for (int i = 0; i < 3; i++) rho[i] = std::exp(nhs2 * omega[i] * omega[i]) * rr;
    const double d13 = std::fma(p1, r3, -r1 * p3), d35 = std::fma(p3, r5, -r3 * p5), d51 = std::fma(p5, r1, -r5 * p1);
    const double rd13 = 1.0 / d13, z15 = d35 * rd13, z35 = d51 * rd13;
    const double A[3][3] = {{p1, p3, p5}, {r1, r3, r5}, {z15, z35, 1.0}};
    const double gam[3] = {1.0, std::fma(radius, radius, -SIGMA * SIGMA), std::fma(z15, rho[0], z35 * rho[1]) + rho[2]};
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    double inv[3][3];
    inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
    inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det; inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
    inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
    inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    BlurK K;
    for (int i = 0; i < 3; i++) {
        const double beta = inv[i][0] * gam[0] + inv[i][1] * gam[1] + inv[i][2] * gam[2];
        K.n2[i] = (float)(-beta * std::cos(omega[i] * (radius + 1.0)));
        K.d1[i] = (float)(-2.0 * std::cos(omega[i]));
    }
    return K;
}

// yuvxyb sRGB EOTF on v/255 (lib.rs:511-513) and palette's Srgb::into_linear on v/255 (lib.rs:1092-1097).
// powf is evaluated through binary64 pow and rounded once.
void make_eotf_tables(float *ssim_eotf, float *lab_eotf) {
    for (int v = 0; v < 256; v++) {
        const float x = (float)v / 255.0f;
        ssim_eotf[v] = x < SSIM2_SRGB_THRESHOLD ? x / SSIM2_SRGB_LINEAR_DIV : (float)std::pow((double)((x + SSIM2_SRGB_OFFSET) / SSIM2_SRGB_SCALE), (double)SSIM2_SRGB_GAMMA); // include/ssimulacra2_constants.h
        lab_eotf[v] = x <= PALETTE_SRGB_THRESHOLD ? (float)(1.0 / PALETTE_SRGB_LINEAR_DIV_D) * x
                                                  : (float)std::pow((double)std::fmaf(x, (float)(1.0 / PALETTE_SRGB_SCALE_D), (float)(PALETTE_SRGB_OFFSET_D / PALETTE_SRGB_SCALE_D)), (double)PALETTE_SRGB_GAMMA);
    }
}

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

// Every workspace allocation goes through dmalloc so that tests can make the n-th one fail (snesimage_debug_fail_alloc):
// the grow-on-demand allocators must leave the context usable after a failed grow.
std::atomic<int> g_fail_alloc_in{-1}; // -1: off; otherwise the number of allocations that still succeed (host threads initialising a batch of images allocate side by side)
template <typename T> hipError_t dmalloc(T **p, size_t bytes) {
    int left = g_fail_alloc_in.load(std::memory_order_relaxed);
    while (left >= 0) { // claim one of the remaining successes, or be the allocation that fails
        if (g_fail_alloc_in.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) { if (left == 0) { *p = nullptr; return hipErrorOutOfMemory; } break; }
    }
    return hipMalloc(p, bytes);
}

} // namespace

struct snesimage_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint32_t W = 0, H = 0, sub_count = 0, sub_size = 0, flags = 0;
    int ncol = 0;
    bool dither = false, perceptual = false, nes = false;
    // Floyd-Steinberg with a quad of lanes per row (k_dither4; RGB distance): a third of the step's latency for ~20 % more
    // instructions and half the runs per CU — B's own run always, the candidates' while they are few (SNES_DITHER4=0: never;
    // SNES_DITHER4_MAX: most runs per launch that still take it)
    bool dither4 = true; uint32_t dither4_max = 512;
    bool ditherw = true; // longer lists: one wave per resumed run (k_ditherw) instead of two waves and a ring in LDS (k_dither MODE 2)
    bool dither_rec = true; // resumed runs take B's search result where their dithered target equals B's (SNES_DITHER_REC=0: always search)
    Geom G{};
    BlurK K{};
    size_t npx = 0, src_floats = 0;
    uint32_t chunk = 4096, chunk_alloc = 0; // chunk: most candidates one launch group takes (a list is split evenly over the lanes up to that); chunk_alloc: dense workspace per lane

    std::vector<uint8_t> h_orig; float h_eotf[256], h_lab_eotf[256];

    // device state
    uint8_t *d_orig = nullptr, *d_tile_pal = nullptr, *d_colors = nullptr, *d_map = nullptr;
    unsigned long long *d_pack = nullptr, *d_packT = nullptr, *d_packC4 = nullptr, *d_packR4 = nullptr;
    float *d_img1C4 = nullptr, *d_mu1R4 = nullptr, *d_sd1R4 = nullptr, *d_a1 = nullptr, *d_a1R4 = nullptr; double *d_r1 = nullptr, *d_r1R4 = nullptr; // blocked copies for the fast kernels; r1: maps_accumulate
    int fast_mask = 0; // bit s: scale s has width and height multiples of 64
    float *d_eotf = nullptr, *d_lab_eotf = nullptr;
    uint32_t *d_pal_rgb8 = nullptr; float *d_pal_lin = nullptr, *d_pal_xyb = nullptr, *d_pal_lab = nullptr;
    float *d_lin0 = nullptr, *d_img1 = nullptr, *d_img1T = nullptr, *d_mu1 = nullptr, *d_sd1 = nullptr;
    float *d_labpx = nullptr, *d_labpxT = nullptr;
    // per-chunk workspace
    float *d_work = nullptr, *d_cand_tab = nullptr, *d_cand_lab = nullptr;
    double *d_part = nullptr;
    uint8_t *d_maps = nullptr, *d_mapsT = nullptr, *d_mapsC4 = nullptr, *d_mapsR4 = nullptr; // dither path: per-candidate maps
    uint8_t *d_subC4 = nullptr, *d_subR4 = nullptr;
    // dither path: every lane keeps the map of the best candidate it has scored in the current list, so that the commit
    // takes the winner's map instead of dithering the image again (lib.rs:237 re-runs optimize() on the winner's palette)
    uint8_t *d_bestmap = nullptr, *d_bestmaps_all = nullptr; BestRec *d_bestrec = nullptr, *d_bestrecs_all = nullptr; int *d_skip = nullptr;
    uint4 *d_rplist = nullptr; int *d_rcount = nullptr; // contested pixels for the perceptual remap-only entry point
    float *d_rtab = nullptr, *d_rlab = nullptr; uint32_t rtab_cap = 0; // the remap-only entry point's candidate tables (it needs none of the scoring workspace)
    bool map_pending = false; // without dither the optimize() that ends a step (lib.rs:237) is deferred until something reads palette_map
    bool best_valid = false, map_synced = false; // records belong to the list being committed; d_map is optimize() of the current palette
    // Additional launch lanes: chunk i of a candidate list runs on lane i % nlanes (lane 0 = the context's stream and the
    // workspace above), so the HBM-bound H pass of one chunk overlaps the VALU-bound V pass of another.
    struct Lane { hipStream_t stream = nullptr; float *d_work = nullptr, *d_cand_tab = nullptr, *d_cand_lab = nullptr; double *d_part = nullptr; uint8_t *d_maps = nullptr, *d_mapsT = nullptr, *d_mapsC4 = nullptr, *d_mapsR4 = nullptr, *d_bestmap = nullptr; BestRec *d_bestrec = nullptr; hipEvent_t done = nullptr; };
    std::vector<Lane> extra; uint32_t nlanes = 2; hipEvent_t ev_ready = nullptr;
    // Row-sparse scoring (kernels_sparse.hpp): one storage array for every lane's candidates plus the base image B
    struct Sparse {
        bool lpt = true; // V pass in descending sweep length (SNES_LPT=0: as listed)
        bool down1 = true; // scale 1 of the candidates' downscale in a kernel of its own, one block per changed group (SNES_DOWN1=0: inside k_sparse_down)
        bool counters_cleared = false; // k_prep cleared B's counters for the current pack
        uint32_t h2q_max = 512; // longest list that takes k_sparse_h2q (SNES_H2Q_MAX; 0 = never)
        bool down_tiles = true; // scales 2.. of the candidates' downscale one block per changed group of scale 3 (SNES_DOWN_TILES=0: k_sparse_down, one block per candidate, scale after scale)
        // The H pass of scale 0 — three quarters of the H pass, bound by its stores — reads the pack and B's checkpoints, not the
        // downscale; the downscale is bound by its arithmetic.  Launch groups of h0_min candidates and more run the two side by side:
        // scale 0's lists on a stream of their own from the scan on, the other wide scales behind the downscale as before.
        uint32_t h0_min = 1024; // (SNES_H0_MIN; 0 = never)
        bool base_down_side = false; // B's planes in place were downscaled on B's stream: their readers wait for ev_base_h (sparse_base_pass)
        uint32_t h0_grid = 8192; // most blocks per list of that launch (SNES_H0_GRID)
        hipStream_t h0_stream[8] = {}; hipEvent_t ev_scan[8] = {}, ev_h0[8] = {}, ev_hn[8] = {}, ev_vn[8] = {};
        bool v0_aside = true; // and scale 0's V pass behind its H pass on that stream, beside everything else of the group (SNES_V0_ASIDE=0: one V launch on the main stream)
        hipEvent_t ev_v0[8] = {};
        bool vn_aside = true; // with it, the narrow scales' V pass on that stream beside the wide scales' (SNES_VN_ASIDE=0: behind it)
        uint32_t tiles_grid = 4096; // most blocks of k_sparse_down_tiles (SNES_TILES_GRID)
        uint32_t down1_grid = 32768; // most blocks of k_sparse_down1 (SNES_DOWN1_GRID)
        bool vsplit = true; // B's wide V sweep with recurrences and maps on two waves (k_sparse_v2_base_split; SNES_VSPLIT=0: one wave does both)
        uint32_t scan4_max = 2048; // longest list whose scan deals a candidate's contested pixels to four waves (SNES_SCAN4_MAX; 0 = never)
        uint32_t hgrid = 8192; // most blocks per scale of k_sparse_h (grid-stride beyond)
        bool enabled = false, side = true; uint32_t min_n = 1; uint32_t cap = 0, lanes = 0; // min_n: shortest list that takes the group-sparse path (SNES_SPARSE_MIN; until round 4: 64 — a channel sweep's 32 candidates, or a rank's share of a 64-candidate call, went the dense way: 0.35 ms against 0.27, 1.7 ms with --perceptual-palettes) // cap = candidates per lane the arrays were sized for
        SparseGeom S{};
        float *store = nullptr, *cand_tab = nullptr, *cand_lab = nullptr, *ckf = nullptr, *ckh = nullptr; long long zeros_off = 0; uint32_t *bitmap = nullptr; double *cka = nullptr, *part = nullptr;
        CandMeta *meta = nullptr; unsigned int *items = nullptr; int *item_count = nullptr; long long item_stride = 0; int *order = nullptr, *first = nullptr;
        uint4 *plist = nullptr; int *plist_count = nullptr; bool plist_valid = false;
        hipStream_t base_stream = nullptr; hipEvent_t ev_base_in = nullptr, ev_base_h = nullptr, ev_base_done = nullptr, ev_base_narrow = nullptr; // B's H and V passes run beside the candidates' scan/down/H
        // --dither (RGB distance): B dithered once per slot (k_dither MODE 1) and the candidates resumed from its checkpoints (MODE 2)
        uint8_t *dmaps = nullptr, *dmapsC4 = nullptr; // [lane][cap][W*H] candidates' palette_maps, row-major and C4
        uint8_t *bmap = nullptr, *bmapC4 = nullptr, *bcand = nullptr; unsigned long long *dpack = nullptr; double *ckd = nullptr; uint32_t slot_ci = 0;
        float *rec_lab = nullptr; // --dither --perceptual-palettes: Lab of B's dithered targets (k_dither_first_lab)
        int base_sp = -1, base_si = -1; // slot B was built for (with --dither the pack does not depend on the slot, B does)
        // B's Floyd-Steinberg run a call ahead: while a call's candidates are scored, B of the scheduler's next slot (lib.rs:881-933
        // walks the slots in raster order) is dithered on B's stream into the second set of buffers, for the palette as it
        // stands.  The next call takes it if it is for that slot and the commit in between changed nothing (a flag on the
        // device: the host does not wait for the commit) — B's run then leaves at once — and dithers B itself otherwise.
        struct Ahead { uint8_t *bmap = nullptr, *bmapC4 = nullptr, *bcand = nullptr; unsigned long long *dpack = nullptr; double *ckd = nullptr; float *btab = nullptr, *blab = nullptr, *rec_lab = nullptr; int *ok = nullptr;
                       hipEvent_t ev = nullptr; bool on = true, have = false; int sp = -1, si = -1; unsigned long long epoch = 0; } ahead;
    } sp;
    // step state
    uint8_t *d_cand = nullptr; uint32_t cand_cap = 0;
    uint8_t *d_cand_sel = nullptr;
    double *d_errs = nullptr, *d_errs_sel = nullptr;
    double *d_inc_err = nullptr; // incumbent error
    StepResult *d_last = nullptr;
    double *d_scratch_err = nullptr;
    uint8_t *d_dummy_cand = nullptr;
    // k-means workspace
    KmeansWork km{};
    // snesimage_reassign_tiles: cost per (tile, subpalette), tiles with an opaque pixel, tiles moved (kept: hipMalloc / hipFree per call synchronise the device)
    double *d_tile_cost = nullptr; int *d_tile_any = nullptr; unsigned int *d_tile_moved = nullptr;

    struct snesimage_batch *owner = nullptr; // set while the context is lent to a batch (batch_host.inc)
    struct snesimage_group *group = nullptr; // set while the context is a member of a group (group_host.inc)
    hipEvent_t ev_own = nullptr;             // marks the end of the work this context queued on its own stream (for its batch)
    struct snesimage_window *win = nullptr;  // slot windows of snesimage_run_slots (window_host.inc), created on first use
    bool win_pend = false;                   // snesimage_slots_begin without its snesimage_slots_commit yet
    bool pack_borrowed = false;              // a slot context of a --dither window: pack and subpalette planes are the parent's

    // cache keys
    bool tables_valid = false, src_valid = false, inc_valid = false;
    unsigned long long epoch = 0; bool epoch_by_commit = false; // palette / tile-map generations, and whether the last one came from a commit (see sp.ahead)
    int pack_mode = -1, pack_sp = -1, pack_si = -1; bool pack_valid = false;
    // pending step (split phase)
    uint32_t pend_n = 0, pend_sp = 0, pend_si = 0, pend_method = 0; bool pend = false;

    // timing
    int timing = 0; // 0 off, 1 every bracket (group, H pass, V pass), 2 the V pass's only (snesimage_timing_enable)
    double t_ms[3] = {0.0, 0.0, 0.0}; uint64_t t_launches = 0, t_cands = 0; // [0] whole launch group, [1] k_hpass scale 0, [2] k_vpass scale 0
    struct TimingRec { hipEvent_t ev[6]; uint32_t n; bool light; }; // light: only ev[3], ev[4] (the V pass) were recorded
    std::vector<TimingRec> t_pending;
};

namespace {

int32_t alloc_lane(snesimage_ctx *c, uint32_t chunk, float *&d_work, float *&d_cand_tab, float *&d_cand_lab, double *&d_part, uint8_t *&d_maps, uint8_t *&d_mapsT, uint8_t *&d_mapsC4, uint8_t *&d_mapsR4) {
    dfree(d_work); dfree(d_cand_tab); dfree(d_cand_lab); dfree(d_part); dfree(d_maps); dfree(d_mapsT); dfree(d_mapsC4); dfree(d_mapsR4);
    HIPCHK(dmalloc(&d_work, sizeof(float) * (size_t)c->G.cand_stride * chunk));
    HIPCHK(dmalloc(&d_cand_tab, sizeof(float) * 8 * (size_t)chunk));
    HIPCHK(dmalloc(&d_cand_lab, sizeof(float) * 3 * (size_t)chunk));
    HIPCHK(dmalloc(&d_part, sizeof(double) * (size_t)chunk * kMaxScales * 18));
    if (c->dither) {
        HIPCHK(dmalloc(&d_maps, c->npx * (size_t)chunk));
        HIPCHK(dmalloc(&d_mapsT, c->npx * (size_t)chunk));
        HIPCHK(dmalloc(&d_mapsC4, c->npx * (size_t)chunk));
        HIPCHK(dmalloc(&d_mapsR4, c->npx * (size_t)chunk));
    }
    return SNES_OK;
}
// grow-only: `chunk` = candidates per lane the dense kernels are about to handle, `lanes` = launch lanes the list at hand is dealt to.
// A lane's stream is created by the first list that needs it: a stream that exists and never runs still takes its turn when the
// runtime deals streams to hardware queues, and with the second lane's idle stream in the process the three streams of an RGB launch
// group shared queues or not depending on GPU_MAX_HW_QUEUES and on whether RCCL had brought streams of its own — 1.41 against 1.67-1.95 ms
// per 4,096-candidate call (profiles/r4_hw_queues_lanes.txt).
int32_t alloc_workspace(snesimage_ctx *c, uint32_t chunk, uint32_t lanes = 1) {
    // (--dither and --perceptual-palettes split every list over the lanes: theirs exist from the first call on, as they always did —
    // created later, behind the caller's stream, the same three active streams drew worse queues: 5.2 -> 6.4 ms and 2.26 -> 2.45 ms per
    // 4,096-candidate call, profiles/r4_hw_queues_dither.txt)
    if (c->dither || c->perceptual) lanes = c->nlanes;
    if (lanes > c->nlanes) lanes = c->nlanes;
    if (lanes < c->extra.size() + 1) lanes = (uint32_t)c->extra.size() + 1;
    if (c->chunk_alloc >= chunk && c->extra.size() + 1 >= lanes) return SNES_OK;
    if (chunk < c->chunk_alloc) chunk = c->chunk_alloc;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &L : c->extra) HIPCHK(hipStreamSynchronize(L.stream));
    c->chunk_alloc = 0; // the buffers are about to be released: a failed grow must not leave the old capacity behind
    CHECK(alloc_lane(c, chunk, c->d_work, c->d_cand_tab, c->d_cand_lab, c->d_part, c->d_maps, c->d_mapsT, c->d_mapsC4, c->d_mapsR4));
    if (!c->ev_ready) HIPCHK(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    while (c->extra.size() + 1 < lanes) {
        snesimage_ctx::Lane L;
        HIPCHK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        c->extra.push_back(L);
    }
    for (auto &L : c->extra) CHECK(alloc_lane(c, chunk, L.d_work, L.d_cand_tab, L.d_cand_lab, L.d_part, L.d_maps, L.d_mapsT, L.d_mapsC4, L.d_mapsR4));
    if (c->dither && !c->d_bestmaps_all) {
        HIPCHK(dmalloc(&c->d_bestmaps_all, c->npx * (size_t)c->nlanes));
        HIPCHK(dmalloc(&c->d_bestrecs_all, sizeof(BestRec) * c->nlanes));
        HIPCHK(dmalloc(&c->d_skip, sizeof(int)));
    }
    if (c->d_bestmaps_all) { // (every lane's slot exists from the start; a lane created later finds its own)
        c->d_bestmap = c->d_bestmaps_all; c->d_bestrec = c->d_bestrecs_all;
        for (size_t l = 0; l < c->extra.size(); l++) { c->extra[l].d_bestmap = c->d_bestmaps_all + (l + 1) * c->npx; c->extra[l].d_bestrec = c->d_bestrecs_all + (l + 1); }
    }
    c->chunk_alloc = chunk;
    return SNES_OK;
}

int32_t ensure_cand_capacity(snesimage_ctx *c, uint32_t n) {
    if (c->cand_cap >= n) return SNES_OK;
    if (c->stream) HIPCHK(hipStreamSynchronize(c->stream)); // the previous list may still be in flight
    c->cand_cap = 0;
    dfree(c->d_cand); dfree(c->d_cand_sel); dfree(c->d_errs); dfree(c->d_errs_sel);
    uint32_t cap = n < 64 ? 64 : n;
    HIPCHK(dmalloc(&c->d_cand, 3 * (size_t)cap));
    HIPCHK(dmalloc(&c->d_cand_sel, 3 * (size_t)cap));
    HIPCHK(dmalloc(&c->d_errs, sizeof(double) * cap));
    HIPCHK(dmalloc(&c->d_errs_sel, sizeof(double) * cap));
    c->cand_cap = cap;
    return SNES_OK;
}

int32_t ensure_tables(snesimage_ctx *c) {
    if (c->tables_valid) return SNES_OK;
    int n = c->ncol + 2;
    hipLaunchKernelGGL(k_palette_tables, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->d_colors, c->ncol, c->d_eotf, c->d_pal_rgb8, c->d_pal_lin, c->d_pal_xyb);
    if (c->perceptual)
        hipLaunchKernelGGL(k_palette_lab, dim3((c->ncol + 63) / 64), dim3(64), 0, c->stream, c->d_pal_rgb8, c->ncol, c->d_lab_eotf, c->d_pal_lab);
    HIPCHK(hipGetLastError());
    c->tables_valid = true;
    return SNES_OK;
}

// Source-side pyramid: img1, img1T, mu1, s11 at every scale (depends only on `original`).
int32_t ensure_source(snesimage_ctx *c) {
    if (c->src_valid) return SNES_OK;
    CHECK(alloc_workspace(c, 1));
    const Geom &G = c->G;
    hipLaunchKernelGGL(k_source_scale0, dim3((unsigned)((c->npx + 255) / 256)), dim3(256), 0, c->stream, c->d_orig, c->d_eotf, (int)c->W, (int)c->H, c->d_lin0,
                       c->d_img1, c->d_img1T);
    if (G.nscales > 1) {
        DownParams D{}; D.G = G; D.lin0 = c->d_lin0; D.work = c->d_img1; D.workT = c->d_img1T;
        hipLaunchKernelGGL(k_downscale_chain<false>, dim3((G.W / 32) * ((G.H + 31) / 32), 1), dim3(256), 0, c->stream, D);
    }
    for (int s = 0; s < G.nscales; s++) {
        HParams Hp{}; Hp.G = G; Hp.K = c->K; Hp.s = s; Hp.npairs = 3; Hp.ncol = c->ncol;
        Hp.in1T = c->d_img1T + G.src_off[s]; Hp.in2T = c->d_img1T + G.src_off[s]; Hp.work = c->d_work;
        int ppw = 256 / G.sh[s];
        hipLaunchKernelGGL((k_hpass<false, false>), dim3((3 + ppw - 1) / ppw), dim3(256), 0, c->stream, Hp);
        VParams Vp{}; Vp.G = G; Vp.K = c->K; Vp.s = s; Vp.npairs = 3; Vp.ncol = c->ncol;
        Vp.img1 = c->d_img1 + G.src_off[s]; Vp.mu1_out = c->d_mu1 + G.src_off[s]; Vp.sd1_out = c->d_sd1 + G.src_off[s]; Vp.a1_out = c->d_a1 + G.src_off[s]; Vp.r1_out = c->d_r1 + G.src_off[s]; Vp.work = c->d_work;
        int ppv = 256 / G.sw[s];
        hipLaunchKernelGGL((k_vpass<false, true, false>), dim3((3 + ppv - 1) / ppv), dim3(256), 0, c->stream, Vp);
    }
    for (int s = 0; s < G.nscales; s++)
        if ((c->fast_mask & (1 << s)) || c->sp.enabled) { // the row-sparse path reads the blocked layouts at every scale
            const int N = G.sw[s] * G.sh[s];
            hipLaunchKernelGGL(k_relayout, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->d_img1 + G.src_off[s], G.sw[s], G.sh[s], (float *)nullptr, c->d_img1C4 + G.src_off[s]);
            hipLaunchKernelGGL(k_relayout, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->d_mu1 + G.src_off[s], G.sw[s], G.sh[s], c->d_mu1R4 + G.src_off[s], (float *)nullptr);
            hipLaunchKernelGGL(k_relayout, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->d_sd1 + G.src_off[s], G.sw[s], G.sh[s], c->d_sd1R4 + G.src_off[s], (float *)nullptr);
            hipLaunchKernelGGL(k_relayout, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->d_a1 + G.src_off[s], G.sw[s], G.sh[s], c->d_a1R4 + G.src_off[s], (float *)nullptr);
            hipLaunchKernelGGL(k_relayout_f64, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->d_r1 + G.src_off[s], G.sw[s], G.sh[s], c->d_r1R4 + G.src_off[s]);
        }
    if (c->perceptual)
        hipLaunchKernelGGL(k_pixel_lab, dim3((unsigned)((c->npx + 255) / 256)), dim3(256), 0, c->stream, c->d_orig, c->d_lab_eotf, (int)c->W, (int)c->H, c->d_labpx, c->d_labpxT);
    HIPCHK(hipGetLastError());
    c->src_valid = true;
    return SNES_OK;
}

int32_t run_prep(snesimage_ctx *c, int mode, int sp, int si) {
    if (c->pack_valid && c->pack_mode == mode && (mode != 2 || (c->pack_sp == sp && c->pack_si == si))) return SNES_OK;
    CHECK(ensure_tables(c));
    if (c->perceptual) CHECK(ensure_source(c));
    PrepParams P{};
    P.orig = c->d_orig; P.tile_pal = c->d_tile_pal; P.pal_rgb8 = c->d_pal_rgb8; P.map = c->d_map; P.pack = c->d_pack; P.packT = c->d_packT; P.packC4 = c->d_packC4; P.packR4 = c->d_packR4; P.subC4 = c->d_subC4; P.subR4 = c->d_subR4;
    P.labpx = c->d_labpx; P.pal_lab = c->d_pal_lab;
    P.W = (int)c->W; P.H = (int)c->H; P.sub_size = (int)c->sub_size; P.ncol = c->ncol; P.mode = mode; P.sp = sp; P.si = si; P.perceptual = c->perceptual ? 1 : 0;
    c->sp.counters_cleared = false;
    if (mode == 2 && c->sp.plist_count) { // the contested-pixel count of the slot
        P.zero = c->sp.plist_count; P.nzero = 1; c->sp.counters_cleared = true; // (B's own item counters are never cleared: sparse_alloc)
    }
    hipLaunchKernelGGL(k_prep, dim3((unsigned)((c->npx + 255) / 256)), dim3(256), 0, c->stream, P);
    HIPCHK(hipGetLastError());
    c->pack_valid = true; c->pack_mode = mode; c->pack_sp = sp; c->pack_si = si;
    c->sp.plist_valid = false; // the base image of the row-sparse path belongs to this pack
    return SNES_OK;
}

// Without dither the pack carries, per pixel, the best non-slot entry and the key the candidate must beat
// (mode 2).  With dither every candidate gets its own map from k_dither; the pack is then only consulted for
// the transparent-pixel marker, which any mode provides (mode 1 is the cheapest).
int32_t prep_for_slot(snesimage_ctx *c, int sp, int si) { return c->dither ? run_prep(c, 1, -1, -1) : run_prep(c, 2, sp, si); }

// k_dither instantiations: the 15-colour subpalettes of the SNES 4bpp modes get a fully unrolled entry search
void launch_dither(snesimage_ctx *c, const DitherParams &Dp, uint32_t nblocks) {
    if (c->perceptual && c->dither4 && nblocks <= c->dither4_max) hipLaunchKernelGGL((k_dither4_lab<0>), dim3(nblocks), dim3(512), 0, c->stream, Dp); // (optimize() of the committed palette: one run)
    else if (c->perceptual) hipLaunchKernelGGL((k_dither<true, 0>), dim3(nblocks), dim3(128), 0, c->stream, Dp);
    else if (c->dither4 && nblocks <= c->dither4_max && c->sub_size == 15) hipLaunchKernelGGL((k_dither4<15, 0>), dim3(nblocks), dim3(512), 0, c->stream, Dp); // a quad of lanes per row: few runs
    else if (c->dither4 && nblocks <= c->dither4_max) hipLaunchKernelGGL((k_dither4<0, 0>), dim3(nblocks), dim3(512), 0, c->stream, Dp);
    else if (c->sub_size == 15) hipLaunchKernelGGL((k_dither<false, 15>), dim3(nblocks), dim3(128), 0, c->stream, Dp);
    else hipLaunchKernelGGL((k_dither<false, 0>), dim3(nblocks), dim3(128), 0, c->stream, Dp);
}

// Score nc candidates (device rgb5 list) given a prepared pack; errors -> d_errors[err_offset + k*err_stride].
// slot_ci: colour index of the slot being replaced (dither path), or -1.
int32_t score_chunk(snesimage_ctx *c, const uint8_t *d_rgb5, uint32_t nc, double *d_errors, int err_stride, int err_offset, int sp, int si, uint8_t *d_maps_out) {
    const Geom &G = c->G;
    const int npairs = (int)nc * 3;
    const bool use_maps = c->dither;
    const uint32_t slot_ci = (sp >= 0) ? (uint32_t)(sp * (int)c->sub_size + si) : 0xffffffffu;
    snesimage_ctx::TimingRec tr{}; tr.n = nc;
    if (c->timing) { for (int i = 0; i < 6; i++) HIPCHK(hipEventCreate(&tr.ev[i])); if (c->timing == 1) HIPCHK(hipEventRecord(tr.ev[0], c->stream)); }
    hipLaunchKernelGGL(k_candidate_tables, dim3((nc + 63) / 64), dim3(64), 0, c->stream, d_rgb5, (int)nc, c->d_eotf, c->d_cand_tab);
    hipLaunchKernelGGL(k_candidate_slot, dim3((nc + 63) / 64), dim3(64), 0, c->stream, c->d_cand_tab, (int)nc, slot_ci);
    if (c->perceptual) hipLaunchKernelGGL(k_candidate_lab, dim3((nc + 63) / 64), dim3(64), 0, c->stream, c->d_cand_tab, (int)nc, c->d_lab_eotf, c->d_cand_lab);
    if (use_maps) {
        DitherParams Dp{};
        Dp.orig = c->d_orig; Dp.tile_pal = c->d_tile_pal; Dp.pal_rgb8 = c->d_pal_rgb8; Dp.pal_lab = c->d_pal_lab; Dp.cand_tab = c->d_cand_tab; Dp.cand_lab = c->d_cand_lab;
        Dp.lab_eotf = c->d_lab_eotf; Dp.maps = c->d_maps; Dp.mapsC4 = c->d_mapsC4;
        Dp.W = (int)c->W; Dp.H = (int)c->H; Dp.sub_size = (int)c->sub_size; Dp.ncol = c->ncol; Dp.slot_ci = slot_ci; Dp.perceptual = c->perceptual ? 1 : 0;
        launch_dither(c, Dp, nc);
        // the transposed copy only feeds the generic scale-0 H pass
        hipLaunchKernelGGL(k_maps_relayout, dim3((unsigned)(c->H / 4), nc), dim3(256), 0, c->stream, c->d_maps, (int)c->W, (int)c->H,
                           reinterpret_cast<uint32_t *>(c->d_mapsR4), (c->fast_mask & 1) ? (uint32_t *)nullptr : reinterpret_cast<uint32_t *>(c->d_mapsT));
        if (d_maps_out) HIPCHK(hipMemcpyAsync(d_maps_out, c->d_maps, c->npx * (size_t)nc, hipMemcpyDeviceToDevice, c->stream));
    } else if (d_maps_out) {
        MapsParams M{}; M.pack = c->d_pack; M.cand_tab = c->d_cand_tab; M.cand_lab = c->d_cand_lab; M.labpx = c->d_labpx; M.maps = d_maps_out;
        M.npx = (int)c->npx; M.ncol = c->ncol; M.sub_size = (int)c->sub_size; M.si = si < 0 ? 0 : si; M.ncand = (int)nc; M.perceptual = c->perceptual ? 1 : 0;
        hipLaunchKernelGGL(k_candidate_maps, dim3((unsigned)((c->npx + 255) / 256), nc), dim3(256), 0, c->stream, M);
    }
    if (G.nscales > 1) {
        DownParams D{}; D.G = G; D.pack = c->d_pack; D.pal_lin = c->d_pal_lin; D.cand_tab = c->d_cand_tab; D.cand_lab = c->d_cand_lab; D.labpx = c->d_labpx;
        D.work = c->d_work; D.ncol = c->ncol; D.perceptual = c->perceptual ? 1 : 0; D.use_maps = use_maps ? 1 : 0; D.fast_mask = c->fast_mask & ~1; D.maps = c->d_maps; D.tile_pal = c->d_tile_pal; D.sub_size = (int)c->sub_size;
        hipLaunchKernelGGL(k_downscale_chain<true>, dim3((G.W / 32) * ((G.H + 31) / 32), nc), dim3(256), 0, c->stream, D);
    }
    const bool fast0 = (c->fast_mask & 1) && (use_maps || !c->perceptual); // scale 0 takes its pixels from the pack (RGB keys) or from the per-candidate maps (dither)
    auto is_fast = [&](int s) { return s == 0 ? fast0 : ((c->fast_mask >> s) & 1) != 0; };
    auto fast_params = [&](int s) {
        FastParams F{}; F.G = G; F.K = c->K; F.s = s; F.npairs = npairs; F.ncol = c->ncol;
        F.packC4 = c->d_packC4; F.packR4 = c->d_packR4; F.pal_xyb = c->d_pal_xyb; F.cand_tab = c->d_cand_tab;
        F.use_maps = use_maps ? 1 : 0; F.mapsC4 = reinterpret_cast<const uint32_t *>(c->d_mapsC4); F.mapsR4 = reinterpret_cast<const uint32_t *>(c->d_mapsR4);
        F.subC4 = reinterpret_cast<const uint32_t *>(c->d_subC4); F.subR4 = reinterpret_cast<const uint32_t *>(c->d_subR4);
        F.img1C4 = c->d_img1C4 + G.src_off[s]; F.mu1R4 = c->d_mu1R4 + G.src_off[s]; F.sd1R4 = c->d_sd1R4 + G.src_off[s]; F.a1R4 = c->d_a1R4 + G.src_off[s]; F.r1R4 = c->d_r1R4 + G.src_off[s];
        F.work = c->d_work; F.part = c->d_part;
        return F;
    };
    for (int s = 0; s < G.nscales; s++) {
        if (s == 0 && c->timing == 1) HIPCHK(hipEventRecord(tr.ev[1], c->stream));
        if (is_fast(s)) {
            FastParams F = fast_params(s);
            dim3 grid((unsigned)(npairs * (G.sh[s] / 64)));
            if (s == 0) hipLaunchKernelGGL((k_hpass_fast<true>), grid, dim3(64), 0, c->stream, F);
            else hipLaunchKernelGGL((k_hpass_fast<false>), grid, dim3(64), 0, c->stream, F);
        } else {
            HParams Hp{}; Hp.G = G; Hp.K = c->K; Hp.s = s; Hp.npairs = npairs; Hp.ncol = c->ncol; Hp.perceptual = c->perceptual ? 1 : 0; Hp.use_maps = use_maps ? 1 : 0; Hp.sub_size = (int)c->sub_size;
            Hp.packT = c->d_packT; Hp.pal_xyb = c->d_pal_xyb; Hp.cand_tab = c->d_cand_tab; Hp.cand_lab = c->d_cand_lab; Hp.labpxT = c->d_labpxT;
            Hp.in1T = c->d_img1T + G.src_off[s]; Hp.in2T = nullptr; Hp.work = c->d_work; Hp.mapsT = c->d_mapsT; Hp.tile_pal = c->d_tile_pal;
            int ppw = 256 / G.sh[s];
            dim3 grid((npairs + ppw - 1) / ppw);
            if (s == 0) {
                if (c->perceptual && !use_maps) hipLaunchKernelGGL((k_hpass<true, true>), grid, dim3(256), 0, c->stream, Hp);
                else hipLaunchKernelGGL((k_hpass<true, false>), grid, dim3(256), 0, c->stream, Hp);
            } else hipLaunchKernelGGL((k_hpass<false, false>), grid, dim3(256), 0, c->stream, Hp);
        }
        if (s == 0 && c->timing == 1) HIPCHK(hipEventRecord(tr.ev[2], c->stream));
        // the V pass of the same scale follows immediately, while its H output is still cache-resident
        if (s == 0 && c->timing) HIPCHK(hipEventRecord(tr.ev[3], c->stream));
        int ppv = 256 / G.sw[s];
        dim3 grid((npairs + ppv - 1) / ppv);
        if (is_fast(s)) {
            FastParams F = fast_params(s);
            if (s == 0) hipLaunchKernelGGL((k_vpass_fast<true>), grid, dim3(256), 0, c->stream, F);
            else hipLaunchKernelGGL((k_vpass_fast<false>), grid, dim3(256), 0, c->stream, F);
        } else {
            VParams Vp{}; Vp.G = G; Vp.K = c->K; Vp.s = s; Vp.npairs = npairs; Vp.ncol = c->ncol; Vp.perceptual = c->perceptual ? 1 : 0; Vp.use_maps = use_maps ? 1 : 0; Vp.sub_size = (int)c->sub_size;
            Vp.pack = c->d_pack; Vp.pal_xyb = c->d_pal_xyb; Vp.cand_tab = c->d_cand_tab; Vp.cand_lab = c->d_cand_lab; Vp.labpx = c->d_labpx;
            Vp.mu1 = c->d_mu1 + G.src_off[s]; Vp.sd1 = c->d_sd1 + G.src_off[s]; Vp.a1 = c->d_a1 + G.src_off[s]; Vp.r1 = c->d_r1 + G.src_off[s]; Vp.work = c->d_work; Vp.part = c->d_part;
            Vp.maps = c->d_maps; Vp.tile_pal = c->d_tile_pal;
            if (s == 0) {
                if (c->perceptual && !use_maps) hipLaunchKernelGGL((k_vpass<true, false, true>), grid, dim3(256), 0, c->stream, Vp);
                else hipLaunchKernelGGL((k_vpass<true, false, false>), grid, dim3(256), 0, c->stream, Vp);
            } else hipLaunchKernelGGL((k_vpass<false, false, false>), grid, dim3(256), 0, c->stream, Vp);
        }
        if (s == 0 && c->timing) HIPCHK(hipEventRecord(tr.ev[4], c->stream));
    }
    hipLaunchKernelGGL(k_final_score, dim3((nc + 63) / 64), dim3(64), 0, c->stream, c->d_part, (int)nc, G, d_errors, err_stride, err_offset);
    if (use_maps) hipLaunchKernelGGL(k_keep_best, dim3(1), dim3(1024), 0, c->stream, d_errors, err_stride, err_offset, (int)nc, c->d_maps, (int)c->npx, c->d_bestrec, c->d_bestmap);
    HIPCHK(hipGetLastError());
    if (c->timing) { tr.light = c->timing != 1; if (!tr.light) HIPCHK(hipEventRecord(tr.ev[5], c->stream)); c->t_pending.push_back(tr); }
    return SNES_OK;
}

// ---- row-sparse path ------------------------------------------------------------------------------------
// grow-only: `need` = candidates per lane of the launch groups to come
SparseParams sparse_params(snesimage_ctx *c, uint32_t lane);

// `lanes` = launch lanes that get candidate storage: a list that fits one launch group runs on lane 0 alone (score_list), so the
// other lanes' planes — 4.45 MB per candidate each — are allocated by the first list that is dealt to them, not before
int32_t sparse_alloc(snesimage_ctx *c, uint32_t need, uint32_t lanes = 1) {
    auto &sp = c->sp;
    if (lanes > c->nlanes) lanes = c->nlanes;
    if (sp.cap >= need && sp.lanes >= lanes) return SNES_OK;
    if (need < sp.cap) need = sp.cap;
    if (lanes < sp.lanes) lanes = sp.lanes;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &L : c->extra) HIPCHK(hipStreamSynchronize(L.stream));
    if (sp.base_stream) HIPCHK(hipStreamSynchronize(sp.base_stream));
    sp.cap = 0; sp.lanes = 0; sp.plist_valid = false; // released below: a failed grow must not leave the old capacity behind
    dfree(sp.store); dfree(sp.cand_tab); dfree(sp.ckf); dfree(sp.cka); dfree(sp.part); dfree(sp.meta); dfree(sp.items); dfree(sp.item_count); dfree(sp.plist); sp.plist_count = nullptr;
    const Geom &G = c->G;
    SparseGeom &S = sp.S;
    long long off = 0, okf = 0, oka = 0, okh = 0; int go = 0;
    for (int s = 0; s < G.nscales; s++) {
        const long long N = (long long)G.sw[s] * G.sh[s];
        S.off_lin[s] = off; if (s >= 1) off += 3 * N;
        S.off_xybC[s] = off; if (s >= 1) off += 3 * N;
        S.off_xybR[s] = off; off += 3 * N; // scale 0 as well: written by the H pass (kernels_sparse2.hpp), read by the V pass
        S.off_hout[s] = off; off += 9 * N;
        S.off_ckf[s] = okf; okf += 3LL * (G.sh[s] / 4 + 2) * 18 * G.sw[s];
        S.off_cka[s] = oka; oka += 3LL * (G.sh[s] / 4 + 2) * 6 * G.sw[s];
        S.off_ckh[s] = okh; if (G.sw[s] >= 64) okh += 3LL * 3 * 18 * G.sh[s];
        S.goff[s] = go; go += G.sh[s] / 4;
    }
    S.cand_stride = off;
    const size_t ncap = (size_t)lanes * need + 1; // + the base image B
    sp.item_stride = (long long)need * (G.sh[0] / 4) * 3;
    HIPCHK(dmalloc(&sp.store, sizeof(float) * (size_t)S.cand_stride * ncap));
    HIPCHK(dmalloc(&sp.cand_tab, sizeof(float) * 8 * ncap));
    if (c->perceptual) {
        dfree(sp.cand_lab); dfree(sp.bitmap);
        HIPCHK(dmalloc(&sp.cand_lab, sizeof(float) * 3 * ncap));
        HIPCHK(dmalloc(&sp.bitmap, sizeof(uint32_t) * (c->npx / 32) * ncap));
    }
    HIPCHK(dmalloc(&sp.ckf, sizeof(float) * (size_t)okf));
    // B's H-pass checkpoints, then 12 W floats of zeros (what the V pass prefetches for the group below the image)
    dfree(sp.ckh); HIPCHK(dmalloc(&sp.ckh, sizeof(float) * (size_t)(okh + 12LL * G.W + 256))); // (+ 256 floats of scratch: SparseParams::trash)
    sp.zeros_off = okh;
    HIPCHK(hipMemsetAsync(sp.ckh + okh, 0, sizeof(float) * 12 * (size_t)G.W, c->stream));
    HIPCHK(dmalloc(&sp.cka, sizeof(double) * (size_t)oka));
    HIPCHK(dmalloc(&sp.part, sizeof(double) * ncap * kMaxScales * 18));
    HIPCHK(dmalloc(&sp.meta, sizeof(CandMeta) * ncap));
    dfree(sp.order); HIPCHK(dmalloc(&sp.order, sizeof(int) * ncap));
    dfree(sp.first); HIPCHK(dmalloc(&sp.first, sizeof(int) * ncap));
    HIPCHK(dmalloc(&sp.items, sizeof(unsigned int) * (size_t)sp.item_stride * kItemLists * (c->nlanes + 1)));
    if (!sp.base_stream && sp.side) {
        int prio_lo = 0, prio_hi = 0; // B's sweeps are the critical path of a step: give them the highest stream priority
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIPCHK(hipStreamCreateWithPriority(&sp.base_stream, hipStreamNonBlocking, prio_hi));
        HIPCHK(hipEventCreateWithFlags(&sp.ev_base_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sp.ev_base_h, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sp.ev_base_narrow, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sp.ev_base_done, hipEventDisableTiming));
    }
    HIPCHK(dmalloc(&sp.item_count, sizeof(int) * (kItemLists * (c->nlanes + 1) + 1))); // + the contested-pixel count, right behind B's counters
    HIPCHK(hipMemsetAsync(sp.item_count, 0, sizeof(int) * (kItemLists * (c->nlanes + 1) + 1), c->stream)); // every launch group leaves its counters cleared
    sp.plist_count = sp.item_count + (size_t)kItemLists * (c->nlanes + 1);
    sp.counters_cleared = true;
    HIPCHK(dmalloc(&sp.plist, sizeof(uint4) * c->npx));
    if (c->dither) {
        dfree(sp.dmaps); dfree(sp.dmapsC4); dfree(sp.bmap); dfree(sp.bmapC4); dfree(sp.bcand); dfree(sp.dpack); dfree(sp.ckd);
        HIPCHK(dmalloc(&sp.dmaps, c->npx * (size_t)need * lanes));
        HIPCHK(dmalloc(&sp.dmapsC4, c->npx * (size_t)need * lanes));
        HIPCHK(dmalloc(&sp.bmap, c->npx)); HIPCHK(dmalloc(&sp.bmapC4, c->npx)); HIPCHK(dmalloc(&sp.bcand, 64));
        HIPCHK(dmalloc(&sp.dpack, sizeof(unsigned long long) * c->npx));
        HIPCHK(dmalloc(&sp.ckd, sizeof(double) * 3 * c->W * (c->H / 4 + 1)));
        dfree(sp.rec_lab); if (c->perceptual) HIPCHK(dmalloc(&sp.rec_lab, sizeof(float) * 3 * c->npx));
        auto &ah = sp.ahead;
        dfree(ah.bmap); dfree(ah.bmapC4); dfree(ah.bcand); dfree(ah.dpack); dfree(ah.ckd); dfree(ah.btab); dfree(ah.blab); dfree(ah.rec_lab); dfree(ah.ok); ah.have = false;
        if (ah.on && sp.side && c->sub_size > 1) {
            HIPCHK(dmalloc(&ah.bmap, c->npx)); HIPCHK(dmalloc(&ah.bmapC4, c->npx)); HIPCHK(dmalloc(&ah.bcand, 64)); HIPCHK(dmalloc(&ah.btab, sizeof(float) * 8)); HIPCHK(dmalloc(&ah.ok, sizeof(int)));
            HIPCHK(dmalloc(&ah.dpack, sizeof(unsigned long long) * c->npx));
            HIPCHK(dmalloc(&ah.ckd, sizeof(double) * 3 * c->W * (c->H / 4 + 1)));
            if (c->perceptual) { HIPCHK(dmalloc(&ah.blab, sizeof(float) * 3)); HIPCHK(dmalloc(&ah.rec_lab, sizeof(float) * 3 * c->npx)); }
            if (!ah.ev) HIPCHK(hipEventCreateWithFlags(&ah.ev, hipEventDisableTiming));
        }
    }
    sp.cap = need; sp.lanes = lanes; sp.plist_valid = false;
    { // B's work items — every group of every scale, from column 0 — and its group tables depend on the geometry alone: published
      // once, here, and left alone (until late in round 4 every call cleared and republished them: a 1,024-thread block of its own,
      // 11-14 us at the head of B's chain, which is what a short call waits for)
        SparseParams P = sparse_params(c, c->nlanes);
        P.is_base = 1; P.ncand = 1; P.k0 = P.base;
        hipLaunchKernelGGL(k_sparse_scan, dim3(1), dim3(1024), 0, c->stream, P);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream)); // the clears above precede whatever the lanes and B's stream launch next
    return SNES_OK;
}

SparseParams sparse_params(snesimage_ctx *c, uint32_t lane) {
    auto &sp = c->sp;
    SparseParams P{};
    P.G = c->G; P.S = sp.S; P.K = c->K; P.ncol = c->ncol; P.base = (int)(sp.lanes * sp.cap);
    P.pack = c->d_pack; P.packC4 = c->d_packC4; P.packR4 = c->d_packR4; P.plist = sp.plist; P.plist_count = sp.plist_count;
    P.pal_lin = c->d_pal_lin; P.pal_xyb = c->d_pal_xyb; P.cand_tab = sp.cand_tab;
    P.perceptual = c->perceptual ? 1 : 0; P.labpx = c->d_labpx; P.cand_lab = sp.cand_lab; P.bitmap = sp.bitmap;
    P.img1C4 = c->d_img1C4; P.mu1R4 = c->d_mu1R4; P.sd1R4 = c->d_sd1R4; P.a1R4 = c->d_a1R4; P.r1R4 = c->d_r1R4;
    P.store = sp.store; P.meta = sp.meta;
    P.items = sp.items + (size_t)lane * sp.item_stride * kItemLists; P.item_count = sp.item_count + (size_t)lane * kItemLists; P.item_stride = sp.item_stride; // lane == nlanes: B
    P.ckf = sp.ckf; P.cka = sp.cka; P.part = sp.part; P.first = sp.first; P.ckh = sp.ckh; P.zeros = sp.ckh + sp.zeros_off; P.trash = sp.ckh + sp.zeros_off + 12LL * c->G.W;
    for (P.s_first = 0; P.s_first < c->G.nscales && c->G.sw[P.s_first] >= 64; P.s_first++) {} // first narrow scale
    if (c->dither) {
        const uint32_t l = lane < c->nlanes ? lane : 0;
        P.use_maps = 1; P.sub_size = (int)c->sub_size; P.slot_ci = sp.slot_ci; P.subC4 = c->d_subC4; P.tile_pal = c->d_tile_pal;
        P.perceptual = 0; P.labpx = sp.rec_lab; // (the scorer reads the runs' maps whatever distance made them; k_dither_first_lab: the targets' Lab)
        P.maps = sp.dmaps + (size_t)l * sp.cap * c->npx; P.mapsC4 = sp.dmapsC4 + (size_t)l * sp.cap * c->npx; P.bmap = sp.bmap; P.bmapC4 = sp.bmapC4;
    }
    return P;
}

static size_t h2_lds(const snesimage_ctx *c) { return sizeof(float) * 3 * (size_t)(c->ncol + 2); } // k_sparse_h2's palette table in LDS

// B of the current slot: compact list of contested pixels, then the pipeline once with checkpoints (main stream)
int32_t sparse_base_pass(snesimage_ctx *c, int sp_idx, int si, uint32_t n_cand) { // n_cand: candidates of the call (decides where B's downscale runs)
    auto &sp = c->sp;
    const Geom &G = c->G;
    if (!sp.plist_valid) {
        if (!sp.counters_cleared) HIPCHK(hipMemsetAsync(sp.plist_count, 0, sizeof(int), c->stream)); // normally k_prep did it
        sp.counters_cleared = false; // about to be used
        const unsigned long long *win_pack = c->d_pack;
        if (c->dither) {
            // B = the image dithered with the slot's entry out of play: the slot takes the colour of another entry j0 of its
            // subpalette (k_dither MODE 1), which records, per pixel, the dithered target and the key a candidate has to beat
            const uint32_t j0 = c->sub_size > 1 ? ((uint32_t)si + 1u) % c->sub_size : (uint32_t)si;
            sp.slot_ci = (uint32_t)(sp_idx * (int)c->sub_size + si);
            auto &ah = sp.ahead;
            const int *b_done = nullptr; // device flag: B's run for this slot and palette is in place already
            const bool same_state = ah.epoch == c->epoch; // nothing has touched the palette since (a sweep of snesimage_score_candidates over the slots)
            if (ah.have && ah.sp == sp_idx && ah.si == si && (same_state || (ah.epoch + 1 == c->epoch && c->epoch_by_commit))) {
                std::swap(sp.bmap, ah.bmap); std::swap(sp.bmapC4, ah.bmapC4); std::swap(sp.dpack, ah.dpack); std::swap(sp.ckd, ah.ckd); std::swap(sp.rec_lab, ah.rec_lab);
                hipLaunchKernelGGL(k_ahead_ok, dim3(1), dim3(1), 0, c->stream, same_state ? (const StepResult *)nullptr : c->d_last, ah.ok); // the one commit since: did it keep the palette?
                HIPCHK(hipStreamWaitEvent(c->stream, ah.ev, 0));
                b_done = ah.ok;
            }
            ah.have = false;
            HIPCHK(hipMemcpyAsync(sp.bcand, c->d_colors + 3 * (size_t)(sp_idx * (int)c->sub_size + (int)j0), 3, hipMemcpyDeviceToDevice, c->stream));
            float *btab = sp.cand_tab + 8 * (size_t)(sp.lanes * sp.cap);
            hipLaunchKernelGGL(k_candidate_tables, dim3(1), dim3(64), 0, c->stream, sp.bcand, 1, c->d_eotf, btab);
            DitherParams Dp{};
            Dp.orig = c->d_orig; Dp.tile_pal = c->d_tile_pal; Dp.pal_rgb8 = c->d_pal_rgb8; Dp.cand_tab = btab; Dp.maps = sp.bmap; Dp.mapsC4 = sp.bmapC4;
            Dp.W = (int)c->W; Dp.H = (int)c->H; Dp.sub_size = (int)c->sub_size; Dp.ncol = c->ncol; Dp.slot_ci = sp.slot_ci;
            Dp.rec_pack = sp.dpack; Dp.ck_out = sp.ckd; Dp.excl_sub = sp_idx; Dp.excl_si = si; Dp.excl_j0 = (int)j0; Dp.skip = b_done;
            if (c->perceptual) { // CIEDE2000: the stand-in's Lab in B's row of the candidates' table, the record in distance bits, the targets' Lab beside it
                float *blab = sp.cand_lab + 3 * (size_t)(sp.lanes * sp.cap);
                hipLaunchKernelGGL(k_candidate_lab, dim3(1), dim3(64), 0, c->stream, btab, 1, c->d_lab_eotf, blab);
                Dp.pal_lab = c->d_pal_lab; Dp.cand_lab = blab; Dp.lab_eotf = c->d_lab_eotf; Dp.perceptual = 1; Dp.rec_lab = sp.rec_lab;
                if (c->dither4) hipLaunchKernelGGL((k_dither4_lab<1>), dim3(1), dim3(512), 0, c->stream, Dp); else hipLaunchKernelGGL((k_dither<true, 0, 1>), dim3(1), dim3(128), 0, c->stream, Dp);
            } else
            if (c->dither4 && c->sub_size == 15) hipLaunchKernelGGL((k_dither4<15, 1>), dim3(1), dim3(512), 0, c->stream, Dp);
            else if (c->dither4) hipLaunchKernelGGL((k_dither4<0, 1>), dim3(1), dim3(512), 0, c->stream, Dp);
            else if (c->sub_size == 15) hipLaunchKernelGGL((k_dither<false, 15, 1>), dim3(1), dim3(128), 0, c->stream, Dp);
            else hipLaunchKernelGGL((k_dither<false, 0, 1>), dim3(1), dim3(128), 0, c->stream, Dp);
            win_pack = sp.dpack;
        }
        hipLaunchKernelGGL(k_build_plist, dim3((unsigned)((c->npx + 255) / 256)), dim3(256), 0, c->stream, win_pack, (int)c->npx, sp.plist, sp.plist_count);
        SparseParams P = sparse_params(c, c->nlanes); // B has its own item list and counters
        P.is_base = 1; P.ncand = 1; P.k0 = P.base;
        // B's H and V passes are long single-image sweeps (latency-bound); they run on their own stream beside the
        // candidates' scan / downscale / H pass, which only need B's linear-RGB rows.  The candidates' V pass waits for them.
        // (SNES_BASE_STREAM=0 keeps them on the context's stream: with many contexts on one device the extra streams only
        // add cross-queue waits.)
        hipStream_t bs = sp.side ? sp.base_stream : c->stream;
        // B's downscale goes with its sweeps where the call is long (round 4: it was the main stream's, 12 us in front of every call's
        // candidates): the pack and the contested list are what the candidates' scan and scale-1 downscale read; B's planes are first
        // read by the downscale of scales 2.., which waits for ev_base_h.  A short call waits for B's chain: there the downscale stays
        // in front of the hand-over, which would only delay it (0.267 -> 0.279 ms per 64-candidate call the other way).
        const bool down_with_sweeps = sp.side && n_cand >= sp.h0_min && sp.h0_min > 0;
        sp.base_down_side = down_with_sweeps;
        if (!down_with_sweeps) hipLaunchKernelGGL(k_base_down, dim3((unsigned)((G.W / 32) * ((G.H + 31) / 32))), dim3(256), 0, c->stream, P); // B: every row of every scale
        if (sp.side) { HIPCHK(hipEventRecord(sp.ev_base_in, c->stream)); HIPCHK(hipStreamWaitEvent(bs, sp.ev_base_in, 0)); }
        if (down_with_sweeps) hipLaunchKernelGGL(k_base_down, dim3((unsigned)((G.W / 32) * ((G.H + 31) / 32))), dim3(256), 0, bs, P);
        // (B's work items — every group, from column 0 — are in place since sparse_alloc)
        // wide scales: B rows start at column 0 (list s*kColBuckets) and leave the per-block H checkpoints and the scale-0 XYB plane
        if (sp.h2q_max > 0) hipLaunchKernelGGL(k_sparse_h2q_base, dim3((unsigned)((G.sh[0] / 4 * 3 + 3) / 4), (unsigned)(P.s_first * kColBuckets)), dim3(64), h2_lds(c), bs, P);
        else hipLaunchKernelGGL(k_sparse_h2_base, dim3((unsigned)((G.sh[0] / 4 * 3 + 15) / 16), (unsigned)(P.s_first * kColBuckets)), dim3(64), h2_lds(c), bs, P);
        if (sp.side) HIPCHK(hipEventRecord(sp.ev_base_h, bs)); // the candidates' H pass resumes from the block checkpoints this launch leaves
        // the wide scales' V sweep first: the candidates' V pass (the bulk of a call) waits for it alone; the narrow scales'
        // sweeps follow and are awaited by the candidates' narrow V pass at the very end of the launch group
        if (P.s_first > 0 && sp.vsplit) hipLaunchKernelGGL(k_sparse_v2_base_split, dim3((unsigned)(G.W >> 6), 3, (unsigned)P.s_first), dim3(128), 0, bs, P); // two waves per 64 columns: recurrences | maps and sums
        else if (P.s_first > 0) hipLaunchKernelGGL(k_sparse_v2_base, dim3(3, (unsigned)P.s_first), dim3(256), 0, bs, P);
        if (sp.side) HIPCHK(hipEventRecord(sp.ev_base_done, bs));
        if (P.s_first < G.nscales) hipLaunchKernelGGL(k_sparse_h, dim3((unsigned)((G.sh[P.s_first] / 4 * 3 + 15) / 16), (unsigned)((G.nscales - P.s_first) * kColBuckets)), dim3(64), 0, bs, P);
        if (P.s_first < G.nscales) hipLaunchKernelGGL(k_sparse_v_base_narrow, dim3(3, (unsigned)(G.nscales - P.s_first)), dim3(256), 0, bs, P);
        HIPCHK(hipGetLastError());
        if (sp.side) HIPCHK(hipEventRecord(sp.ev_base_narrow, bs));
        if (c->dither && sp.side && sp.ahead.dpack) { // B of the scheduler's next slot, behind this B's sweeps on their stream
            auto &ah = sp.ahead;
            int ni = si + 1, np = sp_idx;
            if (ni == (int)c->sub_size) { ni = 0; np = (np + 1) % (int)c->sub_count; }
            const int nj0 = (ni + 1) % (int)c->sub_size;
            HIPCHK(hipMemcpyAsync(ah.bcand, c->d_colors + 3 * (size_t)(np * (int)c->sub_size + nj0), 3, hipMemcpyDeviceToDevice, bs));
            hipLaunchKernelGGL(k_candidate_tables, dim3(1), dim3(64), 0, bs, ah.bcand, 1, c->d_eotf, ah.btab);
            DitherParams Dn{};
            Dn.orig = c->d_orig; Dn.tile_pal = c->d_tile_pal; Dn.pal_rgb8 = c->d_pal_rgb8; Dn.cand_tab = ah.btab; Dn.maps = ah.bmap; Dn.mapsC4 = ah.bmapC4;
            Dn.W = (int)c->W; Dn.H = (int)c->H; Dn.sub_size = (int)c->sub_size; Dn.ncol = c->ncol; Dn.slot_ci = (uint32_t)(np * (int)c->sub_size + ni);
            Dn.rec_pack = ah.dpack; Dn.ck_out = ah.ckd; Dn.excl_sub = np; Dn.excl_si = ni; Dn.excl_j0 = nj0;
            if (c->perceptual) {
                hipLaunchKernelGGL(k_candidate_lab, dim3(1), dim3(64), 0, bs, ah.btab, 1, c->d_lab_eotf, ah.blab);
                Dn.pal_lab = c->d_pal_lab; Dn.cand_lab = ah.blab; Dn.lab_eotf = c->d_lab_eotf; Dn.perceptual = 1; Dn.rec_lab = ah.rec_lab;
                if (c->dither4) hipLaunchKernelGGL((k_dither4_lab<1>), dim3(1), dim3(512), 0, bs, Dn); else hipLaunchKernelGGL((k_dither<true, 0, 1>), dim3(1), dim3(128), 0, bs, Dn);
            } else
            if (c->dither4 && c->sub_size == 15) hipLaunchKernelGGL((k_dither4<15, 1>), dim3(1), dim3(512), 0, bs, Dn);
            else if (c->dither4) hipLaunchKernelGGL((k_dither4<0, 1>), dim3(1), dim3(512), 0, bs, Dn);
            else if (c->sub_size == 15) hipLaunchKernelGGL((k_dither<false, 15, 1>), dim3(1), dim3(128), 0, bs, Dn);
            else hipLaunchKernelGGL((k_dither<false, 0, 1>), dim3(1), dim3(128), 0, bs, Dn);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(ah.ev, bs));
            ah.have = true; ah.sp = np; ah.si = ni; ah.epoch = c->epoch;
        }
        sp.plist_valid = true;
    }
    return SNES_OK;
}

int32_t sparse_score_chunk(snesimage_ctx *c, uint32_t lane, hipStream_t stream, const uint8_t *d_rgb5, uint32_t nc, double *d_errors, int err_stride, int err_offset) {
    auto &sp = c->sp;
    const Geom &G = c->G;
    SparseParams P = sparse_params(c, lane);
    P.is_base = 0; P.ncand = (int)nc; P.k0 = (int)(lane * sp.cap);
    snesimage_ctx::TimingRec tr{}; tr.n = nc;
    if (c->timing) { for (int i = 0; i < 6; i++) HIPCHK(hipEventCreate(&tr.ev[i])); if (c->timing == 1) HIPCHK(hipEventRecord(tr.ev[0], stream)); }
    hipLaunchKernelGGL(k_candidate_tables, dim3((nc + 63) / 64), dim3(64), 0, stream, d_rgb5, (int)nc, c->d_eotf, sp.cand_tab + 8 * (size_t)P.k0);
    if (c->dither) { // first pixel each candidate takes from B, then its own Floyd-Steinberg run from that 4-row group on
        if (c->perceptual) {
            hipLaunchKernelGGL(k_candidate_lab, dim3((nc + 63) / 64), dim3(64), 0, stream, sp.cand_tab + 8 * (size_t)P.k0, (int)nc, c->d_lab_eotf, sp.cand_lab + 3 * (size_t)P.k0);
            hipLaunchKernelGGL(k_dither_first_lab, dim3(nc), dim3(256), 0, stream, P);
        } else
        hipLaunchKernelGGL(k_dither_first, dim3((nc + 15) / 16), dim3(1024), 0, stream, P);
        DitherParams Dp{};
        Dp.orig = c->d_orig; Dp.tile_pal = c->d_tile_pal; Dp.pal_rgb8 = c->d_pal_rgb8; Dp.cand_tab = sp.cand_tab + 8 * (size_t)P.k0;
        Dp.maps = const_cast<uint8_t *>(P.maps); Dp.mapsC4 = const_cast<uint8_t *>(P.mapsC4);
        Dp.W = (int)c->W; Dp.H = (int)c->H; Dp.sub_size = (int)c->sub_size; Dp.ncol = c->ncol; Dp.slot_ci = sp.slot_ci;
        Dp.first_group = sp.first; Dp.first_k0 = P.k0; Dp.ck_in = sp.ckd; Dp.bmap = sp.bmap; Dp.bmapC4 = sp.bmapC4; Dp.rec_in = c->dither_rec ? sp.dpack : nullptr;
        if (c->sp.lpt) { // the resumed runs differ several-fold in length: longest first (the order is rebuilt for the V pass below)
            hipLaunchKernelGGL(k_sparse_order, dim3(1), dim3(1024), 0, stream, P, sp.order + P.k0); Dp.order = sp.order + P.k0;
        }
        if (c->perceptual) {
            Dp.pal_lab = c->d_pal_lab; Dp.cand_lab = sp.cand_lab + 3 * (size_t)P.k0; Dp.lab_eotf = c->d_lab_eotf; Dp.perceptual = 1;
            if (c->dither4 && nc <= c->dither4_max) hipLaunchKernelGGL((k_dither4_lab<2>), dim3(nc), dim3(512), 0, stream, Dp); else hipLaunchKernelGGL((k_dither<true, 0, 2>), dim3(nc), dim3(128), 0, stream, Dp);
        } else
        if (c->dither4 && nc <= c->dither4_max && c->sub_size == 15) hipLaunchKernelGGL((k_dither4<15, 2>), dim3(nc), dim3(512), 0, stream, Dp);
        else if (c->dither4 && nc <= c->dither4_max) hipLaunchKernelGGL((k_dither4<0, 2>), dim3(nc), dim3(512), 0, stream, Dp);
        else if (c->ditherw && c->sub_size == 15) hipLaunchKernelGGL((k_ditherw<15>), dim3((nc + 3) / 4), dim3(256), 0, stream, Dp, (int)nc);
        else if (c->ditherw && c->sub_size > 1) hipLaunchKernelGGL((k_ditherw<0>), dim3((nc + 3) / 4), dim3(256), 0, stream, Dp, (int)nc);
        else if (c->sub_size == 15) hipLaunchKernelGGL((k_dither<false, 15, 2>), dim3(nc), dim3(128), 0, stream, Dp);
        else hipLaunchKernelGGL((k_dither<false, 0, 2>), dim3(nc), dim3(128), 0, stream, Dp);
        hipLaunchKernelGGL(k_dither_diff, dim3((nc + 3) / 4), dim3(1024), 0, stream, P); // changed groups = where the maps differ
    } else if (c->perceptual) {
        hipLaunchKernelGGL(k_candidate_lab, dim3((nc + 63) / 64), dim3(64), 0, stream, sp.cand_tab + 8 * (size_t)P.k0, (int)nc, c->d_lab_eotf, sp.cand_lab + 3 * (size_t)P.k0);
        HIPCHK(hipMemsetAsync(sp.bitmap + (size_t)P.k0 * (c->npx / 32), 0, sizeof(uint32_t) * (c->npx / 32) * nc, stream));
        hipLaunchKernelGGL(k_sparse_scan_lab, dim3(nc), dim3(256), 0, stream, P);
    } else if (nc <= sp.scan4_max) hipLaunchKernelGGL(k_sparse_scan4, dim3((nc + 3) / 4), dim3(1024), 0, stream, P); // a list that leaves CUs idle: four waves per candidate, a quarter of the chain
    else hipLaunchKernelGGL(k_sparse_scan, dim3((nc + 15) / 16), dim3(1024), 0, stream, P);
    bool vn_aside = false;
    const bool h0_ahead = sp.side && sp.h0_min > 0 && nc >= sp.h0_min && nc > sp.h2q_max && lane < 8 && P.s_first > 0;
    const bool v0_aside = h0_ahead && sp.v0_aside;
    vn_aside = h0_ahead && !v0_aside && sp.vn_aside && P.s_first < G.nscales;
    if (c->sp.lpt && nc > 512) P.order = sp.order + P.k0; // (written by k_sparse_order before any V pass reads it)
    if (h0_ahead) {
        if (!sp.h0_stream[lane]) {
            HIPCHK(hipStreamCreateWithFlags(&sp.h0_stream[lane], hipStreamNonBlocking)); // (default priority: a lower or a higher one than the main stream's costs 40-50 % of the step)
            HIPCHK(hipEventCreateWithFlags(&sp.ev_scan[lane], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sp.ev_h0[lane], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sp.ev_hn[lane], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sp.ev_v0[lane], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sp.ev_vn[lane], hipEventDisableTiming));
        }
        hipStream_t hs = sp.h0_stream[lane];
        HIPCHK(hipEventRecord(sp.ev_scan[lane], stream));
        HIPCHK(hipStreamWaitEvent(hs, sp.ev_scan[lane], 0));  // the work items
        HIPCHK(hipStreamWaitEvent(hs, sp.ev_base_h, 0));      // B's H-pass checkpoints
        size_t gx = ((size_t)nc * (G.sh[0] / 4) * 3 + 15) / 16; if (gx > sp.h0_grid) gx = sp.h0_grid;
        hipLaunchKernelGGL(k_sparse_h2, dim3((unsigned)gx, (unsigned)kColBuckets), dim3(64), h2_lds(c), hs, P); // grid.y = scale 0's lists only
        if (c->sp.lpt) hipLaunchKernelGGL(k_sparse_order, dim3(1), dim3(1024), 0, hs, P, sp.order + P.k0); // (the V pass's order needs the scan only: off the main stream)
        HIPCHK(hipEventRecord(sp.ev_h0[lane], hs));
        if (v0_aside) { // scale 0's V pass needs scale 0's H pass, B's sweeps and the order: nothing of the main stream's
            HIPCHK(hipStreamWaitEvent(hs, sp.ev_base_done, 0)); // checkpoints and H output of B
            if (c->timing) HIPCHK(hipEventRecord(tr.ev[3], hs));
            hipLaunchKernelGGL(k_sparse_v2, dim3(nc * 3, 1), dim3(256), 0, hs, P); // grid.y = scale 0 only
            if (c->timing) HIPCHK(hipEventRecord(tr.ev[4], hs)); // ev[3]..ev[4]: k_sparse_v2 (scale 0: three quarters of the V pass) alone
            HIPCHK(hipEventRecord(sp.ev_v0[lane], hs));
        }
    }
    if (sp.down1 && G.nscales > 2) { // scale 1 one block per changed group (the scan's item lists name them), then the scales that do depend on each other
        size_t gd = (size_t)nc * 8; if (gd > sp.down1_grid) gd = sp.down1_grid; // (~6 changed groups per candidate; grid-stride beyond)
        hipLaunchKernelGGL(k_sparse_down1, dim3((unsigned)gd), dim3(256), 0, stream, P);
        if (sp.side && sp.base_down_side) HIPCHK(hipStreamWaitEvent(stream, sp.ev_base_h, 0)); // B's planes (downscaled on B's stream) and, for later, its H-pass checkpoints
        if (sp.down_tiles && G.nscales >= 4) { // scales 2..: one block per changed group of scale 3
            size_t gt = ((size_t)nc * (G.sh[3] / 4) + SNES_DOWN_TILES_U - 1) / SNES_DOWN_TILES_U; if (gt > sp.tiles_grid) gt = sp.tiles_grid; // (4-5 of a candidate's 8 groups change; the block takes SNES_DOWN_TILES_U at a time, grid-stride beyond)
            hipLaunchKernelGGL(k_sparse_down_tiles, dim3((unsigned)gt), dim3(256), 0, stream, P);
        } else
        hipLaunchKernelGGL(k_sparse_down, dim3(nc), dim3(256), 0, stream, P, -2);
    } else { if (sp.side && sp.base_down_side) HIPCHK(hipStreamWaitEvent(stream, sp.ev_base_h, 0)); hipLaunchKernelGGL(k_sparse_down, dim3(nc), dim3(256), 0, stream, P, 0); }
    if (sp.side && !sp.base_down_side) HIPCHK(hipStreamWaitEvent(stream, sp.ev_base_h, 0)); // B's H-pass checkpoints (a short list gets here before B's sweep is through)
    if (c->timing == 1) HIPCHK(hipEventRecord(tr.ev[1], stream)); // ev[1]..ev[2]: the candidates' H pass alone (the wait for B's sweep is before it)
    { size_t gx = ((size_t)nc * (G.sh[0] / 4) * 3 + 15) / 16; if (gx > sp.hgrid) gx = sp.hgrid; // grid-stride over the item quads
      if (nc <= sp.h2q_max) { // a short list: the H pass with a quad of lanes per row (a third of the chain, four times the waves)
          size_t gq = ((size_t)nc * (G.sh[0] / 4) * 3 + 3) / 4; if (gq > sp.hgrid) gq = sp.hgrid;
          hipLaunchKernelGGL(k_sparse_h2q, dim3((unsigned)gq, (unsigned)(P.s_first * kColBuckets)), dim3(64), h2_lds(c), stream, P);
      } else
      if (h0_ahead) { if (P.s_first > 1) hipLaunchKernelGGL(k_sparse_h2_from, dim3((unsigned)gx, (unsigned)((P.s_first - 1) * kColBuckets)), dim3(64), h2_lds(c), stream, P, (int)kColBuckets); } // (scale 0's lists went ahead)
      else
      hipLaunchKernelGGL(k_sparse_h2, dim3((unsigned)gx, (unsigned)(P.s_first * kColBuckets)), dim3(64), h2_lds(c), stream, P);
      if (P.s_first < G.nscales) hipLaunchKernelGGL(k_sparse_h, dim3((unsigned)((gx + 7) / 8), (unsigned)((G.nscales - P.s_first) * kColBuckets)), dim3(64), 0, stream, P);
      if (vn_aside) HIPCHK(hipEventRecord(sp.ev_hn[lane], stream)); // the narrow scales' H output is in place
      if (h0_ahead) HIPCHK(hipStreamWaitEvent(stream, sp.ev_h0[lane], 0)); }
    if (c->timing == 1) HIPCHK(hipEventRecord(tr.ev[2], stream));
    if (sp.side || stream != c->stream) HIPCHK(hipStreamWaitEvent(stream, sp.side ? sp.ev_base_done : c->ev_ready, 0)); // checkpoints and H output of B
    if (c->sp.lpt && nc > 512) { // (a short list's blocks are all resident at once: their order is immaterial)
        if (!h0_ahead) hipLaunchKernelGGL(k_sparse_order, dim3(1), dim3(1024), 0, stream, P, sp.order + P.k0);
        P.order = sp.order + P.k0;
    }
    if (vn_aside) { // the narrow scales' V pass — a tenth of the V pass's time, latency-bound — beside the wide scales' instead of behind it
        hipStream_t hs = sp.h0_stream[lane];
        HIPCHK(hipStreamWaitEvent(hs, sp.ev_hn[lane], 0));
        HIPCHK(hipStreamWaitEvent(hs, sp.ev_base_narrow, 0)); // B's narrow-scale sweeps
        hipLaunchKernelGGL(k_sparse_v, dim3((nc * 3 + 7) / 8, (unsigned)(G.nscales - P.s_first)), dim3(256), 0, hs, P);
        HIPCHK(hipEventRecord(sp.ev_vn[lane], hs));
    }
    if (v0_aside) { // scale 0 runs on the other stream; here the narrow scales first (a latency-bound 55 us that would otherwise end the group alone on the chip), then scales 1..
        if (sp.side) HIPCHK(hipStreamWaitEvent(stream, sp.ev_base_narrow, 0)); // B's narrow-scale sweeps
        if (P.s_first < G.nscales) hipLaunchKernelGGL(k_sparse_v, dim3((nc * 3 + 7) / 8, (unsigned)(G.nscales - P.s_first)), dim3(256), 0, stream, P);
        if (P.s_first > 1) hipLaunchKernelGGL(k_sparse_v2_from, dim3(nc * 3, (unsigned)(P.s_first - 1)), dim3(256), 0, stream, P, 1);
    } else {
    if (c->timing) HIPCHK(hipEventRecord(tr.ev[3], stream));
    hipLaunchKernelGGL(k_sparse_v2, dim3(nc * 3, (unsigned)P.s_first), dim3(256), 0, stream, P);
    if (c->timing) HIPCHK(hipEventRecord(tr.ev[4], stream)); // ev[3]..ev[4]: k_sparse_v2 alone
    }
    if (vn_aside) HIPCHK(hipStreamWaitEvent(stream, sp.ev_vn[lane], 0));
    else if (!v0_aside) {
    if (sp.side) HIPCHK(hipStreamWaitEvent(stream, sp.ev_base_narrow, 0)); // B's narrow-scale sweeps
    if (P.s_first < G.nscales) hipLaunchKernelGGL(k_sparse_v, dim3((nc * 3 + 7) / 8, (unsigned)(G.nscales - P.s_first)), dim3(256), 0, stream, P); // narrow scales: >= 8 pairs per block
    }
    if (v0_aside) HIPCHK(hipStreamWaitEvent(stream, sp.ev_v0[lane], 0));
    hipLaunchKernelGGL(k_final_score_wave, dim3((nc + 3) / 4), dim3(256), 0, stream, sp.part + (size_t)P.k0 * G.nscales * 18, (int)nc, G, d_errors, err_stride, err_offset, P.item_count, (int)kItemLists);
    if (c->dither) { // the lane remembers the map of its best candidate so far: the commit adopts the winner's instead of dithering again
        uint8_t *bm = lane == 0 ? c->d_bestmap : c->extra[lane - 1].d_bestmap; BestRec *br = lane == 0 ? c->d_bestrec : c->extra[lane - 1].d_bestrec;
        hipLaunchKernelGGL(k_keep_best, dim3(1), dim3(1024), 0, stream, d_errors, err_stride, err_offset, (int)nc, P.maps, (int)c->npx, br, bm);
    }
    HIPCHK(hipGetLastError());
    if (c->timing) { tr.light = c->timing != 1; if (!tr.light) HIPCHK(hipEventRecord(tr.ev[5], stream)); c->t_pending.push_back(tr); }
    return SNES_OK;
}

// Run score_chunk on an extra lane: its stream and per-chunk workspace stand in for the context's own.
struct LaneScope {
    snesimage_ctx *c; snesimage_ctx::Lane *L;
    LaneScope(snesimage_ctx *c_, snesimage_ctx::Lane *L_) : c(c_), L(L_) { swap(); }
    ~LaneScope() { swap(); }
    void swap() { std::swap(c->stream, L->stream); std::swap(c->d_work, L->d_work); std::swap(c->d_cand_tab, L->d_cand_tab); std::swap(c->d_cand_lab, L->d_cand_lab);
                  std::swap(c->d_part, L->d_part); std::swap(c->d_maps, L->d_maps); std::swap(c->d_mapsT, L->d_mapsT); std::swap(c->d_mapsC4, L->d_mapsC4); std::swap(c->d_mapsR4, L->d_mapsR4);
                  std::swap(c->d_bestmap, L->d_bestmap); std::swap(c->d_bestrec, L->d_bestrec); }
};

// errors of candidate j of the list go to d_errors[err_offset + j * err_stride]
int32_t score_list(snesimage_ctx *c, const uint8_t *d_rgb5, uint32_t n, double *d_errors, int err_stride, int err_offset, int sp, int si, uint8_t *d_maps_out) {
    // Launch groups of at most `chunk` candidates, dealt to the launch lanes.  With the RGB distance and no dither the stages
    // of a launch group saturate the chip one after the other (V pass: VALU, H pass: its writes): splitting a list that fits
    // one group over two lanes in lock-step only makes them share it (+3 % at 4,096 per call, every kernel's duration
    // doubled), so such a list is one group on one lane; longer lists alternate between the lanes.  The CIEDE2000 scan and the
    // resumed Floyd-Steinberg runs are latency-bound and do overlap the other lane's scoring: those lists are split evenly
    // (+7 % with --perceptual-palettes, +10 % with --dither).
    uint32_t chunk = (c->dither || c->perceptual) ? (n + c->nlanes - 1) / c->nlanes : n;
    if (chunk < 64) chunk = 64;
    if (chunk > c->chunk) chunk = c->chunk;
    // (--dither with one-entry subpalettes stays on the dense path: B then has no other entry to stand in for the slot's, so
    // the slot's index appears in B's map as well and a map comparison cannot tell the candidate's pixels from B's)
    const bool sparse = c->sp.enabled && !d_maps_out && sp >= 0 && n >= c->sp.min_n && c->pack_mode == (c->dither ? 1 : 2) && !(c->dither && c->sub_size == 1);
    const uint32_t nchunks = (n + chunk - 1) / chunk;
    const uint32_t nl = nchunks < c->nlanes ? nchunks : c->nlanes;
    CHECK(alloc_workspace(c, sparse ? 1 : chunk, nl));
    CHECK(ensure_tables(c));
    CHECK(ensure_source(c));
    if (sparse) {
        CHECK(sparse_alloc(c, chunk, nl));
        if (c->dither && (c->sp.base_sp != sp || c->sp.base_si != si)) c->sp.plist_valid = false;
        CHECK(sparse_base_pass(c, sp, si, n));
        c->sp.base_sp = sp; c->sp.base_si = si;
    }
    if (c->dither) { hipLaunchKernelGGL(k_reset_best, dim3(1), dim3(64), 0, c->stream, c->d_bestrecs_all, (int)c->nlanes); c->best_valid = true; }
    if (nl > 1) {
        HIPCHK(hipEventRecord(c->ev_ready, c->stream)); // pack, tables, candidates are ready
        for (uint32_t l = 1; l < nl; l++) HIPCHK(hipStreamWaitEvent(c->extra[l - 1].stream, c->ev_ready, 0));
    }
    uint32_t i = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += chunk, i++) {
        const uint32_t nc = (n - c0 < chunk) ? (n - c0) : chunk;
        const uint32_t lane = i % nl;
        uint8_t *mo = d_maps_out ? d_maps_out + (size_t)c0 * c->npx : nullptr;
        const int eo = err_offset + (int)c0 * err_stride;
        if (sparse) { CHECK(sparse_score_chunk(c, lane, lane == 0 ? c->stream : c->extra[lane - 1].stream, d_rgb5 + 3 * (size_t)c0, nc, d_errors, err_stride, eo)); continue; }
        if (lane == 0) CHECK(score_chunk(c, d_rgb5 + 3 * (size_t)c0, nc, d_errors, err_stride, eo, sp, si, mo));
        else { LaneScope ls(c, &c->extra[lane - 1]); CHECK(score_chunk(c, d_rgb5 + 3 * (size_t)c0, nc, d_errors, err_stride, eo, sp, si, mo)); }
    }
    for (uint32_t l = 1; l < nl; l++) {
        HIPCHK(hipEventRecord(c->extra[l - 1].done, c->extra[l - 1].stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->extra[l - 1].done, 0));
    }
    return SNES_OK;
}

int32_t ensure_map(snesimage_ctx *c);

// optimize() on the current palette: no-dither -> argmin per pixel; dither -> serial error diffusion kernel
int32_t do_optimize(snesimage_ctx *c, bool may_skip = false) {
    CHECK(ensure_tables(c));
    if (!c->dither) {
        c->pack_valid = false;
        CHECK(run_prep(c, 0, -1, -1));
    } else {
        CHECK(alloc_workspace(c, 1));
        if (c->perceptual) CHECK(ensure_source(c));
        // single pseudo-candidate whose slot index matches nothing
        hipLaunchKernelGGL(k_candidate_tables, dim3(1), dim3(64), 0, c->stream, c->d_dummy_cand, 1, c->d_eotf, c->d_cand_tab);
        hipLaunchKernelGGL(k_candidate_slot, dim3(1), dim3(64), 0, c->stream, c->d_cand_tab, 1, 0xffffffffu);
        DitherParams Dp{};
        Dp.orig = c->d_orig; Dp.tile_pal = c->d_tile_pal; Dp.pal_rgb8 = c->d_pal_rgb8; Dp.pal_lab = c->d_pal_lab; Dp.cand_tab = c->d_cand_tab; Dp.cand_lab = c->d_cand_lab;
        Dp.lab_eotf = c->d_lab_eotf; Dp.maps = c->d_map; Dp.skip = may_skip ? c->d_skip : nullptr;
        Dp.W = (int)c->W; Dp.H = (int)c->H; Dp.sub_size = (int)c->sub_size; Dp.ncol = c->ncol; Dp.slot_ci = 0xffffffffu; Dp.perceptual = c->perceptual ? 1 : 0;
        launch_dither(c, Dp, 1);
        HIPCHK(hipGetLastError());
        c->pack_valid = false;
    }
    c->inc_valid = false; c->map_synced = true; c->map_pending = false;
    return SNES_OK;
}

// error() of the stored palette_map -> d_out (device)
int32_t do_error(snesimage_ctx *c, double *d_out) {
    CHECK(ensure_map(c));
    CHECK(alloc_workspace(c, 1));
    CHECK(ensure_tables(c));
    CHECK(ensure_source(c));
    bool saved_dither = c->dither;
    c->dither = false; // evaluate the stored map: pack mode 1 carries every pixel's colour index
    c->pack_valid = false;
    int32_t rc = run_prep(c, 1, -1, -1);
    if (rc == SNES_OK) rc = score_chunk(c, c->d_dummy_cand, 1, d_out, 1, 0, -1, -1, nullptr);
    c->dither = saved_dither;
    c->pack_valid = false;
    return rc;
}

int32_t ensure_incumbent(snesimage_ctx *c) {
    if (c->inc_valid) return SNES_OK;
    CHECK(do_error(c, c->d_inc_err));
    c->inc_valid = true;
    return SNES_OK;
}

int32_t check_slot(snesimage_ctx *c, uint32_t palette, uint32_t index) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    if (palette >= c->sub_count || index >= c->sub_size) return fail(SNES_ERR_ARG, "palette slot out of range");
    return SNES_OK;
}

int32_t batch_quiesce(struct snesimage_batch *b);
int32_t set_device(snesimage_ctx *c) { HIPCHK(hipSetDevice(c->device)); return c->owner ? batch_quiesce(c->owner) : SNES_OK; }

int32_t drain_timing(snesimage_ctx *c) {
    for (auto &r : c->t_pending) {
        float ms = 0.0f;
        HIPCHK(hipEventSynchronize(r.ev[r.light ? 4 : 5]));
        if (!r.light) {
            HIPCHK(hipEventElapsedTime(&ms, r.ev[0], r.ev[5])); c->t_ms[0] += ms;
            HIPCHK(hipEventElapsedTime(&ms, r.ev[1], r.ev[2])); c->t_ms[1] += ms;
        }
        HIPCHK(hipEventElapsedTime(&ms, r.ev[3], r.ev[4])); c->t_ms[2] += ms;
        c->t_launches += 1; c->t_cands += r.n;
        for (int i = 0; i < 6; i++) (void)hipEventDestroy(r.ev[i]);
    }
    c->t_pending.clear();
    return SNES_OK;
}

int32_t gen_candidates(snesimage_ctx *c, uint32_t method, uint32_t palette, uint32_t index, uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n, uint32_t shard_rank = 0,
                       uint32_t shard_count = 1, double *d_errors = nullptr) {
    const uint64_t key = mix64(seed ^ (step_id * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull);
    hipLaunchKernelGGL(k_gen_candidates, dim3((n + 63) / 64), dim3(64), 0, c->stream, (int)method, (int)n, key, c->d_colors, (int)(palette * c->sub_size + index), (int)channel, c->d_cand,
                       (int)shard_rank, (int)shard_count, d_errors ? c->d_cand_sel : (uint8_t *)nullptr, d_errors);
    HIPCHK(hipGetLastError());
    return SNES_OK;
}

uint32_t method_count(uint32_t method, uint32_t n_random) { return method == SNES_METHOD_RANDOM ? (n_random ? n_random : 64u) : (method == SNES_METHOD_CHANNEL ? 32u : kNesColorCount); }

// optimize() owed by the last commit (see commit): run it before anything reads or replaces what it depends on
int32_t ensure_map(snesimage_ctx *c) {
    if (!c->map_pending) return SNES_OK;
    c->map_pending = false;
    const bool inc = c->inc_valid;
    CHECK(do_optimize(c));
    c->inc_valid = inc; // the incumbent error was taken from the winning candidate: it is the error of exactly this map
    return SNES_OK;
}

int32_t commit(snesimage_ctx *c, const double *d_errors, uint32_t n, uint32_t method, uint32_t palette, uint32_t index) {
    // (an optimize() still owed by the previous commit is simply superseded: nobody looked at that map)
    PaletteTables T{};
    if (c->tables_valid) { // keep the tables current: one entry changes
        T.eotf = c->d_eotf; T.lab_eotf = c->d_lab_eotf; T.rgb8 = c->d_pal_rgb8; T.lin = c->d_pal_lin; T.xyb = c->d_pal_xyb; T.lab = c->perceptual ? c->d_pal_lab : nullptr;
    }
    hipLaunchKernelGGL(k_commit, dim3(1), dim3(256), 0, c->stream, d_errors, (int)n, c->d_cand, c->d_colors, (int)(palette * c->sub_size + index), method == SNES_METHOD_NES ? 1 : 0, c->d_inc_err,
                       c->d_last, T);
    HIPCHK(hipGetLastError());
    c->pack_valid = false; c->epoch++; c->epoch_by_commit = true;
    const bool was_synced = c->map_synced;
    if (!c->dither) {
        // lib.rs:237 / 281 / 325 (and :906) re-run optimize() on the committed palette.  Nothing in the optimizer loop reads
        // that map (the next call builds its own pack), so it is computed when palette_map is read, not once per call.
        c->map_pending = true; c->map_synced = true;
        c->inc_valid = was_synced;
        return SNES_OK;
    }
    bool may_skip = false;
    if (c->d_skip) { // the winner's map is optimize() of the committed palette when a lane of this device scored it
        hipLaunchKernelGGL(k_take_best_map, dim3(1), dim3(1024), 0, c->stream, c->d_last, c->d_bestrecs_all, c->best_valid ? (int)c->nlanes : 0, c->d_bestmaps_all, (int)c->npx,
                           c->map_synced ? 1 : 0, c->d_map, c->d_skip);
        may_skip = true;
    }
    c->best_valid = false;
    CHECK(do_optimize(c, may_skip)); // lib.rs:237 / 281 / 325 (and :906)
    // k_commit left the committed state's error in d_inc_err (lib.rs:910 recomputes the same value) — unless the step started
    // from a map that did not belong to its palette and kept the palette: then optimize() has just replaced that map
    c->inc_valid = was_synced;
    return SNES_OK;
}

} // namespace

#include "kmeans_host.inc"

extern "C" {

const char *snesimage_last_error(void) { return g_err.c_str(); }
#ifndef SNES_SRC_HASH
#define SNES_SRC_HASH "unknown"
#endif
const char *snesimage_version(void) { return "snesimage_hip 0.2.0 (gfx950) src:" SNES_SRC_HASH; } // src: hash of the library's sources (csrc/Makefile)

int32_t snesimage_create(const uint8_t *rgba, uint32_t w, uint32_t h, uint32_t sub_count, uint32_t sub_size, uint32_t flags, int32_t device, snesimage_ctx **out) {
    if (!rgba || !out) return fail(SNES_ERR_ARG, "null pointer");
    *out = nullptr;
    if (w != 256) return fail(SNES_ERR_ARG, "image width must be 256 (tile stride is fixed at 32, lib.rs:58)");
    if (h < 8 || h > 256 || (h & (h - 1)) != 0) return fail(SNES_ERR_ARG, "image height must be a power of two in [8,256]");
    if (sub_count < 1 || sub_size < 1 || sub_count > 253 || sub_size > 253 || sub_count * sub_size > 253) return fail(SNES_ERR_ARG, "sub_count*sub_size must be in [1,253]");
    if (device < 0) return fail(SNES_ERR_ARG, "device must be a HIP device ordinal >= 0 (this library has no CPU path)");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device >= ndev) return fail(SNES_ERR_ARG, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    snesimage_ctx *c = new snesimage_ctx();
    c->device = device; c->W = w; c->H = h; c->sub_count = sub_count; c->sub_size = sub_size; c->flags = flags; c->ncol = (int)(sub_count * sub_size);
    c->dither = flags & SNES_DITHER; c->perceptual = flags & SNES_PERCEPTUAL; c->nes = flags & SNES_NES;
    c->npx = (size_t)w * h;
    c->K = make_blur_constants();
    if (const char *e = getenv("SNES_CHUNK")) { int v = atoi(e); if (v > 0) c->chunk = (uint32_t)v; }
    if (const char *e = getenv("SNES_LANES")) { int v = atoi(e); if (v >= 1 && v <= 8) c->nlanes = (uint32_t)v; }
    // the group-sparse path covers the no-dither remap (RGB keys or CIEDE2000) and the RGB Floyd-Steinberg remap
    c->sp.enabled = true; // every height the library takes (8 .. 256 rows; until round 4: 32 and more — B's downscale, which walks 32 x 32 blocks of pixels, read past images of 8 and 16 rows)
    if (const char *e = getenv("SNES_SPARSE")) c->sp.enabled = c->sp.enabled && atoi(e) != 0;
    if (const char *e = getenv("SNES_BASE_STREAM")) c->sp.side = atoi(e) != 0;
    if (const char *e = getenv("SNES_LPT")) c->sp.lpt = atoi(e) != 0;
    if (const char *e = getenv("SNES_DOWN1")) c->sp.down1 = atoi(e) != 0;
    if (const char *e = getenv("SNES_DITHER4")) c->dither4 = atoi(e) != 0;
    if (const char *e = getenv("SNES_DITHERW")) c->ditherw = atoi(e) != 0;
    if (const char *e = getenv("SNES_DITHER_AHEAD")) c->sp.ahead.on = atoi(e) != 0;
    if (const char *e = getenv("SNES_DITHER_REC")) c->dither_rec = atoi(e) != 0;
    if (const char *e = getenv("SNES_DITHER4_MAX")) { int v = atoi(e); if (v >= 0) c->dither4_max = (uint32_t)v; }
    if (const char *e = getenv("SNES_DOWN1_GRID")) { int v = atoi(e); if (v >= 64) c->sp.down1_grid = (uint32_t)v; }
    if (const char *e = getenv("SNES_VSPLIT")) c->sp.vsplit = atoi(e) != 0;
    if (const char *e = getenv("SNES_DOWN_TILES")) c->sp.down_tiles = atoi(e) != 0;
    if (const char *e = getenv("SNES_V0_ASIDE")) c->sp.v0_aside = atoi(e) != 0;
    if (const char *e = getenv("SNES_VN_ASIDE")) c->sp.vn_aside = atoi(e) != 0;
    if (const char *e = getenv("SNES_H0_GRID")) { int v = atoi(e); if (v >= 16) c->sp.h0_grid = (uint32_t)v; }
    if (const char *e = getenv("SNES_H0_MIN")) { int v = atoi(e); if (v >= 0) c->sp.h0_min = (uint32_t)v; }
    if (const char *e = getenv("SNES_TILES_GRID")) { int v = atoi(e); if (v >= 64) c->sp.tiles_grid = (uint32_t)v; }
    if (const char *e = getenv("SNES_SCAN4_MAX")) { int v = atoi(e); if (v >= 0) c->sp.scan4_max = (uint32_t)v; }
    if (const char *e = getenv("SNES_H2Q_MAX")) { int v = atoi(e); if (v >= 0) c->sp.h2q_max = (uint32_t)v; }
    if (const char *e = getenv("SNES_HGRID")) { int v = atoi(e); if (v >= 1) c->sp.hgrid = (uint32_t)v; }
    if (const char *e = getenv("SNES_SPARSE_MIN")) { int v = atoi(e); if (v >= 1) c->sp.min_n = (uint32_t)v; }
    Geom &G = c->G;
    G.W = (int)w; G.H = (int)h; G.nscales = 0;
    // ssimulacra2's scale loop tests the size BEFORE downscaling (`if width < 8 || height < 8 { break }` then
    // `downscale_by_2`), so the last scale computed may be as small as 4 rows: 256x8 -> scales 256x8 and 128x4.
    for (int s = 0; s < kMaxScales; s++) {
        if (s > 0 && (G.sw[s - 1] < 8 || G.sh[s - 1] < 8)) break;
        G.sw[s] = (int)w >> s; G.sh[s] = (int)h >> s; G.nscales = s + 1;
    }
    long long off = 0, soff = 0;
    for (int s = 0; s < G.nscales; s++) {
        long long N = (long long)G.sw[s] * G.sh[s];
        G.src_off[s] = soff; soff += 3 * N;
        if (s >= 1) { G.off_xyb[s] = off; off += 3 * N; G.off_xybT[s] = off; off += 3 * N; } else { G.off_xyb[s] = 0; G.off_xybT[s] = 0; }
        G.off_hout[s] = off; off += 9 * N;
    }
    G.cand_stride = off;
    for (int s = 0; s < G.nscales; s++) if ((G.sw[s] % 64) == 0 && (G.sh[s] % 64) == 0) c->fast_mask |= 1 << s;
    if (const char *e = getenv("SNES_NO_FAST")) { if (atoi(e)) c->fast_mask = 0; }
    c->src_floats = (size_t)soff;

    int32_t rc = SNES_OK;
    auto body = [&]() -> int32_t {
        HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIPCHK(hipMalloc(&c->d_orig, c->npx * 4));
        HIPCHK(hipMalloc(&c->d_tile_pal, 1024));
        HIPCHK(hipMalloc(&c->d_colors, 3 * 256));
        HIPCHK(hipMalloc(&c->d_map, c->npx));
        HIPCHK(hipMalloc(&c->d_pack, c->npx * 8));
        HIPCHK(hipMalloc(&c->d_packT, c->npx * 8));
        HIPCHK(hipMalloc(&c->d_packC4, c->npx * 8));
        HIPCHK(hipMalloc(&c->d_packR4, c->npx * 8));
        if (c->dither) { HIPCHK(hipMalloc(&c->d_subC4, c->npx)); HIPCHK(hipMalloc(&c->d_subR4, c->npx)); }
        HIPCHK(hipMalloc(&c->d_eotf, 256 * 4));
        HIPCHK(hipMalloc(&c->d_lab_eotf, 256 * 4));
        HIPCHK(hipMalloc(&c->d_pal_rgb8, 256 * 4));
        HIPCHK(hipMalloc(&c->d_pal_lin, 256 * 3 * 4));
        HIPCHK(hipMalloc(&c->d_pal_xyb, 256 * 3 * 4));
        HIPCHK(hipMalloc(&c->d_pal_lab, 256 * 3 * 4));
        HIPCHK(hipMalloc(&c->d_lin0, c->npx * 3 * 4));
        HIPCHK(hipMalloc(&c->d_img1, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_img1T, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_mu1, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_sd1, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_a1, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_a1R4, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_r1, c->src_floats * 8));
        HIPCHK(hipMalloc(&c->d_r1R4, c->src_floats * 8));
        HIPCHK(hipMalloc(&c->d_img1C4, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_mu1R4, c->src_floats * 4));
        HIPCHK(hipMalloc(&c->d_sd1R4, c->src_floats * 4));
        if (c->perceptual) { HIPCHK(hipMalloc(&c->d_labpx, c->npx * 3 * 4)); HIPCHK(hipMalloc(&c->d_labpxT, c->npx * 3 * 4)); }
        HIPCHK(hipMalloc(&c->d_inc_err, sizeof(double)));
        HIPCHK(hipMalloc(&c->d_scratch_err, sizeof(double)));
        HIPCHK(hipMalloc(&c->d_last, sizeof(StepResult)));
        HIPCHK(hipMalloc(&c->d_dummy_cand, 64));
        HIPCHK(hipMemsetAsync(c->d_dummy_cand, 0, 64, c->stream));
        HIPCHK(hipMemsetAsync(c->d_last, 0, sizeof(StepResult), c->stream));
        HIPCHK(hipMemcpyAsync(c->d_orig, rgba, c->npx * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemsetAsync(c->d_tile_pal, 0, 1024, c->stream));   // lib.rs:58
        HIPCHK(hipMemsetAsync(c->d_colors, 0, 3 * 256, c->stream));  // lib.rs:756
        HIPCHK(hipMemsetAsync(c->d_map, 0, c->npx, c->stream));      // lib.rs:60
        c->h_orig.assign(rgba, rgba + c->npx * 4);
        make_eotf_tables(c->h_eotf, c->h_lab_eotf);
        HIPCHK(hipMemcpyAsync(c->d_eotf, c->h_eotf, sizeof(c->h_eotf), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_lab_eotf, c->h_lab_eotf, sizeof(c->h_lab_eotf), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        CHECK(ensure_cand_capacity(c, 64));
        return SNES_OK;
    };
    rc = body();
    if (rc != SNES_OK) { std::string keep = g_err; snesimage_destroy(c); g_err = keep; return rc; }
    *out = c;
    return SNES_OK;
}

void batch_forget(struct snesimage_batch *b, snesimage_ctx *c);
void group_forget(struct snesimage_group *g, snesimage_ctx *c);
namespace { void window_free(struct snesimage_window *w); }
void snesimage_destroy(snesimage_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->win) { if (c->stream) (void)hipStreamSynchronize(c->stream); window_free(c->win); c->win = nullptr; } // its slot contexts borrow this context's planes
    if (c->owner) batch_forget(c->owner, c); // waits for the batch's stream and retires the batch
    if (c->group) group_forget(c->group, c); // retires the group: its other members are their own again
    if (c->ev_own) (void)hipEventDestroy(c->ev_own);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)drain_timing(c);
    dfree(c->d_packC4); dfree(c->d_packR4); dfree(c->d_subC4); dfree(c->d_subR4); dfree(c->d_mapsC4); dfree(c->d_mapsR4); dfree(c->d_img1C4); dfree(c->d_mu1R4); dfree(c->d_sd1R4); dfree(c->d_r1); dfree(c->d_r1R4); dfree(c->d_a1); dfree(c->d_a1R4);
    dfree(c->d_orig); dfree(c->d_tile_pal); dfree(c->d_colors); dfree(c->d_map); dfree(c->d_pack); dfree(c->d_packT); dfree(c->d_eotf); dfree(c->d_lab_eotf);
    dfree(c->d_pal_rgb8); dfree(c->d_pal_lin); dfree(c->d_pal_xyb); dfree(c->d_pal_lab); dfree(c->d_lin0); dfree(c->d_img1); dfree(c->d_img1T); dfree(c->d_mu1); dfree(c->d_sd1);
    dfree(c->d_bestmaps_all); dfree(c->d_bestrecs_all); dfree(c->d_skip); dfree(c->d_rplist); dfree(c->d_rcount); dfree(c->d_rtab); dfree(c->d_rlab); dfree(c->d_tile_cost); dfree(c->d_tile_any); dfree(c->d_tile_moved);
    dfree(c->d_labpx); dfree(c->d_labpxT); dfree(c->d_work); dfree(c->d_cand_tab); dfree(c->d_cand_lab); dfree(c->d_part); dfree(c->d_maps); dfree(c->d_mapsT);
    dfree(c->d_cand); dfree(c->d_cand_sel); dfree(c->d_errs); dfree(c->d_errs_sel); dfree(c->d_inc_err); dfree(c->d_last); dfree(c->d_scratch_err); dfree(c->d_dummy_cand);
    for (auto &L : c->extra) { if (L.stream) (void)hipStreamSynchronize(L.stream); dfree(L.d_mapsC4); dfree(L.d_mapsR4); dfree(L.d_work); dfree(L.d_cand_tab); dfree(L.d_cand_lab); dfree(L.d_part); dfree(L.d_maps); dfree(L.d_mapsT); if (L.done) (void)hipEventDestroy(L.done); if (L.stream) (void)hipStreamDestroy(L.stream); }
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    { auto &q = c->sp; for (int i = 0; i < 8; i++) if (q.h0_stream[i]) { (void)hipStreamSynchronize(q.h0_stream[i]); (void)hipStreamDestroy(q.h0_stream[i]); (void)hipEventDestroy(q.ev_scan[i]); (void)hipEventDestroy(q.ev_h0[i]); (void)hipEventDestroy(q.ev_hn[i]); (void)hipEventDestroy(q.ev_vn[i]); (void)hipEventDestroy(q.ev_v0[i]); }
      if (q.base_stream) { (void)hipStreamSynchronize(q.base_stream); (void)hipStreamDestroy(q.base_stream); (void)hipEventDestroy(q.ev_base_in); (void)hipEventDestroy(q.ev_base_h); (void)hipEventDestroy(q.ev_base_narrow); (void)hipEventDestroy(q.ev_base_done); } dfree(q.store); dfree(q.cand_tab); dfree(q.ckf); dfree(q.cka); dfree(q.part); dfree(q.meta); dfree(q.items); dfree(q.item_count); dfree(q.plist); dfree(q.order); dfree(q.first); dfree(q.cand_lab); dfree(q.bitmap); dfree(q.ckh);
      dfree(q.dmaps); dfree(q.dmapsC4); dfree(q.bmap); dfree(q.bmapC4); dfree(q.bcand); dfree(q.dpack); dfree(q.ckd);
      dfree(q.rec_lab); dfree(q.ahead.blab); dfree(q.ahead.rec_lab); dfree(q.ahead.bmap); dfree(q.ahead.bmapC4); dfree(q.ahead.bcand); dfree(q.ahead.dpack); dfree(q.ahead.ckd); dfree(q.ahead.btab); dfree(q.ahead.ok); if (q.ahead.ev) (void)hipEventDestroy(q.ahead.ev); }
    kmeans_free(c->km);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int32_t snesimage_set_stream(snesimage_ctx *c, void *s) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return SNES_OK;
}
int32_t snesimage_sync(snesimage_ctx *c) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}
int32_t snesimage_set_chunk(snesimage_ctx *c, uint32_t chunk) {
    if (!c || chunk == 0 || chunk > 65535) return fail(SNES_ERR_ARG, "chunk must be in [1,65535]");
    if (c->owner) return fail(SNES_ERR_STATE, "the context is lent to a batch, whose launches are sized for the current chunk: destroy the batch first");
    CHECK(set_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->chunk = chunk;
    c->sp.cap = 0; c->sp.lanes = 0; c->sp.plist_valid = false; // storage indices of the row-sparse path depend on the chunk size
    return SNES_OK;
}

int32_t snesimage_optimize(snesimage_ctx *c) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    CHECK(do_optimize(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}

int32_t snesimage_error(snesimage_ctx *c, double *out) {
    if (!c || !out) return fail(SNES_ERR_ARG, "null pointer");
    CHECK(set_device(c));
    CHECK(do_error(c, c->d_scratch_err));
    HIPCHK(hipMemcpyAsync(out, c->d_scratch_err, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}

int32_t snesimage_score_candidates_device(snesimage_ctx *c, uint32_t palette, uint32_t index, const uint8_t *d_rgb5, uint32_t n, double *d_errors, uint8_t *d_maps_out) {
    CHECK(check_slot(c, palette, index));
    if (!d_rgb5 || !d_errors) return fail(SNES_ERR_ARG, "null pointer");
    if (n == 0) return SNES_OK;
    CHECK(set_device(c));
    CHECK(prep_for_slot(c, (int)palette, (int)index));
    CHECK(score_list(c, d_rgb5, n, d_errors, 1, 0, (int)palette, (int)index, d_maps_out));
    return SNES_OK;
}

int32_t snesimage_score_candidates(snesimage_ctx *c, uint32_t palette, uint32_t index, const uint8_t *rgb5, uint32_t n, double *errors) {
    CHECK(check_slot(c, palette, index));
    if (!rgb5 || !errors) return fail(SNES_ERR_ARG, "null pointer");
    if (n == 0) return SNES_OK;
    CHECK(set_device(c));
    CHECK(ensure_cand_capacity(c, n));
    HIPCHK(hipMemcpyAsync(c->d_cand_sel, rgb5, 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    CHECK(snesimage_score_candidates_device(c, palette, index, c->d_cand_sel, n, c->d_errs_sel, nullptr));
    HIPCHK(hipMemcpyAsync(errors, c->d_errs_sel, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}

int32_t snesimage_remap_candidates_device(snesimage_ctx *c, uint32_t palette, uint32_t index, const uint8_t *d_rgb5, uint32_t n, uint8_t *d_maps_out) {
    CHECK(check_slot(c, palette, index));
    if (!d_rgb5 || !d_maps_out) return fail(SNES_ERR_ARG, "null pointer");
    if (n == 0) return SNES_OK;
    CHECK(set_device(c));
    const uint32_t chunk = n < c->chunk ? n : c->chunk;
    CHECK(alloc_workspace(c, 1)); // (lanes, d_skip; the maps go straight to the caller's buffer: no per-candidate planes — until round 4 this
    // call sized the dense scoring workspace, 4.5 MB per candidate and lane, for its chunk and left it with the context)
    if (c->rtab_cap < chunk) {
        HIPCHK(hipStreamSynchronize(c->stream));
        c->rtab_cap = 0; dfree(c->d_rtab); dfree(c->d_rlab);
        HIPCHK(dmalloc(&c->d_rtab, sizeof(float) * 8 * (size_t)chunk)); HIPCHK(dmalloc(&c->d_rlab, sizeof(float) * 3 * (size_t)chunk));
        c->rtab_cap = chunk;
    }
    CHECK(ensure_tables(c));
    if (c->perceptual) CHECK(ensure_source(c));
    CHECK(prep_for_slot(c, (int)palette, (int)index));
    const uint32_t slot_ci = palette * c->sub_size + index;
    for (uint32_t c0 = 0; c0 < n; c0 += chunk) {
        const uint32_t nc = (n - c0 < chunk) ? (n - c0) : chunk;
        const uint8_t *rgb5 = d_rgb5 + 3 * (size_t)c0;
        uint8_t *maps = d_maps_out + (size_t)c0 * c->npx;
        hipLaunchKernelGGL(k_candidate_tables, dim3((nc + 63) / 64), dim3(64), 0, c->stream, rgb5, (int)nc, c->d_eotf, c->d_rtab);
        if (c->perceptual) hipLaunchKernelGGL(k_candidate_lab, dim3((nc + 63) / 64), dim3(64), 0, c->stream, c->d_rtab, (int)nc, c->d_lab_eotf, c->d_rlab);
        if (c->dither) {
            hipLaunchKernelGGL(k_candidate_slot, dim3((nc + 63) / 64), dim3(64), 0, c->stream, c->d_rtab, (int)nc, slot_ci);
            DitherParams Dp{};
            Dp.orig = c->d_orig; Dp.tile_pal = c->d_tile_pal; Dp.pal_rgb8 = c->d_pal_rgb8; Dp.pal_lab = c->d_pal_lab; Dp.cand_tab = c->d_rtab; Dp.cand_lab = c->d_rlab;
            Dp.lab_eotf = c->d_lab_eotf; Dp.maps = maps; Dp.mapsC4 = nullptr;
            Dp.W = (int)c->W; Dp.H = (int)c->H; Dp.sub_size = (int)c->sub_size; Dp.ncol = c->ncol; Dp.slot_ci = slot_ci; Dp.perceptual = c->perceptual ? 1 : 0;
            launch_dither(c, Dp, nc);
        } else {
            MapsParams M{}; M.pack = c->d_pack; M.cand_tab = c->d_rtab; M.cand_lab = c->d_rlab; M.labpx = c->d_labpx; M.maps = maps;
            M.npx = (int)c->npx; M.ncol = c->ncol; M.sub_size = (int)c->sub_size; M.si = (int)index; M.ncand = (int)nc; M.perceptual = c->perceptual ? 1 : 0;
            const dim3 grid((unsigned)((c->npx / 4 + 255) / 256), (nc + kRemapCands - 1) / kRemapCands);
            if (c->perceptual) { // B's map for everyone, then the CIEDE2000 win tests over the slot's contested pixels only
                if (!c->d_rplist) { HIPCHK(hipMalloc(&c->d_rplist, sizeof(uint4) * c->npx)); HIPCHK(hipMalloc(&c->d_rcount, sizeof(int))); }
                if (c0 == 0) {
                    HIPCHK(hipMemsetAsync(c->d_rcount, 0, sizeof(int), c->stream));
                    hipLaunchKernelGGL(k_build_plist, dim3((unsigned)((c->npx + 255) / 256)), dim3(256), 0, c->stream, c->d_pack, (int)c->npx, c->d_rplist, c->d_rcount);
                }
                hipLaunchKernelGGL(k_remap_fill4, grid, dim3(256), 0, c->stream, M);
                hipLaunchKernelGGL(k_remap_won_lab, dim3(nc), dim3(256), 0, c->stream, M, (const uint4 *)c->d_rplist, (const int *)c->d_rcount);
            } else hipLaunchKernelGGL((k_remap4<false>), grid, dim3(256), 0, c->stream, M);
        }
    }
    HIPCHK(hipGetLastError());
    return SNES_OK;
}

int32_t snesimage_step_async(snesimage_ctx *c, uint32_t method, uint32_t palette, uint32_t index, uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_random) {
    CHECK(check_slot(c, palette, index));
    if (method > 2 || channel > 2) return fail(SNES_ERR_ARG, "bad method or channel");
    CHECK(set_device(c));
    const uint32_t n = method_count(method, n_random);
    CHECK(ensure_cand_capacity(c, n));
    if (method != SNES_METHOD_NES) CHECK(ensure_incumbent(c)); // lib.rs:199, 294 (nes: f64::MAX, lib.rs:250)
    CHECK(gen_candidates(c, method, palette, index, channel, seed, step_id, n));
    CHECK(prep_for_slot(c, (int)palette, (int)index));
    CHECK(score_list(c, c->d_cand, n, c->d_errs, 1, 0, (int)palette, (int)index, nullptr));
    CHECK(commit(c, c->d_errs, n, method, palette, index));
    return SNES_OK;
}

int32_t snesimage_last_step(snesimage_ctx *c, double *best_error, uint8_t *best_rgb5, int32_t *best_k) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    StepResult r;
    HIPCHK(hipMemcpyAsync(&r, c->d_last, sizeof(r), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (best_error) *best_error = r.error;
    if (best_rgb5) { best_rgb5[0] = r.rgb5[0]; best_rgb5[1] = r.rgb5[1]; best_rgb5[2] = r.rgb5[2]; }
    if (best_k) *best_k = r.best_k;
    return SNES_OK;
}

int32_t snesimage_step(snesimage_ctx *c, uint32_t method, uint32_t palette, uint32_t index, uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_random, double *best_error,
                       uint8_t *best_rgb5) {
    CHECK(snesimage_step_async(c, method, palette, index, channel, seed, step_id, n_random));
    return snesimage_last_step(c, best_error, best_rgb5, nullptr);
}

int32_t snesimage_step_begin(snesimage_ctx *c, uint32_t method, uint32_t palette, uint32_t index, uint32_t channel, uint64_t seed, uint64_t step_id, uint32_t n_total, uint32_t shard_rank,
                             uint32_t shard_count, double *d_errors) {
    CHECK(check_slot(c, palette, index));
    if (method > 2 || channel > 2 || shard_count == 0 || shard_rank >= shard_count || !d_errors) return fail(SNES_ERR_ARG, "bad step_begin arguments");
    CHECK(set_device(c));
    const uint32_t n = method_count(method, n_total);
    CHECK(ensure_cand_capacity(c, n));
    if (method != SNES_METHOD_NES) CHECK(ensure_incumbent(c));
    CHECK(gen_candidates(c, method, palette, index, channel, seed, step_id, n, shard_rank, shard_count, d_errors)); // + this shard's list, errors preset to +inf
    const uint32_t n_own = (n > shard_rank) ? (n - shard_rank + shard_count - 1) / shard_count : 0;
    if (n_own) {
        CHECK(prep_for_slot(c, (int)palette, (int)index));
        // candidate j of the shard is global candidate shard_rank + j*shard_count
        CHECK(score_list(c, c->d_cand_sel, n_own, d_errors, (int)shard_count, (int)shard_rank, (int)palette, (int)index, nullptr));
    }
    if (!n_own) c->best_valid = false;
    c->pend = true; c->pend_n = n; c->pend_sp = palette; c->pend_si = index; c->pend_method = method;
    return SNES_OK;
}

int32_t snesimage_step_commit(snesimage_ctx *c, const double *d_errors) {
    if (!c || !d_errors) return fail(SNES_ERR_ARG, "null pointer");
    if (!c->pend) return fail(SNES_ERR_STATE, "step_commit without step_begin");
    CHECK(set_device(c));
    c->pend = false;
    return commit(c, d_errors, c->pend_n, c->pend_method, c->pend_sp, c->pend_si);
}

int32_t snesimage_get_tile_palettes(snesimage_ctx *c, uint8_t *out) {
    if (!c || !out) return fail(SNES_ERR_ARG, "null pointer");
    CHECK(set_device(c));
    HIPCHK(hipMemcpyAsync(out, c->d_tile_pal, 1024, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}
int32_t snesimage_set_tile_palettes(snesimage_ctx *c, const uint8_t *in) {
    if (!c || !in) return fail(SNES_ERR_ARG, "null pointer");
    for (int i = 0; i < 1024; i++) if (in[i] >= c->sub_count) return fail(SNES_ERR_ARG, "tile palette index out of range");
    CHECK(set_device(c));
    CHECK(ensure_map(c)); // the owed optimize() belongs to the state before this change
    HIPCHK(hipMemcpyAsync(c->d_tile_pal, in, 1024, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pack_valid = false; c->inc_valid = false; c->map_synced = false; c->epoch++; c->epoch_by_commit = false;
    return SNES_OK;
}
int32_t snesimage_get_palette_rgb5(snesimage_ctx *c, uint8_t *out) {
    if (!c || !out) return fail(SNES_ERR_ARG, "null pointer");
    CHECK(set_device(c));
    HIPCHK(hipMemcpyAsync(out, c->d_colors, 3 * (size_t)c->ncol, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}
int32_t snesimage_set_palette_rgb5(snesimage_ctx *c, const uint8_t *in) {
    if (!c || !in) return fail(SNES_ERR_ARG, "null pointer");
    CHECK(set_device(c));
    CHECK(ensure_map(c));
    HIPCHK(hipMemcpyAsync(c->d_colors, in, 3 * (size_t)c->ncol, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->tables_valid = false; c->pack_valid = false; c->inc_valid = false; c->map_synced = false; c->epoch++; c->epoch_by_commit = false;
    return SNES_OK;
}
int32_t snesimage_get_palette_u16(snesimage_ctx *c, uint16_t *out) {
    if (!c || !out) return fail(SNES_ERR_ARG, "null pointer");
    std::vector<uint8_t> raw(3 * (size_t)c->ncol);
    CHECK(snesimage_get_palette_rgb5(c, raw.data()));
    for (int i = 0; i < c->ncol; i++) out[i] = rgb5_as_u16(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
    return SNES_OK;
}
int32_t snesimage_get_palette_map(snesimage_ctx *c, uint8_t *out) {
    if (!c || !out) return fail(SNES_ERR_ARG, "null pointer");
    CHECK(set_device(c));
    CHECK(ensure_map(c));
    HIPCHK(hipMemcpyAsync(out, c->d_map, c->npx, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SNES_OK;
}
int32_t snesimage_set_palette_map(snesimage_ctx *c, const uint8_t *in) {
    if (!c || !in) return fail(SNES_ERR_ARG, "null pointer");
    for (size_t i = 0; i < c->npx; i++) if (in[i] >= c->sub_size) return fail(SNES_ERR_ARG, "palette_map entry out of range");
    CHECK(set_device(c));
    c->map_pending = false; // replaced wholesale
    HIPCHK(hipMemcpyAsync(c->d_map, in, c->npx, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pack_valid = false; c->inc_valid = false; c->map_synced = false; c->epoch++; c->epoch_by_commit = false;
    return SNES_OK;
}

// lib.rs:550-577 (host-side reconstruction from the device state; not on the hot path)
int32_t snesimage_as_rgba(snesimage_ctx *c, uint8_t *out) {
    if (!c || !out) return fail(SNES_ERR_ARG, "null pointer");
    std::vector<uint8_t> map(c->npx), tp(1024), col(3 * (size_t)c->ncol), orig(c->npx * 4);
    CHECK(snesimage_get_palette_map(c, map.data()));
    CHECK(snesimage_get_tile_palettes(c, tp.data()));
    CHECK(snesimage_get_palette_rgb5(c, col.data()));
    HIPCHK(hipMemcpy(orig.data(), c->d_orig, c->npx * 4, hipMemcpyDeviceToHost));
    memset(out, 0, c->npx * 4);
    for (uint32_t y = 0; y < c->H; y++)
        for (uint32_t x = 0; x < c->W; x++) {
            size_t px = (size_t)y * c->W + x;
            if (orig[4 * px + 3] == 0) continue;
            size_t ci = (size_t)tp[(y / 8) * 32 + (x / 8)] * c->sub_size + map[px];
            uint32_t v = rgb5_to_rgb8(col[3 * ci], col[3 * ci + 1], col[3 * ci + 2]);
            out[4 * px] = v & 0xff; out[4 * px + 1] = (v >> 8) & 0xff; out[4 * px + 2] = (v >> 16) & 0xff; out[4 * px + 3] = 255;
        }
    return SNES_OK;
}

// lib.rs:579-625 + :1002; serde_json (no preserve_order) emits keys sorted: palette, tile_palettes, tiles
int64_t snesimage_as_json(snesimage_ctx *c, char *out, int64_t cap) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    std::vector<uint8_t> map(c->npx), tp(1024), col(3 * (size_t)c->ncol), orig(c->npx * 4);
    if (snesimage_get_palette_map(c, map.data()) || snesimage_get_tile_palettes(c, tp.data()) || snesimage_get_palette_rgb5(c, col.data())) return SNES_ERR_HIP;
    if (hipMemcpy(orig.data(), c->d_orig, c->npx * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(SNES_ERR_HIP, "hipMemcpy failed");
    const uint32_t wt = c->W / 8, ht = c->H / 8;
    std::string s = "{\"palette\":[";
    for (uint32_t p = 0; p < c->sub_count; p++)
        for (uint32_t i = 0; i < 16; i++) {
            unsigned v = 0;
            if (i != 0 && i <= c->sub_size) { size_t k = (size_t)p * c->sub_size + i - 1; v = rgb5_as_u16(col[3 * k], col[3 * k + 1], col[3 * k + 2]); }
            if (p || i) s += ',';
            s += std::to_string(v);
        }
    s += "],\"tile_palettes\":[";
    for (uint32_t t = 0; t < wt * ht; t++) { if (t) s += ','; s += std::to_string((unsigned)tp[t]); }
    s += "],\"tiles\":[";
    for (uint32_t ty = 0; ty < ht; ty++)
        for (uint32_t tx = 0; tx < wt; tx++) {
            if (ty || tx) s += ',';
            s += '[';
            for (uint32_t y = 0; y < 8; y++)
                for (uint32_t x = 0; x < 8; x++) {
                    size_t px = (size_t)(ty * 8 + y) * c->W + (tx * 8 + x);
                    unsigned v = orig[4 * px + 3] == 0 ? 0u : (unsigned)(uint8_t)(map[px] + 1);
                    if (x || y) s += ',';
                    s += std::to_string(v);
                }
            s += ']';
        }
    s += "]}";
    int64_t need = (int64_t)s.size() + 1;
    if (out && cap > 0) { int64_t m = cap - 1 < (int64_t)s.size() ? cap - 1 : (int64_t)s.size(); memcpy(out, s.data(), (size_t)m); out[m] = 0; }
    return need;
}

void snesimage_random_candidates(uint64_t seed, uint64_t step_id, uint32_t n, uint8_t *rgb5) {
    const uint64_t key = mix64(seed ^ (step_id * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull);
    for (uint32_t k = 0; k < n; k++) {
        uint64_t z = mix64(key + ((uint64_t)k + 1) * 0x9E3779B97F4A7C15ull);
        rgb5[3 * k] = (uint8_t)(z & 31); rgb5[3 * k + 1] = (uint8_t)((z >> 5) & 31); rgb5[3 * k + 2] = (uint8_t)((z >> 10) & 31);
    }
}

// lib.rs:890, 917-932
void snesimage_schedule_next(uint32_t sub_count, uint32_t sub_size, int32_t nes, uint32_t *palette, uint32_t *index, uint32_t *channel, uint32_t *step, uint32_t *method) {
    const bool random = (*step % 5) < 4;
    if (method) *method = nes ? SNES_METHOD_NES : (random ? SNES_METHOD_RANDOM : SNES_METHOD_CHANNEL);
    *channel += 1;
    if (*channel == 3 || random) {
        *channel = 0; *index += 1;
        if (*index == sub_size) { *index = 0; *palette += 1; if (*palette == sub_count) { *palette = 0; *step += 1; } }
    }
}

int32_t snesimage_initialize_tiles(snesimage_ctx *c) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    return kmeans_initialize_tiles(c);
}
int32_t snesimage_recalculate_palettes(snesimage_ctx *c) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    return kmeans_recalculate_palettes(c);
}

// Not in the reference (TODO.md:36-37 names it as missing): move every tile to the subpalette that reproduces it best.
int32_t snesimage_reassign_tiles(snesimage_ctx *c, uint32_t *moved_out) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    if (c->pend || c->win_pend) return fail(SNES_ERR_STATE, "a split-phase step is pending: commit it first (its candidates were scored for the current tile assignment)");
    CHECK(set_device(c));
    CHECK(ensure_map(c)); // an optimize() still owed belongs to the state before this change
    CHECK(ensure_tables(c));
    if (c->perceptual) CHECK(ensure_source(c));
    const int ntile = (int)((c->W / 8) * (c->H / 8)), n = ntile * (int)c->sub_count;
    if (!c->d_tile_cost) {
        HIPCHK(dmalloc(&c->d_tile_cost, sizeof(double) * n));
        HIPCHK(dmalloc(&c->d_tile_any, sizeof(int) * ntile));
        HIPCHK(dmalloc(&c->d_tile_moved, sizeof(unsigned int)));
    }
    unsigned int moved = 0;
    HIPCHK(hipMemsetAsync(c->d_tile_moved, 0, sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL(k_tile_costs, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_orig, c->d_pal_rgb8, c->d_pal_lab, c->d_labpx, (int)c->W, (int)c->H, (int)c->sub_count,
                       (int)c->sub_size, c->perceptual ? 1 : 0, c->d_tile_cost, c->d_tile_any);
    hipLaunchKernelGGL(k_tile_move, dim3((ntile + 255) / 256), dim3(256), 0, c->stream, c->d_tile_cost, c->d_tile_any, ntile, (int)c->sub_count, c->d_tile_pal, c->d_tile_moved);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&moved, c->d_tile_moved, sizeof(moved), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (moved) { // as after snesimage_set_tile_palettes, then optimize() (the palettes are kept)
        c->pack_valid = false; c->inc_valid = false; c->map_synced = false; c->epoch++; c->epoch_by_commit = false;
        CHECK(do_optimize(c));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (moved_out) *moved_out = moved;
    return SNES_OK;
}

int32_t snesimage_timing_enable(snesimage_ctx *c, int32_t on) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    CHECK(drain_timing(c));
    c->timing = on == 2 ? 2 : (on != 0 ? 1 : 0); c->t_ms[0] = c->t_ms[1] = c->t_ms[2] = 0.0; c->t_launches = 0; c->t_cands = 0;
    return SNES_OK;
}
int32_t snesimage_timing_read(snesimage_ctx *c, double *ms3, uint64_t *launches, uint64_t *candidates) {
    if (!c) return fail(SNES_ERR_ARG, "null context");
    CHECK(set_device(c));
    CHECK(drain_timing(c));
    if (ms3) { ms3[0] = c->t_ms[0]; ms3[1] = c->t_ms[1]; ms3[2] = c->t_ms[2]; }
    if (launches) *launches = c->t_launches;
    if (candidates) *candidates = c->t_cands;
    return SNES_OK;
}

// test hook: the (n+1)-th workspace allocation from now on fails with hipErrorOutOfMemory (n < 0: off)
void snesimage_debug_fail_alloc(int32_t n) { g_fail_alloc_in.store(n); }

int32_t snesimage_debug_math(int32_t device, int32_t op, const float *x, const float *y, uint32_t n, float *out) {
    if (!x || !out || n == 0) return fail(SNES_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(device));
    const uint32_t per_in = (op == 5 || op == 6) ? 3 : 1, per_out = (op == 6) ? 3 : 1;
    float *dx = nullptr, *dy = nullptr, *dout = nullptr, *dlut = nullptr;
    HIPCHK(hipMalloc(&dx, sizeof(float) * n * per_in));
    HIPCHK(hipMalloc(&dy, sizeof(float) * n * per_in));
    HIPCHK(hipMalloc(&dout, sizeof(float) * n * per_out));
    HIPCHK(hipMalloc(&dlut, sizeof(float) * 256));
    float e1[256], e2[256];
    make_eotf_tables(e1, e2);
    HIPCHK(hipMemcpy(dlut, e2, sizeof(e2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dx, x, sizeof(float) * n * per_in, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy, y ? y : x, sizeof(float) * n * per_in, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, 0, op, dx, dy, (int)n, dlut, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, sizeof(float) * n * per_out, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout); (void)hipFree(dlut);
    return SNES_OK;
}

} // extern "C"

#include "batch_host.inc"
#include "window_host.inc"
#include "group_host.inc"
