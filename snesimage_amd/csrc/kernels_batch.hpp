// snesimage_amd/csrc/kernels_batch.hpp — one launch over many images (throughput mode).
// The kernels of an optimizer call are the single-image bodies unchanged; blockIdx.z picks the image, whose
// arguments sit in a device array instead of the kernel's own argument block.  Every image of a batch has the same
// geometry and the same number of candidates, so one grid shape serves them all; nothing is shared between images.
#pragma once
#include "kernels_sparse2.hpp"

namespace snes {

struct BatchArgs {
    SparseParams Pc, Pb; // the call's candidates / the base image B of the call's slot
    PrepParams prep;     // pack of the slot (mode 2)
    unsigned long long key; const uint8_t *colors_in; uint8_t *colors; uint8_t *cand; const float *eotf; float *cand_tab;
    const double *part; double *errors; double *inc_err; StepResult *last; PaletteTables T;
    int method, n, slot, channel, nes, npx;
    const float *lab_eotf; float *cand_lab; // --perceptual-palettes
    // --dither (slot windows): Floyd-Steinberg of the call's base image B (k_dither4 MODE 1: the slot's entry stands in for
    // entry j0 of its subpalette, colour bcolor -> table row btab) and the candidates' resumed runs (MODE 2)
    DitherParams Db, Dc; const unsigned long long *win_pack; const uint8_t *bcolor; float *btab; int *zero; int nzero; uint8_t *map;
    int dead_base; // slot windows: see SNES_BATCH_IMG (the launch's first member carries it)
};

// `dead` (optional): the sequence number of the last slot window whose commit changed the palette.  A window is built on the
// assumption that the windows in flight before it — numbers dead_base and up — accept nothing: once *dead >= dead_base that is
// false, the window is void and its launches leave at once.  Numbers only grow, so the word is never reset (until round 4 it was a
// flag, cleared by a memset in front of every window that followed an acceptance: ~20 us of host and queue time per window).
#define SNES_BATCH_IMG if (dead && *dead >= A[0].dead_base) return; const BatchArgs &a = A[blockIdx.z]
// XCD-aware block -> (image, x, y) mapping for the heavy stages.  Blocks are dealt round-robin over the chip's 8 XCDs, each
// with its own L2; with blockIdx.z = image every XCD would see every image and fetch its base image, checkpoints and source
// planes (~20 MB per image) once per XCD (PMC: 2.25 MB fetched per candidate in the V pass against 0.37 MB in single-image
// mode).  Here the blocks whose linear index is congruent modulo 8 — one XCD under round-robin placement — take the
// images congruent to that residue: an image's blocks share one L2.  Speed only: any placement gives the same results.
struct BatchBlock { int img, x, y; };
__device__ __forceinline__ BatchBlock batch_block() {
    BatchBlock b;
    const int X = (int)gridDim.x, Y = (int)gridDim.y, K = (int)gridDim.z;
    const int per = X * Y, K8 = K & ~7; // the members beyond the last multiple of eight (a window of channel calls rarely holds a multiple) are placed plainly
    const int L = (int)blockIdx.x + X * ((int)blockIdx.y + Y * (int)blockIdx.z);
    int rem;
    if (L < per * K8) { const int idx = L >> 3; b.img = (L & 7) + 8 * (idx / per); rem = idx % per; }
    else { const int L2 = L - per * K8; b.img = K8 + L2 / per; rem = L2 % per; }
    b.x = rem % X; b.y = rem / X;
    return b;
}
#define SNES_BATCH_XCD if (dead && *dead >= A[0].dead_base) return; const BatchBlock bb = batch_block(); const BatchArgs &a = A[bb.img]
__global__ void kb_gen_candidates(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; gen_candidates_body(a.method, a.n, a.key, a.colors_in, a.slot, a.channel, a.cand, 0, 1, nullptr, nullptr); }
__global__ __launch_bounds__(256) void kb_prep(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; prep_body(a.prep); }
__global__ __launch_bounds__(256) void kb_build_plist(const BatchArgs *__restrict__ A, const int *__restrict__ dead) {
    SNES_BATCH_IMG;
    build_plist_body(a.win_pack ? a.win_pack : a.Pb.pack, a.npx, const_cast<uint4 *>(a.Pb.plist), const_cast<int *>(a.Pb.plist_count)); // (--dither: B's record of targets and keys to beat)
}
// --dither: what k_prep does for the other paths — B's item counters and the contested-pixel count cleared — and B's row of the candidate table
__global__ void kb_dither_prep(const BatchArgs *__restrict__ A, const int *__restrict__ dead) {
    SNES_BATCH_IMG;
    if ((int)threadIdx.x < a.nzero) a.zero[threadIdx.x] = 0;
    candidate_tables_body(a.bcolor, 1, a.eotf, a.btab);
    if (a.Db.perceptual) candidate_lab_body(a.btab, 1, a.lab_eotf, const_cast<float *>(a.Db.cand_lab)); // (the same thread wrote the table's row)
}
template <int SUB> __global__ __launch_bounds__(512) void kb_dither_base(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; dither4_body<SUB, 1>(a.Db, 0); }
__global__ __launch_bounds__(512) void kb_dither_base_lab(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; dither4_body<0, 1, true>(a.Db, 0); }
__global__ __launch_bounds__(256) void kb_dither_first_lab(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; if ((int)blockIdx.x < a.n) dither_first_lab_body(a.Pc); }
__global__ __launch_bounds__(512) void kb_dither_run4_lab(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; if ((int)blockIdx.x < a.n) dither4_body<0, 2, true>(a.Dc, (int)blockIdx.x); }
__global__ __launch_bounds__(128) void kb_dither_run_lab(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; if ((int)blockIdx.x < a.n) dither_body<true, 0, 2>(a.Dc, (int)blockIdx.x); }
__global__ __launch_bounds__(1024) void kb_dither_first(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; dither_first_body(a.Pc); }
template <int SUB> __global__ __launch_bounds__(512) void kb_dither_run4(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; if ((int)blockIdx.x < a.n) dither4_body<SUB, 2>(a.Dc, (int)blockIdx.x); }
template <int SUB> __global__ __launch_bounds__(128) void kb_dither_run(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; if ((int)blockIdx.x < a.n) dither_body<false, SUB, 2>(a.Dc, (int)blockIdx.x); }
template <int SUB> __global__ __launch_bounds__(256) void kb_dither_runw(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; ditherw_body<SUB>(a.Dc, (int)blockIdx.x, a.n); }
__global__ __launch_bounds__(1024) void kb_dither_diff(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; dither_diff_body(a.Pc); }
__global__ __launch_bounds__(1024) void kb_sparse_scan(const BatchArgs *__restrict__ A, int base, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_scan_body<1>(base ? a.Pb : a.Pc); }
__global__ __launch_bounds__(1024) void kb_sparse_scan4(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_scan_body<4>(a.Pc); } // four waves per candidate
__global__ __launch_bounds__(256) void kb_base_down(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; base_down_body(a.Pb); }
__global__ __launch_bounds__(256) void kb_sparse_down(const BatchArgs *__restrict__ A, int only_scale, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_down_body(a.Pc, only_scale, bb.x); }
__global__ __launch_bounds__(256) void kb_sparse_down_tiles(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_down_tiles_body<SNES_DOWN_TILES_U>(a.Pc, bb.x, (int)gridDim.x); }
__global__ __launch_bounds__(256) void kb_sparse_down1(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_down1_body(a.Pc, bb.x, (int)gridDim.x); }
__global__ void kb_candidate_tables(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; candidate_tables_body(a.cand, a.n, a.eotf, a.cand_tab); }
__global__ void kb_candidate_lab(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; candidate_lab_body(a.cand_tab, a.n, a.lab_eotf, a.cand_lab); }
__global__ __launch_bounds__(256) void kb_clear_bitmaps(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { // the won-pixel bitmaps of the call's candidates (npx/32 words each)
    SNES_BATCH_IMG;
    const size_t words = (size_t)a.n * (a.npx / 32);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) a.Pc.bitmap[i] = 0u;
}
__global__ __launch_bounds__(256) void kb_sparse_scan_lab(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_scan_lab_body(a.Pc); }
__global__ __launch_bounds__(64) void kb_sparse_h(const BatchArgs *__restrict__ A, int base, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_h_body(base ? a.Pb : a.Pc); }
__global__ __launch_bounds__(64) void kb_sparse_h2(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_h2_dispatch<false>(a.Pc, bb.y, bb.x, (int)gridDim.x); }
__global__ __launch_bounds__(64) void kb_sparse_h2_base(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_h2_dispatch<true>(a.Pb, bb.y, bb.x, (int)gridDim.x); }
// the quad-per-row H pass (a third of the chain per column quad for four times the waves): short windows, whose stages are latency
// (the lists from list0 on / the wide scales from s0 on: a window's members on two streams, as score_list does for long lists)
__global__ __launch_bounds__(64) void kb_sparse_h2_lists(const BatchArgs *__restrict__ A, const int *__restrict__ dead, int list0) { SNES_BATCH_XCD; sparse_h2_dispatch<false>(a.Pc, list0 + bb.y, bb.x, (int)gridDim.x); }
__global__ __launch_bounds__(256, 5) void kb_sparse_v2_scales(const BatchArgs *__restrict__ A, const int *__restrict__ dead, int s0) { SNES_BATCH_XCD; const int s = s0 + bb.y; if (s < a.Pc.G.nscales && a.Pc.G.sw[s] >= 64) sparse_v2_body<false>(a.Pc, s, bb.x); }
__global__ __launch_bounds__(64) void kb_sparse_h2q(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_h2q_dispatch<false>(a.Pc, bb.y, bb.x, (int)gridDim.x); }
__global__ __launch_bounds__(64) void kb_sparse_h2q_base(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; sparse_h2q_dispatch<true>(a.Pb, bb.y, bb.x, (int)gridDim.x); }
__global__ __launch_bounds__(256, 5) void kb_sparse_v2(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; if (bb.y < a.Pc.G.nscales && a.Pc.G.sw[bb.y] >= 64) sparse_v2_body<false>(a.Pc, bb.y, bb.x); }
__global__ __launch_bounds__(256) void kb_sparse_v_base(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_XCD; if (bb.y < a.Pb.G.nscales && a.Pb.G.sw[bb.y] >= 64) sparse_v2_body<true>(a.Pb, bb.y, bb.x); }
// B's wide V sweep on two waves per 64 columns (sparse_v2_base_split_body): grid.x = column block, grid.y = channel + 3 * scale, grid.z = member
__global__ __launch_bounds__(128) void kb_sparse_v_base_split(const BatchArgs *__restrict__ A, const int *__restrict__ dead) {
    SNES_BATCH_IMG;
    const int s = (int)blockIdx.y / 3, ch = (int)blockIdx.y - 3 * s;
    if (s < a.Pb.G.nscales && a.Pb.G.sw[s] >= 64 && (int)blockIdx.x < (a.Pb.G.sw[s] >> 6)) sparse_v2_base_split_body(a.Pb, s, ch, (int)blockIdx.x);
}
__global__ __launch_bounds__(256, 1) void kb_sparse_v_base_narrow(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_v_base_narrow_dispatch(a.Pb); }
__global__ __launch_bounds__(1024) void kb_sparse_order(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_order_body(a.Pc, const_cast<int *>(a.Pc.order)); }
__global__ __launch_bounds__(256, 2) void kb_sparse_v(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; sparse_v_body<false, 2, 2>(a.Pc, (int)blockIdx.y + a.Pc.s_first); }
__global__ __launch_bounds__(256) void kb_final_score(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; final_score_wave_body(a.part, a.n, a.Pc.G, a.errors, 1, 0, a.Pc.item_count, (int)kItemLists); } // grid.x = (n + 3) / 4: a wave per candidate
__global__ __launch_bounds__(256) void kb_commit(const BatchArgs *__restrict__ A, const int *__restrict__ dead) { SNES_BATCH_IMG; commit_body(a.errors, a.n, a.cand, a.colors, a.slot, a.nes, a.inc_err, a.last, a.T); }
// --dither, image batches: lib.rs:237's optimize() of the committed palette is the winner's own resumed run (every candidate of
// an image is scored on this device, so it is always at hand); nothing accepted: the stored map stands
__global__ __launch_bounds__(1024) void kb_take_map(const BatchArgs *__restrict__ A, const int *__restrict__ dead) {
    SNES_BATCH_IMG;
    const int k = a.last->best_k;
    if (k < 0) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.Pc.maps + (size_t)k * a.npx);
    uint4 *dst = reinterpret_cast<uint4 *>(a.map);
    for (int i = threadIdx.x; i < a.npx / 16; i += 1024) dst[i] = src[i];
}
#undef SNES_BATCH_IMG
#undef SNES_BATCH_XCD

// ---- speculative slot window (snesimage_run_slots, window_host.inc) ----------------------------------------------------
// K consecutive calls of the reference's scheduler (lib.rs:888-933) are scored against the SAME palette, one "image" of
// the batched launches per call (a slot context: its own pack, base image B and candidate storage; source pyramid,
// palette and tables shared).  Call j+1 of the sequential loop sees exactly this state iff call j accepted nothing, so
// the calls are committed in order and the window ends at the first one that changes the state: bit-identical to the
// one-call-at-a-time loop, whatever K.
struct WindowSlot { int method, n, slot, channel, nes, member; unsigned long long key; int cand0, pad; }; // member / cand0: where this rank scored the call (argument block, first candidate in its storage); member < 0: another rank's
struct WindowResult { int consumed, accepted; };
constexpr int kMaxWindow = 1024; // calls per window over all ranks

// The candidate lists of all K calls (every rank generates all of them: the commit needs the winner's colour wherever it
// was scored) and the error vector preset to +inf (a rank fills in the calls it owns; the others arrive by min-all-reduce).
__global__ void kw_gen_candidates(const WindowSlot *__restrict__ S, const uint8_t *__restrict__ colors, uint8_t *__restrict__ cand, double *__restrict__ errors, int stride, const int *__restrict__ dead, int dead_base) {
    if (*dead >= dead_base) return;
    const WindowSlot w = S[blockIdx.y];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < stride) errors[(size_t)blockIdx.y * stride + k] = __longlong_as_double(0x7ff0000000000000ll);
    gen_candidates_body(w.method, w.n, w.key, colors, w.slot, w.channel, cand + 3 * (size_t)blockIdx.y * stride, 0, 1, nullptr, nullptr);
}

// In-order commit: per call the first-lowest error (sixteen lanes per call, lexicographic (error, index) minimum: what the
// ascending strict-< scan of lib.rs:216-219 ends on), then one thread applies the calls' decisions in sequence and stops
// behind the first call that changed the state (a candidate accepted; for the NES method, which always takes its table
// argmin, a colour that differs from the current one).  log[j] = what snesimage_last_step would report after call j.
__global__ __launch_bounds__(1024) void kw_commit(const WindowSlot *__restrict__ S, int K, int stride, const double *__restrict__ errors, const uint8_t *__restrict__ cand, uint8_t *__restrict__ colors,
                                                 double *__restrict__ inc_err, StepResult *__restrict__ last, PaletteTables T, WindowResult *__restrict__ res, StepResult *__restrict__ log, int *__restrict__ dead, int dead_base, int seq) {
    if (*dead >= dead_base) { if (threadIdx.x == 0) { res->consumed = 0; res->accepted = 0; } return; } // voided by an earlier window's commit: nothing of this one happened
    __shared__ double s_e[kMaxWindow];
    __shared__ int s_k[kMaxWindow];
    const int l16 = threadIdx.x & 15, q = threadIdx.x >> 4; // sixteen lanes per call: 64 calls at a time (the loads' latency is what this loop costs)
    for (int j = q; j < K; j += 64) {
        const int n = S[j].n;
        double be = __longlong_as_double(0x7ff0000000000000ll); int bk = 0x7fffffff;
        for (int k = l16; k < n; k += 16) { const double e = errors[(size_t)j * stride + k]; if (e < be) { be = e; bk = k; } } // NaN never wins, as in the reference
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const double e2 = __shfl_xor(be, o); const int k2 = __shfl_xor(bk, o);
            if (e2 < be || (e2 == be && k2 < bk)) { be = e2; bk = k2; }
        }
        if (l16 == 0) { s_e[j] = be; s_k[j] = bk; }
    }
    __syncthreads();
    int consumed = 0, accepted = 0;
    if (K > 0 && !S[0].nes) {
        // Every call up to the first acceptance is measured against the same incumbent error, so which call accepts first is a
        // parallel minimum; the calls before it report "nothing accepted" (incumbent error, the slot's colour as it stands) —
        // written by all threads — and only the accepting call is decided by one thread (a sequential walk over the calls, a
        // few dependent global accesses each, cost ~1.2 us per call: 0.6 ms for the 512 calls of an 8-GPU window)
        __shared__ int s_first;
        if (threadIdx.x == 0) s_first = K;
        __syncthreads();
        const double inc = *inc_err;
        for (int j = threadIdx.x; j < K; j += 1024) if (s_k[j] != 0x7fffffff && s_e[j] < inc) atomicMin(&s_first, j);
        __syncthreads();
        const int first = s_first;
        for (int j = threadIdx.x; j < first; j += 1024) {
            const int slot = S[j].slot;
            StepResult r; r.error = inc; r.best_k = -1; r.rgb5[0] = colors[3 * slot]; r.rgb5[1] = colors[3 * slot + 1]; r.rgb5[2] = colors[3 * slot + 2]; r.changed = 0;
            log[j] = r;
        }
        __syncthreads(); // the records above read `colors`; the accepting call's decision below rewrites its slot (an earlier call of the window may be on the same slot)
        if (threadIdx.x != 0) return;
        consumed = first;
        if (first < K) {
            commit_decide(s_e[first], s_k[first], cand + 3 * (size_t)first * stride, colors, S[first].slot, 0, inc_err, log + first, T);
            consumed = first + 1; accepted = 1;
        } else if (K > 0) { // (this thread's own copy of the last record: the one in log[] may be another thread's store)
            const int slot = S[K - 1].slot;
            StepResult r; r.error = inc; r.best_k = -1; r.rgb5[0] = colors[3 * slot]; r.rgb5[1] = colors[3 * slot + 1]; r.rgb5[2] = colors[3 * slot + 2]; r.changed = 0;
            log[K - 1] = r;
        }
    } else {
        if (threadIdx.x != 0) return;
        for (int j = 0; j < K && !accepted; j++) { // the NES method always takes its table argmin and moves the incumbent error with it: in sequence
            commit_decide(s_e[j], s_k[j], cand + 3 * (size_t)j * stride, colors, S[j].slot, S[j].nes, inc_err, log + j, T);
            consumed = j + 1;
            accepted = log[j].changed ? 1 : 0;
        }
    }
    res->consumed = consumed; res->accepted = accepted;
    if (accepted) *dead = seq; // windows already enqueued behind this one were built for the old palette
    if (consumed) *last = log[consumed - 1];
}

// --dither: the committed state's palette_map (lib.rs:237 re-runs optimize() on the winner's palette).  The winner's own
// resumed run IS that map: adopted if its call was scored on this rank (S[j].member: its argument block, S[j].cand0: its first candidate there);
// *skip = 1 then, and also when nothing was accepted (the stored map stands); 0 = the caller has to dither again.
__global__ __launch_bounds__(1024) void kw_take_map(const BatchArgs *__restrict__ A, const WindowSlot *__restrict__ S, const WindowResult *__restrict__ res, const StepResult *__restrict__ log,
                                                   uint8_t *__restrict__ map, int npx, int *__restrict__ skip) {
    const int consumed = res->consumed, accepted = res->accepted;
    int have = 1;
    if (accepted) {
        const int j = consumed - 1, k = log[j].best_k;
        have = S[j].member >= 0 ? 1 : 0;
        if (have) {
            const uint4 *src = reinterpret_cast<const uint4 *>(A[S[j].member].Pc.maps + (size_t)(S[j].cand0 + k) * npx);
            uint4 *dst = reinterpret_cast<uint4 *>(map);
            for (int i = threadIdx.x; i < npx / 16; i += 1024) dst[i] = src[i];
        }
    }
    if (threadIdx.x == 0) *skip = have;
}

} // namespace snes
