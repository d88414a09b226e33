// snesimage_amd/csrc/kernels_sparse2.hpp — the group-sparse H and V passes for the scales at least 64 pixels wide
// (256, 128 and 64 at the BASELINE size: 98.4 % of the pyramid's pixels).  kernels_sparse.hpp explains the method and keeps
// the general bodies, which still serve the narrow scales and the base image's V pass; the bodies here compute the same
// values operation for operation and differ in how the work is cut and how the data moves:
//
//   * A changed row is recomputed from the 64-column block that holds its first changed input, not from column 0.
//     The recurrence is causal along x, so its state on entering column block b is B's as long as nothing left of the
//     block differs: B's H pass leaves that state behind (18 floats per row, channel and block boundary: `ckh`), the
//     scan lists the work items per (scale, first block), and every wave of sparse_h2_body starts all its rows at the
//     same block.  On the BASELINE workload this removes ~40 % of the H pass's steps and of its output traffic.
//   * H output leaves a wave as 128-byte lines (8 columns x 4 rows of one plane, staged through LDS) instead of the
//     64-byte runs one column quad gives; LDS is kept that small (10 KB per wave: 15 waves per CU) because the kernel is a
//     chain of dependent iterations per wave and lives on the number of waves a CU holds.
//   * The H pass also stores the XYB value of every pixel of a changed group in the R4 layout (looked up here at scale 0,
//     re-laid from the downscale's C4 copy at the other scales; B's H pass does scale 0 for the whole of B), so the V pass
//     reads its map input `img2` from a plane at every scale: it no longer touches the pack, the palette table or the
//     won-pixel bitmap, and has one body for all scales.
//   * In sparse_v2_body a wave's 64 columns belong to one (candidate, channel), so the first changed group, the loop
//     counter and the group -> storage choice are wave-uniform: buffer loads with a uniform descriptor, a 32-bit lane
//     offset and the group offset as the scalar offset — the 64-bit per-lane address arithmetic and the pointer selects of
//     the general body were a fifth of its instructions and the source of its register spills.  Pairs run channel by
//     channel, so that what a channel's waves share fits an XCD's L2.
//     A wave reads a changed group from the candidate's own storage only if the group's first block is not to the
//     right of the wave's columns; otherwise the candidate's H pass has not written those columns and they are B's.
#pragma once
#include "kernels_sparse.hpp"
#include "kernels_fast.hpp"

namespace snes {

// recurrence steps, identical to kernels_fast.hpp / kernels_sparse.hpp
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

// ---- H pass of changed groups, from their first changed column block -----------------------------------------------------
// grid.y = item list (scale * kColBuckets + first block); block = one wave = 16 work items (candidate, slot, channel), a
// lane quad = the four rows of an item.  Column quads g run from 16*block to W/4; iteration g consumes the inputs of quad g
// (the "right" taps in[n+4]) and yields the outputs of quad g-1.
// S0: scale 0 — the candidate's pixels come from the pack (win test) instead of an XYB plane.
// bx / gx: the block's index and count among the blocks of its list (blockIdx.x / gridDim.x unless the caller remaps blocks)
//
// The memory pipeline (round 4).  The body is a chain of dependent iterations per wave, and until round 4 every iteration
// waited for its own prefetch: the loads of the next quad sat under `if (g + 1 < G4)`, the stores of a finished 8-column run
// under `if ((g & 1) == 0)` and behind per-lane predicates, so at every wait the compiler had to assume the path on which
// nothing younger than the awaited load had been issued — `s_waitcnt vmcnt(0)` 49 times in the kernel, the wave parked for a
// full memory round trip per column quad (VALUBusy 40 %).  Now every vector-memory instruction of the steady state is
// issued on every path, so the counter values are static:
//   * inputs are fetched TWO quads ahead into a six-slot register ring, unconditionally (the address is clamped to the last
//     quad; a quad beyond the row is zeroed when it is consumed);
//   * the loop is unrolled by six from an odd quad (iteration gs, which stages nothing, is peeled), so ring slots, the
//     staging half and the flush positions are compile-time constants;
//   * the flush reads its eight lines from LDS in one batch and stores them without predicates — a lane with nothing to
//     store (no item, or a plane the launch does not write) aims at a scratch line (`trash`);
//   * a block is ONE wave: LDS operations of a wave execute in order, so the staging needs no barrier, only the compiler
//     kept from reordering (lds_order) — __syncthreads() also waited for every load and store in flight.
// Same values operation for operation as before (sparse == dense, bit for bit: tests/test_gpu_parity.py).
// LDS of one block (= one wave), shared by the two instantiations of the body (a block runs one of them; declared inside the
// template each instantiation would take its own copy and the kernel would hold both: 32 KB per wave, five waves per CU)
struct H2Shared {
    __attribute__((aligned(16))) float out[16 * (4 * 32 + 4)];  // staging: per item [plane][8 columns][4 rows] + pad
    long long hbase[16], xbase[16];                             // per item: float offset of its H output / XYB planes inside P.store, -1 = no item
};
__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } // one-wave blocks: program order is LDS order
// colour indices of the four pixels of a column quad of row y at scale 0 (pa / pb: the quad's pack words — or, with --dither,
// its map and subpalette-base words; bw: the word of the candidate's won-pixel bitmap that holds the quad's four bits)
template <bool BASE>
__device__ __forceinline__ void h2_resolve(const SparseParams &P, const uint4 pa, const uint4 pb, const uint32_t bw, const int g, const uint32_t crgb, uint32_t (&ci)[4]) {
    if (P.use_maps) resolve4_maps(pa.x, pa.y, BASE ? 0xffffffffu : P.slot_ci, (uint32_t)P.ncol, ci);
    else if (P.perceptual && !BASE) { // four consecutive pixels of a row: four consecutive bits (W = 256: a row is eight words)
        const uint32_t b4 = (bw >> ((g & 7) << 2)) & 0xfu;
        ci[0] = (b4 & 1u) ? (uint32_t)P.ncol : (pa.x >> 24); ci[1] = (b4 & 2u) ? (uint32_t)P.ncol : (pa.z >> 24);
        ci[2] = (b4 & 4u) ? (uint32_t)P.ncol : (pb.x >> 24); ci[3] = (b4 & 8u) ? (uint32_t)P.ncol : (pb.z >> 24);
    } else {
        const uint32_t never = BASE ? 0u : 0xffffffffu; // thr & never == 0 for B: it wins nothing
        ci[0] = sparse_ci(pa.x, pa.y & never, crgb, (uint32_t)P.ncol); ci[1] = sparse_ci(pa.z, pa.w & never, crgb, (uint32_t)P.ncol);
        ci[2] = sparse_ci(pb.x, pb.y & never, crgb, (uint32_t)P.ncol); ci[3] = sparse_ci(pb.z, pb.w & never, crgb, (uint32_t)P.ncol);
    }
}
// BASE: the launch is B's own H pass (every item is B's: no win test, and the block checkpoints are written) / the candidates'
template <bool S0, bool BASE>
__device__ __forceinline__ void sparse_h2_body(const SparseParams &P, const int list, const int bx, const int gx, H2Shared &sh) {
    // staged planes: the three H outputs + the XYB input, which the V pass reads in the R4 layout (at scale 0 it is looked up
    // here; at the other scales it is the downscale's output in the C4 layout, and rewriting it from here — whole lines,
    // only from the row's first block on — is cheaper than a second scattered copy from the downscale; B's are written there)
    constexpr int NP = (S0 || !BASE) ? 4 : 3;
    constexpr int PW = 32;                // staged floats per plane: 8 columns x 4 rows = one 128-byte line of the XT4 / R4 layouts
    constexpr int ISTR = NP * PW + 4;     // staging words per item; the pad spreads the quads' rows over the LDS banks
    extern __shared__ float s_lut[]; // scale 0: XYB of the ncol + 2 colour indices, three planes (dynamic: 12 (ncol + 2) bytes — the waves a CU holds are what this kernel lives on)
    const int lstr = P.ncol + 2;
    float *const s_out = sh.out;
    long long *const s_hbase = sh.hbase, *const s_xbase = sh.xbase;
    const Geom &G = P.G;
    const int s = list / kColBuckets, cb = list % kColBuckets;
    const int W = G.sw[s], H = G.sh[s];
    const int lane = threadIdx.x;
    const int count = P.item_count[list];
    if (bx * 16 >= count) return;
    if (S0) {
        for (int i = lane; i < 3 * lstr; i += 64) { const int c = i / lstr, j = i - c * lstr; s_lut[i] = P.pal_xyb[3 * j + c]; }
        lds_order();
    }
    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
    const int G4 = W >> 2, gs = cb << 4; // first quad of the block: a multiple of 16
    const int r = lane & 3;
    const int half = lane >> 5, fl = lane & 31, fp = fl >> 3, fc = fl & 7; // the flush: half-wave = item, lane = (plane, column) of the staged 8-column run
    float *const trash = P.trash + (lane << 2);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = bx * 16; i0 < count; i0 += gx * 16) { // grid-stride over item quads
        const int qi = i0 + (lane >> 2);
        const bool valid = qi < count;
        const unsigned int it = P.items[(size_t)list * P.item_stride + (valid ? qi : i0)];
        const int k = item_k(it), j = item_j(it), ch = item_ch(it);
        const int y = 4 * item_g(it) + r;
        const size_t ns = (size_t)W * H;
        const float cand_v = (S0 && !BASE) ? P.cand_tab[8 * (size_t)k + 3 + ch] : 0.0f;
        const uint32_t crgb = (S0 && !BASE) ? __float_as_uint(P.cand_tab[8 * (size_t)k + 6]) : 0u;
        const float4 *in1 = reinterpret_cast<const float4 *>(P.img1C4 + G.src_off[s] + (size_t)ch * ns) + y;   // C4: + g*H
        const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.packC4) + 2 * (size_t)y : nullptr;            // + g*2H
        const bool um = S0 && P.use_maps;
        const uint32_t *mw = um ? reinterpret_cast<const uint32_t *>(BASE ? P.bmapC4 : P.mapsC4 + (size_t)(k - P.k0) * ns) + y : nullptr; // C4 bytes: word (g*H + y)
        const uint32_t *sw = um ? reinterpret_cast<const uint32_t *>(P.subC4) + y : nullptr;
        // the row's words of the candidate's won-pixel bitmap (--perceptual-palettes); any readable words otherwise: the load is part of every prefetch
        const uint32_t *bmrow = (S0 && P.perceptual && !BASE && !um) ? P.bitmap + (size_t)k * (G.W * G.H / 32) + (size_t)y * (W >> 5) : reinterpret_cast<const uint32_t *>(P.img1C4);
        const float4 *in2 = S0 ? nullptr
                               : reinterpret_cast<const float4 *>(P.store + (size_t)k * P.S.cand_stride + P.S.off_xybC[s] + (size_t)j * 12 * W + (size_t)ch * 4 * W) + r; // + g*4
        lds_order(); // the previous round's flush has read the bases
        if (r == 0) {
            s_hbase[lane >> 2] = valid ? (long long)k * P.S.cand_stride + P.S.off_hout[s] + (long long)j * 36 * W + (long long)(ch * 3) * 4 * W : -1ll;
            s_xbase[lane >> 2] = valid ? (long long)k * P.S.cand_stride + P.S.off_xybR[s] + (long long)j * 12 * W + (long long)ch * 4 * W : -1ll;
        }
        float sa[3][3], sb[3][3];
        // B's state on entering the block: ckh[s][ch][block-1][18][row]
        const float *ck = P.ckh + P.S.off_ckh[s] + ((size_t)(ch * 3 + (cb > 0 ? cb - 1 : 0)) * 18) * H + y;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                sa[p][q] = gs > 0 ? ck[(size_t)(p * 6 + q) * H] : 0.0f;
                sb[p][q] = gs > 0 ? ck[(size_t)(p * 6 + 3 + q) * H] : 0.0f;
            }
        // six-slot ring: quad gs + t lives in slot t mod 6 (the three quads before gs in slots 3, 4, 5); at scale 0 the raw words
        // of a quad wait in pa / pb / bw[t mod 3] until the quad is consumed
        float4 r1[6], r2[6];
        uint4 pa[3], pb[3]; uint32_t bw[3];
#pragma unroll
        for (int a = 0; a < 6; a++) { r1[a] = zero4; r2[a] = zero4; }
#pragma unroll
        for (int a = 0; a < 3; a++) { pa[a] = make_uint4(0, 0, 0, 0); pb[a] = pa[a]; bw[a] = 0u; }
        auto fetch = [&](const int q, float4 &d1, float4 &d2, uint4 &wa, uint4 &wb, uint32_t &wbm) { // every load on every path (0 <= q < G4)
            d1 = in1[(size_t)q * H];
            if (S0) {
                if (um) { wa.x = mw[(size_t)q * H]; wa.y = sw[(size_t)q * H]; } else { wa = pk[(size_t)q * H * 2]; wb = pk[(size_t)q * H * 2 + 1]; }
                wbm = bmrow[q >> 3];
            } else d2 = in2[(size_t)q * 4];
        };
        auto convert = [&](const int g, const uint4 wa, const uint4 wb, const uint32_t wbm, float4 &d2) { // scale 0: XYB of quad g
            uint32_t ci[4];
            h2_resolve<BASE>(P, wa, wb, wbm, g, crgb, ci);
            const float *lut = s_lut + ch * lstr;
            d2.x = ci[0] == (uint32_t)P.ncol ? cand_v : lut[ci[0]]; d2.y = ci[1] == (uint32_t)P.ncol ? cand_v : lut[ci[1]];
            d2.z = ci[2] == (uint32_t)P.ncol ? cand_v : lut[ci[2]]; d2.w = ci[3] == (uint32_t)P.ncol ? cand_v : lut[ci[3]];
        };
        if (gs > 0) { // the three quads before the block refill the ring (the state is B's checkpoint)
            uint4 ta[3], tb[3]; uint32_t tw[3];
#pragma unroll
            for (int a = 0; a < 3; a++) { ta[a] = make_uint4(0, 0, 0, 0); tb[a] = ta[a]; tw[a] = 0u; fetch(gs - 3 + a, r1[3 + a], r2[3 + a], ta[a], tb[a], tw[a]); }
            if (S0) {
#pragma unroll
                for (int a = 0; a < 3; a++) convert(gs - 3 + a, ta[a], tb[a], tw[a], r2[3 + a]);
            }
        }
        fetch(gs, r1[0], r2[0], pa[0], pb[0], bw[0]);
        fetch(gs + 1, r1[1], r2[1], pa[1], pb[1], bw[1]);
        // one iteration: T = (g - gs) mod 6, STAGE: the outputs of quad g-1 go to the staging area, FLUSH: an 8-column run is complete
#define SNES_H2_ITER(T, STAGE, FLUSH)                                                                                         \
        {                                                                                                                     \
            constexpr int sl_ = (T) % 6, sf_ = ((T) + 2) % 6, ua_ = ((T) + 3) % 6, ub_ = ((T) + 4) % 6, ul_ = ((T) + 5) % 6; \
            constexpr int wc_ = (T) % 3, wf_ = ((T) + 2) % 3;                                                                 \
            { const int qf_ = min(g + 2, G4 - 1); fetch(qf_, r1[sf_], r2[sf_], pa[wf_], pb[wf_], bw[wf_]); }                  \
            if (g >= G4) { r1[sl_] = zero4; r2[sl_] = zero4; }                                                                \
            else if (S0) convert(g, pa[wc_], pb[wc_], bw[wc_], r2[sl_]);                                                      \
            if (BASE && g > 0 && (g & 15) == 0 && g < G4 && valid) { /* B: the state on entering block g/16 */               \
                float *co = P.ckh + P.S.off_ckh[s] + ((size_t)(ch * 3 + (g >> 4) - 1) * 18) * H + y;                          \
                _Pragma("unroll") for (int p = 0; p < 3; p++)                                                                 \
                    _Pragma("unroll") for (int q = 0; q < 3; q++) { co[(size_t)(p * 6 + q) * H] = sa[p][q]; co[(size_t)(p * 6 + 3 + q) * H] = sb[p][q]; } \
            }                                                                                                                 \
            const float v1[4] = {r1[sl_].x, r1[sl_].y, r1[sl_].z, r1[sl_].w}, v2[4] = {r2[sl_].x, r2[sl_].y, r2[sl_].z, r2[sl_].w};   \
            const float l1[4] = {r1[ua_].z, r1[ua_].w, r1[ub_].x, r1[ub_].y}, l2[4] = {r2[ua_].z, r2[ua_].w, r2[ub_].x, r2[ub_].y};   \
            float outp[3][4];                                                                                                 \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                   \
                const float s0 = l2[q] + v2[q];                                                                               \
                const float s1 = (l2[q] * l2[q]) + (v2[q] * v2[q]);                                                           \
                const float s2 = (l1[q] * l2[q]) + (v1[q] * v2[q]);                                                           \
                if ((q & 1) == 0) { SNES_HSTEP(s0, sa[0], sb[0], outp[0][q]) SNES_HSTEP(s1, sa[1], sb[1], outp[1][q]) SNES_HSTEP(s2, sa[2], sb[2], outp[2][q]) } \
                else { SNES_HSTEP(s0, sb[0], sa[0], outp[0][q]) SNES_HSTEP(s1, sb[1], sa[1], outp[1][q]) SNES_HSTEP(s2, sb[2], sa[2], outp[2][q]) }             \
            }                                                                                                                 \
            if (STAGE) { /* outputs of quad g-1 (those of quad gs-1 are B's and stay unwritten): stage [plane][column % 8][row] */ \
                float *so = s_out + (lane >> 2) * ISTR + ((((T) + 1) & 1) << 4) + r; /* (g - 1) & 1: gs is even */            \
                _Pragma("unroll") for (int p = 0; p < 3; p++) { so[p * PW + 0] = outp[p][0]; so[p * PW + 4] = outp[p][1]; so[p * PW + 8] = outp[p][2]; so[p * PW + 12] = outp[p][3]; } \
                if (NP == 4) { so[3 * PW + 0] = r2[ul_].x; so[3 * PW + 4] = r2[ul_].y; so[3 * PW + 8] = r2[ul_].z; so[3 * PW + 12] = r2[ul_].w; } \
            }                                                                                                                 \
            if (FLUSH) { /* 8 columns complete: one store instruction per two items = their NP planes as 128-byte lines */    \
                lds_order();                                                                                                  \
                /* float offset of this lane's 16 bytes behind the item's base: plane fp of the H output (XT4: [x/64][x%64][row] = 4 x */ \
                /* floats into the plane), or the XYB plane (R4: 4 x as well) */                                              \
                const uint32_t o_l = ((NP == 4 && fp == 3) ? 0u : (uint32_t)fp * 4u * (uint32_t)W) + (uint32_t)((g - 2) << 4) + (uint32_t)(fc << 2); \
                _Pragma("unroll") for (int mb = 0; mb < 16; mb += 8) {                                                        \
                    float4 fv[4]; long long fb[4];                                                                            \
                    _Pragma("unroll") for (int m = 0; m < 4; m++) {                                                           \
                        const int im_ = mb + 2 * m + half;                                                                    \
                        fv[m] = *reinterpret_cast<const float4 *>(s_out + im_ * ISTR + fl * 4);                               \
                        fb[m] = (NP == 4 && fp == 3) ? s_xbase[im_] : s_hbase[im_];                                           \
                    }                                                                                                         \
                    _Pragma("unroll") for (int m = 0; m < 4; m++) {                                                           \
                        float *dst = (fb[m] >= 0 && fl < NP * 8) ? P.store + fb[m] + o_l : trash;                             \
                        *reinterpret_cast<float4 *>(dst) = fv[m];                                                             \
                    }                                                                                                         \
                }                                                                                                             \
                asm volatile("" ::: "memory"); /* the staging area is rewritten from the next iteration on */                \
            }                                                                                                                 \
        }
        { const int g = gs; SNES_H2_ITER(0, false, false) }
        for (int g0 = gs + 1; g0 <= G4; g0 += 6) {
            { const int g = g0; SNES_H2_ITER(1, true, false) }
            { const int g = g0 + 1; if (g > G4) break; SNES_H2_ITER(2, true, true) }
            { const int g = g0 + 2; if (g > G4) break; SNES_H2_ITER(3, true, false) }
            { const int g = g0 + 3; if (g > G4) break; SNES_H2_ITER(4, true, true) }
            { const int g = g0 + 4; if (g > G4) break; SNES_H2_ITER(5, true, false) }
            { const int g = g0 + 5; if (g > G4) break; SNES_H2_ITER(6, true, true) }
        }
#undef SNES_H2_ITER
    }
}

// ---- the same H pass with a quad of lanes per row (short lists, and B: a single image is one) --------------------------
// sparse_h2_body runs the three planes of a row one after the other in one lane: ~300 instructions per column quad in one
// chain, and a list that does not fill the chip is bound by the length of that chain (a wave alone issues an instruction
// every ~2.6 ns).  Here lane q < 3 of a quad carries plane q of the row (its own six state floats) and lane 3 stages the XYB
// plane; the inputs are fetched by all four (one address: one request).  Same values operation for operation.  A wave holds
// four items instead of sixteen: four times the waves for the same list, which is why long lists keep the other body.
// The memory pipeline is sparse_h2_body's: every load and store of the steady state on every path.
template <bool S0, bool BASE>
__device__ __forceinline__ void sparse_h2q_body(const SparseParams &P, const int list, const int bx, const int gx, H2Shared &sh) {
    constexpr int NP = (S0 || !BASE) ? 4 : 3, PW = 32, ISTR = 4 * PW + 4; // (the XYB planes of B above scale 0 come from its downscale)
    extern __shared__ float s_lut[];
    const int lstr = P.ncol + 2;
    float *const s_out = sh.out;
    long long *const s_hbase = sh.hbase, *const s_xbase = sh.xbase;
    const Geom &G = P.G;
    const int s = list / kColBuckets, cb = list % kColBuckets;
    const int W = G.sw[s], H = G.sh[s];
    const int lane = threadIdx.x;
    const int count = P.item_count[list];
    if (bx * 4 >= count) return;
    if (S0) {
        for (int i = lane; i < 3 * lstr; i += 64) { const int c = i / lstr, j = i - c * lstr; s_lut[i] = P.pal_xyb[3 * j + c]; }
        lds_order();
    }
    const int q = lane & 3, r = (lane >> 2) & 3, im = lane >> 4; // plane (3: the XYB plane), row of the group, item of the wave
    const int pl = q < 3 ? q : 0;
    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
    const int G4 = W >> 2, gs = cb << 4;
    const int half = lane >> 5, fl = lane & 31, fp = fl >> 3, fc = fl & 7;
    float *const trash = P.trash + (lane << 2);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = bx * 4; i0 < count; i0 += gx * 4) {
        const int qi = i0 + im;
        const bool valid = qi < count;
        const unsigned int it = P.items[(size_t)list * P.item_stride + (valid ? qi : i0)];
        const int k = item_k(it), j = item_j(it), ch = item_ch(it);
        const int y = 4 * item_g(it) + r;
        const size_t ns = (size_t)W * H;
        const float cand_v = (S0 && !BASE) ? P.cand_tab[8 * (size_t)k + 3 + ch] : 0.0f;
        const uint32_t crgb = (S0 && !BASE) ? __float_as_uint(P.cand_tab[8 * (size_t)k + 6]) : 0u;
        const float4 *in1 = reinterpret_cast<const float4 *>(P.img1C4 + G.src_off[s] + (size_t)ch * ns) + y;
        const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.packC4) + 2 * (size_t)y : nullptr;
        const bool um = S0 && P.use_maps;
        const uint32_t *mw = um ? reinterpret_cast<const uint32_t *>(BASE ? P.bmapC4 : P.mapsC4 + (size_t)(k - P.k0) * ns) + y : nullptr;
        const uint32_t *sw = um ? reinterpret_cast<const uint32_t *>(P.subC4) + y : nullptr;
        const uint32_t *bmrow = (S0 && P.perceptual && !BASE && !um) ? P.bitmap + (size_t)k * (G.W * G.H / 32) + (size_t)y * (W >> 5) : reinterpret_cast<const uint32_t *>(P.img1C4);
        const float4 *in2 = S0 ? nullptr
                               : reinterpret_cast<const float4 *>(P.store + (size_t)k * P.S.cand_stride + P.S.off_xybC[s] + (size_t)j * 12 * W + (size_t)ch * 4 * W) + r;
        lds_order(); // the previous round's flush has read the bases
        if ((lane & 15) == 0) {
            s_hbase[im] = valid ? (long long)k * P.S.cand_stride + P.S.off_hout[s] + (long long)j * 36 * W + (long long)(ch * 3) * 4 * W : -1ll;
            s_xbase[im] = valid ? (long long)k * P.S.cand_stride + P.S.off_xybR[s] + (long long)j * 12 * W + (long long)ch * 4 * W : -1ll;
        }
        float sa[3], sb[3]; // this lane's plane
        const float *ck = P.ckh + P.S.off_ckh[s] + ((size_t)(ch * 3 + (cb > 0 ? cb - 1 : 0)) * 18) * H + y;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            sa[e] = gs > 0 ? ck[(size_t)(pl * 6 + e) * H] : 0.0f;
            sb[e] = gs > 0 ? ck[(size_t)(pl * 6 + 3 + e) * H] : 0.0f;
        }
        float4 r1[6], r2[6];
        uint4 pa[3], pb[3]; uint32_t bw[3];
#pragma unroll
        for (int a = 0; a < 6; a++) { r1[a] = zero4; r2[a] = zero4; }
#pragma unroll
        for (int a = 0; a < 3; a++) { pa[a] = make_uint4(0, 0, 0, 0); pb[a] = pa[a]; bw[a] = 0u; }
        auto fetch = [&](const int qq, float4 &d1, float4 &d2, uint4 &wa, uint4 &wb, uint32_t &wbm) {
            d1 = in1[(size_t)qq * H];
            if (S0) {
                if (um) { wa.x = mw[(size_t)qq * H]; wa.y = sw[(size_t)qq * H]; } else { wa = pk[(size_t)qq * H * 2]; wb = pk[(size_t)qq * H * 2 + 1]; }
                wbm = bmrow[qq >> 3];
            } else d2 = in2[(size_t)qq * 4];
        };
        auto convert = [&](const int g, const uint4 wa, const uint4 wb, const uint32_t wbm, float4 &d2) {
            uint32_t ci[4];
            h2_resolve<BASE>(P, wa, wb, wbm, g, crgb, ci);
            const float *lut = s_lut + ch * lstr;
            d2.x = ci[0] == (uint32_t)P.ncol ? cand_v : lut[ci[0]]; d2.y = ci[1] == (uint32_t)P.ncol ? cand_v : lut[ci[1]];
            d2.z = ci[2] == (uint32_t)P.ncol ? cand_v : lut[ci[2]]; d2.w = ci[3] == (uint32_t)P.ncol ? cand_v : lut[ci[3]];
        };
        if (gs > 0) {
            uint4 ta[3], tb[3]; uint32_t tw[3];
#pragma unroll
            for (int a = 0; a < 3; a++) { ta[a] = make_uint4(0, 0, 0, 0); tb[a] = ta[a]; tw[a] = 0u; fetch(gs - 3 + a, r1[3 + a], r2[3 + a], ta[a], tb[a], tw[a]); }
            if (S0) {
#pragma unroll
                for (int a = 0; a < 3; a++) convert(gs - 3 + a, ta[a], tb[a], tw[a], r2[3 + a]);
            }
        }
        fetch(gs, r1[0], r2[0], pa[0], pb[0], bw[0]);
        fetch(gs + 1, r1[1], r2[1], pa[1], pb[1], bw[1]);
#define SNES_H2Q_ITER(T, STAGE, FLUSH)                                                                                        \
        {                                                                                                                     \
            constexpr int sl_ = (T) % 6, sf_ = ((T) + 2) % 6, ua_ = ((T) + 3) % 6, ub_ = ((T) + 4) % 6, ul_ = ((T) + 5) % 6; \
            constexpr int wc_ = (T) % 3, wf_ = ((T) + 2) % 3;                                                                 \
            { const int qf_ = min(g + 2, G4 - 1); fetch(qf_, r1[sf_], r2[sf_], pa[wf_], pb[wf_], bw[wf_]); }                  \
            if (g >= G4) { r1[sl_] = zero4; r2[sl_] = zero4; }                                                                \
            else if (S0) convert(g, pa[wc_], pb[wc_], bw[wc_], r2[sl_]);                                                      \
            if (BASE && q < 3 && g > 0 && (g & 15) == 0 && g < G4 && valid) { /* B: the state of this plane on entering block g/16 */ \
                float *co = P.ckh + P.S.off_ckh[s] + ((size_t)(ch * 3 + (g >> 4) - 1) * 18) * H + y;                          \
                _Pragma("unroll") for (int e = 0; e < 3; e++) { co[(size_t)(pl * 6 + e) * H] = sa[e]; co[(size_t)(pl * 6 + 3 + e) * H] = sb[e]; } \
            }                                                                                                                 \
            const float v1[4] = {r1[sl_].x, r1[sl_].y, r1[sl_].z, r1[sl_].w}, v2[4] = {r2[sl_].x, r2[sl_].y, r2[sl_].z, r2[sl_].w};   \
            const float l1[4] = {r1[ua_].z, r1[ua_].w, r1[ub_].x, r1[ub_].y}, l2[4] = {r2[ua_].z, r2[ua_].w, r2[ub_].x, r2[ub_].y};   \
            float outp[4];                                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 4; c++) {                                                                   \
                /* this lane's plane: mu2 <- img2, s22 <- img2^2, s12 <- img1 * img2 (the same expressions as sparse_h2_body) */ \
                const float s0 = l2[c] + v2[c];                                                                               \
                const float s1 = (l2[c] * l2[c]) + (v2[c] * v2[c]);                                                           \
                const float s2 = (l1[c] * l2[c]) + (v1[c] * v2[c]);                                                           \
                const float sum = q == 0 ? s0 : (q == 1 ? s1 : s2);                                                           \
                if ((c & 1) == 0) { SNES_HSTEP(sum, sa, sb, outp[c]) } else { SNES_HSTEP(sum, sb, sa, outp[c]) }              \
            }                                                                                                                 \
            if (STAGE) {                                                                                                      \
                float *so = s_out + im * ISTR + ((((T) + 1) & 1) << 4) + r + q * PW;                                          \
                if (q < 3) { so[0] = outp[0]; so[4] = outp[1]; so[8] = outp[2]; so[12] = outp[3]; }                           \
                else if (NP == 4) { so[0] = r2[ul_].x; so[4] = r2[ul_].y; so[8] = r2[ul_].z; so[12] = r2[ul_].w; }            \
            }                                                                                                                 \
            if (FLUSH) {                                                                                                      \
                lds_order();                                                                                                  \
                const uint32_t o_l = ((fp == 3) ? 0u : (uint32_t)fp * 4u * (uint32_t)W) + (uint32_t)((g - 2) << 4) + (uint32_t)(fc << 2); \
                float4 fv[2]; long long fb[2];                                                                                \
                _Pragma("unroll") for (int m = 0; m < 2; m++) {                                                               \
                    const int im_ = 2 * m + half;                                                                             \
                    fv[m] = *reinterpret_cast<const float4 *>(s_out + im_ * ISTR + fl * 4);                                   \
                    fb[m] = (fp == 3) ? s_xbase[im_] : s_hbase[im_];                                                          \
                }                                                                                                             \
                _Pragma("unroll") for (int m = 0; m < 2; m++) {                                                               \
                    float *dst = (fb[m] >= 0 && fl < NP * 8) ? P.store + fb[m] + o_l : trash;                                 \
                    *reinterpret_cast<float4 *>(dst) = fv[m];                                                                 \
                }                                                                                                             \
                asm volatile("" ::: "memory");                                                                               \
            }                                                                                                                 \
        }
        { const int g = gs; SNES_H2Q_ITER(0, false, false) }
        for (int g0 = gs + 1; g0 <= G4; g0 += 6) {
            { const int g = g0; SNES_H2Q_ITER(1, true, false) }
            { const int g = g0 + 1; if (g > G4) break; SNES_H2Q_ITER(2, true, true) }
            { const int g = g0 + 2; if (g > G4) break; SNES_H2Q_ITER(3, true, false) }
            { const int g = g0 + 3; if (g > G4) break; SNES_H2Q_ITER(4, true, true) }
            { const int g = g0 + 4; if (g > G4) break; SNES_H2Q_ITER(5, true, false) }
            { const int g = g0 + 5; if (g > G4) break; SNES_H2Q_ITER(6, true, true) }
        }
#undef SNES_H2Q_ITER
    }
}
template <bool BASE>
__device__ __forceinline__ void sparse_h2q_dispatch(const SparseParams &P, const int list, const int bx, const int gx) {
    const int s = list / kColBuckets;
    if (s >= P.G.nscales || P.G.sw[s] < 64 || (list % kColBuckets) >= (P.G.sw[s] >> 6)) return;
    __shared__ H2Shared sh;
    if (s == 0) sparse_h2q_body<true, BASE>(P, list, bx, gx, sh); else sparse_h2q_body<false, BASE>(P, list, bx, gx, sh);
}

// ---- V pass + maps of one (candidate, channel) per 64-column wave, resumed from B's checkpoint ------------------------------
// Same checkpoint records, tail ring and pooling as sparse_v_body (kernels_sparse.hpp); W >= 64 only.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T ld_at(const void *sbase, uint32_t voff) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(sbase) + voff); }
// Buffer loads: a wave-uniform descriptor (four SGPRs), a 32-bit lane offset and a scalar byte offset.  Written as plain
// pointer arithmetic the compiler re-associates (uniform base + group offset) + lane offset into (base + lane offset) + group
// offset: a 64-bit VGPR pair per array and a 64-bit vector add per load.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) { // p must be wave-uniform; the readfirstlanes make that provable (no waterfall loops around the loads)
    const unsigned long long a = (unsigned long long)p;
    const unsigned long long u = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(u), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ double2 buf_ld2d(rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}

// BASE: the image B itself — every group is its own, the sweep starts at the top with a zero state and leaves the
// checkpoint records behind (record g: state before group iteration g and pooling sums of rows < 4g-4; record H/4+1: final sums).
template <bool BASE>
__device__ __forceinline__ void sparse_v2_body(const SparseParams &P, const int s, const int bx) { // bx: blockIdx.x unless the caller remaps blocks
    __shared__ short s_slot[4][64]; // per wave: group -> slot in the candidate's storage, -1 = B's group
    // tails: xy halves two deep, zw halves three deep, [slot][plane][thread]; the pooling reduction reuses the space
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[(2 + 3) * 3 * 256 * sizeof(float2)];
    double (*red)[6] = reinterpret_cast<double (*)[6]>(s_raw);
    static_assert(sizeof(s_raw) >= 256 * 6 * sizeof(double), "reduction scratch must fit the tail ring");
    const Geom &G = P.G;
    // (the geometry arrays are indexed dynamically: without the readfirstlanes their values count as divergent)
    const int W = uni(G.sw[s]), H = uni(G.sh[s]), H4 = H >> 2;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = uni(t >> 6);
    const int wpp = W >> 6, ppw = 4 / wpp; // waves per pair, pairs per block
    const int npairs = BASE ? 3 : P.ncand * 3;
    if (bx * ppw >= npairs) return;
    const int ql = wv / wpp, xw = (wv - ql * wpp) << 6; // pair of the wave inside the block, first column of the wave
    const int pair_raw = bx * ppw + ql;
    const bool active = pair_raw < npairs;
    const int pair = active ? pair_raw : 0;
    // pairs are taken channel by channel: what the waves of a channel share (B's planes and the source's, ~2.4 MB at scale 0)
    // then fits an XCD's L2, which the three channels' together do not
    const int ch = BASE ? pair : pair / P.ncand, ci = BASE ? 0 : pair - ch * P.ncand;
    const int k = BASE ? P.base : uni(P.k0 + (P.order ? P.order[ci] : ci));
    const CandMeta *M = P.meta + k;
    // Effective slot of every group for this wave: the candidate's own rows only where its H pass has written this wave's
    // columns (a group whose first changed block lies to the right leaves these columns as B has them).  The wave resumes
    // from B's checkpoint at ITS first such group — up to there every input of its columns is B's — and takes B's final
    // sums if there is none.
    int gs;
    {
        const int g = lane;
        short sl = -1;
        if (!BASE && g < H4) { sl = M->gslot[P.S.goff[s] + g]; if (sl >= 0 && ((int)M->gcb[P.S.goff[s] + g] << 6) > xw) sl = -1; }
        s_slot[wv][lane] = sl; // (B: every group is read from B's own storage, which is what slot -1 selects)
        const unsigned long long own = __ballot(sl >= 0);
        gs = BASE ? 0 : uni(own ? __ffsll((long long)own) - 1 : H4 + 1);
    }
    __syncthreads();

    const size_t ns = (size_t)W * H;
    const float *store_k = P.store + (size_t)k * P.S.cand_stride, *store_b = P.store + (size_t)P.base * P.S.cand_stride;
    // wave-uniform bases; per group: + g * (36 W | 12 W | 4 W) floats
    const rsrc_t h_own = make_rsrc(store_k + P.S.off_hout[s] + (size_t)(ch * 3) * 4 * W + ((size_t)(xw >> 6) << 8));
    const rsrc_t h_b = make_rsrc(store_b + P.S.off_hout[s] + (size_t)(ch * 3) * 4 * W + ((size_t)(xw >> 6) << 8));
    const rsrc_t h_zero = make_rsrc(P.zeros);
    const rsrc_t x_own = make_rsrc(store_k + P.S.off_xybR[s] + (size_t)ch * 4 * W + (size_t)xw * 4);
    const rsrc_t x_b = make_rsrc(store_b + P.S.off_xybR[s] + (size_t)ch * 4 * W + (size_t)xw * 4);
    const rsrc_t m1_b = make_rsrc(P.mu1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
    const rsrc_t sd1_b = make_rsrc(P.sd1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
    const rsrc_t a1_b = make_rsrc(P.a1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
    const rsrc_t r1_b = make_rsrc(P.r1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
    const uint32_t l4 = (uint32_t)lane << 2, l8 = (uint32_t)lane << 3, l16 = (uint32_t)lane << 4, l32 = (uint32_t)lane << 5;
    const uint32_t W16 = (uint32_t)W * 16u; // bytes per (plane, group): 4 W floats

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
    float sa[3][3], sb[3][3];
    double acc[6];
    float *ck_f = P.ckf + P.S.off_ckf[s] + ((size_t)ch * (H4 + 2)) * 18 * W + xw + lane; // B: record being written
    double *ck_a = P.cka + P.S.off_cka[s] + ((size_t)ch * (H4 + 2)) * 6 * W + xw + lane;
    if (BASE) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = 0.0f; sb[p][q] = 0.0f; }
#pragma unroll
        for (int q = 0; q < 6; q++) acc[q] = 0.0;
    } else { // checkpoint record gs: ckf[s][ch][g][18][W], cka[s][ch][g][6][W]
        const float *cf = P.ckf + P.S.off_ckf[s] + ((size_t)ch * (H4 + 2) + gs) * 18 * W + xw;
        const double *ca = P.cka + P.S.off_cka[s] + ((size_t)ch * (H4 + 2) + gs) * 6 * W + xw;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = ld_at<float>(cf + (size_t)(p * 6 + q) * W, l4); sb[p][q] = ld_at<float>(cf + (size_t)(p * 6 + 3 + q) * W, l4); }
#pragma unroll
        for (int q = 0; q < 6; q++) acc[q] = ld_at<double>(ca + (size_t)q * W, l8);
    }
    // the three planes of group g for this wave: the candidate's if it wrote these columns, else B's; zeros below the image
    // (the prefetch stays unconditional: no per-iteration zeroing of its registers)
#define SNES_HLOAD(GG, DST)                                                                                                   \
    {                                                                                                                         \
        const int gg_ = (GG);                                                                                                 \
        const bool below_ = gg_ >= H4;                                                                                        \
        const int sl_ = uni((int)s_slot[wv][below_ ? 0 : gg_]);                                                               \
        const rsrc_t r_ = below_ ? h_zero : (sl_ >= 0 ? h_own : h_b);                                                         \
        const uint32_t so_ = below_ ? 0u : (uint32_t)(sl_ >= 0 ? sl_ : gg_) * 9u * W16;                                       \
        DST[0] = buf_ld4(r_, l16, so_); DST[1] = buf_ld4(r_, l16, so_ + W16); DST[2] = buf_ld4(r_, l16, so_ + 2u * W16);      \
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // Tail ring slots are relative to gs: group gs+r keeps its xy half in slot r mod 2 and its zw half in slot r mod 3.
    float2 *ring_xy = reinterpret_cast<float2 *>(s_raw) + t;                             // + (slot*3 + plane) * 256
    float2 *ring_zw = reinterpret_cast<float2 *>(s_raw + 2 * 3 * 256 * sizeof(float2)) + t;
    float4 bufA[3], bufB[3];
    float4 t1[3] = {zero4, zero4, zero4}, t2[3] = {zero4, zero4, zero4}, t3[3] = {zero4, zero4, zero4};
#pragma unroll
    for (int p = 0; p < 3; p++) { bufA[p] = zero4; bufB[p] = zero4; }
    if (gs <= H4) {
        if (gs < H4) SNES_HLOAD(gs, bufA)
        if (gs - 1 >= 0) SNES_HLOAD(gs - 1, t1)
        if (gs - 2 >= 0) SNES_HLOAD(gs - 2, t2)
        if (gs - 3 >= 0) SNES_HLOAD(gs - 3, t3)
    }
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const float4 g1 = t1[p], g2 = t2[p], g3 = t3[p];
        ring_xy[(1 * 3 + p) * 256] = make_float2(g1.x, g1.y); ring_zw[(2 * 3 + p) * 256] = make_float2(g1.z, g1.w); // r = -1
        ring_xy[(0 * 3 + p) * 256] = make_float2(g2.x, g2.y); ring_zw[(1 * 3 + p) * 256] = make_float2(g2.z, g2.w); // r = -2
        ring_zw[(0 * 3 + p) * 256] = make_float2(g3.z, g3.w);                                                       // r = -3
    }
    int r2 = 0, r3 = 0; // (g - gs) mod 2, mod 3: wave-uniform ring slots
#define SNES_V2GROUP(CUR, NXT)                                                                                                \
    {                                                                                                                         \
        if (BASE && active) { /* record g */                                                                                  \
            _Pragma("unroll") for (int p = 0; p < 3; p++)                                                                     \
                _Pragma("unroll") for (int q = 0; q < 3; q++) { ck_f[(size_t)(p * 6 + q) * W] = sa[p][q]; ck_f[(size_t)(p * 6 + 3 + q) * W] = sb[p][q]; } \
            _Pragma("unroll") for (int q = 0; q < 6; q++) ck_a[(size_t)q * W] = acc[q];                                       \
            ck_f += (size_t)18 * W; ck_a += (size_t)6 * W;                                                                    \
        }                                                                                                                     \
        SNES_HLOAD(g + 1, NXT)                                                                                                \
        /* inputs of the maps of row group g-1, consumed after the recurrence steps below (g = 0: group 0's, unused) */        \
        float4 c_m1, c_sd1, c_a1, c_x;                                                                                        \
        double2 c_ra, c_rb;                                                                                                   \
        {                                                                                                                     \
            const int gm = g >= 1 ? g - 1 : 0;                                                                                \
            const uint32_t go = (uint32_t)gm * W16;                                                                           \
            c_m1 = buf_ld4(m1_b, l16, go); c_sd1 = buf_ld4(sd1_b, l16, go); c_a1 = buf_ld4(a1_b, l16, go);                    \
            c_ra = buf_ld2d(r1_b, l32, 2u * go); c_rb = buf_ld2d(r1_b, l32 + 16u, 2u * go);                                   \
            const int sl = uni((int)s_slot[wv][gm]);                                                                          \
            c_x = buf_ld4(sl >= 0 ? x_own : x_b, l16, (uint32_t)(sl >= 0 ? sl : gm) * 3u * W16);                              \
        }                                                                                                                     \
        float outp[3][4];                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < 3; p++) {                                                                       \
            float2 *pz = ring_zw + (r3 * 3 + p) * 256, *px = ring_xy + (r2 * 3 + p) * 256;                                    \
            const float2 tz = *pz, tx = *px; /* group g-3 second half, g-2 first half */                                      \
            *pz = make_float2(CUR[p].z, CUR[p].w);                                                                            \
            *px = make_float2(CUR[p].x, CUR[p].y);                                                                            \
            SNES_VSTEP(tz.x + CUR[p].x, sa[p], sb[p], outp[p][0])                                                             \
            SNES_VSTEP(tz.y + CUR[p].y, sb[p], sa[p], outp[p][1])                                                             \
            SNES_VSTEP(tx.x + CUR[p].z, sa[p], sb[p], outp[p][2])                                                             \
            SNES_VSTEP(tx.y + CUR[p].w, sb[p], sa[p], outp[p][3])                                                             \
        }                                                                                                                     \
        /* a1 stays binary32 until the maps: converted where it is loaded, the iteration would open with a wait on memory */  \
        asm volatile("" : "+v"(c_a1.x), "+v"(c_a1.y), "+v"(c_a1.z), "+v"(c_a1.w));                                            \
        if (g >= 1) {                                                                                                         \
            const float m1v[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w}, sd1v[4] = {c_sd1.x, c_sd1.y, c_sd1.z, c_sd1.w};              \
            const float a1v[4] = {c_a1.x, c_a1.y, c_a1.z, c_a1.w}, i2v[4] = {c_x.x, c_x.y, c_x.z, c_x.w};                     \
            const double r1v[4] = {c_ra.x, c_ra.y, c_rb.x, c_rb.y};                                                           \
            _Pragma("unroll") for (int q = 0; q < 4; q++)                                                                     \
                maps_accumulate(acc, m1v[q], sd1v[q], a1v[q], r1v[q], outp[0][q], outp[1][q], outp[2][q], i2v[q]);            \
        }                                                                                                                     \
        r2 ^= 1; r3 = r3 == 2 ? 0 : r3 + 1;                                                                                   \
    }
    for (int g = gs; g <= H4; g++) {
        SNES_V2GROUP(bufA, bufB)
        g++;
        if (g > H4) break;
        SNES_V2GROUP(bufB, bufA)
    }
#undef SNES_V2GROUP
#undef SNES_HLOAD
    if (BASE && active) { // final record (H4 + 1): the pooling sums of the whole column
#pragma unroll
        for (int q = 0; q < 6; q++) ck_a[(size_t)q * W] = acc[q];
    }
    __syncthreads(); // the tail ring is dead: its space becomes the reduction scratch
#pragma unroll
    for (int q = 0; q < 6; q++) red[t][q] = active ? acc[q] : 0.0;
    __syncthreads();
    const int xp = t & (W - 1); // column inside the pair
    for (int stride = W >> 1; stride > 0; stride >>= 1) {
        if (xp < stride) {
#pragma unroll
            for (int q = 0; q < 6; q++) red[t][q] += red[t + stride][q];
        }
        __syncthreads();
    }
    if (xp == 0 && active) {
        double *o = P.part + (((size_t)k * G.nscales + s) * 3 + ch) * 6;
#pragma unroll
        for (int q = 0; q < 6; q++) o[q] = red[t][q];
    }
}
// ---- B's V sweep with the work of a column split over two waves (round 4) ---------------------------------------------------
// sparse_v2_body<true> sweeps a single image: 21 waves in all, each alone on its SIMD, 65 dependent group iterations of ~245
// vector instructions at the widest scale — a wave alone issues an instruction every ~4.6 ns there (binary64 among them):
// 75 us, the longest piece of the fixed chain of a call once the candidates' own stages are short.  Here a block is the 64
// columns of one (channel, scale) and holds TWO waves: wave R runs the three recurrences (vertical_pass, 120 instructions per
// group iteration) and leaves the twelve outputs per column in LDS; wave M, one iteration behind, forms the maps and the
// pooling sums from them (maps_accumulate, ~125).  Same operations on the same values in the same order — the recurrence
// state and the sums are simply held by different waves — and the same checkpoint records: record g = R's state before its
// iteration g (ckf) and M's sums before the maps of row group g-1 (cka); record H/4 + 1 = the final sums.
// One barrier per iteration (s_barrier behind an LDS-only wait: __syncthreads() would also drain the loads in flight).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void sparse_v2_base_split_body(const SparseParams &P, const int s, const int ch, const int cb) { // cb: 64-column block of the scale
    __shared__ __attribute__((aligned(16))) float s_x[2][3][64][4]; // [iteration parity][plane][column][row of the group]
    const Geom &G = P.G;
    const int W = uni(G.sw[s]), H = uni(G.sh[s]), H4 = H >> 2;
    const int lane = threadIdx.x & 63;
    const bool is_r = uni((int)(threadIdx.x >> 6)) == 0;
    const int xw = cb << 6;
    const size_t ns = (size_t)W * H;
    const float *store_b = P.store + (size_t)P.base * P.S.cand_stride;
    const uint32_t l16 = (uint32_t)lane << 4, l32 = (uint32_t)lane << 5;
    const uint32_t W16 = (uint32_t)W * 16u; // bytes per (plane, group): 4 W floats
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (is_r) {
        const rsrc_t h_b = make_rsrc(store_b + P.S.off_hout[s] + (size_t)(ch * 3) * 4 * W + ((size_t)(xw >> 6) << 8));
        const rsrc_t h_zero = make_rsrc(P.zeros);
        const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
        const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
        float sa[3][3], sb[3][3];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = 0.0f; sb[p][q] = 0.0f; }
        float *ck_f = P.ckf + P.S.off_ckf[s] + ((size_t)ch * (H4 + 2)) * 18 * W + xw + lane;
        float4 ring[3][3]; // [plane][group mod 3]: groups g-3, g-2 (tails) and the one being read
        float4 nxt[3];
#pragma unroll
        for (int p = 0; p < 3; p++) { ring[p][0] = zero4; ring[p][1] = zero4; ring[p][2] = zero4; }
#define SNES_V2S_HLOAD(GG, DST) { const int gg_ = (GG); const bool below_ = gg_ >= H4; const rsrc_t r_ = below_ ? h_zero : h_b; const uint32_t so_ = below_ ? 0u : (uint32_t)gg_ * 9u * W16; \
            DST[0] = buf_ld4(r_, l16, so_); DST[1] = buf_ld4(r_, l16, so_ + W16); DST[2] = buf_ld4(r_, l16, so_ + 2u * W16); }
        SNES_V2S_HLOAD(0, nxt)
#define SNES_V2S_RITER(T) /* T = g mod 3 */                                                                                  \
        {                                                                                                                     \
            constexpr int u_ = (T) % 3, ux_ = ((T) + 1) % 3;                                                                  \
            _Pragma("unroll") for (int p = 0; p < 3; p++)                                                                     \
                _Pragma("unroll") for (int q = 0; q < 3; q++) { ck_f[(size_t)(p * 6 + q) * W] = sa[p][q]; ck_f[(size_t)(p * 6 + 3 + q) * W] = sb[p][q]; } \
            ck_f += (size_t)18 * W;                                                                                           \
            float4 cur[3] = {nxt[0], nxt[1], nxt[2]};                                                                         \
            SNES_V2S_HLOAD(g + 1, nxt)                                                                                        \
            _Pragma("unroll") for (int p = 0; p < 3; p++) {                                                                   \
                const float4 tz = ring[p][u_], tx = ring[p][ux_]; /* group g-3 (second half), group g-2 (first half) */       \
                float4 o;                                                                                                     \
                SNES_VSTEP(tz.z + cur[p].x, sa[p], sb[p], o.x)                                                                \
                SNES_VSTEP(tz.w + cur[p].y, sb[p], sa[p], o.y)                                                                \
                SNES_VSTEP(tx.x + cur[p].z, sa[p], sb[p], o.z)                                                                \
                SNES_VSTEP(tx.y + cur[p].w, sb[p], sa[p], o.w)                                                                \
                ring[p][u_] = cur[p];                                                                                         \
                *reinterpret_cast<float4 *>(&s_x[g & 1][p][lane][0]) = o;                                                     \
            }                                                                                                                 \
        }
        for (int g0 = 0; g0 <= H4 + 1; g0 += 3) { // steps 0 .. H4+1: R works on steps 0 .. H4, every step ends at the barrier
            { const int g = g0; if (g <= H4) SNES_V2S_RITER(0) lds_barrier(); }
            { const int g = g0 + 1; if (g > H4 + 1) break; if (g <= H4) SNES_V2S_RITER(1) lds_barrier(); }
            { const int g = g0 + 2; if (g > H4 + 1) break; if (g <= H4) SNES_V2S_RITER(2) lds_barrier(); }
        }
#undef SNES_V2S_RITER
#undef SNES_V2S_HLOAD
    } else {
        const rsrc_t x_b = make_rsrc(store_b + P.S.off_xybR[s] + (size_t)ch * 4 * W + (size_t)xw * 4);
        const rsrc_t m1_b = make_rsrc(P.mu1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
        const rsrc_t sd1_b = make_rsrc(P.sd1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
        const rsrc_t a1_b = make_rsrc(P.a1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
        const rsrc_t r1_b = make_rsrc(P.r1R4 + G.src_off[s] + (size_t)ch * ns + (size_t)xw * 4);
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double *ck_a = P.cka + P.S.off_cka[s] + ((size_t)ch * (H4 + 2)) * 6 * W + xw + lane;
        // the map inputs of row group gm, fetched an iteration ahead
        float4 n_m1, n_sd1, n_a1, n_x; double2 n_ra, n_rb;
#define SNES_V2S_MLOAD(GM) { const uint32_t go_ = (uint32_t)(GM) * W16; n_m1 = buf_ld4(m1_b, l16, go_); n_sd1 = buf_ld4(sd1_b, l16, go_); n_a1 = buf_ld4(a1_b, l16, go_); \
            n_ra = buf_ld2d(r1_b, l32, 2u * go_); n_rb = buf_ld2d(r1_b, l32 + 16u, 2u * go_); n_x = buf_ld4(x_b, l16, (uint32_t)(GM) * 3u * W16); }
        SNES_V2S_MLOAD(0)
        for (int t = 0; t <= H4 + 1; t++) { // step t: the maps of R's iteration g = t - 1, i.e. of row group t - 2
            if (t >= 1) {
                const int g = t - 1;
#pragma unroll
                for (int q = 0; q < 6; q++) ck_a[(size_t)q * W] = acc[q]; // record g: the sums before the maps of row group g-1
                ck_a += (size_t)6 * W;
                if (g >= 1) {
                    const float4 c_m1 = n_m1, c_sd1 = n_sd1, c_a1 = n_a1, c_x = n_x; const double2 c_ra = n_ra, c_rb = n_rb;
                    { const int gn = g < H4 ? g : H4 - 1; SNES_V2S_MLOAD(gn) } // group g's, for the next step (clamped: unused behind the last)
                    const float4 o0 = *reinterpret_cast<const float4 *>(&s_x[g & 1][0][lane][0]), o1 = *reinterpret_cast<const float4 *>(&s_x[g & 1][1][lane][0]),
                                 o2 = *reinterpret_cast<const float4 *>(&s_x[g & 1][2][lane][0]);
                    const float m1v[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w}, sd1v[4] = {c_sd1.x, c_sd1.y, c_sd1.z, c_sd1.w};
                    const float a1v[4] = {c_a1.x, c_a1.y, c_a1.z, c_a1.w}, i2v[4] = {c_x.x, c_x.y, c_x.z, c_x.w};
                    const double r1v[4] = {c_ra.x, c_ra.y, c_rb.x, c_rb.y};
                    const float p0[4] = {o0.x, o0.y, o0.z, o0.w}, p1[4] = {o1.x, o1.y, o1.z, o1.w}, p2[4] = {o2.x, o2.y, o2.z, o2.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) maps_accumulate(acc, m1v[q], sd1v[q], a1v[q], r1v[q], p0[q], p1[q], p2[q], i2v[q]);
                }
            }
            lds_barrier();
        }
#undef SNES_V2S_MLOAD
#pragma unroll
        for (int q = 0; q < 6; q++) ck_a[(size_t)q * W] = acc[q]; // final record (H4 + 1): the pooling sums of the whole column
    }
}

#undef SNES_HSTEP
#undef SNES_VSTEP

// entry points: the H pass takes the lists of the wide scales (grid.y = list), the V pass the wide scales (grid.y = scale)
template <bool BASE>
__device__ __forceinline__ void sparse_h2_dispatch(const SparseParams &P, const int list, const int bx, const int gx) {
    const int s = list / kColBuckets;
    if (s >= P.G.nscales || P.G.sw[s] < 64 || (list % kColBuckets) >= (P.G.sw[s] >> 6)) return;
    __shared__ H2Shared sh;
    if (s == 0) sparse_h2_body<true, BASE>(P, list, bx, gx, sh); else sparse_h2_body<false, BASE>(P, list, bx, gx, sh);
}
__global__ __launch_bounds__(64) void k_sparse_h2(SparseParams P) { sparse_h2_dispatch<false>(P, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x); }
// the lists from list0 on (the launch that follows the downscale when scale 0's lists, which do not read it, went ahead beside it)
__global__ __launch_bounds__(64) void k_sparse_h2_from(SparseParams P, int list0) { sparse_h2_dispatch<false>(P, list0 + (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x); }
__global__ __launch_bounds__(64) void k_sparse_h2q(SparseParams P) { sparse_h2q_dispatch<false>(P, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x); } // short lists
__global__ __launch_bounds__(64) void k_sparse_h2q_base(SparseParams P) { sparse_h2q_dispatch<true>(P, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x); } // B: a single image is a short list
__global__ __launch_bounds__(64) void k_sparse_h2_base(SparseParams P) { sparse_h2_dispatch<true>(P, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x); }
__global__ __launch_bounds__(256, 5) void k_sparse_v2(SparseParams P) { if ((int)blockIdx.y < P.G.nscales && P.G.sw[blockIdx.y] >= 64) sparse_v2_body<false>(P, (int)blockIdx.y, (int)blockIdx.x); }
// the wide scales from s0 on (the launch that follows the other scales' H pass when scale 0, whose H pass went ahead, runs beside it)
__global__ __launch_bounds__(256, 5) void k_sparse_v2_from(SparseParams P, int s0) { const int s = s0 + (int)blockIdx.y; if (s < P.G.nscales && P.G.sw[s] >= 64) sparse_v2_body<false>(P, s, (int)blockIdx.x); }
// B: the wide scales in this body (grid.y = scale), the narrow ones in the general one (grid.y = scale - s_first), two launches:
// one kernel holding both bodies would take the larger register allocation for every block
// B, split: grid.x = 64-column block (W / 64 of the widest scale), grid.y = channel, grid.z = scale; 128 threads
__global__ __launch_bounds__(128) void k_sparse_v2_base_split(SparseParams P) { const int s = (int)blockIdx.z; if (s < P.G.nscales && P.G.sw[s] >= 64 && (int)blockIdx.x < (P.G.sw[s] >> 6)) sparse_v2_base_split_body(P, s, (int)blockIdx.y, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void k_sparse_v2_base(SparseParams P) { if ((int)blockIdx.y < P.G.nscales && P.G.sw[blockIdx.y] >= 64) sparse_v2_body<true>(P, (int)blockIdx.y, (int)blockIdx.x); }
__device__ __forceinline__ void sparse_v_base_narrow_dispatch(const SparseParams &P) { const int s = (int)blockIdx.y + P.s_first; if (s < P.G.nscales) sparse_v_body<false, 0, 1>(P, s); }
__global__ __launch_bounds__(256, 1) void k_sparse_v_base_narrow(SparseParams P) { sparse_v_base_narrow_dispatch(P); }

} // namespace snes
