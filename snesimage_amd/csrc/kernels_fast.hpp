// snesimage_amd/csrc/kernels_fast.hpp — the blur + map kernels for the large scales (every scale whose
// width and height are multiples of 64: 256x256, 128x128, 64x64 at the BASELINE size).
//
// Same arithmetic, operation for operation, as k_hpass / k_vpass in kernels.hpp (the recursive
// Gaussian cannot be re-associated, see there); what changes is how the data moves:
//   * one wavefront = 64 image rows (H pass) or 64 image columns (V pass); the 3 planes x 3 terms of
//     a (candidate, channel) pair give each lane 9 independent recurrences, enough ILP to keep a SIMD
//     issuing without relying on occupancy;
//   * every global access is a 16-byte-per-lane, 1-KiB-per-wave coalesced access: the inputs are
//     stored in "C4" ([x/4][y][x%4], lanes along y) for the H pass and "R4" ([y/4][x][y%4], lanes
//     along x) for the V pass; the H pass hands its output to the V pass in "XT4"
//     ([x/64][y/4][x%64][y%4]), written after a 4x4 transpose through a 3.75 KiB per-wave LDS pad;
//   * loads for the next group of four steps are issued before the current group is computed
//     (software prefetch), so a step never waits on HBM/L2 latency;
//   * the 10-deep delay line of the truncated-cosine filter (in[n-6]) is a five-slot register ring that the
//     prefetch loads write directly; the recurrence state alternates between two register sets instead of
//     being shifted, so the inner loop moves no registers.
#pragma once
#include "kernels.hpp"

namespace snes {

struct FastParams {
    Geom G; BlurK K;
    int s, npairs, ncol;
    const unsigned long long *packC4, *packR4; // scale 0
    // --dither: every candidate has its own palette_map (k_dither); the colour index of a pixel is then
    // sub[px] + map[px] (sub = subpalette base of the pixel's tile, 255 = transparent), and the slot's index means the candidate
    int use_maps; const uint32_t *mapsC4, *mapsR4, *subC4, *subR4; // 4 pixels per word, blocked like the pack
    const float *pal_xyb, *cand_tab;
    const float *img1C4;                       // source, this scale: [3][C4]
    const float *mu1R4, *sd1R4, *a1R4; const double *r1R4; // source, this scale: [3][R4] (maps_accumulate)
    float *work; double *part;
};

// colour index of 4 pixels from their pack words (RGB / redmean keys).  thr == 0 ("candidate never wins")
// needs no special case: an unsigned key is never < 0.
__device__ __forceinline__ void resolve4(const uint4 a, const uint4 b, uint32_t crgb, uint32_t ncol, uint32_t ci[4]) {
    const uint32_t lo[4] = {a.x, a.z, b.x, b.z}, hi[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
    for (int j = 0; j < 4; j++) ci[j] = red_mean_key(crgb, lo[j] & 0x00ffffffu) < hi[j] ? ncol : (lo[j] >> 24);
}

// colour index of 4 pixels from their map bytes (dither path)
__device__ __forceinline__ void resolve4_maps(uint32_t mw, uint32_t sw, uint32_t slot_ci, uint32_t ncol, uint32_t ci[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t sb = (sw >> (8 * j)) & 0xffu, m = (mw >> (8 * j)) & 0xffu;
        const uint32_t c = sb == 255u ? ncol + 1u : sb + m;
        ci[j] = c == slot_ci ? ncol : c;
    }
}

// One step of the three recurrences of a plane.  A holds y[n-1], B holds y[n-2]; the new value is written
// over B, so the caller swaps the roles of A and B on the next step instead of moving registers.
// Horizontal form (blur/gaussian.rs horizontal_row): out = fma(MUL_PREV, prev, fma(-1, prev2, sum * MUL_IN))
#define SNES_HSTEP(SUM, A, B, OUT)                                                   \
    {                                                                                \
        float o1_ = (SUM) * n2_0, o3_ = (SUM) * n2_1, o5_ = (SUM) * n2_2;            \
        o1_ = fmaf(-1.0f, B[0], o1_); o3_ = fmaf(-1.0f, B[1], o3_); o5_ = fmaf(-1.0f, B[2], o5_); \
        o1_ = fmaf(mp_0, A[0], o1_); o3_ = fmaf(mp_1, A[1], o3_); o5_ = fmaf(mp_2, A[2], o5_);    \
        B[0] = o1_; B[1] = o3_; B[2] = o5_;                                          \
        OUT = o1_ + o3_ + o5_;                                                       \
    }
// Vertical form (vertical_pass): out = fma(sum, MUL_IN, -fma(prev, MUL_PREV, prev2))
#define SNES_VSTEP(SUM, A, B, OUT)                                                   \
    {                                                                                \
        float o1_ = fmaf(A[0], d1_0, B[0]), o3_ = fmaf(A[1], d1_1, B[1]), o5_ = fmaf(A[2], d1_2, B[2]); \
        o1_ = fmaf((SUM), n2_0, -o1_); o3_ = fmaf((SUM), n2_1, -o3_); o5_ = fmaf((SUM), n2_2, -o5_);    \
        B[0] = o1_; B[1] = o3_; B[2] = o5_;                                          \
        OUT = o1_ + o3_ + o5_;                                                       \
    }

// ---- H pass: one wave = rows [64*yb, 64*yb+64) of one (candidate, channel) -------------------------
// Groups of four columns; group g carries the "right" inputs in[4g..4g+3] of steps n = 4g-4..4g-1.  A
// five-slot register ring holds groups g-3..g+1: g+1 is the prefetch target (loads land in the ring,
// nothing is copied), g-3 and g-2 supply in[n-6].
template <bool S0>
__global__ __launch_bounds__(64) void k_hpass_fast(FastParams P) {
    __shared__ float s_lut[256];
    __shared__ float s_tr[3][64 * 5];
    const Geom &G = P.G;
    const int s = P.s, W = G.sw[s], H = G.sh[s];
    const int nyb = H >> 6;
    const int lane = threadIdx.x;
    const int yb = blockIdx.x % nyb, pair = blockIdx.x / nyb;
    const int cand = pair / 3, ch = pair - cand * 3;
    const int y = (yb << 6) + lane;
    const size_t ns = (size_t)W * H;

    float cand_v = 0.0f; uint32_t crgb = 0;
    if (S0) {
        for (int i = lane; i < 256; i += 64) s_lut[i] = (i < P.ncol + 2) ? P.pal_xyb[3 * i + ch] : 0.0f;
        cand_v = P.cand_tab[8 * (size_t)cand + 3 + ch];
        crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        __syncthreads();
    }
    const float4 *in1 = reinterpret_cast<const float4 *>(P.img1C4 + (size_t)ch * ns) + y;                                   // advances by H per group
    const float4 *in2 = S0 ? nullptr : reinterpret_cast<const float4 *>(P.work + (size_t)cand * G.cand_stride + G.off_xybT[s] + (size_t)ch * ns) + y;
    const bool use_maps = S0 && P.use_maps;
    const uint32_t slot_ci = S0 ? __float_as_uint(P.cand_tab[8 * (size_t)cand + 7]) : 0u;
    const uint4 *pk = (S0 && !use_maps) ? reinterpret_cast<const uint4 *>(P.packC4) + 2 * (size_t)y : nullptr;                 // advances by 2H per group
    const uint32_t *mp = use_maps ? P.mapsC4 + (size_t)cand * (ns >> 2) + y : nullptr, *sbp = use_maps ? P.subC4 + y : nullptr; // advance by H per group
    float *hout = P.work + (size_t)cand * G.cand_stride + G.off_hout[s] + (size_t)(ch * 3) * ns;
    // XT4 offset of (x = 4(g-1) + (lane&3), y = 64*yb + (lane&~3)); per group x advances by 4 -> +16 floats, and by a whole
    // column block (H*64 floats) every 16 groups
    const int li = lane & 3, k4 = lane & ~3;
    const size_t o_row = ((size_t)(((yb << 6) + k4) >> 2) << 8) + ((size_t)li << 2);

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
    float sa[3][3], sb[3][3]; // recurrence state per plane: roles (prev, prev2) alternate every step
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) { sa[p][k] = 0.0f; sb[p][k] = 0.0f; }
    float4 r1[5], r2[5];
#pragma unroll
    for (int a = 0; a < 5; a++) { r1[a] = make_float4(0.f, 0.f, 0.f, 0.f); r2[a] = r1[a]; }

    const int G4 = W >> 2;
    uint4 n_pa = make_uint4(0, 0, 0, 0), n_pb = n_pa;
    r1[0] = in1[0];
    if (use_maps) { n_pa.x = mp[0]; n_pa.y = sbp[0]; mp += H; sbp += H; }
    else if (S0) { n_pa = pk[0]; n_pb = pk[1]; pk += 2 * (size_t)H; } else { r2[0] = in2[0]; in2 += H; }
    in1 += H;

    for (int g0 = 0; g0 <= G4; g0 += 5) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int g = g0 + u;
            if (g > G4) break;
            const int un = (u + 1) % 5, ua = (u + 2) % 5, ub = (u + 3) % 5; // slots of groups g+1, g-3, g-2
            const uint4 c_pa = n_pa, c_pb = n_pb;
            if (g + 1 < G4) { // prefetch group g+1 straight into its ring slot
                r1[un] = in1[0]; in1 += H;
                if (use_maps) { n_pa.x = mp[0]; n_pa.y = sbp[0]; mp += H; sbp += H; }
                else if (S0) { n_pa = pk[0]; n_pb = pk[1]; pk += 2 * (size_t)H; } else { r2[un] = in2[0]; in2 += H; }
            } else { r1[un] = make_float4(0.f, 0.f, 0.f, 0.f); r2[un] = r1[un]; n_pa = make_uint4(0, 0, 0, 0); n_pb = n_pa; }
            if (S0 && g < G4) {
                uint32_t ci[4];
                if (use_maps) resolve4_maps(c_pa.x, c_pa.y, slot_ci, (uint32_t)P.ncol, ci); else resolve4(c_pa, c_pb, crgb, (uint32_t)P.ncol, ci);
                r2[u].x = (ci[0] == (uint32_t)P.ncol) ? cand_v : s_lut[ci[0]];
                r2[u].y = (ci[1] == (uint32_t)P.ncol) ? cand_v : s_lut[ci[1]];
                r2[u].z = (ci[2] == (uint32_t)P.ncol) ? cand_v : s_lut[ci[2]];
                r2[u].w = (ci[3] == (uint32_t)P.ncol) ? cand_v : s_lut[ci[3]];
            }
            const float v1[4] = {r1[u].x, r1[u].y, r1[u].z, r1[u].w}, v2[4] = {r2[u].x, r2[u].y, r2[u].z, r2[u].w};
            const float l1[4] = {r1[ua].z, r1[ua].w, r1[ub].x, r1[ub].y}, l2[4] = {r2[ua].z, r2[ua].w, r2[ub].x, r2[ub].y}; // in[xr - 10]
            float outp[3][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float s0 = l2[j] + v2[j];
                const float s1 = (l2[j] * l2[j]) + (v2[j] * v2[j]);
                const float s2 = (l1[j] * l2[j]) + (v1[j] * v2[j]);
                if ((j & 1) == 0) { SNES_HSTEP(s0, sa[0], sb[0], outp[0][j]) SNES_HSTEP(s1, sa[1], sb[1], outp[1][j]) SNES_HSTEP(s2, sa[2], sb[2], outp[2][j]) }
                else { SNES_HSTEP(s0, sb[0], sa[0], outp[0][j]) SNES_HSTEP(s1, sb[1], sa[1], outp[1][j]) SNES_HSTEP(s2, sb[2], sa[2], outp[2][j]) }
            }
            if (g >= 1) { // outputs x = 4(g-1)..4(g-1)+3 of row y: 4x4 transpose inside each lane quad, then one 16-byte store per plane
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    s_tr[p][lane * 5 + 0] = outp[p][0]; s_tr[p][lane * 5 + 1] = outp[p][1]; s_tr[p][lane * 5 + 2] = outp[p][2]; s_tr[p][lane * 5 + 3] = outp[p][3];
                }
                __syncthreads();
                const int xg = (g - 1) << 2; // first column of the group
                const size_t o = (((size_t)(xg >> 6) * (size_t)(H >> 2)) << 8) + o_row + ((size_t)(xg & 63) << 2);
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    float4 v;
                    v.x = s_tr[p][(k4 + 0) * 5 + li]; v.y = s_tr[p][(k4 + 1) * 5 + li]; v.z = s_tr[p][(k4 + 2) * 5 + li]; v.w = s_tr[p][(k4 + 3) * 5 + li];
                    *reinterpret_cast<float4 *>(hout + (size_t)p * ns + o) = v;
                }
                __syncthreads();
            }
        }
    }
}

// ---- V pass + maps: one wave = 64 columns, a block = 256 columns = 256/W pairs ------------------------
template <bool S0>
__global__ __launch_bounds__(256) void k_vpass_fast(FastParams P) {
    __shared__ float s_lut[3][256];
    __shared__ double red[256][6];
    const Geom &G = P.G;
    const int s = P.s, W = G.sw[s], H = G.sh[s];
    const int t = threadIdx.x;
    const int ppw = 256 / W;
    const int ql = t / W, x = t - ql * W;
    const int pair_raw = blockIdx.x * ppw + ql;
    const bool active = pair_raw < P.npairs;
    const int pair = active ? pair_raw : 0;
    const int cand = pair / 3, ch = pair - cand * 3;
    const size_t ns = (size_t)W * H;

    float cand_v = 0.0f; uint32_t crgb = 0;
    if (S0) {
        for (int i = t; i < 3 * 256; i += 256) { int c = i >> 8, j = i & 255; s_lut[c][j] = (j < P.ncol + 2) ? P.pal_xyb[3 * j + c] : 0.0f; }
        cand_v = P.cand_tab[8 * (size_t)cand + 3 + ch];
        crgb = __float_as_uint(P.cand_tab[8 * (size_t)cand + 6]);
        __syncthreads();
    }
    const float *lut = s_lut[ch];
    const int H4 = H >> 2;
    const size_t plane4 = ns >> 2;
    // XT4: float4 index ((xb*H4 + g) << 6) + lane; advances by 64 per row group
    const float4 *h0 = reinterpret_cast<const float4 *>(P.work + (size_t)cand * G.cand_stride + G.off_hout[s] + (size_t)(ch * 3) * ns) + (((size_t)(x >> 6) * H4) << 6) + (x & 63);
    const float4 *h1 = h0 + plane4, *h2 = h1 + plane4;
    // R4: float4 index g*W + x; advances by W per row group
    const float4 *mu1 = reinterpret_cast<const float4 *>(P.mu1R4 + (size_t)ch * ns) + x;
    const float4 *sd1 = reinterpret_cast<const float4 *>(P.sd1R4 + (size_t)ch * ns) + x;
    const float4 *a1 = reinterpret_cast<const float4 *>(P.a1R4 + (size_t)ch * ns) + x;
    const double2 *r1 = reinterpret_cast<const double2 *>(P.r1R4 + (size_t)ch * ns) + 2 * (size_t)x; // two 16-byte halves per row group
    const float4 *xyb = S0 ? nullptr : reinterpret_cast<const float4 *>(P.work + (size_t)cand * G.cand_stride + G.off_xyb[s] + (size_t)ch * ns) + x;
    const bool use_maps = S0 && P.use_maps;
    const uint32_t slot_ci = S0 ? __float_as_uint(P.cand_tab[8 * (size_t)cand + 7]) : 0u;
    const uint4 *pk = (S0 && !use_maps) ? reinterpret_cast<const uint4 *>(P.packR4) + 2 * (size_t)x : nullptr;
    const uint32_t *mp = use_maps ? P.mapsR4 + (size_t)cand * (ns >> 2) + x : nullptr, *sbp = use_maps ? P.subR4 + x : nullptr; // advance by W per row group

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
    float sa[3][3], sb[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) { sa[p][k] = 0.0f; sb[p][k] = 0.0f; }
    float4 ring[3][5];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int a = 0; a < 5; a++) ring[p][a] = make_float4(0.f, 0.f, 0.f, 0.f);
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    ring[0][0] = h0[0]; ring[1][0] = h1[0]; ring[2][0] = h2[0];
    h0 += 64; h1 += 64; h2 += 64;
    // map inputs of row group g-1 travel one iteration ahead of their use
    float4 n_m1 = make_float4(0.f, 0.f, 0.f, 0.f), n_sd1 = n_m1, n_a1 = n_m1, n_x = n_m1;
    double2 n_ra = make_double2(1.0, 1.0), n_rb = n_ra;
    uint4 n_pa = make_uint4(0, 0, 0, 0), n_pb = n_pa;

    for (int g0 = 0; g0 <= H4; g0 += 5) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int g = g0 + u;
            if (g > H4) break;
            const int un = (u + 1) % 5, ua = (u + 2) % 5, ub = (u + 3) % 5;
            const float4 c_m1 = n_m1, c_sd1 = n_sd1, c_a1 = n_a1, c_x = n_x; const double2 c_ra = n_ra, c_rb = n_rb; const uint4 c_pa = n_pa, c_pb = n_pb;
            if (g + 1 < H4) { ring[0][un] = h0[0]; ring[1][un] = h1[0]; ring[2][un] = h2[0]; h0 += 64; h1 += 64; h2 += 64; }
            else { ring[0][un] = make_float4(0.f, 0.f, 0.f, 0.f); ring[1][un] = ring[0][un]; ring[2][un] = ring[0][un]; }
            if (g < H4) { // inputs of the maps of rows 4g..4g+3, consumed in the next iteration
                n_m1 = mu1[0]; n_sd1 = sd1[0]; n_a1 = a1[0]; n_ra = r1[0]; n_rb = r1[1]; mu1 += W; sd1 += W; a1 += W; r1 += 2 * W;
                if (use_maps) { n_pa.x = mp[0]; n_pa.y = sbp[0]; mp += W; sbp += W; }
                else if (S0) { n_pa = pk[0]; n_pb = pk[1]; pk += 2 * (size_t)W; } else { n_x = xyb[0]; xyb += W; }
            }
            float i2v[4] = {c_x.x, c_x.y, c_x.z, c_x.w};
            if (S0 && g >= 1) {
                uint32_t ci[4];
                if (use_maps) resolve4_maps(c_pa.x, c_pa.y, slot_ci, (uint32_t)P.ncol, ci); else resolve4(c_pa, c_pb, crgb, (uint32_t)P.ncol, ci);
#pragma unroll
                for (int j = 0; j < 4; j++) i2v[j] = (ci[j] == (uint32_t)P.ncol) ? cand_v : lut[ci[j]];
            }
            const float m1v[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w}, sd1v[4] = {c_sd1.x, c_sd1.y, c_sd1.z, c_sd1.w}, a1v[4] = {c_a1.x, c_a1.y, c_a1.z, c_a1.w};
            const double r1v[4] = {c_ra.x, c_ra.y, c_rb.x, c_rb.y};
            float in[3][4], top[3][4];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                in[p][0] = ring[p][u].x; in[p][1] = ring[p][u].y; in[p][2] = ring[p][u].z; in[p][3] = ring[p][u].w;
                top[p][0] = ring[p][ua].z; top[p][1] = ring[p][ua].w; top[p][2] = ring[p][ub].x; top[p][3] = ring[p][ub].y; // hout[n - 6]
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float outp[3];
                if ((j & 1) == 0) {
                    SNES_VSTEP(top[0][j] + in[0][j], sa[0], sb[0], outp[0]) SNES_VSTEP(top[1][j] + in[1][j], sa[1], sb[1], outp[1]) SNES_VSTEP(top[2][j] + in[2][j], sa[2], sb[2], outp[2])
                } else {
                    SNES_VSTEP(top[0][j] + in[0][j], sb[0], sa[0], outp[0]) SNES_VSTEP(top[1][j] + in[1][j], sb[1], sa[1], outp[1]) SNES_VSTEP(top[2][j] + in[2][j], sb[2], sa[2], outp[2])
                }
                if (g >= 1) maps_accumulate(acc, m1v[j], sd1v[j], a1v[j], r1v[j], outp[0], outp[1], outp[2], i2v[j]); // row n = 4(g-1) + j
            }
        }
    }
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
#undef SNES_HSTEP
#undef SNES_VSTEP

// Row-major [3][H][W] source planes -> R4 and/or C4 copies (one-off, per context)
__global__ __launch_bounds__(256) void k_relayout(const float *__restrict__ src, int W, int H, float *__restrict__ r4, float *__restrict__ c4) {
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= W * H) return;
    int x = px % W, y = px / W;
    size_t ns = (size_t)W * H;
    for (int c = 0; c < 3; c++) {
        float v = src[c * ns + px];
        if (r4) r4[c * ns + idx_r4(x, y, W)] = v;
        if (c4) c4[c * ns + idx_c4(x, y, H)] = v;
    }
}

// the same for the binary64 plane r1: R4 only
__global__ __launch_bounds__(256) void k_relayout_f64(const double *__restrict__ src, int W, int H, double *__restrict__ r4) {
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= W * H) return;
    int x = px % W, y = px / W;
    size_t ns = (size_t)W * H;
    for (int c = 0; c < 3; c++) r4[c * ns + idx_r4(x, y, W)] = src[c * ns + px];
}

} // namespace snes
