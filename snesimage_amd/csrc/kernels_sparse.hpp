// snesimage_amd/csrc/kernels_sparse.hpp — row-sparse ("delta") scoring of candidate palettes.
//
// Observation (measured on the BASELINE workload): replacing one palette entry by a random colour
// changes few pixels.  Let B be the image in which every pixel of the slot's subpalette takes its
// best *other* entry (the pack's fixed colour index); a candidate differs from B only at the pixels
// it wins — a median of ~40 pixels in ~20 of the 256 rows, clustered in the subpalette's tiles.
//
// Every stage of ssimulacra2 is causal along its sweep, so everything computed before the first
// changed input is bit-identical to what the same stage computes for B:
//   * H pass: an output row depends only on its input row  -> only changed rows are recomputed;
//   * V pass + maps + pooling sums: a column's recurrence state and its running sums at step n
//     depend only on rows < n+5                              -> resume from B's checkpoint at the
//     first changed row; columns left of the first changed column (minus the filter's reach) are
//     B's outright;
//   * downscale/XYB: a pixel of scale s depends on its 2^s x 2^s block -> only changed rows.
// B itself is scored once per slot (it is the pseudo-candidate "all rows changed, wins nothing")
// and leaves, in its own compact storage, dense row-major planes plus one checkpoint per row.
// Results are bit-identical to the dense kernels (same operations in the same order; tests compare
// the two paths), the work per candidate drops by the fraction of rows/columns it leaves untouched.
//
// Compact per-candidate storage (slot j = index of the row in the candidate's ascending changed-row
// list of that scale): lin[s][j][3][W_s], xyb[s][j][3][W_s], hout[s][j][9][W_s].
#pragma once
#include "kernels.hpp"

namespace snes {

constexpr int kRowsTotal = 504; // 256 + 128 + 64 + 32 + 16 + 8

struct CandMeta {
    int nrows[kMaxScales];
    int xmin;          // smallest x of a won pixel (W if none)
    int won;           // number of won pixels
    unsigned char rows[kRowsTotal]; // per scale: ascending changed rows
    short slot[kRowsTotal];         // per scale: row -> slot, -1 if unchanged
};

struct SparseGeom {
    long long off_lin[kMaxScales], off_xyb[kMaxScales], off_hout[kMaxScales]; // floats, inside one candidate's storage
    long long cand_stride;                                                   // floats per candidate
    long long off_ckf[kMaxScales], off_cka[kMaxScales];                      // checkpoint arrays (floats / doubles)
    int roff[kMaxScales];                                                    // offset of scale s inside CandMeta::rows / slot
};

struct SparseParams {
    Geom G; SparseGeom S; BlurK K;
    int ncand, k0, base, ncol, is_base; // candidates of this launch occupy storage indices [k0, k0+ncand); base = index of B
    const unsigned long long *pack; // row-major
    const uint4 *plist; const int *plist_count; // contested pixels of the slot: {px, rgb, thr, 0}
    const float *pal_lin, *pal_xyb, *cand_tab;
    const float *img1, *mu1, *s11; // source arrays, row-major, + G.src_off[s]
    float *store; CandMeta *meta;
    unsigned int *items; int *item_count; long long item_stride; // per scale: items[s*item_stride + i] = cand*1024 + j*4 + ch
    float *ckf; double *cka; double *part;
};

__device__ __forceinline__ uint32_t sparse_ci(unsigned long long pk, uint32_t crgb, uint32_t ncol, bool is_base) {
    const uint32_t lo = (uint32_t)pk, thr = (uint32_t)(pk >> 32);
    return (!is_base && red_mean_key(crgb, lo & 0x00ffffffu) < thr) ? ncol : (lo >> 24);
}

// ---- which rows does each candidate change? --------------------------------------------------------
__global__ __launch_bounds__(256) void k_sparse_scan(SparseParams P) {
    __shared__ unsigned int s_mask[8];
    __shared__ int s_xmin, s_won;
    __shared__ int s_flag[256];
    __shared__ int s_base[kMaxScales];
    const Geom &G = P.G;
    const int t = threadIdx.x;
    const int k = P.is_base ? P.base : P.k0 + (int)blockIdx.x;
    if (t < 8) s_mask[t] = P.is_base ? 0xffffffffu : 0u;
    if (t == 0) { s_xmin = P.is_base ? 0 : G.W; s_won = 0; }
    __syncthreads();
    if (!P.is_base) {
        const uint32_t crgb = __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
        const int n = *P.plist_count;
        for (int i = t; i < n; i += 256) {
            const uint4 e = P.plist[i];
            if (red_mean_key(crgb, e.y) < e.z) {
                const int x = (int)(e.x % (unsigned)G.W), y = (int)(e.x / (unsigned)G.W);
                atomicOr(&s_mask[y >> 5], 1u << (y & 31));
                atomicMin(&s_xmin, x);
                atomicAdd(&s_won, 1);
            }
        }
        __syncthreads();
    }
    CandMeta *M = P.meta + k;
    if (t == 0) { M->xmin = s_xmin; M->won = s_won; }
    for (int s = 0; s < G.nscales; s++) {
        const int Hs = G.sh[s];
        int flag = 0;
        if (t < Hs) {
            const int lo = t << s, len = 1 << s; // rows [lo, lo+len) of scale 0 (H = 256: Hs << s == H)
            const unsigned int w = s_mask[lo >> 5];
            const unsigned int m = (len >= 32) ? 0xffffffffu : (((1u << len) - 1u) << (lo & 31));
            flag = (w & m) != 0u;
        }
        s_flag[t] = flag;
        __syncthreads();
        int below = 0, total = 0;
        for (int i = 0; i < Hs; i++) { const int f = s_flag[i]; total += f; if (i < t) below += f; }
        if (t < Hs) {
            M->slot[P.S.roff[s] + t] = flag ? (short)below : (short)-1;
            if (flag) M->rows[P.S.roff[s] + below] = (unsigned char)t;
        }
        if (t == 0) { M->nrows[s] = total; s_base[s] = total ? atomicAdd(&P.item_count[s], total * 3) : 0; }
        __syncthreads();
        // work items of the H pass: (candidate, slot, channel)
        for (int i = t; i < total * 3; i += 256) P.items[(size_t)s * P.item_stride + s_base[s] + i] = (unsigned int)k * 1024u + (unsigned int)(i / 3) * 4u + (unsigned int)(i % 3);
        __syncthreads();
    }
}

// ---- downscale chain + XYB on changed rows only ------------------------------------------------------
__global__ __launch_bounds__(256) void k_sparse_down(SparseParams P) {
    __shared__ float s_lin[256 * 3];
    const Geom &G = P.G;
    const int t = threadIdx.x;
    const int k = P.is_base ? P.base : P.k0 + (int)blockIdx.x;
    const bool is_base = P.is_base != 0;
    for (int i = t; i < (P.ncol + 2) * 3; i += 256) s_lin[i] = P.pal_lin[i];
    __syncthreads();
    if (t < 3 && !is_base) s_lin[3 * P.ncol + t] = P.cand_tab[8 * (size_t)k + t];
    const uint32_t crgb = is_base ? 0u : __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    __syncthreads();
    const CandMeta *M = P.meta + k;
    float *mine = P.store + (size_t)k * P.S.cand_stride;
    const float *basep = P.store + (size_t)P.base * P.S.cand_stride;
    for (int s = 1; s < G.nscales; s++) {
        const int Ws = G.sw[s], Wp = G.sw[s - 1];
        const int n = M->nrows[s];
        for (int i = t; i < n * Ws; i += 256) {
            const int j = i / Ws, x = i - j * Ws;
            const int y = M->rows[P.S.roff[s] + j];
            float v[3];
            if (s == 1) {
                float sum[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int iy = 0; iy < 2; iy++)
#pragma unroll
                    for (int ix = 0; ix < 2; ix++) {
                        const uint32_t ci = sparse_ci(P.pack[(size_t)(2 * y + iy) * G.W + 2 * x + ix], crgb, (uint32_t)P.ncol, is_base);
                        sum[0] += s_lin[3 * ci]; sum[1] += s_lin[3 * ci + 1]; sum[2] += s_lin[3 * ci + 2];
                    }
                v[0] = sum[0] * 0.25f; v[1] = sum[1] * 0.25f; v[2] = sum[2] * 0.25f;
            } else {
                // rows 2y, 2y+1 of scale s-1: the candidate's own compact row if it changed, else B's
                const short sl0 = M->slot[P.S.roff[s - 1] + 2 * y], sl1 = M->slot[P.S.roff[s - 1] + 2 * y + 1];
                const float *r0 = sl0 >= 0 ? mine + P.S.off_lin[s - 1] + (size_t)sl0 * 3 * Wp : basep + P.S.off_lin[s - 1] + (size_t)(2 * y) * 3 * Wp;
                const float *r1 = sl1 >= 0 ? mine + P.S.off_lin[s - 1] + (size_t)sl1 * 3 * Wp : basep + P.S.off_lin[s - 1] + (size_t)(2 * y + 1) * 3 * Wp;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float sum = 0.0f;
                    sum += r0[c * Wp + 2 * x]; sum += r0[c * Wp + 2 * x + 1]; sum += r1[c * Wp + 2 * x]; sum += r1[c * Wp + 2 * x + 1];
                    v[c] = sum * 0.25f;
                }
            }
            float X, Y, B;
            linear_to_positive_xyb(v[0], v[1], v[2], X, Y, B);
            float *ol = mine + P.S.off_lin[s] + (size_t)j * 3 * Ws, *ox = mine + P.S.off_xyb[s] + (size_t)j * 3 * Ws;
            ol[x] = v[0]; ol[Ws + x] = v[1]; ol[2 * Ws + x] = v[2];
            ox[x] = X; ox[Ws + x] = Y; ox[2 * Ws + x] = B;
        }
        __syncthreads(); // the rows written above are read by this block at the next scale
    }
}

// ---- H pass of changed rows: one lane = one (candidate, row slot, channel) ----------------------------
__global__ __launch_bounds__(64) void k_sparse_h(SparseParams P) {
    __shared__ float s_lut[3][256];
    const Geom &G = P.G;
    const int s = blockIdx.y;
    if (s >= G.nscales) return;
    const int lane = threadIdx.x;
    const int count = P.item_count[s];
    const int i0 = blockIdx.x * 64;
    if (i0 >= count) return;
    const bool S0 = (s == 0);
    if (S0) {
        for (int i = lane; i < 3 * 256; i += 64) { int c = i >> 8, j = i & 255; s_lut[c][j] = (j < P.ncol + 2) ? P.pal_xyb[3 * j + c] : 0.0f; }
        __syncthreads();
    }
    const bool valid = (i0 + lane) < count;
    const unsigned int it = P.items[(size_t)s * P.item_stride + (valid ? i0 + lane : i0)];
    const int k = (int)(it >> 10), j = (int)((it >> 2) & 255u), ch = (int)(it & 3u);
    const bool is_base = (k == P.base);
    const CandMeta *M = P.meta + k;
    const int W = G.sw[s], H = G.sh[s];
    const int y = M->rows[P.S.roff[s] + j];
    const size_t ns = (size_t)W * H;
    const float cand_v = is_base ? 0.0f : P.cand_tab[8 * (size_t)k + 3 + ch];
    const uint32_t crgb = is_base ? 0u : __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    const float4 *in1 = reinterpret_cast<const float4 *>(P.img1 + G.src_off[s] + (size_t)ch * ns + (size_t)y * W);
    const float4 *in2 = S0 ? nullptr : reinterpret_cast<const float4 *>(P.store + (size_t)k * P.S.cand_stride + P.S.off_xyb[s] + (size_t)j * 3 * W + (size_t)ch * W);
    const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.pack + (size_t)y * W) : nullptr;
    float *out = P.store + (size_t)k * P.S.cand_stride + P.S.off_hout[s] + (size_t)j * 9 * W + (size_t)(ch * 3) * W;

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
    float pv[3][3], pv2[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int q = 0; q < 3; q++) { pv[p][q] = 0.0f; pv2[p][q] = 0.0f; }
    float4 r1[5], r2[5];
#pragma unroll
    for (int a = 0; a < 5; a++) { r1[a] = make_float4(0.f, 0.f, 0.f, 0.f); r2[a] = r1[a]; }
    const int G4 = W >> 2;
    for (int g0 = 0; g0 <= G4; g0 += 5) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int g = g0 + u;
            if (g > G4) break;
            const int ua = (u + 2) % 5, ub = (u + 3) % 5;
            if (g < G4) {
                r1[u] = in1[g];
                if (S0) {
                    const uint4 a = pk[2 * g], b = pk[2 * g + 1];
                    const unsigned long long w0 = ((unsigned long long)a.y << 32) | a.x, w1 = ((unsigned long long)a.w << 32) | a.z;
                    const unsigned long long w2 = ((unsigned long long)b.y << 32) | b.x, w3 = ((unsigned long long)b.w << 32) | b.z;
                    const uint32_t c0 = sparse_ci(w0, crgb, (uint32_t)P.ncol, is_base), c1 = sparse_ci(w1, crgb, (uint32_t)P.ncol, is_base);
                    const uint32_t c2 = sparse_ci(w2, crgb, (uint32_t)P.ncol, is_base), c3 = sparse_ci(w3, crgb, (uint32_t)P.ncol, is_base);
                    r2[u].x = c0 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c0]; r2[u].y = c1 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c1];
                    r2[u].z = c2 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c2]; r2[u].w = c3 == (uint32_t)P.ncol ? cand_v : s_lut[ch][c3];
                } else r2[u] = in2[g];
            } else { r1[u] = make_float4(0.f, 0.f, 0.f, 0.f); r2[u] = r1[u]; }
            const float v1[4] = {r1[u].x, r1[u].y, r1[u].z, r1[u].w}, v2[4] = {r2[u].x, r2[u].y, r2[u].z, r2[u].w};
            const float l1[4] = {r1[ua].z, r1[ua].w, r1[ub].x, r1[ub].y}, l2[4] = {r2[ua].z, r2[ua].w, r2[ub].x, r2[ub].y};
            float outp[3][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float sums[3] = {l2[q] + v2[q], (l2[q] * l2[q]) + (v2[q] * v2[q]), (l1[q] * l2[q]) + (v1[q] * v2[q])};
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    float o1 = sums[p] * n2_0, o3 = sums[p] * n2_1, o5 = sums[p] * n2_2;
                    o1 = fmaf(-1.0f, pv2[p][0], o1); o3 = fmaf(-1.0f, pv2[p][1], o3); o5 = fmaf(-1.0f, pv2[p][2], o5);
                    pv2[p][0] = pv[p][0]; pv2[p][1] = pv[p][1]; pv2[p][2] = pv[p][2];
                    o1 = fmaf(mp_0, pv[p][0], o1); o3 = fmaf(mp_1, pv[p][1], o3); o5 = fmaf(mp_2, pv[p][2], o5);
                    pv[p][0] = o1; pv[p][1] = o3; pv[p][2] = o5;
                    outp[p][q] = o1 + o3 + o5;
                }
            }
            if (g >= 1 && valid) {
#pragma unroll
                for (int p = 0; p < 3; p++)
                    *reinterpret_cast<float4 *>(out + (size_t)p * W + ((g - 1) << 2)) = make_float4(outp[p][0], outp[p][1], outp[p][2], outp[p][3]);
            }
        }
    }
}

// ---- V pass + maps, resumed from B's checkpoint at the first changed row -------------------------------
// block = 256 threads = 256/W_s (candidate, channel) pairs; thread = one image column.
// Checkpoint record r (0 <= r <= H+4) of B: recurrence state before step n = r-4 and pooling sums of rows < r-4.
__global__ __launch_bounds__(256) void k_sparse_v(SparseParams P) {
    __shared__ float s_lut[3][256];
    __shared__ double red[256][6];
    const Geom &G = P.G;
    const int s = blockIdx.y;
    if (s >= G.nscales) return;
    const int W = G.sw[s], H = G.sh[s];
    const int t = threadIdx.x;
    const int ppw = 256 / W;
    const int npairs = P.is_base ? 3 : P.ncand * 3;
    if ((int)blockIdx.x * ppw >= npairs) return;
    const bool S0 = (s == 0);
    if (S0) {
        for (int i = t; i < 3 * 256; i += 256) { int c = i >> 8, j = i & 255; s_lut[c][j] = (j < P.ncol + 2) ? P.pal_xyb[3 * j + c] : 0.0f; }
        __syncthreads();
    }
    const int ql = t / W, x = t - ql * W;
    const int pair_raw = blockIdx.x * ppw + ql;
    const bool active = pair_raw < npairs;
    const int pair = active ? pair_raw : 0;
    const int k = P.is_base ? P.base : P.k0 + pair / 3, ch = pair % 3;
    const bool is_base = P.is_base != 0;
    const size_t ns = (size_t)W * H;
    const CandMeta *M = P.meta + k;
    const short *slot = M->slot + P.S.roff[s];
    const int nrows = M->nrows[s];
    const int r0 = nrows ? (int)M->rows[P.S.roff[s]] : H + 4;
    const int cmin = (M->xmin >> s) - 5; // columns <= cmin see only unchanged inputs
    const float cand_v = is_base ? 0.0f : P.cand_tab[8 * (size_t)k + 3 + ch];
    const uint32_t crgb = is_base ? 0u : __float_as_uint(P.cand_tab[8 * (size_t)k + 6]);
    const float *mine = P.store + (size_t)k * P.S.cand_stride;
    const float *basep = P.store + (size_t)P.base * P.S.cand_stride;
    const float *img1 = P.img1 + G.src_off[s] + (size_t)ch * ns, *mu1 = P.mu1 + G.src_off[s] + (size_t)ch * ns, *s11 = P.s11 + G.src_off[s] + (size_t)ch * ns;
    // checkpoint arrays: ckf[s][ch][r][18][W], cka[s][ch][r][6][W]
    float *ckf = P.ckf + P.S.off_ckf[s] + (size_t)ch * (H + 5) * 18 * W;
    double *cka = P.cka + P.S.off_cka[s] + (size_t)ch * (H + 5) * 6 * W;

    // does this wave have anything to recompute?  (a whole wave shares k unless W < 64; keep it per lane)
    const bool skip = !is_base && (nrows == 0 || x <= cmin);
    const int rs = skip ? H + 4 : (is_base ? 0 : r0);
    float pv[3][3], pv2[3][3];
    double acc[6];
    if (is_base) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { pv[p][q] = 0.0f; pv2[p][q] = 0.0f; }
#pragma unroll
        for (int q = 0; q < 6; q++) acc[q] = 0.0;
    } else {
        const float *cf = ckf + (size_t)rs * 18 * W + x;
        const double *ca = cka + (size_t)rs * 6 * W + x;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { pv[p][q] = cf[(size_t)(p * 6 + q) * W]; pv2[p][q] = cf[(size_t)(p * 6 + 3 + q) * W]; }
#pragma unroll
        for (int q = 0; q < 6; q++) acc[q] = ca[(size_t)q * W];
    }
    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
    // the block's pairs may start at different rows: run from the smallest start, lanes idle until their own
    for (int n = rs - 4; n < H; n++) {
        if (is_base && active) { // record r = n + 4
            float *cf = ckf + (size_t)(n + 4) * 18 * W + x;
            double *ca = cka + (size_t)(n + 4) * 6 * W + x;
#pragma unroll
            for (int p = 0; p < 3; p++)
#pragma unroll
                for (int q = 0; q < 3; q++) { cf[(size_t)(p * 6 + q) * W] = pv[p][q]; cf[(size_t)(p * 6 + 3 + q) * W] = pv2[p][q]; }
#pragma unroll
            for (int q = 0; q < 6; q++) ca[(size_t)q * W] = acc[q];
        }
        const int b = n + 4, tp = n - 6;
        float in[3] = {0.0f, 0.0f, 0.0f}, top[3] = {0.0f, 0.0f, 0.0f};
        if (b < H) {
            const short sl = slot[b];
            const float *row = sl >= 0 ? mine + P.S.off_hout[s] + (size_t)sl * 9 * W : basep + P.S.off_hout[s] + (size_t)b * 9 * W;
            in[0] = row[(size_t)(ch * 3) * W + x]; in[1] = row[(size_t)(ch * 3 + 1) * W + x]; in[2] = row[(size_t)(ch * 3 + 2) * W + x];
        }
        if (tp >= 0) {
            const short sl = slot[tp];
            const float *row = sl >= 0 ? mine + P.S.off_hout[s] + (size_t)sl * 9 * W : basep + P.S.off_hout[s] + (size_t)tp * 9 * W;
            top[0] = row[(size_t)(ch * 3) * W + x]; top[1] = row[(size_t)(ch * 3 + 1) * W + x]; top[2] = row[(size_t)(ch * 3 + 2) * W + x];
        }
        float outp[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const float sum = top[p] + in[p];
            float o1 = fmaf(pv[p][0], d1_0, pv2[p][0]);
            float o3 = fmaf(pv[p][1], d1_1, pv2[p][1]);
            float o5 = fmaf(pv[p][2], d1_2, pv2[p][2]);
            o1 = fmaf(sum, n2_0, -o1); o3 = fmaf(sum, n2_1, -o3); o5 = fmaf(sum, n2_2, -o5);
            pv2[p][0] = pv[p][0]; pv2[p][1] = pv[p][1]; pv2[p][2] = pv[p][2];
            pv[p][0] = o1; pv[p][1] = o3; pv[p][2] = o5;
            outp[p] = o1 + o3 + o5;
        }
        if (n >= 0) {
            const size_t idx = (size_t)n * W + x;
            const float m1 = mu1[idx], m2 = outp[0], v11 = s11[idx], v22 = outp[1], v12 = outp[2];
            const float i1 = img1[idx];
            float i2;
            if (S0) {
                const uint32_t ci = sparse_ci(P.pack[idx], crgb, (uint32_t)P.ncol, is_base);
                i2 = (ci == (uint32_t)P.ncol) ? cand_v : s_lut[ch][ci];
            } else {
                const short sl = slot[n];
                i2 = sl >= 0 ? mine[P.S.off_xyb[s] + (size_t)sl * 3 * W + (size_t)ch * W + x] : basep[P.S.off_xyb[s] + (size_t)n * 3 * W + (size_t)ch * W + x];
            }
            const float mu11 = m1 * m1, mu22 = m2 * m2, mu12 = m1 * m2;
            const float mu_diff = m1 - m2;
            const float num_m = fmaf(mu_diff, -mu_diff, 1.0f);
            const float num_s = fmaf(2.0f, v12 - mu12, 0.0009f);
            const float denom_s = (v11 - mu11) + (v22 - mu22) + 0.0009f;
            double d = 1.0 - (double)((num_m * num_s) / denom_s);
            d = d > 0.0 ? d : 0.0;
            acc[0] += d;
            const double dd = d * d;
            acc[1] += dd * dd;
            const double d1 = (1.0 + (double)fabsf(i2 - m2)) / (1.0 + (double)fabsf(i1 - m1)) - 1.0;
            const double art = d1 > 0.0 ? d1 : 0.0;
            const double det = (-d1) > 0.0 ? (-d1) : 0.0;
            acc[2] += art;
            const double a2 = art * art;
            acc[3] += a2 * a2;
            acc[4] += det;
            const double l2 = det * det;
            acc[5] += l2 * l2;
        }
    }
    if (is_base && active) { // final record r = H + 4
        double *ca = cka + (size_t)(H + 4) * 6 * W + x;
#pragma unroll
        for (int q = 0; q < 6; q++) ca[(size_t)q * W] = acc[q];
    }
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

// contested pixels of a slot (thr != 0) -> compact list, built once per slot by k_prep's companion
__global__ __launch_bounds__(256) void k_build_plist(const unsigned long long *__restrict__ pack, int npx, uint4 *__restrict__ plist, int *__restrict__ count) {
    int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npx) return;
    const unsigned long long w = pack[px];
    const uint32_t thr = (uint32_t)(w >> 32);
    if (thr != 0u) {
        const int i = atomicAdd(count, 1);
        plist[i] = make_uint4((uint32_t)px, (uint32_t)w & 0x00ffffffu, thr, 0u);
    }
}

} // namespace snes
