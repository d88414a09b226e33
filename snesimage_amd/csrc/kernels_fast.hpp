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
//   * the 10-deep delay line of the truncated-cosine filter (in[n-6]) is a 3-group register ring.
#pragma once
#include "kernels.hpp"

namespace snes {

struct FastParams {
    Geom G; BlurK K;
    int s, npairs, ncol;
    const unsigned long long *packC4, *packR4; // scale 0
    const float *pal_xyb, *cand_tab;
    const float *img1C4;                       // source, this scale: [3][C4]
    const float *img1R4, *mu1R4, *s11R4;       // source, this scale: [3][R4]
    float *work; double *part;
};

// colour index of 4 pixels from their pack words (RGB / redmean keys)
__device__ __forceinline__ void resolve4(const uint4 a, const uint4 b, uint32_t crgb, uint32_t ncol, uint32_t ci[4]) {
    const uint32_t lo[4] = {a.x, a.z, b.x, b.z}, hi[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t ci0 = lo[j] >> 24;
        ci[j] = (hi[j] != 0u && red_mean_key(crgb, lo[j] & 0x00ffffffu) < hi[j]) ? ncol : ci0;
    }
}

// ---- H pass: one wave = rows [64*yb, 64*yb+64) of one (candidate, channel) -------------------------
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
    const float4 *in1 = reinterpret_cast<const float4 *>(P.img1C4 + (size_t)ch * ns) + y;        // + g*H
    const float4 *in2 = S0 ? nullptr : reinterpret_cast<const float4 *>(P.work + (size_t)cand * G.cand_stride + G.off_xybT[s] + (size_t)ch * ns) + y;
    const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.packC4) + 2 * (size_t)y : nullptr;     // + g*H*2
    float *hout = P.work + (size_t)cand * G.cand_stride + G.off_hout[s] + (size_t)(ch * 3) * ns;

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
    float pv[3][3], pv2[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) { pv[p][k] = 0.0f; pv2[p][k] = 0.0f; }
    float r1[4][4], r2[4][4]; // ring of the last four groups: [slot][element]
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) { r1[a][b] = 0.0f; r2[a][b] = 0.0f; }

    const int G4 = W >> 2;
    float4 n_v1 = in1[0];
    float4 n_v2 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 n_pa = make_uint4(0, 0, 0, 0), n_pb = make_uint4(0, 0, 0, 0);
    if (S0) { n_pa = pk[0]; n_pb = pk[1]; } else n_v2 = in2[0];

    for (int g0 = 0; g0 <= G4; g0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = g0 + u;
            if (g > G4) break;
            // take the prefetched group, issue the next group's loads
            const float4 c_v1 = n_v1; float4 c_v2 = n_v2; const uint4 c_pa = n_pa, c_pb = n_pb;
            if (g + 1 < G4) {
                n_v1 = in1[(size_t)(g + 1) * H];
                if (S0) { n_pa = pk[(size_t)(g + 1) * H * 2]; n_pb = pk[(size_t)(g + 1) * H * 2 + 1]; } else n_v2 = in2[(size_t)(g + 1) * H];
            } else { n_v1 = make_float4(0.f, 0.f, 0.f, 0.f); n_v2 = n_v1; n_pa = make_uint4(0, 0, 0, 0); n_pb = n_pa; }
            float v1[4] = {c_v1.x, c_v1.y, c_v1.z, c_v1.w};
            float v2[4];
            if (g >= G4) { v1[0] = v1[1] = v1[2] = v1[3] = 0.0f; v2[0] = v2[1] = v2[2] = v2[3] = 0.0f; }
            else if (S0) {
                uint32_t ci[4];
                resolve4(c_pa, c_pb, crgb, (uint32_t)P.ncol, ci);
#pragma unroll
                for (int j = 0; j < 4; j++) v2[j] = (ci[j] == (uint32_t)P.ncol) ? cand_v : s_lut[ci[j]];
            } else { v2[0] = c_v2.x; v2[1] = c_v2.y; v2[2] = c_v2.z; v2[3] = c_v2.w; }
#pragma unroll
            for (int j = 0; j < 4; j++) { r1[u][j] = v1[j]; r2[u][j] = v2[j]; }
            float outp[3][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // in[xr - 10]: group g-3 (slot u+1) elements 2,3 for j = 0,1; group g-2 (slot u+2) elements 0,1 for j = 2,3
                const float l1v = j < 2 ? r1[(u + 1) & 3][j + 2] : r1[(u + 2) & 3][j - 2];
                const float l2v = j < 2 ? r2[(u + 1) & 3][j + 2] : r2[(u + 2) & 3][j - 2];
                const float sums[3] = {l2v + v2[j], (l2v * l2v) + (v2[j] * v2[j]), (l1v * l2v) + (v1[j] * v2[j])};
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    float o1 = sums[p] * n2_0, o3 = sums[p] * n2_1, o5 = sums[p] * n2_2;
                    o1 = fmaf(-1.0f, pv2[p][0], o1); o3 = fmaf(-1.0f, pv2[p][1], o3); o5 = fmaf(-1.0f, pv2[p][2], o5);
                    pv2[p][0] = pv[p][0]; pv2[p][1] = pv[p][1]; pv2[p][2] = pv[p][2];
                    o1 = fmaf(mp_0, pv[p][0], o1); o3 = fmaf(mp_1, pv[p][1], o3); o5 = fmaf(mp_2, pv[p][2], o5);
                    pv[p][0] = o1; pv[p][1] = o3; pv[p][2] = o5;
                    outp[p][j] = o1 + o3 + o5;
                }
            }
            if (g >= 1) { // outputs x = 4(g-1) .. 4(g-1)+3 of row y: transpose 4x4 across each lane quad, store XT4
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int j = 0; j < 4; j++) s_tr[p][lane * 5 + j] = outp[p][j];
                __syncthreads();
                const int i = lane & 3, k4 = lane & ~3;
                const int x = ((g - 1) << 2) + i;
                const size_t o = (size_t)idx_xt4(x, (yb << 6) + k4, H);
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    float4 v;
                    v.x = s_tr[p][(k4 + 0) * 5 + i]; v.y = s_tr[p][(k4 + 1) * 5 + i]; v.z = s_tr[p][(k4 + 2) * 5 + i]; v.w = s_tr[p][(k4 + 3) * 5 + i];
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
    const int H4 = H >> 2;
    // XT4: float4 index ((xb*H4 + g) << 6) + lane
    const float4 *hout = reinterpret_cast<const float4 *>(P.work + (size_t)cand * G.cand_stride + G.off_hout[s] + (size_t)(ch * 3) * ns) + (((size_t)(x >> 6) * H4) << 6) + (x & 63);
    const size_t plane4 = ns >> 2;
    // R4: float4 index g*W + x
    const float4 *img1 = reinterpret_cast<const float4 *>(P.img1R4 + (size_t)ch * ns) + x;
    const float4 *mu1 = reinterpret_cast<const float4 *>(P.mu1R4 + (size_t)ch * ns) + x;
    const float4 *s11 = reinterpret_cast<const float4 *>(P.s11R4 + (size_t)ch * ns) + x;
    const float4 *xyb = S0 ? nullptr : reinterpret_cast<const float4 *>(P.work + (size_t)cand * G.cand_stride + G.off_xyb[s] + (size_t)ch * ns) + x;
    const uint4 *pk = S0 ? reinterpret_cast<const uint4 *>(P.packR4) + 2 * (size_t)x : nullptr; // + g*W*2

    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
    float pv[3][3], pv2[3][3];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int k = 0; k < 3; k++) { pv[p][k] = 0.0f; pv2[p][k] = 0.0f; }
    float ring[3][4][4]; // [plane][slot][element]
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) ring[p][a][b] = 0.0f;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    float4 n_h[3];
#pragma unroll
    for (int p = 0; p < 3; p++) n_h[p] = hout[(size_t)p * plane4];
    // map inputs of row group g-1 are fetched one group ahead as well
    float4 n_i1 = make_float4(0.f, 0.f, 0.f, 0.f), n_m1 = n_i1, n_s11 = n_i1, n_x = n_i1;
    uint4 n_pa = make_uint4(0, 0, 0, 0), n_pb = n_pa;

    for (int g0 = 0; g0 <= H4; g0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = g0 + u;
            if (g > H4) break;
            float4 c_h[3];
#pragma unroll
            for (int p = 0; p < 3; p++) c_h[p] = n_h[p];
            const float4 c_i1 = n_i1, c_m1 = n_m1, c_s11 = n_s11, c_x = n_x; const uint4 c_pa = n_pa, c_pb = n_pb;
            if (g + 1 < H4) {
#pragma unroll
                for (int p = 0; p < 3; p++) n_h[p] = hout[(size_t)p * plane4 + ((size_t)(g + 1) << 6)];
            } else {
#pragma unroll
                for (int p = 0; p < 3; p++) n_h[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (g < H4) { // inputs of the maps of rows 4g..4g+3, consumed in the next iteration
                n_i1 = img1[(size_t)g * W]; n_m1 = mu1[(size_t)g * W]; n_s11 = s11[(size_t)g * W];
                if (S0) { n_pa = pk[(size_t)g * W * 2]; n_pb = pk[(size_t)g * W * 2 + 1]; } else n_x = xyb[(size_t)g * W];
            }
            float in[3][4];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                if (g >= H4) { in[p][0] = in[p][1] = in[p][2] = in[p][3] = 0.0f; }
                else { in[p][0] = c_h[p].x; in[p][1] = c_h[p].y; in[p][2] = c_h[p].z; in[p][3] = c_h[p].w; }
#pragma unroll
                for (int j = 0; j < 4; j++) ring[p][u][j] = in[p][j];
            }
            float i2v[4] = {c_x.x, c_x.y, c_x.z, c_x.w};
            if (S0 && g >= 1) {
                uint32_t ci[4];
                resolve4(c_pa, c_pb, crgb, (uint32_t)P.ncol, ci);
#pragma unroll
                for (int j = 0; j < 4; j++) i2v[j] = (ci[j] == (uint32_t)P.ncol) ? cand_v : s_lut[ch][ci[j]];
            }
            const float i1v[4] = {c_i1.x, c_i1.y, c_i1.z, c_i1.w}, m1v[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w}, s11v[4] = {c_s11.x, c_s11.y, c_s11.z, c_s11.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float outp[3];
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    const float top = j < 2 ? ring[p][(u + 1) & 3][j + 2] : ring[p][(u + 2) & 3][j - 2];
                    const float sum = top + in[p][j];
                    float o1 = fmaf(pv[p][0], d1_0, pv2[p][0]);
                    float o3 = fmaf(pv[p][1], d1_1, pv2[p][1]);
                    float o5 = fmaf(pv[p][2], d1_2, pv2[p][2]);
                    o1 = fmaf(sum, n2_0, -o1); o3 = fmaf(sum, n2_1, -o3); o5 = fmaf(sum, n2_2, -o5);
                    pv2[p][0] = pv[p][0]; pv2[p][1] = pv[p][1]; pv2[p][2] = pv[p][2];
                    pv[p][0] = o1; pv[p][1] = o3; pv[p][2] = o5;
                    outp[p] = o1 + o3 + o5;
                }
                if (g >= 1 && active) { // row n = 4(g-1) + j
                    const float m1 = m1v[j], m2 = outp[0], v11 = s11v[j], v22 = outp[1], v12 = outp[2];
                    const float i1 = i1v[j], i2 = i2v[j];
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

} // namespace snes
