// snesimage_amd/csrc/kernels_narrow.hpp — the scales narrower than 64 pixels (32, 16 and 8 wide at the BASELINE size: 1.6 %
// of the pyramid's pixels) of one candidate in ONE block, planes in LDS: downscale + XYB, H pass, V pass + maps + pooling.
//
// Until round 4 these scales went the way of the wide ones — changed groups only, resumed from the base image B's
// checkpoints — through four launches of the general bodies (sparse_down_body's last three scales behind a barrier each,
// sparse_h_body, sparse_v_body; for B sparse_h_body and sparse_v_body<.., 1>): 95 us of a 1.47 ms step for the candidates,
// and for B two single-image launches plus a cross-stream hand-off in the fixed chain of every call and every slot window.
// At these sizes sparsity buys little (a group of the 32-wide scale is 32 rows of the image: most candidates change most
// groups) and costs the round trips through memory between the stages and the dependency on B's sweeps.  Here a block
// of three waves (wave = channel) recomputes the narrow scales of its candidate densely:
//   D  the linear planes: a changed group from the scale above (the widest narrow scale reads the candidate's — or, for an
//      unchanged group above, B's — linear rows of the last wide scale from memory; the others read LDS), an unchanged group
//      as B's planes (B's downscale, base_down_body, leaves them); XYB by the same linear_to_positive_xyb;
//   H  lane = row: lanes 0-31 the 32-wide scale, 32-47 the 16-wide, 48-55 the 8-wide, all three sweeping x together
//      (horizontal_row's taps and recurrences operation for operation), outputs to LDS;
//   V  lane = column, the same lane ranges: vertical_pass + ssim_map + edge_diff_map + the six pooling sums per column
//      (maps_accumulate, row after row from the top), then the column tree of the other V bodies (stride W/2, W/4, ...).
// Nothing of B's sweeps is read: the launch depends on the scan, the downscale of the last wide scale and B's downscale
// only, and B needs no narrow sweeps at all.  Dense and checkpoint-resumed evaluation are the same arithmetic on the same
// inputs in the same order: bit-identical (tests: sparse == dense at every height, trajectories, goldens).
#pragma once
#include "kernels_sparse2.hpp"

namespace snes {

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

constexpr int kNarrowMax = 3;  // scales narrower than 64 pixels at W = 256: 32, 16, 8
constexpr int kNarrowPad = 4;  // floats of padding per LDS row: rows stay 16-byte aligned and a lane-per-row access is conflict-free
struct NarrowLayout { int off[kNarrowMax + 1]; }; // float offset of every narrow scale inside one padded plane set
__host__ __device__ inline NarrowLayout narrow_layout(const Geom &G, int s_first) {
    NarrowLayout L; int o = 0;
    for (int j = 0; j < kNarrowMax; j++) { L.off[j] = o; const int s = s_first + j; if (s < G.nscales) o += G.sh[s] * (G.sw[s] + kNarrowPad); }
    L.off[kNarrowMax] = o;
    return L;
}
// dynamic LDS of a block: XYB (3 channels) + H output (3 channels x 3 planes; the linear planes of phase D lie in the same space)
__host__ __device__ inline size_t narrow_lds_bytes(const Geom &G, int s_first) { return sizeof(float) * 12 * (size_t)narrow_layout(G, s_first).off[kNarrowMax]; }

__device__ __forceinline__ void sparse_narrow_body(const SparseParams &P, const int bx) { // bx: the candidate of the launch this block takes
    extern __shared__ __attribute__((aligned(16))) float s_nar[];
    __shared__ short s_gs[kNarrowMax + 1][64]; // group -> slot of the last wide scale (row 0) and of the narrow scales
    const Geom &G = P.G;
    if (bx >= P.ncand) return; // (a batched launch is sized for its longest member)
    const int t = threadIdx.x, lane = t & 63, ch = t >> 6; // 192 threads: wave = channel in phases H and V
    const int k = P.k0 + bx;
    const int s3 = P.s_first, nn = G.nscales - s3;
    const NarrowLayout L = narrow_layout(G, s3);
    const int PS = L.off[kNarrowMax]; // floats per padded plane set
    float *const s_xyb = s_nar;            // [3 ch][PS]
    float *const s_hout = s_nar + 3 * PS;  // [3 ch][3 planes][PS]
    float *const s_lin = s_hout;           // phase D only: [3 ch][PS]
    const CandMeta *M = P.meta + k;
    const float *mine = P.store + (size_t)k * P.S.cand_stride, *basep = P.store + (size_t)P.base * P.S.cand_stride;
    for (int i = t; i < (kNarrowMax + 1) * 64; i += 192) {
        const int row = i >> 6, g = i & 63, s = s3 - 1 + row;
        s_gs[row][g] = (s < G.nscales && g < (G.sh[s] >> 2)) ? M->gslot[P.S.goff[s] + g] : (short)-1;
    }
    __syncthreads();

    // ---- D: linear RGB and XYB of every narrow scale into LDS ---------------------------------------------------------
    for (int j = 0; j < nn; j++) {
        const int s = s3 + j, Ws = G.sw[s], Hs = G.sh[s], Wp = G.sw[s - 1], Q = Ws >> 2, RS = Ws + kNarrowPad;
        for (int i = t; i < Hs * Q; i += 192) { // a thread = four consecutive pixels of a row
            const int xq = i % Q, y = i / Q, g = y >> 2, r = y & 3;
            float4 lin4[3], xyb4[3];
            if (s_gs[1 + j][g] < 0) { // unchanged: B's planes (C4 inside the group: [x/4][row][x%4])
                const size_t o = (size_t)g * 12 * Ws + (size_t)xq * 16 + r * 4;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    lin4[c] = (j + 1 < nn) ? *reinterpret_cast<const float4 *>(basep + P.S.off_lin[s] + o + (size_t)c * 4 * Ws) : make_float4(0.f, 0.f, 0.f, 0.f); // (B keeps no linear rows of its last scale: nobody downscales them)
                    xyb4[c] = *reinterpret_cast<const float4 *>(basep + P.S.off_xybC[s] + o + (size_t)c * 4 * Ws);
                }
            } else { // changed: 2 x 2 box of the scale above in linear RGB (downscale_by_2), then XYB
                float v[3][4];
                if (j == 0) { // rows 2y, 2y+1 of the last wide scale live in one group: the candidate's own if it changed, else B's
                    const int gp = (2 * y) >> 2, rp = (2 * y) & 3;
                    const short sl = s_gs[0][gp];
                    const float *grp = sl >= 0 ? mine + P.S.off_lin[s - 1] + (size_t)sl * 12 * Wp : basep + P.S.off_lin[s - 1] + (size_t)gp * 12 * Wp;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float *q0 = grp + (size_t)c * 4 * Wp + (size_t)(2 * xq) * 16 + rp * 4;
                        const float4 a0 = *reinterpret_cast<const float4 *>(q0), a1 = *reinterpret_cast<const float4 *>(q0 + 4);           // column quad 2xq, rows 2y and 2y+1
                        const float4 b0 = *reinterpret_cast<const float4 *>(q0 + 16), b1 = *reinterpret_cast<const float4 *>(q0 + 20);     // column quad 2xq+1
                        float sm;
                        sm = 0.0f; sm += a0.x; sm += a0.y; sm += a1.x; sm += a1.y; v[c][0] = sm * 0.25f;
                        sm = 0.0f; sm += a0.z; sm += a0.w; sm += a1.z; sm += a1.w; v[c][1] = sm * 0.25f;
                        sm = 0.0f; sm += b0.x; sm += b0.y; sm += b1.x; sm += b1.y; v[c][2] = sm * 0.25f;
                        sm = 0.0f; sm += b0.z; sm += b0.w; sm += b1.z; sm += b1.w; v[c][3] = sm * 0.25f;
                    }
                } else {
                    const int RSp = Wp + kNarrowPad;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float *q0 = s_lin + c * PS + L.off[j - 1] + (2 * y) * RSp + 8 * xq;
                        const float4 a0 = *reinterpret_cast<const float4 *>(q0), b0 = *reinterpret_cast<const float4 *>(q0 + 4);             // row 2y, columns 8xq .. 8xq+7
                        const float4 a1 = *reinterpret_cast<const float4 *>(q0 + RSp), b1 = *reinterpret_cast<const float4 *>(q0 + RSp + 4); // row 2y+1
                        float sm;
                        sm = 0.0f; sm += a0.x; sm += a0.y; sm += a1.x; sm += a1.y; v[c][0] = sm * 0.25f;
                        sm = 0.0f; sm += a0.z; sm += a0.w; sm += a1.z; sm += a1.w; v[c][1] = sm * 0.25f;
                        sm = 0.0f; sm += b0.x; sm += b0.y; sm += b1.x; sm += b1.y; v[c][2] = sm * 0.25f;
                        sm = 0.0f; sm += b0.z; sm += b0.w; sm += b1.z; sm += b1.w; v[c][3] = sm * 0.25f;
                    }
                }
                float xv[3][4];
#pragma unroll
                for (int q = 0; q < 4; q++) linear_to_positive_xyb(v[0][q], v[1][q], v[2][q], xv[0][q], xv[1][q], xv[2][q]);
#pragma unroll
                for (int c = 0; c < 3; c++) { lin4[c] = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]); xyb4[c] = make_float4(xv[c][0], xv[c][1], xv[c][2], xv[c][3]); }
            }
            const int o = L.off[j] + y * RS + 4 * xq;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                *reinterpret_cast<float4 *>(s_lin + c * PS + o) = lin4[c];
                *reinterpret_cast<float4 *>(s_xyb + c * PS + o) = xyb4[c];
            }
        }
        __syncthreads(); // the next scale reads these rows; after the last scale the linear planes are dead (phase H writes over them)
    }

    // ---- lane -> (scale, row | column) of phases H and V: lanes 0-31 / 32-47 / 48-55 --------------------------------------
    const int j = lane < 32 ? 0 : (lane < 48 ? 1 : (lane < 56 ? 2 : 3));
    const int li = lane - (j == 0 ? 0 : (j == 1 ? 32 : 48)); // row (H) or column (V) of the lane inside its scale
    const int sj = s3 + (j < nn ? j : 0);
    const int Wj = G.sw[sj], Hj = G.sh[sj], RSj = Wj + kNarrowPad, offj = L.off[j < nn ? j : 0];
    const size_t nsj = (size_t)Wj * Hj;
    const float n2_0 = P.K.n2[0], n2_1 = P.K.n2[1], n2_2 = P.K.n2[2];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- H: lane = row; iteration g consumes column quad g and yields the outputs of quad g-1 ----------------------------
    {
        const float mp_0 = -P.K.d1[0], mp_1 = -P.K.d1[1], mp_2 = -P.K.d1[2];
        const bool act = j < nn && li < Hj;
        const int y = act ? li : 0, G4 = Wj >> 2;
        const float4 *in1 = reinterpret_cast<const float4 *>(P.img1C4 + G.src_off[sj] + (size_t)ch * nsj) + y; // C4: + g*H
        const float *xrow = s_xyb + ch * PS + offj + y * RSj;
        float *orow = s_hout + (ch * 3) * PS + offj + y * RSj;
        float4 q1[8]; // the row of the source plane: at most 8 column quads
#pragma unroll
        for (int g = 0; g < 8; g++) q1[g] = (act && g < G4) ? in1[(size_t)g * Hj] : zero4;
        float sa[3][3], sb[3][3];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = 0.0f; sb[p][q] = 0.0f; }
        float4 r1[4], r2[4]; // quads g-3 .. g in slots (g-3)&3 .. g&3
#pragma unroll
        for (int a = 0; a < 4; a++) { r1[a] = zero4; r2[a] = zero4; }
#pragma unroll
        for (int g = 0; g <= 8; g++) { // (the widest narrow scale is 32 pixels wide at W = 256: eight quads and the flush)
            const int u = g & 3, ua = (g + 1) & 3, ub = (g + 2) & 3; // slots of quads g, g-3, g-2
            r1[u] = g < 8 ? q1[g < 8 ? g : 0] : zero4;
            r2[u] = (act && g < G4) ? *reinterpret_cast<const float4 *>(xrow + 4 * g) : zero4;
            if (g >= G4) r1[u] = zero4;
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
            if (g >= 1 && act && g - 1 < G4) {
#pragma unroll
                for (int p = 0; p < 3; p++) *reinterpret_cast<float4 *>(orow + p * PS + 4 * (g - 1)) = make_float4(outp[p][0], outp[p][1], outp[p][2], outp[p][3]);
            }
        }
    }
    __syncthreads();

    // ---- V: lane = column; iteration g consumes row group g and yields the outputs of group g-1 -------------------------
    {
        const float d1_0 = P.K.d1[0], d1_1 = P.K.d1[1], d1_2 = P.K.d1[2];
        const bool act = j < nn; // (li < Wj by construction: W = 256 makes the narrow scales 32, 16 and 8 wide)
        const int x = act ? li : 0, H4 = Hj >> 2;
        const int H4max = G.sh[s3] >> 2;
        const float *hcol = s_hout + (ch * 3) * PS + offj + x;
        const float *xcol = s_xyb + ch * PS + offj + x;
        const float4 *mu1 = reinterpret_cast<const float4 *>(P.mu1R4 + G.src_off[sj] + (size_t)ch * nsj) + x; // R4: + g*W
        const float4 *sd1 = reinterpret_cast<const float4 *>(P.sd1R4 + G.src_off[sj] + (size_t)ch * nsj) + x;
        const float4 *a1 = reinterpret_cast<const float4 *>(P.a1R4 + G.src_off[sj] + (size_t)ch * nsj) + x;
        const double2 *r1p = reinterpret_cast<const double2 *>(P.r1R4 + G.src_off[sj] + (size_t)ch * nsj) + 2 * (size_t)x; // + g*2W
        float sa[3][3], sb[3][3];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 3; q++) { sa[p][q] = 0.0f; sb[p][q] = 0.0f; }
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        float4 ring[3][3]; // [plane][group mod 3]
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int a = 0; a < 3; a++) ring[p][a] = zero4;
        // the source side of row group 0, fetched an iteration ahead of the maps that consume it
        float4 n_m1 = zero4, n_sd1 = zero4, n_a1 = zero4; double2 n_ra = make_double2(1.0, 1.0), n_rb = n_ra;
        if (act) { n_m1 = mu1[0]; n_sd1 = sd1[0]; n_a1 = a1[0]; n_ra = r1p[0]; n_rb = r1p[1]; }
#define SNES_NARROW_VITER(T) /* T = g mod 3: the ring slot of group g, which is also group g-3's */                      \
        {                                                                                                                     \
            constexpr int u_ = (T) % 3, ux_ = ((T) + 1) % 3; /* slots of groups g (and g-3), g-2 */                           \
            float4 cur[3];                                                                                                    \
            _Pragma("unroll") for (int p = 0; p < 3; p++) {                                                                   \
                cur[p] = zero4;                                                                                               \
                if (act && g < H4) {                                                                                          \
                    const float *h = hcol + p * PS + (4 * g) * RSj;                                                           \
                    cur[p] = make_float4(h[0], h[RSj], h[2 * RSj], h[3 * RSj]);                                               \
                }                                                                                                             \
            }                                                                                                                 \
            const float4 c_m1 = n_m1, c_sd1 = n_sd1, c_a1 = n_a1; const double2 c_ra = n_ra, c_rb = n_rb; /* source side of group g-1 */ \
            if (act && g < H4 && g >= 1) { n_m1 = mu1[(size_t)g * Wj]; n_sd1 = sd1[(size_t)g * Wj]; n_a1 = a1[(size_t)g * Wj]; n_ra = r1p[(size_t)g * 2 * Wj]; n_rb = r1p[(size_t)g * 2 * Wj + 1]; } \
            float outp[3][4];                                                                                                 \
            _Pragma("unroll") for (int p = 0; p < 3; p++) {                                                                   \
                const float4 tz = ring[p][u_], tx = ring[p][ux_]; /* group g-3 (second half used), group g-2 (first half used) */ \
                SNES_VSTEP(tz.z + cur[p].x, sa[p], sb[p], outp[p][0])                                                         \
                SNES_VSTEP(tz.w + cur[p].y, sb[p], sa[p], outp[p][1])                                                         \
                SNES_VSTEP(tx.x + cur[p].z, sa[p], sb[p], outp[p][2])                                                         \
                SNES_VSTEP(tx.y + cur[p].w, sb[p], sa[p], outp[p][3])                                                         \
                ring[p][u_] = cur[p];                                                                                         \
            }                                                                                                                 \
            if (g >= 1 && act && g - 1 < H4) {                                                                                \
                const float *xr = xcol + (4 * (g - 1)) * RSj;                                                                 \
                const float i2v[4] = {xr[0], xr[RSj], xr[2 * RSj], xr[3 * RSj]};                                              \
                const float m1v[4] = {c_m1.x, c_m1.y, c_m1.z, c_m1.w}, sd1v[4] = {c_sd1.x, c_sd1.y, c_sd1.z, c_sd1.w}, a1v[4] = {c_a1.x, c_a1.y, c_a1.z, c_a1.w}; \
                const double r1v[4] = {c_ra.x, c_ra.y, c_rb.x, c_rb.y};                                                       \
                _Pragma("unroll") for (int q = 0; q < 4; q++) maps_accumulate(acc, m1v[q], sd1v[q], a1v[q], r1v[q], outp[0][q], outp[1][q], outp[2][q], i2v[q]); \
            }                                                                                                                 \
        }
        for (int g0 = 0; g0 <= H4max; g0 += 3) { // (H4max + 1 iterations: 2, 3, 5 or 9)
            { const int g = g0; SNES_NARROW_VITER(0) }
            { const int g = g0 + 1; if (g > H4max) break; SNES_NARROW_VITER(1) }
            { const int g = g0 + 2; if (g > H4max) break; SNES_NARROW_VITER(2) }
        }
#undef SNES_NARROW_VITER
        // the column tree of the other V bodies: stride W/2, W/4, ... inside the scale's lanes
#pragma unroll
        for (int stride = 16; stride > 0; stride >>= 1) {
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const double o = __shfl_down(acc[q], (unsigned)stride, 64);
                if (stride < Wj && li < stride) acc[q] += o;
            }
        }
        if (act && li == 0) {
            double *o = P.part + (((size_t)k * G.nscales + sj) * 3 + ch) * 6;
#pragma unroll
            for (int q = 0; q < 6; q++) o[q] = acc[q];
        }
    }
}
#undef SNES_HSTEP
#undef SNES_VSTEP

__global__ __launch_bounds__(192) void k_sparse_narrow(SparseParams P) { sparse_narrow_body(P, (int)blockIdx.x); }

} // namespace snes
