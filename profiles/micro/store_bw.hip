// profiles/micro/store_bw.hip — what HBM takes in stores on MI355X: the roofline of k_sparse_h2 / k_sparse_down1, which write what
// the next kernel reads (1.07 GB and 0.2 GB per 4,096 candidates) and read next to nothing.
//   hipcc -O3 --offload-arch=gfx950 -o store_bw store_bw.hip && ./store_bw
// Variants: every lane one float4, lanes and waves contiguous (a fill); the same with each wave's 1 KiB going to a scattered 1 KiB
// slot (the H pass's flush: whole lines, but a group's planes lie 4 W floats apart); dword stores, lanes contiguous (the downscale's).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fill16(float4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_fill16_scattered(float4 *p, size_t n4) { // wave w of the grid writes the 1 KiB slot (w * 7919) mod slots
    const size_t slots = n4 / 64;
    for (size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6; w < slots; w += ((size_t)gridDim.x * 256) >> 6) {
        const size_t s = (w * 7919u) % slots;
        p[s * 64 + (threadIdx.x & 63)] = make_float4(1.f, 2.f, 3.f, (float)w);
    }
}
// CH lanes (16 B each) write one contiguous chunk; the chunks of a wave — and of consecutive waves — go to scattered places (the H
// pass's flush: 8 lanes per 128-byte line, a store instruction = eight lines of eight different planes / rows)
template <int CH>
__global__ __launch_bounds__(256) void k_fill16_chunks(float4 *p, size_t n4) {
    const size_t chunks = n4 / CH;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n4; t += (size_t)gridDim.x * 256) {
        const size_t c = t / CH, l = t % CH;
        const size_t s = (c * 7919u) % chunks; // (7919 is prime and chunks a power of two: a permutation)
        p[s * CH + l] = make_float4(1.f, 2.f, 3.f, (float)t);
    }
}
__global__ __launch_bounds__(256) void k_fill4(float *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (float)i;
}
__global__ __launch_bounds__(256) void k_copy16(const float4 *__restrict__ q, float4 *__restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = q[i];
}
__global__ __launch_bounds__(256) void k_read16(const float4 *__restrict__ q, float *out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = q[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30; // 1 GiB: what the H pass writes per 4,096 candidates
    float4 *p, *q; float *o;
    CHECK(hipMalloc(&p, bytes)); CHECK(hipMalloc(&q, bytes)); CHECK(hipMalloc(&o, 4));
    CHECK(hipMemset(q, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t n4 = bytes / 16;
    for (int grid : {2048, 8192, 32768}) {
        for (int v = 0; v < 9; v++) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipEventRecord(e0));
                if (v == 0) hipLaunchKernelGGL(k_fill16, dim3(grid), dim3(256), 0, 0, p, n4);
                if (v == 1) hipLaunchKernelGGL(k_fill16_scattered, dim3(grid), dim3(256), 0, 0, p, n4);
                if (v == 2) hipLaunchKernelGGL(k_fill4, dim3(grid), dim3(256), 0, 0, (float *)p, n4 * 4);
                if (v == 3) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, q, p, n4);
                if (v == 4) hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, q, o, n4);
                if (v == 5) hipLaunchKernelGGL(k_fill16_chunks<4>, dim3(grid), dim3(256), 0, 0, p, n4);
                if (v == 6) hipLaunchKernelGGL(k_fill16_chunks<8>, dim3(grid), dim3(256), 0, 0, p, n4);
                if (v == 7) hipLaunchKernelGGL(k_fill16_chunks<16>, dim3(grid), dim3(256), 0, 0, p, n4);
                if (v == 8) hipLaunchKernelGGL(k_fill16_chunks<32>, dim3(grid), dim3(256), 0, 0, p, n4);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const char *names[] = {"fill, float4 per lane, contiguous", "fill, float4 per lane, a scattered 1 KiB slot per wave", "fill, dword per lane, contiguous", "copy, float4 per lane (1 GiB read + 1 GiB written)", "read, float4 per lane", "fill, scattered chunks of 64 B", "fill, scattered chunks of 128 B (the H pass's lines)", "fill, scattered chunks of 256 B", "fill, scattered chunks of 512 B"};
            printf("grid %6d  %-58s %8.1f us  %6.2f TB/s%s\n", grid, names[v], best * 1e3, (v == 3 ? 2.0 : 1.0) * bytes / (best * 1e-3) / 1e12, v == 3 ? " (read + write)" : "");
        }
    }
    return 0;
}
