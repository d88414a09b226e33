// profiles/micro/stream_handover.hip — what a hand-over between two HIP streams costs, pair by pair, among S streams created in a row.
// The runtime deals streams to GPU_MAX_HW_QUEUES hardware queues (default 4); DESIGN 5 found a step's duration hanging on which of the
// library's streams share one.  For every pair (a, b) of the S streams: N round trips  kernel on a -> event -> wait on b -> kernel on b ->
// event -> wait on a, host time per hand-over (a one-wave kernel's own launch-to-launch time on one stream is printed first).
// Build: hipcc -O2 --offload-arch=gfx950 -o stream_handover stream_handover.hip      Run: GPU_MAX_HW_QUEUES=<q> ./stream_handover [S] [N] [idle]
// `idle` streams are created first and never used (the second launch lane's stream was one such).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_touch(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 6, N = argc > 2 ? atoi(argv[2]) : 200, idle = argc > 3 ? atoi(argv[3]) : 0;
    if (S < 2 || S > 16 || N < 1 || N > 5000 || idle < 0 || idle > 16) { fprintf(stderr, "usage: stream_handover [S 2..16] [N 1..5000] [idle 0..16]\n"); return 2; }
    int *d = nullptr; CHK(hipMalloc(&d, sizeof(int))); CHK(hipMemset(d, 0, sizeof(int)));
    std::vector<hipStream_t> idlers(idle), st(S);
    for (auto &s : idlers) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ea, eb; CHK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    for (auto &s : st) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d); CHK(hipStreamSynchronize(s)); } // every stream has its queue
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s, %d streams (+%d idle ones created first), %d round trips per pair\n", q ? q : "default", S, idle, N);
    { auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2 * N; i++) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[0], d);
      CHK(hipStreamSynchronize(st[0]));
      printf("one stream, launch to launch: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * N)); }
    printf("us per hand-over, row a -> column b:\n     ");
    for (int b = 0; b < S; b++) printf("%6d", b);
    printf("\n");
    for (int a = 0; a < S; a++) {
        printf("%3d :", a);
        for (int b = 0; b < S; b++) {
            if (b <= a) { printf("%6s", b == a ? "." : ""); continue; }
            CHK(hipStreamSynchronize(st[a])); CHK(hipStreamSynchronize(st[b]));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[a], d);
                CHK(hipEventRecord(ea, st[a])); CHK(hipStreamWaitEvent(st[b], ea, 0));
                hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[b], d);
                CHK(hipEventRecord(eb, st[b])); CHK(hipStreamWaitEvent(st[a], eb, 0));
            }
            CHK(hipStreamSynchronize(st[a])); CHK(hipStreamSynchronize(st[b]));
            printf("%6.1f", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * N));
        }
        printf("\n");
    }
    int h = 0; CHK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
    printf("(%d kernels ran)\n", h);
    for (auto &s : st) CHK(hipStreamDestroy(s));
    for (auto &s : idlers) CHK(hipStreamDestroy(s));
    CHK(hipEventDestroy(ea)); CHK(hipEventDestroy(eb)); CHK(hipFree(d));
    return 0;
}
