// profiles/micro/pmc_queue_repro.hip — does rocprofv3's counter collection survive a host that runs far ahead of the device?
// N tiny launches on one stream with no synchronisation in between (what profiles/r3_slots.py --converge 100 does with the
// library: ~240 k launches queued behind one another).  No library code involved.
//   hipcc --offload-arch=gfx950 -O2 -o pmc_queue_repro pmc_queue_repro.hip
//   ./pmc_queue_repro 300000                                                      (plain: prints N)
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/x -o p -f csv -- ./pmc_queue_repro 300000 [sync_every]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_tick(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 300000, sync_every = argc > 2 ? atoi(argv[2]) : 0;
    int *d = nullptr, h = 0;
    if (hipMalloc(&d, 4) != hipSuccess || hipMemset(d, 0, 4) != hipSuccess) return 2;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
    for (int i = 0; i < n; i++) {
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, s, d);
        if (sync_every && (i + 1) % sync_every == 0) (void)hipStreamSynchronize(s);
        if ((i + 1) % 20000 == 0) { fprintf(stderr, "launched %d\n", i + 1); fflush(stderr); }
    }
    if (hipStreamSynchronize(s) != hipSuccess) return 3;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%d of %d launches ran\n", h, n);
    return h == n ? 0 : 1;
}
