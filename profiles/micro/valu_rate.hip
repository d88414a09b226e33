// profiles/micro/valu_rate.hip — issue cost of the vector instructions the V pass is made of, on one MI355X.
// Every wave runs ITER iterations of 16 independent instructions of one kind (inline asm, so the compiler neither fuses
// nor packs them); grids of 1, 2 and 4 waves per SIMD.  Prints nanoseconds and core cycles (s_memtime at launch clock is not
// available: cycles = ns * the clock the runtime reports) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int KIND> __global__ __launch_bounds__(256) void k_rate(float *out, float seed) {
    float a[16]; float2v p[16]; double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x; p[i] = float2v{a[i], a[i] + 0.5f}; d[i] = a[i]; }
    const float m = 0.999f, c = 0.001f; const float2v m2 = {m, m}, c2 = {c, c}; const double md = 0.999, cd = 0.001;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
            if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(md));
            if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            if (KIND == 8) { if (i & 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); }
            if (KIND == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND> void run(const char *name, int cus, double ghz) {
    float *out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD: blocks of 4 waves, wps blocks per CU
        k_rate<KIND><<<cus * wps, 256>>>(out, 1.0f);
        CHECK(hipEventRecord(e0));
        k_rate<KIND><<<cus * wps, 256>>>(out, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms * 1e6 / ((double)ITER * 16 * wps); // ns per wave-instruction per SIMD
        printf("%-14s waves/SIMD %d  %.3f ns = %.2f cycles per wave-instruction (%.3f ms)\n", name, wps, per, per * ghz, ms);
    }
    CHECK(hipFree(out));
}

// the same chain on a grid of `blocks` one-wave blocks (a mostly idle chip): does a lone wave issue at the same rate?
template <int KIND> void run_small(const char *name, int blocks, double ghz) {
    float *out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; i++) k_rate<KIND><<<blocks, 64>>>(out, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / 20 * 1e6 / ((double)ITER * 16);
        printf("%-14s %4d one-wave blocks: %.3f ns = %.2f cycles per wave-instruction (%.3f ms per launch)\n", name, blocks, per, per * ghz, ms / 20);
    }
    CHECK(hipFree(out));
}

int main() {
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const double ghz = pr.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", pr.name, pr.multiProcessorCount, ghz);
    run<0>("v_fma_f32", pr.multiProcessorCount, ghz);
    run<1>("v_pk_fma_f32", pr.multiProcessorCount, ghz);
    run<2>("v_fma_f64", pr.multiProcessorCount, ghz);
    run<3>("v_add_f32", pr.multiProcessorCount, ghz);
    run<4>("v_mul_f64", pr.multiProcessorCount, ghz);
    run<5>("v_add_f64", pr.multiProcessorCount, ghz);
    run<6>("v_pk_add_f32", pr.multiProcessorCount, ghz);
    run<7>("v_pk_mul_f32", pr.multiProcessorCount, ghz);
    run<8>("f32/f64 mixed", pr.multiProcessorCount, ghz);
    run<9>("v_cvt_f64_f32", pr.multiProcessorCount, ghz);
    run_small<0>("v_fma_f32", 1, ghz);
    run_small<0>("v_fma_f32", 12, ghz);
    run_small<2>("v_fma_f64", 1, ghz);
    return 0;
}
