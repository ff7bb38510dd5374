// valu_peak.hip -- what vector-instruction issue rate the chip sustains (wave64, gfx950).
// Each wave runs a long unrolled stream of independent instructions of one kind on 16
// accumulators; waves per SIMD = 1, 2, 4, 8.  Prints wave-instructions per second for the
// whole chip and cycles per instruction per SIMD at the measured clock
// (s_memtime / s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_peak tools/valu_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kAcc = 16, kIters = 4096;

template <int KIND>
__global__ __launch_bounds__(64) void k_stream(float *out, unsigned long long *clocks, float seed) {
    float a[kAcc];
    for (int i = 0; i < kAcc; i++) a[i] = seed + (float)(threadIdx.x + i);
    const float m = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk[kAcc / 2];
    for (int i = 0; i < kAcc / 2; i++) pk[i] = f2{seed + (float)(threadIdx.x + i), seed + 0.5f};
    const f2 pm = {m, m}, pc = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < kAcc; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            // packed fp32: two lanes' worth of results per lane and instruction (register pairs)
            if (KIND == 8 && (i & 1) == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[i / 2]) : "v"(pm));
            if (KIND == 9 && (i & 1) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[i / 2]) : "v"(pc));
            if (KIND == 10 && (i & 1) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[i / 2]) : "v"(pm), "v"(pc));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int i = 0; i < kAcc; i++) s += a[i];
    for (int i = 0; i < kAcc / 2; i++) s += pk[i].x + pk[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = r1 - r0;
    }
}

template <int KIND>
static void run(const char *name, double insts_per_step, int cus, float *out, unsigned long long *clocks) {
    for (int per_simd : {1, 2, 4, 8}) {
        const int waves = cus * 4 * per_simd;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_stream<KIND>, dim3(waves), dim3(64), 0, 0, out, clocks, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_stream<KIND>, dim3(waves), dim3(64), 0, 0, out, clocks, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CHECK(hipMemcpy(h, clocks, sizeof h, hipMemcpyDeviceToHost));
        const double insts = (double)waves * kIters * kAcc * insts_per_step * reps;
        const double rate = insts / (ms * 1e-3);
        const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ginst_per_s\": %.1f, \"clock_ghz\": %.3f, "
               "\"cycles_per_inst_per_simd\": %.3f}\n",
               name, per_simd, rate / 1e9, ghz, ghz * 1e9 * cus * 4 / rate);
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    unsigned long long *clocks;
    CHECK(hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
    CHECK(hipMalloc(&clocks, 16));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    run<0>("v_fma_f32", 1, cus, out, clocks);
    run<1>("v_mul_f32", 1, cus, out, clocks);
    run<2>("v_add_f32", 1, cus, out, clocks);
    run<3>("v_max_f32", 1, cus, out, clocks);
    run<4>("v_cmp+v_cndmask", 2, cus, out, clocks);
    run<5>("v_mul_lo_u32", 1, cus, out, clocks);
    run<6>("v_rcp_f32", 1, cus, out, clocks);
    run<7>("v_and_b32", 1, cus, out, clocks);
    // packed instructions: half as many per step (each does the work of two plain ones)
    run<8>("v_pk_mul_f32", 0.5, cus, out, clocks);
    run<9>("v_pk_add_f32", 0.5, cus, out, clocks);
    run<10>("v_pk_fma_f32", 0.5, cus, out, clocks);
    return 0;
}
