// gather_peak.hip -- what rate of random 64-byte record reads the memory system sustains
// (gfx950), the access pattern of a tree traversal whose records do not fit the caches: every
// lane reads whole 64-byte records (four 16-byte loads, as the traversal's interior step does)
// at pseudo-random places of a table.  Three shapes per table size:
//   independent  kInFlight records per lane in flight at once, addresses from a counter hash:
//                the ceiling for this access pattern
//   chain        one record at a time per lane, the next address taken from the record just
//                read (a dependent chain, like a ray walking down a tree) at 6 and 8 waves per SIMD
// Prints JSON lines: records per second and bytes per second for the whole chip.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/gather_peak tools/gather_peak.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

// every record's first word is a random record index (so chains wander over the whole table)
__global__ void k_fill(uint4 *table, uint32_t n_records) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_records) return;
    for (int q = 0; q < 4; q++)
        table[4 * (size_t)i + q] = make_uint4(mix(i * 4u + q) % n_records, mix(i + 77u), i, q);
}

constexpr int kInFlight = 4;

__global__ __launch_bounds__(256) void k_independent(const uint4 *table, uint32_t n_records, uint32_t iters, uint32_t *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 r[kInFlight][4];
#pragma unroll
        for (int k = 0; k < kInFlight; k++) {
            const uint32_t idx = mix(tid * 2654435761u + it * kInFlight + k) % n_records;
            const uint4 *p = table + 4 * (size_t)idx;
#pragma unroll
            for (int q = 0; q < 4; q++) r[k][q] = p[q];
        }
#pragma unroll
        for (int k = 0; k < kInFlight; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc += r[k][q].x ^ r[k][q].y ^ r[k][q].z ^ r[k][q].w;
    }
    out[tid] = acc;
}

template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_chain(const uint4 *table, uint32_t n_records, uint32_t iters, uint32_t *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = mix(tid) % n_records, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint4 *p = table + 4 * (size_t)idx;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += b.y ^ c.z ^ d.w;
        idx = (a.x + (acc & 1u)) % n_records; // depends on all four loads
    }
    out[tid] = acc;
}

// the same chain over 128-byte records (two adjacent 64-byte lines per step, eight 16-byte loads):
// what a step costs when a record holds four child boxes instead of two
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_chain128(const uint4 *table, uint32_t n_records, uint32_t iters, uint32_t *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = mix(tid) % n_records, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint4 *p = table + 8 * (size_t)idx;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
        acc += b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ h.w;
        idx = (a.x + (acc & 1u)) % n_records;
    }
    out[tid] = acc;
}

template <class K>
static double time_ms(K launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, cus);
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(uint32_t)));
    for (uint32_t mb : {64u, 700u, 1400u, 4096u}) {
        const uint32_t n_records = (uint32_t)((size_t)mb * 1000000 / 64);
        uint4 *table;
        CHECK(hipMalloc(&table, (size_t)n_records * 64));
        hipLaunchKernelGGL(k_fill, dim3((n_records + 255) / 256), dim3(256), 0, 0, table, n_records);
        CHECK(hipDeviceSynchronize());
        {
            const uint32_t iters = 64, blocks = cus * 8;
            const double ms = time_ms([&] { hipLaunchKernelGGL(k_independent, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out); });
            const double recs = (double)blocks * 256 * iters * kInFlight;
            printf("{\"table_MB\": %u, \"shape\": \"independent, %d records in flight per lane, 8 waves/SIMD\", \"Grecords_per_s\": %.2f, \"GB_per_s\": %.1f}\n",
                   mb, kInFlight, recs / ms / 1e6, recs * 64 / ms / 1e6);
        }
        {
            const uint32_t iters = 256, blocks = cus * 6;
            const double ms = time_ms([&] { hipLaunchKernelGGL(k_chain<6>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out); });
            const double recs = (double)blocks * 256 * iters;
            printf("{\"table_MB\": %u, \"shape\": \"dependent chain, 6 waves/SIMD\", \"Grecords_per_s\": %.2f, \"GB_per_s\": %.1f}\n", mb,
                   recs / ms / 1e6, recs * 64 / ms / 1e6);
        }
        {
            const uint32_t iters = 256, blocks = cus * 8;
            const double ms = time_ms([&] { hipLaunchKernelGGL(k_chain<8>, dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out); });
            const double recs = (double)blocks * 256 * iters;
            printf("{\"table_MB\": %u, \"shape\": \"dependent chain, 8 waves/SIMD\", \"Grecords_per_s\": %.2f, \"GB_per_s\": %.1f}\n", mb,
                   recs / ms / 1e6, recs * 64 / ms / 1e6);
        }
        for (int waves : {4, 6}) {
            const uint32_t iters = 256, blocks = cus * waves, n128 = n_records / 2;
            // first words were drawn below n_records: fold them into the half as many 128-byte records
            const double ms = time_ms([&] {
                if (waves == 4) hipLaunchKernelGGL(k_chain128<4>, dim3(blocks), dim3(256), 0, 0, table, n128, iters, out);
                else hipLaunchKernelGGL(k_chain128<6>, dim3(blocks), dim3(256), 0, 0, table, n128, iters, out);
            });
            const double recs = (double)blocks * 256 * iters;
            printf("{\"table_MB\": %u, \"shape\": \"dependent chain of 128-byte records, %d waves/SIMD\", \"Grecords_per_s\": %.2f, \"GB_per_s\": %.1f}\n", mb,
                   waves, recs / ms / 1e6, recs * 128 / ms / 1e6);
        }
        CHECK(hipFree(table));
    }
    return 0;
}
