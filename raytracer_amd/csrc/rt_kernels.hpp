// rt_kernels.hpp -- launch interface between the host runtime (rt_api.cpp) and
// the gfx950 kernels (rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt355.h"

namespace rt {

struct SceneDev;

// One bounce queue, structure of arrays so that a wave's 64 consecutive entries
// are 1 KiB contiguous in each array.
struct Queue {
    float4 *a; // origin.xyz, time
    float4 *b; // direction.xyz, local path index (bits)
    float4 *c; // throughput.rgb, rng state low word (bits)
    uint2 *d;  // rng state high word, global path id
};

struct FrameDev {
    uint32_t width, height, spp;
    uint32_t tiles_x;
    uint32_t shard_index, shard_count;
    uint64_t seed;
};

hipError_t launch_intersect(const SceneDev &S, uint32_t slots, uint32_t grid, const rt_ray *rays, uint32_t n,
                            rt_hit *hits, uint32_t *cursor, unsigned long long *counters,
                            hipStream_t stream);
hipError_t launch_occluded(const SceneDev &S, uint32_t slots, uint32_t grid, const rt_ray *rays, uint32_t n,
                           uint8_t *out, uint32_t *cursor, hipStream_t stream);
hipError_t launch_generate_rays(const rt_camera &cam, uint32_t width, uint32_t spp, uint64_t seed,
                                uint64_t first, uint32_t n, rt_ray *rays, hipStream_t stream);
hipError_t launch_generate(const FrameDev &F, const rt_camera &cam, const Queue &q, uint32_t *tail,
                           uint32_t n_local, uint32_t s0, uint32_t s_count, hipStream_t stream);
hipError_t launch_bounce(const SceneDev &S, uint32_t slots, uint32_t grid, const Queue &qin, const Queue &qout,
                         uint32_t *counts, uint32_t *cursors, uint32_t bounce, uint32_t last, uint64_t seed,
                         float4 *radiance, unsigned long long *counters, hipStream_t stream);
hipError_t launch_resolve(const FrameDev &F, const float4 *radiance, uint32_t n_ranks, uint32_t s_count,
                          int finish, float4 *rgba, hipStream_t stream);

} // namespace rt
