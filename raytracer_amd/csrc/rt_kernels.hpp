// rt_kernels.hpp -- launch interface between the host runtime (rt_api.cpp) and
// the gfx950 kernels (rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt355.h"

namespace rt {

struct SceneDev;

// A bounce queue is kShards independent sub-queues of `cap` entries each; the
// tail counter of sub-queue s for bounce b is counts[(b * kShards + s) * kCountStride],
// the word after it the cursor from which the waves reading that sub-queue claim blocks
// (one counter per 64-byte line).
constexpr uint32_t kShards = 32;
constexpr uint32_t kCountStride = 16;
constexpr size_t kCursorBytes = kShards * kCountStride * sizeof(uint32_t);

// Structure of arrays so that a wave's 64 consecutive entries are 1 KiB
// contiguous in each array.  Entry i of sub-queue s is element s * cap + i.
struct Queue {
    float4 *a; // origin.xyz, time
    float4 *b; // direction.xyz, local path index (bits); 0xFFFFFFFF marks a dead entry
    float4 *c; // throughput.rgb, rng state low word (bits)
    uint2 *d;  // rng state high word, global path id
    uint2 *e;  // scenes with media that enclose them only: what the ray's producer found when it
               // tested those media -- t_max after them (bits), and which of them the ray
               // scatters in (0 = none, k = the k-th)
    uint32_t cap;
};

// Shadow rays of one bounce (RT_RENDER_DIRECT_LIGHT), sub-queues as in Queue: the ray
// Interaction::spawn_ray_to made (t_max = 1 - 0.0001 for all of them), the path's slot and
// what the path gains if the ray reaches the light.
struct ShadowQueue {
    float4 *a; // origin.xyz, time
    float4 *b; // direction.xyz (not normalised), local path index (bits)
    float4 *c; // contribution.rgb, -
    uint32_t cap;
};

// Closest hits of one bounce, entry for entry beside the Queue they were traced from: what the
// trace kernel of the two-kernel bounce (launch_trace) leaves for the shading kernel (launch_shade).
struct HitQueue {
    float4 *a; // leaf slot of the hit (bits; | 0x80000000 if it needs the general shader) or RT_MISS, t, two primitive-specific values
    float *b;  // the third primitive-specific value
};

// Halton::new (colour/src/samplers/halton.rs:25-52) for one frame size, plus the digit
// permutations of the primes 5, 7 and 11 (data/src/lib.rs:16-41) that dimensions 2-4 use.
struct HaltonDev {
    uint32_t scale[2];    // 2^j, 3^k >= min(resolution, 128)
    uint32_t exponent[2]; // j, k
    uint32_t mult_inv[2];
    uint32_t stride;      // scale[0] * scale[1]
    uint8_t perm[28];     // by prime: 2, 3, 5, 7, 11 at offsets 0, 2, 5, 10, 17
};

// Sobol::new (colour/src/samplers/sobol.rs:14-25) for one frame size: the generator matrices
// of dimensions 0-4 (MATRICES_32, data/src/sobol.rs:68) and the two pixel-interval matrices
// of the frame's resolution 2^m (MATRICES[m-1], MATRICES_INV[m-1], data/src/sobol.rs:59978,
// 60151).  Lives in device memory.
struct SobolDev {
    uint32_t m32[5][52];
    uint64_t vdc[52];
    uint64_t inv[52];
};

struct FrameDev {
    uint32_t width, height, spp;
    uint32_t tiles_x;
    uint32_t shard_index, shard_count;
    uint64_t seed;
    uint32_t sampler;     // rt_sampler_kind
    uint32_t sampler_arg; // rt_render_opts::sampler_arg
    HaltonDev halton;
    const SobolDev *sobol; // RT_SAMPLER_SOBOL
    uint32_t sobol_log2;   // m
    // RT_SAMPLER_ZERO_TWO / RT_SAMPLER_STRATIFIED: the camera samples of every pixel, made by
    // launch_pixel_samples before the first ray: five arrays (film x, film y, time, lens u,
    // lens v) of sample_stride floats, sample s of pixel slot p at p * spp + s
    uint32_t sample_stride;
    const float *pixel_samples;
};

// MATRICES / MATRICES_INV of data/src/sobol.rs have 26 rows: resolutions up to 2^26.
constexpr uint32_t kSobolMaxResolution = 1u << 26;
// Fills `h` for a width x height frame (host).
void halton_setup(uint32_t width, uint32_t height, HaltonDev &h);
// Fills `t` for a width x height frame (host); returns m = log2 of the resolution.
uint32_t sobol_setup(uint32_t width, uint32_t height, SobolDev &t);
// PixelSamplerImpl::start_pixel for n_slots pixels (zero_two.rs:23-60, stratified.rs:34-77).
// by_rank: slot = the shard's pixel rank (32x32 tiles); else slot = pixel - first_pixel.
hipError_t launch_pixel_samples(const FrameDev &F, uint32_t n_slots, bool by_rank, uint64_t first_pixel, float *out,
                                hipStream_t stream);

// How the trace kernels are launched for one scene on one device.
struct LaunchShape {
    uint32_t slots;  // traversal stack entries a lane can need
    uint32_t grid;   // persistent workgroups
    bool lds_nodes;  // node array staged in LDS (1024-thread workgroups), else 256-thread workgroups
    // Trees in global memory, SceneDev::quads present: which launches without counters walk the
    // four-box records.  They halve the dependent fetches and cost more arithmetic per level, which
    // pays where rays are incoherent enough to miss the caches (measured on an 11 M primitive scene:
    // camera rays and their first scatter -7 ... -11 %, every later bounce +17 ... +24 %, shuffled rays +13 %).
    //   0 none; 1 the per-bounce trace and shadow launches from bounce kQuadBounce on;
    //   2 those at every bounce and rt_intersect / rt_occluded too
    uint32_t quads;
};

// Resident 256-thread workgroups per CU (= waves per SIMD) the kernels that read the tree
// from global memory are compiled for: more of them hide more memory latency, each gets
// fewer registers and a smaller share of the CU's LDS for its stacks.
#ifndef RT_GLOBAL_WAVES
#define RT_GLOBAL_WAVES 6
#endif
constexpr uint32_t kGlobalWaves = RT_GLOBAL_WAVES;
// Media enclosing the scene that are kept out of the tree, at most.
constexpr uint32_t kMaxEnclosing = 4;
// Stack entries a lane can keep in a private array beyond its LDS levels.
constexpr uint32_t kSpillLevels = 56;
// 64-bit words of a scene's counter block: traversal counters 0-8, tail probe 28-31, profile builds 32-63
constexpr uint32_t kCounterWords = 64;
constexpr uint32_t kQuadBounce = 2;
// Dynamic LDS of a trace workgroup (interior records for the LDS-resident shape + stacks).
size_t lds_bytes(bool lds_nodes, uint32_t n_pairs, uint32_t stack_lds);

// cursor: kCursorBytes of device memory, zero when the kernel starts (kShards counters from
// which the waves claim their ray blocks), or null for a static deal of the blocks
hipError_t launch_intersect(const SceneDev &S, const LaunchShape &L, const rt_ray *rays, uint32_t n,
                            rt_hit *hits, unsigned long long *counters, uint32_t *cursor, hipStream_t stream);
// The cursor block that lives in a file-scope device array instead of allocated memory (one per
// device, shared by every caller on that device).  static_cursor_setup asks the runtime for its
// address on `device`, which must be current; static_cursor returns it, or null before that.
hipError_t static_cursor_setup(int device);
uint32_t *static_cursor(int device);
hipError_t launch_occluded(const SceneDev &S, const LaunchShape &L, const rt_ray *rays, uint32_t n, uint8_t *out,
                           uint32_t *cursor, hipStream_t stream);
// pixel samplers: F.pixel_samples holds the pixels first / spp ... (slot = pixel - first / spp)
hipError_t launch_generate_rays(const FrameDev &F, const rt_camera &cam, uint64_t first, uint32_t n,
                                rt_ray *rays, hipStream_t stream);
// S: the scene the rays will be traced in (camera rays are tested against the media that
// enclose it as they are made)
hipError_t launch_generate(const SceneDev &S, const FrameDev &F, const rt_camera &cam, const Queue &q, uint32_t *counts,
                           uint32_t n_local, uint32_t s0, uint32_t s_count, hipStream_t stream);
// Per-wave memory of the bounce kernel: two lists of parked hit records (kParkSlots x 24
// bytes each) and the lanes' traversal state while a shading batch runs (kSaveWords x 64
// words).
constexpr uint32_t kParkSlots = 128;
constexpr uint32_t kSaveWords = 20;
constexpr size_t kWaveScratchBytes = 2 * kParkSlots * 24 + kSaveWords * 64 * 4;
// Workgroups of the global-memory trace kernels that fit on one CU together with this much
// dynamic LDS each (the smallest answer over the kernels, from the runtime's occupancy query).
hipError_t resident_workgroups(size_t lds_bytes, uint32_t *per_cu);
// Per-wave memory of the kernel that follows paths to their end in one launch (see LoopFeed).
constexpr size_t kWaveLoopBytes = 2 * kParkSlots * 5 * 16 + 64 * 2 * 16 + 512 * 4 * 16 + kSaveWords * 64 * 4;
// waves of the bounce grid (scratch must hold max(kWaveScratchBytes, kWaveLoopBytes) for each)
uint32_t bounce_waves(const LaunchShape &L);
// shadow != null: direct lighting -- hits on Lambertian surfaces also append a shadow ray to
// *shadow (tails at shadow_counts, laid out like counts), radiance is accumulated instead of
// stored, and bit 31 of a queue entry's path index says that the vertex before sampled the lights.
hipError_t launch_bounce(const SceneDev &S, const LaunchShape &L, const Queue &qin, const Queue &qout,
                         uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed, float4 *radiance,
                         uint8_t *scratch, unsigned long long *counters, hipStream_t stream,
                         const ShadowQueue *shadow = nullptr, uint32_t *shadow_counts = nullptr);
// One bounce as two kernels, for trees that stay in global memory: launch_trace finds the closest hit
// of every queued segment and writes it to `hits` (the traversal keeps all of its registers and
// waves), launch_shade reads queue and hits front to back, shades, and compacts the survivors into
// qout (and the shadow rays into *shadow) exactly as launch_bounce does.  Same images, counters
// and queue sizes as launch_bounce.
hipError_t launch_trace(const SceneDev &S, const LaunchShape &L, const Queue &qin, const HitQueue &hits,
                        uint32_t *counts, uint32_t bounce, unsigned long long *counters, hipStream_t stream);
hipError_t launch_shade(const SceneDev &S, const LaunchShape &L, const Queue &qin, const Queue &qout,
                        const HitQueue &hits, uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed,
                        float4 *radiance, hipStream_t stream, const ShadowQueue *shadow = nullptr,
                        uint32_t *shadow_counts = nullptr);
// The shadow rays bounce `bounce` made, through the any-hit traversal (BVH::does_intersect,
// bvh.rs:801-863); an unoccluded ray adds its contribution to its path's radiance.
hipError_t launch_shadow(const SceneDev &S, const LaunchShape &L, const ShadowQueue &q, uint32_t *shadow_counts,
                         uint32_t bounce, float4 *radiance, hipStream_t stream);
// Bounce `bounce` and all later ones of the paths queued in qin (sizes at
// counts[(bounce * kShards + s) * kCountStride]); writes the number of segments traced at each
// later bounce where launch_bounce would have left its queue sizes.
// A wave that takes more than turn_limit feed turns gives up, drops the segments it holds and
// sets counts[kLoopErrorWord] (a word of the counter block no kernel uses otherwise; cleared
// with the block): the frame is then incomplete and the caller must fail it.
constexpr uint32_t kLoopErrorWord = 2;
constexpr uint32_t kLoopTurnLimit = 1u << 24;
hipError_t launch_bounce_loop(const SceneDev &S, const LaunchShape &L, const Queue &qin, uint32_t *counts,
                              uint32_t bounce, uint32_t last, uint64_t seed, float4 *radiance, uint8_t *scratch,
                              unsigned long long *counters, hipStream_t stream, bool steal = true,
                              uint32_t turn_limit = kLoopTurnLimit);
// Film::new's table of filter values (film.rs:99-113), passed to the film kernel by value.
struct FilterDev {
    float w[256];
    float rx, ry;
};
// FilmTile::add_sample (film.rs:281-328) for one batch of samples, gathered per pixel.
hipError_t launch_film(const FrameDev &F, const FilterDev &T, const float4 *radiance, uint32_t s0,
                       uint32_t s_count, float4 *rgba, hipStream_t stream);
// Film::write_image's normalisation (film.rs:236-247), in place.
hipError_t launch_film_resolve(float4 *rgba, size_t n_pixels, hipStream_t stream);
hipError_t launch_resolve(const FrameDev &F, const float4 *radiance, uint32_t n_ranks, uint32_t s_count,
                          int finish, float4 *rgba, hipStream_t stream);

} // namespace rt
