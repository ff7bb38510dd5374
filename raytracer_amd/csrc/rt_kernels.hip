// rt_kernels.hip -- gfx950 kernels: closest-hit / any-hit BVH traversal, camera
// ray generation, the fused trace + shade + compact bounce kernel and the
// framebuffer resolve.
//
// Execution shape: persistent workgroups of 256 threads (4 waves of 64); a
// launch is a fixed grid of (CUs x resident blocks) however many rays are
// queued, and wave w of W works on the 64-ray chunks w, w + W, w + 2W, ...
// A lane's traversal stack lives in LDS as stack[level * 256 + thread]:
// consecutive lanes hit consecutive banks, so pushes and pops are conflict
// free.
//
// Bounce queues are split into kShards independent sub-queues.  A chunk read
// from shard s appends its survivors to shard s of the next queue: one
// ballot + prefix count + one atomic per wave, and the atomics of a launch
// spread over kShards counters that sit on separate cache lines instead of
// serialising on one word.  Because blocks are dealt round-robin to the 8
// XCDs and the wave count is a multiple of kShards, every XCD keeps to its
// own 4 shards from one bounce to the next.
#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rt {

constexpr int kBlock = 256;
// minimum resident waves per SIMD the register allocator must leave room for
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 4
#endif
// lanes holding a leaf that trigger a leaf round (64 = wait for every lane)
#ifndef RT_LEAF_LANES
#define RT_LEAF_LANES 24
#endif
constexpr uint32_t kLeafLanes = RT_LEAF_LANES;
#ifndef RT_BOUNCE_WAVES
#define RT_BOUNCE_WAVES 4
#endif
constexpr uint32_t kDead = 0xFFFFFFFFu;

RT_DEV uint32_t lane_rank(uint64_t mask) { // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
RT_DEV uint32_t wave_index() { return blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); }
RT_DEV uint32_t wave_total() { return gridDim.x * (blockDim.x >> 6); }

// Two launch shapes.  Trees whose interior records fit beside the stacks are
// copied into LDS by every workgroup (1024 threads, one per CU): incoherent rays
// then fetch nodes with ds_read_b128 from four separate arrays (one per 16-byte
// quarter of the record, so that the lanes of one read spread over all 16-byte
// slots of a bank row) instead of queueing 64 distinct cache lines per load in
// the vector L1.  Larger trees stay in global memory (256-thread workgroups, one
// 64-byte line per record).
template <bool LDSN>
struct Shape;
template <>
struct Shape<false> {
    static constexpr int kThreads = kBlock;
    static constexpr int kWaves = RT_TRACE_WAVES;
    using RefT = uint32_t;
};
template <>
struct Shape<true> {
    static constexpr int kThreads = 1024;
    static constexpr int kWaves = 4;
    using RefT = uint16_t; // record index < 2^15, or 0x8000 | first primitive of a leaf
};

// Stack entries beyond the LDS part live in a per-lane private array.
constexpr int kSpill = (int)kSpillLevels;

extern __shared__ float4 lds_raw[];

template <bool LDSN>
struct TreeView {
    const float4 *q;     // LDSN: LDS copy, quarter k of record i at q[k * n_pairs + i]
    uint32_t n_pairs;
    typename Shape<LDSN>::RefT *ref; // stack part in LDS: ref[level * kThreads], near[level * kThreads]
    float *near;
    uint32_t levels;     // stack levels kept in LDS
};

// Carves the dynamic LDS; with LDSN also stages the records (call from all threads).
template <bool LDSN>
RT_DEV TreeView<LDSN> tree_view(const SceneDev &S) {
    constexpr int kThreads = Shape<LDSN>::kThreads;
    TreeView<LDSN> v;
    v.n_pairs = S.n_pairs;
    v.levels = S.stack_lds;
    float4 *base = lds_raw;
    if constexpr (LDSN) {
        const uint32_t n = S.n_pairs;
        for (uint32_t i = threadIdx.x; i < 4 * n; i += kThreads) {
            float4 x = S.nodes[i]; // record i / 4, quarter i % 4
            base[(i & 3u) * n + (i >> 2)] = x;
        }
        __syncthreads();
        v.q = base;
        base += 4 * (size_t)n;
    } else {
        v.q = nullptr;
    }
    v.near = reinterpret_cast<float *>(base) + threadIdx.x;
    v.ref = reinterpret_cast<typename Shape<LDSN>::RefT *>(reinterpret_cast<float *>(base) +
                                                            (size_t)S.stack_lds * kThreads) +
            threadIdx.x;
    return v;
}

// Build with -DRT_PROFILE to have the counting kernels also time their phases
// (shader clock) and tally lane occupancy; rt_render_dev prints the totals.
#ifdef RT_PROFILE
#define PROF(...) __VA_ARGS__
#else
#define PROF(...)
#endif

struct Counters {
    uint32_t nodes;
    uint32_t prims[8];
    // 0 node-loop iterations, 1 lanes stepping in them, 2 leaf rounds, 3 primitive iterations
    // of leaf rounds (longest lane), 4 lanes x primitives tested, 5 cycles in node loops,
    // 6 cycles in leaf rounds, 7 cycles in traverse, 8 cycles shading, 9 cycles per chunk,
    // 10 chunks, 11 rays
    PROF(unsigned long long p[12];)
};

// <BVH as Primitive>::intersect, shapes/src/bvh.rs:736-799, with the ray
// shortened on every accepted hit (ties go to the primitive tested later).
// Returns the leaf slot of the winning primitive or RT_MISS.
//
// Same visiting order and the same box and primitive tests per ray as the
// reference loop, re-associated so that one step handles an interior node's two
// children: the reference pops a node, tests its box, and for an interior node
// continues with the near child while the far child waits on the stack
// (bvh.rs:778-795); both children are tested when they are popped.  Here the
// two boxes sit in the parent's record and are tested together.  The near
// child's test is the one the reference makes next.  The far child's test
// depends on the ray only through `t_near < ray.t_max`, so its t_near goes on
// the stack with it and the comparison is repeated against the ray as it is
// when the entry is popped: exactly the outcome of the reference's later test.
// A child whose box fails is never stacked; the reference would pop it, test
// it and discard it, so the visit counter still counts both children.
//
// Scheduled for a 64-wide wave: every lane walks interior records until it
// holds a leaf (or runs out of nodes), then the lanes that hold a leaf test its
// primitives together, so the primitive code (a switch over seven kinds) runs
// once per leaf round instead of once per node step.
template <bool ANY, bool COUNT, bool LDSN>
RT_DEV uint32_t traverse(const SceneDev &S, const TreeView<LDSN> &tv, Ray &ray, float &a0, float &a1, float &a2,
                         Counters &cnt) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    constexpr int kStride = Shape<LDSN>::kThreads;
    using RefT = typename Shape<LDSN>::RefT;
    RefT *const s_ref = tv.ref;
    float *const s_near = tv.near;
    const uint32_t levels = tv.levels;
    uint32_t spill_ref[kSpill];
    float spill_near[kSpill];
    v3 inv = mk(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z); // bvh.rs:741-750
    bool negx = inv.x < 0.0f, negy = inv.y < 0.0f, negz = inv.z < 0.0f;
    uint32_t sp = 0, cur = kNone, best = RT_MISS;

    auto push = [&](uint32_t r, float t) {
        if (sp < levels) {
            if constexpr (LDSN) s_ref[sp * kStride] = (RefT)((r & 0x7fffu) | ((r >> 31) << 15));
            else s_ref[sp * kStride] = r;
            s_near[sp * kStride] = t;
        } else {
            spill_ref[sp - levels] = r;
            spill_near[sp - levels] = t;
        }
        sp++;
    };
    // next stacked node whose box the shortened ray still enters (bvh.rs:760-762 on a popped node)
    auto pop = [&]() {
        uint32_t r = kNone;
        while (sp > 0) {
            sp--;
            uint32_t e;
            float t;
            if (sp < levels) {
                e = s_ref[sp * kStride];
                if constexpr (LDSN) e = (e & 0x7fffu) | ((e & 0x8000u) << 16);
                t = s_near[sp * kStride];
            } else {
                e = spill_ref[sp - levels];
                t = spill_near[sp - levels];
            }
            if (t < ray.t_max) {
                r = e;
                break;
            }
        }
        return r;
    };

    if (S.n_nodes) { // the root's own box, bvh.rs:757-762
        if (COUNT) cnt.nodes++;
        float t;
        float4 r0 = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.0f);
        float4 r1 = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.0f);
        if (slab(r0, r1, ray, inv, negx, negy, negz, t)) cur = S.root_ref;
    }
    PROF(const unsigned long long tp0 = __builtin_amdgcn_s_memtime();)
    for (;;) {
        // interior steps: lanes that hold a leaf wait here until kLeafLanes of them do (or
        // nobody can step any more)
        PROF(const unsigned long long tn0 = __builtin_amdgcn_s_memtime();)
        for (;;) {
            const bool stepping = (cur >> 31) == 0;
            const bool holding = !stepping && cur != kNone;
            if (__ballot(stepping) == 0) break;
            if ((uint32_t)__popcll(__ballot(holding)) >= kLeafLanes) break;
            PROF(if (COUNT) { cnt.p[0]++; cnt.p[1] += stepping ? 1 : 0; })
            if (stepping) {
                float4 q0, q1, q2, q3;
                if constexpr (LDSN) {
                    const float4 *q = tv.q + cur;
                    const uint32_t n = tv.n_pairs;
                    q0 = q[0];
                    q1 = q[n];
                    q2 = q[2 * (size_t)n];
                    q3 = q[3 * (size_t)n];
                } else {
                    const float4 *q = S.nodes + 4 * (size_t)cur;
                    q0 = q[0];
                    q1 = q[1];
                    q2 = q[2];
                    q3 = q[3];
                }
                if (COUNT) cnt.nodes += 2;
                float t0, t1;
                const bool h0 = slab(q0, q1, ray, inv, negx, negy, negz, t0);
                const bool h1 = slab(q2, q3, ray, inv, negx, negy, negz, t1);
                const uint32_t r0 = __float_as_uint(q0.w), r1 = __float_as_uint(q1.w);
                const uint32_t axis = __float_as_uint(q2.w);
                const bool neg = axis == 0 ? negx : (axis == 1 ? negy : negz); // second child is nearer
                if (h0 && h1) {
                    push(neg ? r0 : r1, neg ? t0 : t1);
                    cur = neg ? r1 : r0;
                } else if (h0) {
                    cur = r0;
                } else if (h1) {
                    cur = r1;
                } else {
                    cur = pop();
                }
            }
        }
        const bool holding = (cur >> 31) != 0 && cur != kNone;
        PROF(const unsigned long long tl0 = __builtin_amdgcn_s_memtime(); if (COUNT) cnt.p[5] += tl0 - tn0;)
        if (__ballot(holding) == 0) break; // every ray of the wave ran out of nodes
        PROF(uint32_t my_iters = 0;)
        if (holding) { // leaf, bvh.rs:763-776
            bool stop = false;
            for (uint32_t i = cur & ~kLeafRef;; i++) {
                PROF(my_iters++;)
                LeafPrim lp = S.leaf[i];
                float t, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                if (COUNT) {
                    uint32_t k = lp.kind & 7u;
#pragma unroll
                    for (uint32_t q = 0; q < 8; q++) cnt.prims[q] += (k == q) ? 1u : 0u;
                }
                if (prim_test(S, lp, ray, t, b0, b1, b2)) {
                    ray.t_max = t;
                    best = i;
                    a0 = b0; a1 = b1; a2 = b2;
                    if (ANY) { stop = true; break; }
                }
                if (lp.kind & kLastInLeaf) break;
            }
            cur = stop ? kNone : pop();
        }
        PROF(if (COUNT) {
            uint32_t longest = my_iters;
            for (int sh = 32; sh > 0; sh >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, sh, 64));
            cnt.p[2]++;
            cnt.p[3] += longest;
            cnt.p[4] += my_iters;
            cnt.p[6] += __builtin_amdgcn_s_memtime() - tl0;
        })
    }
    PROF(if (COUNT) cnt.p[7] += __builtin_amdgcn_s_memtime() - tp0;)
    return best;
}

RT_DEV Ray load_ray(const rt_ray *rays, size_t i) {
    const float4 *p = reinterpret_cast<const float4 *>(rays + i);
    float4 a = p[0], b = p[1];
    Ray r;
    r.o = mk(a.x, a.y, a.z);
    r.d = mk(a.w, b.x, b.y);
    r.t_max = b.z;
    r.time = b.w;
    return r;
}

RT_DEV void flush_counters(const Counters &c, unsigned long long *out) {
    // wave reduction, then one atomic per wave and counter
    uint32_t v[9] = {c.nodes, c.prims[0], c.prims[1], c.prims[2], c.prims[3], c.prims[4], c.prims[5],
                     c.prims[6], c.prims[7]};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t x = v[k];
        for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out[k], (unsigned long long)x);
    }
    PROF(for (int k = 0; k < 12; k++) {
        unsigned long long x = c.p[k];
        const bool per_lane = k == 1 || k == 4 || k == 11;
        if (per_lane)
            for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out[16 + k], x);
    })
}

template <bool COUNT, bool LDSN>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_intersect(
    SceneDev S, const rt_ray *rays, uint32_t n, rt_hit *hits, unsigned long long *counters) {
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t stride = wave_total() * 64;
    Counters cnt = {};
    for (uint32_t base = wave_index() * 64; base < n; base += stride) {
        uint32_t i = base + lane;
        if (i < n) {
            Ray ray = load_ray(rays, i);
            const Ray ray0 = ray;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            uint32_t slot = traverse<false, COUNT, LDSN>(S, tv, ray, a0, a1, a2, cnt);
            float4 h0, h1;
            if (slot != RT_MISS) {
                LeafPrim lp = S.leaf[slot];
                Surf sf;
                prim_finalize(S, lp, ray0, ray.t_max, a0, a1, a2, sf);
                h0 = make_float4(sf.t, __uint_as_float(lp.prim), sf.u, sf.v);
                h1 = make_float4(sf.n.x, sf.n.y, sf.n.z, __uint_as_float(sf.flags));
            } else {
                h0 = make_float4(kInf, __uint_as_float(RT_MISS), 0.0f, 0.0f);
                h1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            float4 *o = reinterpret_cast<float4 *>(hits + i);
            o[0] = h0;
            o[1] = h1;
        }
    }
    if (COUNT) flush_counters(cnt, counters);
}

// <BVH as Primitive>::does_intersect, shapes/src/bvh.rs:801-863
template <bool LDSN>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_occluded(
    SceneDev S, const rt_ray *rays, uint32_t n, uint8_t *out) {
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t stride = wave_total() * 64;
    Counters cnt = {};
    for (uint32_t base = wave_index() * 64; base < n; base += stride) {
        uint32_t i = base + lane;
        if (i < n) {
            Ray ray = load_ray(rays, i);
            float a0, a1, a2;
            uint32_t slot = traverse<true, false, LDSN>(S, tv, ray, a0, a1, a2, cnt);
            out[i] = slot != RT_MISS ? 1 : 0;
        }
    }
}

// ---- camera, colour/src/camera/perspective.rs:54-78 -----------------------------------------
RT_DEV void concentric_disk(float u, float v, float &dx, float &dy) { // sampling.rs:4-22
    float ox = u * 2.0f - 1.0f, oy = v * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) { dx = 0.0f; dy = 0.0f; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        r = ox;
        theta = 0.785398163f * (oy / ox);
    } else {
        r = oy;
        theta = 1.57079637f - 0.785398163f * (ox / oy);
    }
    dx = p_cos(theta) * r;
    dy = p_sin(theta) * r;
}

RT_DEV Ray camera_ray(const rt_camera &cam, float fx, float fy, float lu, float lv, float tu) {
    const float *m = cam.raster_to_camera;
    float xp = m[0] * fx + m[1] * fy + m[2] * 0.0f + m[3];
    float yp = m[4] * fx + m[5] * fy + m[6] * 0.0f + m[7];
    float zp = m[8] * fx + m[9] * fy + m[10] * 0.0f + m[11];
    float wp = m[12] * fx + m[13] * fy + m[14] * 0.0f + m[15];
    v3 pc = mk(xp, yp, zp);
    if (wp != 1.0f) pc = pc / wp;
    v3 o = mk(0.0f, 0.0f, 0.0f);
    v3 d = normalize(pc);
    if (cam.lens_radius > 0.0f) {
        float lx, ly;
        concentric_disk(lu, lv, lx, ly);
        lx *= cam.lens_radius;
        ly *= cam.lens_radius;
        float ft = cam.focal_distance / d.z;
        v3 focus = d * ft;
        o = mk(lx, ly, 0.0f);
        d = normalize(focus - o);
    }
    float time = (1.0f - tu) * cam.shutter_open + tu * cam.shutter_close;
    const float *c = cam.camera_to_world; // ray * transform, transform.rs:530-552
    v3 wo = xf_point(c, o);
    v3 oerr = xf_abs_sum(c, o) * G3;
    v3 wd = xf_vector(c, d);
    float len_sq = dot(wd, wd);
    float t_max = kInf;
    if (len_sq > 0.0f) {
        float dt = dot(vabs(wd), oerr) / len_sq;
        wo = wo + wd * dt;
        t_max -= dt;
    }
    Ray r;
    r.o = wo;
    r.d = wd;
    r.t_max = t_max;
    r.time = time;
    return r;
}

// Path of pixel (px, py), global id `path` = pixel * spp + sample; PCG32 sequence =
// seed * 2^32 + path; draws: film x, film y, time, lens u, lens v.
RT_DEV Ray path_setup(const rt_camera &cam, uint32_t px, uint32_t py, uint64_t seed, uint64_t path, Rng &rng) {
    rng_seq(rng, (seed << 32) + path);
    float u1 = rng_f(rng), u2 = rng_f(rng), ut = rng_f(rng);
    float l1 = rng_f(rng), l2 = rng_f(rng);
    return camera_ray(cam, (float)px + u1, (float)py + u2, l1, l2, ut);
}

__global__ __launch_bounds__(kBlock) void k_generate_rays(rt_camera cam, uint32_t width, uint32_t spp,
                                                          uint64_t seed, uint64_t first, uint32_t n,
                                                          rt_ray *rays) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    uint64_t path = first + i, pixel = path / spp;
    Ray r = path_setup(cam, (uint32_t)(pixel % width), (uint32_t)(pixel / width), seed, path, rng);
    float4 *o = reinterpret_cast<float4 *>(rays + i);
    o[0] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
    o[1] = make_float4(r.d.y, r.d.z, r.t_max, r.time);
}

// ---- bounce queues ------------------------------------------------------------------------------
RT_DEV void queue_push(const Queue &q, size_t slot, const Ray &r, uint32_t local, v3 beta, const Rng &rng,
                       uint32_t gpath) {
    q.a[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.time);
    q.b[slot] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(local));
    q.c[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float((uint32_t)rng.state));
    q.d[slot] = make_uint2((uint32_t)(rng.state >> 32), gpath);
}

// pixel owned by this shard of the frame with rank r: 32x32 tiles, tile t = shard_index + k * shard_count
RT_DEV bool rank_to_pixel(const FrameDev &F, uint32_t rank, uint32_t &px, uint32_t &py) {
    uint32_t tile = F.shard_index + (rank >> 10) * F.shard_count;
    uint32_t within = rank & 1023u;
    uint32_t tx = tile % F.tiles_x, ty = tile / F.tiles_x;
    px = tx * 32 + (within & 31u);
    py = ty * 32 + (within >> 5);
    return px < F.width && py < F.height;
}

// Primary rays, written in place: path `local` goes to chunk local / 64, chunk c to
// sub-queue c % kShards.  Pixels outside the image (partial edge tiles) leave a dead entry.
__global__ __launch_bounds__(kBlock) void k_generate(FrameDev F, rt_camera cam, Queue q, uint32_t *counts,
                                                     uint32_t n_local, uint32_t s0, uint32_t s_count) {
    const uint32_t local = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n_chunks = (n_local + 63) / 64;
    if (blockIdx.x == 0 && threadIdx.x < kShards) {
        uint32_t s = threadIdx.x;
        counts[s * kCountStride] = s < n_chunks ? 64u * ((n_chunks - s + kShards - 1) / kShards) : 0u;
    }
    const uint32_t chunk = local >> 6;
    if (chunk >= n_chunks) return;
    const size_t slot = (size_t)(chunk % kShards) * q.cap + (size_t)(chunk / kShards) * 64 + (local & 63u);
    bool alive = false;
    Ray ray;
    Rng rng;
    uint32_t gpath = 0;
    if (local < n_local) {
        uint32_t rank = local / s_count, s = s0 + local % s_count, px, py;
        if (rank_to_pixel(F, rank, px, py)) {
            gpath = (py * F.width + px) * F.spp + s;
            ray = path_setup(cam, px, py, F.seed, gpath, rng);
            alive = true;
        }
    }
    if (alive) {
        queue_push(q, slot, ray, local, mk(1.0f, 1.0f, 1.0f), rng, gpath);
    } else {
        q.b[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kDead));
    }
}

// ---- textures -----------------------------------------------------------------------------------
RT_DEV float perlin_noise(const SceneDev &S, v3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    const int32_t *px = S.perlin_perm, *py = S.perlin_perm + 256, *pz = S.perlin_perm + 512;
    float accum = 0.0f;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                int idx = px[(i + di) & 255] ^ py[(j + dj) & 255] ^ pz[(k + dk) & 255];
                const float *c = S.perlin_vec + 3 * idx;
                float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
                float dt = c[0] * wx + c[1] * wy + c[2] * wz;
                float fi = di ? uu : 1.0f - uu;
                float fj = dj ? vv : 1.0f - vv;
                float fk = dk ? ww : 1.0f - ww;
                accum += fi * fj * fk * dt;
            }
    return accum;
}
RT_DEV float perlin_turb(const SceneDev &S, v3 p) {
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; i++) {
        accum += weight * perlin_noise(S, p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return fabsf(accum);
}

RT_DEV v3 texture_value(const SceneDev &S, uint32_t tex, float u, float v, v3 p) {
    for (int level = 0; level < 8; level++) {
        rt_texture t = S.textures[tex];
        switch (t.kind) {
        case RT_TEX_CHECKER: {
            float sines = p_sin(t.scale * p.x) * p_sin(t.scale * p.y) * p_sin(t.scale * p.z);
            tex = sines < 0.0f ? t.a : t.b;
            continue;
        }
        case RT_TEX_CHECKER_UV: {
            int cu = (int)floorf(u * t.scale), cv = (int)floorf(v * t.scale);
            tex = ((cu + cv) & 1) ? t.a : t.b;
            continue;
        }
        case RT_TEX_IMAGE: {
            ImageDev im = S.images[t.a];
            float uc = clampf(u, 0.0f, 1.0f), vc = 1.0f - clampf(v, 0.0f, 1.0f);
            int i = (int)(uc * (float)im.w), j = (int)(vc * (float)im.h);
            if (i >= (int)im.w) i = (int)im.w - 1;
            if (j >= (int)im.h) j = (int)im.h - 1;
            const uint8_t *px = im.rgb8 + 3 * ((size_t)j * im.w + (size_t)i);
            const float sc = 1.0f / 255.0f;
            return mk((float)px[0] * sc, (float)px[1] * sc, (float)px[2] * sc);
        }
        case RT_TEX_NOISE: {
            float g = 0.5f * (1.0f + p_sin(t.scale * p.z + 10.0f * perlin_turb(S, p)));
            return mk(g, g, g);
        }
        default: return mk(t.rgb[0], t.rgb[1], t.rgb[2]);
        }
    }
    return mk(0.0f, 0.0f, 0.0f);
}

RT_DEV v3 random_in_unit_sphere(Rng &rng) {
    for (;;) {
        float x = 2.0f * rng_f(rng) - 1.0f;
        float y = 2.0f * rng_f(rng) - 1.0f;
        float z = 2.0f * rng_f(rng) - 1.0f;
        if (x * x + y * y + z * z < 1.0f) return mk(x, y, z);
    }
}
RT_DEV v3 reflect(v3 v, v3 n) { return v - n * (2.0f * dot(v, n)); }

// spawn_ray origin, shapes/src/interaction.rs:86-99 + geometry/src/lib.rs:129-151
RT_DEV v3 offset_origin(v3 p, v3 err, v3 n, v3 w) {
    float d = dot(vabs(n), err);
    v3 off = n * d;
    if (dot(n, w) < 0.0f) off = -off;
    v3 po = p + off;
    if (off.x > 0.0f) po.x = next_float_up(po.x); else if (off.x < 0.0f) po.x = next_float_down(po.x);
    if (off.y > 0.0f) po.y = next_float_up(po.y); else if (off.y < 0.0f) po.y = next_float_down(po.y);
    if (off.z > 0.0f) po.z = next_float_up(po.z); else if (off.z < 0.0f) po.z = next_float_down(po.z);
    return po;
}

// One bounce of every queued path: closest hit, shading, scatter, and the
// survivors compacted into the next queue.  Radiance is stored once, when a
// path ends on a light or leaves the scene.
template <bool COUNT, bool LDSN>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, LDSN ? 4 : RT_BOUNCE_WAVES) void k_bounce(
    SceneDev S, Queue qin, Queue qout, uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed,
    float4 *radiance, unsigned long long *counters) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t *cin = counts + (size_t)bounce * kShards * kCountStride;
    uint32_t *cout = counts + (size_t)(bounce + 1) * kShards * kCountStride;
    // lane s holds the size of sub-queue s; the longest one bounds the chunk loop
    const uint32_t my_count = lane < kShards ? cin[lane * kCountStride] : 0u;
    uint32_t longest = my_count;
    for (int s = 32; s > 0; s >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, s, 64));
    const uint32_t n_chunks = kShards * ((longest + 63) / 64);
    if (blockIdx.x * (blockDim.x >> 6) >= n_chunks) return; // nothing for this workgroup (uniform)
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    for (uint32_t c = wave_index(); c < n_chunks; c += wave_total()) {
        PROF(const unsigned long long tc0 = __builtin_amdgcn_s_memtime(); unsigned long long ts0 = tc0;)
        const uint32_t shard = c % kShards;
        const uint32_t i = (c / kShards) * 64 + lane;
        const uint32_t n = __builtin_amdgcn_readlane(my_count, shard);
        const size_t in_base = (size_t)shard * qin.cap;
        bool alive = false;
        Ray ray;
        Rng rng;
        v3 beta = mk(0.0f, 0.0f, 0.0f);
        uint32_t local = kDead, gpath = 0;
        if (i < n) {
            float4 qb = qin.b[in_base + i];
            local = __float_as_uint(qb.w);
            if (local != kDead) {
                float4 qa = qin.a[in_base + i], qc = qin.c[in_base + i];
                uint2 qd = qin.d[in_base + i];
                ray.o = mk(qa.x, qa.y, qa.z);
                ray.time = qa.w;
                ray.d = mk(qb.x, qb.y, qb.z);
                ray.t_max = kInf;
                beta = mk(qc.x, qc.y, qc.z);
                gpath = qd.y;
                rng.state = ((uint64_t)qd.x << 32) | __float_as_uint(qc.w);
                rng.inc = (((seed << 32) + gpath) << 1) | 1u;
                alive = true;
            }
        }
        if (alive) {
            const Ray ray0 = ray;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            uint32_t slot = traverse<false, COUNT, LDSN>(S, tv, ray, a0, a1, a2, cnt);
            PROF(ts0 = __builtin_amdgcn_s_memtime();)
            PROF(if (COUNT) cnt.p[11]++;)
            if (slot == RT_MISS) {
                radiance[local] = make_float4(beta.x * S.background[0], beta.y * S.background[1],
                                              beta.z * S.background[2], 0.0f);
                alive = false;
            } else {
                LeafPrim lp = S.leaf[slot];
                Surf h;
                prim_finalize(S, lp, ray0, ray.t_max, a0, a1, a2, h);
                const v3 d = ray0.d;
                const bool front = dot(d, h.n) < 0.0f;
                const v3 nf = front ? h.n : -h.n;
                const rt_material m = S.materials[lp.material];
                // The pieces every material shares run once for the whole wave: the
                // texture lookup (all but dielectrics) and the unit-sphere sample
                // (Lambertian, metallic, isotropic).  Per path the RNG draws stay in
                // the same order as when each material is evaluated on its own.
                v3 tex = mk(1.0f, 1.0f, 1.0f), rs = mk(0.0f, 0.0f, 0.0f);
                if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METALLIC || m.kind == RT_MAT_ISOTROPIC)
                    rs = random_in_unit_sphere(rng);
                if (m.kind != RT_MAT_DIELECTRIC) tex = texture_value(S, m.texture, h.u, h.v, h.p);
                v3 dir = nf;
                if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
                    radiance[local] = make_float4(beta.x * tex.x, beta.y * tex.y, beta.z * tex.z, 0.0f);
                    alive = false;
                } else if (m.kind == RT_MAT_LAMBERTIAN) {
                    dir = nf + rs;
                    if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = nf;
                } else if (m.kind == RT_MAT_METALLIC) {
                    dir = reflect(d, nf) + rs * m.fuzz;
                    if (!(dot(dir, nf) > 0.0f)) alive = false; // absorbed
                } else if (m.kind == RT_MAT_DIELECTRIC) {
                    float ratio = front ? 1.0f / m.ior : m.ior;
                    float cos_t = sel_min(dot(-d, nf), 1.0f);
                    float sin_t = sqrtf(1.0f - cos_t * cos_t);
                    bool reflect_it = ratio * sin_t > 1.0f;
                    if (!reflect_it) {
                        float r0 = (1.0f - ratio) / (1.0f + ratio);
                        r0 = r0 * r0;
                        float c5 = 1.0f - cos_t;
                        float refl = r0 + (1.0f - r0) * (c5 * c5 * c5 * c5 * c5);
                        reflect_it = refl > rng_f(rng);
                    }
                    if (reflect_it) {
                        dir = reflect(d, nf);
                    } else {
                        v3 perp = (d + nf * cos_t) * ratio;
                        float k = 1.0f - dot(perp, perp);
                        v3 par = nf * -sqrtf(fabsf(k));
                        dir = perp + par;
                    }
                } else { // RT_MAT_ISOTROPIC
                    dir = rs;
                }
                if (alive) {
                    dir = normalize(dir);
                    beta = mk(beta.x * tex.x, beta.y * tex.y, beta.z * tex.z);
                    ray.o = offset_origin(h.p, h.perr, h.n, dir);
                    ray.d = dir;
                    ray.t_max = kInf;
                }
            }
            if (bounce == last) alive = false; // depth exhausted: the path contributes nothing
        }
        // stream compaction: ballot, rank by prefix count, one atomic per wave on the shard's tail
        const uint64_t mask = __ballot(alive);
        const uint32_t total = (uint32_t)__popcll(mask);
        if (total) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&cout[shard * kCountStride], total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (alive)
                queue_push(qout, (size_t)shard * qout.cap + base + lane_rank(mask), ray, local, beta, rng, gpath);
        }
        PROF(if (COUNT) {
            const unsigned long long te = __builtin_amdgcn_s_memtime();
            uint32_t shade = (uint32_t)(te - ts0); // lanes without a ray still hold the chunk's start
            for (int sh = 32; sh > 0; sh >>= 1) shade = min(shade, (uint32_t)__shfl_xor(shade, sh, 64));
            cnt.p[8] += shade;
            cnt.p[9] += te - tc0;
            cnt.p[10]++;
        })
    }
    if (COUNT) flush_counters(cnt, counters);
}

// Adds the batch's samples to the running per-pixel sums in sample order; the
// last batch divides by spp.  One thread per owned pixel.
__global__ __launch_bounds__(kBlock) void k_resolve(FrameDev F, const float4 *radiance, uint32_t n_ranks,
                                                    uint32_t s_count, int finish, float4 *rgba) {
    uint32_t rank = blockIdx.x * kBlock + threadIdx.x, px, py;
    if (rank >= n_ranks || !rank_to_pixel(F, rank, px, py)) return;
    size_t pix = (size_t)py * F.width + px;
    float4 acc = rgba[pix];
    const float4 *r = radiance + (size_t)rank * s_count;
    for (uint32_t s = 0; s < s_count; s++) {
        float4 v = r[s];
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
    }
    if (finish) {
        float inv = (float)F.spp;
        acc.x = acc.x / inv;
        acc.y = acc.y / inv;
        acc.z = acc.z / inv;
        acc.w = 1.0f;
    }
    rgba[pix] = acc;
}

// ---- host-side launchers ------------------------------------------------------------------------------
// dynamic LDS: the LDS part of the stacks (child reference + entry distance per level and
// lane), preceded by the interior records for the LDS-resident shape
size_t lds_bytes(bool lds_nodes, uint32_t n_pairs, uint32_t stack_lds) {
    return lds_nodes ? (size_t)n_pairs * 64 + (size_t)stack_lds * 1024 * (sizeof(float) + sizeof(uint16_t))
                     : (size_t)stack_lds * kBlock * (sizeof(float) + sizeof(uint32_t));
}

template <class K>
static hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

#define RT_LAUNCH(kernel, blocks, threads, ...)                                                        \
    do {                                                                                               \
        hipError_t e_ = allow_lds(kernel, lds);                                                        \
        if (e_ != hipSuccess) return e_;                                                               \
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, stream, __VA_ARGS__);             \
    } while (0)

hipError_t launch_intersect(const SceneDev &S, const LaunchShape &L, const rt_ray *rays, uint32_t n,
                            rt_hit *hits, unsigned long long *counters, hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    uint32_t blocks = (n + threads - 1) / threads;
    if (blocks > L.grid) blocks = L.grid;
    if (blocks == 0) return hipSuccess;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) {
        if (counters) RT_LAUNCH((k_intersect<true, true>), blocks, threads, S, rays, n, hits, counters);
        else RT_LAUNCH((k_intersect<false, true>), blocks, threads, S, rays, n, hits, counters);
    } else {
        if (counters) RT_LAUNCH((k_intersect<true, false>), blocks, threads, S, rays, n, hits, counters);
        else RT_LAUNCH((k_intersect<false, false>), blocks, threads, S, rays, n, hits, counters);
    }
    return hipGetLastError();
}

hipError_t launch_occluded(const SceneDev &S, const LaunchShape &L, const rt_ray *rays, uint32_t n, uint8_t *out,
                           hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    uint32_t blocks = (n + threads - 1) / threads;
    if (blocks > L.grid) blocks = L.grid;
    if (blocks == 0) return hipSuccess;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) RT_LAUNCH((k_occluded<true>), blocks, threads, S, rays, n, out);
    else RT_LAUNCH((k_occluded<false>), blocks, threads, S, rays, n, out);
    return hipGetLastError();
}

hipError_t launch_generate_rays(const rt_camera &cam, uint32_t width, uint32_t spp, uint64_t seed,
                                uint64_t first, uint32_t n, rt_ray *rays, hipStream_t stream) {
    hipLaunchKernelGGL(k_generate_rays, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cam, width,
                       spp, seed, first, n, rays);
    return hipGetLastError();
}

hipError_t launch_generate(const FrameDev &F, const rt_camera &cam, const Queue &q, uint32_t *counts,
                           uint32_t n_local, uint32_t s0, uint32_t s_count, hipStream_t stream) {
    uint32_t padded = (n_local + 63) / 64 * 64;
    hipLaunchKernelGGL(k_generate, dim3((padded + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, F, cam, q,
                       counts, n_local, s0, s_count);
    return hipGetLastError();
}

hipError_t launch_bounce(const SceneDev &S, const LaunchShape &L, const Queue &qin, const Queue &qout,
                         uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed, float4 *radiance,
                         unsigned long long *counters, hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) {
        if (counters)
            RT_LAUNCH((k_bounce<true, true>), L.grid, threads, S, qin, qout, counts, bounce, last, seed, radiance,
                      counters);
        else
            RT_LAUNCH((k_bounce<false, true>), L.grid, threads, S, qin, qout, counts, bounce, last, seed, radiance,
                      counters);
    } else {
        if (counters)
            RT_LAUNCH((k_bounce<true, false>), L.grid, threads, S, qin, qout, counts, bounce, last, seed, radiance,
                      counters);
        else
            RT_LAUNCH((k_bounce<false, false>), L.grid, threads, S, qin, qout, counts, bounce, last, seed, radiance,
                      counters);
    }
    return hipGetLastError();
}

hipError_t launch_resolve(const FrameDev &F, const float4 *radiance, uint32_t n_ranks, uint32_t s_count,
                          int finish, float4 *rgba, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3((n_ranks + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, F, radiance,
                       n_ranks, s_count, finish, rgba);
    return hipGetLastError();
}

} // namespace rt
