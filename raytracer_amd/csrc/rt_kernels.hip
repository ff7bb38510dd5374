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
// serialising on one word.  The wave count is a multiple of kShards and
// every kShards consecutive waves cover all sub-queues (shard_of_wave).
#include <mutex>
#include <type_traits>

#include "rt_device.hpp"
#include "rt_kernels.hpp"
#include "rt_probe.hpp" // PROF(...), opaque(), probe::double_*: measuring aids, all empty in the shipped build

namespace rt {

constexpr int kBlock = 256;
// lanes holding a leaf that trigger a leaf round (64 = wait for every lane)
#ifndef RT_LEAF_LANES
#define RT_LEAF_LANES 32
#endif
constexpr uint32_t kLeafLanes = RT_LEAF_LANES;
#ifndef RT_LEAF_LANES_GLOBAL
#define RT_LEAF_LANES_GLOBAL 32
#endif
constexpr uint32_t kDead = 0xFFFFFFFFu;

RT_DEV uint32_t lane_rank(uint64_t mask) { // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// wave uniform by construction; readfirstlane tells the compiler so (scalar registers, scalar branches)
RT_DEV uint32_t wave_index() {
    return __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}
RT_DEV uint32_t wave_total() { return gridDim.x * (blockDim.x >> 6); }
// Sub-queue of wave w.  Every kShards consecutive waves cover all sub-queues, and the cover
// shifts by RT_SHARD_SHIFT from one group to the next, so that a sub-queue is served by waves
// of every slot of a workgroup: a SIMD favours its oldest wave, and with plain w % kShards the
// sub-queues of a workgroup's last-started waves were read out 30 % later than those of its
// first (a wave only reads the blocks of its own sub-queue in the per-bounce kernel).
#ifndef RT_SHARD_SHIFT
#define RT_SHARD_SHIFT 4
#endif
RT_DEV uint32_t shard_of_wave(uint32_t w) { return (w + (uint32_t)RT_SHARD_SHIFT * (w / kShards)) % kShards; }

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
    static constexpr int kWaves = (int)kGlobalWaves;
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
        for (uint32_t i = threadIdx.x; i < 4 * n; i += blockDim.x) { // a launch may bring fewer threads
            float4 x = S.nodes[i]; // record i / 4, quarter i % 4
            // The two child references (.w of quarters 0 and 1) get a copy of their leaf flag in bit 15:
            // their low half is then the 16-bit stack entry as it stands (see push and pop).
            if ((i & 2u) == 0) {
                const uint32_t r = __float_as_uint(x.w);
                x.w = __uint_as_float(r | (r >> 16));
            }
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

struct Counters {
    uint32_t nodes;
    uint32_t prims[8];
    // 0 interior steps, 1 lanes stepping in them, 2 leaf rounds, 3 primitive iterations of
    // leaf rounds (longest lane), 4 lanes x primitives tested, 5 cycles in interior steps,
    // 6 cycles in leaf rounds, 7 cycles in shading batches (part of 8), 8 cycles in feed turns,
    // 9 cycles in all, 10 feed turns, 11 rays
    // 12 cycles of a feed turn after the feed returned (reciprocal direction, enclosing media, root box),
    // 13 cycles in Blocks::take, 14 cycles waiting for the new rays' queue entries, 15 cycles parking
    // 16-21 plain shading batches, 24-29 general ones: +0 batches, +1 cycles until the parked records
    // are there, +2 until the queue entries are, +3 in shade(), +4 waiting for the tail atomic,
    // +5 in the producer's tests and the queue stores; 22 cycles saving lane state, 23 how often,
    // 30 cycles reloading it
    PROF(unsigned long long p[32];)
};

constexpr uint32_t kNone = 0xFFFFFFFFu;
// finished or empty lanes that trigger a refill while unread rays remain
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 24
#endif
constexpr uint32_t kRefillLanes = RT_REFILL_LANES;
// the same for the kernels that read the tree from global memory: their waves wait on memory
// rather than on the ALUs, so an idle lane costs more and a refill turn less
#ifndef RT_REFILL_LANES_GLOBAL
#define RT_REFILL_LANES_GLOBAL 24
#endif
#ifndef RT_REFILL_LANES_GLOBAL_HIT
#define RT_REFILL_LANES_GLOBAL_HIT 12 // closest-hit queries without shading: a refill turn is cheap
#endif
#ifndef RT_REFILL_LANES_GLOBAL_TRACE
#define RT_REFILL_LANES_GLOBAL_TRACE 20 // the trace launch of a two-kernel bounce (measured: 12 -1.6 %, 28 -3 %)
#endif
template <bool LDSN>
constexpr uint32_t refill_lanes_of() { return LDSN ? kRefillLanes : (uint32_t)RT_REFILL_LANES_GLOBAL; }
#ifndef RT_PRIMARY_REFILL_LANES
#define RT_PRIMARY_REFILL_LANES 65
#endif
constexpr uint32_t kPrimaryRefillLanes = RT_PRIMARY_REFILL_LANES;
#ifndef RT_PRIMARY_REFILL_LANES_GLOBAL
#define RT_PRIMARY_REFILL_LANES_GLOBAL 16
#endif
#ifndef RT_BOUNCE_REFILL_LANES_GLOBAL
#define RT_BOUNCE_REFILL_LANES_GLOBAL 32 // a refill turn of the bounce kernel also parks and shades
#endif
template <bool LDSN>
constexpr uint32_t primary_refill_lanes_of() { return LDSN ? kPrimaryRefillLanes : (uint32_t)RT_PRIMARY_REFILL_LANES_GLOBAL; }

constexpr uint32_t kBestRare = 0x80000000u;

// What a lane carries for the ray it is tracing.
struct Lane {
    Ray ray;       // t_max shrinks to the closest accepted hit
    uint32_t best; // leaf slot of that hit (| kBestRare if it needs the general shader), or RT_MISS
    float a0, a1, a2;
    bool has_ray;
};

// A lane's whole traversal state.  cur: record index of the interior node to visit,
// kLeafRef | first primitive of a leaf to test, or kNone when the ray is finished (or
// the lane has none); sp: stacked far children.
struct Trav {
    Lane L;
    v3 inv; // 1 / direction, bvh.rs:741-750
    uint32_t cur, sp;
};

// A wave's share of a ray array: blocks of 64 consecutive entries at a fixed
// stride, consumed front to back a few entries at a time.  All members are wave
// uniform.
struct Blocks {
    uint32_t blk, off, n, stride;
    // With a cursor the wave's first block is its static one and every further block is
    // claimed from a counter the waves of the stream share (offset stride: right after the
    // first blocks), so that a wave whose rays run long leaves its later blocks to others.
    uint32_t *cursor = nullptr;
    // claim c (0, 1, ...) is block (first + c) * mul + add of the array: a counter may serve
    // every mul-th block only, so that several counters share the claims of one array (one
    // address takes about 90 atomics per microsecond)
    uint32_t first = 0, mul = 1, add = 0;
    // Sub-queues (bounce kernels that keep no out-queue): `base` is the offset of the current
    // sub-queue's entries, and with `sizes` set a wave whose sub-queue is read out goes on with
    // the next one's cursor (sizes[s * kCountStride], cursor right behind it, entries at s * cap),
    // so that the sub-queues end together however unevenly the work fell on them.
    uint32_t base = 0;
    const uint32_t *sizes = nullptr;
    uint32_t shard = 0, cap = 0, hops = 0;
    RT_DEV bool more() const { return blk < n; }
#ifndef RT_STEAL_MIN
#define RT_STEAL_MIN 256
#endif
    static constexpr uint32_t kStealMin = RT_STEAL_MIN; // blocks a sub-queue must have left to be taken from (mul 1, add 0)
    RT_DEV void claim() {
        static_assert(kShards == 32, "one ballot word covers the sub-queues");
        const uint32_t lane = threadIdx.x & 63;
        for (;;) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(cursor, 1u);
            blk = ((first + __builtin_amdgcn_readfirstlane(c)) * mul + add) * 64u;
            if (blk < n || !sizes || hops >= 2 * kShards) break;
            // This sub-queue is read out.  Lane s looks at sub-queue s (size and cursor), and the
            // wave goes on with the next one that has blocks left: one round trip, not one per
            // sub-queue, and no atomics on cursors that have nothing to give (at the end of a
            // launch every wave comes by, and atomics on one address go one at a time).
            hops++;
            uint32_t size_s = 0, seen = 0;
            if (lane < kShards) {
                size_s = sizes[(size_t)lane * kCountStride];
                seen = __hip_atomic_load(sizes + (size_t)lane * kCountStride + 1, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
            }
            // only sub-queues with plenty left are worth the trip: paths started from a sub-queue's
            // last few blocks by waves that were about to finish only make the launch end later
            // (measured: threshold 16 costs a scene with even sub-queues 4 %, 256 nothing), and
            // thousands of waves going for them spend their atomics for nothing
            const int32_t blocks_left = (int32_t)((size_s + 63u) / 64u) - (int32_t)(first + seen);
            const uint32_t left = (uint32_t)__ballot(lane < kShards && blocks_left >= (int32_t)kStealMin);
            if (left == 0) {
                blk = n;
                break;
            }
            const uint32_t turn = (shard + 1) % kShards; // search from the neighbour on
            const uint32_t rotated = turn ? (left >> turn) | (left << (kShards - turn)) : left;
            shard = (turn + (uint32_t)__builtin_ctz(rotated)) % kShards;
            n = (uint32_t)__builtin_amdgcn_readlane((int)size_s, (int)shard);
            cursor = const_cast<uint32_t *>(sizes) + (size_t)shard * kCountStride + 1;
            base = shard * cap;
        }
        off = 0;
    }
    // a wave with no block of its own looks for one right away
    RT_DEV void start() {
        if (cursor && sizes && blk >= n) claim();
    }
    RT_DEV void next_block() {
        if (cursor) {
            claim();
        } else {
            blk += stride;
            off = 0;
        }
    }
    // Hands the next unread entries to the lanes that want one, in lane order; kNone for a lane
    // that gets none.  When the current block runs out the rest of the lanes are served from the
    // next one in the same call: a block of 64 ends in the middle of two refills out of three,
    // and a refill that brings a handful of rays is followed by another turn at once.
    RT_DEV uint32_t take(bool want) {
        const uint64_t m = __ballot(want);
        const uint32_t n_want = (uint32_t)__popcll(m);
        const uint32_t len = blk < n ? min(64u, n - blk) : 0u;
        const uint32_t avail = len - off;
        const uint32_t rank = lane_rank(m);
        uint32_t idx = (want && rank < avail) ? base + blk + off + rank : kNone;
        off += min(n_want, avail);
        if (len != 0 && off >= len) {
            next_block();
#ifndef RT_TAKE_ONE_BLOCK
            if (n_want > avail) {
                const uint32_t len2 = blk < n ? min(64u, n - blk) : 0u;
                const uint32_t more = min(n_want - avail, len2);
                if (want && rank >= avail && rank - avail < more) idx = base + blk + (rank - avail);
                off = more;
                if (len2 != 0 && off >= len2) next_block();
            }
#endif
        }
        return idx;
    }
};

// A ray that starts inside a ball (BallDev), after the media that enclose the scene have been
// tested (ray.t_max, best0 = what they left): if the ball's own primitives decide the closest
// hit, t_out / won say how and true is returned; the tree need not be walked for this ray.
// They do when
//   - with a shell: the shell is hit at all (t_max = infinity), so the ray does leave through
//     it, and testing the ball's primitives in either order gives the same winner and distance
//     (the traversal may meet them in either order; only an exact tie tells the orders apart);
//   - without a shell: the medium scatters the ray (nothing else can come before that).
// won: 0 / 1 = the ball's primitive in leaf order, 2 = nothing of the ball (the ray ends
// inside it or in an enclosing medium: what came in stands).
RT_DEV bool ball_resolve(const SceneDev &S, const BallDev &b, const Ray &ray, float &t_out, uint32_t &won) {
    float t0, t1;
    if (!sphere_roots(ray.o - mk(b.c[0], b.c[1], b.c[2]), ray.d, b.r, t0, t1)) return false;
    float t_shell = 0.0f;
    if (b.sphere >= 0 && !sphere_pick(t0, t1, kInf, t_shell)) return false;
    // the medium's wish does not depend on the ray's end
    const LeafPrim lm = load_const(S.leaf + (b.best[b.medium] & ~kBestRare));
    const float len = length(ray.d);
    const float hit_dist = __uint_as_float(lm.index) * p_log(medium_uniform(ray, lm.prim));
    float t_end[2];
    uint32_t win[2];
#pragma unroll
    for (uint32_t order = 0; order < 2; order++) {
        float t_max = ray.t_max;
        uint32_t w = 2;
        for (uint32_t q = 0; q < b.n; q++) {
            const uint32_t j = order == 0 ? q : b.n - 1 - q;
            float t;
            bool hit;
            if ((int32_t)j == b.sphere) {
                hit = sphere_pick(t0, t1, t_max, t);
            } else {
                float t_in = t0, t_out_b = t1;
                hit = medium_span(t_in, t_out_b, t_max) && medium_scatter(t_in, t_out_b, t_max, len, hit_dist, t);
            }
            if (hit) {
                t_max = t;
                w = j;
            }
        }
        t_end[order] = t_max;
        win[order] = w;
    }
    if (b.n == 2 && (win[0] != win[1] || __float_as_uint(t_end[0]) != __float_as_uint(t_end[1]))) return false;
    if (b.sphere < 0 && win[0] != 0) return false;
    t_out = t_end[0];
    won = win[0];
    return true;
}

// What the maker of a ray finds out about it before it is queued, where all lanes of a full wave
// work (camera rays, scattered rays) instead of in the traversal that picks the ray up later
// with a few lanes: the media that enclose the scene (the same tests in the same order on the
// same ray as trace_stream makes for rays that come without this) and, for a ray that starts
// inside a ball, its closest hit.  t_max: the ray's end after all that.  code: bits 0-7 = 0, or
// k when the ray scatters in the k-th enclosing medium; bits 8-15 = 0, or 1 + 3 ball + won when
// the ray is resolved (ball_resolve) and needs no traversal.
RT_DEV void pretest(const SceneDev &S, const Ray &ray, float &t_max, uint32_t &code) {
    Ray r = ray;
    uint32_t which = 0;
    const v3 rd = mk(0.0f, 0.0f, 0.0f); // only boxes read the reciprocal direction; these are spheres
    for (uint32_t e = 0; e < S.n_enclosing; e++) {
        const LeafPrim lp = load_const(S.leaf + S.n_tree_prims + e);
        float t;
        if (medium_test(S, lp, r, rd, t)) {
            r.t_max = t;
            which = e + 1;
        }
    }
    code = which;
    // balls do not overlap: the first that holds the origin is the only one.  Every lane goes
    // through all of them, so that k and the ball's record stay wave uniform (scalar registers).
    bool found = false;
    for (uint32_t k = 0; k < S.n_balls; k++) {
        const BallDev &b = S.balls[k];
        const float dx = r.o.x - b.c[0], dy = r.o.y - b.c[1], dz = r.o.z - b.c[2];
        if (found || !(dx * dx + dy * dy + dz * dz < b.r_in2)) continue;
        found = true;
        float t;
        uint32_t won;
        if (ball_resolve(S, b, r, t, won)) {
            r.t_max = t;
            code |= (1u + 3u * k + won) << 8;
        }
    }
    t_max = r.t_max;
}
// What the code says about the ray's hit so far: leaf slot (| kBestRare) of the ball's primitive
// or the enclosing medium it ends in, or RT_MISS; resolved: no traversal needed.
RT_DEV uint32_t pretest_best(const SceneDev &S, uint32_t code, bool &resolved) {
    const uint32_t which = code & 0xffu, r = (code >> 8) & 0xffu;
    resolved = r != 0;
    if (r != 0 && (r - 1u) % 3u != 2u) {
        // chosen among the few candidates, which are scalar values, instead of gathered with the lane's own index
        uint32_t v = RT_MISS;
        for (uint32_t k = 0; k < S.n_balls; k++) {
            v = r - 1u == 3u * k ? S.balls[k].best[0] : v;
            v = r - 1u == 3u * k + 1u ? S.balls[k].best[1] : v;
        }
        return v;
    }
    if (which == 0) return RT_MISS;
    return (S.n_tree_prims + which - 1) | (((S.enclosing_rare >> (which - 1)) & 1u) ? kBestRare : 0u);
}
// counters of a resolved ray: one test per primitive of its ball, no node
RT_DEV void count_resolved(const SceneDev &S, uint32_t code, Counters &cnt) {
    const uint32_t r = (code >> 8) & 0xffu;
    if (r == 0) return;
    const BallDev &b = S.balls[(r - 1u) / 3u];
    cnt.prims[RT_MEDIUM]++;
    if (b.sphere >= 0) cnt.prims[RT_SPHERE]++;
}

// <BVH as Primitive>::intersect / does_intersect, shapes/src/bvh.rs:736-863, for a
// stream of rays, with the ray shortened on every accepted hit (ties go to the
// primitive tested later).
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
// Scheduled for a 64-wide wave.  Each lane is stepping (at an interior record),
// holding (at a leaf), or free (its ray is finished, or it has none).  Every
// turn of the loop the wave does one thing, chosen by head count:
//   * enough free lanes and unread rays left (or nothing else to do): `feed`
//     takes the finished rays' results and hands out new rays;
//   * enough lanes holding a leaf (or nobody stepping): those lanes test their
//     leaf's primitives, so the primitive code (a switch over seven kinds) runs
//     for many lanes at once;
//   * otherwise one interior step for the stepping lanes.
// Rays therefore enter and leave a wave one by one instead of in groups of 64:
// a long traversal does not hold 63 finished lanes idle.
//
// Feed: bool more() (wave uniform: unread rays remain),
// bool turn(Trav &, bool free): for free lanes, consume the finished ray if
// L.has_ray, then load the next ray or clear has_ray; returns true on lanes that
// received a ray (it may save the whole of Trav to memory and reload it), and
// bool drain(), called when nothing is left to hand out: true if the feed found more after
// all (only a feed that turns results into new rays), and void finish(), called once at the end.
//
// WIDE (trees in global memory, no counters): an interior step reads a 128-byte record with the boxes
// of a node's four grandchildren (SceneDev::quads) and so goes down two levels of the tree.  The
// memory system delivers a random 128-byte record as fast as a random 64-byte one, so half as many
// dependent fetches walk the same tree.  The children's own boxes are not tested: a grandchild's
// box lies inside its parent's, every term of the box test is monotone in the box, and a ray only
// gets shorter, so a grandchild that passes its test when it is due implies that its parent passed
// when it was -- provided no term is NaN.  NaN needs 0 x infinity, i.e. a direction component that
// is zero (infinite reciprocal), or a non-finite origin or reciprocal; such rays (`exact`) walk the
// two-box records instead, in the same loop.  The four entries are visited in the order the
// reference would reach them: the near child's side first (split axis of the node), within a side
// the near grandchild first (split axis of the child).
template <bool ANY, bool COUNT, bool LDSN, bool WIDE, class Feed>
RT_DEV void trace_stream(const SceneDev &S, const TreeView<LDSN> &tv, Feed &feed, Counters &cnt,
                         const uint32_t refill_lanes = refill_lanes_of<LDSN>()) {
    static_assert(!WIDE || (!LDSN && !COUNT), "four-box records: global memory, no counters");
    constexpr int kStride = Shape<LDSN>::kThreads;
    constexpr uint32_t leaf_lanes = LDSN ? kLeafLanes : (uint32_t)RT_LEAF_LANES_GLOBAL;
    using RefT = typename Shape<LDSN>::RefT;
    using StackT = typename std::conditional<LDSN, int16_t, uint32_t>::type;
    constexpr uint32_t kLeafIndex = LDSN ? 0x7fffu : ~kLeafRef;
    RefT *const s_ref = tv.ref;
    float *const s_near = tv.near;
    const uint32_t levels = tv.levels;
    uint32_t spill_ref[kSpill];
    float spill_near[kSpill];
    Trav T;
    Lane &L = T.L;
    v3 &inv = T.inv;
    uint32_t &sp = T.sp, &cur = T.cur;
    L.has_ray = false;
    L.best = RT_MISS;
    L.a0 = L.a1 = L.a2 = 0.0f;
    L.ray.o = L.ray.d = mk(0.0f, 0.0f, 0.0f);
    L.ray.t_max = L.ray.time = 0.0f;
    inv = mk(0.0f, 0.0f, 0.0f);
    bool negx = false, negy = false, negz = false;
    uint32_t negbits = 0; // LDSN: the same three as bits 0-2, one shift picks the bit of a record's split axis
    bool exact = false; // WIDE: this lane's ray walks the two-box records
    sp = 0;
    cur = kNone;

    auto push = [&](uint32_t r, float t) {
        if (sp < levels) {
            s_ref[sp * kStride] = (RefT)r; // LDSN: the low half already is the entry (tree_view)
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
            // the LDS read is unconditional (clamped level) and the private array is read in
            // its own branch: two address spaces behind one selected pointer would turn both
            // into slow generic loads
            const uint32_t level = min(sp, levels - 1);
            // LDSN: a 16-bit entry is read sign-extended: a leaf comes back as 0xFFFF8000 | first primitive
            // (bit 31 set as in a record's reference; the leaf round masks the index to 15 bits)
            uint32_t e = (uint32_t)(int32_t)(StackT)s_ref[level * kStride];
            float t = s_near[level * kStride];
            asm volatile("" : "+v"(e), "+v"(t)); // keeps both LDS reads DS instructions (see above)
            if (sp >= levels) {
                e = spill_ref[sp - levels];
                t = spill_near[sp - levels];
            }
            if (t < L.ray.t_max) {
                r = e;
                break;
            }
        }
        return r;
    };

    PROF(unsigned long long tp = __builtin_amdgcn_s_memtime(); const unsigned long long tp0 = tp;)
    for (;;) {
        {
            const bool busy = cur != kNone;
            const uint32_t n_busy = (uint32_t)__popcll(__ballot(busy));
            if (n_busy == 0 || (64u - n_busy >= refill_lanes && feed.more())) {
                const bool fresh = feed.turn(T, !busy);
                PROF(const unsigned long long t_fresh = __builtin_amdgcn_s_memtime();)
                if (fresh) { // bvh.rs:741-762: reciprocal direction, then the root's own box
                    inv = mk(1.0f / L.ray.d.x, 1.0f / L.ray.d.y, 1.0f / L.ray.d.z);
                    sp = 0;
                    // Media that enclose the scene are not in the tree: every ray tests them before it
                    // starts traversing, and a hit shortens it.  A feed whose rays were tested by
                    // their producer (Feed::kPretested) hands over the outcome in L.best / L.ray.t_max.
                    // A ray that starts inside a ball may be resolved by the ball's own primitives
                    // (ball_resolve): it then needs no traversal at all.
                    bool occluded = false;
                    if constexpr (Feed::kPretested) {
                        if (COUNT) {
                            cnt.prims[RT_MEDIUM] += S.n_enclosing;
                            count_resolved(S, feed.code, cnt);
                        }
                        occluded = (feed.code >> 8) != 0; // resolved: skip the tree
                    } else {
                        L.best = RT_MISS;
                        for (uint32_t e = 0; e < S.n_enclosing; e++) {
                            const uint32_t slot = S.n_tree_prims + e;
                            const LeafPrim lp = load_const(S.leaf + slot);
                            if (COUNT) cnt.prims[RT_MEDIUM]++;
                            float t;
                            if (medium_test(S, lp, L.ray, inv, t)) {
                                L.ray.t_max = t;
                                L.best = slot | ((lp.kind & kRareShade) ? kBestRare : 0u);
                                L.a0 = L.a1 = L.a2 = 0.0f;
                                if (ANY) occluded = true;
                            }
                        }
                        if constexpr (!ANY) {
                            bool found = false; // the first ball that holds the origin is the only one
                            for (uint32_t k = 0; k < S.n_balls; k++) {
                                const BallDev &b = S.balls[k];
                                const float dx = L.ray.o.x - b.c[0], dy = L.ray.o.y - b.c[1], dz = L.ray.o.z - b.c[2];
                                if (found || !(dx * dx + dy * dy + dz * dz < b.r_in2)) continue;
                                found = true;
                                float t;
                                uint32_t won;
                                if (ball_resolve(S, b, L.ray, t, won)) {
                                    L.ray.t_max = t;
                                    if (won != 2u) {
                                        L.best = b.best[won];
                                        L.a0 = L.a1 = L.a2 = 0.0f;
                                    }
                                    if (COUNT) count_resolved(S, (1u + 3u * k + won) << 8, cnt);
                                    occluded = true; // resolved: skip the tree
                                }
                            }
                        }
                    }
                    if (S.n_nodes && !occluded) {
                        if (COUNT) cnt.nodes++;
                        float t;
                        float4 r0 = make_float4(S.root_lo[0], S.root_lo[1], S.root_lo[2], 0.0f);
                        float4 r1 = make_float4(S.root_hi[0], S.root_hi[1], S.root_hi[2], 0.0f);
                        if constexpr (WIDE) {
                            auto plain = [](float v) { return fabsf(v) < kInf && v != 0.0f; }; // false for NaN too
                            auto finite = [](float v) { return fabsf(v) < kInf; };
                            exact = !(plain(inv.x) && plain(inv.y) && plain(inv.z) && finite(L.ray.o.x) &&
                                      finite(L.ray.o.y) && finite(L.ray.o.z));
                        }
                        if (slab(r0, r1, L.ray, inv, inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f, t))
                            cur = (WIDE && !exact) ? S.root_quad : S.root_ref;
                    }
                    PROF(if (COUNT) cnt.p[11]++;)
                }
                negx = inv.x < 0.0f;
                negy = inv.y < 0.0f;
                negz = inv.z < 0.0f;
                if constexpr (LDSN) negbits = (negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u);
                PROF(if (COUNT) {
                    const unsigned long long t = __builtin_amdgcn_s_memtime();
                    cnt.p[8] += t - tp;
                    cnt.p[12] += t - t_fresh;
                    cnt.p[10]++;
                    tp = t;
                })
                if (__ballot(L.has_ray) == 0) {
                    if (feed.more()) continue; // only dead entries came: ask again
                    if (feed.drain()) continue; // a feed that makes rays of its own results
                    break;
                }
            }
        }
        // interior steps, bvh.rs:778-795 for both children, until enough lanes hold a leaf,
        // enough lanes are free for a refill, or nobody is left stepping
        const bool can_refill = feed.more();
        for (;;) {
            const bool stepping = (cur >> 31) == 0;
            const uint64_t m_step = __ballot(stepping);
            if (m_step == 0) break;
            const uint32_t n_step = (uint32_t)__popcll(m_step);
            const uint32_t n_hold = (uint32_t)__popcll(__ballot(!stepping && cur != kNone));
            // A leaf round is due when leaf_lanes lanes hold a leaf -- or 4 / RT_LEAF_SHARE of the lanes
            // that have a ray at all: a wave with 30 rays left never gets to 32 holders, and without
            // this rule its holders waited until the last stepping lane had reached a leaf, every time
            // (at the end of a launch every wave is such a wave).  Measured on the headline frame and
            // its eighth: share 4/6 ... 4/8 +1.0 % and -4 %, 4/5 +0.3 %, 4/12 -0.9 %; only once the
            // queue is read out: nothing.
#ifndef RT_LEAF_SHARE
#define RT_LEAF_SHARE 7
#endif
            if (n_hold >= leaf_lanes) break;
            if (n_hold != 0 && n_hold * (uint32_t)RT_LEAF_SHARE >= (n_step + n_hold) * 4u) break;
            if (can_refill && 64u - n_step - n_hold >= refill_lanes) break;
            PROF(if (COUNT) { cnt.p[0]++; cnt.p[1] += stepping ? 1 : 0; })
            // Several steps per head count: the ballots, counts and branches above are a tenth of a step's
            // instructions, and a lane that reaches a leaf a step or two before the wave looks again loses
            // less than that (measured: 2 steps +1.2 %, 3 +1.8 %, 4-8 the same as 3; with the share rule
            // for leaf rounds above 4, 6 and 8 are another 0.5 % ahead of 3).
#ifndef RT_STEPS_PER_CHECK
#define RT_STEPS_PER_CHECK 6
#endif
#pragma unroll 1
            for (int rep = 0; rep < RT_STEPS_PER_CHECK; rep++) {
            const bool stepping = (cur >> 31) == 0; // (shadows the head count's: a lane may have reached a leaf since)
            if (WIDE && stepping && !exact) {
                const float4 *q = S.quads + 8 * (size_t)cur;
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7];
                // entries 0, 1: the first child's children (or the child itself, a leaf, and nothing);
                // entries 2, 3: the second child's
                const uint32_t e0 = __float_as_uint(q0.w), e1 = __float_as_uint(q1.w), e2 = __float_as_uint(q2.w),
                               e3 = __float_as_uint(q3.w), meta = __float_as_uint(q4.w);
                float t0, t1, t2, t3;
                const bool h0 = slab(q0, q1, L.ray, inv, negx, negy, negz, t0);
                const bool h1 = slab(q2, q3, L.ray, inv, negx, negy, negz, t1) && e1 != kNone;
                const bool h2 = slab(q4, q5, L.ray, inv, negx, negy, negz, t2);
                const bool h3 = slab(q6, q7, L.ray, inv, negx, negy, negz, t3) && e3 != kNone;
                auto neg_of = [&](uint32_t axis) { return axis == 0 ? negx : (axis == 1 ? negy : negz); };
                const bool second_side = neg_of(meta & 3u);      // the node's second child is nearer
                const bool swap_a = neg_of((meta >> 2) & 3u);    // the first child's second child is nearer
                const bool swap_b = neg_of((meta >> 4) & 3u);
                // the sides in their own visiting order
                const uint32_t a0r = swap_a ? e1 : e0, a1r = swap_a ? e0 : e1;
                const float a0t = swap_a ? t1 : t0, a1t = swap_a ? t0 : t1;
                const bool a0h = swap_a ? h1 : h0, a1h = swap_a ? h0 : h1;
                const uint32_t b0r = swap_b ? e3 : e2, b1r = swap_b ? e2 : e3;
                const float b0t = swap_b ? t3 : t2, b1t = swap_b ? t2 : t3;
                const bool b0h = swap_b ? h3 : h2, b1h = swap_b ? h2 : h3;
                // all four, last to first: whatever comes later in the visiting order waits on the stack
                uint32_t cand = kNone;
                float cand_t = 0.0f;
                auto offer = [&](bool h, uint32_t r, float t) {
                    if (h) {
                        if (cand != kNone) push(cand, cand_t);
                        cand = r;
                        cand_t = t;
                    }
                };
                offer(second_side ? a1h : b1h, second_side ? a1r : b1r, second_side ? a1t : b1t);
                offer(second_side ? a0h : b0h, second_side ? a0r : b0r, second_side ? a0t : b0t);
                offer(second_side ? b1h : a1h, second_side ? b1r : a1r, second_side ? b1t : a1t);
                offer(second_side ? b0h : a0h, second_side ? b0r : a0r, second_side ? b0t : a0t);
                cur = cand != kNone ? cand : pop();
            } else if (stepping) {
                bool need_pop = true;
                {
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
                    probe::double_slab(q0, q1, q2, q3, L.ray, inv, negx, negy, negz);
                    const bool h0 = slab(q0, q1, L.ray, inv, negx, negy, negz, t0);
                    const bool h1 = slab(q2, q3, L.ray, inv, negx, negy, negz, t1);
                    const uint32_t r0 = __float_as_uint(q0.w), r1 = __float_as_uint(q1.w);
                    const uint32_t axis = __float_as_uint(q2.w);
                    // second child is nearer.  One shift of the three signs as bits is 7 instructions shorter than
                    // selecting among the lane masks; the 80-register shape pays more for the register that
                    // holds the bits (measured: tree in LDS +0.4 %, tree in global memory -3.8 %).
                    const bool neg = LDSN ? ((negbits >> axis) & 1u) != 0
                                          : (axis == 0 ? negx : (axis == 1 ? negy : negz));
                    if (h0 && h1) {
                        push(neg ? r0 : r1, neg ? t0 : t1);
                        cur = neg ? r1 : r0;
                        need_pop = false;
                    } else if (h0 || h1) {
                        cur = h0 ? r0 : r1;
                        need_pop = false;
                    }
                }
                if (need_pop) cur = pop();
            }
            }
        }
        PROF(if (COUNT) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            cnt.p[5] += t - tp;
            tp = t;
        })
        // leaf round, bvh.rs:763-776: when enough lanes hold a leaf (the step loop's rule; nobody
        // stepping is its case of all the lanes with a ray holding one)
        const bool holding = (cur >> 31) != 0 && cur != kNone;
        const uint32_t n_hold = (uint32_t)__popcll(__ballot(holding));
        const uint32_t n_step_left = (uint32_t)__popcll(__ballot((cur >> 31) == 0));
        if (n_hold != 0 &&
            (n_hold >= leaf_lanes || n_hold * (uint32_t)RT_LEAF_SHARE >= (n_step_left + n_hold) * 4u)) {
            PROF(uint32_t my_iters = 0;)
            if (holding) {
                bool stop = false;
                for (uint32_t i = cur & kLeafIndex;; i++) {
                    PROF(my_iters++;)
                    LeafPrim lp = S.leaf[i];
                    float t, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                    if (COUNT) {
                        uint32_t k = lp.kind & 7u;
#pragma unroll
                        for (uint32_t q = 0; q < 8; q++) cnt.prims[q] += (k == q) ? 1u : 0u;
                    }
                    probe::double_prim(S, lp, L.ray, inv);
                    // Origin and direction go through opaque copies inside the loop: otherwise everything
                    // a kind derives from the ray alone (the triangle's permutation and shear with three
                    // divisions, most of the medium's hash, the direction's length) is hoisted in front of
                    // the loop and runs in every leaf round, for every lane, whether or not a primitive of
                    // that kind turns up, and the loop starts with some twenty register copies of the ray
                    // for the kinds that permute it.  (Measured, frames of the five test scenes: no copy
                    // -1.3 to -5 %, d.z alone the base, all six +0.3 to +1.5 %, the reciprocals as well the same.)
                    Ray pr = L.ray;
                    pr.o.x = opaque(pr.o.x); pr.o.y = opaque(pr.o.y); pr.o.z = opaque(pr.o.z);
                    pr.d.x = opaque(pr.d.x); pr.d.y = opaque(pr.d.y); pr.d.z = opaque(pr.d.z);
                    const bool hit = prim_test(S, lp, pr, inv, t, b0, b1, b2);
                    if (hit) {
                        L.ray.t_max = t;
                        L.best = i | ((lp.kind & kRareShade) ? kBestRare : 0u);
                        L.a0 = b0; L.a1 = b1; L.a2 = b2;
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
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                cnt.p[6] += t - tp;
                tp = t;
            })
        }
    }
    feed.finish();
    PROF(if (COUNT) cnt.p[9] += __builtin_amdgcn_s_memtime() - tp0;)
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
    PROF(for (int k = 0; k < 32; k++) {
        unsigned long long x = c.p[k];
        const bool per_lane = k == 1 || k == 4 || k == 11;
        if (per_lane)
            for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out[32 + k], x);
    })
}

// rt_intersect: hit records out, rays in
struct HitFeed {
    static constexpr bool kPretested = false;
    const SceneDev &S;
    const rt_ray *rays;
    rt_hit *hits;
    Blocks blocks;
    uint32_t idx;
    RT_DEV bool more() const { return blocks.more(); }
    RT_DEV bool turn(Trav &T, bool free) {
        Lane &L = T.L;
        if (free && L.has_ray) {
            float4 h0, h1;
            if (L.best != RT_MISS) {
                LeafPrim lp = S.leaf[L.best & ~kBestRare];
                Surf sf;
                prim_finalize(S, lp, L.ray, L.ray.t_max, L.a0, L.a1, L.a2, sf);
                h0 = make_float4(sf.t, __uint_as_float(lp.prim), sf.u, sf.v);
                h1 = make_float4(sf.n.x, sf.n.y, sf.n.z, __uint_as_float(sf.flags));
            } else {
                h0 = make_float4(kInf, __uint_as_float(RT_MISS), 0.0f, 0.0f);
                h1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            float4 *o = reinterpret_cast<float4 *>(hits + idx);
            o[0] = h0;
            o[1] = h1;
        }
        const uint32_t next = blocks.take(free);
        if (free) {
            idx = next;
            L.has_ray = next != kNone;
            if (L.has_ray) L.ray = load_ray(rays, next);
        }
        return free && L.has_ray;
    }
    RT_DEV bool drain() { return false; }
    RT_DEV void finish() {}
};

// A wave's share of a flat ray array: block w first; with cursors (kShards words, one per
// 64-byte line, zero at launch) every further block is claimed, wave w drawing on counter
// k (below), which serves the blocks b with b % kShards == k.
RT_DEV Blocks flat_blocks(uint32_t n, uint32_t *cursors, bool mix_slots) {
    const uint32_t w = wave_index(), total = wave_total();
    Blocks b{w * 64, 0u, n, total * 64};
    if (cursors && total % kShards == 0) {
        // 16-wave workgroups: a counter shared by waves of every workgroup slot (see
        // shard_of_wave; +11 ... 14 % on a small scene).  With four waves per workgroup the
        // slots say nothing about a wave's standing on its SIMD, and the plain deal measured
        // 2 ... 4 % better on a large scene.
        const uint32_t k = mix_slots ? shard_of_wave(w) : w % kShards;
        b.cursor = cursors + k * kCountStride;
        b.first = total / kShards;
        b.mul = kShards;
        b.add = k;
    }
    return b;
}

template <bool COUNT, bool LDSN, bool WIDE = false>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_intersect(
    SceneDev S, const rt_ray *rays, uint32_t n, rt_hit *hits, unsigned long long *counters, uint32_t *cursor) {
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    HitFeed feed{S, rays, hits, flat_blocks(n, cursor, LDSN), kNone};
    trace_stream<false, COUNT, LDSN, WIDE>(S, tv, feed, cnt, LDSN ? kRefillLanes : (uint32_t)RT_REFILL_LANES_GLOBAL_HIT);
    if (COUNT) flush_counters(cnt, counters);
}

// rt_occluded: one byte per ray out
struct AnyFeed {
    static constexpr bool kPretested = false;
    const rt_ray *rays;
    uint8_t *out;
    Blocks blocks;
    uint32_t idx;
    RT_DEV bool more() const { return blocks.more(); }
    RT_DEV bool turn(Trav &T, bool free) {
        Lane &L = T.L;
        if (free && L.has_ray) out[idx] = L.best != RT_MISS ? 1 : 0;
        const uint32_t next = blocks.take(free);
        if (free) {
            idx = next;
            L.has_ray = next != kNone;
            if (L.has_ray) L.ray = load_ray(rays, next);
        }
        return free && L.has_ray;
    }
    RT_DEV bool drain() { return false; }
    RT_DEV void finish() {}
};

// <BVH as Primitive>::does_intersect, shapes/src/bvh.rs:801-863
template <bool LDSN, bool WIDE = false>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_occluded(
    SceneDev S, const rt_ray *rays, uint32_t n, uint8_t *out, uint32_t *cursor) {
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    AnyFeed feed{rays, out, flat_blocks(n, cursor, LDSN), kNone};
    trace_stream<true, false, LDSN, WIDE>(S, tv, feed, cnt);
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

// ---- Halton camera samples: colour/src/samplers/halton.rs + low_discrepancy.rs ------------
template <uint32_t BASE>
RT_DEV float radical_inverse(uint64_t num) { // low_discrepancy.rs:7-21
    const float inv = 1.0f / (float)BASE;
    float inv_base_n = 1.0f;
    uint64_t digits = 0;
    while (num > 0) {
        const uint64_t next = num / BASE, digit = num - next * BASE;
        digits = digits * BASE + digit;
        inv_base_n *= inv;
        num = next;
    }
    return sel_min((float)(double)digits * inv_base_n, kOneMinusEps);
}
template <uint32_t BASE>
RT_DEV float scrambled_radical_inverse(uint64_t num, const uint8_t *perm) { // low_discrepancy.rs:43-62
    const float inv = 1.0f / (float)BASE;
    float inv_base_n = 1.0f;
    uint64_t digits = 0;
    while (num > 0) {
        const uint64_t next = num / BASE, digit = num - next * BASE;
        digits = digits * BASE + perm[digit];
        inv_base_n *= inv;
        num = next;
    }
    return sel_min(inv_base_n * ((float)(double)digits + inv * (float)perm[0] / (1.0f - inv)), kOneMinusEps);
}
template <uint32_t BASE>
RT_DEV uint64_t inverse_radical_inverse(uint64_t inverse, uint32_t n_digits) { // low_discrepancy.rs:31-40
    uint64_t index = 0;
    for (uint32_t i = 0; i < n_digits; i++) {
        const uint64_t digit = inverse % BASE;
        inverse /= BASE;
        index = index * BASE + digit;
    }
    return index;
}

// get_index_for_sample + sample_dimension (halton.rs:62-106) for dimensions 0-4; dimension 1
// divides the index by base_scales[1] as pbrt does (halton.rs:98 shifts by it)
RT_DEV void halton_sample(const HaltonDev &h, uint32_t px, uint32_t py, uint32_t s, float u[5]) {
    uint64_t offset = 0;
    if (h.stride > 1) {
        const uint64_t d0 = inverse_radical_inverse<2>(px % 128u, h.exponent[0]);
        const uint64_t d1 = inverse_radical_inverse<3>(py % 128u, h.exponent[1]);
        offset = d0 * (h.stride / h.scale[0]) * h.mult_inv[0] + d1 * (h.stride / h.scale[1]) * h.mult_inv[1];
        offset %= h.stride;
    }
    const uint64_t index = offset + (uint64_t)s * h.stride;
    u[0] = radical_inverse<2>(index >> h.exponent[0]);
    u[1] = radical_inverse<3>(index / h.scale[1]);
    u[2] = scrambled_radical_inverse<5>(index, h.perm + 5);
    u[3] = scrambled_radical_inverse<7>(index, h.perm + 10);
    u[4] = scrambled_radical_inverse<11>(index, h.perm + 17);
}

// ---- Sobol' camera samples: colour/src/samplers/sobol.rs ---------------------------------------
RT_DEV float unit_float(uint32_t v) { return sel_min((float)v * 2.3283064365386963e-10f, kOneMinusEps); }

// get_index_for_sample (sobol.rs:31-74, with b = ((x << m) | y) ^ delta as in pbrt) and
// sample_dimension (sobol.rs:76-95) for dimensions 0-4.  Loop counts are uniform so the
// table reads are scalar.
RT_DEV void sobol_sample(const FrameDev &F, uint32_t px, uint32_t py, uint32_t s, float u[5]) {
    const SobolDev &T = *F.sobol;
    const uint32_t m = F.sobol_log2;
    uint64_t index = s;
    uint32_t sbits = 32u - (uint32_t)__builtin_clz(F.spp | 1u); // sample numbers are below spp
    if (m) {
        uint64_t delta = 0;
        for (uint32_t c = 0; c < sbits && 2 * m + c < 52; c++) delta ^= ((s >> c) & 1u) ? T.vdc[c] : 0ull;
        const uint64_t b = (((uint64_t)px << m) | py) ^ delta;
        index = (uint64_t)s << (2 * m);
        for (uint32_t c = 0; c < 2 * m; c++) index ^= ((b >> c) & 1u) ? T.inv[c] : 0ull;
    }
    uint32_t v[5] = {0, 0, 0, 0, 0};
    const uint32_t nbits = 2 * m + sbits < 52 ? 2 * m + sbits : 52;
    for (uint32_t c = 0; c < nbits; c++) { // sobol_sample_f32, sobol.rs:111-129
        const bool on = (index >> c) & 1u;
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] ^= on ? T.m32[k][c] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) u[k] = unit_float(v[k]);
    const float res = (float)(1u << m);
    u[0] = sel_min(sel_max(u[0] * res + 0.0f - (float)px, 0.0f), kOneMinusEps);
    u[1] = sel_min(sel_max(u[1] * res + 0.0f - (float)py, 0.0f), kOneMinusEps);
}

// ---- pixel samplers: colour/src/samplers/{pixel_sampler,zero_two,stratified}.rs -----------------
// One thread makes all samples of one pixel, as start_pixel does, from a generator seeded per
// pixel (sequence 2^63 | (seed * 2^32 + pixel)): four sampled dimensions, of which the camera
// uses 1D dimension 0 (time) and 2D dimensions 0 (film) and 1 (lens); the others only advance
// the generator.
constexpr int kPixelSamplerDims = 4;
constexpr uint32_t kPixelSamplesLdsSpp = 256; // 128 KiB of LDS

RT_DEV uint32_t rng_limit(Rng &rng, uint32_t limit) { // uniform_u32_limit, rng.rs:51-59
    for (;;) {
        const uint32_t r = rng_u32(rng);
        // threshold = 2^32 mod limit is below limit: only a draw below limit needs it
        if (r >= limit || r >= (~limit + 1u) % limit) return r % limit;
    }
}
// Where a pixel's samples live while they are made: nowhere (the generator is only advanced),
// the lane's own stretch of global memory, or the wave's LDS block (element i of lane l at
// i * 64 + (l ^ (i & 63)): conflict-free when the lanes share i, and for the transposed
// read-out in which 64 lanes read 64 consecutive i of one lane).
struct NoArr {
    RT_DEV float get(uint32_t) const { return 0.0f; }
    RT_DEV void set(uint32_t, float) const {}
};
struct MemArr {
    float *p;
    RT_DEV float get(uint32_t i) const { return p[i]; }
    RT_DEV void set(uint32_t i, float v) const { p[i] = v; }
};
struct LdsArr {
    float *p;
    uint32_t lane;
    RT_DEV float get(uint32_t i) const { return p[i * 64u + (lane ^ (i & 63u))]; }
    RT_DEV void set(uint32_t i, float v) const { p[i * 64u + (lane ^ (i & 63u))] = v; }
};

template <class A>
RT_DEV void shuffle1(Rng &rng, const A &a, uint32_t n) { // shuffle_dims(.., 1), rng.rs:77-84
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t other = i + rng_limit(rng, n - i);
        const float t = a.get(i), o = a.get(other);
        a.set(i, o);
        a.set(other, t);
    }
}
template <class A>
RT_DEV void shuffle2(Rng &rng, const A &a, const A &b, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t other = i + rng_limit(rng, n - i);
        const float ta = a.get(i), tb = b.get(i), oa = a.get(other), ob = b.get(other);
        a.set(i, oa);
        b.set(i, ob);
        a.set(other, ta);
        b.set(other, tb);
    }
}
// van_der_corput(1, n, ..), zero_two.rs:68-92
template <class A>
RT_DEV void zero_two_1d(Rng &rng, const A &out, uint32_t n) {
    uint32_t v = rng_u32(rng);
    for (uint32_t i = 0; i < n; i++) { // sample_gray_code, zero_two.rs:129-136
        out.set(i, unit_float(v));
        v ^= 0x80000000u >> __builtin_ctz(i + 1);
    }
    for (uint32_t i = 0; i < n; i++) (void)rng_limit(rng, 1u); // one-element shuffles, zero_two.rs:86-89
    shuffle1(rng, out, n);
}
// column c of the second (0,2) generator matrix (zero_two.rs:106-112): row c of Pascal's
// triangle mod 2, most significant bit first
RT_DEV uint32_t sobol2_column(uint32_t c) {
    uint32_t v = 0;
    for (uint32_t k = 0; k <= c; k++)
        if ((c & k) == k) v |= 0x80000000u >> k;
    return v;
}
// sobol_2d(1, n, ..), zero_two.rs:95-126
template <class A>
RT_DEV void zero_two_2d(Rng &rng, const A &x, const A &y, uint32_t n) {
    uint32_t vx = rng_u32(rng), vy = rng_u32(rng);
    for (uint32_t i = 0; i < n; i++) { // sample_gray_code2, zero_two.rs:140-155
        x.set(i, unit_float(vx));
        y.set(i, unit_float(vy));
        const uint32_t c = (uint32_t)__builtin_ctz(i + 1);
        vx ^= 0x80000000u >> c;
        vy ^= sobol2_column(c);
    }
    for (uint32_t i = 0; i < n; i++) (void)rng_limit(rng, 1u);
    shuffle2(rng, x, y, n);
}
// stratified_sample_one + shuffle, stratified.rs:35-38,81-88
template <class A>
RT_DEV void stratified_1d(Rng &rng, const A &out, uint32_t n, bool jitter) {
    const float inv = 1.0f / (float)(int32_t)n;
    for (uint32_t i = 0; i < n; i++) {
        const float delta = jitter ? rng_f(rng) : 0.5f;
        out.set(i, sel_min(((float)(int32_t)i + delta) * inv, kOneMinusEps));
    }
    shuffle1(rng, out, n);
}
// stratified_sample_two + shuffle, stratified.rs:40-49,91-116, strata = cells of the nx x ny grid
template <class A>
RT_DEV void stratified_2d(Rng &rng, const A &x, const A &y, uint32_t nx, uint32_t ny, bool jitter) {
    const float dx = 1.0f / (float)(int32_t)nx, dy = 1.0f / (float)(int32_t)ny;
    for (uint32_t iy = 0, i = 0; iy < ny; iy++)
        for (uint32_t ix = 0; ix < nx; ix++, i++) {
            const float jx = jitter ? rng_f(rng) : 0.5f;
            const float jy = jitter ? rng_f(rng) : 0.5f;
            x.set(i, sel_min(((float)(int32_t)ix + jx) * dx, kOneMinusEps));
            y.set(i, sel_min(((float)(int32_t)iy + jy) * dy, kOneMinusEps));
        }
    shuffle2(rng, x, y, nx * ny);
}

// the 1D dimensions (the first is the camera's time, the others only advance the generator)
template <class A>
RT_DEV void pixel_samples_1d(const FrameDev &F, Rng &rng, const A &time) {
    const uint32_t n = F.spp;
    const bool jitter = !(F.sampler_arg & RT_STRATIFIED_NO_JITTER);
    if (F.sampler == RT_SAMPLER_ZERO_TWO) {
        zero_two_1d(rng, time, n);
        for (int d = 1; d < kPixelSamplerDims; d++) zero_two_1d(rng, NoArr{}, n);
    } else {
        stratified_1d(rng, time, n, jitter);
        for (int d = 1; d < kPixelSamplerDims; d++) stratified_1d(rng, NoArr{}, n, jitter);
    }
}
// one 2D dimension (film, then lens)
template <class A>
RT_DEV void pixel_samples_2d(const FrameDev &F, Rng &rng, const A &x, const A &y) {
    const uint32_t n = F.spp;
    if (F.sampler == RT_SAMPLER_ZERO_TWO) {
        zero_two_2d(rng, x, y, n);
    } else {
        const uint32_t nx = F.sampler_arg & 0xffffu;
        stratified_2d(rng, x, y, nx, n / nx, !(F.sampler_arg & RT_STRATIFIED_NO_JITTER));
    }
}

RT_DEV bool rank_to_pixel(const FrameDev &F, uint32_t rank, uint32_t &px, uint32_t &py);

RT_DEV bool slot_pixel(const FrameDev &F, uint32_t slot, uint32_t n_slots, int by_rank, uint64_t first_pixel,
                       uint64_t &pixel) {
    if (slot >= n_slots) return false;
    if (by_rank) {
        uint32_t px, py;
        if (!rank_to_pixel(F, slot, px, py)) return false;
        pixel = (uint64_t)py * F.width + px;
        return true;
    }
    pixel = first_pixel + slot;
    return pixel < (uint64_t)F.width * F.height;
}

// Any spp: every lane works in place in its own stretch of the sample arrays.
__global__ __launch_bounds__(kBlock) void k_pixel_samples(FrameDev F, uint32_t n_slots, int by_rank,
                                                          uint64_t first_pixel, float *out) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    uint64_t pixel;
    if (!slot_pixel(F, slot, n_slots, by_rank, first_pixel, pixel)) return;
    Rng rng;
    rng_seq(rng, (1ull << 63) | ((F.seed << 32) + pixel));
    float *base = out + (size_t)slot * F.spp;
    const size_t st = F.sample_stride;
    pixel_samples_1d(F, rng, MemArr{base + 2 * st});
    pixel_samples_2d(F, rng, MemArr{base}, MemArr{base + st});
    pixel_samples_2d(F, rng, MemArr{base + 3 * st}, MemArr{base + 4 * st});
}

// spp <= kPixelSamplesLdsSpp: one wave per workgroup works in LDS (2 * spp * 64 floats) and
// copies each finished dimension out with the lanes running along a pixel's samples, so the
// stores are contiguous.
__global__ __launch_bounds__(64) void k_pixel_samples_lds(FrameDev F, uint32_t n_slots, int by_rank,
                                                          uint64_t first_pixel, float *out) {
    extern __shared__ float lds_samples[];
    const uint32_t lane = threadIdx.x, slot0 = blockIdx.x * 64u, n = F.spp;
    uint64_t pixel = 0;
    const bool valid = slot_pixel(F, slot0 + lane, n_slots, by_rank, first_pixel, pixel);
    const uint64_t valid_mask = __ballot(valid);
    Rng rng;
    rng_seq(rng, (1ull << 63) | ((F.seed << 32) + pixel));
    const LdsArr a{lds_samples, lane}, b{lds_samples + (size_t)n * 64u, lane};
    // element e of the wave's 64 * n floats of one array belongs to pixel e / n, sample e % n
    auto copy_out = [&](const float *lds, uint32_t k) {
        __syncthreads();
        float *g = out + (size_t)k * F.sample_stride + (size_t)slot0 * n;
        uint32_t p = lane / n, i = lane - p * n;
        for (uint32_t e = lane; e < 64u * n; e += 64u) {
            if ((valid_mask >> p) & 1ull) g[e] = lds[i * 64u + (p ^ (i & 63u))];
            i += 64u;
            while (i >= n) {
                i -= n;
                p++;
            }
        }
        __syncthreads();
    };
    if (valid) pixel_samples_1d(F, rng, a);
    copy_out(a.p, 2);
    if (valid) pixel_samples_2d(F, rng, a, b);
    copy_out(a.p, 0);
    copy_out(b.p, 1);
    if (valid) pixel_samples_2d(F, rng, a, b);
    copy_out(a.p, 3);
    copy_out(b.p, 4);
}

// Camera sample of path (pixel, s): film offsets u[0], u[1], time u[2], lens u[3], u[4].
// slot: the pixel's place in F.pixel_samples (pixel samplers only).
RT_DEV void camera_sample(const FrameDev &F, uint32_t slot, uint32_t px, uint32_t py, uint32_t s, Rng &rng,
                          float u[5]) {
    if (F.sampler == RT_SAMPLER_HALTON) {
        halton_sample(F.halton, px, py, s, u);
    } else if (F.sampler == RT_SAMPLER_SOBOL) {
        sobol_sample(F, px, py, s, u);
    } else if (F.sampler == RT_SAMPLER_ZERO_TWO || F.sampler == RT_SAMPLER_STRATIFIED) {
        const float *p = F.pixel_samples + (size_t)slot * F.spp + s;
#pragma unroll
        for (int k = 0; k < 5; k++) u[k] = p[(size_t)k * F.sample_stride];
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) u[k] = rng_f(rng);
    }
}

// Path of pixel (px, py), global id `path` = pixel * spp + sample; PCG32 sequence =
// seed * 2^32 + path; camera sample order: film x, film y, time, lens u, lens v.
RT_DEV Ray path_setup(const FrameDev &F, const rt_camera &cam, uint32_t slot, uint32_t px, uint32_t py,
                      uint64_t path, Rng &rng) {
    rng_seq(rng, (F.seed << 32) + path);
    float u[5];
    camera_sample(F, slot, px, py, (uint32_t)(path % F.spp), rng, u);
    return camera_ray(cam, (float)px + u[0], (float)py + u[1], u[3], u[4], u[2]);
}

__global__ __launch_bounds__(kBlock) void k_generate_rays(FrameDev F, rt_camera cam, uint64_t first, uint32_t n,
                                                          rt_ray *rays) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    uint64_t path = first + i, pixel = path / F.spp;
    Ray r = path_setup(F, cam, (uint32_t)(pixel - first / F.spp), (uint32_t)(pixel % F.width),
                       (uint32_t)(pixel / F.width), path, rng);
    float4 *o = reinterpret_cast<float4 *>(rays + i);
    o[0] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
    o[1] = make_float4(r.d.y, r.d.z, r.t_max, r.time);
}

// ---- bounce queues ------------------------------------------------------------------------------
RT_DEV void queue_push(const SceneDev &S, const Queue &q, size_t slot, const Ray &r, uint32_t local, v3 beta,
                       const Rng &rng, uint32_t gpath) {
    if (S.n_enclosing | S.n_balls) {
        float t_max;
        uint32_t code;
        Ray fresh = r;
        fresh.t_max = kInf; // every queued ray starts unbounded
        probe::double_pretest(S, fresh);
        pretest(S, fresh, t_max, code);
        q.e[slot] = make_uint2(__float_as_uint(t_max), code);
    }
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
__global__ __launch_bounds__(kBlock) void k_generate(SceneDev S, FrameDev F, rt_camera cam, Queue q, uint32_t *counts,
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
            ray = path_setup(F, cam, rank, px, py, gpath, rng);
            alive = true;
        }
    }
    if (alive) {
        queue_push(S, q, slot, ray, local, mk(1.0f, 1.0f, 1.0f), rng, gpath);
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
    // A noise lookup is a chain of dependent gathers (permutation entries, then gradient vectors):
    // corner by corner it was sixteen round trips, and a turbulence sum makes seven lookups.  The
    // six permutation entries come in one round trip and the gradient vectors of two corners at a
    // time in another (four at a time gained nothing and cost other scenes registers).  The corners
    // are still added in the order (di, dj, dk) = (0,0,0), (0,0,1), (0,1,0), ...; the loop over the
    // pairs stays rolled so that noise, which is rare, does not set the kernel's register budget.
    const int pxi[2] = {px[i & 255], px[(i + 1) & 255]};
    const int pyj[2] = {py[j & 255], py[(j + 1) & 255]};
    const int pzk[2] = {pz[k & 255], pz[(k + 1) & 255]};
    constexpr int kAtOnce = 2;
#pragma unroll 1
    for (int first = 0; first < 8; first += kAtOnce) {
        float c[kAtOnce][3];
#pragma unroll
        for (int q = 0; q < kAtOnce; q++) {
            const int corner = first + q, di = corner >> 2, dj = (corner >> 1) & 1, dk = corner & 1;
            const float *g = S.perlin_vec + 3 * ((di ? pxi[1] : pxi[0]) ^ (dj ? pyj[1] : pyj[0]) ^ (dk ? pzk[1] : pzk[0]));
            c[q][0] = g[0];
            c[q][1] = g[1];
            c[q][2] = g[2];
        }
#pragma unroll
        for (int q = 0; q < kAtOnce; q++) {
            const int corner = first + q, di = corner >> 2, dj = (corner >> 1) & 1, dk = corner & 1;
            const float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
            const float dt = c[q][0] * wx + c[q][1] * wy + c[q][2] * wz;
            const float fi = di ? uu : 1.0f - uu;
            const float fj = dj ? vv : 1.0f - vv;
            const float fk = dk ? ww : 1.0f - ww;
            accum += fi * fj * fk * dt;
        }
    }
    return accum;
}
RT_DEV float perlin_turb(const SceneDev &S, v3 p) {
    float accum = 0.0f, weight = 1.0f;
#pragma unroll 1
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

// Shading of one finished path segment: emit, or scatter into the next ray
// (L.ray.o / L.ray.d) with the throughput updated.  Returns whether the path
// continues.  PLAIN: the hit does not carry kBestRare (or is a miss), so only the
// four common surface kinds and media, the four materials and the two texture forms that
// flag leaves out can occur; the general code is the same computation with every
// case present.
// Direct lighting (NEE): what a hit on a Lambertian surface hands to the any-hit traversal.
struct Shadow {
    bool want;    // a shadow ray was made
    bool sampled; // this vertex sampled the lights: the next one must not add a light's emission
    v3 o, d, contrib;
};
constexpr float kShadowTMax = 1.0f - 0.0001f; // Interaction::spawn_ray_to, interaction.rs:113

// The end of a path stores its radiance; with direct lighting the shadow rays of earlier
// vertices have already added theirs.
template <bool NEE>
RT_DEV void add_radiance(float4 *radiance, uint32_t local, v3 v) {
    if (NEE) {
        float4 r = radiance[local];
        radiance[local] = make_float4(r.x + v.x, r.y + v.y, r.z + v.z, 0.0f);
    } else {
        radiance[local] = make_float4(v.x, v.y, v.z, 0.0f);
    }
}

template <bool PLAIN, bool NEE = false>
RT_DEV bool shade(const SceneDev &S, Lane &L, v3 &beta, Rng &rng, uint32_t local, float4 *radiance,
                  bool skip_emission = false, Shadow *sh = nullptr) {
    if (L.best == RT_MISS) {
        add_radiance<NEE>(radiance, local, mk(beta.x * S.background[0], beta.y * S.background[1], beta.z * S.background[2]));
        return false;
    }
    bool alive = true;
    LeafPrim lp = S.leaf[L.best & ~kBestRare];
    // Small scenes keep the plain shader's material record once per leaf entry, so that it is fetched beside
    // the entry instead of after it (one dependent round trip less); large ones once per material,
    // where a per-entry table would be a gather over hundreds of megabytes.
    PlainShade ps;
    if (PLAIN && S.plain_per_entry) ps = S.plain[L.best & ~kBestRare];
    Surf h;
    if (PLAIN) prim_finalize_plain(S, lp, L.ray, L.ray.t_max, L.a0, L.a1, L.a2, h);
    else prim_finalize(S, lp, L.ray, L.ray.t_max, L.a0, L.a1, L.a2, h);
    const v3 d = L.ray.d;
    const bool front = dot(d, h.n) < 0.0f;
    const v3 nf = front ? h.n : -h.n;
    // The plain shader reads material and texture from one record per material (a load that
    // runs beside the geometry's); the general one follows the references.
    rt_material m;
    if (PLAIN) {
        if (!S.plain_per_entry) ps = S.plain[lp.material];
        m.kind = ps.material_kind;
        m.texture = 0;
        m.fuzz = ps.fuzz;
        m.ior = ps.ior;
    } else {
        m = S.materials[lp.material];
    }
    // The pieces every material shares run once for the whole wave: the
    // texture lookup (all but dielectrics) and the unit-sphere sample
    // (Lambertian, metallic, isotropic).  Per path the RNG draws stay in
    // the same order as when each material is evaluated on its own.
    v3 tex = mk(1.0f, 1.0f, 1.0f), rs = mk(0.0f, 0.0f, 0.0f);
    if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METALLIC || m.kind == RT_MAT_ISOTROPIC) probe::double_unit_sphere(rng);
    if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METALLIC || m.kind == RT_MAT_ISOTROPIC)
        rs = random_in_unit_sphere(rng);
    if (m.kind != RT_MAT_DIELECTRIC) {
        if (PLAIN) {
            tex = mk(ps.a[0], ps.a[1], ps.a[2]);
            if (ps.checker) { // a solid colour, or a 3-D checker whose two cells are solid colours
                const float sines = p_sin(ps.scale * h.p.x) * p_sin(ps.scale * h.p.y) * p_sin(ps.scale * h.p.z);
                if (!(sines < 0.0f)) tex = mk(ps.b[0], ps.b[1], ps.b[2]);
            }
        } else {
            tex = texture_value(S, m.texture, h.u, h.v, h.p);
        }
    }
    v3 dir = nf;
    if (NEE && m.kind == RT_MAT_LAMBERTIAN && S.n_lights) {
        // one point on one light; the draws follow the scatter direction's
        uint32_t li = 0;
        if (S.n_lights > 1) {
            li = (uint32_t)(rng_f(rng) * (float)S.n_lights);
            if (li >= S.n_lights) li = S.n_lights - 1;
        }
        const float u1 = rng_f(rng), u2 = rng_f(rng);
        const LightDev lt = S.lights[li];
        const bool distant = lt.axis == 3u; // the light's direction is the same from everywhere
        const v3 w = mk(lt.a0, lt.a1, lt.b0);
        const float qa = lt.a0 + u1 * (lt.a1 - lt.a0), qb = lt.b0 + u2 * (lt.b1 - lt.b0);
        const v3 q = distant ? h.p + w * lt.k
                             : (lt.axis == 0 ? mk(lt.k, qa, qb) : (lt.axis == 1 ? mk(qa, lt.k, qb) : mk(qa, qb, lt.k)));
        // Interaction::spawn_ray_to, shapes/src/interaction.rs:102-116
        const v3 so = offset_origin(h.p, h.perr, h.n, q - h.p);
        const v3 sd = q - so;
        const float d2 = dot(sd, sd), cs = distant ? dot(w, nf) : dot(sd, nf);
        const float cl = distant ? 1.0f : fabsf(comp(sd, (int)(lt.axis % 3u)));
        if (cs > 0.0f && cl > 0.0f && d2 > 0.0f) {
            // rectangle: cos cos / distance^2 x area x n_lights / pi; distant light: cos x n_lights / pi
            const float g = distant ? cs * lt.scale : (cs * cl) / (d2 * d2) * lt.scale;
            sh->want = true;
            sh->o = so;
            sh->d = sd;
            sh->contrib = mk(beta.x * tex.x * lt.rgb[0] * g, beta.y * tex.y * lt.rgb[1] * g, beta.z * tex.z * lt.rgb[2] * g);
        }
        sh->sampled = true;
    }
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) {
        // a light the shadow ray of the vertex before has already accounted for adds nothing
        if (!(NEE && skip_emission && (lp.kind & kSampledLight))) add_radiance<NEE>(radiance, local, mk(beta.x * tex.x, beta.y * tex.y, beta.z * tex.z));
        alive = false;
    } else if (m.kind == RT_MAT_LAMBERTIAN) {
        dir = nf + rs;
        if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = nf;
    } else if (m.kind == RT_MAT_METALLIC) {
        dir = reflect(d, nf) + rs * m.fuzz;
        if (!(dot(dir, nf) > 0.0f)) alive = false; // absorbed
    } else if (m.kind == RT_MAT_ISOTROPIC) {
        dir = rs;
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
    }
    if (alive) {
        dir = normalize(dir);
        beta = mk(beta.x * tex.x, beta.y * tex.y, beta.z * tex.z);
        L.ray.o = offset_origin(h.p, h.perr, h.n, dir);
        L.ray.d = dir;
    }
    return alive;
}

// One bounce: queued path segments in, shaded survivors compacted into the next
// queue.  A wave reads and writes one sub-queue (its index modulo kShards).
//
// Shading is the most divergent and register-hungry code of the kernel, so it is
// not run whenever a few rays finish.  A finished ray leaves a 24-byte record
// (queue entry, winning leaf slot, t, three primitive-specific values) in one of
// the wave's two parking lists and its lane is refilled at once.  List 0 takes
// misses and hits the plain shader covers, list 1 the hits that need the general
// shader (textures with lookups or noise, dielectrics, transformed or moving
// primitives).  When a list holds 64 records the whole wave shades them
// together: every lane of every shading instruction works, the short plain shader
// runs for most rays, and the long general one runs once per 64 rays that need it
// rather than whenever one ray of a wave does.  The lanes' traversal state is
// written to memory around a shading batch and read back after it, so shading and
// traversal do not compete for registers.
template <bool NEE>
struct BounceFeed {
    static constexpr bool kPretested = true;
    const SceneDev &S;
    const Queue &qin, &qout;
    uint32_t *tail; // tail counter of the sub-queue written
    size_t in_base, out_base;
    Blocks blocks;
    bool last; // depth exhausted after this bounce: survivors contribute nothing
    uint64_t seed;
    float4 *radiance;
    float4 *park_a;     // 2 x kParkSlots x {entry, leaf slot, t, a0}
    float2 *park_b;     // 2 x kParkSlots x {a1, a2}
    uint32_t *save;     // kSaveWords x 64, word k of lane l at save[k * 64 + l]
    uint32_t parked[2]; // wave uniform
    uint32_t entry;     // queue entry of the ray this lane is tracing
    // direct lighting: the shadow rays this bounce makes (sub-queue and tail as for qout)
    ShadowQueue sq;
    uint32_t *stail;
    size_t shadow_base; // first entry of the shadow sub-queue written (its own capacity, not qout's)
    uint32_t code = 0; // what the producer of the ray a lane just received found out (pretest)
    PROF(unsigned long long prof_shade = 0;) // cycles in shading batches incl. state save and reload
    PROF(unsigned long long prof_take = 0; unsigned long long prof_load = 0; unsigned long long prof_park = 0;)
    PROF(unsigned long long prof_b[16] = {};)
    RT_DEV bool more() const { return blocks.more(); }

    RT_DEV void compact(bool alive, const Lane &P, uint32_t local, v3 beta, const Rng &rng, uint32_t gpath, int prof_at = 0) {
        // stream compaction: ballot, rank by prefix count, one atomic per wave on the tail
        const uint64_t mask = __ballot(alive);
        const uint32_t total = (uint32_t)__popcll(mask);
        if (total) {
            PROF(const unsigned long long c0 = __builtin_amdgcn_s_memtime();)
            uint32_t base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(tail, total);
            base = __builtin_amdgcn_readfirstlane(base);
            PROF(const unsigned long long c1 = __builtin_amdgcn_s_memtime(); prof_b[prof_at + 4] += c1 - c0;)
            if (alive) queue_push(S, qout, out_base + base + lane_rank(mask), P.ray, local, beta, rng, gpath);
            PROF(prof_b[prof_at + 5] += __builtin_amdgcn_s_memtime() - c1;)
        }
    }
    RT_DEV void compact_shadow(const Shadow &sh, float time, uint32_t local) {
        const uint64_t mask = __ballot(sh.want);
        const uint32_t total = (uint32_t)__popcll(mask);
        if (total) {
            uint32_t base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(stail, total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (sh.want) {
                const size_t slot = shadow_base + base + lane_rank(mask);
                sq.a[slot] = make_float4(sh.o.x, sh.o.y, sh.o.z, time);
                sq.b[slot] = make_float4(sh.d.x, sh.d.y, sh.d.z, __uint_as_float(local));
                sq.c[slot] = make_float4(sh.contrib.x, sh.contrib.y, sh.contrib.z, 0.0f);
            }
        }
    }

    template <bool PLAIN>
    RT_DEV void shade_batch(uint32_t first, uint32_t count) {
        const uint32_t lane = threadIdx.x & 63;
        bool alive = false;
        Lane P;
        v3 beta = mk(0.0f, 0.0f, 0.0f);
        Rng rng{0, 0};
        uint32_t gpath = 0, local = kDead;
        Shadow sh;
        sh.want = sh.sampled = false;
        sh.o = sh.d = sh.contrib = mk(0.0f, 0.0f, 0.0f);
        PROF(const int pa = PLAIN ? 0 : 8; prof_b[pa]++; const unsigned long long s0 = __builtin_amdgcn_s_memtime();)
        if (lane < count) {
            const float4 ra = park_a[first + lane];
            const float2 rb = park_b[first + lane];
            PROF(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long s1 = __builtin_amdgcn_s_memtime(); prof_b[pa + 1] += s1 - s0;)
            const size_t e = in_base + __float_as_uint(ra.x);
            const float4 qa = qin.a[e], qb = qin.b[e], qc = qin.c[e];
            const uint2 qd = qin.d[e];
            PROF(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long s2 = __builtin_amdgcn_s_memtime(); prof_b[pa + 2] += s2 - s1;)
            P.ray.o = mk(qa.x, qa.y, qa.z);
            P.ray.time = qa.w;
            P.ray.d = mk(qb.x, qb.y, qb.z);
            P.ray.t_max = ra.z;
            P.best = __float_as_uint(ra.y);
            P.a0 = ra.w;
            P.a1 = rb.x;
            P.a2 = rb.y;
            P.has_ray = true;
            local = __float_as_uint(qb.w);
            beta = mk(qc.x, qc.y, qc.z);
            gpath = qd.y;
            rng.state = ((uint64_t)qd.x << 32) | __float_as_uint(qc.w);
            rng.inc = (((seed << 32) + gpath) << 1) | 1u;
#ifdef RT_DOUBLE_SHADE
            // cost probe (rt_probe.hpp); frames without direct lighting only: with it `local` still carries
            // a flag in bit 31 here and radiance is accumulated, so a second run would write out of bounds
            if constexpr (!NEE) {
                Lane P2 = P;
                P2.ray.o.x = opaque(P2.ray.o.x);
                v3 beta2 = beta;
                Rng rng2 = rng;
                // radiance of an unused slot far away is not touched: hits on lights and misses
                // write to `local`, so give the copy the same slot and let the real run overwrite it
                bool g = shade<PLAIN>(S, P2, beta2, rng2, local, radiance);
                float sink = g ? P2.ray.o.x + P2.ray.d.x + beta2.x : 0.0f;
                asm volatile("" ::"v"(sink));
            }
#endif
            if (NEE) {
                // bit 31 of the path index: the vertex before this one sampled the lights
                const bool skip_emission = (local >> 31) != 0;
                local &= 0x7fffffffu;
                alive = shade<PLAIN, true>(S, P, beta, rng, local, radiance, skip_emission, &sh) && !last;
            } else {
                alive = shade<PLAIN>(S, P, beta, rng, local, radiance) && !last;
            }
            PROF(prof_b[pa + 3] += __builtin_amdgcn_s_memtime() - s2;)
        }
        if (NEE) compact_shadow(sh, P.ray.time, local);
        compact(alive, P, NEE && sh.sampled ? (local | 0x80000000u) : local, beta, rng, gpath PROF(, pa));
    }

    RT_DEV bool turn(Trav &T, bool free) {
        Lane &L = T.L;
        const uint32_t lane = threadIdx.x & 63;
        // park the finished rays
        PROF(const unsigned long long prof_tp = __builtin_amdgcn_s_memtime();)
        const bool finished = free && L.has_ray;
        const bool general = L.best != RT_MISS && (L.best & kBestRare) != 0;
        const uint64_t m0 = __ballot(finished && !general), m1 = __ballot(finished && general);
        if (finished) {
            const uint32_t slot = general ? kParkSlots + parked[1] + lane_rank(m1) : parked[0] + lane_rank(m0);
            park_a[slot] = make_float4(__uint_as_float(entry), __uint_as_float(L.best), L.ray.t_max, L.a0);
            park_b[slot] = make_float2(L.a1, L.a2);
        }
        parked[0] += (uint32_t)__popcll(m0);
        parked[1] += (uint32_t)__popcll(m1);
        PROF(const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(); prof_park += prof_t0 - prof_tp;)
        if (parked[0] >= 64 || parked[1] >= 64) {
            // The traversing lanes' state leaves the registers while a batch is shaded.  When no
            // lane is traversing (small scenes: the rays of a wave finish together) there is
            // nothing to keep: free lanes are reloaded or cleared below.
            const bool any_busy = __ballot(T.cur != kNone) != 0;
            uint32_t *sv = save + lane;
            PROF(const unsigned long long v0 = __builtin_amdgcn_s_memtime();)
            if (any_busy) {
                sv[0 * 64] = __float_as_uint(L.ray.o.x); sv[1 * 64] = __float_as_uint(L.ray.o.y);
                sv[2 * 64] = __float_as_uint(L.ray.o.z); sv[3 * 64] = __float_as_uint(L.ray.d.x);
                sv[4 * 64] = __float_as_uint(L.ray.d.y); sv[5 * 64] = __float_as_uint(L.ray.d.z);
                sv[6 * 64] = __float_as_uint(L.ray.t_max); sv[7 * 64] = __float_as_uint(L.ray.time);
                sv[8 * 64] = L.best; sv[9 * 64] = __float_as_uint(L.a0);
                sv[10 * 64] = __float_as_uint(L.a1); sv[11 * 64] = __float_as_uint(L.a2);
                sv[12 * 64] = __float_as_uint(T.inv.x); sv[13 * 64] = __float_as_uint(T.inv.y);
                sv[14 * 64] = __float_as_uint(T.inv.z); sv[15 * 64] = T.cur;
                sv[16 * 64] = T.sp; sv[17 * 64] = entry;
            }
            // the compiler must really store and reload (no forwarding through registers),
            // and records parked by other lanes are read below
            asm volatile("" ::: "memory");
            __threadfence_block();
            PROF(prof_b[6] += __builtin_amdgcn_s_memtime() - v0; prof_b[7]++;)
            if (parked[0] >= 64) {
                parked[0] -= 64;
                shade_batch<true>(parked[0], 64);
            }
            if (parked[1] >= 64) {
                parked[1] -= 64;
                shade_batch<false>(kParkSlots + parked[1], 64);
            }
            asm volatile("" ::: "memory");
            PROF(const unsigned long long v1 = __builtin_amdgcn_s_memtime();)
            if (any_busy) {
                L.ray.o = mk(__uint_as_float(sv[0 * 64]), __uint_as_float(sv[1 * 64]), __uint_as_float(sv[2 * 64]));
                L.ray.d = mk(__uint_as_float(sv[3 * 64]), __uint_as_float(sv[4 * 64]), __uint_as_float(sv[5 * 64]));
                L.ray.t_max = __uint_as_float(sv[6 * 64]);
                L.ray.time = __uint_as_float(sv[7 * 64]);
                L.best = sv[8 * 64];
                L.a0 = __uint_as_float(sv[9 * 64]);
                L.a1 = __uint_as_float(sv[10 * 64]);
                L.a2 = __uint_as_float(sv[11 * 64]);
                T.inv = mk(__uint_as_float(sv[12 * 64]), __uint_as_float(sv[13 * 64]), __uint_as_float(sv[14 * 64]));
                T.cur = sv[15 * 64];
                T.sp = sv[16 * 64];
                entry = sv[17 * 64];
                PROF(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); prof_b[14] += __builtin_amdgcn_s_memtime() - v1;)
            } else {
                L.ray.o = L.ray.d = T.inv = mk(0.0f, 0.0f, 0.0f);
                L.ray.t_max = L.ray.time = L.a0 = L.a1 = L.a2 = 0.0f;
                L.best = RT_MISS;
                T.cur = kNone;
                T.sp = 0;
                entry = 0;
            }
        }
        PROF(const unsigned long long prof_t1 = __builtin_amdgcn_s_memtime(); prof_shade += prof_t1 - prof_t0;)
        // next rays
        const uint32_t next = blocks.take(free);
        PROF(const unsigned long long prof_t2 = __builtin_amdgcn_s_memtime(); prof_take += prof_t2 - prof_t1;)
        bool fresh = false;
        if (free) {
            L.has_ray = false;
            if (next != kNone) {
                const float4 qa = qin.a[in_base + next], qb = qin.b[in_base + next];
                uint2 qe = make_uint2(__float_as_uint(kInf), 0u);
                if (S.n_enclosing | S.n_balls) qe = qin.e[in_base + next];
                PROF(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");)
                if (__float_as_uint(qb.w) != kDead) {
                    L.ray.o = mk(qa.x, qa.y, qa.z);
                    L.ray.time = qa.w;
                    L.ray.d = mk(qb.x, qb.y, qb.z);
                    // what the ray's producer found in the media that enclose the scene and, for a
                    // ray that starts inside a ball, in the ball
                    L.ray.t_max = __uint_as_float(qe.x);
                    bool resolved;
                    L.best = pretest_best(S, qe.y, resolved);
                    code = qe.y;
                    L.a0 = L.a1 = L.a2 = 0.0f;
                    entry = next;
                    L.has_ray = true;
                    fresh = true;
                }
            }
        }
        PROF(prof_load += __builtin_amdgcn_s_memtime() - prof_t2;)
        return fresh;
    }

    RT_DEV bool drain() { return false; }
    // after the last ray of the stream: shade what is still parked
    RT_DEV void finish() {
        __threadfence_block();
        while (parked[0] != 0) {
            const uint32_t count = parked[0] < 64u ? parked[0] : 64u;
            parked[0] -= count;
            shade_batch<true>(parked[0], count);
        }
        while (parked[1] != 0) {
            const uint32_t count = parked[1] < 64u ? parked[1] : 64u;
            parked[1] -= count;
            shade_batch<false>(kParkSlots + parked[1], count);
        }
    }
};

// One bounce of every queued path: closest hit, shading, scatter, and the
// survivors compacted into the next queue.  Radiance is stored once, when a
// path ends on a light or leaves the scene.  The launch must have a wave count
// that is a multiple of kShards, so that wave w reads and writes sub-queue
// shard_of_wave(w) only.
template <bool COUNT, bool LDSN, bool NEE>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_bounce(
    SceneDev S, Queue qin, Queue qout, uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed,
    float4 *radiance, uint8_t *scratch, unsigned long long *counters, ShadowQueue sq, uint32_t *scounts) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t *cin = counts + (size_t)bounce * kShards * kCountStride;
    uint32_t *cout = counts + (size_t)(bounce + 1) * kShards * kCountStride;
    // lane s holds the size of sub-queue s; the longest one tells which workgroups have work
    const uint32_t my_count = lane < kShards ? cin[lane * kCountStride] : 0u;
    uint32_t longest = my_count;
    for (int s = 32; s > 0; s >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, s, 64));
    const uint32_t n_chunks = kShards * ((longest + 63) / 64);
    if (blockIdx.x * (blockDim.x >> 6) >= n_chunks) return; // nothing for this workgroup (uniform)
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    const uint32_t w = wave_index(), shard = shard_of_wave(w), per_shard = wave_total() / kShards;
    const uint32_t n = __builtin_amdgcn_readlane(my_count, shard);
    uint8_t *mine = scratch + (size_t)w * kWaveScratchBytes;
    BounceFeed<NEE> feed{S, qin, qout, cout + shard * kCountStride, (size_t)shard * qin.cap, (size_t)shard * qout.cap,
                    // blocks after a wave's first are claimed from the word next to the sub-queue's
                    // size (zero at launch: counts are cleared once per batch)
                    Blocks{(w / kShards) * 64, 0u, n, per_shard * 64,
                           counts + ((size_t)bounce * kShards + shard) * kCountStride + 1, per_shard, 1u, 0u},
                    bounce == last, seed, radiance,
                    reinterpret_cast<float4 *>(mine), reinterpret_cast<float2 *>(mine + 2 * kParkSlots * 16),
                    reinterpret_cast<uint32_t *>(mine + 2 * kParkSlots * 24), {0u, 0u}, 0u, sq,
                    NEE ? scounts + ((size_t)bounce * kShards + shard) * kCountStride : nullptr,
                    (size_t)shard * sq.cap};
    // camera rays of neighbouring pixels take about equally long, so a wave's 64 finish almost
    // together and refilling in between only adds turns; scattered rays do not
    trace_stream<false, COUNT, LDSN, false>(S, tv, feed, cnt, bounce == 0 ? primary_refill_lanes_of<LDSN>() : (LDSN ? kRefillLanes : (uint32_t)RT_BOUNCE_REFILL_LANES_GLOBAL));
    PROF(cnt.p[7] = feed.prof_shade; cnt.p[13] = feed.prof_take; cnt.p[14] = feed.prof_load; cnt.p[15] = feed.prof_park;
         for (int k = 0; k < 16; k++) cnt.p[16 + k] = feed.prof_b[k];)
    if (COUNT) flush_counters(cnt, counters);
}

// ---- one bounce as two kernels ------------------------------------------------------------------------
// The fused kernel above shades inside the traversal's register budget and parks, saves and
// reloads around every batch.  That pays while the tree sits in LDS (one workgroup per CU either
// way).  A tree in global memory wants as many waves as it can get to hide its fetches, and then
// the shader's registers are spilled ones: there the bounce is two launches.  This feed only
// traces: a finished ray leaves its hit beside its queue entry and the lane is refilled.
struct TraceFeed {
    static constexpr bool kPretested = true;
    const SceneDev &S;
    const Queue &qin;
    HitQueue hits;
    size_t in_base;
    Blocks blocks;
    uint32_t entry;
    uint32_t code = 0; // what the producer of the ray a lane just received found out (pretest)
    RT_DEV bool more() const { return blocks.more(); }
    RT_DEV bool turn(Trav &T, bool free) {
        Lane &L = T.L;
        if (free && L.has_ray) {
            hits.a[in_base + entry] = make_float4(__uint_as_float(L.best), L.ray.t_max, L.a0, L.a1);
            hits.b[in_base + entry] = L.a2;
        }
        const uint32_t next = blocks.take(free);
        bool fresh = false;
        if (free) {
            L.has_ray = false;
            if (next != kNone) {
                const float4 qa = qin.a[in_base + next], qb = qin.b[in_base + next];
                uint2 qe = make_uint2(__float_as_uint(kInf), 0u);
                if (S.n_enclosing | S.n_balls) qe = qin.e[in_base + next];
                if (__float_as_uint(qb.w) != kDead) {
                    L.ray.o = mk(qa.x, qa.y, qa.z);
                    L.ray.time = qa.w;
                    L.ray.d = mk(qb.x, qb.y, qb.z);
                    L.ray.t_max = __uint_as_float(qe.x);
                    bool resolved;
                    L.best = pretest_best(S, qe.y, resolved);
                    code = qe.y;
                    L.a0 = L.a1 = L.a2 = 0.0f;
                    entry = next;
                    L.has_ray = true;
                    fresh = true;
                }
            }
        }
        return fresh;
    }
    RT_DEV bool drain() { return false; }
    RT_DEV void finish() {}
};

template <bool COUNT, bool LDSN, bool WIDE = false>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_trace(
    SceneDev S, Queue qin, HitQueue hits, uint32_t *counts, uint32_t bounce, unsigned long long *counters) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t *cin = counts + (size_t)bounce * kShards * kCountStride;
    const uint32_t my_count = lane < kShards ? cin[lane * kCountStride] : 0u;
    uint32_t longest = my_count;
    for (int s = 32; s > 0; s >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, s, 64));
    const uint32_t n_chunks = kShards * ((longest + 63) / 64);
    if (blockIdx.x * (blockDim.x >> 6) >= n_chunks) return; // nothing for this workgroup (uniform)
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    const uint32_t w = wave_index(), shard = shard_of_wave(w), per_shard = wave_total() / kShards;
    const uint32_t n = __builtin_amdgcn_readlane(my_count, shard);
    TraceFeed feed{S, qin, hits, (size_t)shard * qin.cap,
                   Blocks{(w / kShards) * 64, 0u, n, per_shard * 64,
                          counts + ((size_t)bounce * kShards + shard) * kCountStride + 1, per_shard, 1u, 0u},
                   0u};
    trace_stream<false, COUNT, LDSN, WIDE>(S, tv, feed, cnt,
                                     bounce == 0 ? primary_refill_lanes_of<LDSN>()
                                                 : (LDSN ? kRefillLanes : (uint32_t)RT_REFILL_LANES_GLOBAL_TRACE));
    if (COUNT) flush_counters(cnt, counters);
}

// Shades the segments of one bounce from their queue entries and the hits k_trace left, 64
// consecutive entries per wave and step, and compacts the survivors into the next queue: one
// ballot, one prefix count and one atomic per wave and step, as BounceFeed::compact does.
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
template <bool NEE>
__global__ __launch_bounds__(kBlock, RT_SHADE_WAVES) void k_shade(
    SceneDev S, Queue qin, Queue qout, HitQueue hits, uint32_t *counts, uint32_t bounce, uint32_t last_bounce,
    uint64_t seed, float4 *radiance, ShadowQueue sq, uint32_t *scounts) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t *cin = counts + (size_t)bounce * kShards * kCountStride;
    uint32_t *cout = counts + (size_t)(bounce + 1) * kShards * kCountStride;
    const uint32_t my_count = lane < kShards ? cin[lane * kCountStride] : 0u;
    const uint32_t w = wave_index(), shard = w % kShards, per_shard = wave_total() / kShards;
    const uint32_t n = __builtin_amdgcn_readlane(my_count, shard);
    const size_t in_base = (size_t)shard * qin.cap, out_base = (size_t)shard * qout.cap;
    uint32_t *tail = cout + shard * kCountStride;
    uint32_t *stail = NEE ? scounts + ((size_t)bounce * kShards + shard) * kCountStride : nullptr;
    const size_t shadow_base = (size_t)shard * sq.cap;
    const bool last = bounce == last_bounce;
    for (uint32_t blk = (w / kShards) * 64u; blk < n; blk += per_shard * 64u) {
        bool alive = false;
        Lane P;
        P.ray.o = P.ray.d = mk(0.0f, 0.0f, 0.0f);
        P.ray.t_max = P.ray.time = 0.0f;
        v3 beta = mk(0.0f, 0.0f, 0.0f);
        Rng rng{0, 0};
        uint32_t gpath = 0, local = kDead;
        Shadow sh;
        sh.want = sh.sampled = false;
        sh.o = sh.d = sh.contrib = mk(0.0f, 0.0f, 0.0f);
        const size_t e = in_base + blk + lane;
        float4 qb = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kDead));
        if (blk + lane < n) qb = qin.b[e];
        if (__float_as_uint(qb.w) != kDead) {
            const float4 qa = qin.a[e], qc = qin.c[e], ha = hits.a[e];
            const uint2 qd = qin.d[e];
            P.ray.o = mk(qa.x, qa.y, qa.z);
            P.ray.time = qa.w;
            P.ray.d = mk(qb.x, qb.y, qb.z);
            P.ray.t_max = ha.y;
            P.best = __float_as_uint(ha.x);
            P.a0 = ha.z;
            P.a1 = ha.w;
            P.a2 = hits.b[e];
            P.has_ray = true;
            local = __float_as_uint(qb.w);
            beta = mk(qc.x, qc.y, qc.z);
            gpath = qd.y;
            rng.state = ((uint64_t)qd.x << 32) | __float_as_uint(qc.w);
            rng.inc = (((seed << 32) + gpath) << 1) | 1u;
            const bool general = P.best != RT_MISS && (P.best & kBestRare) != 0;
            bool skip_emission = false;
            if (NEE) { // bit 31 of the path index: the vertex before this one sampled the lights
                skip_emission = (local >> 31) != 0;
                local &= 0x7fffffffu;
            }
            if (general) alive = shade<false, NEE>(S, P, beta, rng, local, radiance, skip_emission, &sh) && !last;
            else alive = shade<true, NEE>(S, P, beta, rng, local, radiance, skip_emission, &sh) && !last;
        }
        if (NEE) {
            const uint64_t mask = __ballot(sh.want);
            const uint32_t total = (uint32_t)__popcll(mask);
            if (total) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(stail, total);
                base = __builtin_amdgcn_readfirstlane(base);
                if (sh.want) {
                    const size_t slot = shadow_base + base + lane_rank(mask);
                    sq.a[slot] = make_float4(sh.o.x, sh.o.y, sh.o.z, P.ray.time);
                    sq.b[slot] = make_float4(sh.d.x, sh.d.y, sh.d.z, __uint_as_float(local));
                    sq.c[slot] = make_float4(sh.contrib.x, sh.contrib.y, sh.contrib.z, 0.0f);
                }
            }
        }
        const uint64_t mask = __ballot(alive);
        const uint32_t total = (uint32_t)__popcll(mask);
        if (total) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(tail, total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (alive)
                queue_push(S, qout, out_base + base + lane_rank(mask), P.ray,
                           NEE && sh.sampled ? (local | 0x80000000u) : local, beta, rng, gpath);
        }
    }
}

// ---- shadow rays (direct lighting) ----------------------------------------------------------------------
// The rays Interaction::spawn_ray_to made during one bounce, through the any-hit traversal
// (<BVH as Primitive>::does_intersect, shapes/src/bvh.rs:801-863).  A ray that reaches its
// light adds its contribution to its path's radiance; a path has at most one shadow ray per
// bounce, so no two lanes touch the same slot.
struct ShadowFeed {
    static constexpr bool kPretested = false;
    const ShadowQueue &q;
    size_t base;
    Blocks blocks;
    float4 *radiance;
    uint32_t idx;
    RT_DEV bool more() const { return blocks.more(); }
    RT_DEV bool turn(Trav &T, bool free) {
        Lane &L = T.L;
        if (free && L.has_ray && L.best == RT_MISS) {
            const float4 c = q.c[base + idx];
            const uint32_t local = __float_as_uint(q.b[base + idx].w);
            const float4 r = radiance[local];
            radiance[local] = make_float4(r.x + c.x, r.y + c.y, r.z + c.z, 0.0f);
        }
        const uint32_t next = blocks.take(free);
        if (free) {
            idx = next;
            L.has_ray = next != kNone;
            if (L.has_ray) {
                const float4 a = q.a[base + next], b = q.b[base + next];
                L.ray.o = mk(a.x, a.y, a.z);
                L.ray.time = a.w;
                L.ray.d = mk(b.x, b.y, b.z);
                L.ray.t_max = kShadowTMax;
            }
        }
        return free && L.has_ray;
    }
    RT_DEV bool drain() { return false; }
    RT_DEV void finish() {}
};

template <bool LDSN, bool WIDE = false>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_shadow(
    SceneDev S, ShadowQueue q, uint32_t *scounts, uint32_t bounce, float4 *radiance) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t *cin = scounts + (size_t)bounce * kShards * kCountStride;
    const uint32_t my_count = lane < kShards ? cin[lane * kCountStride] : 0u;
    uint32_t longest = my_count;
    for (int s = 32; s > 0; s >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, s, 64));
    const uint32_t n_chunks = kShards * ((longest + 63) / 64);
    if (blockIdx.x * (blockDim.x >> 6) >= n_chunks) return; // nothing for this workgroup (uniform)
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    const uint32_t w = wave_index(), shard = shard_of_wave(w), per_shard = wave_total() / kShards;
    const uint32_t n = __builtin_amdgcn_readlane(my_count, shard);
    ShadowFeed feed{q, (size_t)shard * q.cap,
                    Blocks{(w / kShards) * 64, 0u, n, per_shard * 64,
                           scounts + ((size_t)bounce * kShards + shard) * kCountStride + 1, per_shard, 1u, 0u},
                    radiance, kNone};
    trace_stream<true, false, LDSN, WIDE>(S, tv, feed, cnt);
}

// ---- the rest of the path in one launch ------------------------------------------------------------
// Late bounces carry few rays, and a launch lasts a wave's whole chain fetch -> traverse ->
// park -> shade -> push however few it has: fifty such launches are a floor under the frame
// time that matters when the frame is split over several GPUs.  This feed does not hand a
// bounce's survivors to the next launch: a wave keeps the survivors of its own shading batches
// in a ring of its own and feeds them back to its free lanes, so one launch takes the rays
// of bounce `bounce0` to the end of their paths.  Per wave, in global memory:
//   parking lists  2 x kParkSlots x 5 float4: ray, hit and path state of a finished segment
//   lane area      64 x 2 float4: path state of the segment a lane is tracing (throughput,
//                  generator, path id, pixel slot, depth), written when the lane is fed
//   ring           kRingSlots x 4 float4: segments waiting for a lane, first in first out
//   save area      as in BounceFeed
// A wave never holds more than 64 + 2 x 127 + 128 segments, so the ring cannot overflow.
// free lanes that trigger a refill turn of this kernel: its turn is the dearest of all (whole segments
// are parked, ring and queue are both looked at), so it waits for more lanes than the others do
// (measured on the headline frame: 24 lanes 69.0 ms, 32 68.4, 40 68.0, 48 68.2, 56 68.6)
#ifndef RT_LOOP_REFILL_LANES
#define RT_LOOP_REFILL_LANES 40
#endif
constexpr uint32_t kRingSlots = 512;
constexpr size_t kLoopParkBytes = 2 * kParkSlots * 5 * 16, kLoopLaneBytes = 64 * 2 * 16,
                 kLoopRingBytes = kRingSlots * 4 * 16;
static_assert(kWaveLoopBytes == kLoopParkBytes + kLoopLaneBytes + kLoopRingBytes + kSaveWords * 64 * 4,
              "per-wave memory of the loop kernel");

struct LoopFeed {
    static constexpr bool kPretested = true;
    const SceneDev &S;
    const Queue &qin;
    size_t in_base;
    Blocks blocks;
    uint32_t *counts; // queue sizes by bounce: here only statistics (segments traced per bounce)
    uint32_t shard, bounce0, last;
    uint64_t seed;
    float4 *radiance;
    float4 *park;   // [list][slot][5]
    float4 *lane_c; // throughput.rgb, generator state low word
    uint4 *lane_d;  // generator state high word, path id, pixel slot, depth
    float4 *ring;   // [slot][4]: a, b, c as in Queue, d = lane_d
    uint32_t *save;
    uint32_t parked[2];
    uint32_t ring_head, ring_count, turns, turn_limit;
    uint32_t code = 0; // what the producer of the ray a lane just received found out (pretest)
    Counters *cnt = nullptr; // counting launches: where segments shaded without a traversal are counted
#ifdef RT_TAIL_PROBE
    unsigned long long probe_dry = 0; // clock when this wave first found the queue read out
    uint32_t probe_held = 0, probe_turns_dry = 0, probe_batches = 0, probe_batch_fill = 0;
#endif

    RT_DEV bool more() const { return ring_count != 0 || blocks.more(); }

    template <bool PLAIN>
    RT_DEV void shade_batch(uint32_t first, uint32_t count) {
        const uint32_t lane = threadIdx.x & 63;
        bool alive = false;
        Lane P;
        v3 beta = mk(0.0f, 0.0f, 0.0f);
        Rng rng{0, 0};
        uint32_t gpath = 0, local = kDead, depth = 0;
#ifdef RT_TAIL_PROBE
        if (probe_dry) {
            probe_batches++;
            probe_batch_fill += count;
        }
#endif
        // records parked by other lanes (also by the batch before this one) are read here
        asm volatile("" ::: "memory");
        __threadfence_block();
        if (lane < count) {
            const float4 *r = park + (size_t)(first + lane) * 5;
            const float4 p0 = r[0], p1 = r[1], p2 = r[2], p3 = r[3], p4 = r[4];
            P.ray.o = mk(p0.x, p0.y, p0.z);
            P.ray.time = p0.w;
            P.ray.d = mk(p1.x, p1.y, p1.z);
            P.ray.t_max = p1.w;
            P.best = __float_as_uint(p2.x);
            P.a0 = p2.y;
            P.a1 = p2.z;
            P.a2 = p2.w;
            P.has_ray = true;
            beta = mk(p3.x, p3.y, p3.z);
            gpath = __float_as_uint(p4.y);
            local = __float_as_uint(p4.z);
            depth = __float_as_uint(p4.w);
            rng.state = ((uint64_t)__float_as_uint(p4.x) << 32) | __float_as_uint(p3.w);
            rng.inc = (((seed << 32) + gpath) << 1) | 1u;
        }
        if (lane < count) alive = shade<PLAIN>(S, P, beta, rng, local, radiance) && depth < last;
        // the new ray against the media that enclose the scene and its ball, here where the
        // whole batch works: t_max after them, and what was found (the pretest's code)
        float t_max = kInf;
        uint32_t which = 0;
        if (alive && (S.n_enclosing | S.n_balls)) {
            Ray fresh = P.ray;
            fresh.t_max = kInf;
            pretest(S, fresh, t_max, which);
        }
        // Survivors join the ring as segments of the next bounce -- unless the pretest already
        // decides the next segment's hit (a ray inside a ball).  That segment is a finished one:
        // it needs no lane of the traversal.  If the plain shader covers its hit it goes straight
        // back to the parking list this batch came from, whose slots are free again (all lanes
        // have read their records), to be shaded with the next full batch: a path scattering
        // inside a fog-filled ball runs from shading to shading without ring entry or lane.
        // (Shading it at once, in this batch, was tried too: as long as at least N lanes wanted
        // to, for N = 32 ... 56; the fuller batches of the parking list won at every N.)
        bool resolved = false;
        uint32_t next_best = RT_MISS;
        if (PLAIN && alive) next_best = pretest_best(S, which, resolved);
        const bool done = PLAIN && alive && resolved && next_best != RT_MISS && (next_best & kBestRare) == 0;
        const uint64_t m_done = __ballot(done);
        if (done) {
            float4 *r = park + (size_t)(parked[0] + lane_rank(m_done)) * 5;
            r[0] = make_float4(P.ray.o.x, P.ray.o.y, P.ray.o.z, P.ray.time);
            r[1] = make_float4(P.ray.d.x, P.ray.d.y, P.ray.d.z, t_max);
            r[2] = make_float4(__uint_as_float(next_best), 0.0f, 0.0f, 0.0f);
            r[3] = make_float4(beta.x, beta.y, beta.z, __uint_as_float((uint32_t)rng.state));
            r[4] = make_float4(__uint_as_float((uint32_t)(rng.state >> 32)), __uint_as_float(gpath),
                               __uint_as_float(local), __uint_as_float(depth + 1));
            if (cnt) { // what the traversal counts for a resolved ray
                cnt->prims[RT_MEDIUM] += S.n_enclosing;
                count_resolved(S, which, *cnt);
            }
        }
        parked[0] += (uint32_t)__popcll(m_done);
        const bool queued = alive && !done;
        const uint64_t mask = __ballot(queued);
        if (queued) {
            float4 *e = ring + (size_t)((ring_head + ring_count + lane_rank(mask)) % kRingSlots) * 4;
            e[0] = make_float4(P.ray.o.x, P.ray.o.y, P.ray.o.z, P.ray.time);
            e[1] = make_float4(P.ray.d.x, P.ray.d.y, P.ray.d.z, t_max);
            e[2] = make_float4(beta.x, beta.y, beta.z, __uint_as_float((uint32_t)rng.state));
            e[3] = make_float4(__uint_as_float((uint32_t)(rng.state >> 32)), __uint_as_float(gpath),
                               __uint_as_float(local), __uint_as_float((depth + 1) | (which << 16)));
        }
        ring_count += (uint32_t)__popcll(mask);
        // segments per bounce, where the per-bounce kernels keep their queue sizes
        uint64_t todo = __ballot(alive);
        while (todo) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
            const uint32_t dd = (uint32_t)__builtin_amdgcn_readlane((int)depth, (int)leader);
            const uint64_t same = __ballot(alive && depth == dd);
            if (lane == leader)
                atomicAdd(&counts[((size_t)(dd + 1) * kShards + shard) * kCountStride], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }

    RT_DEV void shade_lists(uint32_t at_least) {
        while (parked[0] >= at_least && parked[0] != 0) {
            const uint32_t count = parked[0] < 64u ? parked[0] : 64u;
            parked[0] -= count;
            shade_batch<true>(parked[0], count);
        }
        while (parked[1] >= at_least && parked[1] != 0) {
            const uint32_t count = parked[1] < 64u ? parked[1] : 64u;
            parked[1] -= count;
            shade_batch<false>(kParkSlots + parked[1], count);
        }
    }

    RT_DEV bool turn(Trav &T, bool free) {
        Lane &L = T.L;
        const uint32_t lane = threadIdx.x & 63;
        if (++turns > turn_limit) {
            // No frame comes near the limit; it only keeps a wave from spinning forever.  The wave
            // drops what it holds, so the frame is incomplete: say so where the host looks after
            // every batch (it then fails the call instead of returning a wrong image).
            ring_count = 0;
            blocks.blk = blocks.n;
            if (lane == 0) atomicOr(&counts[kLoopErrorWord], 1u);
        }
#ifdef RT_TAIL_PROBE
        if (!blocks.more()) {
            if (!probe_dry) {
                probe_dry = wall_clock64();
                probe_held = ring_count + parked[0] + parked[1] + (uint32_t)__popcll(__ballot(L.has_ray));
            }
            probe_turns_dry++;
        }
#endif
        // park the finished segments, whole
        const bool finished = free && L.has_ray;
        const bool general = L.best != RT_MISS && (L.best & kBestRare) != 0;
        const uint64_t m0 = __ballot(finished && !general), m1 = __ballot(finished && general);
        if (finished) {
            const uint32_t slot = general ? kParkSlots + parked[1] + lane_rank(m1) : parked[0] + lane_rank(m0);
            float4 *r = park + (size_t)slot * 5;
            const uint4 d4 = lane_d[lane];
            r[0] = make_float4(L.ray.o.x, L.ray.o.y, L.ray.o.z, L.ray.time);
            r[1] = make_float4(L.ray.d.x, L.ray.d.y, L.ray.d.z, L.ray.t_max);
            r[2] = make_float4(__uint_as_float(L.best), L.a0, L.a1, L.a2);
            r[3] = lane_c[lane];
            r[4] = make_float4(__uint_as_float(d4.x), __uint_as_float(d4.y), __uint_as_float(d4.z), __uint_as_float(d4.w));
        }
        parked[0] += (uint32_t)__popcll(m0);
        parked[1] += (uint32_t)__popcll(m1);
        if (parked[0] >= 64 || parked[1] >= 64) {
            const bool any_busy = __ballot(T.cur != kNone) != 0;
            uint32_t *sv = save + lane;
            if (any_busy) {
                sv[0 * 64] = __float_as_uint(L.ray.o.x); sv[1 * 64] = __float_as_uint(L.ray.o.y);
                sv[2 * 64] = __float_as_uint(L.ray.o.z); sv[3 * 64] = __float_as_uint(L.ray.d.x);
                sv[4 * 64] = __float_as_uint(L.ray.d.y); sv[5 * 64] = __float_as_uint(L.ray.d.z);
                sv[6 * 64] = __float_as_uint(L.ray.t_max); sv[7 * 64] = __float_as_uint(L.ray.time);
                sv[8 * 64] = L.best; sv[9 * 64] = __float_as_uint(L.a0);
                sv[10 * 64] = __float_as_uint(L.a1); sv[11 * 64] = __float_as_uint(L.a2);
                sv[12 * 64] = __float_as_uint(T.inv.x); sv[13 * 64] = __float_as_uint(T.inv.y);
                sv[14 * 64] = __float_as_uint(T.inv.z); sv[15 * 64] = T.cur;
                sv[16 * 64] = T.sp;
            }
            asm volatile("" ::: "memory");
            __threadfence_block();
            shade_lists(64);
            asm volatile("" ::: "memory");
            if (any_busy) {
                L.ray.o = mk(__uint_as_float(sv[0 * 64]), __uint_as_float(sv[1 * 64]), __uint_as_float(sv[2 * 64]));
                L.ray.d = mk(__uint_as_float(sv[3 * 64]), __uint_as_float(sv[4 * 64]), __uint_as_float(sv[5 * 64]));
                L.ray.t_max = __uint_as_float(sv[6 * 64]);
                L.ray.time = __uint_as_float(sv[7 * 64]);
                L.best = sv[8 * 64];
                L.a0 = __uint_as_float(sv[9 * 64]);
                L.a1 = __uint_as_float(sv[10 * 64]);
                L.a2 = __uint_as_float(sv[11 * 64]);
                T.inv = mk(__uint_as_float(sv[12 * 64]), __uint_as_float(sv[13 * 64]), __uint_as_float(sv[14 * 64]));
                T.cur = sv[15 * 64];
                T.sp = sv[16 * 64];
            } else {
                L.ray.o = L.ray.d = T.inv = mk(0.0f, 0.0f, 0.0f);
                L.ray.t_max = L.ray.time = L.a0 = L.a1 = L.a2 = 0.0f;
                L.best = RT_MISS;
                T.cur = kNone;
                T.sp = 0;
            }
        }
        // ring entries written by other lanes are read below
        asm volatile("" ::: "memory");
        __threadfence_block();
        // next segments: the ring first, then unread rays of the queue
        const uint64_t want = __ballot(free);
        const uint32_t rank = lane_rank(want);
        const uint32_t from_ring = min((uint32_t)__popcll(want), ring_count);
        const bool ring_lane = free && rank < from_ring;
        const uint32_t next = blocks.take(free && !ring_lane);
        bool fresh = false;
        if (free) {
            L.has_ray = false;
            float4 qa, qb, qc;
            uint4 qd = make_uint4(0u, 0u, 0u, 0u);
            bool got = false;
            float t0 = kInf;    // the producer's findings in the media that enclose the scene:
            uint32_t which = 0; // t_max after them, and which of them the ray scatters in
            if (ring_lane) {
                const float4 *e = ring + (size_t)((ring_head + rank) % kRingSlots) * 4;
                qa = e[0];
                qb = e[1];
                qc = e[2];
                const float4 d4 = e[3];
                t0 = qb.w;
                // depth in the low half (rt_render_dev keeps max_depth below 2^16), the pretest's
                // 16-bit code in the high half
                which = __float_as_uint(d4.w) >> 16;
                qd = make_uint4(__float_as_uint(d4.x), __float_as_uint(d4.y), __float_as_uint(d4.z),
                                __float_as_uint(d4.w) & 0xffffu);
                got = true;
            } else if (next != kNone) {
                qa = qin.a[in_base + next];
                qb = qin.b[in_base + next];
                if (__float_as_uint(qb.w) != kDead) {
                    qc = qin.c[in_base + next];
                    const uint2 d2 = qin.d[in_base + next];
                    if (S.n_enclosing | S.n_balls) {
                        const uint2 qe = qin.e[in_base + next];
                        t0 = __uint_as_float(qe.x);
                        which = qe.y;
                    }
                    qd = make_uint4(d2.x, d2.y, __float_as_uint(qb.w), bounce0);
                    got = true;
                }
            }
            if (got) {
                L.ray.o = mk(qa.x, qa.y, qa.z);
                L.ray.time = qa.w;
                L.ray.d = mk(qb.x, qb.y, qb.z);
                L.ray.t_max = t0;
                bool resolved;
                L.best = pretest_best(S, which, resolved);
                code = which;
                L.a0 = L.a1 = L.a2 = 0.0f;
                lane_c[lane] = qc;
                lane_d[lane] = qd;
                L.has_ray = true;
                fresh = true;
            }
        }
        ring_count -= from_ring;
        // an empty ring starts over at slot 0: the same few kilobytes stay in cache instead of
        // the head walking through all kRingSlots
        ring_head = ring_count ? (ring_head + from_ring) % kRingSlots : 0u;
        return fresh;
    }

    // Nothing in the lanes, the ring or the queue: the paths parked in incomplete batches go on.
    // One batch per list and call, so that survivors which need a traversal get their lanes at once
    // instead of waiting in the ring while the resolved ones (which come straight back to the
    // list) are followed to their ends: every path has dozens of generations to go, and what
    // counts at the end of a launch is that they all advance side by side.
    RT_DEV bool drain() {
        asm volatile("" ::: "memory");
        __threadfence_block();
        if (parked[0] != 0) {
            const uint32_t count = parked[0] < 64u ? parked[0] : 64u;
            parked[0] -= count;
            shade_batch<true>(parked[0], count);
        }
        if (parked[1] != 0) {
            const uint32_t count = parked[1] < 64u ? parked[1] : 64u;
            parked[1] -= count;
            shade_batch<false>(kParkSlots + parked[1], count);
        }
        asm volatile("" ::: "memory");
        __threadfence_block();
        return (ring_count | parked[0] | parked[1]) != 0;
    }
    RT_DEV void finish() {}
};

// Bounce `bounce` and every later one of the paths queued for it, in one launch (LoopFeed).
template <bool COUNT, bool LDSN>
__global__ __launch_bounds__(Shape<LDSN>::kThreads, Shape<LDSN>::kWaves) void k_bounce_loop(
    SceneDev S, Queue qin, uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed, float4 *radiance,
    uint8_t *scratch, unsigned long long *counters, uint32_t steal, uint32_t turn_limit) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t *cin = counts + (size_t)bounce * kShards * kCountStride;
    const uint32_t my_count = lane < kShards ? cin[lane * kCountStride] : 0u;
    uint32_t longest = my_count;
    for (int s = 32; s > 0; s >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, s, 64));
    const uint32_t n_chunks = kShards * ((longest + 63) / 64);
    if (blockIdx.x * (blockDim.x >> 6) >= n_chunks) return; // nothing for this workgroup (uniform)
#ifdef RT_TAIL_PROBE
    const unsigned long long probe_t0 = wall_clock64();
#endif
    const TreeView<LDSN> tv = tree_view<LDSN>(S);
    Counters cnt = {};
    const uint32_t w = wave_index(), shard = shard_of_wave(w), per_shard = wave_total() / kShards;
    const uint32_t n = __builtin_amdgcn_readlane(my_count, shard);
    uint8_t *mine = scratch + (size_t)w * kWaveLoopBytes;
    // the block claimer hands out entry numbers of the whole queue (it may move on to other
    // sub-queues), so the feed's own offset is zero
    LoopFeed feed{S, qin, 0,
                  Blocks{(w / kShards) * 64, 0u, n, per_shard * 64,
                         counts + ((size_t)bounce * kShards + shard) * kCountStride + 1, per_shard, 1u, 0u,
                         shard * qin.cap, steal ? cin : nullptr, shard, qin.cap, 0u},
                  counts, shard, bounce, last, seed, radiance,
                  reinterpret_cast<float4 *>(mine),
                  reinterpret_cast<float4 *>(mine + kLoopParkBytes),
                  reinterpret_cast<uint4 *>(mine + kLoopParkBytes + 64 * 16),
                  reinterpret_cast<float4 *>(mine + kLoopParkBytes + kLoopLaneBytes),
                  reinterpret_cast<uint32_t *>(mine + kLoopParkBytes + kLoopLaneBytes + kLoopRingBytes),
                  {0u, 0u}, 0u, 0u, 0u, turn_limit};
    if (COUNT) feed.cnt = &cnt;
    feed.blocks.start();
    trace_stream<false, COUNT, LDSN, false>(S, tv, feed, cnt, LDSN ? (uint32_t)RT_LOOP_REFILL_LANES : (uint32_t)RT_BOUNCE_REFILL_LANES_GLOBAL);
    if (COUNT) flush_counters(cnt, counters);
#ifdef RT_TAIL_PROBE
    // when do the waves of a launch finish?  (100 MHz clock; slots 28-31 of the counters)
    if (COUNT && lane == 0) {
        const unsigned long long t1 = wall_clock64();
        // per-wave record over the start of the wave's own memory (no longer in use)
        unsigned long long *rec = reinterpret_cast<unsigned long long *>(mine);
        rec[0] = probe_t0;
        rec[1] = feed.probe_dry;
        rec[2] = t1;
        rec[3] = ((unsigned long long)feed.probe_held << 32) | feed.probe_turns_dry;
        rec[4] = ((unsigned long long)feed.probe_batches << 32) | feed.probe_batch_fill;
        atomicMax(&counters[28], ~probe_t0);
        atomicMax(&counters[29], t1);
        atomicAdd(&counters[30], t1 - probe_t0);
        atomicAdd(&counters[31], 1ull);
    }
#endif
}

// Adds the batch's samples to the running per-pixel sums in sample order; the
// last batch divides by spp.  One thread per owned pixel.  A pixel's samples are
// consecutive in memory, so a wave's 64 pixels are 64 rows of s_count entries: the wave
// reads them eight columns at a time with its lanes running along the rows (8 rows x 128
// contiguous bytes per load), passes them through LDS, and every lane then adds its own
// row's eight values in order.
constexpr uint32_t kResolveCols = 8;
__global__ __launch_bounds__(kBlock) void k_resolve(FrameDev F, const float4 *radiance, uint32_t n_ranks,
                                                    uint32_t s_count, int finish, float4 *rgba) {
    __shared__ float4 tile[kBlock / 64][64][kResolveCols + 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t rank0 = blockIdx.x * kBlock + wave * 64u, rank = rank0 + lane;
    uint32_t px = 0, py = 0;
    const bool owned = rank < n_ranks && rank_to_pixel(F, rank, px, py);
    const size_t pix = (size_t)py * F.width + px;
    float4 acc = owned ? rgba[pix] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t col = lane % kResolveCols, row_in_load = lane / kResolveCols; // 8 rows per load
    for (uint32_t s0 = 0; s0 < s_count; s0 += kResolveCols) {
#pragma unroll
        for (uint32_t i = 0; i < 64u / (64u / kResolveCols); i++) {
            const uint32_t row = row_in_load + i * (64u / kResolveCols);
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (rank0 + row < n_ranks && s0 + col < s_count) v = radiance[(size_t)(rank0 + row) * s_count + s0 + col];
            tile[wave][row][col] = v;
        }
        __syncthreads();
        const uint32_t cols = s_count - s0 < kResolveCols ? s_count - s0 : kResolveCols;
        for (uint32_t c = 0; c < cols; c++) {
            const float4 v = tile[wave][lane][c];
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
        }
        __syncthreads();
    }
    if (!owned) return;
    if (finish) {
        float inv = (float)F.spp;
        acc.x = acc.x / inv;
        acc.y = acc.y / inv;
        acc.z = acc.z / inv;
        acc.w = 1.0f;
    }
    rgba[pix] = acc;
}

// FilmTile::add_sample (film.rs:281-328) in gather form: one thread per image pixel adds,
// for each sample index of the batch in turn, the samples of the pixels around it row by
// row (those this shard traced), weighted by the filter table.  Sample positions are the
// first two draws of the path's generator, regenerated here.  The running sums
// {sum w*L, sum w} stay in rgba between batches; the order of the additions is fixed, so
// the result does not depend on how the samples are batched.
__global__ __launch_bounds__(kBlock) void k_film(FrameDev F, FilterDev T, const float4 *radiance, uint32_t s0,
                                                 uint32_t s_count, float4 *rgba) {
    const size_t pix = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (pix >= (size_t)F.width * F.height) return;
    const int x = (int)(pix % F.width), y = (int)(pix / F.width);
    const float inv_rx = 1.0f / T.rx, inv_ry = 1.0f / T.ry;
    const int Rx = (int)ceilf(T.rx + 0.5f), Ry = (int)ceilf(T.ry + 0.5f);
    const float fxp = (float)x, fyp = (float)y;
    float4 acc = rgba[pix];
    for (uint32_t k = 0; k < s_count; k++)
        for (int qy = y - Ry; qy <= y + Ry; qy++)
            for (int qx = x - Rx; qx <= x + Rx; qx++) {
                if (qx < 0 || qy < 0 || qx >= (int)F.width || qy >= (int)F.height) continue;
                const uint32_t tile = ((uint32_t)qy >> 5) * F.tiles_x + ((uint32_t)qx >> 5);
                if (tile % F.shard_count != F.shard_index) continue;
                const uint32_t rank = (tile / F.shard_count) * 1024u + (((uint32_t)qy & 31u) << 5) + ((uint32_t)qx & 31u);
                const uint64_t path = ((uint64_t)qy * F.width + (uint64_t)qx) * F.spp + (s0 + k);
                Rng rng;
                float u[5];
                rng_seq(rng, (F.seed << 32) + path);
                camera_sample(F, rank, (uint32_t)qx, (uint32_t)qy, s0 + k, rng, u);
                const float dx = ((float)qx + u[0]) - 0.5f, dy = ((float)qy + u[1]) - 0.5f;
                if (fxp < ceilf(dx - T.rx) || fxp > floorf(dx + T.rx)) continue;
                if (fyp < ceilf(dy - T.ry) || fyp > floorf(dy + T.ry)) continue;
                int ix = (int)floorf(fabsf((fxp - dx) * inv_rx * 16.0f));
                int iy = (int)floorf(fabsf((fyp - dy) * inv_ry * 16.0f));
                if (ix > 15) ix = 15;
                if (iy > 15) iy = 15;
                const float w = T.w[iy * 16 + ix];
                const float4 L = radiance[(size_t)rank * s_count + k];
                acc.x += L.x * w;
                acc.y += L.y * w;
                acc.z += L.z * w;
                acc.w += w;
            }
    rgba[pix] = acc;
}

__global__ __launch_bounds__(kBlock) void k_film_resolve(float4 *rgba, size_t n) { // film.rs:236-247
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = rgba[i];
    if (v.w != 0.0f) {
        const float inv = 1.0f / v.w;
        v.x = sel_max(v.x * inv, 0.0f);
        v.y = sel_max(v.y * inv, 0.0f);
        v.z = sel_max(v.z * inv, 0.0f);
    }
    v.w = 1.0f;
    rgba[i] = v;
}

// ---- host-side launchers ------------------------------------------------------------------------------
// dynamic LDS: the LDS part of the stacks (child reference + entry distance per level and
// lane), preceded by the interior records for the LDS-resident shape
size_t lds_bytes(bool lds_nodes, uint32_t n_pairs, uint32_t stack_lds) {
    return (lds_nodes ? (size_t)n_pairs * 64 + (size_t)stack_lds * 1024 * (sizeof(float) + sizeof(uint16_t))
                      : (size_t)stack_lds * kBlock * (sizeof(float) + sizeof(uint32_t))) + 256; // + slack
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
                            rt_hit *hits, unsigned long long *counters, uint32_t *cursor, hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    uint32_t blocks = (n + threads - 1) / threads;
    if (blocks > L.grid) blocks = L.grid;
    if (blocks == 0) return hipSuccess;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) {
        if (counters) RT_LAUNCH((k_intersect<true, true>), blocks, threads, S, rays, n, hits, counters, cursor);
        else RT_LAUNCH((k_intersect<false, true>), blocks, threads, S, rays, n, hits, counters, cursor);
    } else {
        if (counters) RT_LAUNCH((k_intersect<true, false>), blocks, threads, S, rays, n, hits, counters, cursor);
        else if (L.quads == 2) RT_LAUNCH((k_intersect<false, false, true>), blocks, threads, S, rays, n, hits, counters, cursor);
        else RT_LAUNCH((k_intersect<false, false>), blocks, threads, S, rays, n, hits, counters, cursor);
    }
    return hipGetLastError();
}

hipError_t launch_occluded(const SceneDev &S, const LaunchShape &L, const rt_ray *rays, uint32_t n, uint8_t *out,
                           uint32_t *cursor, hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    uint32_t blocks = (n + threads - 1) / threads;
    if (blocks > L.grid) blocks = L.grid;
    if (blocks == 0) return hipSuccess;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) RT_LAUNCH((k_occluded<true>), blocks, threads, S, rays, n, out, cursor);
    else if (L.quads == 2) RT_LAUNCH((k_occluded<false, true>), blocks, threads, S, rays, n, out, cursor);
    else RT_LAUNCH((k_occluded<false>), blocks, threads, S, rays, n, out, cursor);
    return hipGetLastError();
}

// ---- block cursors of the entry point that allocates nothing ------------------------------------------
// The trace kernels claim their ray blocks from kShards counters (one per 64-byte line) that are
// zero at launch.  The ordinary entry points keep one such block per caller stream in memory
// from hipMalloc; rt_intersect_dev_noalloc uses this file-scope array instead.  Every device
// that loads the library holds its own copy at its own address, which host code must ask the
// runtime for: the name below is only a stand-in on the host side.
static __device__ __align__(256) uint32_t scratch[kShards * kCountStride];
static_assert(sizeof(scratch) == kCursorBytes, "one cursor block");

namespace {
constexpr int kMaxDevices = 64;
uint32_t *g_scratch_ptr[kMaxDevices]; // address of `scratch` on device d, or null before static_cursor_setup(d)
std::mutex g_scratch_lock;
} // namespace

// Once per device, with that device current (the scene upload does it).
hipError_t static_cursor_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_scratch_lock);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = static_cast<uint32_t *>(p);
    return hipSuccess;
}

uint32_t *static_cursor(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> g(g_scratch_lock);
    return g_scratch_ptr[device];
}

hipError_t launch_pixel_samples(const FrameDev &F, uint32_t n_slots, bool by_rank, uint64_t first_pixel, float *out,
                                hipStream_t stream) {
    if (!n_slots) return hipSuccess;
    if (F.spp <= kPixelSamplesLdsSpp) {
        const size_t lds = (size_t)F.spp * 2 * 64 * sizeof(float);
        hipError_t e = allow_lds(k_pixel_samples_lds, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_pixel_samples_lds, dim3((n_slots + 63) / 64), dim3(64), lds, stream, F, n_slots,
                           by_rank ? 1 : 0, first_pixel, out);
    } else {
        hipLaunchKernelGGL(k_pixel_samples, dim3((n_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, F,
                           n_slots, by_rank ? 1 : 0, first_pixel, out);
    }
    return hipGetLastError();
}

hipError_t launch_generate_rays(const FrameDev &F, const rt_camera &cam, uint64_t first, uint32_t n,
                                rt_ray *rays, hipStream_t stream) {
    hipLaunchKernelGGL(k_generate_rays, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, F, cam, first, n,
                       rays);
    return hipGetLastError();
}

hipError_t launch_generate(const SceneDev &S, const FrameDev &F, const rt_camera &cam, const Queue &q, uint32_t *counts,
                           uint32_t n_local, uint32_t s0, uint32_t s_count, hipStream_t stream) {
    uint32_t padded = (n_local + 63) / 64 * 64;
    hipLaunchKernelGGL(k_generate, dim3((padded + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, S, F, cam, q,
                       counts, n_local, s0, s_count);
    return hipGetLastError();
}

hipError_t resident_workgroups(size_t lds_bytes, uint32_t *per_cu) {
    int n[7] = {0, 0, 0, 0, 0, 0, 0};
    hipError_t e = allow_lds(k_bounce<false, false, false>, lds_bytes);
    if (e == hipSuccess) e = allow_lds(k_trace<false, false, true>, lds_bytes);
    if (e == hipSuccess) e = allow_lds(k_intersect<false, false, true>, lds_bytes);
    if (e == hipSuccess) e = allow_lds(k_occluded<false, true>, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[4], k_trace<false, false, true>, kBlock, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[5], k_intersect<false, false, true>, kBlock, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[6], k_occluded<false, true>, kBlock, lds_bytes);
    if (e == hipSuccess) e = allow_lds(k_bounce<true, false, false>, lds_bytes);
    if (e == hipSuccess) e = allow_lds(k_intersect<false, false>, lds_bytes);
    if (e == hipSuccess) e = allow_lds(k_occluded<false>, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[0], k_bounce<false, false, false>, kBlock, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[1], k_bounce<true, false, false>, kBlock, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[2], k_intersect<false, false>, kBlock, lds_bytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n[3], k_occluded<false>, kBlock, lds_bytes);
    if (e != hipSuccess) return e;
    int m = n[0];
    for (int k = 1; k < 7; k++) m = n[k] < m ? n[k] : m;
    *per_cu = m > 0 ? (uint32_t)m : 1u;
    return hipSuccess;
}

uint32_t bounce_waves(const LaunchShape &L) { return L.grid * ((L.lds_nodes ? 1024u : (uint32_t)kBlock) / 64u); }

hipError_t launch_bounce(const SceneDev &S, const LaunchShape &L, const Queue &qin, const Queue &qout,
                         uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed, float4 *radiance,
                         uint8_t *scratch, unsigned long long *counters, hipStream_t stream,
                         const ShadowQueue *shadow, uint32_t *shadow_counts) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    const ShadowQueue sq = shadow ? *shadow : ShadowQueue{nullptr, nullptr, nullptr, 0u};
#define RT_BOUNCE(C, N, E)                                                                              \
    RT_LAUNCH((k_bounce<C, N, E>), L.grid, threads, S, qin, qout, counts, bounce, last, seed, radiance, \
              scratch, counters, sq, shadow_counts)
    if (shadow) {
        if (L.lds_nodes) {
            if (counters) RT_BOUNCE(true, true, true);
            else RT_BOUNCE(false, true, true);
        } else {
            if (counters) RT_BOUNCE(true, false, true);
            else RT_BOUNCE(false, false, true);
        }
    } else {
        if (L.lds_nodes) {
            if (counters) RT_BOUNCE(true, true, false);
            else RT_BOUNCE(false, true, false);
        } else {
            if (counters) RT_BOUNCE(true, false, false);
            else RT_BOUNCE(false, false, false);
        }
    }
#undef RT_BOUNCE
    return hipGetLastError();
}

hipError_t launch_trace(const SceneDev &S, const LaunchShape &L, const Queue &qin, const HitQueue &hits,
                        uint32_t *counts, uint32_t bounce, unsigned long long *counters, hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) {
        if (counters) RT_LAUNCH((k_trace<true, true>), L.grid, threads, S, qin, hits, counts, bounce, counters);
        else RT_LAUNCH((k_trace<false, true>), L.grid, threads, S, qin, hits, counts, bounce, counters);
    } else {
        if (counters) RT_LAUNCH((k_trace<true, false>), L.grid, threads, S, qin, hits, counts, bounce, counters);
        else if (L.quads == 2 || (L.quads == 1 && bounce >= kQuadBounce)) RT_LAUNCH((k_trace<false, false, true>), L.grid, threads, S, qin, hits, counts, bounce, counters);
        else RT_LAUNCH((k_trace<false, false>), L.grid, threads, S, qin, hits, counts, bounce, counters);
    }
    return hipGetLastError();
}

hipError_t launch_shade(const SceneDev &S, const LaunchShape &L, const Queue &qin, const Queue &qout,
                        const HitQueue &hits, uint32_t *counts, uint32_t bounce, uint32_t last, uint64_t seed,
                        float4 *radiance, hipStream_t stream, const ShadowQueue *shadow, uint32_t *shadow_counts) {
    // every workgroup resident at once (each wave strides over its share of the blocks), and a wave
    // count that is a multiple of kShards
    static int per_cu[2] = {0, 0}, cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[0], k_shade<false>, kBlock, 0);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[1], k_shade<true>, kBlock, 0);
        if (e != hipSuccess) return e;
        cus = prop.multiProcessorCount;
    }
    (void)L;
    uint32_t blocks = (uint32_t)cus * (uint32_t)(per_cu[shadow ? 1 : 0] > 0 ? per_cu[shadow ? 1 : 0] : 1);
    blocks = blocks / 8u * 8u;
    if (blocks == 0) blocks = 8;
    const ShadowQueue sq = shadow ? *shadow : ShadowQueue{nullptr, nullptr, nullptr, 0u};
    if (shadow)
        hipLaunchKernelGGL(k_shade<true>, dim3(blocks), dim3(kBlock), 0, stream, S, qin, qout, hits, counts, bounce,
                           last, seed, radiance, sq, shadow_counts);
    else
        hipLaunchKernelGGL(k_shade<false>, dim3(blocks), dim3(kBlock), 0, stream, S, qin, qout, hits, counts, bounce,
                           last, seed, radiance, sq, shadow_counts);
    return hipGetLastError();
}

hipError_t launch_shadow(const SceneDev &S, const LaunchShape &L, const ShadowQueue &q, uint32_t *shadow_counts,
                         uint32_t bounce, float4 *radiance, hipStream_t stream) {
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) RT_LAUNCH((k_shadow<true>), L.grid, threads, S, q, shadow_counts, bounce, radiance);
    else if (L.quads == 2 || (L.quads == 1 && bounce >= kQuadBounce)) RT_LAUNCH((k_shadow<false, true>), L.grid, threads, S, q, shadow_counts, bounce, radiance);
    else RT_LAUNCH((k_shadow<false>), L.grid, threads, S, q, shadow_counts, bounce, radiance);
    return hipGetLastError();
}

hipError_t launch_bounce_loop(const SceneDev &S, const LaunchShape &L, const Queue &qin, uint32_t *counts,
                              uint32_t bounce, uint32_t last, uint64_t seed, float4 *radiance, uint8_t *scratch,
                              unsigned long long *counters, hipStream_t stream, bool steal, uint32_t turn_limit) {
    const uint32_t st = steal ? 1u : 0u;
    const uint32_t threads = L.lds_nodes ? 1024 : kBlock;
    const size_t lds = lds_bytes(L.lds_nodes, S.n_pairs, S.stack_lds);
    if (L.lds_nodes) {
        if (counters)
            RT_LAUNCH((k_bounce_loop<true, true>), L.grid, threads, S, qin, counts, bounce, last, seed, radiance, scratch,
                      counters, st, turn_limit);
        else
            RT_LAUNCH((k_bounce_loop<false, true>), L.grid, threads, S, qin, counts, bounce, last, seed, radiance,
                      scratch, counters, st, turn_limit);
    } else {
        if (counters)
            RT_LAUNCH((k_bounce_loop<true, false>), L.grid, threads, S, qin, counts, bounce, last, seed, radiance,
                      scratch, counters, st, turn_limit);
        else
            RT_LAUNCH((k_bounce_loop<false, false>), L.grid, threads, S, qin, counts, bounce, last, seed, radiance,
                      scratch, counters, st, turn_limit);
    }
    return hipGetLastError();
}

hipError_t launch_film(const FrameDev &F, const FilterDev &T, const float4 *radiance, uint32_t s0,
                       uint32_t s_count, float4 *rgba, hipStream_t stream) {
    const size_t n = (size_t)F.width * F.height;
    hipLaunchKernelGGL(k_film, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, F, T, radiance,
                       s0, s_count, rgba);
    return hipGetLastError();
}

hipError_t launch_film_resolve(float4 *rgba, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_film_resolve, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, rgba, n);
    return hipGetLastError();
}

hipError_t launch_resolve(const FrameDev &F, const float4 *radiance, uint32_t n_ranks, uint32_t s_count,
                          int finish, float4 *rgba, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3((n_ranks + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, F, radiance,
                       n_ranks, s_count, finish, rgba);
    return hipGetLastError();
}

} // namespace rt
