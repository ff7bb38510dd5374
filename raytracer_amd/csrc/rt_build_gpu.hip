// rt_build_gpu.hip -- HLBVH construction on the GPU (hlbvh_build, shapes/src/bvh.rs:385-449):
// Morton codes (:396-406, morton_encode3 :903-919), the radix sort (:923-966: five stable
// passes of six bits, least significant first -- 64 buckets, one per lane of a wave, see
// k_sort_count / k_sort_scatter), and the treelets (emit_lvbh :452-538).  Only the small SAH tree above the at most 4096
// treelets (build_upper_sah :541-658) runs on the host.  The output is the host builder's
// tree node for node: boxes are unions (min / max, exact in any order), splits are found
// with the same binary search, and node numbers are those of the pre-order flattening
// (flatten_tree :661-691), assigned here from subtree sizes.
//
// emit_lvbh is a recursion; on the GPU its call tree is built breadth first.  A node is a
// range [lo, hi) of the sorted primitives and the bit to test next; one thread per node of
// the current level finds the split and appends two children (consecutive numbers from an
// atomic counter, so a level is a contiguous block of nodes).  Then one pass per level
// upwards computes boxes and subtree sizes, one pass per level downwards the pre-order
// indices, and a last pass writes the 32-byte linear nodes.
#include <hip/hip_runtime.h>
#include <string.h>

#include <cstring>

#include <algorithm>
#include <unordered_map>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "rt_bvh.hpp"

namespace rt {
namespace {

struct GNode {
    uint32_t lo, hi;   // range of sorted primitives
    int32_t bit;       // in: first bit to test; out: the bit this node splits at
    int32_t child[2];  // -1: leaf
    uint32_t treelet;
    uint32_t size;     // nodes in the subtree
    uint32_t index;    // pre-order index inside the treelet
    uint32_t height;   // levels below this node (a leaf: 0)
    Aabb box;
};

constexpr int kThreads = 256;

__device__ inline uint32_t spread3(uint32_t x) { // left_shift3, bvh.rs:903-912
    if (x == (1u << 10)) x -= 1;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

// Floats as unsigned integers of the same order, so that atomicMin / atomicMax on the integers
// give the smallest / largest float (no NaNs here: primitive bounds are checked to be finite).
__device__ inline uint32_t ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float unordered(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}

// bounds of the centroids (bvh.rs:389-394): lo[3], hi[3] as ordered integers.  A thread walks
// the primitives with the grid's stride, a wave reduces its lanes' results with shuffles and
// adds six atomics (a few thousand waves: one wave per 64 primitives spent 12 ms on them).
__global__ void k_centroid_bounds(const Aabb *bounds, uint32_t n, uint32_t *lohi) {
    uint32_t lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const Aabb b = bounds[i];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t c = ordered(b.lo[k] * 0.5f + b.hi[k] * 0.5f); // BVHPrimitiveInfo::new, bvh.rs:694-703
            lo[k] = min(lo[k], c);
            hi[k] = max(hi[k], c);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int s = 32; s > 0; s >>= 1) {
            lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], s, 64));
            hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], s, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&lohi[k], lo[k]);
            atomicMax(&lohi[3 + k], hi[k]);
        }
    }
}

// treelets are maximal runs of sorted codes with equal top 12 bits (bvh.rs:414-431):
// first[v] = where the run with top bits v starts (untouched when there is none)
__global__ void k_treelet_starts(const uint32_t *codes, uint32_t n, uint32_t *first) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = codes[i] >> 18;
    if (i == 0 || (codes[i - 1] >> 18) != v) first[v] = i;
}

// bvh.rs:396-406 with the scale applied before the cast (the text casts first)
__global__ void k_morton(const Aabb *bounds, uint32_t n, Aabb cb, uint32_t *codes, uint32_t *ids) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const Aabb b = bounds[i];
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float c = b.lo[k] * 0.5f + b.hi[k] * 0.5f; // BVHPrimitiveInfo::new, bvh.rs:694-703
        float o = c - cb.lo[k];                          // Bounds3::offset, bounds3.rs:144-157
        if (cb.hi[k] > cb.lo[k]) o /= cb.hi[k] - cb.lo[k];
        q[k] = (uint32_t)(o * 1024.0f);
    }
    codes[i] = (spread3(q[2]) << 2) | (spread3(q[1]) << 1) | spread3(q[0]);
    ids[i] = i;
}

// ---- radix_sort, bvh.rs:923-966: stable LSD passes of kRadixBits bits -------------------------
// A pass sorts (code, id) pairs by one 6-bit digit and keeps the order of equal digits.  The 64
// buckets of a pass are the 64 lanes of a wave: lane b keeps bucket b's counter.  A 256-thread
// workgroup owns a tile of 4096 consecutive pairs, each of its four waves 1024 of them, which it
// walks in 16 rounds of 64 consecutive pairs (lane l takes pair 64 r + l, so loads are
// coalesced and lane order is input order).  In a round six ballots (one per digit bit) give
// every lane two masks at once: the lanes whose digit equals its own (its rank among them is
// the count of lower lanes in that mask) and the lanes whose digit equals its lane number (the
// round's contribution to the bucket it keeps).  Pass = count per tile (and per bucket in all)
// -> exclusive scan over (bucket, tile), one workgroup per bucket -> scatter; the scatter
// repeats the count from registers to know where each wave of the tile starts inside every
// bucket.
constexpr int kRadixBits = 6, kSortPasses = 5; // bits_per_pass, n_passes, bvh.rs:924-926
constexpr int kSortWaves = kThreads / 64, kSortRounds = 16;
constexpr uint32_t kSortTile = kThreads * kSortRounds;

__device__ inline void digit_masks(uint32_t digit, bool valid, uint32_t lane, uint64_t &same, uint64_t &kept) {
    same = kept = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kRadixBits; b++) {
        const uint64_t set = __ballot(valid && ((digit >> b) & 1u));
        same &= ((digit >> b) & 1u) ? set : ~set;
        kept &= ((lane >> b) & 1u) ? set : ~set;
    }
}

// hist[bucket * n_tiles + tile] = pairs of the tile whose digit is `bucket`; totals[bucket] += the same
__global__ __launch_bounds__(kThreads) void k_sort_count(const uint32_t *codes, uint32_t n, int shift, uint32_t *hist,
                                                         uint32_t n_tiles, uint32_t *totals) {
    __shared__ uint32_t wave_count[kSortWaves][64];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t first = (size_t)blockIdx.x * kSortTile + (size_t)w * 64 * kSortRounds;
    uint32_t count = 0;
    for (int r = 0; r < kSortRounds; r++) {
        const size_t i = first + (size_t)r * 64 + lane;
        const bool valid = i < n;
        const uint32_t digit = valid ? (codes[i] >> shift) & 63u : 0u;
        uint64_t same, kept;
        digit_masks(digit, valid, lane, same, kept);
        count += (uint32_t)__popcll(kept);
    }
    wave_count[w][lane] = count;
    __syncthreads();
    if (w == 0) {
        uint32_t total = 0;
        for (int k = 0; k < kSortWaves; k++) total += wave_count[k][lane];
        hist[(size_t)lane * n_tiles + blockIdx.x] = total;
        if (total) atomicAdd(&totals[lane], total);
    }
}

// Workgroup b turns row b of hist (bucket b, all tiles) into where each tile's pairs of that
// bucket start: the pairs of lower buckets (from totals) plus the row's exclusive prefix sum.
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t *hist, uint32_t n_tiles, const uint32_t *totals) {
    __shared__ uint32_t part[1024];
    uint32_t *v = hist + (size_t)blockIdx.x * n_tiles;
    const uint32_t t = threadIdx.x, per = (n_tiles + 1023) / 1024;
    const uint32_t lo = min(t * per, n_tiles), hi = min(lo + per, n_tiles);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < 1024; step <<= 1) { // inclusive scan of the 1024 partial sums
        const uint32_t add = t >= step ? part[t - step] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t k = 0; k < blockIdx.x; k++) run += totals[k];
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t x = v[i];
        v[i] = run;
        run += x;
    }
}

__global__ __launch_bounds__(kThreads) void k_sort_scatter(const uint32_t *codes, const uint32_t *ids, uint32_t n, int shift,
                                                           const uint32_t *offsets, uint32_t n_tiles, uint32_t *codes_out,
                                                           uint32_t *ids_out) {
    __shared__ uint32_t wave_count[kSortWaves][64];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t first = (size_t)blockIdx.x * kSortTile + (size_t)w * 64 * kSortRounds;
    uint32_t code[kSortRounds], id[kSortRounds];
    uint32_t count = 0;
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const size_t i = first + (size_t)r * 64 + lane;
        const bool valid = i < n;
        code[r] = valid ? codes[i] : 0u;
        id[r] = valid ? ids[i] : 0u;
        uint64_t same, kept;
        digit_masks((code[r] >> shift) & 63u, valid, lane, same, kept);
        count += (uint32_t)__popcll(kept);
    }
    wave_count[w][lane] = count;
    __syncthreads();
    // where this wave's pairs of bucket `lane` go: after all pairs of lower buckets, of the bucket
    // in earlier tiles, and in earlier waves of this tile
    uint32_t next = offsets[(size_t)lane * n_tiles + blockIdx.x];
    for (uint32_t k = 0; k < w; k++) next += wave_count[k][lane];
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const size_t i = first + (size_t)r * 64 + lane;
        const bool valid = i < n;
        const uint32_t digit = (code[r] >> shift) & 63u;
        uint64_t same, kept;
        digit_masks(digit, valid, lane, same, kept);
        const uint32_t at = (uint32_t)__shfl((int)next, (int)digit, 64) + (uint32_t)__popcll(same & below);
        if (valid) {
            codes_out[at] = code[r];
            ids_out[at] = id[r];
        }
        next += (uint32_t)__popcll(kept);
    }
}

__global__ void k_roots(const uint32_t *starts, uint32_t n_treelets, GNode *nodes) {
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= n_treelets) return;
    GNode g;
    g.lo = starts[t];
    g.hi = starts[t + 1];
    g.bit = 29 - 12; // the top 12 bits are shared inside a treelet (bvh.rs:418)
    g.child[0] = g.child[1] = -1;
    g.treelet = t;
    g.size = 1;
    g.index = 0;
    g.height = 0;
    g.box = Aabb{{0, 0, 0}, {0, 0, 0}};
    nodes[t] = g;
}

// One step of emit_lvbh (bvh.rs:452-538) for every node of a level.
__global__ void k_split(GNode *nodes, uint32_t begin, uint32_t end, const uint32_t *codes, uint32_t max_prims,
                        uint32_t *counter) {
    const uint32_t id = begin + blockIdx.x * kThreads + threadIdx.x;
    if (id >= end) return;
    GNode g = nodes[id];
    const uint32_t n = g.hi - g.lo;
    const uint32_t *c = codes + g.lo;
    int bit = g.bit;
    for (;;) {
        if (bit == -1 || n < max_prims) { // leaf, bvh.rs:459-480
            nodes[id].child[0] = nodes[id].child[1] = -1;
            return;
        }
        const uint32_t mask = 1u << bit;
        if ((c[0] & mask) != (c[n - 1] & mask)) break;
        bit--; // every primitive on the same side of this plane, bvh.rs:485-494
    }
    const uint32_t mask = 1u << bit;
    uint32_t lo = 0, hi = n - 1; // bvh.rs:497-509
    while (lo + 1 != hi) {
        const uint32_t mid = (lo + hi) / 2;
        if ((c[lo] & mask) == (c[mid] & mask)) lo = mid; else hi = mid;
    }
    const uint32_t first = atomicAdd(counter, 2u);
    GNode a;
    a.bit = bit - 1;
    a.child[0] = a.child[1] = -1;
    a.treelet = g.treelet;
    a.size = 1;
    a.index = 0;
    a.height = 0;
    a.box = Aabb{{0, 0, 0}, {0, 0, 0}};
    a.lo = g.lo;
    a.hi = g.lo + hi;
    nodes[first] = a;
    a.lo = g.lo + hi;
    a.hi = g.hi;
    nodes[first + 1] = a;
    nodes[id].bit = bit;
    nodes[id].child[0] = (int32_t)first;
    nodes[id].child[1] = (int32_t)first + 1;
}

__device__ inline float pick_min(float a, float b) { return b < a ? b : a; }
__device__ inline float pick_max(float a, float b) { return b > a ? b : a; }

// boxes and subtree sizes, one level at a time from the deepest
__global__ void k_up(GNode *nodes, uint32_t begin, uint32_t end, const Aabb *bounds, const uint32_t *ids,
                     uint32_t *too_many) {
    const uint32_t id = begin + blockIdx.x * kThreads + threadIdx.x;
    if (id >= end) return;
    GNode g = nodes[id];
    const float m = 3.402823466e+38f;
    Aabb box{{m, m, m}, {-m, -m, -m}};
    uint32_t size = 1, height = 0;
    if (g.child[0] < 0) {
        if (g.hi - g.lo > 65535u) atomicAdd(too_many, 1u);
        for (uint32_t i = g.lo; i < g.hi; i++) {
            const Aabb b = bounds[ids[i]];
            for (int k = 0; k < 3; k++) {
                box.lo[k] = pick_min(box.lo[k], b.lo[k]);
                box.hi[k] = pick_max(box.hi[k], b.hi[k]);
            }
        }
    } else {
        const GNode &a = nodes[g.child[0]], &b = nodes[g.child[1]];
        for (int k = 0; k < 3; k++) {
            box.lo[k] = pick_min(a.box.lo[k], b.box.lo[k]);
            box.hi[k] = pick_max(a.box.hi[k], b.box.hi[k]);
        }
        size = 1 + a.size + b.size;
        height = 1 + max(a.height, b.height);
    }
    nodes[id].box = box;
    nodes[id].size = size;
    nodes[id].height = height;
}

// pre-order indices inside the treelet, one level at a time from the roots
__global__ void k_down(GNode *nodes, uint32_t begin, uint32_t end) {
    const uint32_t id = begin + blockIdx.x * kThreads + threadIdx.x;
    if (id >= end) return;
    const GNode g = nodes[id];
    if (g.child[0] < 0) return;
    nodes[g.child[0]].index = g.index + 1;
    nodes[g.child[1]].index = g.index + 1 + nodes[g.child[0]].size;
}

__global__ void k_emit(const GNode *nodes, uint32_t n_nodes, const uint32_t *base, rt_linear_node *out) {
    const uint32_t id = blockIdx.x * kThreads + threadIdx.x;
    if (id >= n_nodes) return;
    const GNode g = nodes[id];
    const uint32_t b = base[g.treelet];
    rt_linear_node ln;
    for (int k = 0; k < 3; k++) {
        ln.lo[k] = g.box.lo[k];
        ln.hi[k] = g.box.hi[k];
    }
    ln._pad = 0;
    if (g.child[0] < 0) {
        ln.child_offset = g.lo;
        ln.primitive_count = (uint16_t)(g.hi - g.lo);
        ln.axis = 0;
    } else {
        ln.child_offset = b + nodes[g.child[1]].index;
        ln.primitive_count = 0;
        ln.axis = (uint8_t)(g.bit % 3);
    }
    out[b + g.index] = ln;
}

struct DeviceMem { // frees what it allocated when the build ends, on every path
    std::vector<void *> blocks;
    ~DeviceMem() {
        for (void *p : blocks) (void)hipFree(p);
    }
    template <class T>
    hipError_t get(T **p, size_t count) {
        void *v = nullptr;
        hipError_t e = hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) blocks.push_back(v);
        *p = static_cast<T *>(v);
        return e;
    }
};

inline unsigned blocks_for(size_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

} // namespace

#define GPU_TRY(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return RT_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

int build_tree_hlbvh_device(const std::vector<Aabb> &bounds, uint32_t max_prims_in_node, int device,
                            BuiltTree &out, std::string &err) {
    out = BuiltTree();
    const size_t n64 = bounds.size();
    if (n64 == 0) return RT_OK;
    if (n64 >= (1ull << 31)) {
        err = "too many primitives for a device build";
        return RT_ERR_LIMIT;
    }
    const uint32_t n = (uint32_t)n64;
    uint32_t max_prims = max_prims_in_node ? max_prims_in_node : 4;
    if (max_prims > 255) max_prims = 255; // bvh.rs:114
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        err = "no such HIP device for the build";
        return RT_ERR_NO_DEVICE;
    }
    GPU_TRY(hipSetDevice(device));
    // RT355_BUILD_TIMES=1 prints where the build's time goes (stderr, milliseconds)
    const bool times = getenv("RT355_BUILD_TIMES") != nullptr;
    auto clock0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!times) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rt355 build: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - clock0).count());
        clock0 = now;
    };

    DeviceMem mem;
    Aabb *d_bounds;
    uint32_t *d_codes, *d_ids, *d_codes2, *d_ids2, *d_counter;
    GPU_TRY(mem.get(&d_bounds, n));
    GPU_TRY(mem.get(&d_codes, n));
    GPU_TRY(mem.get(&d_ids, n));
    GPU_TRY(mem.get(&d_codes2, n));
    GPU_TRY(mem.get(&d_ids2, n));
    GPU_TRY(mem.get(&d_counter, 2));
    mark("allocations");
    GPU_TRY(hipMemcpy(d_bounds, bounds.data(), n64 * sizeof(Aabb), hipMemcpyHostToDevice));
    mark("bounds to device");
    // bounds of the centroids (bvh.rs:389-394): min / max, exact in any order
    Aabb cb;
    {
        uint32_t *d_lohi, lohi[6] = {~0u, ~0u, ~0u, 0u, 0u, 0u};
        GPU_TRY(mem.get(&d_lohi, 6));
        GPU_TRY(hipMemcpy(d_lohi, lohi, sizeof lohi, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min(blocks_for(n), 1024u)), dim3(kThreads), 0, nullptr, d_bounds, n, d_lohi);
        GPU_TRY(hipMemcpy(lohi, d_lohi, sizeof lohi, hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; k++) {
            cb.lo[k] = unordered(lohi[k]);
            cb.hi[k] = unordered(lohi[3 + k]);
        }
    }
    mark("centroid bounds");
    hipLaunchKernelGGL(k_morton, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_bounds, n, cb, d_codes, d_ids);
    GPU_TRY(hipGetLastError());

    // radix_sort, bvh.rs:923-966: an odd number of passes leaves the result in the second pair of buffers
    {
        const uint32_t n_tiles = (uint32_t)((n64 + kSortTile - 1) / kSortTile);
        uint32_t *d_hist, *d_totals;
        GPU_TRY(mem.get(&d_hist, (size_t)64 * n_tiles));
        GPU_TRY(mem.get(&d_totals, (size_t)64 * kSortPasses));
        GPU_TRY(hipMemset(d_totals, 0, sizeof(uint32_t) * 64 * kSortPasses));
        uint32_t *in_c = d_codes, *in_i = d_ids, *out_c = d_codes2, *out_i = d_ids2;
        for (int pass = 0; pass < kSortPasses; pass++) {
            const int shift = pass * kRadixBits;
            hipLaunchKernelGGL(k_sort_count, dim3(n_tiles), dim3(kThreads), 0, nullptr, in_c, n, shift, d_hist, n_tiles,
                               d_totals + 64 * pass);
            hipLaunchKernelGGL(k_sort_scan, dim3(64), dim3(1024), 0, nullptr, d_hist, n_tiles, d_totals + 64 * pass);
            hipLaunchKernelGGL(k_sort_scatter, dim3(n_tiles), dim3(kThreads), 0, nullptr, in_c, in_i, n, shift, d_hist,
                               n_tiles, out_c, out_i);
            std::swap(in_c, out_c);
            std::swap(in_i, out_i);
        }
        GPU_TRY(hipGetLastError());
        static_assert(kSortPasses % 2 == 1, "the sorted pairs are read from d_codes2 / d_ids2 below");
    }

    mark("codes + sort");
    // treelets: maximal runs with equal top 12 bits (bvh.rs:414-431)
    out.order.resize(n);
    GPU_TRY(hipMemcpy(out.order.data(), d_ids2, n64 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> starts;
    {
        uint32_t *d_first;
        std::vector<uint32_t> first(4096);
        GPU_TRY(mem.get(&d_first, first.size()));
        GPU_TRY(hipMemset(d_first, 0xff, first.size() * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_treelet_starts, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_codes2, n, d_first);
        GPU_TRY(hipMemcpy(first.data(), d_first, first.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t f : first) // increasing top bits = increasing position in the sorted array
            if (f != ~0u) starts.push_back(f);
    }
    const uint32_t n_treelets = (uint32_t)starts.size();
    starts.push_back(n);

    mark("order to host; starts");
    uint32_t *d_starts, *d_base;
    GNode *d_nodes;
    const size_t capacity = 2 * n64 + n_treelets;
    GPU_TRY(mem.get(&d_starts, starts.size()));
    GPU_TRY(mem.get(&d_base, n_treelets));
    GPU_TRY(mem.get(&d_nodes, capacity));
    GPU_TRY(hipMemcpy(d_starts, starts.data(), starts.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_roots, dim3(blocks_for(n_treelets)), dim3(kThreads), 0, nullptr, d_starts, n_treelets, d_nodes);
    const uint32_t init[2] = {n_treelets, 0}; // [0] next free node, [1] leaves above 65535 primitives
    GPU_TRY(hipMemcpy(d_counter, init, sizeof init, hipMemcpyHostToDevice));

    // breadth first: level k is nodes [level[k], level[k + 1])
    std::vector<uint32_t> level{0, n_treelets};
    for (;;) {
        const uint32_t begin = level[level.size() - 2], end = level.back();
        hipLaunchKernelGGL(k_split, dim3(blocks_for(end - begin)), dim3(kThreads), 0, nullptr, d_nodes, begin, end,
                           d_codes2, max_prims, d_counter);
        uint32_t next = 0;
        GPU_TRY(hipMemcpy(&next, d_counter, sizeof next, hipMemcpyDeviceToHost));
        if (next == end) break;
        level.push_back(next);
    }
    const uint32_t n_nodes = level.back();
    mark("allocate nodes; split levels");
    for (size_t k = level.size() - 1; k-- > 0;)
        hipLaunchKernelGGL(k_up, dim3(blocks_for(level[k + 1] - level[k])), dim3(kThreads), 0, nullptr, d_nodes, level[k],
                           level[k + 1], d_bounds, d_ids2, d_counter + 1);
    for (size_t k = 0; k + 1 < level.size(); k++)
        hipLaunchKernelGGL(k_down, dim3(blocks_for(level[k + 1] - level[k])), dim3(kThreads), 0, nullptr, d_nodes,
                           level[k], level[k + 1]);
    GPU_TRY(hipGetLastError());
    uint32_t flags[2];
    GPU_TRY(hipMemcpy(flags, d_counter, sizeof flags, hipMemcpyDeviceToHost));
    if (flags[1]) {
        err = "leaf with more than 65535 primitives";
        return RT_ERR_LIMIT;
    }

    mark("up, down");
    // the tree above the treelets, on the host
    std::vector<GNode> roots(n_treelets);
    GPU_TRY(hipMemcpy(roots.data(), d_nodes, (size_t)n_treelets * sizeof(GNode), hipMemcpyDeviceToHost));
    std::vector<Aabb> boxes(n_treelets);
    std::vector<uint32_t> sizes(n_treelets), base;
    for (uint32_t t = 0; t < n_treelets; t++) {
        boxes[t] = roots[t].box;
        sizes[t] = roots[t].size;
    }
    std::vector<UpperNode> upper;
    const uint32_t total = hlbvh_upper(boxes, sizes, upper, base);

    mark("upper tree (host)");
    rt_linear_node *d_out;
    GPU_TRY(mem.get(&d_out, total));
    GPU_TRY(hipMemcpy(d_base, base.data(), (size_t)n_treelets * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_emit, dim3(blocks_for(n_nodes)), dim3(kThreads), 0, nullptr, d_nodes, n_nodes, d_base, d_out);
    GPU_TRY(hipGetLastError());
    mark("emit");
    out.nodes.resize(total);
    mark("resize host nodes");
    GPU_TRY(hipMemcpy(out.nodes.data(), d_out, (size_t)total * sizeof(rt_linear_node), hipMemcpyDeviceToHost));
    for (const UpperNode &u : upper) out.nodes[u.index] = u.node;
    mark("nodes to host");

    // Stack entries the traversal can need (tree_depth): the deepest leaf.  Inside a treelet that
    // is its root's height (from k_up); the treelet roots' own depths come from a walk over the
    // few nodes above them.
    {
        std::unordered_map<uint32_t, uint32_t> treelet_at, upper_at; // node index -> treelet / upper node
        for (uint32_t t = 0; t < n_treelets; t++) treelet_at[base[t]] = t;
        for (size_t k = 0; k < upper.size(); k++) upper_at[upper[k].index] = (uint32_t)k;
        uint32_t deepest = 0;
        std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};
        while (!todo.empty()) {
            const auto [idx, depth] = todo.back();
            todo.pop_back();
            const auto t = treelet_at.find(idx);
            if (t != treelet_at.end()) {
                deepest = std::max(deepest, depth + roots[t->second].height);
            } else {
                const rt_linear_node &u = upper[upper_at.at(idx)].node;
                todo.push_back({idx + 1, depth + 1});
                todo.push_back({u.child_offset, depth + 1});
            }
        }
        out.max_depth = deepest + 1;
    }
    mark("tree depth");
    if (out.max_depth > 64) { // nodes_to_visit = [0; 64], bvh.rs:753
        err = "tree needs " + std::to_string(out.max_depth) + " stack entries, the traversal stack holds 64";
        return RT_ERR_DEPTH;
    }
    return RT_OK;
}

} // namespace rt
