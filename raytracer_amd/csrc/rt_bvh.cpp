// rt_bvh.cpp -- host BVH builder of librt355: SAH / Middle / EqualCounts
// top-down builds, HLBVH (Morton + radix sort + treelets + upper SAH) and the
// pre-order flattening into 32-byte linear nodes.
//
// Follows shapes/src/bvh.rs (construct :124-151, recursive_build :154-238,
// middle :261-276, equal_counts :282-295, sah :300-382, hlbvh_build :385-449,
// emit_lvbh :452-538, build_upper_sah :541-658, flatten_tree :661-691,
// partition :881-900, morton_encode3 :903-919, radix_sort :923-966).  The
// places where the literal text cannot run are rewritten to the behaviour it
// transcribes; each is marked.
#include "rt_bvh.hpp"

#include <string.h>

#include <stdlib.h>

#include <stdio.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

namespace rt {
namespace {

struct PrimInfo { // BVHPrimitiveInfo, bvh.rs:42-51
    uint32_t prim;
    Aabb box;
    float centroid[3];
};

struct BuildNode { // BVHBuildNode, bvh.rs:54-66
    Aabb box;
    int32_t child[2] = {-1, -1};
    int axis = 0;
    uint32_t first = 0, count = 0;
};

inline float pick_min(float a, float b) { return b < a ? b : a; }
inline float pick_max(float a, float b) { return b > a ? b : a; }

Aabb merge(const Aabb &a, const Aabb &b) { // union_box, bounds3.rs:70-75
    Aabb r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = pick_min(a.lo[k], b.lo[k]);
        r.hi[k] = pick_max(a.hi[k], b.hi[k]);
    }
    return r;
}
Aabb merge_point(const Aabb &a, const float p[3]) { // union_point, bounds3.rs:62-67
    Aabb r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = pick_min(a.lo[k], p[k]);
        r.hi[k] = pick_max(a.hi[k], p[k]);
    }
    return r;
}
// 2(dx*dy + dx*dz + dy*dz); bounds3.rs:114-117 computes 2*(d.d) instead
float area(const Aabb &b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
int longest_axis(const Aabb &b) { // maximum_extent, bounds3.rs:126-135
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
}
float relative(const Aabb &b, const float p[3], int dim) { // offset, bounds3.rs:144-157
    float o = p[dim] - b.lo[dim];
    if (b.hi[dim] > b.lo[dim]) o /= b.hi[dim] - b.lo[dim];
    return o;
}
int bucket_index(const Aabb &cb, const float c[3], int dim) {
    // bvh.rs:325-329 multiplies after the integer cast; multiply first
    int b = (int)(12.0f * relative(cb, c, dim));
    return b >= 12 ? 11 : (b < 0 ? 0 : b);
}
void centre_of(const Aabb &b, float out[3]) {
    for (int k = 0; k < 3; k++) out[k] = (b.lo[k] + b.hi[k]) * 0.5f;
}

// partition(), bvh.rs:881-900: same scan and swap sequence
template <class T, class Pred>
uint32_t split_by(T *a, uint32_t len, Pred keep_left) {
    if (len == 0) return 0;
    uint32_t left = 0, right = len - 1;
    for (;;) {
        while (left < len && keep_left(a[left])) left++;
        while (right > 0 && !keep_left(a[right])) right--;
        if (left >= right) return left;
        std::swap(a[left], a[right]);
    }
}

class Builder {
  public:
    Builder(const std::vector<Aabb> &bounds, uint32_t max_prims, uint32_t method)
        : max_prims_(max_prims), method_(method) {
        info_.resize(bounds.size());
        for (uint32_t i = 0; i < bounds.size(); i++) { // BVHPrimitiveInfo::new, bvh.rs:694-703
            info_[i].prim = i;
            info_[i].box = bounds[i];
            for (int k = 0; k < 3; k++)
                info_[i].centroid[k] = bounds[i].lo[k] * 0.5f + bounds[i].hi[k] * 0.5f;
        }
        pool_.reserve(2 * bounds.size() + 1);
        order_.reserve(bounds.size());
    }

    int32_t run() {
        uint32_t n = (uint32_t)info_.size();
        if (method_ == RT_SPLIT_HLBVH) return hlbvh(n);
        int32_t root = parallel_top_down(n);
        order_.resize(n); // leaves cover info_ left to right: the final order of info_ is the leaf order
        for (uint32_t i = 0; i < n; i++) order_[i] = info_[i].prim;
        return root;
    }

    // build_upper_sah over nodes the caller put into pool_ (one per treelet)
    int32_t upper_of(std::vector<int32_t> &roots) { return upper(roots.data(), (uint32_t)roots.size()); }

    std::vector<BuildNode> pool_;
    std::vector<uint32_t> order_;

  private:
    // total order on (centroid[dim], primitive number): fixes what
    // select_nth_unstable_by (bvh.rs:288) leaves open
    static bool before(const PrimInfo &a, const PrimInfo &b, int dim) {
        if (a.centroid[dim] < b.centroid[dim]) return true;
        if (b.centroid[dim] < a.centroid[dim]) return false;
        return a.prim < b.prim;
    }

    // k-th smallest to position k: median-of-three quickselect, Hoare scan
    static void nth(PrimInfo *a, uint32_t n, uint32_t k, int dim) {
        uint32_t lo = 0, hi = n;
        while (hi - lo > 2) {
            uint32_t mid = lo + (hi - lo) / 2;
            if (before(a[mid], a[lo], dim)) std::swap(a[mid], a[lo]);
            if (before(a[hi - 1], a[lo], dim)) std::swap(a[hi - 1], a[lo]);
            if (before(a[hi - 1], a[mid], dim)) std::swap(a[hi - 1], a[mid]);
            const PrimInfo pivot = a[mid];
            uint32_t i = lo, j = hi - 1;
            for (;;) {
                while (before(a[i], pivot, dim)) i++;
                while (before(pivot, a[j], dim)) j--;
                if (i >= j) break;
                std::swap(a[i], a[j]);
                i++;
                j--;
            }
            if (k <= j) hi = j + 1; else lo = j + 1;
        }
        if (hi - lo == 2 && before(a[lo + 1], a[lo], dim)) std::swap(a[lo], a[lo + 1]);
    }

    static uint32_t halves(PrimInfo *a, uint32_t n, int dim) { // equal_counts, bvh.rs:282-295
        nth(a, n, n / 2, dim);
        return n / 2;
    }

    using Pool = std::vector<BuildNode>;

    static int32_t alloc(Pool &pool) {
        pool.emplace_back();
        return (int32_t)pool.size() - 1;
    }
    int32_t alloc() { return alloc(pool_); }

    // bvh.rs:241-258, range start..start+n; the leaf's primitives are info_[start..end)
    static int32_t leaf(Pool &pool, uint32_t start, uint32_t end, const Aabb &box) {
        int32_t id = alloc(pool);
        pool[id].first = start;
        pool[id].count = end - start;
        pool[id].box = box;
        return id;
    }

    static int32_t interior(Pool &pool, int axis, int32_t c0, int32_t c1) {
        int32_t id = alloc(pool);
        pool[id].child[0] = c0;
        pool[id].child[1] = c1;
        pool[id].axis = axis;
        pool[id].box = merge(pool[c0].box, pool[c1].box);
        return id;
    }
    int32_t interior(int axis, int32_t c0, int32_t c1) { return interior(pool_, axis, c0, c1); }

    // Subtrees of at most grain_ primitives found while the top of the tree is built are put
    // aside (a placeholder node holds their box) and built by worker threads into pools of
    // their own, which are then spliced in.  Subtrees share nothing but disjoint ranges of
    // info_, and node numbers are assigned later by the pre-order flattening, so the tree does
    // not depend on the number of threads.
    struct Task {
        uint32_t start, end;
        int32_t node;
    };

    int32_t parallel_top_down(uint32_t n) {
        const unsigned threads = build_threads();
        if (threads == 1 || n < (1u << 16)) return top_down(pool_, 0, n, nullptr);
        grain_ = std::max(4096u, n / (threads * 8));
        std::vector<Task> tasks;
        int32_t root = top_down(pool_, 0, n, &tasks);
        std::vector<Pool> pools(tasks.size());
        std::vector<int32_t> roots(tasks.size(), -1);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t t = next++; t < tasks.size(); t = next++)
                roots[t] = top_down(pools[t], tasks[t].start, tasks[t].end, nullptr);
        };
        std::vector<std::thread> pool_threads;
        for (unsigned t = 1; t < threads; t++) pool_threads.emplace_back(work);
        work();
        for (auto &t : pool_threads) t.join();
        for (size_t t = 0; t < tasks.size(); t++) {
            const int32_t base = (int32_t)pool_.size();
            for (BuildNode nd : pools[t]) {
                if (nd.child[0] >= 0) {
                    nd.child[0] += base;
                    nd.child[1] += base;
                }
                pool_.push_back(nd);
            }
            pool_[tasks[t].node] = pool_[base + roots[t]];
            Pool().swap(pools[t]);
        }
        return root;
    }

    // sah(), bvh.rs:300-382.  Returns the split position or -1 for a leaf.
    int64_t sah(PrimInfo *a, uint32_t n, const Aabb &cb, int dim, const Aabb &box) {
        if (n <= 4) return halves(a, n, dim);
        uint32_t count[12] = {0};
        Aabb bb[12];
        for (auto &b : bb) b = empty_box();
        for (uint32_t i = 0; i < n; i++) {
            int b = bucket_index(cb, a[i].centroid, dim);
            count[b]++;
            bb[b] = merge(bb[b], a[i].box);
        }
        float best_cost = 0.0f;
        int best = 0;
        for (int i = 0; i < 11; i++) { // both sides summed (bvh.rs:349 assigns); 11 candidate planes
            Aabb b0 = empty_box(), b1 = empty_box();
            uint32_t c0 = 0, c1 = 0;
            for (int j = 0; j <= i; j++) { b0 = merge(b0, bb[j]); c0 += count[j]; }
            for (int j = i + 1; j < 12; j++) { b1 = merge(b1, bb[j]); c1 += count[j]; }
            float cost = 0.125f + ((float)c0 * area(b0) + (float)c1 * area(b1)) / area(box);
            if (i == 0 || cost < best_cost) { best_cost = cost; best = i; }
        }
        if (n > max_prims_ || best_cost < (float)n) // leaf cost compared as float (bvh.rs:368-370)
            return split_by(a, n, [&](const PrimInfo &p) { return bucket_index(cb, p.centroid, dim) <= best; });
        return -1;
    }

    // recursive_build, bvh.rs:154-238
    int32_t top_down(Pool &pool, uint32_t start, uint32_t end, std::vector<Task> *defer) {
        PrimInfo *a = info_.data() + start;
        uint32_t n = end - start;
        Aabb box = empty_box();
        for (uint32_t i = 0; i < n; i++) box = merge(box, a[i].box);
        if (n == 1) return leaf(pool, start, end, box);
        if (defer && n <= grain_) {
            int32_t id = alloc(pool);
            pool[id].box = box;
            defer->push_back(Task{start, end, id});
            return id;
        }
        Aabb cb = empty_box();
        for (uint32_t i = 0; i < n; i++) cb = merge_point(cb, a[i].centroid);
        int dim = longest_axis(cb);
        if (cb.hi[dim] == cb.lo[dim]) return leaf(pool, start, end, box);
        int64_t mid;
        if (method_ == RT_SPLIT_MIDDLE) { // bvh.rs:261-276
            float pmid = (cb.lo[dim] + cb.hi[dim]) / 2.0f;
            mid = split_by(a, n, [&](const PrimInfo &p) { return p.centroid[dim] < pmid; });
            if (mid == 0 || mid == (int64_t)n) mid = halves(a, n, dim);
        } else if (method_ == RT_SPLIT_EQUAL_COUNTS) {
            mid = halves(a, n, dim);
        } else {
            mid = sah(a, n, cb, dim, box);
            if (mid < 0) return leaf(pool, start, end, box);
            if (mid == 0 || mid == (int64_t)n) mid = halves(a, n, dim);
        }
        int32_t c0 = top_down(pool, start, start + (uint32_t)mid, defer);
        int32_t c1 = top_down(pool, start + (uint32_t)mid, end, defer);
        return interior(pool, dim, c0, c1);
    }

    // ---- HLBVH ----
    struct Morton {
        uint32_t prim, code;
    };
    static uint32_t spread3(uint32_t x) { // left_shift3, bvh.rs:903-912
        if (x == (1u << 10)) x -= 1;
        x = (x | (x << 16)) & 0x030000ffu;
        x = (x | (x << 8)) & 0x0300f00fu;
        x = (x | (x << 4)) & 0x030c30c3u;
        x = (x | (x << 2)) & 0x09249249u;
        return x;
    }
    static void radix(std::vector<Morton> &v) { // radix_sort, bvh.rs:923-966 with real scratch space
        std::vector<Morton> tmp(v.size());
        Morton *in = v.data(), *out = tmp.data();
        const size_t n = v.size();
        for (int pass = 0; pass < 5; pass++) {
            const int low = pass * 6;
            uint32_t hist[64] = {0}, at[64];
            for (size_t i = 0; i < n; i++) hist[(in[i].code >> low) & 63u]++;
            at[0] = 0;
            for (int i = 1; i < 64; i++) at[i] = at[i - 1] + hist[i - 1];
            for (size_t i = 0; i < n; i++) out[at[(in[i].code >> low) & 63u]++] = in[i];
            std::swap(in, out);
        }
        if (in != v.data()) memcpy(v.data(), in, n * sizeof(Morton));
    }

    int32_t emit(const Morton *mp, uint32_t n, int bit) { // emit_lvbh, bvh.rs:452-538
        if (bit == -1 || n < max_prims_) {
            int32_t id = alloc();
            Aabb box = empty_box();
            pool_[id].first = (uint32_t)order_.size();
            pool_[id].count = n;
            for (uint32_t i = 0; i < n; i++) {
                order_.push_back(mp[i].prim);
                box = merge(box, info_[mp[i].prim].box);
            }
            pool_[id].box = box;
            return id;
        }
        const uint32_t mask = 1u << bit;
        if ((mp[0].code & mask) == (mp[n - 1].code & mask)) return emit(mp, n, bit - 1);
        uint32_t lo = 0, hi = n - 1;
        while (lo + 1 != hi) {
            uint32_t mid = (lo + hi) / 2;
            if ((mp[lo].code & mask) == (mp[mid].code & mask)) lo = mid; else hi = mid;
        }
        // a split at n-1 is legal (bvh.rs:512 asserts split < n-1)
        int32_t c0 = emit(mp, hi, bit - 1);
        int32_t c1 = emit(mp + hi, n - hi, bit - 1);
        return interior(bit % 3, c0, c1);
    }

    int32_t upper(int32_t *roots, uint32_t n) { // build_upper_sah, bvh.rs:541-658
        if (n == 1) return roots[0];
        Aabb box = empty_box(), cb = empty_box();
        for (uint32_t i = 0; i < n; i++) {
            float c[3];
            box = merge(box, pool_[roots[i]].box);
            centre_of(pool_[roots[i]].box, c);
            cb = merge_point(cb, c);
        }
        int dim = longest_axis(cb);
        uint32_t mid;
        if (cb.hi[dim] == cb.lo[dim]) {
            mid = n / 2; // bvh.rs:570 asserts here
        } else {
            uint32_t count[12] = {0};
            Aabb bb[12];
            for (auto &b : bb) b = empty_box();
            for (uint32_t i = 0; i < n; i++) {
                float c[3];
                centre_of(pool_[roots[i]].box, c);
                int b = bucket_index(cb, c, dim);
                count[b]++;
                bb[b] = merge(bb[b], pool_[roots[i]].box);
            }
            float best_cost = 0.0f;
            int best = 0;
            for (int i = 0; i < 11; i++) {
                Aabb b0 = empty_box(), b1 = empty_box();
                uint32_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) { b0 = merge(b0, bb[j]); c0 += count[j]; }
                for (int j = i + 1; j < 12; j++) { b1 = merge(b1, bb[j]); c1 += count[j]; }
                float cost = 0.125f + ((float)c0 * area(b0) + (float)c1 * area(b1)) / area(box);
                if (i == 0 || cost < best_cost) { best_cost = cost; best = i; }
            }
            mid = split_by(roots, n, [&](int32_t r) {
                float c[3];
                centre_of(pool_[r].box, c);
                return bucket_index(cb, c, dim) <= best;
            });
            if (mid == 0 || mid == n) mid = n / 2;
        }
        int32_t c0 = upper(roots, mid);
        int32_t c1 = upper(roots + mid, n - mid);
        return interior(dim, c0, c1);
    }

    int32_t hlbvh(uint32_t n) { // hlbvh_build, bvh.rs:385-449
        Aabb cb = empty_box();
        for (uint32_t i = 0; i < n; i++) cb = merge_point(cb, info_[i].centroid);
        std::vector<Morton> mp(n);
        for (uint32_t i = 0; i < n; i++) { // scale, then cast (bvh.rs:400-406 casts first)
            float ox = relative(cb, info_[i].centroid, 0) * 1024.0f;
            float oy = relative(cb, info_[i].centroid, 1) * 1024.0f;
            float oz = relative(cb, info_[i].centroid, 2) * 1024.0f;
            mp[i].prim = info_[i].prim;
            mp[i].code = (spread3((uint32_t)oz) << 2) | (spread3((uint32_t)oy) << 1) | spread3((uint32_t)ox);
        }
        radix(mp);
        std::vector<int32_t> roots;
        const uint32_t mask = 0x3ffc0000u; // top 12 of 30 bits
        for (uint32_t start = 0, end = 1; end <= n; end++) { // bvh.rs:414-431 with a bounded scan
            if (end == n || (mp[start].code & mask) != (mp[end].code & mask)) {
                roots.push_back(emit(mp.data() + start, end - start, 29 - 12));
                start = end;
            }
        }
        return upper(roots.data(), (uint32_t)roots.size());
    }

    std::vector<PrimInfo> info_;
    uint32_t max_prims_, method_;
    uint32_t grain_ = 0;
};

uint32_t flatten(const std::vector<BuildNode> &pool, int32_t id, std::vector<rt_linear_node> &out) {
    uint32_t idx = (uint32_t)out.size(); // flatten_tree, bvh.rs:661-691
    out.emplace_back();
    const BuildNode &n = pool[id];
    rt_linear_node ln;
    memset(&ln, 0, sizeof ln);
    memcpy(ln.lo, n.box.lo, sizeof ln.lo);
    memcpy(ln.hi, n.box.hi, sizeof ln.hi);
    if (n.child[0] >= 0) {
        ln.axis = (uint8_t)n.axis;
        out[idx] = ln;
        flatten(pool, n.child[0], out);
        out[idx].child_offset = flatten(pool, n.child[1], out);
    } else {
        ln.child_offset = n.first;
        ln.primitive_count = (uint16_t)n.count;
        out[idx] = ln;
    }
    return idx;
}

} // namespace

uint32_t hlbvh_upper(const std::vector<Aabb> &boxes, const std::vector<uint32_t> &sizes,
                     std::vector<UpperNode> &upper, std::vector<uint32_t> &base) {
    const uint32_t n = (uint32_t)boxes.size();
    upper.clear();
    base.assign(n, 0);
    if (n == 0) return 0;
    Builder b(std::vector<Aabb>(), 1, RT_SPLIT_HLBVH);
    std::vector<int32_t> roots(n);
    for (uint32_t t = 0; t < n; t++) { // a treelet stands in the pool as a leaf whose `first` is its number
        b.pool_.emplace_back();
        b.pool_.back().box = boxes[t];
        b.pool_.back().first = t;
        roots[t] = (int32_t)t;
    }
    const int32_t root = b.upper_of(roots);
    // pre-order numbering (flatten_tree, bvh.rs:661-691) with a treelet taking sizes[t] slots
    uint32_t next = 0;
    struct Walk {
        const std::vector<BuildNode> &pool;
        const std::vector<uint32_t> &sizes;
        std::vector<UpperNode> &upper;
        std::vector<uint32_t> &base;
        uint32_t &next;
        uint32_t go(int32_t id) {
            const BuildNode &nd = pool[id];
            const uint32_t idx = next;
            if (nd.child[0] < 0) {
                base[nd.first] = idx;
                next += sizes[nd.first];
                return idx;
            }
            next++;
            const size_t slot = upper.size();
            upper.emplace_back();
            rt_linear_node ln;
            memset(&ln, 0, sizeof ln);
            memcpy(ln.lo, nd.box.lo, sizeof ln.lo);
            memcpy(ln.hi, nd.box.hi, sizeof ln.hi);
            ln.axis = (uint8_t)nd.axis;
            go(nd.child[0]);
            ln.child_offset = go(nd.child[1]);
            upper[slot].index = idx;
            upper[slot].node = ln;
            return idx;
        }
    } walk{b.pool_, sizes, upper, base, next};
    walk.go(root);
    return next;
}

unsigned build_threads() {
    unsigned threads = std::thread::hardware_concurrency();
    if (const char *e = getenv("RT355_BUILD_THREADS")) threads = (unsigned)atoi(e);
    else if (const char *e2 = getenv("OMP_NUM_THREADS")) threads = std::min(threads, (unsigned)atoi(e2));
    return std::max(1u, std::min(threads, 64u));
}

uint32_t tree_depth(const std::vector<rt_linear_node> &nodes) {
    if (nodes.empty()) return 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};
    uint32_t deepest = 0;
    while (!todo.empty()) {
        auto [idx, depth] = todo.back();
        todo.pop_back();
        deepest = std::max(deepest, depth);
        if (nodes[idx].primitive_count == 0) { // either child may be visited with the other one stacked
            todo.push_back({idx + 1, depth + 1});
            todo.push_back({nodes[idx].child_offset, depth + 1});
        }
    }
    return deepest + 1;
}

int build_tree(const std::vector<Aabb> &bounds, uint32_t max_prims_in_node, uint32_t split_method,
               BuiltTree &out, std::string &err) {
    out = BuiltTree();
    if (bounds.empty()) return RT_OK;
    uint32_t max_prims = max_prims_in_node ? max_prims_in_node : 4;
    if (max_prims > 255) max_prims = 255; // bvh.rs:114
    const bool timing = getenv("RT355_BUILD_TIMING") != nullptr; // phase times on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();
    Builder b(bounds, max_prims, split_method);
    auto t1 = now();
    int32_t root = b.run();
    auto t2 = now();
    for (const BuildNode &n : b.pool_)
        if (n.child[0] < 0 && n.count > 65535) {
            err = "leaf with more than 65535 primitives";
            return RT_ERR_LIMIT;
        }
    out.nodes.reserve(b.pool_.size());
    flatten(b.pool_, root, out.nodes);
    out.order = std::move(b.order_);
    auto t3 = now();
    out.max_depth = tree_depth(out.nodes);
    if (timing)
        fprintf(stderr, "rt355 build: info %.0f ms, tree %.0f ms, flatten %.0f ms, depth %.0f ms\n", ms(t0, t1),
                ms(t1, t2), ms(t2, t3), ms(t3, now()));
    if (out.max_depth > 64) { // nodes_to_visit = [0; 64], bvh.rs:753
        err = "tree needs " + std::to_string(out.max_depth) + " stack entries, the traversal stack holds 64";
        return RT_ERR_DEPTH;
    }
    return RT_OK;
}

} // namespace rt
