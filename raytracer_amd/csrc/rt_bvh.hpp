// rt_bvh.hpp -- host-side BVH construction and flattening
// (BVH::new / construct / flatten_tree, shapes/src/bvh.rs:108-151, 661-691).
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

#include "rt355.h"

namespace rt {

struct Aabb {
    float lo[3], hi[3];
};

// Bounds3::ZERO, geometry/src/bounds3.rs:23-26
inline Aabb empty_box() {
    const float m = 3.402823466e+38f;
    return Aabb{{m, m, m}, {-m, -m, -m}};
}

struct BuiltTree {
    std::vector<rt_linear_node> nodes; // pre-order, first child = index + 1
    std::vector<uint32_t> order;       // leaf order -> caller's primitive index
    uint32_t max_depth = 0;            // stack entries the traversal can need, + 1
};

// Builds the tree over `bounds` (one box per primitive, caller order).
// Returns RT_OK, RT_ERR_LIMIT (a leaf above 65535 primitives) or RT_ERR_DEPTH.
int build_tree(const std::vector<Aabb> &bounds, uint32_t max_prims_in_node, uint32_t split_method,
               BuiltTree &out, std::string &err);

uint32_t tree_depth(const std::vector<rt_linear_node> &nodes);

// Host threads the builders use: the machine's, capped by OMP_NUM_THREADS and 64, or
// RT355_BUILD_THREADS when set.
unsigned build_threads();

// The part of an HLBVH above its treelets (build_upper_sah, bvh.rs:541-658) for treelets
// built elsewhere.  In: box and node count of every treelet, in Morton order.  Out: the
// interior nodes above the treelets with their indices in the final pre-order array, and
// base[t] = index of treelet t's root there.  Returns the size of the final array.
struct UpperNode {
    uint32_t index;
    rt_linear_node node;
};
uint32_t hlbvh_upper(const std::vector<Aabb> &boxes, const std::vector<uint32_t> &sizes,
                     std::vector<UpperNode> &upper, std::vector<uint32_t> &base);

// HLBVH (hlbvh_build, bvh.rs:385-449) with Morton codes, the radix sort and the treelets
// (emit_lvbh, bvh.rs:452-538) computed on HIP device `device`; same tree as build_tree with
// RT_SPLIT_HLBVH, node for node.  Defined in rt_build_gpu.hip.
int build_tree_hlbvh_device(const std::vector<Aabb> &bounds, uint32_t max_prims_in_node, int device,
                            BuiltTree &out, std::string &err);

} // namespace rt
