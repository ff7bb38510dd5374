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

} // namespace rt
