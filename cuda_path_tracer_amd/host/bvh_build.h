// Host BVH build that reproduces the reference's tree exactly.
//
// The reference builds its hierarchy on the CPU before any rendering (reference:
// src/scene/bvh.cpp:101-220): recursive median-window SAH over one shared primitive array that is
// re-sorted in place, range by range, with std::sort.  Node visit sets and hit tie-breaks are only
// comparable with the reference if the tree is the same tree, so this build keeps the reference's
// decisions (axis choice, sort key, candidate window, cost expression, first strict minimum) and
// changes only how the candidate boxes are obtained: one prefix sweep and one suffix sweep per
// node instead of two full box unions per candidate.
#pragma once
#include <vector>

#include "dvec.h"

namespace cpt {

struct PrimBounds {
    dvec3 lo, hi;   // box of the triangle's three vertices
    dvec3 center;   // ((v0 + v1) + v2) / 3, the sort key
};

struct BvhNode {    // canonical order: preorder, root first, left subtree before right
    dvec3 lo, hi;
    int left = -1, right = -1;  // canonical ids, -1 for a leaf
    int prim = -1;              // primitive index for a leaf
};

struct Bvh {
    std::vector<BvhNode> nodes;
    int height = 0;  // levels, a single leaf has height 1
};

Bvh build_bvh(const std::vector<PrimBounds>& prims);

}  // namespace cpt
