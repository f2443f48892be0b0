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
#include <stdint.h>

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

// A tree of the library's own for the fast mode, over the same leaves: top-down, every range split where the
// surface-area cost  area(left box) * n_left + area(right box) * n_right  is smallest over all three axes and all
// positions (primitives ordered by box centre along the axis, ties by index), one primitive per leaf.  The reference's
// walk tests the boxes of both children of every node a ray enters and never culls, so the number of boxes a ray
// tests is exactly what this cost estimates; its own build (above) looks at one axis and, in ranges of more than a
// thousand primitives, only at positions near the median.  Leaf boxes are the same exact bounds, so -- with the
// reference's tie-break ranks carried over -- every ray has the same closest hit on both trees.  A branch deeper than
// max_height levels is finished with median splits, which bounds the height for any input.
Bvh build_bvh_sah(const std::vector<PrimBounds>& prims, int max_height = 32);

// Position of every primitive's leaf in the test order of a stack walk that expands nodes depth first, left first, and
// tests a node's left child before its right one (reference: src/gpu/GPUBVH.cu:38-99): 2 * (expansion order of the
// parent) + side.  Among hits at equal distance the reference keeps the one it tested first, i.e. the smallest rank.
std::vector<uint32_t> test_ranks(const Bvh& bvh, size_t n_prims);

}  // namespace cpt
