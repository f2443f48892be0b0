// Scene file -> the flat records the GPU library takes (include/cpt_amd.h).
//
// Replaces the reference's object-by-object managed-memory mirror of the host scene graph
// (reference: src/gpu/GPUScene.cuh:25-57, src/gpu/GPUTrimesh.cuh:129-168, src/gpu/GPUBVH.cu:9-31):
// one array per record kind, triangles in BVH-input order, internal nodes in expansion order with
// both child boxes inline, one material per mesh instead of one per face.
#pragma once
#include <string>
#include <vector>

#include "bvh_build.h"
#include "cpt_amd.h"
#include "ray_file.h"

namespace cpt {

struct FlatScene {
    std::vector<cpt_node> nodes;
    std::vector<cpt_tri> tris;
    std::vector<cpt_tri_normals> tri_normals;
    std::vector<cpt_material> materials;
    cpt_camera camera{};
    double aspect = 1.0;
    Bvh bvh;  // canonical tree, kept for inspection and tests
    std::vector<std::string> warnings;
    // The fast mode's tree over the same triangles (build_bvh_sah), built on first use.  The triangle records carry
    // the reference tree's tie-break ranks, so a walk of either tree returns the same hit for every ray.
    std::vector<PrimBounds> bounds;
    Bvh fast_bvh;
    std::vector<cpt_node> fast_nodes;

    cpt_scene_desc describe() const;
    cpt_scene_desc describe_fast();
};

// build_tree = false leaves `bvh` and `nodes` empty, for callers that have the GPU build the tree (cpt_build_scene).
FlatScene flatten_scene(const RayScene& scene, bool build_tree = true);
FlatScene load_flat_scene(const std::string& ray_path, bool build_tree = true);

}  // namespace cpt
