// Reader for the reference's ".ray" scene description (the "SBT-raytracer 1.0" text format).
//
// The GPU path of the reference renders triangle meshes only (reference: README.md:21,
// src/gpu/GPUScene.cuh:36-49), so this reader keeps every polymesh/trimesh, the camera and the
// material state, accepts the rest of the grammar (lights, analytic primitives, groups,
// transforms) and reports what it dropped.  Transforms are applied to the vertices here, so
// transformed meshes render where the scene file puts them.
//
// Grammar reference: src/parser/Parser.cpp:27-155 (scene, camera), :157-361 (groups and
// transforms), :551-717 (trimesh, faces), :986-1070 (material); lexical rules:
// src/parser/Tokenizer.cpp:131-237; reserved words: src/parser/Token.cpp:113-178.
#pragma once
#include <string>
#include <vector>

#include "dvec.h"

namespace cpt {

struct RayMaterial {  // defaults: src/scene/material.h:164-174
    dvec3 ke{0, 0, 0}, ka{0, 0, 0}, ks{0, 0, 0}, kd{0, 0, 0}, kr{0, 0, 0}, kt{0, 0, 0};
    dvec3 shininess{0, 0, 0}, index{1, 1, 1};
};

struct RayMesh {
    std::string name;
    RayMaterial material;
    std::vector<dvec3> vertices;  // world space
    std::vector<dvec3> normals;   // empty, or one per vertex (as given or generated)
    std::vector<int> faces;       // 3 indices per non-degenerate triangle, file order
};

struct RayCamera {  // src/scene/camera.cpp:9-100
    dvec3 eye{0, 0, 0}, look{0, 0, -1}, u{1, 0, 0}, v{0, 1, 0};
    double aspect = 1, normalized_height = 1;
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // m[column][row]
};

struct RayScene {
    RayCamera camera;
    std::vector<RayMesh> meshes;
    std::vector<std::string> warnings;
};

// Throws std::runtime_error with "line N: message" on malformed input.
RayScene read_ray_text(const std::string& text);
RayScene read_ray_file(const std::string& path);

}  // namespace cpt
