// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// A plain, single-purpose restatement of the reference path tracer's algorithm, used as the
// checker for the HIP path and as the host-CPU baseline.  Nothing that ships (the HIP library,
// the host scene library, bin/ray) may include, link or execute anything in this directory;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// Parity status: the reference itself cannot be built in this environment (it needs nvcc,
// cuRAND, Thrust, and GLM 0.9.6 which its build fetches with git clone), and it ships no
// known-answer tests.  The oracle is therefore pinned by (a) the reference's own scene files,
// (b) the mean luminance of the reference's golden renders (and, through the HIP path that is
// compared with this restatement path by path, the 64 x 64 block means of all fifteen of them),
// and (c) the published Philox known-answer vectors for the counter RNG.  Primary-ray hit ids / BVH visit sets are
// "parity unpinned" against a running reference: they are pinned to this restatement.
//
// Arithmetic: IEEE-754 binary64, built with -ffp-contract=off, operations in the order the
// reference source writes them, with GLM 0.9.6 evaluation order for dot/cross/normalize.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace orc {

struct V3 { double x, y, z; };

// --- GLM 0.9.6 evaluation order (glm/detail/func_geometric.inl) -----------------------------
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
double length(V3 v);
V3 normalize(V3 v);
inline double comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// reference: src/scene/material.h:160-174 (defaults), src/gpu/GPUMaterial.cuh:33-38 (flags)
struct Material {
    V3 ke{0, 0, 0}, ka{0, 0, 0}, ks{0, 0, 0}, kd{0, 0, 0}, kr{0, 0, 0}, kt{0, 0, 0};
    V3 shininess{0, 0, 0}, index{1, 1, 1};
    bool diff() const;
    bool refl() const;
    bool trans() const;
    bool recur() const { return refl() || trans(); }
    bool spec() const;
    double ior() const;  // GPUMaterial.cuh:21
};

struct Mesh {
    std::string name;
    std::vector<V3> verts, normals;
    std::vector<int> face_ids;  // 3 per kept (non-degenerate) face
    std::vector<V3> face_n;     // unit face normal per kept face
    Material mat;
    bool parsed_normals = false;
    bool use_vertex_normals() const { return normals.size() == verts.size() && !verts.empty(); }
};

struct Camera {  // reference: src/scene/camera.cpp:9-100
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // m[col][row], GLM column-major
    double normalizedHeight = 1, aspect = 1;
    V3 eye{0, 0, 0}, look{0, 0, -1}, u{1, 0, 0}, v{0, 1, 0};
    void update();
    void ray_through(double x, double y, V3& o, V3& d) const;
};

struct Prim {  // one kept triangle, in BVH-input order (scene order, then face order)
    int mesh, face;
    V3 a, b, c, n;
    V3 bmin, bmax;
    V3 center;
};

struct Node {  // pointer tree exactly as BVHTree::buildHelper creates it (bvh.cpp:178-220)
    V3 bmin, bmax;
    Node *left = nullptr, *right = nullptr;
    int prim = -1;
    int id = -1;  // canonical id: preorder index (root 0, left subtree, then right)
    bool leaf() const { return prim >= 0; }
};

struct Scene {
    Camera cam;
    std::vector<Mesh> meshes;
    std::vector<Prim> prims;  // fixed index = BVH-input order (before any sort)
    std::vector<int> order;   // the shared array the build sorts in place (prim indices)
    std::vector<Node*> nodes; // by canonical id
    Node* root = nullptr;
    int height = 0;
    ~Scene();
};

Scene* load_scene(const std::string& path, std::string& err);
void build_bvh(Scene& s);
bool replace_tree(Scene& s, int n, const double* boxes, const int* topo);

// --- GPU-semantics traversal (src/gpu/GPUBVH.cu:38-99) --------------------------------------
struct TraceOut {
    int prim = -1;
    double t = -1.0;
    V3 N{0, 0, 0};
    uint64_t visit_hash = 0;
    int node_steps = 0, box_tests = 0, tri_tests = 0, max_stack = 0;
};
uint64_t visit_mix(uint32_t node_id);
void trace_gpu(const Scene& s, V3 o, V3 d, TraceOut& out, uint8_t* visit_bitmap /*nullable*/);

// --- counter RNG ------------------------------------------------------------------------------
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double u01_open0(uint32_t lo, uint32_t hi);  // (0,1], 53 bits

struct RenderParams {
    int w = 0, h = 0, s0 = 0, s1 = 0, depth = 0;
    uint64_t seed = 1984;
    int dof = 0;
    double focal_distance = 8.5, aperture = 0.6;
    int rr = 0;
};
struct RenderStats {
    uint64_t paths = 0, segments = 0, hits = 0, node_steps = 0, tri_tests = 0;
};
// accum: w*h*3 doubles, bottom-up rows, += per-pixel sum of radiance over samples [s0,s1)
void render_gpu_semantics(const Scene& s, const RenderParams& p, double* accum, RenderStats* st);
// one path: (pixel i,j, sample) -> radiance
V3 path_radiance_gpu(const Scene& s, const RenderParams& p, int i, int j, int sample, RenderStats* st);

// --- the reference's CPU estimator (src/path_tracer.cu:27-238 + src/scene/bvh.cpp:223-297) ---
void render_cpu_reference(const Scene& s, int w, int h, int spp, int depth, int threads,
                          uint64_t seed, uint8_t* rgb, double* accum_mean /*nullable*/);

}  // namespace orc
