#include "flatten.h"

#include <cstring>
#include <stdexcept>

namespace cpt {
namespace {

inline void put(double* dst, dvec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
inline void put_box(double* dst, const BvhNode& n) { put(dst, n.lo); put(dst + 3, n.hi); }

cpt_material flat_material(const RayMaterial& m) {
    cpt_material o;
    std::memset(&o, 0, sizeof o);
    put(o.ke, m.ke); put(o.kd, m.kd); put(o.ks, m.ks); put(o.kr, m.kr); put(o.kt, m.kt);
    o.index = (0.299 * m.index.x) + (0.587 * m.index.y) + (0.114 * m.index.z);  // src/gpu/GPUMaterial.cuh:21
    // src/gpu/GPUMaterial.cuh:33-38
    bool refl = !(length(m.kr) == 0), trans = !(length(m.kt) == 0);
    bool spec = refl || !(length(m.ks) == 0), diff = !(length(m.kd) == 0);
    o.flags = (diff ? CPT_MAT_DIFF : 0u) | (spec ? CPT_MAT_SPEC : 0u) | (trans ? CPT_MAT_TRANS : 0u) |
              ((refl || trans) ? CPT_MAT_RECUR : 0u);
    return o;
}

// Emit internal-node records level by level (top of the tree first), so that a prefix of the array
// is the part of the tree every ray walks through.  Parents always precede their children.
void emit_records(const Bvh& bvh, std::vector<cpt_node>& out) {
    std::vector<int> record_of(bvh.nodes.size(), -1), order;
    if (bvh.nodes.empty() || bvh.nodes[0].prim >= 0) return;
    order.push_back(0);
    for (size_t q = 0; q < order.size(); ++q) {
        const BvhNode& n = bvh.nodes[order[q]];
        record_of[order[q]] = (int)q;
        if (bvh.nodes[n.left].prim < 0) order.push_back(n.left);
        if (bvh.nodes[n.right].prim < 0) order.push_back(n.right);
    }
    for (size_t q = 0; q < order.size(); ++q) record_of[order[q]] = (int)q;
    out.resize(order.size());
    for (size_t q = 0; q < order.size(); ++q) {
        const BvhNode& n = bvh.nodes[order[q]];
        cpt_node& r = out[q];
        std::memset(&r, 0, sizeof r);
        put_box(r.lbox, bvh.nodes[n.left]);
        put_box(r.rbox, bvh.nodes[n.right]);
        r.lref = bvh.nodes[n.left].prim >= 0 ? ~bvh.nodes[n.left].prim : record_of[n.left];
        r.rref = bvh.nodes[n.right].prim >= 0 ? ~bvh.nodes[n.right].prim : record_of[n.right];
        r.lid = n.left;
        r.rid = n.right;
        r.self_id = order[q];
    }
}

}  // namespace

cpt_scene_desc FlatScene::describe_fast() {
    if (fast_bvh.nodes.empty() && !bounds.empty()) {
        fast_bvh = build_bvh_sah(bounds);
        emit_records(fast_bvh, fast_nodes);
    }
    cpt_scene_desc d = describe();
    d.n_nodes = (int)fast_nodes.size();
    d.n_tree_nodes = (int)fast_bvh.nodes.size();
    d.tree_height = fast_bvh.height;
    d.nodes = fast_nodes.data();
    d.flags = CPT_SCENE_KEEP_RANKS;
    return d;
}

cpt_scene_desc FlatScene::describe() const {
    cpt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_nodes = (int)nodes.size();
    d.n_tris = (int)tris.size();
    d.n_materials = (int)materials.size();
    d.n_tree_nodes = (int)bvh.nodes.size();
    d.tree_height = bvh.height;
    d.nodes = nodes.data();
    d.tris = tris.data();
    d.tri_normals = tri_normals.data();
    d.materials = materials.data();
    d.camera = camera;
    return d;
}

FlatScene flatten_scene(const RayScene& scene, bool build_tree) {
    FlatScene f;
    f.warnings = scene.warnings;
    put(f.camera.eye, scene.camera.eye);
    put(f.camera.look, scene.camera.look);
    put(f.camera.u, scene.camera.u);
    put(f.camera.v, scene.camera.v);
    f.aspect = scene.camera.aspect;

    // triangles in the order the reference hands them to its BVH build: scene objects in file
    // order, faces in face order (src/scene/bvh.cpp:108-113, src/SceneObjects/trimesh.cpp:201-208)
    std::vector<PrimBounds> bounds;
    for (size_t mi = 0; mi < scene.meshes.size(); ++mi) {
        const RayMesh& me = scene.meshes[mi];
        cpt_material fm = flat_material(me.material);
        f.materials.push_back(fm);
        const bool vn = !me.vertices.empty() && me.normals.size() == me.vertices.size();  // src/gpu/GPUTrimesh.cuh:149,165
        for (size_t fi = 0; fi + 2 < me.faces.size(); fi += 3) {
            dvec3 a = me.vertices[me.faces[fi]], b = me.vertices[me.faces[fi + 1]], c = me.vertices[me.faces[fi + 2]];
            cpt_tri t;
            std::memset(&t, 0, sizeof t);
            put(t.a, a); put(t.b, b); put(t.c, c);
            put(t.n, normalize(cross(b - a, c - a)));  // src/gpu/GPUTrimesh.cuh:47-49
            t.material = (int)mi;
            t.flags = ((fm.flags & CPT_MAT_RECUR) ? CPT_TRI_RECUR : 0u) | (vn ? CPT_TRI_VERTEX_NORMALS : 0u);
            f.tris.push_back(t);
            cpt_tri_normals tn;
            std::memset(&tn, 0, sizeof tn);
            if (vn) { put(tn.na, me.normals[me.faces[fi]]); put(tn.nb, me.normals[me.faces[fi + 1]]); put(tn.nc, me.normals[me.faces[fi + 2]]); }
            f.tri_normals.push_back(tn);
            PrimBounds pb;  // src/SceneObjects/trimesh.h:148-161 and trimesh.cpp:197-199
            auto mx = [](double x, double y) { return (x < y) ? y : x; };
            auto mn = [](double x, double y) { return (y < x) ? y : x; };
            dvec3 hi = dv(mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z)), lo = dv(mn(a.x, b.x), mn(a.y, b.y), mn(a.z, b.z));
            pb.hi = dv(mx(c.x, hi.x), mx(c.y, hi.y), mx(c.z, hi.z));
            pb.lo = dv(mn(c.x, lo.x), mn(c.y, lo.y), mn(c.z, lo.z));
            dvec3 sum = (a + b) + c;
            pb.center = dv(sum.x / 3.0, sum.y / 3.0, sum.z / 3.0);
            bounds.push_back(pb);
        }
    }
    if (f.tris.empty()) throw std::runtime_error("scene has no polymesh triangles to render");
    if (build_tree) {
        f.bvh = build_bvh(bounds);
        emit_records(f.bvh, f.nodes);
        // what cpt_upload_scene derives from these records anyway; filled in here so that the same triangle records can
        // go with another tree (describe_fast) and still break ties the reference's way
        const std::vector<uint32_t> rank = test_ranks(f.bvh, f.tris.size());
        for (size_t i = 0; i < f.tris.size(); ++i) f.tris[i].test_rank = rank[i];
        f.bounds = std::move(bounds);
    }
    return f;
}

FlatScene load_flat_scene(const std::string& ray_path, bool build_tree) { return flatten_scene(read_ray_file(ray_path), build_tree); }

}  // namespace cpt
