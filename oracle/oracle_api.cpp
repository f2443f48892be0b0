// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
// extern "C" surface so tests/ and bench.py can drive the oracle through ctypes.
#include "oracle.h"

#include <chrono>
#include <cstdio>
#include <cstring>

using namespace orc;

extern "C" {

void* orc_load(const char* path, char* err, int errlen) {
    std::string e;
    Scene* s = load_scene(path, e);
    if (!s && err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return s;
}
void orc_free(void* h) { delete (Scene*)h; }

// counts: [n_prims, n_nodes, n_meshes, height]
void orc_counts(void* h, int* out4) {
    Scene* s = (Scene*)h;
    out4[0] = (int)s->prims.size();
    out4[1] = (int)s->nodes.size();
    out4[2] = (int)s->meshes.size();
    out4[3] = s->height;
}

// Triangles in BVH-input order.  geom: n*12 doubles (a, b, c, unit normal); vn: n*9 doubles
// (per-vertex normals, zeros when the mesh has none); ids: n*3 ints (mesh, recur, has_vn)
void orc_get_prims(void* h, double* geom, double* vn, int* ids) {
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->prims.size(); ++i) {
        const Prim& p = s->prims[i];
        const V3 g[4] = {p.a, p.b, p.c, p.n};
        for (int k = 0; k < 4; ++k) { geom[12 * i + 3 * k] = g[k].x; geom[12 * i + 3 * k + 1] = g[k].y; geom[12 * i + 3 * k + 2] = g[k].z; }
        const Mesh& me = s->meshes[p.mesh];
        bool has = me.use_vertex_normals();
        for (int k = 0; k < 3; ++k) {
            V3 n = has ? me.normals[me.face_ids[3 * p.face + k]] : V3{0, 0, 0};
            vn[9 * i + 3 * k] = n.x; vn[9 * i + 3 * k + 1] = n.y; vn[9 * i + 3 * k + 2] = n.z;
        }
        ids[3 * i] = p.mesh;
        ids[3 * i + 1] = me.mat.recur() ? 1 : 0;
        ids[3 * i + 2] = has ? 1 : 0;
    }
}

// Per-mesh material: 16 doubles (ke, kd, ks, kr, kt, ior) + flags bit0 diff, bit1 spec, bit2 trans, bit3 recur
void orc_get_materials(void* h, double* vals, int* flags) {
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->meshes.size(); ++i) {
        const Material& m = s->meshes[i].mat;
        const V3 c[5] = {m.ke, m.kd, m.ks, m.kr, m.kt};
        for (int k = 0; k < 5; ++k) { vals[16 * i + 3 * k] = c[k].x; vals[16 * i + 3 * k + 1] = c[k].y; vals[16 * i + 3 * k + 2] = c[k].z; }
        vals[16 * i + 15] = m.ior();
        flags[i] = (m.diff() ? 1 : 0) | (m.spec() ? 2 : 0) | (m.trans() ? 4 : 0) | (m.recur() ? 8 : 0);
    }
}

// Canonical (preorder) node table: boxes n*6 (min, max); topo n*3 (left id, right id, prim) with -1 for none
void orc_get_nodes(void* h, double* boxes, int* topo) {
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->nodes.size(); ++i) {
        const Node* n = s->nodes[i];
        boxes[6 * i] = n->bmin.x; boxes[6 * i + 1] = n->bmin.y; boxes[6 * i + 2] = n->bmin.z;
        boxes[6 * i + 3] = n->bmax.x; boxes[6 * i + 4] = n->bmax.y; boxes[6 * i + 5] = n->bmax.z;
        topo[3 * i] = n->left ? n->left->id : -1;
        topo[3 * i + 1] = n->right ? n->right->id : -1;
        topo[3 * i + 2] = n->prim;
    }
}

// Swap in a tree built elsewhere (same table format as orc_get_nodes, row 0 = root); 1 on success.
int orc_set_tree(void* h, int n, const double* boxes, const int* topo) { return replace_tree(*(Scene*)h, n, boxes, topo) ? 1 : 0; }

// eye, look, u, v (12 doubles)
void orc_get_camera(void* h, double* out12) {
    Scene* s = (Scene*)h;
    const V3 c[4] = {s->cam.eye, s->cam.look, s->cam.u, s->cam.v};
    for (int k = 0; k < 4; ++k) { out12[3 * k] = c[k].x; out12[3 * k + 1] = c[k].y; out12[3 * k + 2] = c[k].z; }
}

// Camera rays through (i + jx)/w, (j + jy)/h for every pixel, index i + j*w.
void orc_camera_rays(void* h, int w, int hh, double jx, double jy, double* o, double* d) {
    Scene* s = (Scene*)h;
    for (int j = 0; j < hh; ++j)
        for (int i = 0; i < w; ++i) {
            V3 ro, rd;
            s->cam.ray_through((i + jx) / double(w), (j + jy) / double(hh), ro, rd);
            size_t k = (size_t)(i + j * w) * 3;
            o[k] = ro.x; o[k + 1] = ro.y; o[k + 2] = ro.z;
            d[k] = rd.x; d[k + 1] = rd.y; d[k + 2] = rd.z;
        }
}

// Trace n rays.  counts: n*4 ints (node_steps, box_tests, tri_tests, max_stack).
// bitmap (nullable): n rows of bitmap_stride bytes, bit k = node k's box was tested.
void orc_trace(void* h, int n, const double* o, const double* d, int* prim, double* t, double* normal,
               uint64_t* vhash, int* counts, uint8_t* bitmap, int bitmap_stride) {
    Scene* s = (Scene*)h;
    for (int r = 0; r < n; ++r) {
        TraceOut out;
        uint8_t* bm = bitmap ? bitmap + (size_t)r * bitmap_stride : nullptr;
        if (bm) std::memset(bm, 0, (size_t)bitmap_stride);
        trace_gpu(*s, V3{o[3 * r], o[3 * r + 1], o[3 * r + 2]}, V3{d[3 * r], d[3 * r + 1], d[3 * r + 2]}, out, bm);
        prim[r] = out.prim;
        t[r] = out.t;
        if (normal) { normal[3 * r] = out.N.x; normal[3 * r + 1] = out.N.y; normal[3 * r + 2] = out.N.z; }
        vhash[r] = out.visit_hash;
        counts[4 * r] = out.node_steps; counts[4 * r + 1] = out.box_tests;
        counts[4 * r + 2] = out.tri_tests; counts[4 * r + 3] = out.max_stack;
    }
}

void orc_philox(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) { philox4x32_10(ctr4, key2, out4); }
double orc_u01(uint32_t lo, uint32_t hi) { return u01_open0(lo, hi); }

// GPU-semantics estimator.  accum (w*h*3 doubles) is added to.  stats: 5 uint64
// (paths, segments, hits, node_steps, tri_tests), nullable.
void orc_render(void* h, int w, int hh, int s0, int s1, int depth, uint64_t seed, int dof, double focal,
                double aperture, int rr, double* accum, uint64_t* stats) {
    Scene* s = (Scene*)h;
    RenderParams p;
    p.w = w; p.h = hh; p.s0 = s0; p.s1 = s1; p.depth = depth; p.seed = seed;
    p.dof = dof; p.focal_distance = focal; p.aperture = aperture; p.rr = rr;
    RenderStats st;
    render_gpu_semantics(*s, p, accum, &st);
    if (stats) { stats[0] = st.paths; stats[1] = st.segments; stats[2] = st.hits; stats[3] = st.node_steps; stats[4] = st.tri_tests; }
}

// Radiance of single paths: n entries of (i, j, sample) -> n*3 doubles.
void orc_paths(void* h, int w, int hh, int depth, uint64_t seed, int dof, double focal, double aperture, int rr,
               int n, const int* ijs, double* rad) {
    Scene* s = (Scene*)h;
    RenderParams p;
    p.w = w; p.h = hh; p.depth = depth; p.seed = seed; p.dof = dof; p.focal_distance = focal; p.aperture = aperture; p.rr = rr;
    for (int k = 0; k < n; ++k) {
        V3 c = path_radiance_gpu(*s, p, ijs[3 * k], ijs[3 * k + 1], ijs[3 * k + 2], nullptr);
        rad[3 * k] = c.x; rad[3 * k + 1] = c.y; rad[3 * k + 2] = c.z;
    }
}

// mean -> clamp -> (int)(255 c), reference: src/path_tracer.cu:436-443
void orc_tonemap(const double* accum, int npix, int spp, uint8_t* rgb) {
    for (int k = 0; k < npix * 3; ++k) {
        double c = accum[k] / spp;
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        rgb[k] = (uint8_t)(int)(255.0 * c);
    }
}

// The reference's CPU path tracer; returns wall seconds.
double orc_render_cpu_reference(void* h, int w, int hh, int spp, int depth, int threads, uint64_t seed,
                                uint8_t* rgb, double* mean) {
    Scene* s = (Scene*)h;
    auto t0 = std::chrono::steady_clock::now();
    render_cpu_reference(*s, w, hh, spp, depth, threads, seed, rgb, mean);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
