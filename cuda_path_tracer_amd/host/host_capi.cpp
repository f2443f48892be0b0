// C ABI of the host scene library (include/cpt_host.h).
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "cpt_host.h"
#include "flatten.h"
#include "png_write.h"

struct cpth_scene {
    cpt::FlatScene flat;
};

static int set_err(char* err, int errlen, const std::string& msg) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
    return -1;
}

extern "C" {

int cpth_load(const char* ray_path, cpth_scene** out, char* err, int errlen) {
    if (!ray_path || !out) return set_err(err, errlen, "cpth_load: NULL argument");
    *out = nullptr;
    try {
        cpth_scene* s = new cpth_scene;
        try { s->flat = cpt::load_flat_scene(ray_path); } catch (...) { delete s; throw; }
        *out = s;
        return 0;
    } catch (const std::exception& e) {
        return set_err(err, errlen, e.what());
    }
}

int cpth_load_text(const char* ray_text, cpth_scene** out, char* err, int errlen) {
    if (!ray_text || !out) return set_err(err, errlen, "cpth_load_text: NULL argument");
    *out = nullptr;
    try {
        cpth_scene* s = new cpth_scene;
        try { s->flat = cpt::flatten_scene(cpt::read_ray_text(ray_text)); } catch (...) { delete s; throw; }
        *out = s;
        return 0;
    } catch (const std::exception& e) {
        return set_err(err, errlen, e.what());
    }
}

void cpth_free(cpth_scene* s) { delete s; }

int cpth_describe(const cpth_scene* s, cpt_scene_desc* out) {
    if (!s || !out) return -1;
    *out = s->flat.describe();
    return 0;
}

int cpth_describe_fast(cpth_scene* s, cpt_scene_desc* out) {
    if (!s || !out) return -1;
    if (s->flat.bounds.empty()) return -1;  // loaded without a tree
    try {
        *out = s->flat.describe_fast();
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}

double cpth_aspect(const cpth_scene* s) { return s ? s->flat.aspect : 1.0; }
int cpth_warning_count(const cpth_scene* s) { return s ? (int)s->flat.warnings.size() : 0; }
const char* cpth_warning(const cpth_scene* s, int k) {
    if (!s || k < 0 || k >= (int)s->flat.warnings.size()) return "";
    return s->flat.warnings[(size_t)k].c_str();
}

static int tree_table(const std::vector<cpt::BvhNode>& nodes, double* boxes, int32_t* topo);
int cpth_tree_nodes(const cpth_scene* s, double* boxes, int32_t* topo) {
    if (!s || !boxes || !topo) return -1;
    return tree_table(s->flat.bvh.nodes, boxes, topo);
}
int cpth_fast_tree_nodes(cpth_scene* s, double* boxes, int32_t* topo) {
    if (!s || !boxes || !topo) return -1;
    cpt_scene_desc d;
    if (cpth_describe_fast(s, &d) != 0) return -1;
    return tree_table(s->flat.fast_bvh.nodes, boxes, topo);
}
static int tree_table(const std::vector<cpt::BvhNode>& nodes, double* boxes, int32_t* topo) {
    for (size_t i = 0; i < nodes.size(); ++i) {
        boxes[6 * i] = nodes[i].lo.x; boxes[6 * i + 1] = nodes[i].lo.y; boxes[6 * i + 2] = nodes[i].lo.z;
        boxes[6 * i + 3] = nodes[i].hi.x; boxes[6 * i + 4] = nodes[i].hi.y; boxes[6 * i + 5] = nodes[i].hi.z;
        topo[3 * i] = nodes[i].left; topo[3 * i + 1] = nodes[i].right; topo[3 * i + 2] = nodes[i].prim;
    }
    return 0;
}

int cpth_write_image(const char* path, int32_t w, int32_t h, const uint8_t* rgb, char* err, int errlen) {
    if (!path) return set_err(err, errlen, "cpth_write_image: NULL path");
    std::string e;
    if (!cpt::write_image(path, w, h, rgb, e)) return set_err(err, errlen, e);
    return 0;
}

}  // extern "C"
