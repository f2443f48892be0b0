// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
// Scene file -> meshes -> primitive list -> the reference's host BVH.
#include "oracle.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>

namespace orc {

double length(V3 v) { return std::sqrt(dot(v, v)); }
V3 normalize(V3 v) { return v * (1.0 / std::sqrt(dot(v, v))); }

// reference: src/gpu/GPUMaterial.cuh:33-38, src/scene/material.h:273-278
bool Material::diff() const { return !(length(kd) == 0); }
bool Material::refl() const { return !(length(kr) == 0); }
bool Material::trans() const { return !(length(kt) == 0); }
bool Material::spec() const { return refl() || !(length(ks) == 0); }
double Material::ior() const { return (0.299 * index.x) + (0.587 * index.y) + (0.114 * index.z); }

// reference: src/scene/camera.cpp:94-100 (GLM mat3 * vec3, then scalar products left to right)
static V3 mat_mul(const double m[3][3], V3 v) {
    return {m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
            m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
            m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z};
}
void Camera::update() {
    u = mat_mul(m, V3{1, 0, 0}) * normalizedHeight * aspect;
    v = mat_mul(m, V3{0, 1, 0}) * normalizedHeight;
    look = mat_mul(m, V3{0, 0, -1});
}
// reference: src/gpu/GPUCamera.cu:22-32
void Camera::ray_through(double x, double y, V3& o, V3& d) const {
    x -= 0.5;
    y -= 0.5;
    d = normalize(look + x * u + y * v);
    o = eye;
}

Scene::~Scene() {
    for (Node* n : nodes) delete n;
}

// ------------------------------------------------------------------------------------------
// Minimal reader for the polymesh subset of the .ray format.
// reference: src/parser/Tokenizer.cpp:131-237 (comments, scalars via atof over [0-9.-e]),
//            src/parser/Parser.cpp:27-155 (top level, camera), :551-717 (trimesh, faces),
//            :986-1070 (material)
namespace {

struct Lex {
    std::string s;
    size_t p = 0;
    explicit Lex(std::string text) : s(std::move(text)) {}
    void ws() {
        for (;;) {
            while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
            if (p + 1 < s.size() && s[p] == '/' && s[p + 1] == '/') {
                while (p < s.size() && s[p] != '\n') ++p;
            } else if (p + 1 < s.size() && s[p] == '/' && s[p + 1] == '*') {
                size_t e = s.find("*/", p + 2);
                if (e == std::string::npos) throw std::runtime_error("unterminated comment");
                p = e + 2;
            } else {
                return;
            }
        }
    }
    bool eof() { ws(); return p >= s.size(); }
    char peek() { ws(); return p < s.size() ? s[p] : '\0'; }
    bool accept(char c) { if (peek() == c) { ++p; return true; } return false; }
    void expect(char c) {
        if (!accept(c)) {
            std::ostringstream o;
            o << "expected '" << c << "' at offset " << p;
            throw std::runtime_error(o.str());
        }
    }
    bool at_word() { char c = peek(); return std::isalpha((unsigned char)c) || c == '_' || c == '"'; }
    std::string word() {
        ws();
        std::string r;
        if (p < s.size() && s[p] == '"') {
            ++p;
            while (p < s.size() && s[p] != '"') r += s[p++];
            ++p;
            return r;
        }
        while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-')) r += s[p++];
        if (r.empty()) throw std::runtime_error("expected identifier");
        return r;
    }
    double scalar() {
        ws();
        std::string r;
        while (p < s.size() && (std::isdigit((unsigned char)s[p]) || s[p] == '-' || s[p] == '.' || s[p] == 'e')) r += s[p++];
        if (r.empty()) throw std::runtime_error("expected scalar");
        return std::atof(r.c_str());
    }
    V3 vec3() {
        V3 v;
        expect('(');
        v.x = scalar(); expect(',');
        v.y = scalar(); expect(',');
        v.z = scalar();
        expect(')');
        return v;
    }
    std::vector<double> scalar_list() {
        std::vector<double> r;
        expect('(');
        if (peek() != ')') {
            r.push_back(scalar());
            while (peek() != ')') { expect(','); r.push_back(scalar()); }
        }
        expect(')');
        return r;
    }
    void skip_block() {  // skip a balanced {...}
        expect('{');
        int d = 1;
        while (d > 0) {
            if (p >= s.size()) throw std::runtime_error("unterminated block");
            if (s[p] == '{') ++d;
            if (s[p] == '}') --d;
            ++p;
        }
    }
};

Material read_material(Lex& L, const Material& parent, std::map<std::string, Material>& named) {
    if (L.peek() != '{') {
        std::string ref = L.word();
        return named[ref];
    }
    L.expect('{');
    Material m = parent;
    std::string name;
    while (!L.accept('}')) {
        std::string k = L.word();
        if (k == "name") {
            L.accept('=');
            name = L.word();
            L.accept(';');
            continue;
        }
        L.expect('=');
        if (k == "emissive") m.ke = L.vec3();
        else if (k == "ambient") m.ka = L.vec3();
        else if (k == "specular") m.ks = L.vec3();
        else if (k == "diffuse") m.kd = L.vec3();
        else if (k == "reflective") m.kr = L.vec3();
        else if (k == "transmissive") m.kt = L.vec3();
        else if (k == "index") { double s = L.scalar(); m.index = {s, s, s}; }
        else if (k == "shininess") { double s = L.scalar(); m.shininess = {s, s, s}; }
        else throw std::runtime_error("unsupported material attribute '" + k + "'");
        L.accept(';');
    }
    if (!name.empty()) named[name] = m;
    return m;
}

void read_camera(Lex& L, Camera& cam) {
    bool hv = false, hu = false;
    V3 view{0, 0, -1}, up{0, 1, 0};
    L.expect('{');
    while (!L.accept('}')) {
        std::string k = L.word();
        L.expect('=');
        if (k == "position") cam.eye = L.vec3();
        else if (k == "fov") {
            double fov = L.scalar();
            fov /= (180.0 / 3.14159265359);  // camera.cpp:4,80-82 (short PI)
            cam.normalizedHeight = 2 * std::tan(fov / 2);
            cam.update();
        } else if (k == "aspectratio") { cam.aspect = L.scalar(); cam.update(); }
        else if (k == "viewdir") { view = L.vec3(); hv = true; }
        else if (k == "updir") { up = L.vec3(); hu = true; }
        else if (k == "quaternian") {
            L.expect('(');
            double r = L.scalar(); L.expect(',');
            double i = L.scalar(); L.expect(',');
            double j = L.scalar(); L.expect(',');
            double q = L.scalar(); L.expect(')');
            double a[3][3];  // camera.cpp:48-60, then transpose
            a[0][0] = 1.0 - 2.0 * (i * i + j * j); a[0][1] = 2.0 * (r * i - j * q); a[0][2] = 2.0 * (j * r + i * q);
            a[1][0] = 2.0 * (r * i + j * q); a[1][1] = 1.0 - 2.0 * (j * j + r * r); a[1][2] = 2.0 * (i * j - r * q);
            a[2][0] = 2.0 * (j * r - i * q); a[2][1] = 2.0 * (i * j + r * q); a[2][2] = 1.0 - 2.0 * (i * i + r * r);
            for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) cam.m[c][rr] = a[rr][c];
            cam.update();
        } else throw std::runtime_error("unsupported camera attribute '" + k + "'");
        L.accept(';');
    }
    if (hv != hu) throw std::runtime_error("viewdir and updir must be given together");
    if (hv) {  // camera.cpp:64-75: columns (x, y, z), not normalised
        V3 z = -view, y = up, x = cross(y, z);
        cam.m[0][0] = x.x; cam.m[0][1] = x.y; cam.m[0][2] = x.z;
        cam.m[1][0] = y.x; cam.m[1][1] = y.y; cam.m[1][2] = y.z;
        cam.m[2][0] = z.x; cam.m[2][1] = z.y; cam.m[2][2] = z.z;
        cam.update();
    }
}

void read_mesh(Lex& L, Scene& sc, const Material& parent, std::map<std::string, Material>& named) {
    Mesh me;
    me.mat = parent;
    bool gen = false;
    std::vector<int> faces;  // fan-triangulated index triples, Parser.cpp:699-717
    L.expect('{');
    while (!L.accept('}')) {
        std::string k = L.word();
        if (k == "gennormals") { L.expect(';'); gen = true; continue; }
        L.expect('=');
        if (k == "material") me.mat = read_material(L, parent, named);
        else if (k == "name") me.name = L.word();
        else if (k == "points") {
            L.expect('(');
            if (L.peek() != ')') {
                me.verts.push_back(L.vec3());
                while (L.peek() != ')') { L.expect(','); me.verts.push_back(L.vec3()); }
            }
            L.expect(')');
        } else if (k == "normals") {
            L.expect('(');
            if (L.peek() != ')') {
                me.normals.push_back(L.vec3());
                while (L.peek() != ')') { L.expect(','); me.normals.push_back(L.vec3()); }
            }
            L.expect(')');
            me.parsed_normals = true;
        } else if (k == "faces") {
            L.expect('(');
            auto one = [&]() {
                std::vector<double> pts = L.scalar_list();
                if (pts.size() < 3) throw std::runtime_error("faces must have at least 3 vertices");
                double a = pts[0], b = pts[1];
                for (size_t q = 2; q < pts.size(); ++q) {
                    double c = pts[q];
                    faces.push_back((int)a); faces.push_back((int)b); faces.push_back((int)c);
                    b = c;
                }
            };
            if (L.peek() != ')') {
                one();
                while (L.peek() != ')') { L.expect(','); one(); }
            }
            L.expect(')');
        } else throw std::runtime_error("unsupported polymesh attribute '" + k + "'");
        L.accept(';');
    }
    // Trimesh::addFace (trimesh.cpp:40-58) + TrimeshFace ctor (trimesh.h:101-134)
    int vc = (int)me.verts.size();
    for (size_t f = 0; f + 2 < faces.size(); f += 3) {
        int a = faces[f], b = faces[f + 1], c = faces[f + 2];
        if (a < 0 || b < 0 || c < 0 || a >= vc || b >= vc || c >= vc) throw std::runtime_error("bad face in trimesh");  // the reference checks the upper bound only
        V3 A = me.verts[a], B = me.verts[b], C = me.verts[c];
        if (length(B - A) == 0.0 || length(C - A) == 0.0 || length(B - C) == 0.0) continue;  // degenerate: dropped
        me.face_ids.push_back(a); me.face_ids.push_back(b); me.face_ids.push_back(c);
        me.face_n.push_back(normalize(cross(B - A, C - A)));
    }
    if (gen) {  // trimesh.cpp:174-195
        me.normals.resize(vc, V3{0, 0, 0});
        std::vector<int> cnt(vc, 0);
        for (size_t f = 0; f < me.face_n.size(); ++f)
            for (int q = 0; q < 3; ++q) {
                int vi = me.face_ids[3 * f + q];
                me.normals[vi] = me.normals[vi] + me.face_n[f];
                ++cnt[vi];
            }
        for (int i = 0; i < vc; ++i)
            if (cnt[i]) me.normals[i] = me.normals[i] / (double)cnt[i];
    }
    if (!me.normals.empty() && me.normals.size() != me.verts.size())
        throw std::runtime_error("Bad Trimesh: Wrong number of normals.");
    sc.meshes.push_back(std::move(me));
}

}  // namespace

Scene* load_scene(const std::string& path, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "cannot read " + path; return nullptr; }
    std::stringstream ss;
    ss << f.rdbuf();
    Scene* sc = new Scene;
    try {
        Lex L(ss.str());
        if (L.word() != "SBT-raytracer") throw std::runtime_error("missing SBT-raytracer header");
        if (L.scalar() > 1.1) throw std::runtime_error("version too high");
        Material cur;
        std::map<std::string, Material> named;
        while (!L.eof()) {
            if (L.accept(';')) continue;
            std::string k = L.word();
            if (k == "camera") read_camera(L, sc->cam);
            else if (k == "polymesh" || k == "trimesh") read_mesh(L, *sc, cur, named);
            else if (k == "material") { L.expect('='); cur = read_material(L, cur, named); L.accept(';'); }
            else if (k == "point_light" || k == "directional_light" || k == "ambient_light") L.skip_block();
            else throw std::runtime_error("unsupported scene element '" + k + "' (oracle reads the polymesh subset)");
        }
        // primitive list: BVHTree::build (bvh.cpp:108-113) + Trimesh::getAll (trimesh.cpp:201-208)
        for (size_t mi = 0; mi < sc->meshes.size(); ++mi) {
            const Mesh& me = sc->meshes[mi];
            for (size_t fi = 0; fi < me.face_n.size(); ++fi) {
                Prim p;
                p.mesh = (int)mi; p.face = (int)fi;
                p.a = me.verts[me.face_ids[3 * fi]];
                p.b = me.verts[me.face_ids[3 * fi + 1]];
                p.c = me.verts[me.face_ids[3 * fi + 2]];
                p.n = me.face_n[fi];
                auto mx = [](double x, double y) { return (x < y) ? y : x; };  // glm::max
                auto mn = [](double x, double y) { return (y < x) ? y : x; };  // glm::min
                V3 hi{mx(p.a.x, p.b.x), mx(p.a.y, p.b.y), mx(p.a.z, p.b.z)};   // trimesh.h:148-161
                V3 lo{mn(p.a.x, p.b.x), mn(p.a.y, p.b.y), mn(p.a.z, p.b.z)};
                p.bmax = {mx(p.c.x, hi.x), mx(p.c.y, hi.y), mx(p.c.z, hi.z)};
                p.bmin = {mn(p.c.x, lo.x), mn(p.c.y, lo.y), mn(p.c.z, lo.z)};
                p.center = ((p.a + p.b) + p.c) / V3{3.0, 3.0, 3.0};           // trimesh.cpp:197-199
                sc->prims.push_back(p);
            }
        }
        build_bvh(*sc);
    } catch (const std::exception& e) {
        err = e.what();
        delete sc;
        return nullptr;
    }
    return sc;
}

// ------------------------------------------------------------------------------------------
// Host BVH build, restated naively (O(window * n) box unions per split, as the reference).
// reference: src/scene/bvh.cpp:128-220, src/scene/bbox.cc:82-93
namespace {

struct Box { V3 lo, hi; };

Box range_box(const Scene& s, int st, int end) {
    Box b{s.prims[s.order[st]].bmin, s.prims[s.order[st]].bmax};
    for (int i = st + 1; i < end; ++i) {
        const Prim& p = s.prims[s.order[i]];
        b.lo = {(p.bmin.x < b.lo.x) ? p.bmin.x : b.lo.x, (p.bmin.y < b.lo.y) ? p.bmin.y : b.lo.y, (p.bmin.z < b.lo.z) ? p.bmin.z : b.lo.z};
        b.hi = {(b.hi.x < p.bmax.x) ? p.bmax.x : b.hi.x, (b.hi.y < p.bmax.y) ? p.bmax.y : b.hi.y, (b.hi.z < p.bmax.z) ? p.bmax.z : b.hi.z};
    }
    return b;
}
double area(const Box& b) {
    return 2.0 * ((b.hi.x - b.lo.x) * (b.hi.y - b.lo.y) + (b.hi.y - b.lo.y) * (b.hi.z - b.lo.z) +
                  (b.hi.z - b.lo.z) * (b.hi.x - b.lo.x));
}
int choose_split(const Scene& s, int st, int end) {
    int mid = st + (end - st) / 2;
    const int samples = 500;
    const double trimeshCost = 1, bbCost = 1;
    int best = mid;
    double cost = std::numeric_limits<double>::max();
    Box total = range_box(s, st, end);
    for (int i = std::max(mid - samples, st + 1); i < std::min(mid + samples, end - 1); i++) {
        Box sA = range_box(s, st, i);
        Box sB = range_box(s, i, end);
        double myCost = bbCost + (area(sA) / area(total)) * (i - st) * trimeshCost +
                        (area(sB) / area(total)) * (end - i) * trimeshCost;
        if (myCost < cost) { cost = myCost; best = i; }
    }
    return best;
}
Node* build_range(Scene& s, int st, int end) {
    Node* node = new Node;
    Box b = range_box(s, st, end);
    node->bmin = b.lo;
    node->bmax = b.hi;
    if (end - st <= 1) { node->prim = s.order[st]; return node; }
    V3 diff = b.hi - b.lo;
    int axis;
    if (diff.x >= diff.y && diff.x >= diff.z) axis = 0;
    else if (diff.y >= diff.x && diff.y >= diff.z) axis = 1;
    else axis = 2;
    std::sort(s.order.begin() + st, s.order.begin() + end, [&](int a, int c) {
        return comp(s.prims[a].center, axis) < comp(s.prims[c].center, axis);
    });
    int mid = choose_split(s, st, end);
    node->left = build_range(s, st, mid);
    node->right = build_range(s, mid, end);
    return node;
}
int number_preorder(Scene& s, Node* n) {
    n->id = (int)s.nodes.size();
    s.nodes.push_back(n);
    if (n->leaf()) return 1;
    int hl = number_preorder(s, n->left);
    int hr = number_preorder(s, n->right);
    return 1 + std::max(hl, hr);
}

}  // namespace

void build_bvh(Scene& s) {
    for (Node* n : s.nodes) delete n;
    s.nodes.clear();
    s.root = nullptr;
    s.height = 0;
    s.order.resize(s.prims.size());
    for (size_t i = 0; i < s.prims.size(); ++i) s.order[i] = (int)i;
    if (s.prims.empty()) return;
    s.root = build_range(s, 0, (int)s.prims.size());
    s.height = number_preorder(s, s.root);
}

// Replaces the scene's tree by one given as a table (row 0 = root): boxes n*6 (min, max), topo n*3
// (left row, right row, prim; -1 for none).  The traversal under test (GPUBVH.cu:38-99) is defined for any
// binary tree, so this lets the checker walk a tree that some other builder produced.  Returns false if the
// table is not a tree over this scene's triangles.
bool replace_tree(Scene& s, int n, const double* boxes, const int* topo) {
    if (n != 2 * (int)s.prims.size() - 1) return false;
    std::vector<Node*> rows((size_t)n);
    for (int i = 0; i < n; ++i) rows[(size_t)i] = new Node();
    std::vector<int> parents((size_t)n, 0), used((size_t)s.prims.size(), 0);
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
        Node* nd = rows[(size_t)i];
        nd->bmin = V3{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]};
        nd->bmax = V3{boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]};
        int l = topo[3 * i], r = topo[3 * i + 1], p = topo[3 * i + 2];
        if (p >= 0) {
            ok = l < 0 && r < 0 && p < (int)s.prims.size() && !used[(size_t)p]++;
            nd->prim = p;
        } else {
            ok = l > 0 && r > 0 && l < n && r < n && l != r && !parents[(size_t)l]++ && !parents[(size_t)r]++;
            if (ok) { nd->left = rows[(size_t)l]; nd->right = rows[(size_t)r]; }
        }
    }
    for (int i = 1; i < n && ok; ++i) ok = parents[(size_t)i] == 1;
    if (!ok) {
        for (Node* nd : rows) delete nd;
        return false;
    }
    for (Node* nd : s.nodes) delete nd;
    s.nodes.clear();
    s.root = rows[0];
    s.height = number_preorder(s, s.root);
    return true;
}

}  // namespace orc
