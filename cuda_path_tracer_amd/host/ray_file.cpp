#include "ray_file.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace cpt {
namespace {

// ---- 4x4 affine transforms, row-major, column vectors ---------------------------------------
struct Xform {
    double a[4][4];
    bool identity;
};
Xform xf_identity() {
    Xform m{};
    for (int i = 0; i < 4; ++i) m.a[i][i] = 1.0;
    m.identity = true;
    return m;
}
Xform xf_mul(const Xform& p, const Xform& q) {
    if (p.identity) return q;
    if (q.identity) return p;
    Xform r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += p.a[i][k] * q.a[k][j];
            r.a[i][j] = s;
        }
    r.identity = false;
    return r;
}
dvec3 xf_point(const Xform& m, dvec3 v) {
    if (m.identity) return v;
    return dv(m.a[0][0] * v.x + m.a[0][1] * v.y + m.a[0][2] * v.z + m.a[0][3],
              m.a[1][0] * v.x + m.a[1][1] * v.y + m.a[1][2] * v.z + m.a[1][3],
              m.a[2][0] * v.x + m.a[2][1] * v.y + m.a[2][2] * v.z + m.a[2][3]);
}
// normals transform by the inverse transpose of the upper 3x3 (src/scene/scene.h:108-129)
dvec3 xf_normal(const Xform& m, dvec3 n) {
    if (m.identity) return n;
    const double(*a)[4] = m.a;
    double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2], c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    double c10 = a[0][2] * a[2][1] - a[0][1] * a[2][2], c11 = a[0][0] * a[2][2] - a[0][2] * a[2][0], c12 = a[0][1] * a[2][0] - a[0][0] * a[2][1];
    double c20 = a[0][1] * a[1][2] - a[0][2] * a[1][1], c21 = a[0][2] * a[1][0] - a[0][0] * a[1][2], c22 = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    if (det == 0.0) throw std::runtime_error("singular transform");
    // cofactor matrix / det == inverse transpose
    return dv((c00 * n.x + c01 * n.y + c02 * n.z) / det, (c10 * n.x + c11 * n.y + c12 * n.z) / det,
              (c20 * n.x + c21 * n.y + c22 * n.z) / det);
}

// ---- tokens ------------------------------------------------------------------------------------
enum class Tok { Word, Number, Punct, End };
struct Token {
    Tok kind = Tok::End;
    std::string text;
    double value = 0;
    char punct = 0;
    int line = 0;
};

class Lexer {
public:
    explicit Lexer(const std::string& text) : s_(text) { advance(); }
    const Token& peek() const { return cur_; }
    Token take() { Token t = cur_; advance(); return t; }
    bool is_punct(char c) const { return cur_.kind == Tok::Punct && cur_.punct == c; }
    bool is_word(const char* w) const { return cur_.kind == Tok::Word && cur_.text == w; }
    [[noreturn]] void error(const std::string& msg) const {
        std::ostringstream o;
        o << "line " << cur_.line << ": " << msg;
        throw std::runtime_error(o.str());
    }
    void need(char c) {
        if (!is_punct(c)) error(std::string("expected '") + c + "'");
        advance();
    }
    bool maybe(char c) {
        if (!is_punct(c)) return false;
        advance();
        return true;
    }
    std::string word() {
        if (cur_.kind != Tok::Word) error("expected an identifier");
        return take().text;
    }
    double number() {
        if (cur_.kind != Tok::Number) error("expected a number");
        return take().value;
    }

private:
    void skip_blank() {
        for (;;) {
            while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) {
                if (s_[p_] == '\n') ++line_;
                ++p_;
            }
            if (p_ + 1 < s_.size() && s_[p_] == '/' && s_[p_ + 1] == '/') {
                while (p_ < s_.size() && s_[p_] != '\n') ++p_;
            } else if (p_ + 1 < s_.size() && s_[p_] == '/' && s_[p_ + 1] == '*') {
                size_t e = s_.find("*/", p_ + 2);
                if (e == std::string::npos) { cur_.line = line_; error("unterminated comment"); }
                for (size_t k = p_; k < e; ++k) if (s_[k] == '\n') ++line_;
                p_ = e + 2;
            } else {
                return;
            }
        }
    }
    void advance() {
        skip_blank();
        cur_ = Token();
        cur_.line = line_;
        if (p_ >= s_.size()) return;
        unsigned char c = (unsigned char)s_[p_];
        if (std::isalpha(c) || c == '_') {
            cur_.kind = Tok::Word;
            while (p_ < s_.size() && (std::isalnum((unsigned char)s_[p_]) || s_[p_] == '_' || s_[p_] == '-')) cur_.text += s_[p_++];
        } else if (c == '"') {
            cur_.kind = Tok::Word;
            ++p_;
            while (p_ < s_.size() && s_[p_] != '"') {
                if (s_[p_] == '\n') error("unterminated string constant");
                cur_.text += s_[p_++];
            }
            if (p_ >= s_.size()) error("unterminated string constant");
            ++p_;
        } else if (std::isdigit(c) || c == '-' || c == '.') {
            cur_.kind = Tok::Number;
            while (p_ < s_.size() && (std::isdigit((unsigned char)s_[p_]) || s_[p_] == '-' || s_[p_] == '.' || s_[p_] == 'e')) cur_.text += s_[p_++];
            cur_.value = std::atof(cur_.text.c_str());
        } else if (std::strchr("(){},=;", c)) {
            cur_.kind = Tok::Punct;
            cur_.punct = (char)c;
            ++p_;
        } else {
            error(std::string("unexpected character '") + (char)c + "'");
        }
    }
    const std::string& s_;
    size_t p_ = 0;
    int line_ = 1;
    Token cur_;
};

// ---- parser --------------------------------------------------------------------------------------
class Reader {
public:
    explicit Reader(const std::string& text) : lx_(text) {}

    RayScene run() {
        if (!lx_.is_word("SBT-raytracer")) lx_.error("expected 'SBT-raytracer'");
        lx_.take();
        if (lx_.number() > 1.1) lx_.error("scene file version too high; 1.1 and below are readable");
        RayMaterial current;
        Xform root = xf_identity();
        while (lx_.peek().kind != Tok::End) {
            if (lx_.maybe(';')) continue;
            if (lx_.peek().kind != Tok::Word && !lx_.is_punct('{')) lx_.error("expected geometry, camera, or light information");
            if (lx_.is_word("camera")) { lx_.take(); camera(); }
            else if (lx_.is_word("material")) { lx_.take(); lx_.need('='); current = material(current); lx_.maybe(';'); }
            else if (lx_.is_word("point_light") || lx_.is_word("directional_light") || lx_.is_word("ambient_light")) {
                std::string k = lx_.take().text;
                skip_braces();
                note("light '" + k + "' ignored: the path tracer lights the scene with emissive surfaces only");
            } else element(root, current);
        }
        return std::move(out_);
    }

private:
    void note(const std::string& w) { out_.warnings.push_back(w); }

    void skip_braces() {
        lx_.need('{');
        int depth = 1;
        while (depth > 0) {
            if (lx_.peek().kind == Tok::End) lx_.error("unterminated block");
            if (lx_.is_punct('{')) ++depth;
            if (lx_.is_punct('}')) --depth;
            lx_.take();
        }
    }

    dvec3 vec3() {
        lx_.need('(');
        double x = lx_.number(); lx_.need(',');
        double y = lx_.number(); lx_.need(',');
        double z = lx_.number();
        lx_.need(')');
        return dv(x, y, z);
    }
    void vec4(double out[4]) {
        lx_.need('(');
        for (int k = 0; k < 4; ++k) { out[k] = lx_.number(); if (k < 3) lx_.need(','); }
        lx_.need(')');
    }

    // material := IDENT | '{' { key '=' value [';'] } '}'
    RayMaterial material(const RayMaterial& parent) {
        if (lx_.peek().kind == Tok::Word) {
            std::string ref = lx_.take().text;
            auto it = named_.find(ref);
            if (it == named_.end()) lx_.error("unknown material '" + ref + "'");
            return it->second;
        }
        lx_.need('{');
        RayMaterial m = parent;
        std::string name;
        while (!lx_.maybe('}')) {
            std::string key = lx_.word();
            if (key == "name") { lx_.maybe('='); name = lx_.word(); lx_.maybe(';'); continue; }
            lx_.need('=');
            if (lx_.is_word("map")) lx_.error("texture-mapped material parameters are not supported by the GPU path");
            if (key == "emissive") m.ke = vec3();
            else if (key == "ambient") m.ka = vec3();
            else if (key == "specular") m.ks = vec3();
            else if (key == "diffuse") m.kd = vec3();
            else if (key == "reflective") m.kr = vec3();
            else if (key == "transmissive") m.kt = vec3();
            else if (key == "index") { double s = lx_.number(); m.index = dv(s, s, s); }
            else if (key == "shininess") { double s = lx_.number(); m.shininess = dv(s, s, s); }
            else lx_.error("expected a material attribute, got '" + key + "'");
            lx_.maybe(';');
        }
        if (!name.empty()) {
            if (named_.count(name)) lx_.error("redefinition of material '" + name + "'");
            named_[name] = m;
        }
        return m;
    }

    static void mat_apply(const double m[3][3], dvec3 v, dvec3& out) {
        out = dv(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z, m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                 m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
    }
    void camera_update() {  // src/scene/camera.cpp:94-100
        RayCamera& c = out_.camera;
        dvec3 t;
        mat_apply(c.m, dv(1, 0, 0), t);
        c.u = t * c.normalized_height * c.aspect;
        mat_apply(c.m, dv(0, 1, 0), t);
        c.v = t * c.normalized_height;
        mat_apply(c.m, dv(0, 0, -1), c.look);
    }
    void camera() {
        RayCamera& c = out_.camera;
        bool have_view = false, have_up = false;
        dvec3 view = dv(0, 0, -1), up = dv(0, 1, 0);
        lx_.need('{');
        while (!lx_.maybe('}')) {
            std::string key = lx_.word();
            lx_.need('=');
            if (key == "position") c.eye = vec3();
            else if (key == "fov") {
                double fov = lx_.number();
                fov /= (180.0 / 3.14159265359);  // the reference's camera uses this short pi (camera.cpp:4)
                c.normalized_height = 2 * std::tan(fov / 2);
                camera_update();
            } else if (key == "aspectratio") { c.aspect = lx_.number(); camera_update(); }
            else if (key == "viewdir") { view = vec3(); have_view = true; }
            else if (key == "updir") { up = vec3(); have_up = true; }
            else if (key == "quaternian") {
                double q[4];
                vec4(q);
                double r = q[0], i = q[1], j = q[2], k = q[3], a[3][3];
                a[0][0] = 1.0 - 2.0 * (i * i + j * j); a[0][1] = 2.0 * (r * i - j * k); a[0][2] = 2.0 * (j * r + i * k);
                a[1][0] = 2.0 * (r * i + j * k); a[1][1] = 1.0 - 2.0 * (j * j + r * r); a[1][2] = 2.0 * (i * j - r * k);
                a[2][0] = 2.0 * (j * r - i * k); a[2][1] = 2.0 * (i * j + r * k); a[2][2] = 1.0 - 2.0 * (i * i + r * r);
                for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) c.m[col][row] = a[row][col];
                camera_update();
            } else lx_.error("expected a camera attribute, got '" + key + "'");
            lx_.maybe(';');
        }
        if (have_view != have_up) lx_.error(have_view ? "expected 'updir'" : "expected 'viewdir'");
        if (have_view) {  // camera.cpp:64-75: matrix columns are (up x -view, up, -view), not normalised
            dvec3 z = -view, y = up, x = cross(y, z);
            c.m[0][0] = x.x; c.m[0][1] = x.y; c.m[0][2] = x.z;
            c.m[1][0] = y.x; c.m[1][1] = y.y; c.m[1][2] = y.z;
            c.m[2][0] = z.x; c.m[2][1] = z.y; c.m[2][2] = z.z;
            camera_update();
        }
    }

    // element := mesh | analytic primitive | '{' group '}' | transform '(' args ',' element ')' [';']
    void element(const Xform& xf, const RayMaterial& mat) {
        if (lx_.is_punct('{')) { group(xf, mat); return; }
        std::string k = lx_.word();
        if (k == "polymesh" || k == "trimesh") { mesh(xf, mat); return; }
        if (k == "sphere" || k == "box" || k == "square" || k == "cylinder" || k == "cone") {
            skip_braces();
            note("'" + k + "' ignored: the GPU path renders polymesh geometry only");
            return;
        }
        Xform local = xf_identity();
        local.identity = false;
        lx_.need('(');
        if (k == "translate") {
            for (int r = 0; r < 3; ++r) { local.a[r][3] = lx_.number(); lx_.need(','); }
        } else if (k == "scale") {
            double x = lx_.number(), y = x, z = x;
            lx_.need(',');
            if (lx_.peek().kind == Tok::Number) { y = lx_.number(); lx_.need(','); z = lx_.number(); lx_.need(','); }
            local.a[0][0] = x; local.a[1][1] = y; local.a[2][2] = z;
        } else if (k == "rotate") {
            double x = lx_.number(); lx_.need(',');
            double y = lx_.number(); lx_.need(',');
            double z = lx_.number(); lx_.need(',');
            double w = lx_.number(); lx_.need(',');
            dvec3 ax = normalize(dv(x, y, z));
            double c = std::cos(w), s = std::sin(w);
            dvec3 t = ax * (1 - c);
            local.a[0][0] = c + t.x * ax.x;        local.a[0][1] = t.y * ax.x - s * ax.z; local.a[0][2] = t.z * ax.x + s * ax.y;
            local.a[1][0] = t.x * ax.y + s * ax.z; local.a[1][1] = c + t.y * ax.y;        local.a[1][2] = t.z * ax.y - s * ax.x;
            local.a[2][0] = t.x * ax.z - s * ax.y; local.a[2][1] = t.y * ax.z + s * ax.x; local.a[2][2] = c + t.z * ax.z;
        } else if (k == "transform") {
            for (int r = 0; r < 4; ++r) { vec4(local.a[r]); lx_.need(','); }
        } else {
            lx_.error("expected geometry, camera, or light information, got '" + k + "'");
        }
        element(xf_mul(xf, local), mat);
        lx_.need(')');
        lx_.maybe(';');
    }

    void group(const Xform& xf, const RayMaterial& mat) {
        lx_.need('{');
        RayMaterial cur = mat;
        while (!lx_.maybe('}')) {
            if (lx_.is_word("material")) { lx_.take(); lx_.need('='); cur = material(mat); lx_.maybe(';'); }
            else element(xf, cur);
        }
    }

    void mesh(const Xform& xf, const RayMaterial& parent) {
        RayMesh me;
        me.material = parent;
        bool generate = false;
        std::vector<dvec3> local_vertices, local_normals;
        std::vector<int> tris;
        lx_.need('{');
        while (!lx_.maybe('}')) {
            std::string key = lx_.word();
            if (key == "gennormals") { lx_.need(';'); generate = true; continue; }
            lx_.need('=');
            if (key == "material") me.material = material(parent);
            else if (key == "name") me.name = lx_.word();
            else if (key == "points" || key == "normals") {
                std::vector<dvec3>& dst = key == "points" ? local_vertices : local_normals;
                lx_.need('(');
                if (!lx_.is_punct(')')) {
                    dst.push_back(vec3());
                    while (!lx_.is_punct(')')) { lx_.need(','); dst.push_back(vec3()); }
                }
                lx_.need(')');
            } else if (key == "faces") {
                lx_.need('(');
                auto polygon = [&]() {  // fan triangulation about the first index (Parser.cpp:699-717)
                    std::vector<double> idx;
                    lx_.need('(');
                    if (!lx_.is_punct(')')) {
                        idx.push_back(lx_.number());
                        while (!lx_.is_punct(')')) { lx_.need(','); idx.push_back(lx_.number()); }
                    }
                    lx_.need(')');
                    if (idx.size() < 3) lx_.error("faces must have at least 3 vertices");
                    for (size_t q = 2; q < idx.size(); ++q) {
                        tris.push_back((int)idx[0]); tris.push_back((int)idx[q - 1]); tris.push_back((int)idx[q]);
                    }
                };
                if (!lx_.is_punct(')')) {
                    polygon();
                    while (!lx_.is_punct(')')) { lx_.need(','); polygon(); }
                }
                lx_.need(')');
            } else if (key == "materials") {
                lx_.need('(');
                if (!lx_.is_punct(')')) {
                    (void)material(me.material);
                    while (!lx_.is_punct(')')) { lx_.need(','); (void)material(me.material); }
                }
                lx_.need(')');
                note("per-vertex materials ignored: the GPU path shades with the mesh material");
            } else lx_.error("expected a trimesh attribute, got '" + key + "'");
            lx_.maybe(';');
        }
        const int vc = (int)local_vertices.size();
        for (const dvec3& v : local_vertices) me.vertices.push_back(xf_point(xf, v));
        // keep non-degenerate faces only (src/SceneObjects/trimesh.cpp:40-58, trimesh.h:118-124)
        std::vector<dvec3> face_n;
        for (size_t f = 0; f + 2 < tris.size(); f += 3) {
            int a = tris[f], b = tris[f + 1], c = tris[f + 2];
            if (a < 0 || b < 0 || c < 0 || a >= vc || b >= vc || c >= vc) {
                std::ostringstream o;
                o << "bad face in trimesh: (" << a << ", " << b << ", " << c << ")";
                lx_.error(o.str());
            }
            dvec3 A = me.vertices[a], B = me.vertices[b], C = me.vertices[c];
            if (length(B - A) == 0.0 || length(C - A) == 0.0 || length(B - C) == 0.0) continue;
            me.faces.push_back(a); me.faces.push_back(b); me.faces.push_back(c);
            face_n.push_back(normalize(cross(B - A, C - A)));
        }
        for (const dvec3& n : local_normals) me.normals.push_back(xf_normal(xf, n));
        if (generate) {  // mean of the adjacent unit face normals (trimesh.cpp:174-195)
            me.normals.resize(vc, dv(0, 0, 0));
            std::vector<int> count(vc, 0);
            for (size_t f = 0; f < face_n.size(); ++f)
                for (int q = 0; q < 3; ++q) {
                    int vi = me.faces[3 * f + q];
                    me.normals[vi] = me.normals[vi] + face_n[f];
                    ++count[vi];
                }
            for (int i = 0; i < vc; ++i)
                if (count[i]) me.normals[i] = me.normals[i] / (double)count[i];
        }
        if (!me.normals.empty() && me.normals.size() != me.vertices.size()) lx_.error("bad trimesh: wrong number of normals");
        out_.meshes.push_back(std::move(me));
    }

    Lexer lx_;
    RayScene out_;
    std::map<std::string, RayMaterial> named_;
};

}  // namespace

RayScene read_ray_text(const std::string& text) {
    Reader r(text);
    return r.run();
}

RayScene read_ray_file(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("couldn't read scene file " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return read_ray_text(ss.str());
}

}  // namespace cpt
