// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
// The reference's *CPU* path tracer, restated: branching recursive estimator over a recursive,
// unordered BVH walk with the host epsilon (1e-8) and the world->local ray round trip.
// Used as the same-run host-CPU baseline and as the converged-image comparison target.
#include "oracle.h"

#include <atomic>
#include <cmath>
#include <random>
#include <thread>

namespace orc {
namespace {

const double PI_RT = 3.14159265358979311600;  // src/RayTracer.h:5
const double RAY_EPSILON_HOST = 0.00000001;   // src/scene/ray.h:135

// reference: src/scene/bbox.cc:33-70
bool slab_host(const V3& o, const V3& d, const V3& bmin, const V3& bmax) {
    double tMin = -1.0e308, tMax = 1.0e308;
    for (int a = 0; a < 3; ++a) {
        double vd = comp(d, a);
        if (vd == 0.0) continue;
        double v1 = comp(bmin, a) - comp(o, a);
        double v2 = comp(bmax, a) - comp(o, a);
        double t1 = v1 / vd, t2 = v2 / vd;
        if (t1 > t2) { double tt = t1; t1 = t2; t2 = tt; }
        if (t1 > tMin) tMin = t1;
        if (t2 < tMax) tMax = t2;
        if (tMin > tMax) return false;
        if (tMax < RAY_EPSILON_HOST) return false;
    }
    return true;
}

struct HostHit { double t = -1; V3 N{0, 0, 0}; int prim = -1; };

// reference: src/scene/scene.cpp:14-39 wrapping src/SceneObjects/trimesh.cpp:98-170.
// All bundled scenes have identity transforms, so inverse*p == p; what survives of the
// world->local round trip is the direction perturbation (p+d)-p and the t/length rescale.
bool face_host(const Scene& s, const Prim& p, const V3& ro, const V3& rd, HostHit& out) {
    if (!slab_host(ro, rd, p.bmin, p.bmax)) return false;
    V3 pos = ro;
    V3 dir = (ro + rd) - pos;
    double len = length(dir);
    dir = normalize(dir);
    const Mesh& me = s.meshes[p.mesh];
    double numerator = -dot(pos - p.a, p.n);
    double denominator = dot(dir, p.n);
    if (denominator >= 0) {
        if (!me.mat.recur() || denominator == 0) return false;
    }
    double t = numerator / denominator;
    if (t < 0) return false;
    V3 P = pos + dir * t;
    V3 vba = p.b - p.a, vcb = p.c - p.b, vac = p.a - p.c;
    V3 vpa = P - p.a, vpb = P - p.b, vpc = P - p.c;
    bool res = dot(cross(vba, vpa), p.n) >= 0 && dot(cross(vcb, vpb), p.n) >= 0 &&
               dot(cross(vac, vpc), p.n) >= 0;
    if (!res) return false;
    V3 N;
    if (me.parsed_normals || me.use_vertex_normals()) {
        double alpha = length(cross(p.b - P, p.c - P));
        double beta = length(cross(P - p.a, p.c - p.a));
        double denom = length(cross(vba, p.c - p.a));
        alpha /= denom;
        beta /= denom;
        N = normalize(alpha * me.normals[me.face_ids[3 * p.face]] + beta * me.normals[me.face_ids[3 * p.face + 1]] +
                      (1 - alpha - beta) * me.normals[me.face_ids[3 * p.face + 2]]);
    } else {
        N = normalize(p.n);
    }
    out.N = normalize(N);  // localToGlobalCoordsNormal with an identity normal matrix
    out.t = t / len;
    return true;
}

// reference: src/scene/bvh.cpp:244-297 (both children always visited; right wins equal t)
HostHit walk_host(const Scene& s, const Node* n, const V3& o, const V3& d) {
    HostHit res;
    if (!n) return res;
    if (!slab_host(o, d, n->bmin, n->bmax)) { res.t = -2; return res; }
    if (n->leaf()) {
        HostHit h;
        if (face_host(s, s.prims[n->prim], o, d, h)) { h.prim = n->prim; return h; }
        res.t = -1;
        return res;
    }
    HostHit li = walk_host(s, n->left, o, d);
    HostHit ri = walk_host(s, n->right, o, d);
    if (li.t < 0) return ri;
    if (ri.t < 0) return li;
    return li.t < ri.t ? li : ri;
}

struct HostRng {
    std::default_random_engine re;
    std::uniform_real_distribution<double> unif{0, 1};
    std::uniform_real_distribution<double> unbounded;
    std::uniform_real_distribution<double> jitter{-0.5, 0.5};
};

// reference: src/path_tracer.cu:27-198
V3 path_host(const Scene& s, V3 o, V3 d, double curIndex, int depth, HostRng& g) {
    V3 colorC{0, 0, 0};
    HostHit i = walk_host(s, s.root, o, d);
    if (!(i.t >= 0)) return V3{0, 0, 0};
    const Material& m = s.meshes[s.prims[i.prim].mesh].mat;
    colorC = colorC + m.ke * 1.0;
    if (depth <= 0) return colorC;
    V3 at = o + (i.t * d);
    if (m.diff()) {
        V3 normal = normalize(i.N);
        double r1 = 2 * PI_RT * g.unif(g.re);
        double r2 = g.unif(g.re);
        double r2s = std::sqrt(r2);
        V3 w = normal;
        V3 u = normalize(cross((std::fabs(w.x) > .1 ? V3{0, 1, 0} : V3{1, 0, 0}), w));
        V3 v = cross(w, u);
        V3 rand_dir = normalize(u * std::cos(r1) * r2s + v * std::sin(r1) * r2s + w * std::sqrt(1 - r2));
        double p = 1 / PI_RT, theta = 1;
        V3 brdf = m.kd / PI_RT;
        V3 res = path_host(s, at + normal * 1e-12, rand_dir, 1.0, depth - 1, g);
        colorC = colorC + ((brdf * res * theta / p) * 1.0);
    }
    if (m.spec()) {
        V3 w_in = d;
        V3 normal = i.N;
        if (curIndex != 1.0) normal = normal * -1.0;
        V3 w_normal = dot(w_in, normal) * normal;
        V3 w_tan = w_in - w_normal;
        V3 w_ref = normalize(-w_normal + w_tan);
        V3 temp = path_host(s, at + normal * 1e-12, w_ref, curIndex, depth - 1, g);
        colorC = colorC + ((m.ks * temp) * 1.0);
    }
    if (m.trans()) {
        V3 w_in = d;
        V3 normal = i.N;
        double n1, n2;
        V3 tr{1, 1, 1};
        if (curIndex == 1.0) { n1 = curIndex; n2 = m.ior(); }
        else {
            n1 = m.ior(); n2 = 1.0; normal = normal * -1.0;
            tr = {std::pow(m.kt.x, i.t), std::pow(m.kt.y, i.t), std::pow(m.kt.z, i.t)};
        }
        double r0 = (n1 - n2) / (n1 + n2);
        r0 *= r0;
        double n = n1 / n2;
        w_in = -normalize(w_in);
        double cosI = dot(normal, w_in);
        double cosR = 1 - n * n * (1 - cosI * cosI);
        if (cosR >= 0) {
            double reflCoeff = r0 + (1 - r0) * std::pow((1 - cosI), 5.0);
            double sampleFresnel = g.unbounded(g.re);
            if (sampleFresnel > reflCoeff) {
                double cosT = std::sqrt(cosR);
                V3 refrac = (n * cosI - cosT) * normal - n * w_in;
                V3 temp = path_host(s, at - normal * 1e-12, refrac, n2, depth - 1, g);
                colorC = colorC + ((tr * temp) * 1.0);
            } else {
                V3 w2 = d;
                V3 nn = i.N;
                if (curIndex != 1.0) nn = nn * -1.0;
                V3 w_normal = dot(w2, nn) * nn;
                V3 w_tan = w2 - w_normal;
                V3 w_ref = normalize(-w_normal + w_tan);
                V3 temp = path_host(s, at + nn * 1e-12, w_ref, curIndex, depth - 1, g);
                colorC = colorC + ((m.ks * temp) * 1.0);
            }
        } else {
            V3 w2 = d;
            V3 w_normal = dot(w2, normal) * normal;
            V3 w_tan = w2 - w_normal;
            V3 w_ref = normalize(-w_normal + w_tan);
            V3 temp = path_host(s, at + normal * 1e-12, w_ref, curIndex, depth - 1, g);
            colorC = colorC + ((m.kr * tr * temp) * 1.0);
        }
    }
    return colorC;
}

}  // namespace

// reference: src/RayTracer.cpp:657-694 (pixels strided over threads, i = idx / h, j = idx % h)
//            src/path_tracer.cu:201-238 (jitter U(-.5,.5) with clamping, mean, clamp, (int)(255 c))
// The reference shares one static engine between threads (a data race); here each thread owns one.
void render_cpu_reference(const Scene& s, int w, int h, int spp, int depth, int threads, uint64_t seed,
                          uint8_t* rgb, double* accum_mean) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int tid = 0; tid < threads; ++tid) {
        pool.emplace_back([=, &s]() {
            HostRng g;
            g.re.seed((unsigned)(seed + 7919u * (unsigned)tid + 1u));
            for (int index1d = tid; index1d < w * h; index1d += threads) {
                int i = index1d / h, j = index1d % h;
                V3 colSum{0, 0, 0};
                for (int smp = 0; smp < spp; ++smp) {
                    double iShift = g.jitter(g.re), jShift = g.jitter(g.re);
                    double newI = i + iShift, newJ = j + jShift;
                    if (newI < 0) newI = 0;
                    if (newI >= w) newI = w - 1;
                    if (newJ < 0) newJ = 0;
                    if (newJ >= h) newJ = h - 1;
                    double x = newI / double(w), y = newJ / double(h);
                    V3 o, d;
                    s.cam.ray_through(x, y, o, d);
                    colSum = colSum + path_host(s, o, d, 1.0, depth, g);
                }
                colSum = V3{colSum.x / spp, colSum.y / spp, colSum.z / spp};
                if (accum_mean) {
                    accum_mean[3 * (i + j * w)] = colSum.x;
                    accum_mean[3 * (i + j * w) + 1] = colSum.y;
                    accum_mean[3 * (i + j * w) + 2] = colSum.z;
                }
                auto cl = [](double c) { return c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c); };
                uint8_t* px = rgb + (i + j * w) * 3;
                px[0] = (uint8_t)(int)(255.0 * cl(colSum.x));
                px[1] = (uint8_t)(int)(255.0 * cl(colSum.y));
                px[2] = (uint8_t)(int)(255.0 * cl(colSum.z));
            }
        });
    }
    for (auto& t : pool) t.join();
}

}  // namespace orc
