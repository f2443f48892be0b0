// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.h).
// GPU-semantics traversal, intersection, shading and estimator, restated on the CPU.
#include "oracle.h"

#include <cmath>
#include <cstring>

namespace orc {

// reference: src/gpu/GPUbb.cuh:16-53 (RAY_EPSILON 1e-12 at :12)
static bool slab_gpu(const V3& o, const V3& d, const V3& bmin, const V3& bmax) {
    double tMin = -1.0e308, tMax = 1.0e308;
    for (int a = 0; a < 3; ++a) {
        double vd = comp(d, a);
        if (vd == 0.0) continue;
        double v1 = comp(bmin, a) - comp(o, a);
        double v2 = comp(bmax, a) - comp(o, a);
        double t1 = v1 / vd;
        double t2 = v2 / vd;
        if (t1 > t2) { double tt = t1; t1 = t2; t2 = tt; }
        if (t1 > tMin) tMin = t1;
        if (t2 < tMax) tMax = t2;
        if (tMin > tMax) return false;
        if (tMax < 1e-12) return false;
    }
    return true;
}

// reference: src/gpu/GPUTrimesh.cuh:59-113 (hit decision and t only; the normal is derived
// from the winning hit afterwards -- same values, see shading_normal below)
static bool tri_gpu(const V3& o, const V3& d, const Prim& p, bool recur, double& t_out) {
    double numerator = -dot(o - p.a, p.n);
    double denominator = dot(d, p.n);
    if (denominator >= 0) {
        if (!recur || denominator == 0) return false;
    }
    double t = numerator / denominator;
    if (t < 0) return false;
    V3 P = o + d * t;
    V3 vba = p.b - p.a, vcb = p.c - p.b, vac = p.a - p.c;
    V3 vpa = P - p.a, vpb = P - p.b, vpc = P - p.c;
    bool res = dot(cross(vba, vpa), p.n) >= 0 && dot(cross(vcb, vpb), p.n) >= 0 &&
               dot(cross(vac, vpc), p.n) >= 0;
    if (res) t_out = t;
    return res;
}

// reference: src/gpu/GPUTrimesh.cuh:91-108
static V3 shading_normal(const Scene& s, const Prim& p, const V3& o, const V3& d, double t) {
    const Mesh& me = s.meshes[p.mesh];
    if (!me.use_vertex_normals()) return normalize(p.n);
    V3 P = o + d * t;
    double alpha = length(cross(p.b - P, p.c - P));
    double beta = length(cross(P - p.a, p.c - p.a));
    double denom = length(cross(p.b - p.a, p.c - p.a));
    alpha /= denom;
    beta /= denom;
    V3 na = me.normals[me.face_ids[3 * p.face]];
    V3 nb = me.normals[me.face_ids[3 * p.face + 1]];
    V3 nc = me.normals[me.face_ids[3 * p.face + 2]];
    return normalize(alpha * na + beta * nb + (1 - alpha - beta) * nc);
}

uint64_t visit_mix(uint32_t node_id) {  // splitmix64 finaliser of (id + 1)
    uint64_t z = (uint64_t)node_id + 1u;
    z *= 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// reference: src/gpu/GPUBVH.cu:38-99.  Same control flow: pointer stack, NULL sentinel,
// test L box, test R box, leaf children intersected immediately (L then R, strict '<' keeps the
// earlier hit on equal t), descend L, postpone R.
void trace_gpu(const Scene& s, V3 o, V3 d, TraceOut& out, uint8_t* bitmap) {
    out = TraceOut();
    if (!s.root) return;
    auto mark = [&](const Node* n) {
        out.visit_hash += visit_mix((uint32_t)n->id);
        out.box_tests++;
        if (bitmap) bitmap[n->id >> 3] |= (uint8_t)(1u << (n->id & 7));
    };
    if (s.root->leaf()) {
        // The reference assumes an internal root (it dereferences root->left).  A one-triangle
        // scene is defined here as: test that triangle, visit nothing.
        double t;
        const Prim& p = s.prims[s.root->prim];
        out.tri_tests++;
        if (tri_gpu(o, d, p, s.meshes[p.mesh].mat.recur(), t)) { out.t = t; out.prim = s.root->prim; }
        if (out.prim >= 0) out.N = shading_normal(s, p, o, d, out.t);
        return;
    }
    std::vector<const Node*> stack;
    stack.reserve(64);
    stack.push_back(nullptr);
    const Node* node = s.root;
    double best_t = -1;
    int best = -1;
    do {
        out.node_steps++;
        const Node* L = node->left;
        const Node* R = node->right;
        mark(L);
        bool overlapL = slab_gpu(o, d, L->bmin, L->bmax);
        mark(R);
        bool overlapR = slab_gpu(o, d, R->bmin, R->bmax);
        if (overlapL && L->leaf()) {
            double t;
            const Prim& p = s.prims[L->prim];
            out.tri_tests++;
            if (tri_gpu(o, d, p, s.meshes[p.mesh].mat.recur(), t))
                if (best_t < 0 || t < best_t) { best_t = t; best = L->prim; }
        }
        if (overlapR && R->leaf()) {
            double t;
            const Prim& p = s.prims[R->prim];
            out.tri_tests++;
            if (tri_gpu(o, d, p, s.meshes[p.mesh].mat.recur(), t))
                if (best_t < 0 || t < best_t) { best_t = t; best = R->prim; }
        }
        bool travL = overlapL && !L->leaf();
        bool travR = overlapR && !R->leaf();
        if (!travL && !travR) {
            node = stack.back();
            stack.pop_back();
        } else {
            node = travL ? L : R;
            if (travL && travR) {
                stack.push_back(R);
                if ((int)stack.size() - 1 > out.max_stack) out.max_stack = (int)stack.size() - 1;
            }
        }
    } while (node != nullptr);
    out.t = best_t;
    out.prim = best_t >= 0 ? best : -1;
    if (out.prim >= 0) out.N = shading_normal(s, s.prims[out.prim], o, d, out.t);
}

// ------------------------------------------------------------------------------------------
// Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// Replaces the reference's per-pixel XORWOW stream (curand_init(1984, pixel, 0), path_tracer.cu:412).
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Uniform double in (0,1] (the convention of curand_uniform_double, path_tracer.cu:282).
double u01_open0(uint32_t lo, uint32_t hi) {
    uint64_t bits = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)(bits + 1) * (1.0 / 9007199254740992.0);
}

namespace {
// Random-number addressing: counter = (pixel, sample, bounce, stream), key = seed.
// stream 0 -> (lobe choice, Fresnel choice); stream 1 -> (cosine r1, r2); stream 2 -> (roulette, -).
// The camera draw uses bounce = 0xFFFFFFFF, stream 0 -> (iShift, jShift).
struct Draw { double a, b; };
Draw draw2(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream) {
    uint32_t ctr[4] = {pixel, sample, bounce, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t r[4];
    philox4x32_10(ctr, key, r);
    return {u01_open0(r[0], r[1]), u01_open0(r[2], r[3])};
}
const double PI_RT = 3.14159265358979311600;  // src/RayTracer.h:5

// reference: src/path_tracer.cu:242-253
V3 cosine_hemisphere(const V3& normal, double xa, double xb) {
    double r1 = 2 * PI_RT * xa;
    double r2 = xb;
    double r2s = std::sqrt(r2);
    V3 w = normal;
    V3 u = normalize(cross((std::fabs(w.x) > .1 ? V3{0, 1, 0} : V3{1, 0, 0}), w));
    V3 v = cross(w, u);
    return normalize(u * std::cos(r1) * r2s + v * std::sin(r1) * r2s + w * std::sqrt(1 - r2));
}
V3 vpow(V3 b, double e) { return {std::pow(b.x, e), std::pow(b.y, e), std::pow(b.z, e)}; }
}  // namespace

// reference: src/path_tracer.cu:255-400 (loop), :403-444 / :446-492 (camera ray per sample);
// termination test as in src/improved_path_tracer.cu:80 (Diff || Spec || Trans).
V3 path_radiance_gpu(const Scene& s, const RenderParams& p, int i, int j, int sample, RenderStats* st) {
    const uint32_t pixel = (uint32_t)(i + j * p.w);
    V3 o, d;
    Draw jit = draw2(p.seed, pixel, (uint32_t)sample, 0xFFFFFFFFu, 0);
    if (!p.dof) {
        double newI = i + jit.a, newJ = j + jit.b;
        double x = newI / double(p.w), y = newJ / double(p.h);
        s.cam.ray_through(x, y, o, d);
    } else {  // path_tracer.cu:462-479, including the one-sided (0, aperture] lens sample
        V3 to, td;
        s.cam.ray_through(double(i) / double(p.w), double(j) / double(p.h), to, td);
        V3 focal = to + (p.focal_distance * td);
        double iShift = jit.a, jShift = jit.b;
        iShift *= (2 * p.aperture) - p.aperture;
        jShift *= (2 * p.aperture) - p.aperture;
        V3 eye = s.cam.eye + s.cam.u * iShift + s.cam.v * jShift;
        o = eye;
        d = normalize(focal - eye);
    }
    double curIndex = 1.0;
    V3 colorC{0, 0, 0}, curAtten{1, 1, 1};
    if (st) st->paths++;
    for (int bounce = 0; bounce <= p.depth; bounce++) {
        if (p.rr && bounce >= (p.depth < 3 ? p.depth : 3)) {  // path_tracer.cu:38-46 (disabled there)
            double probRR = 0.30;
            Draw rr = draw2(p.seed, pixel, (uint32_t)sample, (uint32_t)bounce, 2);
            if (rr.a >= probRR) break;
            curAtten = curAtten * (1 / probRR);
        }
        TraceOut h;
        trace_gpu(s, o, d, h, nullptr);
        if (st) { st->segments++; st->node_steps += h.node_steps; st->tri_tests += h.tri_tests; }
        if (h.prim < 0) break;  // the reference re-traverses the same missing ray; same result
        if (st) st->hits++;
        const Material& m = s.meshes[s.prims[h.prim].mesh].mat;
        colorC = colorC + m.ke * curAtten;
        if (!(m.diff() || m.spec() || m.trans())) break;
        double spec = (m.ks.x + m.ks.y + m.ks.z) / 3.0;
        double diff = (m.kd.x + m.kd.y + m.kd.z) / 3.0;
        double trans = (m.kt.x + m.kt.y + m.kt.z) / 3.0;
        double total = spec + diff + trans;
        double diffProb = diff / total, specProb = spec / total, transProb = trans / total;
        Draw sel = draw2(p.seed, pixel, (uint32_t)sample, (uint32_t)bounce, 0);
        double prob = sel.a;
        V3 at = o + (h.t * d);
        if (prob < diffProb) {
            V3 normal = normalize(h.N);
            Draw cs = draw2(p.seed, pixel, (uint32_t)sample, (uint32_t)bounce, 1);
            V3 rand_dir = cosine_hemisphere(normal, cs.a, cs.b);
            const double pdf = 1 / PI_RT, theta = 1;
            o = at + normal * 1e-12;
            d = rand_dir;
            curIndex = 1.0;
            V3 brdf = m.kd / PI_RT;
            curAtten = curAtten * (brdf * theta / (pdf * diffProb));
        } else if (prob < diffProb + specProb) {
            V3 w_in = normalize(d);
            V3 normal = normalize(h.N);
            if (curIndex != 1.0) normal = normal * -1.0;
            V3 w_normal = dot(w_in, normal) * normal;
            V3 w_tan = w_in - w_normal;
            V3 w_ref = normalize(-w_normal + w_tan);
            o = at + normal * 1e-12;
            d = w_ref;
            curIndex = 1.0;  // path_tracer.cu:306: the Ray constructor resets the medium
            curAtten = curAtten * (m.ks * 1.0 / specProb);
        } else {
            V3 w_in = d;
            V3 normal = h.N;
            double n1, n2;
            V3 tr{1, 1, 1};
            if (curIndex == 1.0) { n1 = curIndex; n2 = m.ior(); }
            else { n1 = m.ior(); n2 = 1.0; normal = normal * -1.0; tr = vpow(m.kt, h.t); }
            double r0 = (n1 - n2) / (n1 + n2);
            r0 *= r0;
            double n = n1 / n2;
            w_in = -normalize(w_in);
            double cosI = dot(normal, w_in);
            double cosR = 1 - n * n * (1 - cosI * cosI);
            if (cosR >= 0) {
                double reflCoeff = r0 + (1 - r0) * std::pow((1 - cosI), 5.0);
                double sampleFresnel = sel.b;
                if (sampleFresnel > reflCoeff) {
                    double cosT = std::sqrt(cosR);
                    V3 refrac = (n * cosI - cosT) * normal - n * w_in;
                    o = at - normal * 1e-12;
                    d = refrac;
                    curIndex = n2;
                    curAtten = curAtten * tr;
                } else {
                    V3 w2 = d;
                    V3 nn = h.N;
                    if (curIndex != 1.0) nn = nn * -1.0;
                    V3 w_normal = dot(w2, nn) * nn;
                    V3 w_tan = w2 - w_normal;
                    V3 w_ref = normalize(-w_normal + w_tan);
                    o = at + nn * 1e-12;
                    d = w_ref;  // medium index kept
                    curAtten = curAtten * m.ks;
                }
            } else {
                V3 w2 = d;
                V3 w_normal = dot(w2, normal) * normal;
                V3 w_tan = w2 - w_normal;
                V3 w_ref = normalize(-w_normal + w_tan);
                o = at + normal * 1e-12;
                d = w_ref;  // medium index kept
                curAtten = curAtten * (m.kr * tr);
            }
            curAtten = curAtten * (1 / transProb);
        }
    }
    return colorC;
}

void render_gpu_semantics(const Scene& s, const RenderParams& p, double* accum, RenderStats* st) {
    for (int j = 0; j < p.h; ++j)
        for (int i = 0; i < p.w; ++i) {
            V3 sum{accum[3 * (i + j * p.w)], accum[3 * (i + j * p.w) + 1], accum[3 * (i + j * p.w) + 2]};
            for (int smp = p.s0; smp < p.s1; ++smp) sum = sum + path_radiance_gpu(s, p, i, j, smp, st);
            accum[3 * (i + j * p.w)] = sum.x;
            accum[3 * (i + j * p.w) + 1] = sum.y;
            accum[3 * (i + j * p.w) + 2] = sum.z;
        }
}

}  // namespace orc
