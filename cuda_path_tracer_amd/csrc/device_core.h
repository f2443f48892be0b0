// Device-side building blocks of the path tracer for gfx950: slab test, triangle test,
// stack traversal with the stack in LDS, counter RNG, camera ray, one bounce of shading.
// Every kernel in kernels.hip is assembled from these, so the wavefront pipeline, the
// one-thread-per-pixel kernel and the parity hooks compute bit-identical values.
//
// All arithmetic is binary64 with contraction off (the translation unit is compiled with
// -ffp-contract=off); divisions and square roots are the correctly rounded ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpt_amd.h"
#include "dvec.h"
#include "launch.h"

#define CPT_WAVE 64
#define CPT_STACK_ENTRIES 32  // per-lane traversal stack entries held in LDS

// Per-ray observation record filled only by the parity hooks.
struct TraceRec {
    uint64_t visit_hash;
    int node_steps, box_tests, tri_tests, max_stack;
    uint8_t* bitmap;  // nullable
};

__device__ __forceinline__ uint64_t visit_mix(uint32_t id) {
    uint64_t z = (uint64_t)id + 1u;
    z *= 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Kay-Kajiya slab test with the reference's exact sequence of divisions and comparisons
// (reference: src/gpu/GPUbb.cuh:16-53, epsilon 1e-12 at :12).  The reference returns from inside
// the axis loop; a failed axis can never be undone by a later one, so the axes are evaluated
// without branches and the verdicts are combined.
__device__ __forceinline__ bool slab_hit(double lx, double ly, double lz, double hx, double hy, double hz,
                                         const dvec3& o, const dvec3& d, double& t_enter) {
    double tMin = -1.0e308, tMax = 1.0e308;
    bool ok = true;
#define CPT_SLAB_AXIS(LO, HI, O, D)                         \
    {                                                       \
        double vd = (D);                                    \
        double t1 = ((LO) - (O)) / vd;                      \
        double t2 = ((HI) - (O)) / vd;                      \
        double tn = (t1 > t2) ? t2 : t1;                    \
        double tf = (t1 > t2) ? t1 : t2;                    \
        bool live = (vd != 0.0);                            \
        double nMin = (tn > tMin) ? tn : tMin;              \
        double nMax = (tf < tMax) ? tf : tMax;              \
        tMin = live ? nMin : tMin;                          \
        tMax = live ? nMax : tMax;                          \
        ok = ok && !(live && ((tMin > tMax) || (tMax < 1e-12))); \
    }
    CPT_SLAB_AXIS(lx, hx, o.x, d.x)
    CPT_SLAB_AXIS(ly, hy, o.y, d.y)
    CPT_SLAB_AXIS(lz, hz, o.z, d.z)
#undef CPT_SLAB_AXIS
    t_enter = tMin;
    return ok;
}
__device__ __forceinline__ bool slab_hit(double lx, double ly, double lz, double hx, double hy, double hz,
                                         const dvec3& o, const dvec3& d) {
    double unused;
    return slab_hit(lx, ly, lz, hx, hy, hz, o, d, unused);
}

// The same verdict as slab_hit, usually without dividing.  With r = 1/d rounded once per ray,
// a = (plane - o) * r differs from the reference's rounded quotient t = (plane - o) / d by a relative 4.1 * 2^-53 at
// most (three roundings, one more for a reciprocal that is within one ulp; the subtraction is the reference's own, so
// there is no absolute error term and rays that start on a box plane are decided like any other).
//
// The reference's verdict is "every axis' near t <= every axis' far t, and the smallest far t is
// >= 1e-12" (its early returns can never be undone by a later axis).  On one axis near <= far always
// holds because rounding is monotonic, and a box that is flat on an axis has near == far there, so
// only the six cross-axis orderings may be judged by their distance from equality:
//   G = min over axes k of (far_k - largest near of the other two axes).
// With TF the smallest far product and M = 4e-15 |TF|:
//   hit   is certain when  G >  M  and TF > 1e-12 (1 + 1e-15)
//   miss  is certain when  G < -M  or  TF < 1e-12 (1 - 1e-15)
// One margin serves all six orderings, and it needs TF only.  The epsilon test alone says miss unless TF > 0, so let
// TF > 0 (then every far is positive, in exact arithmetic as well: a product has its quotient's sign).
// G > M: on TF's axis j the far is above every other axis' near by G and above its own near by monotonicity, so no near
// exceeds TF.  An ordering far_k >= near_i with near_i <= 0 holds in exact arithmetic by the signs alone; one with
// 0 < near_i <= TF <= far_k has far_k - near_i >= G > 4e-15 near_i, and a difference x > 4e-15 near_i clears the
// 4.6e-16 (2 near_i + x) that the two products can be off by together (the roundings of the difference and of M are
// relative 2^-53 of numbers that have four times that room).
// G < -M: the largest near TN lies on another axis than TF (on TF's own axis every near is <= TF and G would be >= 0),
// so G = TF - TN and TN - TF > 4e-15 TF clears the same bound: that ordering is violated in exact arithmetic too.
// (M needs no floor: it only decides where TF >= 1e-12 (1 - 1e-15), and there it is at least 4e-27; products down in the
// subnormal range, whose error is absolute, are nears that far below every far that counts.)  Anything else is "not sure" and the caller redoes the box with
// slab_hit.  Infinite intermediates end up as "not sure"; the caller must not use this test at all
// for a ray with a zero direction component (the reference skips such an axis, which the products
// cannot express).
__device__ __forceinline__ double fast_reciprocal(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
#else
    return 1.0 / x;
#endif
}
struct SlabRay {
    dvec3 r;          // 1/d
    bool exact_only;  // the products cannot stand in for this ray's quotients at all
};
__device__ __forceinline__ SlabRay slab_prepare(const dvec3& o, const dvec3& d) {
    SlabRay s;
    // The reciprocal need not be the correctly rounded one: the margin of the test (slab_verdict) has four times the
    // room that the products of a reciprocal within one ulp need, and a correctly rounded one would only take a quarter
    // off that need.  The hardware estimate refined twice by Newton's step (error below one ulp) costs five
    // instructions where the IEEE division takes thirteen, three times per ray.  Zero, infinite and NaN components
    // come out infinite or NaN and send the ray down the exact path, as before.
    s.r = dv(fast_reciprocal(d.x), fast_reciprocal(d.y), fast_reciprocal(d.z));
    // also rules out NaN products: box planes are finite (checked at upload), so a product can only
    // be NaN through an infinite reciprocal or a non-finite origin
    s.exact_only = !(fabs(s.r.x) < 1e300 && fabs(s.r.y) < 1e300 && fabs(s.r.z) < 1e300 &&
                     fabs(o.x) < 1e300 && fabs(o.y) < 1e300 && fabs(o.z) < 1e300);
    return s;
}
// The part of the test that follows the products: n* / f* are each axis' near and far product.
__device__ __forceinline__ bool slab_verdict(double nx, double ny, double nz, double fx, double fy, double fz, bool& sure,
                                             double& t_enter) {
    double cx = fmax(ny, nz), cy = fmax(nx, nz), cz = fmax(nx, ny);
    double tf = fmin(fx, fmin(fy, fz));
    double g = fmin(fx - cx, fmin(fy - cy, fz - cz));
    double m = fabs(tf) * 4e-15;
    bool hit = (g > m) && (tf > 1.000000000000001e-12);
    sure = hit || (g < -m) || (tf < 0.999999999999999e-12);
    t_enter = fmax(cz, nz);  // only the culling walks look at it
    return hit;
}
__device__ __forceinline__ bool slab_filtered(double lx, double ly, double lz, double hx, double hy, double hz,
                                              const dvec3& o, const SlabRay& s, bool& sure, double& t_enter) {
    double x1 = (lx - o.x) * s.r.x, x2 = (hx - o.x) * s.r.x;
    double y1 = (ly - o.y) * s.r.y, y2 = (hy - o.y) * s.r.y;
    double z1 = (lz - o.z) * s.r.z, z2 = (hz - o.z) * s.r.z;
    return slab_verdict(fmin(x1, x2), fmin(y1, y2), fmin(z1, z2), fmax(x1, x2), fmax(y1, y2), fmax(z1, z2), sure, t_enter);
}
__device__ __forceinline__ bool slab_filtered(double lx, double ly, double lz, double hx, double hy, double hz,
                                              const dvec3& o, const SlabRay& s, bool& sure) {
    double unused;
    return slab_filtered(lx, ly, lz, hx, hy, hz, o, s, sure, unused);
}
// The same test for a caller that already knows, per axis, which of the box's two planes the ray meets first: `near`
// holds the lower plane of every axis on which the ray moves up (r > 0) and the upper plane of the others, `far` the
// opposite ones.  Rounding is monotonic, so on an axis with r > 0 the lower plane's product is the smaller one (or they
// are equal), with r < 0 the upper plane's: the minima and maxima of slab_filtered select exactly these values and need
// not be taken.  (Which plane goes where is immaterial to slab_hit, whose swap is symmetric, and to a ray with a
// non-finite reciprocal, which is sent there.)
__device__ __forceinline__ bool slab_selected(double near_x, double near_y, double near_z, double far_x, double far_y, double far_z,
                                              const dvec3& o, const SlabRay& s, bool& sure, double& t_enter) {
    return slab_verdict((near_x - o.x) * s.r.x, (near_y - o.y) * s.r.y, (near_z - o.z) * s.r.z,
                        (far_x - o.x) * s.r.x, (far_y - o.y) * s.r.y, (far_z - o.z) * s.r.z, sure, t_enter);
}

// Ray/triangle test (reference: src/gpu/GPUTrimesh.cuh:59-113): plane hit, back-face cull unless
// the material is reflective/transmissive, t >= 0, three edge tests.  Returns t through t_out.
// The plane term of the test, -dot(o - a, n) (src/gpu/GPUTrimesh.cuh:62): a function of the ray's origin and the triangle alone.
__device__ __forceinline__ double tri_numerator(const dvec3& o, const dvec3& a, const dvec3& n) { return -dot(o - a, n); }
// `numerator` is tri_numerator(o, a, n), worked out by the caller: per test, or once per triangle for rays that share an origin.
__device__ __forceinline__ bool tri_hit_regs(double2 q0, double2 q1, double2 q2, double2 q3, double2 q4, double2 q5,
                                             uint32_t flags, double numerator, const dvec3& o, const dvec3& d, double& t_out) {
    dvec3 a = dv(q0.x, q0.y, q1.x), b = dv(q1.y, q2.x, q2.y), c = dv(q3.x, q3.y, q4.x), n = dv(q4.y, q5.x, q5.y);
    bool recur = (flags & CPT_TRI_RECUR) != 0;
    double denominator = dot(d, n);
    if (denominator >= 0) {
        if (!recur || denominator == 0) return false;
    }
    double t = numerator / denominator;
    if (t < 0) return false;
    dvec3 P = o + d * t;
    dvec3 vba = b - a, vcb = c - b, vac = a - c;
    dvec3 vpa = P - a, vpb = P - b, vpc = P - c;
    bool res = dot(cross(vba, vpa), n) >= 0 && dot(cross(vcb, vpb), n) >= 0 && dot(cross(vac, vpc), n) >= 0;
    t_out = t;
    return res;
}
__device__ __forceinline__ bool tri_hit_regs(double2 q0, double2 q1, double2 q2, double2 q3, double2 q4, double2 q5,
                                             uint32_t flags, const dvec3& o, const dvec3& d, double& t_out) {
    return tri_hit_regs(q0, q1, q2, q3, q4, q5, flags, tri_numerator(o, dv(q0.x, q0.y, q1.x), dv(q4.y, q5.x, q5.y)), o, d, t_out);
}
__device__ __forceinline__ bool tri_hit(const cpt_tri* __restrict__ tp, const dvec3& o, const dvec3& d,
                                        double& t_out) {
    const double2* q = (const double2*)tp;  // 128-byte aligned record: a, b, c, n, material, flags
    double2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
    uint2 mf = ((const uint2*)tp)[12];
    return tri_hit_regs(q0, q1, q2, q3, q4, q5, mf.y, o, d, t_out);
}

// Closest hit over the scene, visiting exactly the nodes the reference's stack traversal visits
// (reference: src/gpu/GPUBVH.cu:38-99): both child boxes of every expanded node are tested, leaf
// children are intersected, the left child is descended and the right one postponed.  There is
// no near/far ordering and no culling by the current closest hit, so the set of tested nodes is a
// property of the ray alone.  On equal t the reference keeps the leaf it tested first; its test
// order is (expansion order of the parent, left before right), a total order that does not depend
// on the ray; every triangle record carries that position as test_rank, so the rule is "smaller
// test_rank wins" whatever order we walk in.
//
// `stack` points at this lane's column of an LDS array laid out [entry][lane]: entry e of lane l
// lives at stack[e * CPT_WAVE], which puts the 64 lanes of a wave on distinct banks.
template <bool TRACE>
__device__ __forceinline__ void traverse(const DScene& sc, const dvec3& o, const dvec3& d, uint32_t* stack,
                                         double& best_t, int& best_tri, TraceRec* rec) {
    best_t = -1.0;
    best_tri = -1;
    uint32_t best_rank = 0xFFFFFFFFu;
    if (sc.n_nodes == 0) {  // one-triangle scene: the reference would dereference a null child
        if (sc.n_tris > 0) {
            double t;
            if (TRACE) rec->tri_tests++;
            if (tri_hit(sc.tris, o, d, t)) { best_t = t; best_tri = 0; }
        }
        return;
    }
    int sp = 0;
    int node = 0;
    const SlabRay sr = slab_prepare(o, d);
    for (;;) {
        const double2* p = (const double2*)(sc.nodes + node);
        double2 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3], b4 = p[4], b5 = p[5];
        int4 refs = ((const int4*)p)[6];  // lref, rref, lid, rid
        bool sure_l, sure_r;
        bool hl = slab_filtered(b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, o, sr, sure_l);
        bool hr = slab_filtered(b3.x, b3.y, b4.x, b4.y, b5.x, b5.y, o, sr, sure_r);
        if (sr.exact_only || !(sure_l && sure_r)) {  // rare: a comparison too close to call without the exact quotients
            hl = slab_hit(b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, o, d);
            hr = slab_hit(b3.x, b3.y, b4.x, b4.y, b5.x, b5.y, o, d);
        }
        if (TRACE) {
            rec->node_steps++;
            rec->box_tests += 2;
            rec->visit_hash += visit_mix((uint32_t)refs.z) + visit_mix((uint32_t)refs.w);
            if (rec->bitmap) {
                rec->bitmap[refs.z >> 3] |= (uint8_t)(1u << (refs.z & 7));
                rec->bitmap[refs.w >> 3] |= (uint8_t)(1u << (refs.w & 7));
            }
        }
        if (hl && refs.x < 0) {
            double t;
            if (TRACE) rec->tri_tests++;
            if (tri_hit(sc.tris + (~refs.x), o, d, t)) {
                uint32_t rank = sc.tris[~refs.x].test_rank;
                if (best_t < 0 || t < best_t || (t == best_t && rank < best_rank)) { best_t = t; best_tri = ~refs.x; best_rank = rank; }
            }
        }
        if (hr && refs.y < 0) {
            double t;
            if (TRACE) rec->tri_tests++;
            if (tri_hit(sc.tris + (~refs.y), o, d, t)) {
                uint32_t rank = sc.tris[~refs.y].test_rank;
                if (best_t < 0 || t < best_t || (t == best_t && rank < best_rank)) { best_t = t; best_tri = ~refs.y; best_rank = rank; }
            }
        }
        bool tl = hl && refs.x >= 0;
        bool tr = hr && refs.y >= 0;
        if (!tl && !tr) {
            if (sp == 0) break;
            --sp;
            node = (int)stack[sp * CPT_WAVE];
        } else {
            node = tl ? refs.x : refs.y;
            if (tl && tr) {
                stack[sp * CPT_WAVE] = (uint32_t)refs.y;
                ++sp;
                if (TRACE && sp > rec->max_stack) rec->max_stack = sp;
            }
        }
    }
}

// Shading normal of a hit (reference: src/gpu/GPUTrimesh.cuh:91-108).  The reference computes the
// barycentric weights inside every triangle test; only the winning hit's are ever used, so they
// are computed once here from the same expressions.
__device__ __forceinline__ dvec3 shading_normal(const ShadeRec* tp, const dvec3& o, const dvec3& d, double t) {
    if (!(tp->tri_flags & CPT_TRI_VERTEX_NORMALS)) return dv(tp->n1);  // normalize(face normal), from the record
    dvec3 a = dv(tp->a), b = dv(tp->b), c = dv(tp->c);
    dvec3 P = o + d * t;
    double alpha = length(cross(b - P, c - P));
    double beta = length(cross(P - a, dv(tp->ca)));  // c - a and the denominator do not depend on the ray: from the record (k_build_shade)
    double denom = tp->denom;
    alpha /= denom;
    beta /= denom;
    return normalize(alpha * dv(tp->na) + beta * dv(tp->nb) + (1 - alpha - beta) * dv(tp->nc));
}

// ---- counter RNG -----------------------------------------------------------------------
// Philox-4x32-10 keyed by the seed and addressed by (pixel, sample, bounce, stream): any path's
// numbers can be produced anywhere, in any order, on any GPU.  It stands in for the reference's
// per-pixel XORWOW state (curand_init(1984, pixel, 0), reference: src/path_tracer.cu:412).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of separate low and high halves
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// 53 random bits -> (0,1], the range of curand_uniform_double (reference: src/path_tracer.cu:282)
__device__ __forceinline__ double u01_open0(uint32_t lo, uint32_t hi) {
    uint64_t bits = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)(bits + 1) * (1.0 / 9007199254740992.0);
}
#define CPT_STREAM_SELECT 0u   // (lobe choice, Fresnel choice)
#define CPT_STREAM_COSINE 1u   // (r1, r2) of the cosine-weighted direction
#define CPT_STREAM_ROULETTE 2u // (survival, -)
#define CPT_BOUNCE_CAMERA 0xFFFFFFFFu
__device__ __forceinline__ void draw2(const DParams& P, uint32_t pixel, uint32_t sample, uint32_t bounce,
                                      uint32_t stream, double& a, double& b) {
    uint32_t r[4];
    philox4x32_10(pixel, sample, bounce, stream, P.seed_lo, P.seed_hi, r);
    a = u01_open0(r[0], r[1]);
    b = u01_open0(r[2], r[3]);
}

// ---- camera ----------------------------------------------------------------------------
// reference: src/gpu/GPUCamera.cu:22-32
__device__ __forceinline__ void camera_ray(const cpt_camera& cam, double x, double y, dvec3& o, dvec3& d) {
    x -= 0.5;
    y -= 0.5;
    // element by element: taking the address of a by-value kernel argument would send it through scratch memory
    const dvec3 look = dv(cam.look[0], cam.look[1], cam.look[2]), u = dv(cam.u[0], cam.u[1], cam.u[2]), v = dv(cam.v[0], cam.v[1], cam.v[2]);
    d = normalize(look + x * u + y * v);
    o = dv(cam.eye[0], cam.eye[1], cam.eye[2]);
}

// First ray of sample `sample` of pixel (i, j).
// reference: src/path_tracer.cu:419-431 (pin-hole, jitter in (0,1]); :462-479 (thin lens, keeping
// the reference's lens offset range (0, aperture] on both axes).
__device__ __forceinline__ void first_ray(const DScene& sc, const DParams& P, int i, int j, int sample, dvec3& o,
                                          dvec3& d) {
    double ja, jb;
    draw2(P, (uint32_t)(i + j * P.w), (uint32_t)sample, CPT_BOUNCE_CAMERA, CPT_STREAM_SELECT, ja, jb);
    if (!P.dof) {
        double newI = i + ja, newJ = j + jb;
        camera_ray(sc.cam, newI / double(P.w), newJ / double(P.h), o, d);
    } else {
        dvec3 to, td;
        camera_ray(sc.cam, double(i) / double(P.w), double(j) / double(P.h), to, td);
        dvec3 focal = to + (P.focal_distance * td);
        double iShift = ja, jShift = jb;
        iShift *= (2 * P.aperture) - P.aperture;
        jShift *= (2 * P.aperture) - P.aperture;
        const cpt_camera& cam = sc.cam;
        dvec3 eye = dv(cam.eye[0], cam.eye[1], cam.eye[2]) + dv(cam.u[0], cam.u[1], cam.u[2]) * iShift + dv(cam.v[0], cam.v[1], cam.v[2]) * jShift;
        o = eye;
        d = normalize(focal - eye);
    }
}

// ---- one bounce --------------------------------------------------------------------------

// reference: src/path_tracer.cu:242-253
__device__ __forceinline__ dvec3 cosine_hemisphere(const dvec3& normal, double xa, double xb) {
    double r1 = 2 * CPT_PI * xa;
    double r2 = xb;
    double r2s = sqrt(r2);
    dvec3 w = normal;
    dvec3 u = normalize(cross((fabs(w.x) > .1 ? dv(0, 1, 0) : dv(1, 0, 0)), w));
    dvec3 v = cross(w, u);
    return normalize(u * cos(r1) * r2s + v * sin(r1) * r2s + w * sqrt(1 - r2));
}

struct PathRegs {  // a path's live state between bounces
    dvec3 o, d;      // current ray
    double medium;   // refractive index of the medium the ray travels in (1.0 = air)
    dvec3 atten;     // throughput
    dvec3 color;     // radiance gathered so far
};

// Russian roulette before bounce `bounce` (off in the reference, which has it commented out at
// src/path_tracer.cu:38-46: survive with probability 0.30 once min(3, depth) bounces are done).
// Returns false when the path dies.
__device__ __forceinline__ bool roulette(const DParams& P, uint32_t pixel, uint32_t sample, int bounce, PathRegs& s) {
    if (!P.rr || bounce < (P.depth < 3 ? P.depth : 3)) return true;
    const double probRR = 0.30;
    double xa, xb;
    draw2(P, pixel, sample, (uint32_t)bounce, CPT_STREAM_ROULETTE, xa, xb);
    if (xa >= probRR) return false;
    s.atten = s.atten * (1 / probRR);
    return true;
}

// One bounce is split in two so the wavefront pipeline can run each half as its own kernel: first
// the part every hit goes through (emission, termination, lobe choice), then one of three scatter
// routines.  shade_bounce below chains them for the callers that keep a path in registers, so
// every path sees the same arithmetic.
// reference: src/path_tracer.cu:263-397; termination test as src/improved_path_tracer.cu:80.
#define CPT_LOBE_NONE (-1)
#define CPT_LOBE_DIFFUSE 0
#define CPT_LOBE_MIRROR 1
#define CPT_LOBE_DIELECTRIC 2

// Emission and lobe choice for a hit on triangle `tri` (tri < 0: the ray missed).  `emitted` receives
// ke * throughput, which the caller adds to the path's radiance; returns the lobe to scatter with, or
// CPT_LOBE_NONE when the path ends.
__device__ __forceinline__ int choose_lobe(const DScene& sc, const DParams& P, uint32_t pixel, uint32_t sample, int bounce,
                                           int tri, const dvec3& atten, dvec3& emitted) {
    emitted = dv(0, 0, 0);
    if (tri < 0) return CPT_LOBE_NONE;  // the reference re-intersects the same missing ray until the loop ends
    const ShadeRec* m = sc.shade + tri;
    emitted = dv(m->ke) * atten;
    if (!(m->mat_flags & (CPT_MAT_DIFF | CPT_MAT_SPEC | CPT_MAT_TRANS))) return CPT_LOBE_NONE;
    if (bounce >= P.depth) return CPT_LOBE_NONE;  // the new ray would never be traced
    double prob, unused;
    draw2(P, pixel, sample, (uint32_t)bounce, CPT_STREAM_SELECT, prob, unused);
    if (prob < m->p_diff) return CPT_LOBE_DIFFUSE;
    if (prob < m->p_diff + m->p_spec) return CPT_LOBE_MIRROR;
    return CPT_LOBE_DIELECTRIC;
}

// pow(x, y) for x < 0 is +-|x|^y when y is an integer (minus for an odd one) and NaN otherwise: the factor
// that |x|^y is to be multiplied by.
__device__ __forceinline__ double pow_negative_base(double y) {
    if (y != rint(y)) return __builtin_nan("");
    return (rint(y * 0.5) * 2.0 != y) ? -1.0 : 1.0;
}

// |kt|^t as exp(t * log|kt|), log|kt| given.  Where the product is NaN although neither factor is (t = 0 with
// kt = 0, t infinite with |kt| = 1) pow's value is 1.
__device__ __forceinline__ double kt_power(double t, double log_kt) {
    const double e = t * log_kt;
    const double v = exp(e);
    return (e != e && (t == 0.0 || log_kt == 0.0)) ? 1.0 : v;
}

// Scatter with the chosen lobe: replace the ray, scale the throughput.
// reference: src/path_tracer.cu:284-396.
//
// The three lobes are written as one routine so that a wave whose lanes chose different lobes runs
// the expensive common pieces once: the shading normal, and the mirror-reflection block that the
// mirror lobe, the dielectric lobe's Fresnel reflection and total internal reflection all end in
// (each with its own incoming direction, normal and throughput factor, exactly as the reference
// writes them).
__device__ __forceinline__ void scatter(int lobe, const DScene& sc, const DParams& P, uint32_t pixel, uint32_t sample,
                                        int bounce, double t, int tri, PathRegs& s) {
    const ShadeRec* m = sc.shade + tri;
    const bool flat = !(m->tri_flags & CPT_TRI_VERTEX_NORMALS);
    const dvec3 N = shading_normal(m, s.o, s.d, t);
    // normalize(N) as the diffuse and mirror lobes take it: for a flat triangle that is in the record too
    const dvec3 NN = flat ? dv(m->n2) : normalize(N);
    const dvec3 at = s.o + (t * s.d);

    bool reflects = false;        // ends in the reflection block below
    dvec3 rw = s.d, rn = N;       // incoming direction and normal the reflection uses
    dvec3 f1 = dv(1, 1, 1);       // throughput factor of the reflection
    bool lobe_weight = false;     // dielectric outcomes are also divided by the lobe's probability
    bool reset_medium = false;

    if (lobe == CPT_LOBE_DIFFUSE) {  // :284-293
        dvec3 normal = NN;
        double xa, xb;
        draw2(P, pixel, sample, (uint32_t)bounce, CPT_STREAM_COSINE, xa, xb);
        dvec3 rand_dir = cosine_hemisphere(normal, xa, xb);
        s.o = at + normal * 1e-12;
        s.d = rand_dir;
        s.medium = 1.0;
        s.atten = s.atten * dv(m->diff_w);  // brdf * theta / (pdf * p_diff), brdf = kd / pi, pdf = 1 / pi, theta = 1 (:289-292)
    } else if (lobe == CPT_LOBE_MIRROR) {  // :294-308
        rw = normalize(s.d);
        rn = NN;
        if (s.medium != 1.0) rn = rn * -1.0;
        f1 = dv(m->spec_w);  // ks * 1.0 / p_spec
        reset_medium = true;  // the reference builds a fresh Ray here, which resets the medium (:306)
        reflects = true;
    } else {  // :310-396
        dvec3 kt = dv(m->kt);
        dvec3 w_in = s.d;
        dvec3 normal = N;
        double n1, n2;
        dvec3 tr = dv(1, 1, 1);
        if (s.medium == 1.0) {
            n1 = s.medium;
            n2 = m->index;
        } else {
            n1 = m->index;
            n2 = 1.0;
            normal = normal * -1.0;
            // kt^t as exp(t log|kt|) with log|kt| from the shading record: a few ulp from pow (relative error
            // about |t log kt| ulp), a quarter of its instructions and none of its registers.  kt = 0 gives
            // exp(-inf) = 0 like pow; a negative kt follows pow's rule for a negative base (pow_negative_base).
            // t is a hit distance (>= 0, finite on finite scenes); the two products that pow defines but the
            // exponential form turns into NaN -- 0 * log 0 (pow(0, 0) = 1) and inf * log 1 (pow(1, inf) = 1) -- are put
            // right by kt_power.
            const dvec3 lk = dv(m->log_kt);
            tr.x = kt_power(t, lk.x);
            tr.y = (kt.y == kt.x) ? tr.x : kt_power(t, lk.y);  // equal inputs give equal results: computed once
            tr.z = (kt.z == kt.x) ? tr.x : ((kt.z == kt.y) ? tr.y : kt_power(t, lk.z));
            if (kt.x < 0 || kt.y < 0 || kt.z < 0) {
                const double sign = pow_negative_base(t);
                if (kt.x < 0) tr.x = sign * tr.x;
                if (kt.y < 0) tr.y = sign * tr.y;
                if (kt.z < 0) tr.z = sign * tr.z;
            }
        }
        const double r0 = m->r0;  // ((n1 - n2) / (n1 + n2))^2: the two sides give quotients of opposite sign, one square
        double n = n1 / n2;
        w_in = -normalize(w_in);
        double cosI = dot(normal, w_in);
        double cosR = 1 - n * n * (1 - cosI * cosI);
        lobe_weight = true;
        if (cosR >= 0) {
            // pow(x, 5.0) by three multiplications: within 3 ulp of the library's result (which is itself not
            // correctly rounded on either side), and a fifth of a general pow's instructions on the lanes that
            // make their whole wave wait
            const double omc = 1 - cosI, omc2 = omc * omc;
            double reflCoeff = r0 + (1 - r0) * (omc2 * omc2 * omc);
            double unused, sampleFresnel;
            draw2(P, pixel, sample, (uint32_t)bounce, CPT_STREAM_SELECT, unused, sampleFresnel);
            if (sampleFresnel > reflCoeff) {  // refraction
                double cosT = sqrt(cosR);
                dvec3 refrac = (n * cosI - cosT) * normal - n * w_in;
                s.o = at - normal * 1e-12;
                s.d = refrac;
                s.medium = n2;
                s.atten = s.atten * tr;
                s.atten = s.atten * m->inv_p_trans;
            } else {  // Fresnel reflection about the geometric-side normal, medium kept
                rw = s.d;
                rn = N;
                if (s.medium != 1.0) rn = rn * -1.0;
                f1 = dv(m->ks);
                reflects = true;
            }
        } else {  // total internal reflection, medium kept
            rw = s.d;
            rn = normal;
            f1 = dv(m->kr) * tr;
            reflects = true;
        }
    }
    if (reflects) {
        dvec3 w_normal = dot(rw, rn) * rn;
        dvec3 w_tan = rw - w_normal;
        dvec3 w_ref = normalize(-w_normal + w_tan);
        s.o = at + rn * 1e-12;
        s.d = w_ref;
        if (reset_medium) s.medium = 1.0;
        s.atten = s.atten * f1;
        if (lobe_weight) s.atten = s.atten * m->inv_p_trans;
    }
}

// Both halves in sequence.  Returns true when the path goes on.
__device__ __forceinline__ bool shade_bounce(const DScene& sc, const DParams& P, uint32_t pixel, uint32_t sample,
                                             int bounce, double t, int tri, PathRegs& s) {
    dvec3 emitted;
    int lobe = choose_lobe(sc, P, pixel, sample, bounce, tri, s.atten, emitted);
    s.color = s.color + emitted;
    if (lobe == CPT_LOBE_NONE) return false;
    scatter(lobe, sc, P, pixel, sample, bounce, t, tri, s);
    return true;
}

// Pixel slot <-> pixel.  Slots enumerate 8x8 pixel tiles, so the 64 lanes of a wave that
// processes 64 consecutive slots hold one tile: neighbouring first rays, one coalesced block of
// state.  Slots whose pixel falls outside the image are never generated.
__device__ __forceinline__ bool slot_to_pixel(const DParams& P, int slot, int& i, int& j) {
    int tile = P.tile0 + (slot >> 6), in = slot & 63;
    int ty = (int)fast_div((uint32_t)tile, P.by_tiles_x), tx = tile - ty * P.tiles_x;
    i = tx * 8 + (in & 7);
    j = ty * 8 + (in >> 3);
    return i < P.w && j < P.h;
}
