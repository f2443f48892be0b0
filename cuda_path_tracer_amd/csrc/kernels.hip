// gfx950 kernels of the path tracer and their launchers.
//
// Wavefront pipeline (the product): a batch holds S samples of every pixel as S * n_pix_slots path
// slots.  generate writes every slot's first ray; then, once per bounce, extend intersects the rays
// of the current stream and shade consumes the hits and writes the surviving paths, compacted, to
// the other stream (workgroup-wide ballot + popcount prefix, one atomic per workgroup).  resolve
// folds the batch's per-path radiance into the per-pixel accumulation buffer in sample order, so
// the sum a pixel receives is the same sequence of fp64 additions whatever the batch size.
//
// Ray state is structure-of-arrays in HBM and is indexed by position in the stream, not by path:
// extend and shade read and write contiguous runs.  Slots enumerate 8x8 pixel tiles, so a wave's 64
// first rays are neighbours.
#include <hip/hip_runtime.h>

#include "device_core.h"
#include "launch.h"

#define CPT_BLOCK 256
#define CPT_WAVES_PER_BLOCK (CPT_BLOCK / CPT_WAVE)

// LDS traversal stacks for one workgroup: [wave][entry][lane]
#define CPT_DECLARE_STACK                                                                  \
    __shared__ uint32_t s_stack[CPT_WAVES_PER_BLOCK][CPT_STACK_ENTRIES][CPT_WAVE];          \
    uint32_t* my_stack = &s_stack[threadIdx.x / CPT_WAVE][0][threadIdx.x % CPT_WAVE];

// Streams are written once and read once, each time by another kernel: non-temporal accesses keep them from
// displacing the scene records in L2.
#define CPT_NT_LOAD(p) __builtin_nontemporal_load(&(p))
#define CPT_NT_STORE(p, v) __builtin_nontemporal_store((v), &(p))
__device__ __forceinline__ void load_ray_geometry(const RayStream& R, uint32_t p, dvec3& o, dvec3& d) {
    const cpt_v2d a = CPT_NT_LOAD(R.oxy[p]), b = CPT_NT_LOAD(R.ozdx[p]), c = CPT_NT_LOAD(R.dyz[p]);
    o = dv(a.x, a.y, b.x);
    d = dv(b.y, c.x, c.y);
}
__device__ __forceinline__ void load_ray(const RayStream& R, uint32_t p, PathRegs& s) {
    load_ray_geometry(R, p, s.o, s.d);
    const cpt_v2d m = CPT_NT_LOAD(R.mdax[p]), a = CPT_NT_LOAD(R.ayz[p]);
    s.medium = m.x;
    s.atten = dv(m.y, a.x, a.y);
}
__device__ __forceinline__ void store_ray(const RayStream& R, uint32_t p, const dvec3& o, const dvec3& d, double medium,
                                          const dvec3& atten, uint32_t slot) {
    cpt_v2d v;
    v.x = o.x; v.y = o.y; CPT_NT_STORE(R.oxy[p], v);
    v.x = o.z; v.y = d.x; CPT_NT_STORE(R.ozdx[p], v);
    v.x = d.y; v.y = d.z; CPT_NT_STORE(R.dyz[p], v);
    v.x = medium; v.y = atten.x; CPT_NT_STORE(R.mdax[p], v);
    v.x = atten.y; v.y = atten.z; CPT_NT_STORE(R.ayz[p], v);
    CPT_NT_STORE(R.slot[p], slot);
}
__device__ __forceinline__ void store_ray(const RayStream& R, uint32_t p, const PathRegs& s, uint32_t slot) {
    store_ray(R, p, s.o, s.d, s.medium, s.atten, slot);
}

__device__ __forceinline__ void store_hit(const HitBuf& H, uint32_t p, double t, int tri) {
    cpt_v2d v;
    v.x = t;
    v.y = __hiloint2double(0, tri);
    // an ordinary store, unlike the ray streams' non-temporal ones: the lanes of a wave retire at different times, and
    // the four 16-byte records of a line are put together in L2 instead of reaching HBM as partial lines
    ((cpt_v2d*)H.rec)[p] = v;
}
__device__ __forceinline__ void load_hit(const HitBuf& H, uint32_t p, double& t, int& tri) {
    const cpt_v2d v = CPT_NT_LOAD(((const cpt_v2d*)H.rec)[p]);
    t = v.x;
    tri = __double2loint(v.y);
}

// Compaction: positions in the output stream for the threads of a workgroup that keep their path,
// in thread order, behind one atomic.  A single queue counter takes on the order of 10^8 atomics
// per second however many CUs issue them; at one atomic per wave that limit, not the shading, set
// the pace of the kernel, so the count is aggregated over the whole workgroup first.
// Stands in for thrust::partition over whole path records (reference: src/improved_path_tracer.cu:321-324).
// Must be reached by every thread of the workgroup; returns the position (valid when keep).
#define CPT_SHADE_BLOCK 256
#define CPT_SHADE_WAVES (CPT_SHADE_BLOCK / CPT_WAVE)
__device__ __forceinline__ uint32_t block_claim(uint32_t* counter, bool keep) {
    __shared__ uint32_t s_off[CPT_SHADE_WAVES];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & (CPT_WAVE - 1), wave = threadIdx.x / CPT_WAVE;
    const uint64_t mask = __ballot(keep);
    if (lane == 0) s_off[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < CPT_SHADE_WAVES; ++w) {
            uint32_t c = s_off[w];
            s_off[w] = total;
            total += c;
        }
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t pos = s_base + s_off[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();
    return pos;
}

// ---- generate ----------------------------------------------------------------------------
// One thread per slot of the batch: slot = k * n_pix_slots + pixel slot, sample = batch_s0 + k, and
// the ray goes to stream position `slot`.  Every slot gets a ray: slots that have no pixel (image
// sides that are not multiples of 8) or whose path ends before its first segment get a ray that
// starts far outside the scene and leaves after one node, and are never resolved.
// reference: initRays (src/improved_path_tracer.cu:221-251)
__global__ void __launch_bounds__(CPT_BLOCK) k_generate(DScene sc, DParams P, RayStream out, Radiance col, int batch_s0,
                                                        int batch_n) {
    int total = batch_n * P.n_pix_slots;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += gridDim.x * blockDim.x) {
        int k = (int)fast_div((uint32_t)slot, P.by_pix_slots), ps = slot - k * P.n_pix_slots;
        int i, j;
        bool live = slot_to_pixel(P, ps, i, j);
        dvec3 o = dv(1e300, 1e300, 1e300), d = dv(1, 0, 0), atten = dv(0, 0, 0);
        if (live) {
            int sample = batch_s0 + k;
            first_ray(sc, P, i, j, sample, o, d);
            atten = dv(1, 1, 1);
            if (P.rr) {  // roulette before the first segment, as before every other one
                PathRegs s;
                s.atten = atten;
                live = roulette(P, (uint32_t)(i + j * P.w), (uint32_t)sample, 0, s);
                atten = s.atten;
            }
        }
        if (!live) {
            o = dv(1e300, 1e300, 1e300);
            d = dv(1, 0, 0);
            atten = dv(0, 0, 0);
        }
        const uint32_t p = (uint32_t)slot;
        if (P.rr) {
            store_ray(out, p, o, d, 1.0, atten, p);
        } else {
            // without roulette a first ray's medium, throughput and slot are constants (1, (1,1,1), its own position):
            // only the geometry is stored, and the first shade launch (FIRST) does not read the rest
            cpt_v2d v;
            v.x = o.x; v.y = o.y; CPT_NT_STORE(out.oxy[p], v);
            v.x = o.z; v.y = d.x; CPT_NT_STORE(out.ozdx[p], v);
            v.x = d.y; v.y = d.z; CPT_NT_STORE(out.dyz[p], v);
        }
        // the path has gathered nothing yet: one byte says so (the 24 bytes of radiance are written by the first emitter the
        // path meets and read by k_resolve only if there was one -- most paths of most scenes never meet one)
        col.lit[slot] = 0;
    }
}

// ---- segmented streams -------------------------------------------------------------------------
// A stream is CPT_PW_SEGMENTS independent runs: segment s holds its rays at positions [s * cap, s * cap + count[s]),
// count[s] kept in word s * 32 of the stream's counter block (a 128-byte line each).  Shade compacts the survivors
// of segment s into segment s of the other stream, so the workgroups of a launch spread their output claims over
// CPT_PW_SEGMENTS counters instead of one: a single address takes on the order of 10^8 atomics per second, which
// 156 000 workgroup claims per launch of 40 M rays ran into.
// A dense run of n rays at positions 0 .. n-1 (first rays, caller-supplied rays) is the same thing with every
// segment but the last ones full.
__global__ void k_dense_counts(uint32_t* __restrict__ counts, uint32_t n, uint32_t cap) {
    const uint32_t s = threadIdx.x;
    if (s < CPT_PW_SEGMENTS) {
        const unsigned long long lo = (unsigned long long)s * cap;
        counts[s * 32u] = lo >= n ? 0u : (n - lo < cap ? (uint32_t)(n - lo) : cap);
    }
}

// ---- extend --------------------------------------------------------------------------------
// Closest hit for every ray of the stream, one ray per thread.
// reference: the intersection half of pathTrace (src/improved_path_tracer.cu:56-76)
template <bool COUNT>
__global__ void __launch_bounds__(CPT_BLOCK) k_extend(DScene sc, RayStream in, HitBuf hit,
                                                      const uint32_t* __restrict__ counter, uint32_t cap, unsigned long long* work,
                                                      ExtendObserve obs) {
    CPT_DECLARE_STACK
    __shared__ uint32_t s_count[CPT_PW_SEGMENTS];
    if (threadIdx.x < CPT_PW_SEGMENTS) s_count[threadIdx.x] = counter[threadIdx.x * 32u];
    __syncthreads();
    unsigned long long steps = 0, tris = 0;
    const unsigned long long span = (unsigned long long)cap * CPT_PW_SEGMENTS;
    for (unsigned long long x = blockIdx.x * blockDim.x + threadIdx.x; x < span; x += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t seg = (uint32_t)(x / cap);
        if ((uint32_t)(x - (unsigned long long)seg * cap) >= s_count[seg]) continue;
        const uint32_t q = (uint32_t)x;
        dvec3 o, d;
        load_ray_geometry(in, q, o, d);
        double t;
        int tri;
        if (COUNT) {
            TraceRec rec = {0, 0, 0, 0, 0, obs.bitmap ? obs.bitmap + (size_t)q * (size_t)obs.bitmap_stride : nullptr};
            traverse<true>(sc, o, d, my_stack, t, tri, &rec);
            steps += (unsigned long long)rec.node_steps;
            tris += (unsigned long long)rec.tri_tests;
            if (obs.vhash) obs.vhash[q] = rec.visit_hash;
        } else {
            traverse<false>(sc, o, d, my_stack, t, tri, nullptr);
        }
        store_hit(hit, q, t, tri);
    }
    if (COUNT) {
        if (steps) atomicAdd(&work[3], steps);
        if (tris) atomicAdd(&work[4], tris);
    }
}

// ---- extend, persistent waves ------------------------------------------------------------------
// One 1024-thread workgroup per CU; each of its 16 waves keeps 64 rays in flight and refills
// finished lanes from the queue, so a wave is not held up by its longest traversal.  Work per ray
// is a fixed set of node expansions and triangle tests (the reference never orders or culls), so
// their order is free: leaf hits are parked on a second per-lane LDS stack and tested when enough
// lanes have one, which keeps both the box code and the triangle code running with most lanes
// active instead of alternating half-empty.
//
// Incoherent rays make every lane fetch a different 128-byte record, and then the vector memory
// path, not arithmetic, bounds the kernel.  The workgroup therefore keeps the head of the node
// array (records are stored top of the tree first) and the head of the triangle array in LDS, 64 KB
// shared by its 16 waves: small scenes live there entirely, large ones keep their upper levels
// there.  The exact variants keep the cache as seven planes (piece k of every cached record side by side, see
// fetch_record): consecutive records are 16 bytes apart in a plane, so lanes reading the same piece of different
// records spread over the LDS banks, and a fetch computes one address.  The binary32 variants keep whole records,
// their pieces rotated by an XOR (with (r >> 1) & 7 for a 128-byte record) to the same end.
//
// Queue entries are claimed in chunks: one atomic per `chunk` rays, the rest is wave-local.
// The per-lane node stack keeps its first entries in LDS (PwPlan::STACK_LDS); deeper entries (rare)
// go to a per-lane column of `spill` in global memory.
#define CPT_PW_THREADS CPT_PW_THREADS_HOST
#define CPT_PW_WAVES (CPT_PW_THREADS / CPT_WAVE)
#define CPT_PW_CACHE_RECORDS CPT_PW_CACHE_RECORDS_HOST  // 66 KB of 128-byte records
#ifndef CPT_PW_LEAF_CAP
#define CPT_PW_LEAF_CAP 10
#endif
#ifndef CPT_PW_HOT_TRIS
#define CPT_PW_HOT_TRIS 32    // record-cache slots kept for triangles when the node records alone would fill it
#endif
// How an instance of the persistent kernel divides the CU's 160 KB of LDS: 66 KB of scene records, 12 node-stack entries
// and 10 parked leaves per lane (88 KB); the 6 KB left over are what the shade workgroups running beside it need.  The exact variants keep the records as seven planes (planar_record), which
// holds CPT_PW_PLANAR_SLOTS_HOST of them because the never-read eighth piece is not stored; the binary32 variants keep
// CPT_PW_CACHE_RECORDS whole 128-byte slots (two 64-byte node records per slot).
template <bool F32>
struct PwPlan {
    static constexpr int STACK_LDS = CPT_PW_STACK_LDS_HOST;  // node-stack entries per lane in LDS
    static constexpr int LEAF_CAP = CPT_PW_LEAF_CAP;         // parked leaves per lane (a node expansion adds at most two)
    static constexpr int PLANE_SLOTS = CPT_PW_PLANAR_SLOTS_HOST;
    static constexpr int PLANE_BYTES = PLANE_SLOTS * 16;
    static constexpr int CACHED_SLOTS = F32 ? CPT_PW_CACHE_RECORDS : PLANE_SLOTS;  // record slots of the cache
    static constexpr int CACHE_BYTES = CPT_PW_CACHE_RECORDS * 128;
    static_assert(7 * PLANE_BYTES <= CACHE_BYTES, "the seven planes must fit the cache");
    static_assert(PLANE_SLOTS & 1, "an odd number of slots per plane puts the pieces of one record into different bank groups");
};

struct Rec7 {  // the first seven 16-byte pieces of a 128-byte record
    double2 p0, p1, p2, p3, p4, p5, p6;
};
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_v2d;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) int32_t lds_i32;
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ double2 lds_piece(const lds_byte* p) {
    v2d v = *(const lds_v2d*)p;
    return make_double2(v.x, v.y);
}
// Planar cache: piece k of slot s at k * PLANE_BYTES + s * 16 (seven planes of 16-byte pieces; the eighth piece of a
// record is never read): one address per fetch, the planes are reached through the instruction's offset field, and
// lanes reading different slots spread over the banks without any swizzle.
template <int PLANE_BYTES>
__device__ __forceinline__ Rec7 planar_record(const lds_byte* cache, uint32_t slot) {
    Rec7 r;
    const lds_byte* rec = cache + slot * 16u;
    r.p0 = lds_piece(rec); r.p1 = lds_piece(rec + PLANE_BYTES);
    r.p2 = lds_piece(rec + 2 * PLANE_BYTES); r.p3 = lds_piece(rec + 3 * PLANE_BYTES);
    r.p4 = lds_piece(rec + 4 * PLANE_BYTES); r.p5 = lds_piece(rec + 5 * PLANE_BYTES);
    r.p6 = lds_piece(rec + 6 * PLANE_BYTES);
    return r;
}
__device__ __forceinline__ Rec7 global_record(const void* __restrict__ base, int idx) {
    Rec7 r;
    const double2* g = (const double2*)((const unsigned char*)base + (size_t)idx * 128);
    r.p0 = g[0]; r.p1 = g[1]; r.p2 = g[2]; r.p3 = g[3]; r.p4 = g[4]; r.p5 = g[5]; r.p6 = g[6];
    return r;
}
// PLANE_BYTES > 0: planar cache; 0: whole 128-byte records, their pieces rotated by an XOR with (slot >> 1) & 7.
template <bool RESIDENT, int PLANE_BYTES>
__device__ __forceinline__ Rec7 fetch_record(const void* __restrict__ base, const lds_byte* cache, int cache_first,
                                             int cached, int idx) {
    if (PLANE_BYTES > 0 && (RESIDENT || idx < cached)) {
        return planar_record<PLANE_BYTES ? PLANE_BYTES : 16>(cache, (uint32_t)(cache_first + idx));
    } else if (RESIDENT || idx < cached) {
        Rec7 r;
        const uint32_t slot = (uint32_t)(cache_first + idx);
        const lds_byte* rec = cache + slot * 128u;
        const uint32_t z = (slot >> 1) & 7u;
        r.p0 = lds_piece(rec + ((0u ^ z) << 4)); r.p1 = lds_piece(rec + ((1u ^ z) << 4));
        r.p2 = lds_piece(rec + ((2u ^ z) << 4)); r.p3 = lds_piece(rec + ((3u ^ z) << 4));
        r.p4 = lds_piece(rec + ((4u ^ z) << 4)); r.p5 = lds_piece(rec + ((5u ^ z) << 4));
        r.p6 = lds_piece(rec + ((6u ^ z) << 4));
        return r;
    }
    return global_record(base, idx);
}

struct Rec4f {  // a 64-byte binary32 node record as four 16-byte pieces
    float4 p0, p1, p2, p3;
};
typedef float v4f_lds __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f_lds lds_v4f;
__device__ __forceinline__ float4 lds_piece4(const lds_byte* p) {
    v4f_lds v = *(const lds_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// Cached binary32 node records start at byte 0 of the cache, 64 bytes each, their four pieces rotated by an XOR with
// (index >> 2) & 3 so that lanes reading the same piece of different records spread over the LDS banks.
template <bool RESIDENT>
__device__ __forceinline__ Rec4f fetch_node32(const void* __restrict__ base, const lds_byte* cache, int cached, int idx) {
    Rec4f r;
    if (RESIDENT || idx < cached) {
        const lds_byte* rec = cache + (uint32_t)idx * 64u;
        const uint32_t z = ((uint32_t)idx >> 2) & 3u;
        r.p0 = lds_piece4(rec + ((0u ^ z) << 4)); r.p1 = lds_piece4(rec + ((1u ^ z) << 4));
        r.p2 = lds_piece4(rec + ((2u ^ z) << 4)); r.p3 = lds_piece4(rec + ((3u ^ z) << 4));
    } else {
        const float4* g = (const float4*)((const unsigned char*)base + (size_t)idx * 64);
        r.p0 = g[0]; r.p1 = g[1]; r.p2 = g[2]; r.p3 = g[3];
    }
    return r;
}

// ---- binary32 node records ------------------------------------------------------------------------
// What the reference's walk returns depends on the tree only through its leaves: a child box always contains
// its descendants' boxes, correctly rounded subtraction and division are monotonic, so whenever a leaf's own box
// passes the reference's slab test every box above it passes too.  The closest hit is therefore "the smallest t
// (ties: smallest test_rank) over the triangles whose OWN leaf box passes the exact test and whose triangle test
// hits", whatever inner boxes stand above them.  The F32 variant of the persistent kernel uses that: inner nodes
// are tested in binary32 against outward-rounded copies of the boxes with error margins that make the test
// conservative (it never rejects a box the exact test accepts), so it walks a superset of the reference's nodes
// with a third of the arithmetic and half the record bytes, and every candidate leaf is then put to the exact
// binary64 test of its own box (the bounds of the triangle's three vertices, exactly what the tree stores for a
// leaf) before its triangle is tested.  Hits -- triangle, t bits -- are those of the exact walk; the visit set is
// not the reference's, so the parity hooks and the default variant stay on the exact walk.
struct __attribute__((aligned(64))) Node32 {
    // pairs (left child, right child): lo.x, lo.y | lo.z, hi.x | hi.y, hi.z; each pair is one packed-fp32 operand
    float lox[2], loy[2], loz[2], hix[2], hiy[2], hiz[2];
    int32_t lref, rref;
    int32_t lid, rid;  // canonical ids of the children (visit-set reporting only)
};
static_assert(sizeof(Node32) == 64, "Node32 is half a node record");
__device__ __forceinline__ float f32_down(double x) {  // largest float <= x
    float f = (float)x;
    return ((double)f > x) ? nextafterf(f, -INFINITY) : f;
}
__device__ __forceinline__ float f32_up(double x) {  // smallest float >= x
    float f = (float)x;
    return ((double)f < x) ? nextafterf(f, INFINITY) : f;
}
// A plane beyond the binary32 range is stored as the largest finite float, not as an infinity (0 * inf would put
// a NaN into the test of an axis the ray does not move on, and minimum/maximum drop NaNs silently).  That is no
// longer outward for such a plane, and need not be: the scene's extent, taken from the binary64 boxes, is then
// infinite in binary32, which leaves every axis of every ray unconstrained (see ray32_axis).
__device__ __forceinline__ float plane_down(double x) { return fmaxf(f32_down(x), -3.4028234663852886e38f); }
__device__ __forceinline__ float plane_up(double x) { return fminf(f32_up(x), 3.4028234663852886e38f); }
__global__ void __launch_bounds__(CPT_BLOCK) k_nodes32(const cpt_node* __restrict__ nodes, int n, Node32* __restrict__ out) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const cpt_node nd = nodes[k];
        Node32 r;
        r.lox[0] = plane_down(nd.lbox[0]); r.lox[1] = plane_down(nd.rbox[0]);
        r.loy[0] = plane_down(nd.lbox[1]); r.loy[1] = plane_down(nd.rbox[1]);
        r.loz[0] = plane_down(nd.lbox[2]); r.loz[1] = plane_down(nd.rbox[2]);
        r.hix[0] = plane_up(nd.lbox[3]); r.hix[1] = plane_up(nd.rbox[3]);
        r.hiy[0] = plane_up(nd.lbox[4]); r.hiy[1] = plane_up(nd.rbox[4]);
        r.hiz[0] = plane_up(nd.lbox[5]); r.hiz[1] = plane_up(nd.rbox[5]);
        r.lref = nd.lref; r.rref = nd.rref;
        r.lid = nd.lid; r.rid = nd.rid;
        out[k] = r;
    }
}

// A ray as the binary32 box test wants it: per axis the reciprocal direction and the two additive constants of
//   t(plane) = plane * r + c,   c = -(o * r) -/+ E,
// one for the plane that is the near one for this ray's direction (lowered by E) and one for the far one (raised
// by E).  E = 2^-20 (S |r| + |o r|), S = the largest plane magnitude in the scene, is sixteen times what the
// roundings of r, of o * r and of c can amount to; the outward rounding of the planes is on the safe side by
// itself, and the last rounding (of the fma) is covered by the relative slack of the final comparison.  An axis
// whose products could overflow binary32, or on which the ray does not move, is left unconstrained (r = 0,
// constants -inf / +inf): the exact test of the leaves decides.
typedef float v2f __attribute__((ext_vector_type(2)));
struct Ray32 {
    float rx, ry, rz;
    float clx, chx, cly, chy, clz, chz;  // constants for the lo plane and the hi plane of each axis
};
__device__ __forceinline__ void ray32_axis(double o, double r, float S, float& r32, float& c_lo, float& c_hi) {
    // o * r once in binary64 (o and r can be large with a small product's worth of cancellation to come), the rest in
    // binary32: its own roundings are relative 2^-24 of terms that E carries at 2^-20
    const float orr = (float)(o * r), rf = (float)r;
    const float ar = fabsf(rf), ao = fabsf(orr);
    const bool ok = ar < 1e30f && ao < 1e30f && S * ar < 1e30f;  // also false for NaN
    const float E = fmaf(fmaf(S, ar, ao), 9.5367431640625e-07f, 1e-30f);  // 2^-20 (S |r| + |o r|)
    const float c_near = -orr - E, c_far = -orr + E;
    r32 = ok ? rf : 0.0f;
    c_lo = ok ? (rf >= 0 ? c_near : c_far) : -INFINITY;
    c_hi = ok ? (rf >= 0 ? c_far : c_near) : INFINITY;
}
__device__ __forceinline__ Ray32 ray32_prepare(const dvec3& o, const SlabRay& sr, float S) {
    Ray32 q;
    ray32_axis(o.x, sr.r.x, S, q.rx, q.clx, q.chx);
    ray32_axis(o.y, sr.r.y, S, q.ry, q.cly, q.chy);
    ray32_axis(o.z, sr.r.z, S, q.rz, q.clz, q.chz);
    return q;
}
// Both child boxes of a node at once (component 0: left child, 1: right child).  `keep` is false only when the
// exact test is certain to miss; `enter` is a lower bound of the distance at which the ray enters the box.
__device__ __forceinline__ void slab32_pair(v2f lox, v2f loy, v2f loz, v2f hix, v2f hiy, v2f hiz, const Ray32& q,
                                            bool& keep_l, bool& keep_r, float& enter_l, float& enter_r) {
    const v2f x1 = __builtin_elementwise_fma(lox, (v2f)(q.rx), (v2f)(q.clx)), x2 = __builtin_elementwise_fma(hix, (v2f)(q.rx), (v2f)(q.chx));
    const v2f y1 = __builtin_elementwise_fma(loy, (v2f)(q.ry), (v2f)(q.cly)), y2 = __builtin_elementwise_fma(hiy, (v2f)(q.ry), (v2f)(q.chy));
    const v2f z1 = __builtin_elementwise_fma(loz, (v2f)(q.rz), (v2f)(q.clz)), z2 = __builtin_elementwise_fma(hiz, (v2f)(q.rz), (v2f)(q.chz));
#define CPT_SLAB32(K, KEEP, ENTER)                                                                    \
    {                                                                                                 \
        const float tn = fmaxf(fmaxf(fminf(x1[K], x2[K]), fminf(y1[K], y2[K])), fminf(z1[K], z2[K])); \
        const float tf = fminf(fminf(fmaxf(x1[K], x2[K]), fmaxf(y1[K], y2[K])), fmaxf(z1[K], z2[K])); \
        const float slack = fmaf(fabsf(tn) + fabsf(tf), 9.5367431640625e-07f, 1e-30f);                \
        KEEP = !((tn - tf > slack) || (tf < 0.0f));                                                   \
        ENTER = tn;                                                                                   \
    }
    CPT_SLAB32(0, keep_l, enter_l)
    CPT_SLAB32(1, keep_r, enter_r)
#undef CPT_SLAB32
}

// ---- records for rays that share an origin ----------------------------------------------------------
// Every first ray of a pin-hole render starts at the camera's eye.  The slab test's first operation on a box plane is
// plane - origin (reference: src/gpu/GPUbb.cuh:33-34) and the triangle test's plane term is -dot(origin - a, n)
// (src/gpu/GPUTrimesh.cuh:62): with the origin fixed both are properties of the record, so they are worked out here once
// per camera, with the very operations the per-ray code uses, and the EYE instance of the persistent kernel reads them
// instead of redoing them at every node step and triangle test -- 12 of a node step's subtractions and 8 of a triangle
// test's operations, with bit-identical results.
// Node records keep their layout.  Triangle records keep vertices and normal; bytes 96..111 become
// (test_rank, flags, plane term): the material index is not needed by the walk.
__global__ void __launch_bounds__(CPT_BLOCK) k_eye_records(const cpt_node* __restrict__ nodes, int n_nodes, const cpt_tri* __restrict__ tris,
                                                           int n_tris, dvec3 eye, cpt_node* __restrict__ nodes_out, cpt_tri* __restrict__ tris_out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes + n_tris; i += gridDim.x * blockDim.x) {
        if (i < n_nodes) {
            cpt_node r = nodes[i];
            r.lbox[0] = r.lbox[0] - eye.x; r.lbox[1] = r.lbox[1] - eye.y; r.lbox[2] = r.lbox[2] - eye.z;
            r.lbox[3] = r.lbox[3] - eye.x; r.lbox[4] = r.lbox[4] - eye.y; r.lbox[5] = r.lbox[5] - eye.z;
            r.rbox[0] = r.rbox[0] - eye.x; r.rbox[1] = r.rbox[1] - eye.y; r.rbox[2] = r.rbox[2] - eye.z;
            r.rbox[3] = r.rbox[3] - eye.x; r.rbox[4] = r.rbox[4] - eye.y; r.rbox[5] = r.rbox[5] - eye.z;
            nodes_out[i] = r;
        } else {
            const int k = i - n_nodes;
            cpt_tri r = tris[k];
            const double num = tri_numerator(eye, dv(r.a[0], r.a[1], r.a[2]), dv(r.n[0], r.n[1], r.n[2]));
            r.material = (int32_t)r.test_rank;
            double2* q = (double2*)&r;
            q[6].y = num;
            tris_out[k] = r;
        }
    }
}

//
// CULL selects the fast mode: of two child boxes that are both hit the nearer one is walked first, and
// a box the ray enters beyond the closest hit found so far (with a relative margin of 1e-6) is not
// walked at all.  The closest hit is the same one; the set of visited nodes, which the reference
// mode reproduces exactly, is not.
// MODE (checked by the host, see scene_mode): 2 = RESIDENT, the whole scene is in the record cache and the walk's node
// stack never needs more than its LDS entries; 1 = the stack fits its LDS entries but records may have to come from
// global memory; 0 = neither.  Instances with an LDS-only stack address it by pointer and have no spill path.
// RESIDENT: the whole scene is in the record cache and its tree is low enough for the LDS part of the node
// stack, so the loop has no vector-memory record or spill path and none of the waits that come with
// them; the only loads left are the rays.
// F32: inner nodes through the conservative binary32 test, leaves through the exact test of their own box (see above).
// EYE: every live ray of the stream starts at `eye` (the first rays of a pin-hole render) and the records come from
// k_eye_records: the origin lives in scalar registers, is not loaded per ray, and the box and triangle tests skip the
// operations the records already contain.  A slot without a path (origin 1e300, see k_generate) is retired at once with
// the counts its walk would have had: one node step through the exact test, no triangle.
template <bool COUNT, bool CULL, int MODE, bool F32, int REPS, bool EYE>
__global__ void __launch_bounds__(CPT_PW_THREADS) k_extend_pw(DScene sc, RayStream in, HitBuf hit,
                                                              const uint32_t* __restrict__ counter, uint32_t* cursor, uint32_t cap,
                                                              uint32_t chunk, uint32_t refill_at, uint32_t leaf_go, uint32_t full_go, uint32_t* spill, unsigned long long* work,
                                                              ExtendObserve obs, dvec3 eye) {
    static_assert(!(F32 && EYE), "the shared-origin records exist for the exact variants only");
    constexpr bool RESIDENT = MODE == 2, LDS_STACK = MODE >= 1;
    // SEL: the resident instance of the exact variants keeps its records in the layout described at
    // CPT_PW_RESIDENT_NODE_BYTES: every lane reads, per child and axis, the plane its ray meets first and the one it
    // meets last (slab_selected), and the instance without culling gives the node-stack entries such a low tree cannot need
    // to the records.
    constexpr bool SEL = RESIDENT && !F32;
    const void* const node_recs = EYE ? (const void*)sc.nodes_eye : (const void*)sc.nodes_pw;
    const void* const tri_recs = EYE ? (const void*)sc.tris_eye : (const void*)sc.tris;
    typedef PwPlan<F32> Plan;
    constexpr int kStackLds = (SEL && !CULL) ? CPT_PW_RESIDENT_STACK : Plan::STACK_LDS, kLeafCap = Plan::LEAF_CAP;
    constexpr int kPlaneBytes = F32 ? 0 : Plan::PLANE_BYTES;  // 0: whole records with rotated pieces (fetch_record)
    constexpr int kCacheBytes = Plan::CACHE_BYTES + (Plan::STACK_LDS - kStackLds) * 4 * CPT_PW_THREADS;
    __shared__ __attribute__((aligned(128))) unsigned char s_cache[kCacheBytes];
    __shared__ uint32_t s_nodes[CPT_PW_WAVES][kStackLds][CPT_WAVE];
    __shared__ uint32_t s_leaves[CPT_PW_WAVES][kLeafCap][CPT_WAVE];
    const uint32_t lane = threadIdx.x & (CPT_WAVE - 1), wave = threadIdx.x / CPT_WAVE;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    // lengths of the stream's segments, one per lane of every wave (lanes beyond the segments hold 0)
    const uint32_t my_count = lane < CPT_PW_SEGMENTS ? counter[lane * 32u] : 0u;
    uint32_t n = my_count;
    for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off);
    // A wave's last chunk is the tail of the launch (its lanes finish one by one while the other waves have
    // nothing left), so a short stream is handed out in smaller chunks: at least eight claims per wave.
    {
        const uint32_t fair = n / (gridDim.x * CPT_PW_WAVES * 8u);
        chunk = fair < chunk ? (fair < 32u ? 32u : fair) : chunk;
    }

    // fill the record cache: node records first, triangle records in what is left
    // (F32: 64-byte binary32 node records, twice as many per byte; triangle records start at the next 128-byte slot)
    // A scene that does not fit whole keeps CPT_PW_HOT_TRIS slots for its first triangles, which the library has put
    // largest bounding box first: those few take most of the triangle tests.
    constexpr int kSlots = Plan::CACHED_SLOTS;  // record slots of this instance's cache layout
    const bool fits = (F32 ? (sc.n_nodes + 1) / 2 : sc.n_nodes) + sc.n_tris <= kSlots;
    const int tri_reserve = fits ? 0 : (sc.n_tris < CPT_PW_HOT_TRIS ? sc.n_tris : CPT_PW_HOT_TRIS);
    const int node_room = F32 ? (kSlots - tri_reserve) * 2 : kSlots - tri_reserve;
    const int cached_nodes = sc.n_nodes < node_room ? sc.n_nodes : node_room;
    const int tri_first = F32 ? (cached_nodes + 1) / 2 : cached_nodes;  // first 128-byte slot of the cached triangles
    const int room = kSlots - tri_first;
    const int cached_tris = sc.n_tris < room ? sc.n_tris : room;
    if (F32) {
        for (int i = (int)threadIdx.x; i < cached_nodes * 4; i += CPT_PW_THREADS) {
            int r = i >> 2, c = i & 3;
            *(float4*)(s_cache + (size_t)r * 64 + (((uint32_t)c ^ (((uint32_t)r >> 2) & 3u)) << 4)) =
                *(const float4*)((const unsigned char*)sc.nodes32 + (size_t)r * 64 + c * 16);
        }
        for (int i = (int)threadIdx.x; i < cached_tris * 8; i += CPT_PW_THREADS) {
            int r = tri_first + (i >> 3), c = i & 7;
            *(double2*)(s_cache + (size_t)r * 128 + (((uint32_t)c ^ (((uint32_t)r >> 1) & 7u)) << 4)) =
                *(const double2*)((const unsigned char*)sc.tris + (size_t)(i >> 3) * 128 + c * 16);
        }
    } else if (SEL) {  // the host has checked that all of it fits (scene_mode)
        for (int i = (int)threadIdx.x; i < sc.n_nodes * 21; i += CPT_PW_THREADS) {
            const int r = i / 21, e = i - r * 21;
            const double* src = (const double*)((const unsigned char*)node_recs + (size_t)r * 128);  // lower x y z, upper x y z of the left child, of the right child, references, ids
            double v = 0.0;
            if (e < 18) {
                const int child = e / 9, axis = (e - child * 9) / 3, j = e - child * 9 - axis * 3;  // (upper, lower, upper)
                v = src[child * 6 + (j == 1 ? axis : 3 + axis)];
            } else if (e < 20) {
                v = src[12 + (e - 18)];
            }
            *(double*)(s_cache + (size_t)r * CPT_PW_RESIDENT_NODE_BYTES + (size_t)e * 8) = v;
        }
        const int tri_lds = (sc.n_nodes * CPT_PW_RESIDENT_NODE_BYTES + 15) / 16 * 16;
        for (int i = (int)threadIdx.x; i < sc.n_tris * 7; i += CPT_PW_THREADS) {
            const int r = i / 7, c = i - r * 7;
            *(double2*)(s_cache + tri_lds + (size_t)r * CPT_PW_RESIDENT_TRI_BYTES + (size_t)c * 16) =
                *(const double2*)((const unsigned char*)tri_recs + (size_t)r * 128 + c * 16);
        }
    } else {
        for (int i = (int)threadIdx.x; i < (cached_nodes + cached_tris) * 8; i += CPT_PW_THREADS) {
            int r = i >> 3, c = i & 7;
            const unsigned char* src = r < cached_nodes ? (const unsigned char*)node_recs + (size_t)r * 128
                                                        : (const unsigned char*)tri_recs + (size_t)(r - cached_nodes) * 128;
            if (c < 7) *(double2*)(s_cache + (size_t)c * Plan::PLANE_BYTES + (size_t)r * 16) = *(const double2*)(src + c * 16);  // planar, see planar_record
        }
    }
    // the largest plane magnitude of the scene (the root's two child boxes contain everything): scale of the
    // binary32 test's error margins
    float scene_extent = 0.0f;
    if (F32 && sc.n_nodes > 0) {
        const double* b = (const double*)sc.nodes_pw;
        double m = 0.0;
        for (int k = 0; k < 12; ++k) m = fmax(m, fabs(b[k]));
        scene_extent = f32_up(m);
    }
    __syncthreads();
    const lds_byte* cache = (const lds_byte*)s_cache;

    uint32_t* my_nodes = &s_nodes[wave][0][lane];    // entry e at my_nodes[e * CPT_WAVE]
    // The parked-leaf list is addressed by a pointer to its next free entry, not by a count: a push is one LDS write
    // and one add.  Entries hold the child reference as the node record has it (~triangle index).
    lds_u32* const leaf_base = (lds_u32*)&s_leaves[wave][0][lane];  // entries are CPT_WAVE words apart
    lds_u32* leaf_top = leaf_base;
#define CPT_LEAVES_AT_MOST(N) (leaf_top <= leaf_base + (N) * CPT_WAVE)

    // likewise the node stack of the instances whose stack never outgrows LDS (the others count entries: theirs continues in global memory)
    lds_u32* const node_base = (lds_u32*)&s_nodes[wave][0][lane];
    lds_u32* node_top = node_base;
    uint32_t* my_spill = spill + (size_t)blockIdx.x * CPT_PW_THREADS + threadIdx.x;  // entry e at my_spill[e * spill_stride]
    const size_t spill_stride = (size_t)gridDim.x * CPT_PW_THREADS;
    uint32_t chunk_pos = 0, chunk_end = 0;  // wave-uniform: unclaimed part of this wave's chunk
    bool drained = false;                   // wave-uniform: the queue has been handed out completely
    uint32_t seg = (blockIdx.x * CPT_PW_WAVES + wave) % CPT_PW_SEGMENTS, segs_empty = 0;  // wave-uniform: segment being claimed from

    bool has_ray = false;
    int node = -1, sp = 0;
    uint32_t pos = 0;  // stream position of this lane's ray
    dvec3 o = EYE ? eye : dv(0, 0, 0), d = dv(0, 0, 1);
    SlabRay sr = slab_prepare(o, dv(1, 1, 1));
    // SEL: byte offsets into a node's record of the left child's (near, far) plane pair of each axis for this lane's ray:
    // the axis' triple starts at 24 * axis and holds (upper, lower, upper), so a ray that moves up the axis reads from
    // its second double on, one that moves down from its first
    uint32_t sel_x = 8, sel_y = 32, sel_z = 56;
    const uint32_t tri_lds = (uint32_t)((sc.n_nodes * CPT_PW_RESIDENT_NODE_BYTES + 15) / 16 * 16);
    Ray32 q32 = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double best_t = -1.0;
    int best_tri = -1;
    uint32_t best_rank = 0xFFFFFFFFu;
    unsigned long long steps = 0, tris = 0, exact = 0, box_passes = 0, leaf_passes = 0;
    unsigned long long vhash = 0;  // COUNT: order-free hash of the canonical ids of the nodes whose boxes this lane's ray has tested

    for (;;) {
        // ---- refill idle lanes ----
        uint64_t idle = __ballot(!has_ray);
        if (!drained && (uint32_t)__popcll(idle) >= refill_at) {
            while (chunk_pos == chunk_end && !drained) {  // claim the next chunk of this wave's current segment
                const uint32_t seg_size = (uint32_t)__builtin_amdgcn_readlane((int)my_count, (int)seg), size = chunk;
                const uint32_t seg_lo = seg * cap, seg_hi = seg_lo + seg_size;
                uint32_t off = 0;
                if (lane == 0) off = atomicAdd(cursor + seg * 32u, size);
                off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
                if (off < seg_size) {
                    chunk_pos = seg_lo + off;
                    chunk_end = (seg_size - off) < size ? seg_hi : chunk_pos + size;
                } else {  // used up: on to the next segment; once all have been seen empty the stream is handed out
                    seg = (seg + 1) % CPT_PW_SEGMENTS;
                    if (++segs_empty == CPT_PW_SEGMENTS) drained = true;
                }
            }
            uint32_t avail = chunk_end - chunk_pos;
            uint32_t want = (uint32_t)__popcll(idle);
            uint32_t take = want < avail ? want : avail;
            uint32_t mine = (uint32_t)__popcll(idle & lanes_below);
            if (!has_ray && mine < take) {
                pos = chunk_pos + mine;  // idle lanes take consecutive positions: one contiguous run per refill
                bool no_path = false;
                if (EYE) {
                    const cpt_v2d b = CPT_NT_LOAD(in.ozdx[pos]), cc = CPT_NT_LOAD(in.dyz[pos]);
                    d = dv(b.y, cc.x, cc.y);
                    no_path = b.x == 1e300;
                } else {
                    load_ray_geometry(in, pos, o, d);
                }
                sr = slab_prepare(o, d);
                if (SEL) {
                    sel_x = sr.r.x < 0 ? 0u : 8u;
                    sel_y = sr.r.y < 0 ? 24u : 32u;
                    sel_z = sr.r.z < 0 ? 48u : 56u;
                }
                if (F32) q32 = ray32_prepare(o, sr, scene_extent);
                best_t = -1.0;
                best_tri = -1;
                best_rank = 0xFFFFFFFFu;
                if (COUNT) vhash = 0;
                sp = 0;
                node_top = node_base;
                leaf_top = leaf_base;
                has_ray = true;
                if (EYE && no_path) {
                    node = -1;
                    if (COUNT) { steps++; exact++; }
                } else if (sc.n_nodes > 0) {
                    node = 0;
                } else {  // one-triangle scene
                    node = -1;
                    *leaf_base = ~0u;  // ~(triangle 0)
                    leaf_top = leaf_base + CPT_WAVE;
                }
            }
            chunk_pos += take;
        }
        if (__ballot(has_ray) == 0) {
            if (drained) break;
            continue;
        }
    

        // ---- box phase: expand up to REPS nodes per lane that has one (and room to park two more leaves) ----
        // (several expansions per turn of the loop spread its fixed costs -- ballots, refill and triangle-pass
        // decisions -- over more node steps)
#pragma unroll
        for (int rep = 0; rep < REPS; ++rep) {
            const bool boxing = node >= 0 && CPT_LEAVES_AT_MOST(kLeafCap - 2);
            if (COUNT) {
                uint64_t any_boxing = __ballot(boxing);
                if (lane == 0 && any_boxing) box_passes++;
            }
            if (boxing) {
                int2 refs;  // lref, rref
                int2 ids;   // COUNT: canonical ids of the two children
                bool hl, hr;
                double enter_l, enter_r;
                if (F32) {
                    const Rec4f r = fetch_node32<RESIDENT>(sc.nodes32, cache, cached_nodes, node);
                    refs = make_int2(__float_as_int(r.p3.x), __float_as_int(r.p3.y));
                    ids = make_int2(__float_as_int(r.p3.z), __float_as_int(r.p3.w));
                    float el, er;
                    v2f lox = {r.p0.x, r.p0.y}, loy = {r.p0.z, r.p0.w}, loz = {r.p1.x, r.p1.y};
                    v2f hix = {r.p1.z, r.p1.w}, hiy = {r.p2.x, r.p2.y}, hiz = {r.p2.z, r.p2.w};
                    slab32_pair(lox, loy, loz, hix, hiy, hiz, q32, hl, hr, el, er);
                    enter_l = (double)el;
                    enter_r = (double)er;
                } else if (SEL) {
                    const uint32_t rec = __umul24((uint32_t)node, CPT_PW_RESIDENT_NODE_BYTES);
                    // (volatile keeps the twelve reads apart: paired up as ds_read2_b64 they would take the LDS array twice as long)
                    const volatile lds_f64* px = (const volatile lds_f64*)(cache + rec + sel_x);  // [0], [1]: the left child's near and far plane; [9], [10]: the right child's
                    const volatile lds_f64* py = (const volatile lds_f64*)(cache + rec + sel_y);
                    const volatile lds_f64* pz = (const volatile lds_f64*)(cache + rec + sel_z);
                    const double lnx = px[0], lfx = px[1], rnx = px[9], rfx = px[10];
                    const double lny = py[0], lfy = py[1], rny = py[9], rfy = py[10];
                    const double lnz = pz[0], lfz = pz[1], rnz = pz[9], rfz = pz[10];
                    const lds_i32* pr = (const lds_i32*)(cache + rec + 144);
                    refs = make_int2(pr[0], pr[1]);
                    if (COUNT) ids = make_int2(pr[2], pr[3]);
                    bool sure_l, sure_r;
                    const dvec3 box_o = EYE ? dv(0, 0, 0) : o;
                    hl = slab_selected(lnx, lny, lnz, lfx, lfy, lfz, box_o, sr, sure_l, enter_l);
                    hr = slab_selected(rnx, rny, rnz, rfx, rfy, rfz, box_o, sr, sure_r, enter_r);
                    if (sr.exact_only || !(sure_l && sure_r)) {
                        if (COUNT) exact++;
                        hl = slab_hit(lnx, lny, lnz, lfx, lfy, lfz, box_o, d, enter_l);
                        hr = slab_hit(rnx, rny, rnz, rfx, rfy, rfz, box_o, d, enter_r);
                    }
                } else {
                    const Rec7 r = fetch_record<RESIDENT, kPlaneBytes>(node_recs, cache, 0, cached_nodes, node);
                    refs = make_int2(__double2loint(r.p6.x), __double2hiint(r.p6.x));
                    ids = make_int2(__double2loint(r.p6.y), __double2hiint(r.p6.y));
                    bool sure_l, sure_r;
                    const dvec3 box_o = EYE ? dv(0, 0, 0) : o;  // what the tests subtract from a plane: nothing when the record already has (x - 0 is x)
                    hl = slab_filtered(r.p0.x, r.p0.y, r.p1.x, r.p1.y, r.p2.x, r.p2.y, box_o, sr, sure_l, enter_l);
                    hr = slab_filtered(r.p3.x, r.p3.y, r.p4.x, r.p4.y, r.p5.x, r.p5.y, box_o, sr, sure_r, enter_r);
                    if (sr.exact_only || !(sure_l && sure_r)) {
                        if (COUNT) exact++;
                        hl = slab_hit(r.p0.x, r.p0.y, r.p1.x, r.p1.y, r.p2.x, r.p2.y, box_o, d, enter_l);
                        hr = slab_hit(r.p3.x, r.p3.y, r.p4.x, r.p4.y, r.p5.x, r.p5.y, box_o, d, enter_r);
                    }
                }
                if (COUNT) {  // both child boxes have been tested (reference: src/gpu/GPUBVH.cu:56-57), whatever the walk does next
                    vhash += visit_mix((uint32_t)ids.x) + visit_mix((uint32_t)ids.y);
                    if (obs.bitmap) {
                        uint8_t* row = obs.bitmap + (size_t)pos * (size_t)obs.bitmap_stride;
                        row[ids.x >> 3] |= (uint8_t)(1u << (ids.x & 7));
                        row[ids.y >> 3] |= (uint8_t)(1u << (ids.y & 7));
                    }
                }
                if (CULL) {
                    if (best_t >= 0) {
                        const double limit = best_t * 1.000001 + 1e-9;
                        hl = hl && !(enter_l > limit);
                        hr = hr && !(enter_r > limit);
                    }
                    if (hl && hr && refs.x >= 0 && refs.y >= 0 && enter_r < enter_l) {  // nearer child first
                        int tmp = refs.x; refs.x = refs.y; refs.y = tmp;
                    }
                }
                if (COUNT) steps++;
                // parked without a branch: the entry is written whether or not the child is a leaf that was hit (there
                // is room for two, and nothing beyond the top is ever read), only the top moves conditionally
                {   // (both entries in one LDS store: the first is the left child if that is to be parked, else the right one)
                    const bool park_l = hl && refs.x < 0, park_r = hr && refs.y < 0;
                    leaf_top[0] = (uint32_t)(park_l ? refs.x : refs.y);
                    leaf_top[CPT_WAVE] = (uint32_t)refs.y;
                    leaf_top += ((park_l ? CPT_WAVE : 0) + (park_r ? CPT_WAVE : 0));
                }
                bool tl = hl && refs.x >= 0, tr = hr && refs.y >= 0;
                if (LDS_STACK) {
                    if (tl) {
                        node = refs.x;
                        if (tr) { *node_top = (uint32_t)refs.y; node_top += CPT_WAVE; }
                    } else if (tr) {
                        node = refs.y;
                    } else if (node_top != node_base) {
                        node_top -= CPT_WAVE;
                        node = (int)*node_top;
                    } else {
                        node = -1;
                    }
                } else if (tl) {
                    node = refs.x;
                    if (tr) {
                        if (sp < kStackLds) my_nodes[sp * CPT_WAVE] = (uint32_t)refs.y;
                        else my_spill[(size_t)(sp - kStackLds) * spill_stride] = (uint32_t)refs.y;
                        ++sp;
                    }
                } else if (tr) {
                    node = refs.y;
                } else if (sp > 0) {
                    --sp;
                    if (sp < kStackLds) node = (int)my_nodes[sp * CPT_WAVE];
                    else node = (int)my_spill[(size_t)(sp - kStackLds) * spill_stride];
                } else {
                    node = -1;
                }
            }
        }

        // ---- triangle phase: when enough lanes have a parked leaf, or nothing else can run ----
        uint64_t have_leaf = __ballot(leaf_top != leaf_base);
        if (have_leaf) {
            uint64_t have_node = __ballot(node >= 0);
            uint64_t nearly_full = __ballot(!CPT_LEAVES_AT_MOST(kLeafCap - 2));
            if ((uint32_t)__popcll(have_leaf) >= leaf_go || (have_node & ~nearly_full) == 0 || (uint32_t)__popcll(nearly_full) >= full_go) {
                if (COUNT && lane == 0) leaf_passes++;
                if (leaf_top != leaf_base) {
                    leaf_top -= CPT_WAVE;
                    int tri = (int)~*leaf_top;
                    Rec7 q;
                    if (SEL) {
                        const lds_byte* rec = cache + tri_lds + __umul24((uint32_t)tri, CPT_PW_RESIDENT_TRI_BYTES);
                        q.p0 = lds_piece(rec); q.p1 = lds_piece(rec + 16); q.p2 = lds_piece(rec + 32); q.p3 = lds_piece(rec + 48);
                        q.p4 = lds_piece(rec + 64); q.p5 = lds_piece(rec + 80); q.p6 = lds_piece(rec + 96);
                    } else {
                        q = fetch_record<RESIDENT, kPlaneBytes>(tri_recs, cache, tri_first, cached_tris, tri);
                    }
                    const uint32_t flags = (uint32_t)__double2hiint(q.p6.x);
                    const uint32_t rank = (uint32_t)__double2loint(EYE ? q.p6.x : q.p6.y);  // see k_eye_records
                    bool gate = true;
                    if (F32 && sc.n_nodes > 0) {  // (a one-triangle scene has no boxes at all) the exact test of the leaf's own box: the bounds of the three vertices (reference: src/scene/bvh.cpp:128-146)
                        const double lx = fmin(fmin(q.p0.x, q.p1.y), q.p3.x), hx = fmax(fmax(q.p0.x, q.p1.y), q.p3.x);
                        const double ly = fmin(fmin(q.p0.y, q.p2.x), q.p3.y), hy = fmax(fmax(q.p0.y, q.p2.x), q.p3.y);
                        const double lz = fmin(fmin(q.p1.x, q.p2.y), q.p4.x), hz = fmax(fmax(q.p1.x, q.p2.y), q.p4.x);
                        bool sure;
                        double enter;
                        gate = slab_filtered(lx, ly, lz, hx, hy, hz, o, sr, sure, enter);
                        if (sr.exact_only || !sure) {
                            if (COUNT) exact++;
                            gate = slab_hit(lx, ly, lz, hx, hy, hz, o, d, enter);
                        }
                        if (CULL && best_t >= 0 && enter > best_t * 1.000001 + 1e-9) gate = false;
                    }
                    double t;
                    if (gate) {
                        if (COUNT) tris++;
                        const double numerator = EYE ? q.p6.y : tri_numerator(o, dv(q.p0.x, q.p0.y, q.p1.x), dv(q.p4.y, q.p5.x, q.p5.y));
                        if (tri_hit_regs(q.p0, q.p1, q.p2, q.p3, q.p4, q.p5, flags, numerator, o, d, t)) {
                            if (best_t < 0 || t < best_t || (t == best_t && rank < best_rank)) { best_t = t; best_tri = tri; best_rank = rank; }
                        }
                    }
                }
            }
        }

        // ---- retire ----
        if (has_ray && node < 0 && leaf_top == leaf_base) {
            store_hit(hit, pos, best_t, best_tri);
            if (COUNT && obs.vhash) obs.vhash[pos] = vhash;
            has_ray = false;
        }
    }
    if (COUNT) {
        if (steps) atomicAdd(&work[3], steps);
        if (tris) atomicAdd(&work[4], tris);
        if (exact) atomicAdd(&work[5], exact);
        if (box_passes) atomicAdd(&work[6], box_passes);
        if (leaf_passes) atomicAdd(&work[7], leaf_passes);
    }
}

// ---- shade + compact -------------------------------------------------------------------------
// reference: the shading half of pathTrace (src/improved_path_tracer.cu:77-215)
// FIRST: the stream is the one k_generate wrote without roulette: geometry only, position = slot, medium 1,
// throughput (1,1,1).  A slot outside the image holds a ray that leaves the scene, so it ends here as a miss.
// Register budget: 96 VGPRs (five waves per SIMD) at the price of 28 spilled registers.  Run alone the kernel is 5 %
// slower than with 128 VGPRs and no spills, but the pipeline keeps two batches in flight, and beside the other
// batch's extend kernel the smaller footprint wins: +4 % on whole renders (six waves: the same; eight: -17 %).
#ifndef CPT_SHADE_MIN_WAVES
#define CPT_SHADE_MIN_WAVES 5
#endif
template <bool COUNT, bool FIRST>
__global__ void __launch_bounds__(CPT_SHADE_BLOCK, CPT_SHADE_MIN_WAVES) k_shade(DScene sc, DParams P, RayStream in, HitBuf hit, RayStream out,
                                                           Radiance col, int batch_s0, int bounce,
                                                           const uint32_t* __restrict__ counter_in, uint32_t* counter_out, uint32_t cap,
                                                           unsigned long long* work) {
    __shared__ uint32_t s_count[CPT_PW_SEGMENTS];
    if (threadIdx.x < CPT_PW_SEGMENTS) s_count[threadIdx.x] = counter_in[threadIdx.x * 32u];
    __syncthreads();
    uint32_t longest = 0;
    for (int k = 0; k < CPT_PW_SEGMENTS; ++k) longest = s_count[k] > longest ? s_count[k] : longest;
    // work items: (segment, 256-ray piece of it), enumerated piece-major so that the workgroups running at any one
    // time are working on -- and claiming output in -- all the segments
    const uint32_t pieces = (longest + CPT_SHADE_BLOCK - 1) / CPT_SHADE_BLOCK;
    unsigned long long hits = 0, segs = 0;
    for (uint32_t item = blockIdx.x; item < pieces * CPT_PW_SEGMENTS; item += gridDim.x) {
        const uint32_t seg = item % CPT_PW_SEGMENTS, piece = item / CPT_PW_SEGMENTS;
        const uint32_t n_seg = s_count[seg];
        if (piece * CPT_SHADE_BLOCK >= n_seg) continue;  // the whole workgroup: nothing reaches block_claim
        const uint32_t off = piece * CPT_SHADE_BLOCK + threadIdx.x;
        const uint32_t q = seg * cap + off;
        bool go = false;
        uint32_t slot = 0;
        PathRegs s;
        if (off < n_seg) {
            int tri;
            double t;
            // the ray is asked for together with its hit record, not after it: one memory round trip less on the
            // critical path of a kernel that waits on memory most of the time (a miss has read 84 bytes for nothing)
            slot = FIRST ? q : CPT_NT_LOAD(in.slot[q]);
            if (FIRST) {
                load_ray_geometry(in, q, s.o, s.d);
                s.medium = 1.0;
                s.atten = dv(1, 1, 1);
            } else {
                load_ray(in, q, s);
            }
            load_hit(hit, q, t, tri);
            if (COUNT) { segs++; hits += (tri >= 0); }
            if (tri >= 0) {
                int k = (int)fast_div(slot, P.by_pix_slots), ps = (int)slot - k * P.n_pix_slots;
                int i, j;
                slot_to_pixel(P, ps, i, j);
                uint32_t pixel = (uint32_t)(i + j * P.w), sample = (uint32_t)(batch_s0 + k);
                dvec3 emitted;
                int lobe = choose_lobe(sc, P, pixel, sample, bounce, tri, s.atten, emitted);
                if (emitted.x != 0.0 || emitted.y != 0.0 || emitted.z != 0.0) {  // adding zero changes nothing: skip the round trip
                    if (!FIRST && col.lit[slot]) {
                        col.cx[slot] += emitted.x; col.cy[slot] += emitted.y; col.cz[slot] += emitted.z;
                    } else {  // the path's first emitter: 0 + e = e, stored without reading what the slot held before
                        col.cx[slot] = emitted.x; col.cy[slot] = emitted.y; col.cz[slot] = emitted.z;
                        col.lit[slot] = 1;
                    }
                }
                if (lobe != CPT_LOBE_NONE) {
                    scatter(lobe, sc, P, pixel, sample, bounce, t, tri, s);
                    go = roulette(P, pixel, sample, bounce + 1, s);
                }
            }
        }
        uint32_t p = block_claim(counter_out + seg * 32u, go);
        if (go) store_ray(out, seg * cap + p, s, slot);
    }
    if (COUNT) {
        if (segs) atomicAdd(&work[1], segs);
        if (hits) atomicAdd(&work[2], hits);
    }
}

// ---- resolve ---------------------------------------------------------------------------------
// accum[pixel] += radiance of sample batch_s0, then batch_s0 + 1, ... : the order pathTraceKernel sums
// its samples in (reference: src/path_tracer.cu:419-435).
__global__ void __launch_bounds__(CPT_BLOCK) k_resolve(DParams P, Radiance col, int batch_n, double* __restrict__ accum) {
    for (int ps = blockIdx.x * blockDim.x + threadIdx.x; ps < P.n_pix_slots; ps += gridDim.x * blockDim.x) {
        int i, j;
        if (!slot_to_pixel(P, ps, i, j)) continue;
        double* px = accum + 3 * (size_t)(i + j * P.w);
        dvec3 sum = dv(px[0], px[1], px[2]);
        for (int k = 0; k < batch_n; ++k) {
            size_t slot = (size_t)k * P.n_pix_slots + ps;
            // a path that met no emitter has radiance +0, and adding it would change nothing (a sum is never -0: it starts
            // from the caller's buffer and only ever receives finite throughput times ke)
            if (col.lit[slot]) sum = sum + dv(col.cx[slot], col.cy[slot], col.cz[slot]);
        }
        px[0] = sum.x; px[1] = sum.y; px[2] = sum.z;
    }
}

// ---- one thread per pixel ----------------------------------------------------------------------
// All samples and bounces of a pixel in registers; the per-pixel loop of the reference's live kernel.
// reference: pathTraceKernel / pathTraceDOFKernel (src/path_tracer.cu:403-492)
template <bool COUNT>
__global__ void __launch_bounds__(CPT_BLOCK) k_megakernel(DScene sc, DParams P, double* __restrict__ accum,
                                                          unsigned long long* work) {
    CPT_DECLARE_STACK
    unsigned long long steps = 0, tris = 0, hits = 0, segs = 0;
    for (int ps = blockIdx.x * blockDim.x + threadIdx.x; ps < P.n_pix_slots; ps += gridDim.x * blockDim.x) {
        int i, j;
        if (!slot_to_pixel(P, ps, i, j)) continue;
        uint32_t pixel = (uint32_t)(i + j * P.w);
        double* px = accum + 3 * (size_t)pixel;
        dvec3 sum = dv(px[0], px[1], px[2]);
        for (int sample = P.s0; sample < P.s1; ++sample) {
            PathRegs s;
            first_ray(sc, P, i, j, sample, s.o, s.d);
            s.medium = 1.0;
            s.atten = dv(1, 1, 1);
            s.color = dv(0, 0, 0);
            bool go = roulette(P, pixel, (uint32_t)sample, 0, s);
            for (int bounce = 0; go && bounce <= P.depth; ++bounce) {
                double t;
                int tri;
                if (COUNT) {
                    TraceRec rec = {0, 0, 0, 0, 0, nullptr};
                    traverse<true>(sc, s.o, s.d, my_stack, t, tri, &rec);
                    steps += (unsigned long long)rec.node_steps;
                    tris += (unsigned long long)rec.tri_tests;
                    segs++;
                    hits += (tri >= 0);
                } else {
                    traverse<false>(sc, s.o, s.d, my_stack, t, tri, nullptr);
                }
                go = shade_bounce(sc, P, pixel, (uint32_t)sample, bounce, t, tri, s);
                if (go) go = roulette(P, pixel, (uint32_t)sample, bounce + 1, s);
            }
            sum = sum + s.color;
        }
        px[0] = sum.x; px[1] = sum.y; px[2] = sum.z;
    }
    if (COUNT) {
        if (segs) atomicAdd(&work[1], segs);
        if (hits) atomicAdd(&work[2], hits);
        if (steps) atomicAdd(&work[3], steps);
        if (tris) atomicAdd(&work[4], tris);
    }
}

// ---- tonemap -----------------------------------------------------------------------------------
// reference: src/path_tracer.cu:436-443
// dst[i] += src[i]: radiance sums of another GPU, already copied to this one, joining this GPU's (the reduce of a render
// split over several GPUs of one process)
__global__ void __launch_bounds__(CPT_BLOCK) k_add_values(double* __restrict__ dst, const double* __restrict__ src, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = dst[i] + src[i];
}

__global__ void __launch_bounds__(CPT_BLOCK) k_tonemap(const double* __restrict__ accum, long long n_values, int spp,
                                                       uint8_t* __restrict__ rgb) {
    for (long long k = blockIdx.x * blockDim.x + threadIdx.x; k < n_values; k += (long long)gridDim.x * blockDim.x) {
        double c = accum[k] / spp;
        c = fmin(fmax(c, 0.0), 1.0);
        rgb[k] = (uint8_t)(int)(255.0 * c);
    }
}

// ---- parity hooks ------------------------------------------------------------------------------
__global__ void __launch_bounds__(CPT_BLOCK) k_trace_rays(DScene sc, int n, const double* __restrict__ o3,
                                                          const double* __restrict__ d3, int* tri_out, double* t_out,
                                                          double* normal_out, unsigned long long* vhash_out,
                                                          int* counts_out, uint8_t* bitmap, int bitmap_stride) {
    CPT_DECLARE_STACK
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        dvec3 o = dv(o3 + 3 * (size_t)r), d = dv(d3 + 3 * (size_t)r);
        TraceRec rec = {0, 0, 0, 0, 0, bitmap ? bitmap + (size_t)r * bitmap_stride : nullptr};
        double t;
        int tri;
        traverse<true>(sc, o, d, my_stack, t, tri, &rec);
        tri_out[r] = tri;
        t_out[r] = t;
        if (normal_out) {
            dvec3 N = tri >= 0 ? shading_normal(sc.shade + tri, o, d, t) : dv(0, 0, 0);
            normal_out[3 * (size_t)r] = N.x; normal_out[3 * (size_t)r + 1] = N.y; normal_out[3 * (size_t)r + 2] = N.z;
        }
        if (vhash_out) vhash_out[r] = rec.visit_hash;
        if (counts_out) {
            counts_out[4 * (size_t)r] = rec.node_steps; counts_out[4 * (size_t)r + 1] = rec.box_tests;
            counts_out[4 * (size_t)r + 2] = rec.tri_tests; counts_out[4 * (size_t)r + 3] = rec.max_stack;
        }
    }
}

__global__ void __launch_bounds__(CPT_BLOCK) k_camera_rays(DScene sc, int w, int h, double jx, double jy, double* o3,
                                                           double* d3) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < w * h; p += gridDim.x * blockDim.x) {
        int i = p % w, j = p / w;
        dvec3 o, d;
        camera_ray(sc.cam, (i + jx) / double(w), (j + jy) / double(h), o, d);
        o3[3 * (size_t)p] = o.x; o3[3 * (size_t)p + 1] = o.y; o3[3 * (size_t)p + 2] = o.z;
        d3[3 * (size_t)p] = d.x; d3[3 * (size_t)p + 1] = d.y; d3[3 * (size_t)p + 2] = d.z;
    }
}

__global__ void __launch_bounds__(CPT_BLOCK) k_path_radiance(DScene sc, DParams P, int n, const int* __restrict__ ijs,
                                                             double* rad) {
    CPT_DECLARE_STACK
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        int i = ijs[3 * r], j = ijs[3 * r + 1], sample = ijs[3 * r + 2];
        uint32_t pixel = (uint32_t)(i + j * P.w);
        PathRegs s;
        first_ray(sc, P, i, j, sample, s.o, s.d);
        s.medium = 1.0;
        s.atten = dv(1, 1, 1);
        s.color = dv(0, 0, 0);
        bool go = roulette(P, pixel, (uint32_t)sample, 0, s);
        for (int bounce = 0; go && bounce <= P.depth; ++bounce) {
            double t;
            int tri;
            traverse<false>(sc, s.o, s.d, my_stack, t, tri, nullptr);
            go = shade_bounce(sc, P, pixel, (uint32_t)sample, bounce, t, tri, s);
            if (go) go = roulette(P, pixel, (uint32_t)sample, bounce + 1, s);
        }
        rad[3 * (size_t)r] = s.color.x; rad[3 * (size_t)r + 1] = s.color.y; rad[3 * (size_t)r + 2] = s.color.z;
    }
}

__global__ void k_probe_math(int n, const double* a, const double* b, double* out_sqrt, double* out_div) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        out_sqrt[k] = sqrt(a[k]);
        out_div[k] = a[k] / (b ? b[k] : 3.0);
    }
}

__global__ void k_probe_rng(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out,
                            double* out_u) {
    uint32_t r[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, r);
    for (int q = 0; q < 4; ++q) out[q] = r[q];
    out_u[0] = u01_open0(r[0], r[1]);
    out_u[1] = u01_open0(r[2], r[3]);
}

// ---- launchers (called from the C ABI in capi.cpp) ------------------------------------------------
static inline int grid_for(long long n_threads, int max_blocks) {
    long long b = (n_threads + CPT_BLOCK - 1) / CPT_BLOCK;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

void launch_generate(const LaunchCfg& L, const DScene& sc, const DParams& P, const RayStream& out, const Radiance& col,
                     int batch_s0, int batch_n) {
    int g = grid_for((long long)batch_n * P.n_pix_slots, L.max_blocks);
    hipLaunchKernelGGL(k_generate, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, P, out, col, batch_s0, batch_n);
}
void launch_nodes32(hipStream_t st, const cpt_node* d_nodes, int n_nodes, void* d_nodes32) {
    if (n_nodes < 1) return;
    int g = (n_nodes + CPT_BLOCK - 1) / CPT_BLOCK;
    hipLaunchKernelGGL(k_nodes32, dim3(g > 4096 ? 4096 : g), dim3(CPT_BLOCK), 0, st, d_nodes, n_nodes, (Node32*)d_nodes32);
}
void launch_eye_records(hipStream_t st, const DScene& sc, cpt_node* d_nodes_eye, cpt_tri* d_tris_eye) {
    const int n = sc.n_nodes + sc.n_tris, g = (n + CPT_BLOCK - 1) / CPT_BLOCK;
    hipLaunchKernelGGL(k_eye_records, dim3(g > 8192 ? 8192 : g), dim3(CPT_BLOCK), 0, st, sc.nodes_pw, sc.n_nodes, sc.tris, sc.n_tris,
                       dv(sc.cam.eye[0], sc.cam.eye[1], sc.cam.eye[2]), d_nodes_eye, d_tris_eye);
}
void launch_dense_counts(const LaunchCfg& L, uint32_t* counts, uint32_t n, uint32_t cap) {
    hipLaunchKernelGGL(k_dense_counts, dim3(1), dim3(64), 0, L.stream, counts, n, cap);
}
void launch_extend(const LaunchCfg& L, const DScene& sc, const RayStream& in, const HitBuf& hit, const uint32_t* counter,
                   uint32_t cap, long long max_entries, unsigned long long* work, const ExtendObserve& obs) {
    int g = grid_for(max_entries, L.max_blocks);
    if (work) hipLaunchKernelGGL(k_extend<true>, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, in, hit, counter, cap, work, obs);
    else hipLaunchKernelGGL(k_extend<false>, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, in, hit, counter, cap, work, obs);
}
void launch_extend_pw(const LaunchCfg& L, const DScene& sc, const RayStream& in, const HitBuf& hit, const uint32_t* counter,
                      uint32_t cap, uint32_t* cursor, uint32_t* spill, unsigned long long* work, const ExtendObserve& obs) {
    int g = L.pw_blocks;
    const dvec3 eye = dv(sc.cam.eye[0], sc.cam.eye[1], sc.cam.eye[2]);
    const bool use_eye = L.pw_eye && !L.pw_f32;
#define CPT_PW_LAUNCH(COUNT, CULL, RES, F32, REPS, EYE)                                                                                  \
    hipLaunchKernelGGL((k_extend_pw<COUNT, CULL, RES, F32, REPS, EYE>), dim3(g), dim3(CPT_PW_THREADS), 0, L.stream, sc, in, hit, counter, \
                       cursor, cap, (uint32_t)L.pw_chunk, (uint32_t)L.pw_refill, (uint32_t)L.pw_leaf_go,                       \
                       (uint32_t)L.pw_full_go, spill, work, obs, eye)
#define CPT_PW_LAUNCH_EYE(COUNT, CULL, MODE, F32, REPS)                                   \
    if (!F32 && use_eye) CPT_PW_LAUNCH(COUNT, CULL, MODE, F32, REPS, (!F32));                 \
    else CPT_PW_LAUNCH(COUNT, CULL, MODE, F32, REPS, false)
#define CPT_PW_LAUNCH_REPS(COUNT, CULL, MODE, F32)                                   \
    if (L.pw_box_repeat == 2) { CPT_PW_LAUNCH_EYE(COUNT, CULL, MODE, F32, 2); }                 \
    else { CPT_PW_LAUNCH_EYE(COUNT, CULL, MODE, F32, 3); }
#define CPT_PW_LAUNCH_MODE(COUNT, CULL, F32)                                            \
    switch (L.pw_mode) {                                                                \
        case 2: CPT_PW_LAUNCH_REPS(COUNT, CULL, 2, F32); break;                         \
        case 1: CPT_PW_LAUNCH_REPS(COUNT, CULL, 1, F32); break;                         \
        default: CPT_PW_LAUNCH_REPS(COUNT, CULL, 0, F32); break;                        \
    }
    const int which = (L.pw_f32 ? 4 : 0) | (work ? 2 : 0) | (L.pw_cull ? 1 : 0);
    switch (which) {
        case 0: CPT_PW_LAUNCH_MODE(false, false, false); break;
        case 1: CPT_PW_LAUNCH_MODE(false, true, false); break;
        case 2: CPT_PW_LAUNCH_MODE(true, false, false); break;
        case 3: CPT_PW_LAUNCH_MODE(true, true, false); break;
        case 4: CPT_PW_LAUNCH_MODE(false, false, true); break;
        case 5: CPT_PW_LAUNCH_MODE(false, true, true); break;
        case 6: CPT_PW_LAUNCH_MODE(true, false, true); break;
        default: CPT_PW_LAUNCH_MODE(true, true, true); break;
    }
#undef CPT_PW_LAUNCH_MODE
#undef CPT_PW_LAUNCH_REPS
#undef CPT_PW_LAUNCH_EYE
#undef CPT_PW_LAUNCH
}
void launch_shade(const LaunchCfg& L, const DScene& sc, const DParams& P, const RayStream& in, const HitBuf& hit,
                  const RayStream& out, const Radiance& col, int batch_s0, int bounce, const uint32_t* counter_in,
                  uint32_t* counter_out, uint32_t cap, long long max_entries, unsigned long long* work) {
    // Five workgroups fit a CU (CPT_SHADE_MIN_WAVES waves per SIMD), so the default cap of eight per CU would run as
    // one full round and one that fills three fifths of the chip; five times the cap is forty per CU, eight full rounds
    // (run alone the kernel is 8 % faster for it).
    const long long b = (max_entries + CPT_SHADE_BLOCK - 1) / CPT_SHADE_BLOCK, most = (long long)L.max_blocks * CPT_SHADE_MIN_WAVES;
    int g = (int)(b < 1 ? 1 : (b > most ? most : b));
    const bool first = bounce == 0 && !P.rr;  // the stream k_generate wrote in its lean form
#define CPT_SHADE_LAUNCH(COUNT, FIRST)                                                                                      \
    hipLaunchKernelGGL((k_shade<COUNT, FIRST>), dim3(g), dim3(CPT_SHADE_BLOCK), 0, L.stream, sc, P, in, hit, out, col, batch_s0, \
                       bounce, counter_in, counter_out, cap, work)
    if (work) { if (first) CPT_SHADE_LAUNCH(true, true); else CPT_SHADE_LAUNCH(true, false); }
    else { if (first) CPT_SHADE_LAUNCH(false, true); else CPT_SHADE_LAUNCH(false, false); }
#undef CPT_SHADE_LAUNCH
}
void launch_resolve(const LaunchCfg& L, const DParams& P, const Radiance& col, int batch_n, double* accum) {
    int g = grid_for(P.n_pix_slots, L.max_blocks);
    hipLaunchKernelGGL(k_resolve, dim3(g), dim3(CPT_BLOCK), 0, L.stream, P, col, batch_n, accum);
}
void launch_megakernel(const LaunchCfg& L, const DScene& sc, const DParams& P, double* accum, unsigned long long* work) {
    int g = grid_for(P.n_pix_slots, 1 << 30);
    if (work) hipLaunchKernelGGL(k_megakernel<true>, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, P, accum, work);
    else hipLaunchKernelGGL(k_megakernel<false>, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, P, accum, work);
}
void launch_add_values(hipStream_t st, double* dst, const double* src, long long n_values, int max_blocks) {
    hipLaunchKernelGGL(k_add_values, dim3(grid_for(n_values, max_blocks)), dim3(CPT_BLOCK), 0, st, dst, src, n_values);
}
void launch_tonemap(const LaunchCfg& L, const double* accum, long long n_values, int spp, uint8_t* rgb) {
    int g = grid_for(n_values, L.max_blocks);
    hipLaunchKernelGGL(k_tonemap, dim3(g), dim3(CPT_BLOCK), 0, L.stream, accum, n_values, spp, rgb);
}
void launch_trace_rays(const LaunchCfg& L, const DScene& sc, int n, const double* o3, const double* d3, int* tri, double* t,
                       double* normal, unsigned long long* vhash, int* counts, uint8_t* bitmap, int bitmap_stride) {
    int g = grid_for(n, L.max_blocks);
    hipLaunchKernelGGL(k_trace_rays, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, n, o3, d3, tri, t, normal, vhash, counts, bitmap, bitmap_stride);
}
void launch_camera_rays(const LaunchCfg& L, const DScene& sc, int w, int h, double jx, double jy, double* o3, double* d3) {
    int g = grid_for((long long)w * h, L.max_blocks);
    hipLaunchKernelGGL(k_camera_rays, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, w, h, jx, jy, o3, d3);
}
void launch_path_radiance(const LaunchCfg& L, const DScene& sc, const DParams& P, int n, const int* ijs, double* rad) {
    int g = grid_for(n, L.max_blocks);
    hipLaunchKernelGGL(k_path_radiance, dim3(g), dim3(CPT_BLOCK), 0, L.stream, sc, P, n, ijs, rad);
}
void launch_probe_math(const LaunchCfg& L, int n, const double* a, const double* b, double* s, double* q) {
    int g = grid_for(n, L.max_blocks);
    hipLaunchKernelGGL(k_probe_math, dim3(g), dim3(CPT_BLOCK), 0, L.stream, n, a, b, s, q);
}
void launch_probe_rng(const LaunchCfg& L, const uint32_t c[4], const uint32_t k[2], uint32_t* out, double* out_u) {
    hipLaunchKernelGGL(k_probe_rng, dim3(1), dim3(1), 0, L.stream, c[0], c[1], c[2], c[3], k[0], k[1], out, out_u);
}
