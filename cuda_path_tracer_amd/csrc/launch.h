// Types shared between the kernels (kernels.hip) and the C ABI implementation (capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpt_amd.h"
#include "dvec.h"

// Everything the shading half of a bounce needs to know about a triangle, in one record, so a hit
// costs one dependent fetch (hit -> record) instead of a chain (hit -> triangle -> material, plus the
// vertex normals), and everything that is the same at every hit of the triangle is worked out once, when
// the scene is prepared, by the device with the very expressions the reference evaluates per hit
// (src/path_tracer.cu:272-293, 294-308, 340-343): the normalised face normal and its second
// normalisation, the lobe probabilities, the diffuse and mirror throughput factors, 1 / p_trans, r0.
// Four 128-byte lines: the first serves a flat-shaded diffuse hit on its own, the second adds what the
// mirror and dielectric lobes need, the last two are read for smooth-shaded triangles (and log kt inside
// a medium).
struct __attribute__((aligned(128))) ShadeRec {
    uint32_t tri_flags, mat_flags;     // CPT_TRI_*, CPT_MAT_*
    double n1[3];                      // normalize(face normal): the shading normal of a flat triangle
    double n2[3];                      // normalize(n1): what the diffuse and mirror lobes normalise it to again
    double ke[3];
    double diff_w[3];                  // (kd / pi) * 1 / ((1 / pi) * p_diff); holds kd until the device fills it in
    double p_diff, p_spec;             // lobe probabilities
    double inv_p_trans;                // 1 / p_trans; holds p_trans until the device fills it in
    double ks[3], kr[3], kt[3];        // second line
    double spec_w[3];                  // ks * 1.0 / p_spec
    double index;
    double r0;                         // ((1 - index) / (1 + index))^2, the same from either side
    double pad1[2];
    double a[3], b[3], c[3];           // third line: vertices, smooth-shaded triangles only
    double log_kt[3];                  // log|kt|, so that kt^t is one exp() at a hit inside the medium
    double ca[3];                      // c - a, and ...
    double denom;                      // ... length(cross(b - a, c - a)): the ray-independent operands of the barycentric weights (src/gpu/GPUTrimesh.cuh:91-97)
    double na[3], nb[3], nc[3];        // fourth line: vertex normals
    double pad3[7];
};
static_assert(sizeof(ShadeRec) == 512, "ShadeRec is four 128-byte lines");

struct LobeProb {
    double diff, spec, trans;
};
// reference: src/path_tracer.cu:272-279.  Evaluated on the host when a scene is uploaded; a
// material's probabilities are the same at every hit.
CPT_HD LobeProb lobe_probabilities(const double* kd3, const double* ks3, const double* kt3) {
    double spec = (ks3[0] + ks3[1] + ks3[2]) / 3.0;
    double diff = (kd3[0] + kd3[1] + kd3[2]) / 3.0;
    double trans = (kt3[0] + kt3[1] + kt3[2]) / 3.0;
    double total = spec + diff + trans;
    LobeProb p;
    p.diff = diff / total;
    p.spec = spec / total;
    p.trans = trans / total;
    return p;
}

struct DScene {
    const cpt_node* __restrict__ nodes;      // the tree as the caller (or the device build) gave it: hooks, one-ray-per-thread kernels
    const cpt_node* __restrict__ nodes_pw;   // the same records with each node's children in stack-minimal order (persistent extend kernel);
                                             // the same array when no such copy was made
    const cpt_tri* __restrict__ tris;       // intersection records (extend)
    const ShadeRec* __restrict__ shade;     // shading records (shade), one per triangle
    const void* __restrict__ nodes32;       // 64-byte binary32 copies of the node records, boxes rounded outwards (extend variants 3 and 4)
    // nodes_pw and tris restated for rays that start at the pin-hole camera's eye (k_eye_records): every box plane with the
    // eye's coordinate already subtracted, every triangle with its plane term for that origin
    const cpt_node* __restrict__ nodes_eye;
    const cpt_tri* __restrict__ tris_eye;
    cpt_camera cam;
    int n_nodes, n_tris;
};

// Division of a number below 2^28 by a divisor fixed at launch, as a multiplication and a shift: with
// L = ceil(log2 d) and m = ceil(2^(28+L) / d) < 2^29, (n * m) >> (28 + L) == n / d for every n < 2^28
// (Granlund and Montgomery's round-up method).  Path slots, pixel slots and tile numbers are all below 2^28.
struct FastDiv {
    uint32_t m, s;
};
inline FastDiv fast_div_for(uint32_t d) {
    uint32_t L = 0;
    while ((1u << L) < d) ++L;
    FastDiv f;
    f.s = 28 + L;
    f.m = (uint32_t)(((1ull << f.s) + d - 1) / d);
    return f;
}
CPT_HD uint32_t fast_div(uint32_t n, FastDiv f) { return (uint32_t)(((unsigned long long)n * f.m) >> f.s); }

struct DParams {  // cpt_render_params plus what the kernels derive from it
    int w, h, s0, s1, depth;
    int dof, rr;
    uint32_t seed_lo, seed_hi;
    double focal_distance, aperture;
    int tiles_x, tiles_y;  // 8x8 pixel tiles
    int n_pix_slots;       // 64 per tile of the window being rendered (the whole image: tiles_x*tiles_y*64)
    int tile0;             // first tile of the window
    FastDiv by_pix_slots, by_tiles_x;  // division by n_pix_slots and by tiles_x
};


// Path state of the wavefront pipeline, structure-of-arrays in HBM.
//
// The rays of one bounce form a dense stream indexed by queue position 0 .. count-1: extend reads
// position p and writes the hit for p, shade reads p and writes the surviving paths, compacted, at
// consecutive positions of the other stream.  Nothing is gathered through an index; every access
// is a contiguous run per wave.  `slot` says which path (pixel slot and sample) a position holds.
// Components are stored in pairs so that every access is a 16-byte load or store per lane (half the
// memory instructions of one array per component).
typedef double cpt_v2d __attribute__((ext_vector_type(2)));
struct RayStream {
    cpt_v2d *oxy, *ozdx, *dyz;  // ray origin x,y | origin z, direction x | direction y,z: all extend reads
    cpt_v2d *mdax, *ayz;        // refractive index of the current medium, throughput x | throughput y,z
    uint32_t* slot;             // path slot = k * n_pix_slots + pixel slot
};
#define CPT_STREAM_DOUBLES 10
#define CPT_STREAM_BYTES (CPT_STREAM_DOUBLES * 8 + 4)
struct HitRec {             // 16 bytes: one store in extend, one load in shade
    double t;               // hit distance (meaningful when tri >= 0)
    int32_t tri;            // triangle index or -1
    int32_t pad_;
};
struct HitBuf {             // extend -> shade, indexed by queue position
    HitRec* rec;
};
struct Radiance {           // radiance gathered per path, indexed by slot; touched only at emitters
    double *cx, *cy, *cz;   // meaningful only where `lit` is set: generate clears the flag, not the 24 bytes of radiance
    uint8_t* lit;           // 1 once an emitter has contributed to the slot's path
};
#define CPT_SLOT_BYTES (2 * CPT_STREAM_BYTES + 16 + 24 + 1)  // device bytes per path slot of a batch: two streams, hit record, radiance, its flag

// traversal stack of the persistent-wave extend kernel: entries in LDS, and the deepest tree it accepts
#ifndef CPT_PW_STACK_LDS_HOST
#define CPT_PW_STACK_LDS_HOST 12
#endif
#define CPT_PW_STACK_MAX 62
// The rays of a stream are handed to the persistent extend kernel's waves in small chunks.  One claim
// counter for the whole stream would serialise the claims (an address takes on the order of 10^8
// atomics per second), so the stream is cut into CPT_PW_SEGMENTS equal segments, each with a counter
// of its own in its own 128-byte line; a wave works through its home segment and then helps with the
// others.
#define CPT_PW_SEGMENTS 16
#define CPT_PW_CURSOR_WORDS (CPT_PW_SEGMENTS * 32)  // u32 words of claim counters per bounce; a stream's segment lengths take as many
// Positions a stream of n rays needs, and the capacity of one of its CPT_PW_SEGMENTS segments.
inline uint32_t cpt_segment_cap(size_t n) { return (uint32_t)(((n + CPT_PW_SEGMENTS - 1) / CPT_PW_SEGMENTS + 63) / 64 * 64); }
inline size_t cpt_stream_positions(size_t n) { return (size_t)cpt_segment_cap(n) * CPT_PW_SEGMENTS; }
#ifndef CPT_PW_CACHE_RECORDS_HOST
#define CPT_PW_CACHE_RECORDS_HOST 528  // 128-byte records of the scene the persistent extend kernel keeps in LDS (its 66 KB record cache) ...
#endif
// ... and how many the exact variants keep, which store only the seven 16-byte pieces of a record that are ever read
// (603 for 66 KB; kept odd, see planar_record)
#define CPT_PW_PLANAR_SLOTS_HOST (((CPT_PW_CACHE_RECORDS_HOST * 128 / 112) & 1) ? (CPT_PW_CACHE_RECORDS_HOST * 128 / 112) : (CPT_PW_CACHE_RECORDS_HOST * 128 / 112) - 1)
#ifndef CPT_PW_THREADS_HOST
#define CPT_PW_THREADS_HOST 1024
#endif
// The exact variants' instance for a scene that lives in LDS whole ("resident") keeps its records in a layout of its
// own: a node as 21 doubles -- per child and axis the triple (upper, lower, upper) plane, so that a lane reads the plane
// its ray meets first and the one it meets last as two consecutive doubles at an offset it fixes once per ray, then the
// two child references and the two canonical ids, then one double of padding that makes the stride an odd number of
// 8-byte bank pairs -- and a triangle as the seven 16-byte pieces of its record that are ever read (an odd number of
// 16-byte bank groups).  Such a scene's tree is low: the instance without culling holds CPT_PW_RESIDENT_STACK node-stack
// entries per lane -- enough for any tree over the 285 triangles that fit, whose walk in stack-minimal order postpones
// at most log2(inner nodes) + 1 of them -- and gives the rest of the stack's LDS to the records.
#define CPT_PW_RESIDENT_NODE_BYTES 168
#define CPT_PW_RESIDENT_TRI_BYTES 112
#define CPT_PW_RESIDENT_STACK 9
inline int cpt_pw_resident_stack(bool cull) { return cull ? CPT_PW_STACK_LDS_HOST : CPT_PW_RESIDENT_STACK; }
inline long long cpt_pw_resident_cache_bytes(bool cull) {
    return (long long)CPT_PW_CACHE_RECORDS_HOST * 128 + (long long)(CPT_PW_STACK_LDS_HOST - cpt_pw_resident_stack(cull)) * 4 * CPT_PW_THREADS_HOST;
}
inline long long cpt_pw_resident_tri_base(long long n_nodes) { return (n_nodes * CPT_PW_RESIDENT_NODE_BYTES + 15) / 16 * 16; }
inline long long cpt_pw_resident_bytes(long long n_nodes, long long n_tris) {
    return cpt_pw_resident_tri_base(n_nodes) + n_tris * CPT_PW_RESIDENT_TRI_BYTES;
}

// What the counting instances of the extend kernels can report per ray besides the hit (all nullable; indexed by
// stream position): the order-free hash of the canonical ids of every node whose box was tested, and one bitmap row
// per ray with a bit per canonical id.  The parity hooks fill it in; the render path passes an empty one.
struct ExtendObserve {
    unsigned long long* vhash;
    uint8_t* bitmap;
    int bitmap_stride;
};

struct LaunchCfg {
    hipStream_t stream;
    int max_blocks;  // grid cap for grid-stride kernels
    int pw_blocks;   // number of persistent 1024-thread workgroups of k_extend_pw (one per CU)
    int pw_chunk;    // queue entries a wave takes per atomic
    int pw_refill;   // a wave refills when at least this many of its lanes are idle
    int pw_mode;     // 2: the scene fits the LDS record cache and the walk's stack its LDS entries; 1: the stack only; 0: neither (checked by the host)
    int pw_cull;     // 1: nearest-first walk with closest-hit culling (fast mode), 0: the reference's visit set
    int pw_f32;      // 1: inner nodes through the conservative binary32 test, leaves through the exact test of their own box
    int pw_full_go;  // ... or when at least this many lanes cannot expand a node because their parked-leaf list is full
    int pw_leaf_go;  // a wave runs its triangle phase when at least this many lanes have a parked leaf
    int pw_box_repeat; // node expansions per lane between two looks at the triangle phase: 2 or 3 (compiled instances)
    int pw_eye;        // 1: every live ray of the stream starts at sc.cam.eye and sc.nodes_eye / sc.tris_eye are current (exact variants only)
};

void launch_generate(const LaunchCfg& L, const DScene& sc, const DParams& P, const RayStream& out, const Radiance& col,
                     int batch_s0, int batch_n);
void launch_nodes32(hipStream_t st, const cpt_node* d_nodes, int n_nodes, void* d_nodes32);
void launch_eye_records(hipStream_t st, const DScene& sc, cpt_node* d_nodes_eye, cpt_tri* d_tris_eye);
void launch_dense_counts(const LaunchCfg& L, uint32_t* counts, uint32_t n, uint32_t cap);
void launch_extend(const LaunchCfg& L, const DScene& sc, const RayStream& in, const HitBuf& hit, const uint32_t* counter,
                   uint32_t cap, long long max_entries, unsigned long long* work, const ExtendObserve& obs);
void launch_extend_pw(const LaunchCfg& L, const DScene& sc, const RayStream& in, const HitBuf& hit, const uint32_t* counter,
                      uint32_t cap, uint32_t* cursor, uint32_t* spill, unsigned long long* work, const ExtendObserve& obs);
void launch_shade(const LaunchCfg& L, const DScene& sc, const DParams& P, const RayStream& in, const HitBuf& hit,
                  const RayStream& out, const Radiance& col, int batch_s0, int bounce, const uint32_t* counter_in,
                  uint32_t* counter_out, uint32_t cap, long long max_entries, unsigned long long* work);
void launch_resolve(const LaunchCfg& L, const DParams& P, const Radiance& col, int batch_n, double* accum);
void launch_megakernel(const LaunchCfg& L, const DScene& sc, const DParams& P, double* accum, unsigned long long* work);
void launch_add_values(hipStream_t st, double* dst, const double* src, long long n_values, int max_blocks);
void launch_tonemap(const LaunchCfg& L, const double* accum, long long n_values, int spp, uint8_t* rgb);
void launch_trace_rays(const LaunchCfg& L, const DScene& sc, int n, const double* o3, const double* d3, int* tri, double* t,
                       double* normal, unsigned long long* vhash, int* counts, uint8_t* bitmap, int bitmap_stride);
void launch_camera_rays(const LaunchCfg& L, const DScene& sc, int w, int h, double jx, double jy, double* o3, double* d3);
void launch_path_radiance(const LaunchCfg& L, const DScene& sc, const DParams& P, int n, const int* ijs, double* rad);
void launch_probe_math(const LaunchCfg& L, int n, const double* a, const double* b, double* s, double* q);
void launch_probe_rng(const LaunchCfg& L, const uint32_t c[4], const uint32_t k[2], uint32_t* out, double* out_u);

// Device-side scene preparation (bvh_device.hip).
void launch_build_shade(hipStream_t st, const cpt_tri* d_tris, const cpt_tri_normals* d_normals, const ShadeRec* d_by_material,
                        int n_tris, ShadeRec* d_out);
void launch_update_tris(hipStream_t st, const cpt_tri* d_fresh, int n_tris, cpt_tri* d_tris);
void launch_refit(hipStream_t st, const int* d_order, const int* level_start, int n_levels, const cpt_tri* d_tris, cpt_node* d_nodes);
size_t lbvh_work_bytes(int n_tris);
hipError_t launch_lbvh_build(hipStream_t st, cpt_tri* d_tris, int n_tris, cpt_node* d_nodes, void* d_work, int morton_bits,
                             int* h_height, bool keep_ranks);
