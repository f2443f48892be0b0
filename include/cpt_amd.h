/* cpt_amd.h -- C ABI of the MI355X (gfx950) path-tracing library, libcpt_amd.so.
 *
 * This is the drop-in boundary for the reference's GPU render path.  The reference has no
 * FFI: the seam is the member call RayTracer::traceImage -> traceImageGPU(w, h)
 * (reference: src/RayTracer.cpp:698-707, src/path_tracer.cu:507-551) and the dormant
 * RayTracer::betterPathTracing(w, h) (src/improved_path_tracer.cu:277-347).  Everything those
 * two functions do after "the scene and its BVH exist on the host" is behind these entry
 * points: plain pointers and sizes, int status codes, no exceptions, no exit(), no globals.
 *
 * Conventions
 *   - every call returns CPT_OK (0) or a negative CPT_E* code; cpt_last_error() gives text.
 *   - pointers named h_* are host memory, d_* are device memory on the context's GPU.
 *   - images are RGB, row j = 0 is the BOTTOM row (camera up), index (i + j*w)*3, exactly the
 *     reference's RayTracer::buffer (src/RayTracer.h:82, src/path_tracer.cu:440-443).
 *   - all arithmetic is IEEE binary64 without FMA contraction.
 *   - a context is bound to one GPU and must be used from one host thread at a time.
 */
#ifndef CPT_AMD_H
#define CPT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPT_OK 0
#define CPT_EINVAL (-1)   /* bad argument / scene failed validation            */
#define CPT_EHIP (-2)     /* a HIP runtime call failed                         */
#define CPT_ENOSCENE (-3) /* render/trace before cpt_upload_scene              */
#define CPT_ELIMIT (-4)   /* a compiled-in limit would be exceeded             */

/* ---- flat scene records (host fills them, the library copies them to HBM) ------------- */

/* One expanded internal BVH node: both child boxes, both child references.  128 bytes,
 * 128-byte aligned = one cache line pair per traversal step.
 * Replaces GPU::BVHNode + GPU::BoundingBox (reference: src/gpu/GPUBVH.cuh:17-27,
 * src/gpu/GPUbb.cuh:10-57), which are 80-byte managed allocations reached by pointer.
 * Record 0 is the root and a child's record always comes after its parent's; the host library
 * stores them level by level, so a prefix of the array is the top of the tree. */
typedef struct cpt_node {
    double lbox[6];   /* left child  box: min x,y,z then max x,y,z */
    double rbox[6];   /* right child box                            */
    int32_t lref;     /* >= 0: record index of the left child; < 0: leaf, triangle index is ~lref */
    int32_t rref;
    int32_t lid;      /* canonical id (preorder over ALL nodes, root = 0) of the left child  */
    int32_t rid;      /* ... of the right child; used only for visit-set reporting            */
    int32_t self_id;  /* canonical id of this node                                            */
    int32_t pad_[3];
} cpt_node;

/* One triangle as the intersection test needs it.  128 bytes.
 * Replaces GPU::TrimeshFace (reference: src/gpu/GPUTrimesh.cuh:17-57). */
typedef struct cpt_tri {
    double a[3], b[3], c[3]; /* vertices                                      */
    double n[3];             /* unit face normal, normalize(cross(b-a, c-a))  */
    int32_t material;        /* index into the material table                 */
    uint32_t flags;          /* CPT_TRI_*                                     */
    uint32_t test_rank;      /* position of this triangle's leaf in the reference's traversal order
                                (2 * expansion order of the parent + side), used to break ties in t.
                                Filled in by cpt_upload_scene; callers may leave it 0.             */
    int32_t pad_[5];
} cpt_tri;
#define CPT_TRI_RECUR 1u         /* material is reflective or transmissive: no back-face cull */
#define CPT_TRI_VERTEX_NORMALS 2u /* interpolate cpt_tri_normals instead of the face normal   */

/* Per-vertex normals of one triangle (read only by shading, only for the winning hit). */
typedef struct cpt_tri_normals {
    double na[3], nb[3], nc[3];
} cpt_tri_normals;

/* Replaces GPU::Material (reference: src/gpu/GPUMaterial.cuh:10-58); one per mesh. */
typedef struct cpt_material {
    double ke[3], kd[3], ks[3], kr[3], kt[3];
    double index;   /* 0.299 r + 0.587 g + 0.114 b of the index triple (GPUMaterial.cuh:21) */
    uint32_t flags; /* CPT_MAT_* */
    uint32_t pad_;
} cpt_material;
#define CPT_MAT_DIFF 1u
#define CPT_MAT_SPEC 2u
#define CPT_MAT_TRANS 4u
#define CPT_MAT_RECUR 8u

/* Replaces GPU::Camera (reference: src/gpu/GPUCamera.cu:11-32). */
typedef struct cpt_camera {
    double eye[3], look[3], u[3], v[3];
} cpt_camera;

typedef struct cpt_scene_desc {
    int32_t n_nodes;                 /* internal-node records; 0 for a one-triangle scene */
    int32_t n_tris;
    int32_t n_materials;
    int32_t n_tree_nodes;            /* canonical node count (2*n_tris - 1)               */
    int32_t tree_height;             /* levels, root = 1                                   */
    uint32_t flags;                  /* CPT_SCENE_*                                        */
    const cpt_node* nodes;           /* host arrays, caller-owned, copied by the call      */
    const cpt_tri* tris;
    const cpt_tri_normals* tri_normals; /* n_tris entries                                  */
    const cpt_material* materials;
    cpt_camera camera;
} cpt_scene_desc;

/* cpt_upload_scene and cpt_build_scene keep the test_rank of the triangle records as given instead of deriving it from
 * the tree: for a tree other than the one whose tie-breaks are wanted (the host library's fast tree, or a tree built on
 * the device, carrying the reference tree's ranks). */
#define CPT_SCENE_KEEP_RANKS 1u

/* ---- render parameters ---------------------------------------------------------------- */

#define CPT_MODE_WAVEFRONT 0 /* generate -> extend -> shade -> compact queues (betterPathTracing) */
#define CPT_MODE_MEGAKERNEL 1 /* one thread per pixel, all samples and bounces (traceImageGPU)    */

typedef struct cpt_render_params {
    int32_t width, height;
    int32_t sample_begin, sample_end; /* this call renders sample indices [begin, end) of every pixel */
    int32_t depth;                    /* -r: a path has at most depth+1 segments                      */
    int32_t mode;                     /* CPT_MODE_*                                                   */
    uint64_t seed;                    /* reference seed is 1984 (src/path_tracer.cu:412)              */
    int32_t dof;                      /* 0: pin-hole; 1: thin lens (src/path_tracer.cu:446-492)       */
    int32_t russian_roulette;         /* 0 (reference behaviour) or 1                                 */
    double focal_distance, aperture;  /* reference constants: 8.5, 0.6 (src/path_tracer.cu:512-514)   */
    int32_t tile_begin, tile_count;   /* render only the 8x8-pixel tiles [tile_begin, tile_begin + tile_count) of the
                                         image, tiles numbered row by row, (width + 7) / 8 per row; tile_count 0 =
                                         the whole image.  The accumulation buffer is always the whole image's; pixels
                                         outside the window are not touched.  A pixel's paths do not depend on the
                                         window, so disjoint windows add up to the full render bit for bit: the way to
                                         shard one render over GPUs by area when there are fewer samples than GPUs. */
} cpt_render_params;

typedef struct cpt_stats {
    uint64_t paths, segments, hits, node_steps, tri_tests; /* counted only when counting is on */
    uint64_t launches_generate, launches_extend, launches_shade, launches_mega;
    double ms_generate, ms_extend, ms_shade, ms_mega, ms_total; /* HIP-event time on the render stream */
    uint64_t iterations;
    uint64_t exact_node_steps;  /* node steps whose boxes were re-tested with the reference's divisions */
    uint64_t box_passes;        /* wave-level passes over the box code of the persistent extend kernel (64 lane slots each) */
    uint64_t leaf_passes;       /* wave-level passes over its triangle code */
} cpt_stats;

typedef struct cpt_ctx cpt_ctx;

/* ---- lifecycle -------------------------------------------------------------------------- */

/* Create a context on GPU `device`.  Replaces the implicit CUDA context of traceImageGPU. */
int cpt_create(int device, cpt_ctx** out);
void cpt_destroy(cpt_ctx* ctx);
/* Text of the last error on this context (ctx may be NULL: error of the last failed cpt_create). */
const char* cpt_last_error(const cpt_ctx* ctx);
/* Launch on this hipStream_t (passed as void*) instead of the context's own stream.  As everywhere
 * in HIP, NULL is the legacy default stream; CPT_OWN_STREAM goes back to the context's own stream,
 * which is an ordinary (blocking) stream: it is ordered after work queued earlier on the legacy
 * default stream, so buffers prepared there can be handed to cpt_render without a fence.
 * Work the context queued earlier (cpt_render, cpt_render_noalloc, cpt_tonemap) is ordered before what it queues
 * on the new stream: the new stream is made to wait for an event those calls leave behind.  The previous stream's
 * handle is not used by this call, so it may already have been destroyed; the context moves to the new stream even
 * when the call returns an error (the wait could not be queued).  Not for use while either stream is being captured. */
#define CPT_OWN_STREAM ((void*)-1)
int cpt_set_stream(cpt_ctx* ctx, void* hip_stream);
/* Tuning knobs, by name: "batch_paths" (path slots of one batch of the wavefront pipeline,
 * default 2^25 = up to 7 GB of ray streams, hit records and radiance per batch in flight; 2^27 renders large scenes ~6 % faster,
 * but a short-lived process that allocates tens of GB makes the next one wait for the driver to reclaim them; launches of tens of millions of rays
 * keep the persistent kernel's tail short), "streams" (batches in flight at once, each on its own HIP stream, so that one
 * batch's intersection kernel overlaps another's shading kernel; default 2, at most 4; the result
 * does not depend on it), "max_blocks" (grid cap of the grid-stride kernels, default 8 per CU; the shade kernel takes five times that),
 * "extend_variant" (1: persistent waves with an LDS record cache, default; 0: one ray per thread;
 * 2: as 1 but walking the nearer child first and skipping boxes entered beyond the closest hit so
 * far -- a fast mode that returns the same hits without reproducing the reference's visit set;
 * 3: as 1 but inner nodes are tested in binary32 against outward-rounded boxes with margins that make the test
 * conservative, and every candidate leaf is then put to the exact binary64 test of its own box before its triangle
 * is tested: because a box that contains a leaf's box can never fail the reference's test when the leaf's box
 * passes it, the closest hits are exactly those of variant 1, from a slightly larger visit set and a third less
 * arithmetic per node; 4: as 3 with the nearest-first walk and culling of 2.  2, 3 and 4 rest on two properties of the
 * tree: every child box lies inside its parent's box, and every leaf box is exactly the bounds of its triangle's
 * three vertices.  Trees built or refitted by the library have them; cpt_upload_scene checks the caller's tree, and on
 * a tree without them (padded or tightened leaf boxes, boxes that are not nested -- which the reference's walk, and
 * so variants 0 and 1, take as they come) the context runs 2, 3 and 4 as 1),
 * "pw_blocks" / "pw_chunk" (workgroups of the persistent extend kernel / rays a wave claims per
 * atomic, 0 = chosen from the scene size),
 * "pw_box_repeat" (node expansions per lane between two looks at the triangle phase: 3 or 2; 0 = 3, or 2 for a tree
 * uploaded with CPT_SCENE_KEEP_RANKS, whose walks are shorter),
 * "pw_eye" (1, default: the first rays of a pin-hole render, which all start at the camera's eye, are intersected by an
 * instance of that kernel whose node and triangle records already contain the terms that depend on the origin alone --
 * plane - eye, reference src/gpu/GPUbb.cuh:33-34, and -dot(eye - a, n), src/gpu/GPUTrimesh.cuh:62 -- worked out once per
 * camera with the same operations; same hits, same visit sets, a fifth less arithmetic per node step of those rays, and
 * one more copy of the node and triangle records on the device; 0: every bounce through the general instance),
 * "pw_refill" / "pw_leaf_go" / "pw_full_go" (a wave of that kernel refills when this many lanes are
 * idle / runs its triangle pass when this many lanes have a parked leaf, 0 = 32 for the reference walk and 16 for the
 * culling walks / or when this many lanes cannot go on because their parked-leaf list is full). */
int cpt_set_option(cpt_ctx* ctx, const char* key, long long value);
/* Turn per-launch HIP-event timing and device-side work counters on or off (both default off). */
int cpt_set_instrumentation(cpt_ctx* ctx, int time_kernels, int count_work);

/* Copy a flattened scene to the GPU.  Replaces `new GPU::Scene(scene)` + `copyBVH(&bvhTree)`
 * (reference: src/path_tracer.cu:519-520, src/gpu/GPUScene.cuh:25-57, src/gpu/GPUBVH.cu:9-31). */
int cpt_upload_scene(cpt_ctx* ctx, const cpt_scene_desc* scene);

/* Copy the triangles, normals, materials and camera of `scene` to the GPU and build the BVH there
 * (`scene->nodes`, `n_nodes`, `n_tree_nodes` and `tree_height` are ignored).  Replaces the host build
 * `BVHTree::build` (reference: src/scene/bvh.cpp:101-220) together with the upload above, for callers
 * that do not need the reference's tree: the result is a Morton-order tree with exact boxes, so renders
 * are statistically the same image but visit sets, step counts and the winner among equal-t hits are
 * this tree's own -- unless scene->flags has CPT_SCENE_KEEP_RANKS and the triangle records carry the tie-break ranks of
 * the reference's tree (the host library fills them in): then every ray's closest hit, ties included, is the one the
 * reference's tree gives, and the image is the same bit for bit.  A tree too tall for the traversal stacks is rebuilt
 * with coarser Morton codes; CPT_ELIMIT if even that fails.  Synchronous. */
int cpt_build_scene(cpt_ctx* ctx, const cpt_scene_desc* scene);

/* New vertices (and normals, materials, camera) for the scene on the GPU, same triangles in the same order, same
 * tree: every box of the tree in place is recomputed on the device, bottom-up, from the new triangle bounds;
 * topology, node numbering and tie-break ranks stay.  For geometry that moves a little between renders, where a
 * rebuild is not worth it; the tree stays correct for any motion, only less tight.  `scene->nodes` is ignored.
 * No reference counterpart (the reference rebuilds and re-uploads everything per process).  Synchronous. */
int cpt_refit_scene(cpt_ctx* ctx, const cpt_scene_desc* scene);

/* Read back the node records (and, if h_test_rank is not NULL, every triangle's tie-break rank) of the
 * scene on the GPU, however it got there.  h_nodes may be NULL to query *n_nodes / *height only.
 * No reference counterpart: lets a host inspect, cache or cross-check a device-built tree. */
int cpt_download_tree(cpt_ctx* ctx, cpt_node* h_nodes, int32_t node_capacity, uint32_t* h_test_rank, int32_t tri_capacity,
                      int32_t* n_nodes, int32_t* height);

/* ---- rendering -------------------------------------------------------------------------- */

/* Add the radiance of samples [sample_begin, sample_end) of every pixel to d_accum
 * (device memory, width*height*3 doubles, layout (i + j*w)*3, caller-owned, e.g. a torch tensor).
 * Asynchronous on the context's stream unless instrumentation is on.
 * Replaces the kernel launches of traceImageGPU / betterPathTracing
 * (reference: src/path_tracer.cu:543-547, src/improved_path_tracer.cu:296-333). */
int cpt_render(cpt_ctx* ctx, const cpt_render_params* params, double* d_accum);

/* cpt_render for callers that cannot allocate device memory (code under HIP graph capture, serving loops with a
 * fixed memory budget, harnesses that forbid hipMalloc): the same pipeline, the same kernels and bit for bit the
 * same sums as cpt_render (wavefront mode only), but all working memory -- ray streams, hit records, per-path
 * radiance, queue counters, stack-spill columns -- comes from one CPT_NOALLOC_SCRATCH_BYTES array that lives in the
 * library's device code, so the call makes no hipMalloc / hipFree and creates no events or streams.  It has no
 * counterpart in the reference (whose render path allocates per object, src/gpu/GPUManaged.cuh:19-24).
 * Limits: one sample of the tile window must fit, i.e.
 *   2 * 2 KB * (depth + 2)  +  4 KB * CUs * max(0, tree height - 14)  +  209 B * (pixels of the window, in whole 8x8 tiles)
 * must not exceed CPT_NOALLOC_SCRATCH_BYTES (a 512x512 image at any accepted tree height does; a 1024x1024 image is
 * rendered as four tile windows); otherwise CPT_ELIMIT.  Samples are rendered in as large batches as fit, one batch
 * at a time on the context's stream; instrumentation settings are ignored.  Every context on a GPU shares that
 * GPU's array: two cpt_render_noalloc calls that overlap in time on one GPU (two contexts, or two streams) race;
 * calls on one stream, or ordered by events, are safe, and the array needs no clearing between calls (every call
 * zeroes the counters it uses).  Asynchronous on the context's stream. */
#define CPT_NOALLOC_SCRATCH_BYTES (128ull << 20)
int cpt_render_noalloc(cpt_ctx* ctx, const cpt_render_params* params, double* d_accum);

/* mean over total_spp, clamp to [0,1], (int)(255*c) -> d_rgb (device, width*height*3 bytes).
 * Replaces the tail of pathTraceKernel (reference: src/path_tracer.cu:436-443). */
int cpt_tonemap(cpt_ctx* ctx, const double* d_accum, int32_t width, int32_t height, int32_t total_spp,
                uint8_t* d_rgb);

/* The whole of traceImageGPU in one call: allocate, render all samples, tonemap, copy the RGB8
 * image into h_rgb (host, width*height*3 bytes), free.  Blocking.
 * Replaces RayTracer::traceImageGPU(w, h) (reference: src/path_tracer.cu:507-551). */
int cpt_render_image(cpt_ctx* ctx, const cpt_render_params* params, uint8_t* h_rgb);

/* cpt_render_image in chunks, for long renders: after every `chunk_spp` samples (and at the end) the
 * image so far is tonemapped, copied into h_rgb and `progress` (nullable) is called with the number of
 * samples the image now holds; a non-zero return from it stops the render (the call then returns
 * CPT_OK with what was accumulated).  h_accum (nullable; host, width*height*3 doubles) makes the
 * render resumable: on entry it holds the radiance sums of the `accum_samples` samples
 * [params->sample_begin - accum_samples, params->sample_begin) of an earlier call (ignored when
 * accum_samples is 0), on return the sums including this call's samples.  Because a sample's paths
 * depend only on (pixel, sample index, seed), a render resumed this way is bit-identical to one
 * made in a single call.  *samples_done (nullable) receives the samples this call added.  Blocking.
 * No counterpart in the reference, whose render is one blocking kernel launch
 * (src/path_tracer.cu:543-550). */
typedef int (*cpt_progress_fn)(void* user, int32_t samples_in_image, const uint8_t* h_rgb);
int cpt_render_progressive(cpt_ctx* ctx, const cpt_render_params* params, int32_t chunk_spp, double* h_accum,
                           int32_t accum_samples, uint8_t* h_rgb, cpt_progress_fn progress, void* user,
                           int32_t* samples_done);

/* One whole-image render on several GPUs of this process (cpt_render_image split by samples).  contexts[0..n) are distinct
 * contexts, normally one per GPU, each with the same scene in place.  Context k renders a contiguous share of the sample
 * indices [sample_begin, sample_end) of every pixel -- the first (count % n) contexts one sample more -- from a host thread
 * of its own into radiance sums on its own GPU.  The sums are then copied device to device to contexts[0]'s GPU
 * (hipMemcpyPeerAsync: over xGMI between the GPUs of a node) and added there by a kernel, one peer after the other in list
 * order, and the total is tonemapped there into h_rgb (host, width*height*3 bytes, bottom row first).  A sample's paths
 * depend only on (pixel, sample index, seed), so the sample set is the single-GPU one; the sums are those of a
 * single-GPU render except for the association of the n - 1 final additions per value.  *samples_done (nullable)
 * receives the samples in the image.  Blocking; errors of any context are reported through contexts[0].
 * No counterpart in the reference, which drives one GPU (src/path_tracer.cu:507-551). */
int cpt_render_image_multi(cpt_ctx* const* contexts, int32_t n_contexts, const cpt_render_params* params, uint8_t* h_rgb,
                           int32_t* samples_done);

/* Wait for everything queued on the context's stream. */
int cpt_synchronize(cpt_ctx* ctx);
int cpt_get_stats(cpt_ctx* ctx, cpt_stats* out);
int cpt_reset_stats(cpt_ctx* ctx);

/* ---- parity hooks ----------------------------------------------------------------------- */

/* Closest hits of n explicit rays through the wavefront pipeline's own extend kernel (the variant
 * selected by the "extend_variant" option), i.e. the kernel cpt_render launches once per bounce, fed
 * with the caller's rays instead of a generated stream.  Host arrays: h_origin/h_dir n*3 doubles in,
 * h_tri n (triangle index or -1) and h_t n out.  With work counting on, cpt_get_stats afterwards
 * reports the node steps, triangle tests and exact-box fallbacks of the call.
 * Observes the extend half of pathTrace (reference: src/improved_path_tracer.cu:60-75), which calls
 * GPU::BVH::traverseIterative (src/gpu/GPUBVH.cu:38-99). */
int cpt_extend_rays(cpt_ctx* ctx, int32_t n, const double* h_origin, const double* h_dir, int32_t* h_tri, double* h_t);

/* cpt_extend_rays through the counting instance of that same kernel, which also reports, per ray, what its walk
 * touched: h_visit_hash n (nullable; sum over the nodes whose box was tested of mix64(canonical id + 1), the same
 * order-free hash cpt_trace_rays reports) and h_bitmap (nullable) n rows of bitmap_stride bytes with bit k set iff
 * node k's box was tested.  For extend variants 0 and 1 both equal the reference walk's; variants 3 and 4 test a
 * superset of inner boxes, 2 and 4 skip culled subtrees.
 * Observes GPU::BVH::traverseIterative (reference: src/gpu/GPUBVH.cu:38-99) as the render path runs it. */
int cpt_extend_rays_observed(cpt_ctx* ctx, int32_t n, const double* h_origin, const double* h_dir, int32_t* h_tri, double* h_t,
                             uint64_t* h_visit_hash, uint8_t* h_bitmap, int32_t bitmap_stride);

/* Intersect n explicit rays with the scene using exactly the render path's traversal.
 * Host arrays: h_origin/h_dir n*3 doubles.  Outputs (host): h_tri n (triangle index or -1),
 * h_t n, h_normal n*3 (shading normal of the hit, nullable), h_visit_hash n (sum over tested
 * nodes of mix64(canonical id + 1), nullable), h_counts n*4 (node steps, box tests, triangle
 * tests, deepest stack; nullable), h_bitmap (nullable) n rows of bitmap_stride bytes with bit k
 * set iff node k's box was tested.
 * Observes GPU::BVH::traverseIterative (reference: src/gpu/GPUBVH.cu:38-99). */
int cpt_trace_rays(cpt_ctx* ctx, int32_t n, const double* h_origin, const double* h_dir, int32_t* h_tri,
                   double* h_t, double* h_normal, uint64_t* h_visit_hash, int32_t* h_counts,
                   uint8_t* h_bitmap, int32_t bitmap_stride);

/* Camera rays through ((i + jx)/w, (j + jy)/h) for every pixel (index i + j*w), generated on the
 * GPU by the render path's camera code; h_origin/h_dir: width*height*3 doubles.
 * Observes GPU::Camera::rayThrough (reference: src/gpu/GPUCamera.cu:22-32). */
int cpt_camera_rays(cpt_ctx* ctx, int32_t width, int32_t height, double jx, double jy, double* h_origin,
                    double* h_dir);

/* Radiance of individual paths (i, j, sample) -> h_radiance n*3 doubles, computed by the same
 * device functions as cpt_render.  Observes pathTraceRayGPU (reference: src/path_tracer.cu:255-400). */
int cpt_path_radiance(cpt_ctx* ctx, const cpt_render_params* params, int32_t n, const int32_t* h_ijs,
                      double* h_radiance);

/* Device-side checks of the arithmetic the parity claims rest on: n doubles in, per element
 * sqrt(x) and 1/x and a/b as the device computes them (h_b nullable -> b = 3). */
int cpt_probe_math(cpt_ctx* ctx, int32_t n, const double* h_a, const double* h_b, double* h_sqrt,
                   double* h_div);

/* Philox-4x32-10 block and its (0,1] double, as the device computes them (known-answer hook). */
int cpt_probe_rng(cpt_ctx* ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], double out_u01[2]);

#ifdef __cplusplus
}
#endif
#endif /* CPT_AMD_H */
