/* cpt_host.h -- C ABI of the host scene library, libcpt_host.so (no GPU code).
 *
 * Everything the reference does on the CPU before it hands a scene to its GPU path: read the
 * .ray file (reference: RayTracer::loadScene, src/RayTracer.cpp:554-596), build the BVH
 * (BVHTree::build, src/scene/bvh.cpp:101-126, called from RayTracer::traceSetup,
 * src/RayTracer.cpp:622), and lay the result out as the flat records of cpt_amd.h; plus the image
 * writer used after the render (writeImage, src/fileio/images.cc:59-68).
 */
#ifndef CPT_HOST_H
#define CPT_HOST_H

#include <stdint.h>

#include "cpt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cpth_scene cpth_scene;

/* Parse + build + flatten.  Returns 0 and *out, or -1 with a message in err. */
int cpth_load(const char* ray_path, cpth_scene** out, char* err, int errlen);
/* Same from scene text already in memory. */
int cpth_load_text(const char* ray_text, cpth_scene** out, char* err, int errlen);
void cpth_free(cpth_scene* s);

/* Fill *out with pointers into the scene's own arrays (valid until cpth_free). */
int cpth_describe(const cpth_scene* s, cpt_scene_desc* out);
/* The same scene with the fast mode's tree instead of the reference's: a full-sweep surface-area build over the same
 * triangles (about a third fewer box tests per ray on the bundled scenes, because the reference's walk tests every box a
 * ray pierces and that is what the surface-area cost counts), the triangle records carrying the reference tree's
 * tie-break ranks (out->flags has CPT_SCENE_KEEP_RANKS), so that every ray's closest hit -- triangle and distance -- is the
 * one the reference's tree gives.  Visit sets and step counts are this tree's own.  Built on the first call.  No
 * counterpart in the reference (BVHTree::build, src/scene/bvh.cpp:101-220, is its only build). */
int cpth_describe_fast(cpth_scene* s, cpt_scene_desc* out);
/* cpth_tree_nodes for that tree (n_tree_nodes rows as well). */
int cpth_fast_tree_nodes(cpth_scene* s, double* boxes, int32_t* topo);
/* width / height of the camera (reference: RayTracer::aspectRatio, src/RayTracer.cpp:549-552). */
double cpth_aspect(const cpth_scene* s);
/* Number of warnings collected while reading, and the k-th one. */
int cpth_warning_count(const cpth_scene* s);
const char* cpth_warning(const cpth_scene* s, int k);

/* The canonical tree (preorder): boxes n*6 doubles (min, max), topo n*3 ints (left id, right id,
 * triangle) with -1 for "none".  For inspection and parity tests. */
int cpth_tree_nodes(const cpth_scene* s, double* boxes, int32_t* topo);

/* Write an RGB8 buffer whose row 0 is the bottom of the image to a .png or .bmp file. */
int cpth_write_image(const char* path, int32_t width, int32_t height, const uint8_t* rgb, char* err, int errlen);

#ifdef __cplusplus
}
#endif
#endif /* CPT_HOST_H */
