// The C ABI of the GPU library, driven from C++ with the library's HOST code built under AddressSanitizer and
// UndefinedBehaviorSanitizer (device code cannot be sanitized on this pool): create, upload, render, tonemap,
// progressive render, explicit rays, device-side build, refit, tree download, malformed input, destroy.
// Built by tests/sanitize/build_capi_driver.sh here, run on the GPU by tests/test_gpu_sanitized_capi.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>


#include "cpt_amd.h"
#include "cpt_host.h"

// AddressSanitizer's own interface (sanitizer/allocator_interface.h): empty the quarantine of freed blocks now.
extern "C" void __sanitizer_purge_allocator(void);

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != CPT_OK) { std::printf("FAILED %s: %d %s\n", #call, rc_, cpt_last_error(ctx)); return 1; } \
    } while (0)

static int progress(void* user, int32_t samples, const uint8_t*) {
    *(int*)user = samples;
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    cpt_ctx* ctx = nullptr;
    if (cpt_create(0, &ctx) != CPT_OK) { std::printf("cpt_create: %s\n", cpt_last_error(nullptr)); return 1; }
    for (int a = 1; a < argc; ++a) {
        cpth_scene* hs = nullptr;
        char err[256];
        if (cpth_load(argv[a], &hs, err, sizeof err) != 0) { std::printf("load %s: %s\n", argv[a], err); return 1; }
        cpt_scene_desc d;
        cpth_describe(hs, &d);
        CHECK(cpt_upload_scene(ctx, &d));
        const int w = 40, h = 24;
        cpt_render_params p;
        std::memset(&p, 0, sizeof p);
        p.width = w; p.height = h; p.sample_begin = 0; p.sample_end = 6; p.depth = 5; p.mode = CPT_MODE_WAVEFRONT; p.seed = 1984;
        p.focal_distance = 8.5; p.aperture = 0.6;
        std::vector<uint8_t> rgb((size_t)w * h * 3), rgb2(rgb.size());
        CHECK(cpt_render_image(ctx, &p, rgb.data()));
        std::vector<double> accum((size_t)w * h * 3, 0.0);
        int seen = 0, done = 0;
        CHECK(cpt_render_progressive(ctx, &p, 4, accum.data(), 0, rgb2.data(), progress, &seen, &done));
        if (done != 6 || seen != 6 || rgb != rgb2) { std::printf("progressive render differs\n"); return 1; }
        p.mode = CPT_MODE_MEGAKERNEL;
        CHECK(cpt_render_image(ctx, &p, rgb2.data()));
        if (rgb != rgb2) { std::printf("per-pixel mode differs\n"); return 1; }
        // explicit rays through both hooks, on the host tree and on the device-built tree
        const int n = 999;
        std::vector<double> o((size_t)n * 3), dir((size_t)n * 3), t(n), t2(n), nrm((size_t)n * 3);
        std::vector<int32_t> tri(n), tri2(n), counts((size_t)n * 4);
        std::vector<uint64_t> vh(n);
        for (int k = 0; k < n; ++k) {
            double ang = 0.01 * k;
            o[3 * k] = d.camera.eye[0]; o[3 * k + 1] = d.camera.eye[1]; o[3 * k + 2] = d.camera.eye[2];
            double x = 0.3 * std::cos(ang), y = 0.3 * std::sin(ang), z = -1.0, len = std::sqrt(x * x + y * y + z * z);
            dir[3 * k] = x / len; dir[3 * k + 1] = y / len; dir[3 * k + 2] = z / len;
        }
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) CHECK(cpt_build_scene(ctx, &d));
            CHECK(cpt_extend_rays(ctx, n, o.data(), dir.data(), tri.data(), t.data()));
            CHECK(cpt_trace_rays(ctx, n, o.data(), dir.data(), tri2.data(), t2.data(), nrm.data(), vh.data(), counts.data(), nullptr, 0));
            if (tri != tri2 || std::memcmp(t.data(), t2.data(), sizeof(double) * n) != 0) { std::printf("hooks disagree\n"); return 1; }
        }
        CHECK(cpt_refit_scene(ctx, &d));  // same vertices: the boxes must come out as they were
        CHECK(cpt_extend_rays(ctx, n, o.data(), dir.data(), tri2.data(), t2.data()));
        if (tri != tri2 || std::memcmp(t.data(), t2.data(), sizeof(double) * n) != 0) { std::printf("refit changed the hits\n"); return 1; }
        int32_t n_nodes = 0, height = 0;
        CHECK(cpt_download_tree(ctx, nullptr, 0, nullptr, 0, &n_nodes, &height));
        std::vector<cpt_node> nodes((size_t)(n_nodes > 0 ? n_nodes : 1));
        std::vector<uint32_t> rank((size_t)d.n_tris);
        CHECK(cpt_download_tree(ctx, nodes.data(), n_nodes, rank.data(), d.n_tris, &n_nodes, &height));
        if (cpt_download_tree(ctx, nodes.data(), n_nodes - 1, nullptr, 0, nullptr, nullptr) == CPT_OK && n_nodes > 0) { std::printf("short buffer accepted\n"); return 1; }
        // malformed input must be refused, not read out of bounds
        cpt_scene_desc bad = d;
        bad.n_nodes = d.n_nodes + 1;
        if (cpt_upload_scene(ctx, &bad) == CPT_OK) { std::printf("bad scene accepted\n"); return 1; }
        cpt_stats st;
        CHECK(cpt_get_stats(ctx, &st));
        std::printf("%s: %d triangles, device tree height %d, ok\n", argv[a], d.n_tris, height);
        cpth_free(hs);
    }
    cpt_destroy(ctx);
    std::printf("capi driver ok\n");
    std::fflush(stdout);
    // ROCm's AddressSanitizer runtime tracks hipMalloc'ed blocks in a device allocator of its own and, like host
    // blocks, parks them in its quarantine when they are freed.  The blocks cpt_destroy has just freed would still sit
    // there when the process exits; the HSA runtime's exit handler then frees host objects of its own, the quarantine
    // overflows, and recycling one of OUR long-freed device blocks after the HSA runtime has shut down trips an
    // internal check of the sanitizer (sanitizer_allocator_device.h: "!dev_runtime_unloaded_") -- a teardown-order
    // limitation inside the sanitizer runtime, not a memory error in anyone's code.  Draining the quarantine here,
    // while the runtime is still up, recycles those blocks at a time when that is legal.
    __sanitizer_purge_allocator();
    return 0;  // through the exit handlers and static destructors of everything linked in
}
