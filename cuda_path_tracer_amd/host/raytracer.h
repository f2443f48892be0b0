// Host-side render driver with the reference's RayTracer interface for the path-tracing path
// (reference: src/RayTracer.h:30-96): loadScene, traceSetup, traceImage, waitRender, getBuffer,
// aspectRatio, plus the two GPU entry points traceImageGPU and betterPathTracing.  Each of those
// two is now one call through the C ABI of libcpt_amd.so.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "cpt_amd.h"
#include "flatten.h"

namespace cpt {

struct PathTracingOptions {  // what the reference hard-codes in traceImageGPU (src/path_tracer.cu:509-514)
    bool use_dof = false;
    double focal_distance = 8.5;
    double aperture = 0.6;
    bool russian_roulette = false;
    bool work_queue = true;  // true: betterPathTracing (wavefront); false: traceImageGPU (one thread per pixel)
    unsigned long long seed = 1984;
    int device = 0;
    long long batch_paths = 0;      // > 0: path slots per batch (cpt_set_option "batch_paths"); 0: the library's default
    std::vector<int> devices;       // two or more GPU ordinals: samples are split across them (one host thread and one
                                    // context each), sums added on the host in list order; empty: `device` alone
    // long renders (no counterpart in the reference)
    bool fast = false;              // the fast mode: the host library's surface-area tree (same triangles, the reference
                                    // tree's tie-break ranks, so the same hits and the same image bit for bit) and, for
                                    // scenes of more than 2048 triangles, the nearest-first walk with culling
    bool device_bvh = false;        // build the BVH on the GPU (cpt_build_scene) instead of on the host; set before loadScene
    int progressive_spp = 0;        // > 0: refresh the image every this many samples and call on_progress
    std::string checkpoint;         // non-empty: resume from this file if it matches, and save the sums to it
    bool (*on_progress)(void* user, int samples, const unsigned char* rgb, int w, int h) = nullptr;  // true = stop
    void* progress_user = nullptr;
};

class RayTracer {
public:
    RayTracer();
    ~RayTracer();

    bool loadScene(const char* fn);
    bool sceneLoaded() const { return scene_ != nullptr; }
    double aspectRatio() const;

    void traceSetup(int w, int h, bool useGPU, int pathTracingSamples, int depth);
    void traceImage(int w, int h);
    void traceImageGPU(int w, int h);       // reference: src/path_tracer.cu:507-551
    void betterPathTracing(int w, int h);   // reference: src/improved_path_tracer.cu:277-347
    void waitRender() {}                    // the GPU path is synchronous (reference: src/RayTracer.cpp:735-752)
    void getBuffer(unsigned char*& buf, int& w, int& h);

    PathTracingOptions options;
    const std::string& lastError() const { return error_; }
    int samplesInImage() const { return samples_in_image_; }  // differs from the request after a resume or an early stop
    bool stats(cpt_stats& out) const { return ctx_ && cpt_get_stats(ctx_, &out) == CPT_OK; }
    bool failed() const { return !error_.empty(); }

private:
    void renderGPU(int w, int h, int mode);
    void renderMultiGPU(int w, int h, const cpt_render_params& p);
    cpt_scene_desc sceneDesc();            // the scene over the tree the options ask for
    bool applyWalkOptions(cpt_ctx* c);     // the extend variant that goes with it
    std::unique_ptr<FlatScene> scene_;
    cpt_ctx* ctx_ = nullptr;
    std::vector<cpt_ctx*> more_ctx_;  // contexts of options.devices[1..]
    bool uploaded_ = false;
    std::vector<unsigned char> buffer_;
    int buffer_width_ = 0, buffer_height_ = 0;
    bool useGPU_ = false;
    int samples_per_pixel_ = 32, depth_ = 0, samples_in_image_ = 0;
    std::string error_;
};

}  // namespace cpt
