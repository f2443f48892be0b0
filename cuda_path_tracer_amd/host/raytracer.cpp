#include "raytracer.h"

#include <algorithm>
#include <iostream>

namespace cpt {

RayTracer::RayTracer() = default;

RayTracer::~RayTracer() {
    if (ctx_) cpt_destroy(ctx_);
}

bool RayTracer::loadScene(const char* fn) {
    error_.clear();
    try {
        scene_.reset(new FlatScene(load_flat_scene(fn)));
    } catch (const std::exception& e) {
        scene_.reset();
        error_ = e.what();
        std::cerr << "Parser: " << error_ << std::endl;
        return false;
    }
    for (const std::string& w : scene_->warnings) std::cerr << "warning: " << w << std::endl;
    uploaded_ = false;
    return true;
}

double RayTracer::aspectRatio() const { return sceneLoaded() ? scene_->aspect : 1; }

// reference: src/RayTracer.cpp:598-645.  The BVH is already built by loadScene.
void RayTracer::traceSetup(int w, int h, bool useGPU, int pathTracingSamples, int depth) {
    size_t n = (size_t)w * h * 3;
    if (n != buffer_.size()) buffer_.resize(n);
    buffer_width_ = w;
    buffer_height_ = h;
    std::fill(buffer_.begin(), buffer_.end(), 0);
    useGPU_ = useGPU;
    if (useGPU_) std::cout << "GPU Enabled\n";
    samples_per_pixel_ = pathTracingSamples;
    depth_ = depth;
    std::cout << "Current samples per pixel: " << samples_per_pixel_ << std::endl;
}

// reference: src/RayTracer.cpp:698-707
void RayTracer::traceImage(int w, int h) {
    traceSetup(w, h, useGPU_, samples_per_pixel_, depth_);
    if (!useGPU_) {
        error_ = "the CPU renderer is not part of this build: run with -g (the GPU path is the product; there is no CPU fallback)";
        return;
    }
    if (options.work_queue) betterPathTracing(w, h);
    else traceImageGPU(w, h);
}

void RayTracer::traceImageGPU(int w, int h) { renderGPU(w, h, CPT_MODE_MEGAKERNEL); }
void RayTracer::betterPathTracing(int w, int h) { renderGPU(w, h, CPT_MODE_WAVEFRONT); }

void RayTracer::renderGPU(int w, int h, int mode) {
    error_.clear();
    if (!sceneLoaded()) { error_ = "no scene loaded"; return; }
    if (!ctx_) {
        if (cpt_create(options.device, &ctx_) != CPT_OK) { error_ = cpt_last_error(nullptr); ctx_ = nullptr; return; }
    }
    if (!uploaded_) {
        cpt_scene_desc d = scene_->describe();
        if (cpt_upload_scene(ctx_, &d) != CPT_OK) { error_ = cpt_last_error(ctx_); return; }
        uploaded_ = true;
    }
    if ((size_t)w * h * 3 != buffer_.size()) traceSetup(w, h, true, samples_per_pixel_, depth_);
    cpt_render_params p{};
    p.width = w; p.height = h;
    p.sample_begin = 0; p.sample_end = samples_per_pixel_;
    p.depth = depth_;
    p.mode = mode;
    p.seed = options.seed;
    p.dof = options.use_dof ? 1 : 0;
    p.russian_roulette = options.russian_roulette ? 1 : 0;
    p.focal_distance = options.focal_distance;
    p.aperture = options.aperture;
    if (cpt_render_image(ctx_, &p, buffer_.data()) != CPT_OK) error_ = cpt_last_error(ctx_);
}

void RayTracer::getBuffer(unsigned char*& buf, int& w, int& h) {
    buf = buffer_.data();
    w = buffer_width_;
    h = buffer_height_;
}

}  // namespace cpt
