#include "raytracer.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>

namespace cpt {

RayTracer::RayTracer() = default;

RayTracer::~RayTracer() {
    if (ctx_) cpt_destroy(ctx_);
}

bool RayTracer::loadScene(const char* fn) {
    error_.clear();
    try {
        scene_.reset(new FlatScene(load_flat_scene(fn, !options.device_bvh)));
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
        const bool on_device = scene_->nodes.empty() && scene_->tris.size() > 1;  // loaded without a host tree
        if ((on_device ? cpt_build_scene(ctx_, &d) : cpt_upload_scene(ctx_, &d)) != CPT_OK) { error_ = cpt_last_error(ctx_); return; }
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
    samples_in_image_ = 0;
    if (options.progressive_spp <= 0 && options.checkpoint.empty()) {
        if (cpt_render_image(ctx_, &p, buffer_.data()) != CPT_OK) error_ = cpt_last_error(ctx_);
        else samples_in_image_ = samples_per_pixel_;
        return;
    }
    // Long render: chunks, optional checkpoint.  The checkpoint holds the fp64 radiance sums of samples
    // [0, have) and the parameters they depend on; a sample's paths depend only on (pixel, sample
    // index, seed), so resuming adds exactly the samples a single run would have added.
    struct Header { char magic[8]; int32_t w, h, depth, mode, dof, rr, samples, pad; uint64_t seed; double focal, aperture; };
    Header want{};
    memcpy(want.magic, "CPTCKPT1", 8);
    want.w = w; want.h = h; want.depth = depth_; want.mode = mode; want.dof = p.dof; want.rr = p.russian_roulette;
    want.seed = p.seed; want.focal = p.focal_distance; want.aperture = p.aperture;
    std::vector<double> accum((size_t)w * h * 3, 0.0);
    int have = 0;
    if (!options.checkpoint.empty()) {
        if (FILE* f = fopen(options.checkpoint.c_str(), "rb")) {
            Header got{};
            bool ok = fread(&got, sizeof got, 1, f) == 1;
            const int samples = got.samples;
            got.samples = 0;
            ok = ok && memcmp(&got, &want, sizeof got) == 0 && samples > 0 && samples <= samples_per_pixel_;
            ok = ok && fread(accum.data(), sizeof(double), accum.size(), f) == accum.size();
            fclose(f);
            if (ok) have = samples;
            else std::fill(accum.begin(), accum.end(), 0.0);
            std::cout << (ok ? "resuming from " : "ignoring ") << options.checkpoint;
            if (ok) std::cout << " at " << have << " samples";
            std::cout << std::endl;
        }
    }
    samples_in_image_ = have;
    if (have < samples_per_pixel_) {
        p.sample_begin = have;
        struct Ctx { RayTracer* rt; int w, h; } cb{this, w, h};
        auto tramp = [](void* u, int32_t samples, const uint8_t* rgb) -> int {
            Ctx* x = (Ctx*)u;
            x->rt->samples_in_image_ = samples;
            const PathTracingOptions& o = x->rt->options;
            return (o.on_progress && o.on_progress(o.progress_user, samples, rgb, x->w, x->h)) ? 1 : 0;
        };
        int chunk = options.progressive_spp > 0 ? options.progressive_spp : samples_per_pixel_ - have;
        int32_t done = 0;
        if (cpt_render_progressive(ctx_, &p, chunk, accum.data(), have, buffer_.data(), tramp, &cb, &done) != CPT_OK) {
            error_ = cpt_last_error(ctx_);
            return;
        }
        samples_in_image_ = have + done;
    } else {  // the checkpoint already holds every requested sample: tonemap it on the host, as k_tonemap does
        for (size_t i = 0; i < accum.size(); ++i) {
            double c = accum[i] / have;
            c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
            buffer_[i] = (unsigned char)(int)(255.0 * c);
        }
    }
    if (!options.checkpoint.empty() && samples_in_image_ > 0) {
        want.samples = samples_in_image_;
        std::string tmp = options.checkpoint + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        bool ok = f && fwrite(&want, sizeof want, 1, f) == 1 && fwrite(accum.data(), sizeof(double), accum.size(), f) == accum.size();
        if (f) ok = (fclose(f) == 0) && ok;
        if (!ok || rename(tmp.c_str(), options.checkpoint.c_str()) != 0) error_ = "cannot write checkpoint " + options.checkpoint;
    }
}

void RayTracer::getBuffer(unsigned char*& buf, int& w, int& h) {
    buf = buffer_.data();
    w = buffer_width_;
    h = buffer_height_;
}

}  // namespace cpt
