#include "raytracer.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <thread>

namespace cpt {

RayTracer::RayTracer() = default;

RayTracer::~RayTracer() {
    if (ctx_) cpt_destroy(ctx_);
    for (cpt_ctx* c : more_ctx_) if (c) cpt_destroy(c);
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

// FNV-1a over the values a render depends on: triangles (vertices, normal, material, flags), vertex normals,
// materials, camera.  Field by field, so that padding never takes part.
static uint64_t scene_digest(const FlatScene& s) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    for (const cpt_tri& t : s.tris) { mix(t.a, sizeof t.a); mix(t.b, sizeof t.b); mix(t.c, sizeof t.c); mix(t.n, sizeof t.n); mix(&t.material, 4); mix(&t.flags, 4); }
    for (const cpt_tri_normals& n : s.tri_normals) { mix(n.na, sizeof n.na); mix(n.nb, sizeof n.nb); mix(n.nc, sizeof n.nc); }
    for (const cpt_material& m : s.materials) { mix(m.ke, sizeof m.ke); mix(m.kd, sizeof m.kd); mix(m.ks, sizeof m.ks); mix(m.kr, sizeof m.kr); mix(m.kt, sizeof m.kt); mix(&m.index, 8); mix(&m.flags, 4); }
    mix(s.camera.eye, sizeof s.camera.eye); mix(s.camera.look, sizeof s.camera.look); mix(s.camera.u, sizeof s.camera.u); mix(s.camera.v, sizeof s.camera.v);
    return h;
}

cpt_scene_desc RayTracer::sceneDesc() {
    const bool has_host_tree = !scene_->nodes.empty();
    return (options.fast && has_host_tree) ? scene_->describe_fast() : scene_->describe();
}
// Culling pays on scenes whose rays could otherwise walk into many boxes behind their closest hit; on the small
// bundled scenes (closed rooms of a few hundred triangles) the plain walk of the better tree is the faster one.
bool RayTracer::applyWalkOptions(cpt_ctx* c) {
    if (!options.fast) return true;
    return cpt_set_option(c, "extend_variant", scene_->tris.size() > 2048 ? 2 : 1) == CPT_OK;
}

void RayTracer::traceImageGPU(int w, int h) { renderGPU(w, h, CPT_MODE_MEGAKERNEL); }
void RayTracer::betterPathTracing(int w, int h) { renderGPU(w, h, CPT_MODE_WAVEFRONT); }

void RayTracer::renderGPU(int w, int h, int mode) {
    error_.clear();
    if (!sceneLoaded()) { error_ = "no scene loaded"; return; }
    const bool multi = options.devices.size() > 1;
    if (multi && (options.progressive_spp > 0 || !options.checkpoint.empty())) {
        error_ = "progressive and checkpointed renders run on one GPU";
        return;
    }
    if (!ctx_) {
        if (cpt_create(multi ? options.devices[0] : options.device, &ctx_) != CPT_OK) { error_ = cpt_last_error(nullptr); ctx_ = nullptr; return; }
        if (options.batch_paths > 0 && cpt_set_option(ctx_, "batch_paths", options.batch_paths) != CPT_OK) { error_ = cpt_last_error(ctx_); return; }
    }
    if (!uploaded_) {
        cpt_scene_desc d = sceneDesc();
        if (!applyWalkOptions(ctx_)) { error_ = cpt_last_error(ctx_); return; }
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
    if (multi) { renderMultiGPU(w, h, p); return; }
    if (options.progressive_spp <= 0 && options.checkpoint.empty()) {
        if (cpt_render_image(ctx_, &p, buffer_.data()) != CPT_OK) error_ = cpt_last_error(ctx_);
        else samples_in_image_ = samples_per_pixel_;
        return;
    }
    // Long render: chunks, optional checkpoint.  The checkpoint holds the fp64 radiance sums of samples
    // [0, have) and the parameters they depend on; a sample's paths depend only on (pixel, sample
    // index, seed), so resuming adds exactly the samples a single run would have added.
    // ... and a digest of the scene and the kind of tree it is rendered on: sums of another scene, or of the same
    // scene on the other tree (whose ties between equal hits fall differently), must not be continued.
    struct Header { char magic[8]; int32_t w, h, depth, mode, dof, rr, samples, n_tris; uint64_t seed; double focal, aperture;
                    uint64_t scene_digest; int32_t device_tree, pad; };
    Header want{};
    memcpy(want.magic, "CPTCKPT2", 8);
    want.n_tris = (int32_t)scene_->tris.size();
    want.scene_digest = scene_digest(*scene_);
    want.device_tree = (scene_->nodes.empty() && scene_->tris.size() > 1) ? 1 : 0;
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

// One render on several GPUs of this machine: a context per listed GPU is created and given the scene from a host
// thread of its own (uploads of large scenes run side by side), then cpt_render_image_multi deals the samples to
// them, adds the radiance sums on the first GPU (device-to-device copies, one add kernel per peer, list order) and
// tonemaps there.  A sample's paths depend only on (pixel, sample index, seed), so the sample set is the single-GPU
// one; only the association of the final additions differs.  No reference counterpart (the reference drives one GPU).
void RayTracer::renderMultiGPU(int /*w*/, int /*h*/, const cpt_render_params& p) {
    const size_t n = options.devices.size();
    more_ctx_.resize(n - 1, nullptr);
    const cpt_scene_desc d = sceneDesc();
    const bool on_device = scene_->nodes.empty() && scene_->tris.size() > 1;
    std::vector<std::string> errors(n);
    auto prepare = [&](size_t k) {
        cpt_ctx*& c = k == 0 ? ctx_ : more_ctx_[k - 1];
        if (!c && cpt_create(options.devices[k], &c) != CPT_OK) { errors[k] = cpt_last_error(nullptr); c = nullptr; return; }
        if (k == 0) return;  // the first context was set up by the caller
        if (options.batch_paths > 0 && cpt_set_option(c, "batch_paths", options.batch_paths) != CPT_OK) { errors[k] = cpt_last_error(c); return; }
        if (!applyWalkOptions(c)) { errors[k] = cpt_last_error(c); return; }
        if ((on_device ? cpt_build_scene(c, &d) : cpt_upload_scene(c, &d)) != CPT_OK) { errors[k] = cpt_last_error(c); return; }
    };
    std::vector<std::thread> pool;
    for (size_t k = 1; k < n; ++k) pool.emplace_back(prepare, k);
    prepare(0);
    for (std::thread& t : pool) t.join();
    for (size_t k = 0; k < n; ++k)
        if (!errors[k].empty()) { error_ = "GPU " + std::to_string(options.devices[k]) + ": " + errors[k]; return; }
    std::vector<cpt_ctx*> all(1, ctx_);
    all.insert(all.end(), more_ctx_.begin(), more_ctx_.end());
    int32_t have = 0;
    if (cpt_render_image_multi(all.data(), (int32_t)all.size(), &p, buffer_.data(), &have) != CPT_OK) {
        error_ = cpt_last_error(ctx_);
        return;
    }
    samples_in_image_ = have;
}

void RayTracer::getBuffer(unsigned char*& buf, int& w, int& h) {
    buf = buffer_.data();
    w = buffer_width_;
    h = buffer_height_;
}

}  // namespace cpt
