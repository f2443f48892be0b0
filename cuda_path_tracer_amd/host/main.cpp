// bin/ray: the reference's command line for the path tracer,
//   ray [options] input.ray output.png
// with the reference's option letters (reference: src/ui/CommandLineUI.cpp:23-137, src/main.cu:22-41):
//   -r depth  -w width  -s samples  -g (GPU)  -j json  -c cubemap  -h
// and a few long options for what the reference fixes at compile time.
#include <getopt.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "png_write.h"
#include "raytracer.h"

namespace {

struct Cli {
    int depth = 0;      // reference default: src/ui/TraceUI.h:101
    int size = 512;     // reference default: src/ui/TraceUI.h:102
    int samples = 32;   // reference default: src/ui/CommandLineUI.cpp:30
    bool gpu = false;
    std::string json, cubemap, metrics;
    cpt::PathTracingOptions pt;
};

struct ProgressiveOutput {  // rewrites the output image while a long render runs
    const char* name;
};
bool write_progress(void* user, int samples, const unsigned char* rgb, int w, int h) {
    const ProgressiveOutput* po = (const ProgressiveOutput*)user;
    std::string err;
    if (!cpt::write_image(po->name, w, h, rgb, err)) std::cerr << err << std::endl;
    std::cout << "  " << samples << " samples per pixel written to " << po->name << std::endl;
    return false;
}

void usage(const char* prog, const Cli& c) {
    std::cerr << "usage: " << prog << " [options] [input.ray output.png]\n"
              << "  -r <#>      set recursion level (default " << c.depth << ")\n"
              << "  -w <#>      set output image width (default " << c.size << ")\n"
              << "  -j <FILE>   set parameters from JSON file\n"
              << "  -c <FILE>   one Cubemap file (accepted; the path tracer does not use cubemaps)\n"
              << "  -g          Enable GPU rendering (MI355X, HIP)\n"
              << "  -s <#>      Set number of samples per pixel (path tracing)\n"
              << "  --mode wavefront|mega   work-queue pipeline (default) or one thread per pixel\n"
              << "  --dof <focal>,<aperture>   thin-lens camera\n"
              << "  --rr        Russian roulette\n"
              << "  --seed <#>  random seed (default 1984)\n"
              << "  --device <#> GPU ordinal (default 0)\n"
              << "  --batch <#>  path slots per batch (209 bytes each, two batches in flight; default 2^25, at most 2^28)\n"
              << "  --devices <#,#,...>  split the samples over these GPUs (one host thread and context each)\n"
              << "  --device-bvh        build the BVH on the GPU (a Morton-order tree, not the reference's)\n"
              << "  --fast              fast mode: surface-area tree with the reference's tie-break ranks (same image, bit for bit,\n"
              << "                      from about a third fewer box tests), nearest-first culling walk on large scenes\n"
              << "  --progressive <#>   rewrite the output image every # samples per pixel\n"
              << "  --checkpoint <FILE> resume from FILE if it matches this render, and save the radiance sums to it\n"
              << "  --metrics <FILE>    write timing and work counters as JSON\n";
}

// Flat JSON object with numeric values: of the keys the reference reads (src/ui/TraceUI.cc:46-76) "size" and
// "recursion_depth" are the two this renderer has a setting for.
bool json_number(const std::string& text, const char* key, double& out) {
    std::string pat = std::string("\"") + key + "\"";
    size_t p = text.find(pat);
    if (p == std::string::npos) return false;
    p = text.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    char* end = nullptr;
    double v = std::strtod(text.c_str() + p + 1, &end);
    if (end == text.c_str() + p + 1) return false;
    out = v;
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    Cli c;
    static option longopts[] = {{"mode", required_argument, nullptr, 1000}, {"dof", required_argument, nullptr, 1001},
                                {"rr", no_argument, nullptr, 1002},         {"seed", required_argument, nullptr, 1003},
                                {"device", required_argument, nullptr, 1004}, {"progressive", required_argument, nullptr, 1005},
                                {"checkpoint", required_argument, nullptr, 1006}, {"metrics", required_argument, nullptr, 1007},
                                {"device-bvh", no_argument, nullptr, 1008},     {"devices", required_argument, nullptr, 1009},
                                {"batch", required_argument, nullptr, 1010},    {"fast", no_argument, nullptr, 1011},
                                {nullptr, 0, nullptr, 0}};
    int o;
    while ((o = getopt_long(argc, argv, "tr:w:hj:c:gs:", longopts, nullptr)) != -1) {
        switch (o) {
            case 'r': c.depth = atoi(optarg); break;
            case 'w': c.size = atoi(optarg); break;
            case 'j': c.json = optarg; break;
            case 'c': c.cubemap = optarg; break;
            case 'h': usage(argv[0], c); return 1;
            case 'g': c.gpu = true; break;
            case 's': c.samples = atoi(optarg); break;
            case 't': break;
            case 1000:
                if (!strcmp(optarg, "wavefront")) c.pt.work_queue = true;
                else if (!strcmp(optarg, "mega")) c.pt.work_queue = false;
                else { std::cerr << "unknown mode '" << optarg << "'\n"; return 1; }
                break;
            case 1001:
                if (sscanf(optarg, "%lf,%lf", &c.pt.focal_distance, &c.pt.aperture) != 2) { std::cerr << "--dof wants <focal>,<aperture>\n"; return 1; }
                c.pt.use_dof = true;
                break;
            case 1002: c.pt.russian_roulette = true; break;
            case 1003: c.pt.seed = strtoull(optarg, nullptr, 10); break;
            case 1004: c.pt.device = atoi(optarg); break;
            case 1005: c.pt.progressive_spp = atoi(optarg); break;
            case 1006: c.pt.checkpoint = optarg; break;
            case 1007: c.metrics = optarg; break;
            case 1008: c.pt.device_bvh = true; break;
            case 1010: c.pt.batch_paths = atoll(optarg); break;
            case 1011: c.pt.fast = true; break;
            case 1009:
                c.pt.devices.clear();
                for (const char* q = optarg; *q;) {
                    char* end = nullptr;
                    long v = strtol(q, &end, 10);
                    if (end == q || v < 0) { std::cerr << "--devices wants a comma-separated list of GPU ordinals\n"; return 1; }
                    c.pt.devices.push_back((int)v);
                    q = (*end == ',') ? end + 1 : end;
                    if (*end && *end != ',') { std::cerr << "--devices wants a comma-separated list of GPU ordinals\n"; return 1; }
                }
                break;
            default:
                std::cerr << "Invalid argument: '" << o << "'." << std::endl;
                usage(argv[0], c);
                return 1;
        }
    }
    if (!c.json.empty()) {
        std::ifstream f(c.json);
        std::stringstream ss;
        ss << f.rdbuf();
        double v;
        if (json_number(ss.str(), "size", v)) c.size = (int)v;
        if (json_number(ss.str(), "recursion_depth", v)) c.depth = (int)v;
        // The other keys the reference reads (src/ui/TraceUI.cc:52-75) steer its CPU thread pool, its Whitted tracer
        // and its kd-tree; the GPU path tracer has nothing for them to set, so they are accepted and named.
        static const char* const unused[] = {"threads", "threshold", "blocksize", "supersamples", "aa_threshold", "tree_depth",
                                             "leaf_size", "filter_width", "anti_alias", "kdtree", "shadows", "smoothshade",
                                             "backface_culling", "internal_reflection", "backface_specular"};
        for (const char* key : unused)
            if (ss.str().find(std::string("\"") + key + "\"") != std::string::npos)
                std::cerr << "note: -j key '" << key << "' has no effect on the GPU path tracer" << std::endl;
    }
    if (optind >= argc - 1) {
        std::cerr << "no input and/or output name." << std::endl;
        return 1;
    }
    const char* rayName = argv[optind];
    const char* imgName = argv[optind + 1];

    ProgressiveOutput po{imgName};
    if (c.pt.progressive_spp > 0) { c.pt.on_progress = write_progress; c.pt.progress_user = &po; }
    cpt::RayTracer rt;
    rt.options = c.pt;
    if (!rt.loadScene(rayName)) {
        std::cerr << "Unable to load ray file '" << rayName << "'" << std::endl;
        return 1;
    }
    int width = c.size;
    int height = (int)(width / rt.aspectRatio() + 0.5);
    if (width < 1 || height < 1 || c.samples < 1 || c.depth < 0) { std::cerr << "bad size, samples or depth" << std::endl; return 1; }
    rt.traceSetup(width, height, c.gpu, c.samples, c.depth);
    auto t0 = std::chrono::steady_clock::now();
    rt.traceImage(width, height);
    rt.waitRender();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rt.failed()) {
        std::cerr << "render failed: " << rt.lastError() << std::endl;
        return 2;
    }
    unsigned char* buf;
    rt.getBuffer(buf, width, height);
    std::string err;
    if (!cpt::write_image(imgName, width, height, buf, err)) {
        std::cerr << err << std::endl;
        return 3;
    }
    std::cout << "total time = " << secs << " seconds, " << (double)width * height * c.samples / secs / 1e6 << " Mpaths/s" << std::endl;
    if (!c.metrics.empty()) {
        cpt_stats st{};
        rt.stats(st);
        std::ofstream m(c.metrics);
        m << "{\"scene\": \"" << rayName << "\", \"width\": " << width << ", \"height\": " << height
          << ", \"samples_requested\": " << c.samples << ", \"samples_in_image\": " << rt.samplesInImage()
          << ", \"depth\": " << c.depth << ", \"mode\": \"" << (c.pt.work_queue ? "wavefront" : "mega") << "\""
          << ", \"seconds\": " << secs << ", \"paths\": " << st.paths
          << ", \"mpaths_per_s\": " << (secs > 0 ? (double)st.paths / secs / 1e6 : 0.0)
          << ", \"launches\": {\"generate\": " << st.launches_generate << ", \"extend\": " << st.launches_extend
          << ", \"shade\": " << st.launches_shade << ", \"mega\": " << st.launches_mega << "}}" << std::endl;
        if (!m) { std::cerr << "cannot write " << c.metrics << std::endl; return 3; }
    }
    return 0;
}
