// Host-side code under AddressSanitizer and UndefinedBehaviorSanitizer: the scene reader, BVH build, flattening and
// PNG writer of the product's host library, and the oracle's loader, build, traversal and both estimators, on the
// scene files given on the command line.  Built and run by tests/test_sanitizers.py.
#include <cstdio>
#include <string>
#include <vector>
#include "flatten.h"
#include "png_write.h"
#include "oracle.h"
int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        std::string path = argv[a];
        cpt::FlatScene f = cpt::load_flat_scene(path);
        cpt::FlatScene g = cpt::load_flat_scene(path, false);
        std::printf("%s: host %zu tris %zu nodes height %d (no-tree load: %zu nodes)\n", path.c_str(), f.tris.size(), f.nodes.size(), f.bvh.height, g.nodes.size());
        std::string err;
        orc::Scene* s = orc::load_scene(path, err);
        if (!s) { std::printf("oracle load failed: %s\n", err.c_str()); return 1; }
        orc::build_bvh(*s);
        const int w = 12, h = 12;
        std::vector<double> acc((size_t)w * h * 3, 0.0);
        orc::RenderParams p; p.w = w; p.h = h; p.s0 = 0; p.s1 = 2; p.depth = 4; p.seed = 1984;
        orc::RenderStats st;
        orc::render_gpu_semantics(*s, p, acc.data(), &st);
        std::vector<unsigned char> rgb((size_t)w * h * 3);
        std::vector<double> mean((size_t)w * h * 3);
        orc::render_cpu_reference(*s, w, h, 1, 2, 2, 1, rgb.data(), mean.data());
        std::string e2;
        cpt::write_image((std::string(argv[0]) + ".png").c_str(), w, h, rgb.data(), e2);
        std::printf("   oracle: %zu prims, %zu nodes, segments %llu\n", s->prims.size(), s->nodes.size(), (unsigned long long)st.segments);
        delete s;
    }
    return 0;
}
