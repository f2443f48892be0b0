// Mutated scene files (truncations, byte substitutions, deletions) through both .ray readers under the sanitizers:
// every input must be either accepted or refused with an exception / error string, never crash or read out of bounds.
// usage: fuzz <scene.ray> <iterations> <scratch-file>
#include <cstdlib>
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include "flatten.h"
#include "oracle.h"
int main(int argc, char** argv) {
    std::ifstream f(argv[1]);
    std::stringstream ss; ss << f.rdbuf();
    const std::string base = ss.str();
    std::mt19937 rng(7);
    int ok = 0, bad = 0;
    const char alphabet[] = "(){};=,.-0123456789e \n\"/*abcxyz";
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 300;
    const std::string path = argc > 3 ? argv[3] : "fuzz_scratch.ray";
    for (int it = 0; it < iterations; ++it) {
        std::string t = base;
        int kind = it % 3;
        if (kind == 0) t.resize(rng() % t.size());
        else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 8); ++k) t[rng() % t.size()] = alphabet[rng() % (sizeof alphabet - 1)];
        else { size_t a = rng() % t.size(), b = rng() % t.size(); if (a > b) std::swap(a, b); t.erase(a, (b - a) % 200); }
        try { cpt::FlatScene s = cpt::flatten_scene(cpt::read_ray_text(t)); ++ok; } catch (const std::exception&) { ++bad; }
        { std::ofstream o(path); o << t; }
        std::string err;
        orc::Scene* s = orc::load_scene(path, err);
        if (s) { orc::build_bvh(*s); delete s; }
    }
    std::printf("host reader: %d accepted, %d rejected\n", ok, bad);
    return 0;
}
