#!/usr/bin/env python3
"""Differential soak of the extend kernels: random triangle soups of random sizes, built on the device, random and
axis-parallel rays; every persistent-wave variant must return the hits of the one-ray-per-thread kernel bit for bit.

    python tools/soak_variants.py [rounds] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    from cuda_path_tracer_amd import HostScene, Renderer
    from test_gpu_device_bvh import _mesh_text, _small_tri, bits
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    gpu = Renderer(0)
    for k in range(rounds):
        n = int(rng.choice([1, 2, 3, 17, 150, 291, 292, 293, 340, 341, 342, 700, 2500, 9000]))
        spread = 10.0 ** rng.uniform(-1, 1)
        tris = [_small_tri(rng.normal(size=3) * spread, spread * 10.0 ** rng.uniform(-2.5, -0.3), rng) for _ in range(n)]
        if rng.random() < 0.3:  # some flat, axis-aligned sheets: flat boxes, coplanar faces
            for t in tris[: n // 2]:
                for p in t:
                    p[2] = round(p[2], 1)
        gpu.build(HostScene(text=_mesh_text(tris)))
        m = 20000
        o = rng.normal(size=(m, 3)) * spread * 1.5
        d = rng.normal(size=(m, 3))
        d[: m // 10, int(rng.integers(0, 3))] = 0.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        hits = {}
        for variant in (0, 1, 2, 3, 4):
            gpu.set_option("extend_variant", variant)
            hits[variant] = gpu.extend_rays(o, d)
        for variant in (1, 2, 3, 4):
            assert np.array_equal(hits[variant][0], hits[0][0]), (k, n, variant)
            assert np.array_equal(bits(hits[variant][1]), bits(hits[0][1])), (k, n, variant)
        print("round %d: %d triangles, %d of %d rays hit, all variants agree" % (k, n, int((hits[0][0] >= 0).sum()), m), flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
