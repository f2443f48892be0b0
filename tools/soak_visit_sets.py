#!/usr/bin/env python3
"""Soak of the persistent extend kernel's reference walk against the CPU oracle: per ray the hit triangle, the bits of
the hit distance and the order-free hash of the visited-node set must be the oracle's, on many more rays than the test
suite takes, of the kinds that sit close to the box filter's margins: rays aimed at box corners or starting on box
planes, axis-parallel rays, random rays, and rays that leave the scene's own surfaces (a first generation's hit points,
pushed 1e-12 along a random direction of the hemisphere, as shading would).

    python tools/soak_visit_sets.py [rounds] [rays per kind] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from cuda_path_tracer_amd import HostScene, Renderer
    from oracle import pyoracle
    from test_gpu_parity import _tie_prone_rays, bits
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    gpu = Renderer(0)
    total = 0
    for name in ("cornell-box2-transmissive", "teapot-trans", "dragon-spec"):
        path = os.path.join(ROOT, "tests", "golden", "scenes", name + ".ray")
        gpu.upload(HostScene(path))
        orc = pyoracle.OracleScene(path)
        boxes, _ = orc.nodes()
        lo, hi = boxes[0, :3], boxes[0, 3:]
        for k in range(rounds):
            o1 = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
            d1 = rng.normal(size=(n, 3))
            d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
            o2, d2 = _tie_prone_rays(orc, rng, n)
            first = orc.trace(o1, d1)
            hit = first["prim"] >= 0
            p = o1[hit] + d1[hit] * first["t"][hit][:, None]
            d3 = rng.normal(size=p.shape)
            d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
            o3 = p + d3 * 1e-12
            o = np.concatenate([o1, o2, o3])
            d = np.concatenate([d1, d2, d3])
            want = orc.trace(o, d)
            got = gpu.extend_rays_observed(o, d)
            assert np.array_equal(got["prim"], want["prim"]), (name, k, "hit triangles differ")
            assert np.array_equal(bits(got["t"]), bits(want["t"])), (name, k, "hit distances differ")
            assert np.array_equal(got["vhash"], want["vhash"]), (name, k, "visited-node sets differ")
            total += len(o)
            print("%s round %d: %d rays (%d from surfaces), hits, distances and visit sets equal the oracle's" %
                  (name, k, len(o), len(o3)), flush=True)
    gpu.close()
    print("soak done: %d rays" % total)


if __name__ == "__main__":
    main()
