#!/usr/bin/env python3
"""Times the device-side BVH build (cpt_build_scene) on a synthetic soup of n small triangles handed over
through the C ABI, and checks the result by tracing rays against it (every ray aimed at a triangle's centre
must hit something no farther than that triangle).

    python tools/bvh_build_bench.py --tris 1000000
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    from cuda_path_tracer_amd import ArrayScene, Renderer
    n = args.tris
    rng = np.random.default_rng(1)
    centre = rng.uniform(-1, 1, (n, 1, 3))
    hs = ArrayScene(centre + rng.normal(size=(n, 3, 3)) * (0.5 / n ** (1 / 3)))
    tris = hs.tris()
    r = Renderer(0)
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        r.build(hs)
        dt = time.perf_counter() - t0
        print("cpt_build_scene: %d triangles in %.2f ms (upload of %.0f MB of triangles and vertex normals included); tree height %d" %
              (n, dt * 1e3, n * (128 + 72) / 1e6, r.download_tree()[2]), flush=True)
    m = min(n, 20000)
    pick = rng.integers(0, n, m)
    target = (tris["a"][pick] + tris["b"][pick] + tris["c"][pick]) / 3.0
    o = target + rng.normal(size=(m, 3)) * 3.0
    d = target - o
    dist = np.linalg.norm(d, axis=1)
    d /= dist[:, None]
    tri, t = r.extend_rays(o, d)
    ok = (tri >= 0) & (t <= dist * (1 + 1e-9))
    print("rays aimed at triangle centres: %d of %d hit at or before their target" % (int(ok.sum()), m))
    assert ok.all()


if __name__ == "__main__":
    main()
