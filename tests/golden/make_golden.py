#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference cannot run in this environment (see oracle/oracle.h), so these vectors pin the
oracle's own output: any later change to the oracle, the host BVH build or the HIP traversal that
alters a primary hit, a hit distance or a visited-node set shows up as a diff against them.

primary_<scene>_64.npz     64x64 pixel-centre camera rays: direction bits, hit triangle, t bits,
                           visit hash, node steps, triangle tests
primary_digests.json       sha256 of the same arrays for 512x512 rays (too large to store)
luminance.json             mean Rec.709 luminance of the reference's own golden renders
                           (important_assets/Renders/benchmark-renders/bvh_renders, measured with PIL)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle  # noqa: E402

SCENES = ["cornell-box2", "cornell-box2-transmissive", "teapot-trans", "dragon-spec"]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    digests = {}
    for name in SCENES:
        sc = pyoracle.OracleScene(os.path.join(HERE, "scenes", name + ".ray"))
        o, d = sc.camera_rays(64, 64)
        r = sc.trace(o, d)
        np.savez_compressed(os.path.join(HERE, "primary_%s_64.npz" % name), dir_bits=d.view(np.uint64),
                            origin_bits=o[0].view(np.uint64), prim=r["prim"], t_bits=r["t"].view(np.uint64),
                            vhash=r["vhash"], node_steps=r["node_steps"].astype(np.int32),
                            tri_tests=r["tri_tests"].astype(np.int32))
        o, d = sc.camera_rays(512, 512)
        r = sc.trace(o, d)
        digests[name] = dict(n_prims=sc.n_prims, n_nodes=sc.n_nodes, height=sc.height, dir=digest(d), prim=digest(r["prim"]),
                             t=digest(r["t"]), vhash=digest(r["vhash"]), node_steps=int(r["node_steps"].sum()),
                             tri_tests=int(r["tri_tests"].sum()), hits=int((r["prim"] >= 0).sum()))
        print(name, digests[name]["node_steps"], digests[name]["hits"])
    json.dump(digests, open(os.path.join(HERE, "primary_digests.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
