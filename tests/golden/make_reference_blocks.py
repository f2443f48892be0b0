#!/usr/bin/env python3
"""Block means of the reference's own golden renders, as a fixture for a per-block RMSE test.

    python tests/golden/make_reference_blocks.py [/root/reference]

For every (scene, depth, estimator) the reference holds 512x512 renders for
(important_assets/Renders/benchmark-renders/bvh_renders/{scene}.ray_{depth}_{spp}_{cpu|gpu}.png) the
best-converged image is reduced to the mean RGB of its 64x64 blocks of 8x8 pixels and stored,
bottom row first, in reference_blocks.npz under "{scene}|{depth}|{cpu|gpu}" (float32, 64 x 64 x 3,
0-255 scale), with the file name it came from in reference_blocks.json.  Only these means are stored,
not the images.
"""
import json
import os
import re
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    d = os.path.join(ref, "important_assets", "Renders", "benchmark-renders", "bvh_renders")
    best = {}
    for f in sorted(os.listdir(d)):
        m = re.match(r"(.+)\.ray_(\d+)_(\d+)_(cpu|gpu)\.png$", f)
        if not m:
            continue
        key = (m.group(1), int(m.group(2)), m.group(4))
        spp = int(m.group(3))
        if key not in best or spp > best[key][0]:
            best[key] = (spp, f)
    arrays, names = {}, {}
    for (scene, depth, est), (spp, f) in sorted(best.items()):
        img = np.asarray(Image.open(os.path.join(d, f)).convert("RGB"), np.float64)
        if img.shape != (512, 512, 3):
            continue
        img = img[::-1]  # PNG rows run top-down; the renderer's buffer is bottom-up
        blocks = img.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3)).astype(np.float32)
        arrays["%s|%d|%s" % (scene, depth, est)] = blocks
        names["%s|%d|%s" % (scene, depth, est)] = f
        print(scene, depth, est, f, "mean", blocks.mean())
    np.savez_compressed(os.path.join(HERE, "reference_blocks.npz"), **arrays)
    json.dump(names, open(os.path.join(HERE, "reference_blocks.json"), "w"), indent=1, sort_keys=True)
    # The showcase renders (important_assets/Renders/*.bmp|png, 512x512, 16 384 samples per pixel by their names;
    # README.md:49).  Their file names do not say at which depth they were made: the tests that use them say so.
    # (dragon-split-spec-16384.bmp and dragon-split-trans-16384.bmp are left out: their mean levels, 32.4 and 45.2,
    # are far below what the shipped dragon-spec.ray / dragon-trans.ray give at any depth (57.7 at depth 1, rising
    # with depth), so they were rendered from another revision of those scenes and pin nothing.)
    show = {"cornell-box": "cornell-box-16384.bmp", "teapot-trans": "teapot-trans-16384-5.png"}
    arrays = {}
    for scene, f in sorted(show.items()):
        img = np.asarray(Image.open(os.path.join(ref, "important_assets", "Renders", f)).convert("RGB"), np.float64)
        assert img.shape == (512, 512, 3), (f, img.shape)
        arrays[scene] = img[::-1].reshape(64, 8, 64, 8, 3).mean(axis=(1, 3)).astype(np.float32)
        print("showcase", scene, f, "mean", arrays[scene].mean())
    np.savez_compressed(os.path.join(HERE, "showcase_blocks.npz"), **arrays)
    json.dump(show, open(os.path.join(HERE, "showcase_blocks.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
