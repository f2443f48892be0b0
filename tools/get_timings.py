#!/usr/bin/env python3
"""End-to-end wall-clock timings of `bin/ray -g`, in the form of the reference's own timing tables
(reference: timings/get_timings.py:10-31 and timings/gpu_timing_bvh.txt): for every bundled Cornell scene, depths 3
and 5, 768 / 1536 / 3072 samples per pixel at the default width of 512, the time of the whole process -- start-up,
scene load, BVH build, render, PNG -- measured around the subprocess, one line per run in the reference's format
plus the Mpaths/s it corresponds to.

    python tools/get_timings.py [out.txt]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = ["cornell-box", "cornell-box2", "cornell-box2-specular", "cornell-box2-transmissive"]
DEPTHS = [3, 5]
SAMPLES = [768, 1536, 3072]


def main():
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as tmp:
        png = os.path.join(tmp, "o.png")
        subprocess.run([exe, "-g", "-s", "1", "-r", "1", os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box.ray"), png],
                       check=True, capture_output=True)  # first touch of the GPU on a fresh machine is not a render time
        for scene in SCENES:
            path = os.path.join(ROOT, "tests", "golden", "scenes", scene + ".ray")
            for depth in DEPTHS:
                for spp in SAMPLES:
                    t0 = time.time()
                    subprocess.run([exe, "-g", "-r", str(depth), "-s", str(spp), path, png], check=True, capture_output=True)
                    dt = time.time() - t0
                    line = "%s.ray - GPU|%d,%d: %.4f   (%.1f Mpaths/s end to end)" % (scene, spp, depth, dt, 512 * 512 * spp / dt / 1e6)
                    print(line, flush=True)
                    if out:
                        out.write(line + "\n")
    if out:
        out.close()


if __name__ == "__main__":
    main()
