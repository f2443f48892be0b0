#!/usr/bin/env python3
"""One render through the C ABI, for use under rocprofv3 (kernel trace or PMC passes).

    python tools/prof_render.py --scene cornell-box2-transmissive --width 512 --spp 16 --depth 8
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell-box2-transmissive")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--stats", action="store_true", help="time kernels with HIP events and count work")
    ap.add_argument("--count", action="store_true", help="one extra render with work counters on")
    ap.add_argument("--device-build", action="store_true", help="build the BVH on the GPU instead of uploading the host tree")
    ap.add_argument("--opt", action="append", default=[], help="key=value passed to cpt_set_option")
    args = ap.parse_args()
    import torch
    from cuda_path_tracer_amd import HostScene, Renderer
    hs = HostScene(os.path.join(ROOT, "tests", "golden", "scenes", args.scene + ".ray"))
    r = Renderer(0)
    for kv in args.opt:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    t0 = time.perf_counter()
    hs2 = HostScene(os.path.join(ROOT, "tests", "golden", "scenes", args.scene + ".ray"))
    t_host = time.perf_counter() - t0
    del hs2
    for rep in range(3):
        t0 = time.perf_counter()
        if args.device_build:
            r.build(hs)
        else:
            r.upload(hs)
        r.synchronize()
        t_up = time.perf_counter() - t0
    print("%s: %d triangles; host parse+build+flatten %.1f ms; %s %.2f ms; tree height %d" %
          (args.scene, hs.n_tris, t_host * 1e3, "device build (upload, build, sync)" if args.device_build else "upload of host tree",
           t_up * 1e3, r.download_tree()[2]), flush=True)
    w = args.width
    h = hs.height_for_width(w)
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    p = r.params(w, h, 0, args.spp, args.depth, mode=args.mode)
    if args.stats:
        r.set_instrumentation(time_kernels=True, count_work=False)
    for _ in range(args.repeat):
        acc.zero_()
        r.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.render_into(acc.data_ptr(), p)
        r.synchronize()
        dt = time.perf_counter() - t0
        print("%s %dx%d spp %d depth %d: %.2f ms, %.1f Mpaths/s, sum %.6f" %
              (args.scene, w, h, args.spp, args.depth, dt * 1e3, w * h * args.spp / dt / 1e6, float(acc.sum())), flush=True)
        if args.stats:
            st = r.stats()
            print("   ms: generate %.2f extend %.2f shade %.2f total %.2f (launches: %d extend)" %
                  (st["ms_generate"], st["ms_extend"], st["ms_shade"], st["ms_total"], st["launches_extend"]), flush=True)
    if args.count:
        r.set_instrumentation(time_kernels=False, count_work=True)
        acc.zero_()
        r.reset_stats()
        r.render_into(acc.data_ptr(), p)
        r.synchronize()
        st = r.stats()
        print("   work: segments %d node_steps %d tri_tests %d exact_node_steps %d (%.3f%%)" %
              (st["segments"], st["node_steps"], st["tri_tests"], st["exact_node_steps"],
               100.0 * st["exact_node_steps"] / max(1, st["node_steps"])), flush=True)
        print("   lanes: box passes %d (%.1f%% of lane slots used), leaf passes %d (%.1f%%)" %
              (st["box_passes"], 100.0 * st["node_steps"] / max(1, 64 * st["box_passes"]),
               st["leaf_passes"], 100.0 * st["tri_tests"] / max(1, 64 * st["leaf_passes"])), flush=True)


if __name__ == "__main__":
    main()
