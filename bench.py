#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|dragon|cornell2]

A step renders the whole workload once (all samples of every pixel of every scene in it) with the
wavefront pipeline, scenes already resident in HBM.  With N > 1 (launched under torch.distributed.run,
one rank per GPU) every rank renders its own, disjoint range of sample indices of the same image
(weak scaling: per-GPU work fixed) and the per-rank radiance sums are reduced to rank 0 over RCCL.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak

WORKLOADS = {
    # the headline configuration: both mixed-lobe Cornell boxes at 512x512, 256 spp, depth 8
    "cfg2": dict(desc="cornell-box2-specular + cornell-box2-transmissive, 512x512, 256 spp, depth 8",
                 scenes=["cornell-box2-specular", "cornell-box2-transmissive"], width=512, spp=256, depth=8),
    # the large-mesh configuration
    "dragon": dict(desc="dragon-spec, 1024x1024, 1024 spp, depth 8", scenes=["dragon-spec"], width=1024, spp=1024,
                   depth=8),
    # the size the reference's CPU estimator can run
    "cornell2": dict(desc="cornell-box2, 512x512, 16 spp, depth 5", scenes=["cornell-box2"], width=512, spp=16,
                     depth=5),
}


def scene_file(name):
    return os.path.join(ROOT, "tests", "golden", "scenes", name + ".ray")


def extend_bytes(st):
    """Algorithmic bytes the extend kernel moves: per expanded node one 104-byte record (two child
    boxes + two child references), per triangle test one 100-byte record (three vertices, unit
    normal, material/flags), per segment the ray it reads (origin + direction, 48 bytes) and the hit
    record it writes (t + triangle id + padding, 16 bytes)."""
    return 104 * st["node_steps"] + 100 * st["tri_tests"] + (48 + 16) * st["segments"]


def pipeline_bytes(st):
    """Algorithmic bytes of the whole path: node and triangle records as above, plus per hit the
    shading inputs (vertex normals 72 + material 136), per segment the path state written once and
    read once (88 x 2) and the hit record written and read (16 x 2), per path the 24-byte
    accumulation update."""
    return (104 * st["node_steps"] + 100 * st["tri_tests"] + 208 * st["hits"] + (176 + 32) * st["segments"]
            + 24 * st["paths"])


def cpu_baseline(workload, seconds_target=12.0):
    """The reference's CPU path tracer (oracle restatement), timed on this box's host cores on a
    bounded sample of the same workload: a 64x64-pixel render of each scene at the workload's depth."""
    from oracle import pyoracle
    threads = min(os.cpu_count() or 1, 16)
    paths = 0
    secs = 0.0
    w = 64
    spp = 1
    for name in workload["scenes"]:
        sc = pyoracle.OracleScene(scene_file(name))
        per_scene = seconds_target / len(workload["scenes"])
        t_scene = 0.0
        n_scene = 0
        while t_scene < per_scene and spp <= 65536:
            _, _, s = sc.render_cpu_reference(w, w, spp, workload["depth"], threads)
            t_scene += s
            n_scene += w * w * spp
            if s < per_scene / 3:
                spp *= 2
        paths += n_scene
        secs += t_scene
    return {"value": paths / secs / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": "oracle restatement of the reference CPU path tracer (branching estimator, recursive BVH walk), "
                      "64x64 px of each scene at depth %d, %d paths in %.1f s" % (workload["depth"], paths, secs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="wavefront", choices=["wavefront", "mega"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-paths", type=int, default=1 << 27,
                    help="path slots per batch; the library's own default (2^25) suits short-lived processes, a long-lived one "
                         "can afford the larger allocation and gets shorter launch tails (0: library default)")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight at once (library default when 0)")
    ap.add_argument("--device-bvh", action="store_true",
                    help="have the GPU build the BVH (a Morton-order tree) instead of uploading the reference's tree; "
                         "same image up to ties, different visit sets, so not the parity configuration")
    ap.add_argument("--opt", action="append", default=[], help="key=value passed to cpt_set_option (tuning experiments)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from cuda_path_tracer_amd import MODE_MEGAKERNEL, MODE_WAVEFRONT, HostScene, Renderer
    from cuda_path_tracer_amd.dist import reduce_radiance

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
    # CPT_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share GPUs)
    backend = os.environ.get("CPT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # communicator set-up (seconds for RCCL) must never land in the timed region, whatever --warmup says
        probe = torch.ones(1, dtype=torch.float64, device="cuda")
        reduce_radiance(probe, dst=0)
        torch.cuda.synchronize()

    verbose = bool(os.environ.get("CPT_BENCH_VERBOSE"))

    def say(msg):
        if verbose:
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    say("process group up, backend %s, device %d" % (backend if world > 1 else "-", local_rank))
    wl = WORKLOADS[args.workload]
    mode = MODE_WAVEFRONT if args.mode == "wavefront" else MODE_MEGAKERNEL
    # a stream of our own: the legacy default stream serialises against every other stream of the process,
    # which would keep the renderer's second lane from overlapping the first
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    renderers = []
    for name in wl["scenes"]:
        hs = HostScene(scene_file(name))
        r = Renderer(local_rank)
        r.set_stream(stream.cuda_stream)
        if args.batch_paths:
            r.set_option("batch_paths", args.batch_paths)
        if args.streams:
            r.set_option("streams", args.streams)
        for kv in args.opt:
            k, v = kv.split("=")
            r.set_option(k, int(v))
        if args.device_bvh:
            r.build(hs)
        else:
            r.upload(hs)
        w = wl["width"]
        h = hs.height_for_width(w)
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        rgb = torch.zeros(h, w, 3, dtype=torch.uint8, device="cuda")
        renderers.append((r, hs, w, h, acc, rgb))
    spp = wl["spp"]
    s0, s1 = rank * spp, (rank + 1) * spp  # weak scaling: every rank adds `spp` new sample indices

    def step(collective=True):
        for (r, hs, w, h, acc, rgb) in renderers:
            acc.zero_()
            r.render_into(acc.data_ptr(), r.params(w, h, s0, s1, wl["depth"], mode=mode))
            if collective:
                reduce_radiance(acc, dst=0)
            if rank == 0:
                r.tonemap_into(acc.data_ptr(), w, h, spp * world, rgb.data_ptr())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    say("scenes uploaded")
    for _ in range(args.warmup):
        step()
        say("warmup step queued")
    fence()
    say("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    say("timed steps done in %.3f s" % elapsed)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    paths_per_step_per_gpu = sum(w * h * spp for (_, _, w, h, _, _) in renderers)
    value = world * args.steps * paths_per_step_per_gpu / elapsed / 1e6

    out = None
    if rank == 0:
        # one instrumented step for per-kernel HIP-event time, one counting step for the work done
        roof = None
        for (r, *_rest) in renderers:
            r.reset_stats()
            r.set_instrumentation(time_kernels=True, count_work=False)
        step(collective=False)  # rank 0 alone: the other ranks are already at the final barrier
        torch.cuda.synchronize()
        timed = [r.stats() for (r, *_rest) in renderers]
        for (r, *_rest) in renderers:
            r.reset_stats()
            r.set_instrumentation(time_kernels=False, count_work=True)
        step(collective=False)
        torch.cuda.synchronize()
        counted = [r.stats() for (r, *_rest) in renderers]
        for (r, *_rest) in renderers:
            r.set_instrumentation(False, False)
        agg = {k: sum(c[k] for c in counted) for k in ("paths", "segments", "hits", "node_steps", "tri_tests",
                                                        "exact_node_steps", "box_passes", "leaf_passes")}
        if mode == MODE_WAVEFRONT:
            ms = sum(t["ms_extend"] for t in timed)
            launches = sum(t["launches_extend"] for t in timed)
            kernel, kbytes = "k_extend_pw", extend_bytes(agg)
        else:
            ms = sum(t["ms_mega"] for t in timed)
            launches = sum(t["launches_mega"] for t in timed)
            kernel, kbytes = "k_megakernel", pipeline_bytes(agg)
        achieved = kbytes / (ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get(kernel)
            except Exception:
                traffic = None
        # fp64 instruction issue, the resource the extend kernel actually runs out of: instructions per lane on its two
        # hot paths (counted in the kernel's ISA: 75 per node expansion with both box tests, 89 per triangle test)
        # against 256 CUs x 4 SIMDs x 16 lanes per cycle at 2.4 GHz.  `useful` counts lanes that had work,
        # `occupied` every lane of the passes that ran.
        fp64_peak = 256 * 4 * 16 * 2.4e9
        fp64 = None
        if mode == MODE_WAVEFRONT and ms > 0:
            useful = 75.0 * agg["node_steps"] + 89.0 * agg["tri_tests"]
            occupied = 75.0 * 64 * agg["box_passes"] + 89.0 * 64 * agg["leaf_passes"]
            fp64 = {"per_node_step": 75, "per_triangle_test": 89, "peak_Tinstr_per_s": fp64_peak / 1e12,
                    "useful_frac": useful / (ms * 1e-3) / fp64_peak, "occupied_frac": occupied / (ms * 1e-3) / fp64_peak}
        roof = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "bytes_per_launch": kbytes / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                "launches_per_step": launches,
                "note": "algorithmic bytes count every node and triangle record a ray touches; the kernel serves them "
                        "from a per-CU LDS copy (and L2), so the rate can exceed the HBM peak and `traffic` (measured HBM "
                        "bytes per launch) is far below bytes_per_launch; the kernel's physical bound is fp64 VALU issue",
                "lane_slots_used": {"box": agg["node_steps"] / max(64 * agg["box_passes"], 1),
                                    "triangle": agg["tri_tests"] / max(64 * agg["leaf_passes"], 1)},
                "exact_box_fallbacks": agg["exact_node_steps"], "fp64_issue": fp64,
                "pipeline_GBps": pipeline_bytes(agg) / (sum(t["ms_total"] for t in timed) * 1e-3) / 1e9,
                "per_segment": {"node_steps": agg["node_steps"] / max(agg["segments"], 1),
                                "tri_tests": agg["tri_tests"] / max(agg["segments"], 1),
                                "segments_per_path": agg["segments"] / max(agg["paths"], 1)},
                "kernel_ms_per_step": {k: sum(t[k] for t in timed) for k in
                                       ("ms_generate", "ms_extend", "ms_shade", "ms_mega", "ms_total")}}
        out = {"metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": wl["desc"], "mode": args.mode, "batch_paths": args.batch_paths,
                          "bvh": "built on the GPU (Morton order)" if args.device_bvh else "reference tree (host build)", "paths_per_step": world * paths_per_step_per_gpu,
                          "sharding": "spp range per rank, fp64 radiance reduce to rank 0" if world > 1 else "single GPU"},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
