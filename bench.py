#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|dragon|teapot_dof|cornell2|cfg4|cfg5_cornell|cfg5_dragon]
                    [--extra dragon,teapot_dof|none] [--scaling weak|strong]

A step renders the whole workload once (all samples of every pixel of every scene in it) with the
wavefront pipeline, scenes already resident in HBM.  The headline workload (default: BASELINE configs[1], both
mixed-lobe Cornell boxes) gives `value`; the other workloads the project is asked to report -- the dragon
(the BVH-traversal configuration) and the teapot with the thin lens -- are measured in the same run with a few
steps each and reported under `workloads`, every one with its own roofline record.

With N > 1 there is one rank per GPU (under torch.distributed.run; run as plain `python bench.py --gpus N` the script
starts that launcher itself, before anything touches a GPU).  Ranks render disjoint shares of the same image and
the per-rank fp64 radiance sums are reduced to rank 0 over RCCL: `--scaling weak` gives every rank the workload's
sample count as new sample indices (per-GPU work fixed), `--scaling strong` deals the workload's samples out
(total work fixed; by image tiles when there are fewer samples than ranks).  With N > 1 the line also carries, under
`workloads`, the full-node shapes of BASELINE configs[3] and [4] (cfg4: teapot-trans with the thin lens, 1024x1024, 4096
spp; cfg5: cornell-box2 512x512 at 16384 spp and dragon-spec 2048x2048 at 8192 spp), always with their fixed totals dealt
over the N GPUs, each with the slowest rank's render and reduce times.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak
# fp64 vector issue peak: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz lane-instructions per second
# (the 78.6 TFLOP/s vector fp64 peak counts an FMA as two)
FP64_ISSUE_PEAK = 256 * 4 * 16 * 2.4e9
# Algorithmic constants of the roofline's numerator, frozen: fp64 operations per lane of the two units of work the
# reference walk is made of.  They were counted once in the extend kernel's ISA (tools/count_fp64_isa.py, output
# committed as profiles/r02_isa_fp64_counts.txt) and are NOT re-counted when the kernel changes, so that adding
# instructions to the kernel cannot raise `achieved`.
#   node step = the division-free test of both child boxes, 2 x 35: per box 6 subtractions (plane - origin), 6 products
#     with the reciprocal direction, 6 min/max (near and far per axis), 3 max (largest near of the other two axes),
#     1 max (entry distance), 2 min (smallest far), 3 subtractions + 2 min (the gap G), 1 add + 1 fma (the margin),
#     4 compares.
#   triangle test that runs through all three edge tests, 89: o - a, two dot products, the division's 13-instruction
#     expansion, the hit point, three edge vectors, three cross products and three dot products with the normal.
F64_PER_NODE_STEP = 70
F64_PER_TRIANGLE_TEST = 89

WORKLOADS = {
    # the headline configuration: both mixed-lobe Cornell boxes at 512x512, 256 spp, depth 8
    "cfg2": dict(desc="cornell-box2-specular + cornell-box2-transmissive, 512x512, 256 spp, depth 8",
                 scenes=["cornell-box2-specular", "cornell-box2-transmissive"], width=512, spp=256, depth=8),
    # the large-mesh configuration
    "dragon": dict(desc="dragon-spec, 1024x1024, 1024 spp, depth 8", scenes=["dragon-spec"], width=1024, spp=1024,
                   depth=8),
    # thin lens through the glass teapot; BASELINE quotes 4096 spp spread over 2/4/8 GPUs, here 1024 per GPU
    "teapot_dof": dict(desc="teapot-trans, thin lens (focal distance 8.1, aperture 0.6), 1024x1024, 1024 spp, depth 8",
                       scenes=["teapot-trans"], width=1024, spp=1024, depth=8,
                       render=dict(dof=True, focal_distance=8.1, aperture=0.6)),
    # the size the reference's CPU estimator is quoted on
    "cornell2": dict(desc="cornell-box2, 512x512, 16 spp, depth 5", scenes=["cornell-box2"], width=512, spp=16,
                     depth=5),
    # BASELINE configs[3] and [4]: fixed totals dealt out over the GPUs of the run (strong scaling whatever --scaling
    # says), measured by default when N > 1
    "cfg4": dict(desc="teapot-trans, thin lens (focal distance 8.1, aperture 0.6), 1024x1024, 4096 spp in all, depth 8",
                 scenes=["teapot-trans"], width=1024, spp=4096, depth=8, scaling="strong",
                 render=dict(dof=True, focal_distance=8.1, aperture=0.6)),
    "cfg5_cornell": dict(desc="cornell-box2, 512x512, 16384 spp in all, depth 8", scenes=["cornell-box2"], width=512,
                         spp=16384, depth=8, scaling="strong"),
    "cfg5_dragon": dict(desc="dragon-spec, 2048x2048, 8192 spp in all, depth 8", scenes=["dragon-spec"], width=2048,
                        spp=8192, depth=8, scaling="strong"),
}
MULTI_GPU_EXTRAS = "cfg4,cfg5_cornell,cfg5_dragon"


if os.environ.get("CPT_BENCH_CORNELL2_SPP"):  # test hook: fewer samples than ranks exercises the tile split
    WORKLOADS["cornell2"]["spp"] = int(os.environ["CPT_BENCH_CORNELL2_SPP"])
if os.environ.get("CPT_BENCH_SHRINK"):  # test hook: the full-node shapes at 1/k of their width and 1/k^2 of their samples
    _k = int(os.environ["CPT_BENCH_SHRINK"])
    for _name in ("cfg4", "cfg5_cornell", "cfg5_dragon"):
        WORKLOADS[_name]["width"] //= _k
        WORKLOADS[_name]["spp"] //= _k * _k


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


def host_cpus():
    """(CPUs the machine has, CPUs this process may use): the second honours the affinity mask and a cgroup quota."""
    total = os.cpu_count() or 1
    usable = total
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = min(usable, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    forced = os.environ.get("CPT_BENCH_CPU_THREADS")
    if forced:
        usable = max(1, int(forced))
    return total, usable


def cpu_baseline(workload, seconds_target=10.0, also=()):
    """The reference's CPU path tracer (oracle restatement: branching estimator, recursive BVH walk, one
    generator per thread), timed on this box's host cores.  Two measurements: BASELINE configs[0] exactly
    (cornell-box2, 512x512, 16 spp, depth 5 -- the size the reference's own CPU timings are quoted on), and a bounded
    sample of the headline workload (64x64 pixels of each scene at its depth, where the branching estimator costs
    up to 3^depth rays per path)."""
    from oracle import pyoracle
    cores, threads = host_cpus()
    cfg1 = WORKLOADS["cornell2"]
    sc = pyoracle.OracleScene(scene_file(cfg1["scenes"][0]))
    w = cfg1["width"]
    _, _, s1 = sc.render_cpu_reference(w, w, cfg1["spp"], cfg1["depth"], threads)
    cfg1_rec = {"workload": cfg1["desc"], "value": w * w * cfg1["spp"] / s1 / 1e6, "unit": "Mpaths/s", "seconds": s1,
                "paths": w * w * cfg1["spp"], "sample": "the whole configuration, not a sample"}
    def sample(wl, seconds):
        paths, secs, w, spp = 0, 0.0, 64, 1
        for name in wl["scenes"]:
            sc = pyoracle.OracleScene(scene_file(name))
            per_scene = seconds / len(wl["scenes"])
            t_scene = 0.0
            n_scene = 0
            while t_scene < per_scene and spp <= 65536:
                _, _, s = sc.render_cpu_reference(w, w, spp, wl["depth"], threads)
                t_scene += s
                n_scene += w * w * spp
                if s < per_scene / 3:
                    spp *= 2
            paths += n_scene
            secs += t_scene
        return paths, secs

    paths, secs = sample(workload, seconds_target)
    rec = {"value": paths / secs / 1e6, "unit": "Mpaths/s", "cores": cores, "threads": threads, "kind": "port",
           "sample": "oracle restatement of the reference CPU path tracer (src/path_tracer.cu:27-238 on the thread pool of "
                     "src/RayTracer.cpp:657-694), 64x64 px of each scene of the headline workload at depth %d, %d paths in "
                     "%.1f s on %d threads (machine: %d CPUs)" % (workload["depth"], paths, secs, threads, cores),
           "cfg1": cfg1_rec}
    for name in also:  # the other workloads of the line, a shorter sample each
        p2, s2 = sample(WORKLOADS[name], seconds_target / 2)
        rec[name] = {"value": p2 / s2 / 1e6, "unit": "Mpaths/s",
                     "sample": "64x64 px of %s at depth %d, %d paths in %.1f s on %d threads; per-path cost does not depend on the image "
                               "size or the sample count, so the rate carries over to the full configuration" %
                               (" + ".join(WORKLOADS[name]["scenes"]), WORKLOADS[name]["depth"], p2, s2, threads)}
    return rec


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run with N ranks as a child process --
    this process has not touched a GPU and never will -- relay its output and exit with its status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


class Bench:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
        # CPT_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share GPUs)
        self.backend = os.environ.get("CPT_BENCH_BACKEND", "nccl")
        if self.backend != "nccl":
            local_rank %= torch.cuda.device_count()
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        self.ranks_seen = 1
        # CPT_BENCH_FORCE_DIST=1 (under torch.distributed.run with one rank): bring the process group up and send every
        # reduce through it anyway -- what a one-GPU box can rehearse of the RCCL path (communicator set-up, an fp64 SUM
        # reduce of a device tensor)
        self.distributed = self.world > 1 or (bool(os.environ.get("CPT_BENCH_FORCE_DIST")) and "WORLD_SIZE" in os.environ)
        if self.distributed:
            from cuda_path_tracer_amd.dist import reduce_radiance
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(self.backend)
            # communicator set-up (seconds for RCCL) must never land in a timed region, whatever --warmup says
            probe = torch.ones(1, dtype=torch.float64, device="cuda")
            reduce_radiance(probe, dst=0, single_rank_too=True)
            torch.cuda.synchronize()
            self.ranks_seen = dist.get_world_size()
            if self.rank == 0 and int(round(float(probe.item()))) != self.world:
                raise SystemExit("bench.py: warm-up reduce summed %r, expected %d ranks" % (probe.item(), self.world))
        self.verbose = bool(os.environ.get("CPT_BENCH_VERBOSE"))
        # a stream of our own: the legacy default stream serialises against every other stream of the process,
        # which would keep the renderer's second lane from overlapping the first
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)

    def say(self, msg):
        if self.verbose:
            print("[bench rank %d] %s" % (self.rank, msg), file=sys.stderr, flush=True)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, key, steps, warmup, with_roofline=True, options=(), tree=None):
        """Time `steps` steps of workload `key` after `warmup` untimed ones; returns the record (rank 0) or None.
        `options`: (key, value) pairs for cpt_set_option on top of the command line's; `tree`: "reference" or "fast"
        (the command line's choice when None)."""
        torch, dist, args = self.torch, self.dist, self.args
        from cuda_path_tracer_amd import MODE_MEGAKERNEL, MODE_WAVEFRONT, HostScene, Renderer
        from cuda_path_tracer_amd.dist import reduce_radiance, shard
        wl = WORKLOADS[key]
        world, rank = self.world, self.rank
        mode = MODE_WAVEFRONT if args.mode == "wavefront" else MODE_MEGAKERNEL
        renderers = []
        for name in wl["scenes"]:
            hs = HostScene(scene_file(name))
            r = Renderer(self.local_rank)
            r.set_stream(self.stream.cuda_stream)
            if args.batch_paths:
                r.set_option("batch_paths", args.batch_paths)
            if args.streams:
                r.set_option("streams", args.streams)
            for kv in args.opt:
                k, v = kv.split("=")
                r.set_option(k, int(v))
            for (k, v) in options:
                r.set_option(k, int(v))
            if args.device_bvh:
                r.build(hs)
            elif (tree or args.tree) == "fast":
                r.upload(hs.fast())
            else:
                r.upload(hs)
            w = wl["width"]
            h = hs.height_for_width(w)
            acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
            rgb = torch.zeros(h, w, 3, dtype=torch.uint8, device="cuda")
            renderers.append((r, hs, w, h, acc, rgb))
        spp = wl["spp"]
        scaling = wl.get("scaling", args.scaling)
        total_spp = spp * world if scaling == "weak" else spp
        extra = wl.get("render", {})
        phase = {"render": 0.0, "reduce": 0.0}

        def step(collective=True, clocked=False):
            """clocked: wait for the render and for the reduce and note how long each took on this rank (the timed steps
            never do: there the phases of consecutive scenes and steps overlap)."""
            for (r, hs, w, h, acc, rgb) in renderers:
                acc.zero_()
                if clocked:
                    torch.cuda.synchronize()
                    t_a = time.perf_counter()
                share = shard(rank, world, spp, w, h, scaling)
                if share is not None:
                    r.render_into(acc.data_ptr(), r.params(w, h, share["sample_begin"], share["sample_end"], wl["depth"], mode=mode,
                                                           tile_begin=share["tile_begin"], tile_count=share["tile_count"], **extra))
                if clocked:
                    torch.cuda.synchronize()
                    t_b = time.perf_counter()
                if collective:
                    reduce_radiance(acc, dst=0, single_rank_too=self.distributed)
                if clocked:
                    torch.cuda.synchronize()
                    phase["render"] += t_b - t_a
                    phase["reduce"] += time.perf_counter() - t_b
                if rank == 0:
                    r.tonemap_into(acc.data_ptr(), w, h, total_spp, rgb.data_ptr())

        self.say("%s: scenes uploaded" % key)
        for _ in range(warmup):
            step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        elapsed = time.perf_counter() - t0
        self.say("%s: timed steps done in %.3f s" % (key, elapsed))
        render_ms = reduce_ms = None
        if self.distributed:
            # one more step, outside the timed region, with a wait after each phase: the slowest rank's render time and
            # reduce time (the reduce includes waiting for the slowest rank's render to arrive)
            step(clocked=True)
            t = torch.tensor([elapsed, phase["render"], phase["reduce"]], dtype=torch.float64,
                             device="cuda" if self.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, render_ms, reduce_ms = float(t[0]), float(t[1]) * 1e3, float(t[2]) * 1e3
        paths_per_step = sum(w * h * total_spp for (_, _, w, h, _, _) in renderers)  # all ranks together
        rec = None
        if rank == 0:
            rec = {"workload": wl["desc"], "value": steps * paths_per_step / elapsed / 1e6, "unit": "Mpaths/s",
                   "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "paths_per_step": paths_per_step,
                   "scaling": scaling, "n_gpus": world}
            if self.distributed:
                rec.update({"render_ms": render_ms, "reduce_ms": reduce_ms, "rccl_ranks": self.ranks_seen,
                            "phase_note": "one extra step with a wait after each phase, max over ranks; reduce_ms includes "
                                          "waiting for the slowest rank"})
            if with_roofline:
                rec["roofline"] = self.roofline(key, renderers, step, mode)
        for (r, *_rest) in renderers:
            r.close()
        del renderers
        torch.cuda.empty_cache()
        if self.distributed:
            dist.barrier()  # the other ranks wait here while rank 0 runs its instrumented steps
        return rec

    def roofline(self, key, renderers, step, mode):
        """One step with per-launch HIP-event timing (one batch at a time, so a kernel's time is its own) and one with
        the device-side work counters, on rank 0 alone; from them the extend kernel's rates."""
        torch = self.torch
        from cuda_path_tracer_amd import MODE_WAVEFRONT
        for (r, *_rest) in renderers:
            r.reset_stats()
            r.set_instrumentation(time_kernels=True, count_work=False)
        step(collective=False)
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
        kernel_ms = {k: sum(t[k] for t in timed) for k in ("ms_generate", "ms_extend", "ms_shade", "ms_mega", "ms_total")}
        if mode != MODE_WAVEFRONT:
            ms, launches = kernel_ms["ms_mega"], sum(t["launches_mega"] for t in timed)
            gbps = pipeline_bytes(agg) / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "k_megakernel", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBPS, "traffic": None, "avg_launch_ms": ms / max(launches, 1),
                    "note": "one thread per pixel (the reference's kernel shape), kept as an A/B; algorithmic bytes of the whole path",
                    "kernel_ms_per_step": kernel_ms}
        ms = kernel_ms["ms_extend"]
        launches = sum(t["launches_extend"] for t in timed)
        secs = ms * 1e-3
        useful = F64_PER_NODE_STEP * agg["node_steps"] + F64_PER_TRIANGLE_TEST * agg["tri_tests"]
        occupied = 64.0 * (F64_PER_NODE_STEP * agg["box_passes"] + F64_PER_TRIANGLE_TEST * agg["leaf_passes"])
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(key, {}).get("k_extend_pw")
            except Exception:
                traffic = None
        kbytes = extend_bytes(agg)
        avg_ms = ms / max(launches, 1)
        hbm_measured = None
        if traffic:
            hbm_measured = {"bytes_per_launch": traffic, "GBps": traffic / (avg_ms * 1e-3) / 1e9,
                            "frac_of_hbm_peak": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "source": "profiles/traffic.json (rocprofv3 PMC passes, TCC_EA0 read/write requests, one batch at a time)"}
        return {
            # the resource the kernel runs out of: fp64 vector instruction issue.  achieved = fp64 lane-instructions of
            # lanes that had work (node expansions and triangle tests are fixed by the rays, the per-unit instruction
            # counts by the kernel's ISA) per second of extend-kernel time; peak = every SIMD issuing fp64 every cycle.
            "bound": "fp64_valu_issue", "kernel": "k_extend_pw", "achieved": useful / secs / 1e12,
            "peak": FP64_ISSUE_PEAK / 1e12, "unit": "T fp64 lane-instructions/s", "frac": useful / secs / FP64_ISSUE_PEAK,
            # the byte yardstick next to it: algorithmic record bytes per second over the HBM peak.  Above 1 means the
            # records cannot be coming from HBM, and they are not: the kernel serves them from its LDS copy of the scene
            # (and L2), which is why the byte rate is reported but is not this kernel's bound
            "hbm_algorithmic_ratio_to_hbm_peak": kbytes / secs / 1e9 / HBM_PEAK_GBPS,
            "hbm_algorithmic_note": "104 B per node step + 100 B per triangle test + 64 B per segment, per second, over 8 TB/s; "
                                    "> 1 => served from LDS/L2, not HBM",
            "traffic": traffic, "avg_launch_ms": avg_ms, "launches_per_step": launches,
            "instr_per_node_step": F64_PER_NODE_STEP, "instr_per_triangle_test": F64_PER_TRIANGLE_TEST,
            "instr_constants": "frozen algorithmic constants (minimal fp64 op list in bench.py), first counted in "
                               "profiles/r02_isa_fp64_counts.txt (tools/count_fp64_isa.py); not re-counted per build",
            # what the kernel issues for those units today, for the record: fewer than the frozen algorithmic counts, so
            # `frac` is algorithmic work per peak issue slot, not the share of issue slots that were filled
            "instr_executed": {"node_step": {"default": 66, "scene_resident_in_lds": 54, "first_rays_of_a_pinhole_camera": "12 fewer"},
                               "triangle_test": {"default": 89, "first_rays_of_a_pinhole_camera": 81},
                               "note": "counted in the ISA of this build's kernels (make -C cuda_path_tracer_amd asm); the "
                                       "numerator of `achieved` keeps the frozen 70 and 89"},
            "occupied_frac": occupied / secs / FP64_ISSUE_PEAK,
            "lane_slots_used": {"box": agg["node_steps"] / max(64 * agg["box_passes"], 1),
                                "triangle": agg["tri_tests"] / max(64 * agg["leaf_passes"], 1)},
            # secondary yardsticks, neither of them this kernel's bound: the algorithmic-byte rate (every node and
            # triangle record a ray touches, served from the per-CU LDS copy and L2, so it may exceed the HBM peak) and
            # the HBM traffic the counters measured
            "hbm_algorithmic": {"bytes_per_launch": kbytes / max(launches, 1), "GBps": kbytes / secs / 1e9,
                                "ratio_to_hbm_peak": kbytes / secs / 1e9 / HBM_PEAK_GBPS,
                                "note": "records are served from LDS/L2, not HBM: a yardstick across implementations, not a bound"},
            "hbm_measured": hbm_measured,
            "exact_box_fallbacks": agg["exact_node_steps"],
            "pipeline_algorithmic_GBps": pipeline_bytes(agg) / (kernel_ms["ms_total"] * 1e-3) / 1e9,
            "per_segment": {"node_steps": agg["node_steps"] / max(agg["segments"], 1),
                            "tri_tests": agg["tri_tests"] / max(agg["segments"], 1),
                            "segments_per_path": agg["segments"] / max(agg["paths"], 1)},
            "kernel_ms_per_step": kernel_ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS), help="the headline workload (`value`)")
    ap.add_argument("--extra", default=None,
                    help="comma-separated workloads also measured and reported under `workloads`; 'none' for none.  Default: "
                         "dragon,teapot_dof (2 steps each) on one GPU; with N > 1 the full-node shapes of BASELINE configs[3] and "
                         "[4], " + MULTI_GPU_EXTRAS + " (fixed totals dealt over the N GPUs, 1 step each)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank renders the workload's sample count (new sample indices per rank); strong: the "
                         "workload's samples are dealt out over the ranks")
    ap.add_argument("--mode", default="wavefront", choices=["wavefront", "mega"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-modes", action="store_true", help="skip the extra measurements of extend variants 3 and 4")
    ap.add_argument("--batch-paths", type=int, default=1 << 28,
                    help="path slots per batch; the library's own default (2^25) suits short-lived processes, a long-lived one "
                         "can afford the larger allocation and gets shorter launch tails (0: library default)")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight at once (library default when 0)")
    ap.add_argument("--device-bvh", action="store_true",
                    help="have the GPU build the BVH (a Morton-order tree) instead of uploading the reference's tree; "
                         "same image up to ties, different visit sets, so not the parity configuration")
    ap.add_argument("--tree", default="reference", choices=["reference", "fast"],
                    help="reference: the reference's BVH, which makes every ray test exactly the boxes the reference tests (the parity "
                         "configuration); fast: the host library's surface-area tree over the same triangles with the reference's "
                         "tie-break ranks -- same hits, same image bit for bit, about a third fewer box tests")
    ap.add_argument("--opt", action="append", default=[], help="key=value passed to cpt_set_option (tuning experiments)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.extra is None:
        args.extra = "dragon,teapot_dof" if args.gpus == 1 else MULTI_GPU_EXTRAS
    names = [] if args.extra in ("", "none") else [n for n in args.extra.split(",") if n and n != args.workload]
    for name in names:
        if name not in WORKLOADS:
            raise SystemExit("bench.py: unknown workload '%s' in --extra" % name)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    b = Bench(args)
    wl = WORKLOADS[args.workload]
    head = b.measure(args.workload, args.steps, args.warmup)
    extras = {}
    for name in names:
        heavy = WORKLOADS[name].get("scaling") == "strong"  # the full-node shapes: tens of seconds per step on two GPUs
        extras[name] = b.measure(name, 1 if heavy else 2, 1, with_roofline=not heavy)  # their kernels are the headline's and the dragon's
    # The fast modes, beside the numbers above and never instead of them: the same closest hits (so the same image,
    # bit for bit) from a walk that is not the reference's.  3: inner nodes through a conservative binary32 box test,
    # every leaf through the exact test of its own box; 4: the same, nearer child first, boxes beyond the closest hit
    # so far skipped.
    fast = {}
    if not args.no_fast_modes and args.mode == "wavefront" and b.world == 1:
        for name in [args.workload] + names:
            rec = {}
            for label, variant, tree in (("binary32_inner_nodes", 3, None), ("binary32_inner_nodes_and_culling", 4, None),
                                         ("surface_area_tree", 1, "fast"), ("surface_area_tree_and_culling", 2, "fast"),
                                         ("surface_area_tree_binary32_inner_nodes_and_culling", 4, "fast")):
                r = b.measure(name, 2, 1, with_roofline=False, options=(("extend_variant", variant),), tree=tree)
                if r is not None:
                    rec[label] = {"value": r["value"], "unit": "Mpaths/s", "ms_per_step": r["ms_per_step"]}
            if b.rank == 0:
                base = head["value"] if name == args.workload else extras[name]["value"]
                for label in rec:
                    rec[label]["speedup_over_reference_walk"] = rec[label]["value"] / base
                fast[name] = rec
    if b.rank == 0:
        world = b.world
        out = {"metric": "Mpaths/s", "value": head["value"], "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
               "scaling": wl.get("scaling", args.scaling), "vs_baseline": None, "dtype": "f64",
               "data": "the reference's bundled scenes (tests/golden/scenes/*.ray: " + ", ".join(
                   sorted({n for k in [args.workload] + names for n in WORKLOADS[k]["scenes"]})) + "), rendered, not synthetic",
               "config": {"workload": wl["desc"], "mode": args.mode, "batch_paths": args.batch_paths,
                          "bvh": "built on the GPU (Morton order)" if args.device_bvh else
                                 ("surface-area tree (host build), reference tie-break ranks" if args.tree == "fast" else "reference tree (host build)"),
                          "paths_per_step": head["paths_per_step"],
                          "sharding": ("spp range per rank (%s scaling), fp64 radiance reduce to rank 0" % wl.get("scaling", args.scaling))
                                      if world > 1 else "single GPU",
                          "rccl_ranks": b.ranks_seen, "backend": b.backend if b.distributed else None},
               "roofline": head["roofline"], "workloads": extras, "fast_modes": fast}
        if b.distributed:
            out["render_ms"], out["reduce_ms"] = head["render_ms"], head["reduce_ms"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, also=[n for n in names if n == "dragon"])
        print(json.dumps(out), flush=True)
    if b.distributed:
        b.dist.barrier()
        b.dist.destroy_process_group()


if __name__ == "__main__":
    main()
