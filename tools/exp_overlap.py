#!/usr/bin/env python3
"""Experiment: do two renders on two HIP streams overlap (extend of one under shade of the other)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cuda_path_tracer_amd import HostScene, Renderer
name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box2-transmissive"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
hs = HostScene(os.path.join(ROOT, "tests", "golden", "scenes", name + ".ray"))
h = hs.height_for_width(w)
rs = []
for i in range(2):
    r = Renderer(0); r.upload(hs)
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    rs.append((r, acc))
def run(pairs, reps):
    for r, acc in rs: r.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for (r, acc), (s0, s1) in pairs:
            r.render_into(acc.data_ptr(), r.params(w, h, s0, s1, 8))
    for r, acc in rs: r.synchronize()
    return (time.perf_counter() - t0) / reps
for trial in range(2):
    t_seq = run([(rs[0], (0, spp)), (rs[0], (spp, 2 * spp))], 3)
    t_par = run([(rs[0], (0, spp)), (rs[1], (spp, 2 * spp))], 3)
    print("%s %d spp x2: one stream %.2f ms, two streams %.2f ms (%.1f%% faster)" % (name, spp, t_seq * 1e3, t_par * 1e3, 100 * (t_seq / t_par - 1)), flush=True)
