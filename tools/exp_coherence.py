#!/usr/bin/env python3
"""Experiment: how much does ordering incoherent rays by direction help the per-lane traversal kernel?
Builds first-bounce diffuse rays for a scene (oracle for the primary hits), then times cpt_trace_rays'
kernel for several orderings of the same rays (use under rocprofv3 --kernel-trace for kernel time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cuda_path_tracer_amd import HostScene, Renderer
from oracle import pyoracle as po

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell-box2"
path = os.path.join(ROOT, "tests", "golden", "scenes", scene + ".ray")
orc = po.OracleScene(path)
W = 384
o, d = orc.camera_rays(W, W)
t0 = time.time(); pr = orc.trace(o, d); print("oracle primary trace %.1fs" % (time.time() - t0), flush=True)
hit = pr["prim"] >= 0
P = o + d * pr["t"][:, None]
N = pr["normal"]
rng = np.random.default_rng(1)
reps = 12
O = []; D = []
for r in range(reps):
    u1 = rng.random(len(P)); u2 = rng.random(len(P))
    r1 = 2 * np.pi * u1; r2s = np.sqrt(u2)
    w = N
    a = np.where(np.abs(w[:, :1]) > 0.1, np.array([[0., 1, 0]]), np.array([[1., 0, 0]]))
    uu = np.cross(a, w); uu /= np.linalg.norm(uu, axis=1, keepdims=True)
    vv = np.cross(w, uu)
    dd = uu * (np.cos(r1) * r2s)[:, None] + vv * (np.sin(r1) * r2s)[:, None] + w * np.sqrt(1 - u2)[:, None]
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    O.append((P + N * 1e-12)[hit]); D.append(dd[hit])
O = np.concatenate(O); D = np.concatenate(D)
print("rays", len(O), flush=True)
hs = HostScene(path); g = Renderer(0); g.upload(hs)

def run(name, order):
    oo = np.ascontiguousarray(O[order]); dd = np.ascontiguousarray(D[order])
    g.trace_rays(oo[:1000], dd[:1000])
    t0 = time.time(); res = g.trace_rays(oo, dd); dt = time.time() - t0
    print("%-28s wall %.1f ms  mean steps %.1f" % (name, dt * 1e3, res["node_steps"].mean()), flush=True)

n = len(O)
def dirbin(D, k):
    # cube-map face (6) x k x k cells
    ax = np.argmax(np.abs(D), axis=1)
    sgn = (D[np.arange(len(D)), ax] > 0).astype(np.int64)
    face = ax * 2 + sgn
    oth = np.stack([D[np.arange(len(D)), (ax + 1) % 3], D[np.arange(len(D)), (ax + 2) % 3]], 1) / np.abs(D[np.arange(len(D)), ax])[:, None]
    ij = np.clip(((oth + 1) * 0.5 * k).astype(np.int64), 0, k - 1)
    return face * k * k + ij[:, 0] * k + ij[:, 1]

run("arrival order (pixel)", np.arange(n))
run("random shuffle", rng.permutation(n))
run("octant bins (8)", np.argsort(((D > 0) * np.array([1, 2, 4])).sum(1), kind="stable"))
for k in (2, 4, 8, 16):
    run("cube %dx%d bins (%d)" % (k, k, 6 * k * k), np.argsort(dirbin(D, k), kind="stable"))
# primary rays themselves for reference
OO = np.tile(o, (reps, 1)); DD = np.tile(d, (reps, 1))
O, D = OO, DD
run("primary rays (coherent)", np.arange(len(O)))
