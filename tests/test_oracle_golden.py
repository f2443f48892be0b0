"""CPU tests of the oracle (test infrastructure): known answers, scene statistics, the committed
golden vectors, and the reference's golden renders' mean luminance."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, scene_path

GOLD = os.path.join(ROOT, "tests", "golden")


def test_philox_known_answers(oracle):
    # Philox-4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(x) for x in oracle.philox(ctr, key)] == want
    assert oracle.u01(0, 0) == 2.0 ** -53                   # never 0
    assert oracle.u01(0xffffffff, 0xffffffff) == 1.0        # 1 included: the range is (0, 1]


@pytest.mark.parametrize("name,tris,nodes", [("cornell-box2", 36, 71), ("cornell-box2-specular", 36, 71),
                                              ("cornell-box2-transmissive", 36, 71), ("teapot-trans", 201, 401),
                                              ("dragon-spec", 10005, 20009), ("dragon-trans", 10005, 20009)])
def test_scene_statistics(name, tris, nodes, oracle_scenes):
    """Triangle / node counts of the bundled scenes; the dragon's light has a duplicated point, so one
    of its faces is degenerate and dropped (reference: src/SceneObjects/trimesh.cpp:50-53)."""
    sc = oracle_scenes(name)
    assert (sc.n_prims, sc.n_nodes) == (tris, nodes)
    boxes, topo = sc.nodes()
    leaves = topo[:, 2] >= 0
    assert leaves.sum() == tris and sorted(topo[leaves, 2]) == list(range(tris))
    # every child box lies inside its parent's box
    for k in np.nonzero(~leaves)[0][:2000]:
        for c in topo[k, :2]:
            assert (boxes[c, :3] >= boxes[k, :3]).all() and (boxes[c, 3:] <= boxes[k, 3:]).all()


@pytest.mark.parametrize("name", ["cornell-box2", "cornell-box2-transmissive", "teapot-trans", "dragon-spec"])
def test_primary_ray_golden_vectors(name, oracle_scenes):
    sc = oracle_scenes(name)
    g = np.load(os.path.join(GOLD, "primary_%s_64.npz" % name))
    o, d = sc.camera_rays(64, 64)
    assert np.array_equal(d.view(np.uint64), g["dir_bits"])
    r = sc.trace(o, d)
    assert np.array_equal(r["prim"], g["prim"])
    assert np.array_equal(r["t"].view(np.uint64), g["t_bits"])
    assert np.array_equal(r["vhash"], g["vhash"])
    assert np.array_equal(r["node_steps"], g["node_steps"]) and np.array_equal(r["tri_tests"], g["tri_tests"])


def test_primary_ray_digests_512(oracle_scenes):
    want = json.load(open(os.path.join(GOLD, "primary_digests.json")))["cornell-box2-transmissive"]
    sc = oracle_scenes("cornell-box2-transmissive")
    o, d = sc.camera_rays(512, 512)
    r = sc.trace(o, d)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert sha(d) == want["dir"] and sha(r["prim"]) == want["prim"] and sha(r["t"]) == want["t"]
    assert sha(r["vhash"]) == want["vhash"] and int(r["node_steps"].sum()) == want["node_steps"]


def test_visit_bitmap_consistent_with_hash_and_counts(oracle, oracle_scenes):
    sc = oracle_scenes("teapot-trans")
    o, d = sc.camera_rays(24, 24)
    r = sc.trace(o, d, bitmap=True)
    bits = np.unpackbits(r["bitmap"], axis=1, bitorder="little")[:, :sc.n_nodes]
    assert np.array_equal(bits.sum(1), r["box_tests"])
    assert not bits[:, 0].any()  # the root's own box is never tested (reference: src/gpu/GPUBVH.cu:46-53)
    _, topo = sc.nodes()
    for k in range(len(o)):
        visited = np.nonzero(bits[k])[0]
        parents = {int(c): int(p) for p in range(sc.n_nodes) for c in topo[p, :2] if c >= 0}
        for v in visited:  # a node is tested iff its parent was expanded, i.e. is the root or was itself hit
            p = parents[int(v)]
            assert p == 0 or bits[k, p]
        if k > 40:
            break


def test_tie_break_prefers_first_tested_leaf(oracle, tmp_path):
    """Two coincident triangles: the reference keeps the hit it found first (strict '<',
    reference: src/gpu/GPUBVH.cu:68,77)."""
    text = """SBT-raytracer 1.0
    camera { position = (0,0,3); }
    polymesh { material = { emissive = (1,0,0); }; points = ((-1,-1,0),(1,-1,0),(0,1,0)); faces = ((0,1,2)); }
    polymesh { material = { emissive = (0,1,0); }; points = ((-1,-1,0),(1,-1,0),(0,1,0)); faces = ((0,1,2)); }"""
    p = tmp_path / "tie.ray"
    p.write_text(text)
    sc = oracle.OracleScene(str(p))
    o, d = sc.camera_rays(9, 9)
    r = sc.trace(o, d)
    _, topo = sc.nodes()
    first_leaf = topo[topo[0, 0], 2]  # left child of the root is tested first
    assert set(r["prim"][r["prim"] >= 0]) == {first_leaf}


def test_gpu_estimator_against_reference_golden_luminance(oracle, oracle_scenes):
    """The oracle's GPU-semantics estimator reproduces the mean luminance of the reference's own
    renders of cornell-box at depth 3 (69.86 at 256 spp; clamping makes the mean depend on spp)."""
    lum = json.load(open(os.path.join(GOLD, "luminance.json")))["cornell-box"]["3"]
    want = [v for (est, spp, v) in lum if spp == 256][0]
    sc = oracle_scenes("cornell-box")
    acc, stats = sc.render(40, 40, 0, 256, 3)
    img = oracle.tonemap(acc, 256).astype(np.float64)
    got = (0.2126 * img[..., 0] + 0.7152 * img[..., 1] + 0.0722 * img[..., 2]).mean()
    assert abs(got - want) < 2.0, got
    assert stats["paths"] == 40 * 40 * 256 and stats["segments"] <= 4 * stats["paths"]


def test_cpu_reference_estimator_agrees_with_gpu_estimator(oracle, oracle_scenes):
    """Same expectation: the branching CPU estimator and the lobe-sampling GPU estimator converge to the
    same image (compared on unclamped pixel means, away from the light)."""
    sc = oracle_scenes("cornell-box2")
    w = h = 24
    _, mean_cpu, _ = sc.render_cpu_reference(w, h, 192, 3, threads=8)
    acc, _ = sc.render(w, h, 0, 768, 3)
    mean_gpu = acc / 768
    a, b = mean_cpu.mean(), mean_gpu.mean()
    assert abs(a - b) / b < 0.05, (a, b)


def test_depth_of_field_and_roulette_are_unbiased_variants(oracle_scenes):
    sc = oracle_scenes("cornell-box2")
    base, _ = sc.render(16, 16, 0, 384, 4)
    rr, st = sc.render(16, 16, 0, 384, 4, rr=1)
    assert abs(rr.mean() - base.mean()) / base.mean() < 0.15
    dof, _ = sc.render(16, 16, 0, 8, 2, dof=1, focal=6.0, aperture=0.2)
    assert np.isfinite(dof).all() and dof.sum() > 0


@pytest.mark.parametrize("scene,depth", [("cornell-box", 3), ("cornell-box2-specular", 3), ("cornell-box2-transmissive", 3)])
def test_oracle_estimator_against_reference_renders_blockwise(scene, depth, oracle_scenes):
    """The oracle itself (not the GPU product) against the reference's own golden renders, one scene per lobe:
    all-diffuse, with a mirror lobe, with a dielectric lobe.

    tests/golden/reference_blocks.npz holds the mean RGB of every 8x8-pixel block of the reference's 512x512
    renders.  Rendering whole images at the golden files' sample counts is beyond a CPU test, so the oracle renders
    64 of the 4096 blocks -- a regular 8x8 lattice over the image -- at the reference's resolution and pixel
    semantics (jittered samples of the 512x512 pixel, mean, clamp, truncation to 8 bits, then the block mean),
    1024 samples per pixel in two halves.  The halves give the noise sigma of a block mean; the reference's block
    must be no further from the oracle's than 1.5 sigma + 0.25 in RMS over the lattice (0-255 scale, the two
    worst blocks set aside as in the GPU test: a light's outline falls a fraction of a pixel differently in the
    golden files), for the reference's GPU render and for its CPU render."""
    ref = np.load(os.path.join(GOLD, "reference_blocks.npz"))
    sc = oracle_scenes(scene)
    half = 512
    lattice = [(8 * gx + 3, 8 * gy + 4) for gy in range(8) for gx in range(8)]
    px = np.array([(8 * bx + i, 8 * by + j) for (bx, by) in lattice for j in range(8) for i in range(8)], np.int32)
    sums = []
    for s0 in (0, half):
        ijs = np.empty((px.shape[0] * half, 3), np.int32)
        ijs[:, 0] = np.repeat(px[:, 0], half)
        ijs[:, 1] = np.repeat(px[:, 1], half)
        ijs[:, 2] = np.tile(np.arange(s0, s0 + half, dtype=np.int32), px.shape[0])
        sums.append(sc.paths(512, 512, depth, ijs).reshape(px.shape[0], half, 3).sum(axis=1))

    def block_means(acc, n):  # (64 blocks, 3): the reference's quantiser per pixel (src/path_tracer.cu:436-443), then the mean
        return np.floor(255.0 * np.clip(acc / n, 0.0, 1.0)).reshape(64, 64, 3).mean(axis=1)

    def rms(a, b, drop=0):
        sq = ((a - b) ** 2).mean(axis=1)
        sq.sort()
        return float(np.sqrt(sq[:sq.size - drop].mean()))
    both = block_means(sums[0] + sums[1], 2 * half)
    sigma = rms(block_means(sums[0], half), block_means(sums[1], half)) / 2.0
    for est in ("gpu", "cpu"):
        blocks = ref["%s|%d|%s" % (scene, depth, est)].astype(np.float64)
        want = np.array([blocks[by, bx] for (bx, by) in lattice])
        got = rms(both, want, drop=2)
        assert got < 1.5 * sigma + 0.25, (est, got, sigma)
        assert abs(both.mean() - want.mean()) < 1.5, (est, both.mean(), want.mean())
