"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars: bit-exact for everything discrete or made of +,-,*,/,sqrt (camera rays, hit triangle, hit t,
visited-node sets, step counts); 1e-9 relative for radiance, whose only inexact ingredients are the
device's sin/cos/pow (a few ulp from glibc's) -- paths where such an ulp flips a discrete choice are
counted and must stay below 1 in 10^4.
"""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

SCENES = ["cornell-box2", "cornell-box2-specular", "cornell-box2-transmissive", "teapot-trans", "dragon-spec",
          "dragon-trans", "3-dragon-spec"]


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_device_sqrt_and_division_are_correctly_rounded(gpu):
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.uniform(0, 4, 200000), np.exp(rng.uniform(-300, 300, 200000)),
                        np.array([0.0, 1.0, 2.0, 4.0, 1e-310, 5e-324, 1e308])])
    b = np.concatenate([rng.uniform(-3, 3, 200000), np.exp(rng.uniform(-30, 30, 200000)) * rng.choice([-1, 1], 200000),
                        np.array([3.0, -7.0, 1e-300, 1e300, 2.0, 2.0, 0.5])])
    s, q = gpu.probe_math(a, b)
    assert np.array_equal(bits(s), bits(np.sqrt(a)))
    with np.errstate(over="ignore", under="ignore"):
        assert np.array_equal(bits(q), bits(a / b))


def test_device_philox_known_answers(gpu):
    # known-answer vectors of Philox-4x32-10 from the Random123 distribution (kat_vectors)
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        out, u = gpu.probe_rng(ctr, key)
        assert [int(x) for x in out] == want
        assert 0.0 < u[0] <= 1.0 and 0.0 < u[1] <= 1.0


@pytest.mark.parametrize("name", SCENES)
def test_camera_rays_bit_exact(name, gpu_scene, oracle_scenes):
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    for (w, h, jx, jy) in [(64, 64, 0.5, 0.5), (37, 23, 0.25, 1.0)]:
        o, d = g.camera_rays(w, h, jx, jy)
        oo, od = orc.camera_rays(w, h, jx, jy)
        assert np.array_equal(bits(o), bits(oo))
        assert np.array_equal(bits(d), bits(od))


def _compare_trace(g, orc, o, d, bitmap):
    a = g.trace_rays(o, d, bitmap=bitmap)
    b = orc.trace(o, d, bitmap=bitmap)
    assert np.array_equal(a["prim"], b["prim"]), "hit triangle ids differ"
    assert np.array_equal(bits(a["t"]), bits(b["t"])), "hit distances differ"
    assert np.array_equal(a["vhash"], b["vhash"]), "visited-node sets differ"
    for k in ("node_steps", "box_tests", "tri_tests", "max_stack"):
        assert np.array_equal(a[k], b[k]), k
    hit = b["prim"] >= 0
    assert np.array_equal(bits(a["normal"][hit]), bits(b["normal"][hit])), "shading normals differ"
    if bitmap:
        assert np.array_equal(a["bitmap"], b["bitmap"]), "visit bitmaps differ"
    return b


@pytest.mark.parametrize("name", SCENES)
def test_primary_rays_hit_ids_and_visit_sets_bit_exact(name, gpu_scene, oracle_scenes):
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    o, d = orc.camera_rays(64, 64)
    b = _compare_trace(g, orc, o, d, bitmap=True)
    assert (b["prim"] >= 0).mean() > 0.5


@pytest.mark.parametrize("name", ["cornell-box2-transmissive", "dragon-trans", "3-dragon-spec"])
def test_incoherent_rays_bit_exact(name, gpu_scene, oracle_scenes):
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    rng = np.random.default_rng(11)
    n = 20000
    boxes, _ = orc.nodes()
    lo, hi = boxes[0, :3], boxes[0, 3:]
    o = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # axis-parallel and grazing rays: zero direction components exercise the slab test's skip
    d[:200, 0] = 0.0
    d[200:400, 1] = 0.0
    d[400:500, :2] = 0.0
    d[400:500, 2] = 1.0
    _compare_trace(g, orc, o, d, bitmap=False)


@pytest.mark.parametrize("name,depth", [("3-dragon-spec", 8), ("cornell-box2", 5), ("cornell-box2-specular", 8),
                                         ("cornell-box2-transmissive", 8), ("teapot-trans", 8), ("dragon-trans", 5)])
def test_path_radiance_matches_oracle(name, depth, gpu_scene, oracle_scenes):
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    w = h = 48
    rng = np.random.default_rng(3)
    n = 6000
    ijs = np.stack([rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 64, n)], 1).astype(np.int32)
    p = g.params(w, h, 0, 64, depth)
    a = g.path_radiance(p, ijs)
    b = orc.paths(w, h, depth, ijs)
    close = np.isclose(a, b, rtol=1e-9, atol=1e-12).all(axis=1)
    assert (~close).sum() <= max(1, n // 10000), "%d of %d paths differ" % ((~close).sum(), n)
    assert np.abs(b).sum() > 0


@pytest.mark.parametrize("kt", ["(0.0,0.9,0.9)", "(-0.5,0.9,0.0)"])
def test_transmission_through_a_medium_with_zero_or_negative_kt(kt, gpu, oracle, tmp_path):
    """kt^t inside a medium (reference: glm::pow at src/path_tracer.cu:326) is computed without pow: a zero kt must
    still give 0, a negative one pow's NaN (t is never an integer here), channel by channel, as the oracle's pow does."""
    from cuda_path_tracer_amd import HostScene
    text = open(scene_path("cornell-box2-transmissive")).read()
    assert "transmissive=(0.9,0.9,0.9)" in text
    path = tmp_path / "odd-kt.ray"
    path.write_text(text.replace("transmissive=(0.9,0.9,0.9)", "transmissive=" + kt))
    gpu.upload(HostScene(str(path)))
    orc = oracle.OracleScene(str(path))
    w = h = 48
    rng = np.random.default_rng(5)
    n = 6000
    ijs = np.stack([rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 64, n)], 1).astype(np.int32)
    a = gpu.path_radiance(gpu.params(w, h, 0, 64, 8), ijs)
    b = orc.paths(w, h, 8, ijs)
    assert (np.isnan(a) == np.isnan(b)).all()
    close = np.isclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True).all(axis=1)
    assert (~close).sum() <= 1, "%d of %d paths differ" % ((~close).sum(), n)
    if "-" in kt:
        assert np.isnan(b[:, 0]).any() and not np.isnan(b[:, 1]).all()
    else:
        assert not np.isnan(b).any() and np.abs(b).sum() > 0


@pytest.mark.parametrize("name,depth,dof,rr", [("cornell-box2", 5, 0, 0), ("cornell-box2-transmissive", 6, 0, 0),
                                                ("teapot-trans", 4, 1, 0), ("cornell-box2-specular", 6, 0, 1)])
def test_wavefront_render_matches_oracle(name, depth, dof, rr, gpu_scene, oracle_scenes):
    import torch
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    w, h, spp = 40, 28, 6
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    p = g.params(w, h, 0, spp, depth, dof=dof, focal_distance=8.1, aperture=0.3, russian_roulette=rr)
    g.render_into(acc.data_ptr(), p)
    g.synchronize()
    got = acc.cpu().numpy()
    want, _ = orc.render(w, h, 0, spp, depth, dof=dof, focal=8.1, aperture=0.3, rr=rr)
    close = np.isclose(got, want, rtol=1e-9, atol=1e-12).all(axis=2)
    assert (~close).sum() <= 1, "%d of %d pixels differ" % ((~close).sum(), w * h)
    assert want.sum() > 0


def test_wavefront_and_one_thread_per_pixel_agree_bit_for_bit(gpu_scene):
    import torch
    g = gpu_scene("cornell-box2-transmissive")
    w, h, spp, depth = 72, 50, 5, 6
    out = []
    for mode in (0, 1):
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth, mode=mode))
        g.synchronize()
        out.append(acc.cpu().numpy())
    assert np.array_equal(bits(out[0]), bits(out[1]))


def test_sample_ranges_add_up_and_batching_does_not_matter(gpu_scene):
    import torch
    g = gpu_scene("cornell-box2")
    w, h, depth = 64, 64, 4
    whole = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.render_into(whole.data_ptr(), g.params(w, h, 0, 12, depth))
    parts = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.set_option("batch_paths", 64 * 64 * 2)  # two samples per batch instead of all twelve
    try:
        g.render_into(parts.data_ptr(), g.params(w, h, 0, 5, depth))
        g.render_into(parts.data_ptr(), g.params(w, h, 5, 12, depth))
    finally:
        g.set_option("batch_paths", 1 << 25)
    g.synchronize()
    assert np.array_equal(bits(whole.cpu().numpy()), bits(parts.cpu().numpy()))


@pytest.mark.parametrize("name,width,first,spp,dof", [
    ("cornell-box2-specular", 512, 0, 256, False), ("dragon-spec", 1024, 0, 1024, False), ("teapot-trans", 1024, 0, 4096, True),
    # the full-node configuration's one-GPU shares: dragon at 2048x2048, the 1024 samples rank 3 of 8 renders of the
    # 8192; cornell-box2 at 512x512 with all 16 384 samples
    ("dragon-spec", 2048, 3 * 1024, 1024, False), ("cornell-box2", 512, 0, 16384, False)])
def test_full_size_renders_split_anywhere_add_up(name, width, first, spp, dof, gpu_scene):
    """The benchmark configurations at their full sizes, through a property that needs no oracle: a render in
    one call and the same sample range split at an odd index (and, within each part, into whatever batches and
    streams the library picks for that size) hold the same radiance sums bit for bit; the image is finite and
    not empty; with the thin lens on, it is not the pin-hole image."""
    import torch
    g = gpu_scene(name)
    depth = 8
    h = width
    p = dict(dof=dof, focal_distance=8.1, aperture=0.6)
    whole = torch.zeros(h, width, 3, dtype=torch.float64, device="cuda")
    g.render_into(whole.data_ptr(), g.params(width, h, first, first + spp, depth, **p))
    parts = torch.zeros(h, width, 3, dtype=torch.float64, device="cuda")
    cut = first + spp // 3 + 1
    g.render_into(parts.data_ptr(), g.params(width, h, first, cut, depth, **p))
    g.render_into(parts.data_ptr(), g.params(width, h, cut, first + spp, depth, **p))
    g.synchronize()
    assert torch.equal(whole.view(torch.int64), parts.view(torch.int64))
    assert bool(torch.isfinite(whole).all()) and float(whole.sum()) > 0.0
    if dof:
        pin = torch.zeros(h, width, 3, dtype=torch.float64, device="cuda")
        g.render_into(pin.data_ptr(), g.params(width, h, 0, 16, depth))
        g.synchronize()
        assert float((pin / 16 - whole / spp).abs().mean()) > 1e-3
    if first > 0:  # a rank's share is not the first rank's: other sample indices, other paths, same expectation
        base = torch.zeros(h, width, 3, dtype=torch.float64, device="cuda")
        g.render_into(base.data_ptr(), g.params(width, h, 0, 64, depth))
        g.synchronize()
        assert not torch.equal(base, whole)
        assert abs(float(whole.mean()) / spp - float(base.mean()) / 64) < 0.05 * float(base.mean()) / 64


def test_tile_windows_add_up_to_the_full_render(gpu_scene):
    """Rendering disjoint tile windows into one buffer gives the full render bit for bit (both kernels shapes,
    ragged image sides); a window leaves the pixels outside it alone; bad windows are refused."""
    import torch
    from cuda_path_tracer_amd import CptError
    from cuda_path_tracer_amd.dist import tile_range
    g = gpu_scene("cornell-box2-transmissive")
    w, h, spp, depth = 77, 43, 5, 5
    n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
    for mode in (0, 1):
        whole = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        g.render_into(whole.data_ptr(), g.params(w, h, 0, spp, depth, mode=mode))
        parts = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        for r in range(3):
            b, c = tile_range(r, 3, w, h)
            g.render_into(parts.data_ptr(), g.params(w, h, 0, spp, depth, mode=mode, tile_begin=b, tile_count=c))
        g.synchronize()
        assert torch.equal(whole.view(torch.int64), parts.view(torch.int64))
    one = torch.full((h, w, 3), 7.0, dtype=torch.float64, device="cuda")
    g.render_into(one.data_ptr(), g.params(w, h, 0, spp, depth, tile_begin=11, tile_count=1))  # tile (1, 1): pixels 8..15 x 8..15
    g.synchronize()
    touched = (one != 7.0).any(dim=2).cpu().numpy()
    assert touched[8:16, 8:16].any() and not touched[:8].any() and not touched[16:].any() and not touched[:, :8].any() and not touched[:, 16:].any()
    g.reset_stats()
    g.set_instrumentation(False, True)
    try:
        g.render_into(one.data_ptr(), g.params(w, h, 0, 2, depth, tile_begin=n_tiles - 1, tile_count=1))  # the ragged corner tile
        g.synchronize()
        assert g.stats()["paths"] == 2 * (w - 72) * (h - 40)
    finally:
        g.set_instrumentation(False, False)
    for (b, c) in [(0, n_tiles + 1), (n_tiles, 1), (-1, 2), (3, 0), (0, -1)]:
        with pytest.raises(CptError):
            g.render_into(one.data_ptr(), g.params(w, h, 0, 1, depth, tile_begin=b, tile_count=c))


def test_tonemap_and_render_image(gpu_scene, oracle):
    import torch
    g = gpu_scene("cornell-box2")
    w, h, spp, depth = 50, 50, 8, 3
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth))
    rgb = torch.zeros(h, w, 3, dtype=torch.uint8, device="cuda")
    g.tonemap_into(acc.data_ptr(), w, h, spp, rgb.data_ptr())
    g.synchronize()
    want = oracle.tonemap(acc.cpu().numpy(), spp)
    assert np.array_equal(rgb.cpu().numpy(), want)
    img = g.render_image(g.params(w, h, 0, spp, depth))
    assert np.array_equal(img, want)
    assert img.max() > 0


def _tie_prone_rays(orc, rng, n):
    """Rays whose box-test comparisons tie or nearly tie: aimed at corners of node boxes (two or three
    slabs are crossed at the same t up to rounding), a third of them starting exactly on a plane of the
    box they aim at, plus axis-parallel ones (zero direction components)."""
    boxes, _ = orc.nodes()
    lo, hi = boxes[0, :3], boxes[0, 3:]
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n, 3))
    pick = boxes[rng.integers(0, len(boxes), n)]
    corner = np.where(rng.integers(0, 2, (n, 3)) == 1, pick[:, 3:], pick[:, :3])
    k = n // 3
    ax = rng.integers(0, 3, k)
    o[np.arange(k), ax] = pick[np.arange(k), ax]
    d = corner - o
    keep = np.linalg.norm(d, axis=1) > 1e-9
    o, d = o[keep], d[keep]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[-60:-30, 0] = 0.0
    d[-30:, :2] = 0.0
    d[-30:, 2] = -1.0
    return o, d


@pytest.mark.parametrize("name", ["cornell-box2-transmissive", "dragon-trans", "teapot", "3-dragon-spec"])
def test_pipeline_extend_kernel_on_explicit_rays_bit_exact(name, gpu_scene, oracle_scenes):
    """The kernel the renderer launches per bounce (both variants), fed with incoherent and tie-prone
    rays; hit ids and distances are the oracle's bit for bit, its work counters are the oracle's, and
    the tie-prone set does reach the exact-division fallback of the box test."""
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    rng = np.random.default_rng(5)
    boxes, _ = orc.nodes()
    lo, hi = boxes[0, :3], boxes[0, 3:]
    n = 30000
    o1 = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d1 = rng.normal(size=(n, 3))
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    o2, d2 = _tie_prone_rays(orc, rng, n)
    try:
        for (o, d, expect_fallback) in [(o1, d1, False), (o2, d2, True)]:
            want = orc.trace(o, d)
            for variant in (1, 0, 2, 3, 4):
                g.set_option("extend_variant", variant)
                g.reset_stats()
                g.set_instrumentation(time_kernels=False, count_work=True)
                tri, t = g.extend_rays(o, d)
                st = g.stats()
                assert np.array_equal(tri, want["prim"]), "hit triangle ids differ (variant %d)" % variant
                assert np.array_equal(bits(t), bits(want["t"])), "hit distances differ (variant %d)" % variant
                if variant == 2:  # the culling walk may skip work, never add any
                    assert st["node_steps"] <= int(want["node_steps"].sum())
                    continue
                if variant == 3:
                    # inner nodes through the conservative binary32 test: a superset of the reference's nodes, but only
                    # just (the margins are a few parts in a million); every leaf goes through the exact test of its own
                    # box, so the triangles tested are exactly the reference's
                    assert int(want["node_steps"].sum()) <= st["node_steps"] <= 1.02 * int(want["node_steps"].sum()) + 64
                    assert st["tri_tests"] == int(want["tri_tests"].sum())
                    continue
                if variant == 4:  # the same with culling: fewer triangles, never more
                    assert st["tri_tests"] <= int(want["tri_tests"].sum())
                    continue
                assert st["node_steps"] == int(want["node_steps"].sum())
                assert st["tri_tests"] == int(want["tri_tests"].sum())
                if variant == 1 and expect_fallback:
                    assert st["exact_node_steps"] > 0
    finally:
        g.set_option("extend_variant", 1)
        g.set_instrumentation(False, False)
        g.reset_stats()


@pytest.mark.parametrize("name", SCENES)
def test_render_extend_kernel_visit_sets_bit_exact(name, gpu_scene, oracle_scenes):
    """The visit set of GPU::BVH::traverseIterative (reference: src/gpu/GPUBVH.cu:38-99), reported per ray by the
    extend kernel the renderer launches (its counting instance): the order-free hash over the canonical ids of every
    node whose box was tested equals the oracle's for the one-ray-per-thread kernel and for the persistent kernel's
    reference walk (variants 0 and 1), on camera rays, random rays, tie-prone rays (aimed at box corners, starting
    on box planes), axis-parallel rays and rays leaving the scene's surfaces; hit triangle and t bits likewise.  On one
    scene also bitmap against bitmap."""
    g = gpu_scene(name)
    orc = oracle_scenes(name)
    rng = np.random.default_rng(17)
    boxes, _ = orc.nodes()
    lo, hi = boxes[0, :3], boxes[0, 3:]
    n = 12000
    o1 = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d1 = rng.normal(size=(n, 3))
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    o2, d2 = _tie_prone_rays(orc, rng, n)
    o3, d3 = orc.camera_rays(64, 64)
    # rays that leave the scene's own surfaces, as every ray after the first does: the random rays' hit points, pushed
    # 1e-12 along a random direction (they start on, or within rounding of, planes of the boxes around the triangle they left)
    first = orc.trace(o1, d1)
    hit = first["prim"] >= 0
    d4 = rng.normal(size=(int(hit.sum()), 3))
    d4 /= np.linalg.norm(d4, axis=1, keepdims=True)
    o4 = o1[hit] + d1[hit] * first["t"][hit][:, None] + d4 * 1e-12
    o = np.concatenate([o1, o2, o3, o4])
    d = np.concatenate([d1, d2, d3, d4])
    want = orc.trace(o, d, bitmap=(name == "teapot-trans"))
    assert len(np.unique(want["vhash"])) > 100 and len(o4) > 500
    try:
        for variant in (1, 0):
            g.set_option("extend_variant", variant)
            got = g.extend_rays_observed(o, d, bitmap=(name == "teapot-trans"))
            assert np.array_equal(got["prim"], want["prim"]), "hit triangle ids differ (variant %d)" % variant
            assert np.array_equal(bits(got["t"]), bits(want["t"])), "hit distances differ (variant %d)" % variant
            assert np.array_equal(got["vhash"], want["vhash"]), "visited-node sets differ (variant %d)" % variant
            if got["bitmap"] is not None:
                assert np.array_equal(got["bitmap"], want["bitmap"]), "visit bitmaps differ (variant %d)" % variant
    finally:
        g.set_option("extend_variant", 1)


def test_fast_walks_visit_supersets_and_subsets(gpu_scene, oracle_scenes):
    """What the non-reference walks touch, per ray, as bitmaps: the binary32 inner-node test (variant 3) is conservative, so it
    tests every box the reference tests and a few more; the culling walk (variant 2) tests only boxes the reference tests.
    Hits are the reference's in both."""
    g = gpu_scene("dragon-spec")
    orc = oracle_scenes("dragon-spec")
    rng = np.random.default_rng(23)
    boxes, _ = orc.nodes()
    lo, hi = boxes[0, :3], boxes[0, 3:]
    n = 3000
    o = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    oc, dc = orc.camera_rays(32, 32)
    o, d = np.concatenate([o, oc]), np.concatenate([d, dc])
    want = orc.trace(o, d, bitmap=True)
    try:
        for variant in (3, 2, 4):
            g.set_option("extend_variant", variant)
            got = g.extend_rays_observed(o, d, bitmap=True)
            assert np.array_equal(got["prim"], want["prim"]) and np.array_equal(bits(got["t"]), bits(want["t"]))
            extra = got["bitmap"] & ~want["bitmap"]
            lost = want["bitmap"] & ~got["bitmap"]
            if variant == 3:
                assert not lost.any(), "the binary32 walk skipped a box the reference tests"
                # "a few": the margins are absolute in position (2^-20 of the scene's extent), so a ray that barely moves
                # on an axis (this camera's middle columns: origin x = 0, direction x near 0) is loose on that axis; the
                # random rays, whose directions have no such axis, stay within a few per cent
                n_extra = int(np.unpackbits(extra[:n]).sum()), int(np.unpackbits(want["bitmap"][:n]).sum())
                assert n_extra[0] <= 0.03 * n_extra[1], n_extra
            elif variant == 2:
                assert not extra.any(), "the culling walk tested a box the reference does not"
                assert lost.any()
            else:
                assert lost.any()
    finally:
        g.set_option("extend_variant", 1)


def test_edge_cases(gpu, host_scenes, oracle):
    import torch
    from cuda_path_tracer_amd import HostScene
    # 1x1 image, depth 0, image sides that are not multiples of the 8x8 tile
    gpu.upload(host_scenes("cornell-box2"))
    orc = oracle.OracleScene(scene_path("cornell-box2"))
    for (w, h, depth) in [(1, 1, 0), (9, 3, 1), (8, 8, 0), (131, 2, 2), (8, 8, 100)]:  # ragged sides; far more bounces than any path survives
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        gpu.render_into(acc.data_ptr(), gpu.params(w, h, 0, 3, depth))
        gpu.synchronize()
        want, _ = orc.render(w, h, 0, 3, depth)
        assert np.allclose(acc.cpu().numpy(), want, rtol=1e-9, atol=1e-12)
    # empty sample range is a no-op
    acc = torch.ones(4, 4, 3, dtype=torch.float64, device="cuda")
    gpu.render_into(acc.data_ptr(), gpu.params(4, 4, 5, 5, 2))
    gpu.synchronize()
    assert float(acc.sum()) == 48.0
    # a one-triangle scene (the reference's traversal assumes an internal root)
    one = """SBT-raytracer 1.0
    camera { position = (0,0,3); }
    polymesh { material = { emissive = (0.5,0.25,1.0); }; points = ((-1,-1,0),(1,-1,0),(0,1,0)); faces = ((0,1,2)); }"""
    hs = HostScene(text=one)
    assert hs.n_tris == 1 and hs.n_nodes == 0
    gpu.upload(hs)
    img = gpu.render_image(gpu.params(16, 16, 0, 2, 2))
    assert tuple(img[8, 8]) == (127, 63, 255) and tuple(img[0, 0]) == (0, 0, 0)
    # errors are reported, not fatal
    from cuda_path_tracer_amd import CptError
    with pytest.raises(CptError):
        gpu.render_image(gpu.params(0, 16, 0, 2, 2))
    with pytest.raises(CptError):
        gpu.render_image(gpu.params(16, 16, 0, 2, -1))


def _golden_luminance_cases():
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    table = json.load(open(os.path.join(here, "golden", "luminance.json")))
    cases = []
    for scene, by_depth in sorted(table.items()):
        if scene.startswith("_"):
            continue
        for depth, rows in sorted(by_depth.items()):
            estimator, spp, lum = max(rows, key=lambda r: r[1])  # the best-converged golden render
            cases.append(pytest.param(scene, int(depth), lum, id="%s-depth%s-%s%d" % (scene, depth, estimator, spp)))
    return cases


@pytest.mark.parametrize("scene,depth,golden", _golden_luminance_cases())
def test_converged_image_against_reference_golden_luminance(scene, depth, golden, gpu_scene):
    """Soft known-answer: mean Rec.709 luminance of the reference's own 512x512 golden renders
    (important_assets/Renders/benchmark-renders/bvh_renders/{scene}.ray_{depth}_{spp}_{cpu|gpu}.png of
    the reference, tabulated in tests/golden/luminance.json), for every scene and depth it holds:
    diffuse, mirror and dielectric lobes at depths 3 to 7.  Tolerance +-1.0 of 255."""
    g = gpu_scene(scene)
    img = g.render_image(g.params(512, 512, 0, 1024, depth)).astype(np.float64)
    lum = (0.2126 * img[..., 0] + 0.7152 * img[..., 1] + 0.0722 * img[..., 2]).mean()
    assert abs(lum - golden) < 1.0, (lum, golden)


@pytest.mark.parametrize("name", ["cornell-box2", "cornell-box2-transmissive", "teapot-trans", "dragon-spec"])
def test_primary_rays_against_committed_golden_vectors(name, gpu_scene):
    """The same comparison against the fixtures under tests/golden/ (no oracle in the loop)."""
    import os
    from conftest import ROOT
    g = gpu_scene(name)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "primary_%s_64.npz" % name))
    o, d = g.camera_rays(64, 64)
    assert np.array_equal(d.view(np.uint64), gold["dir_bits"])
    r = g.trace_rays(o, d)
    assert np.array_equal(r["prim"], gold["prim"])
    assert np.array_equal(r["t"].view(np.uint64), gold["t_bits"])
    assert np.array_equal(r["vhash"], gold["vhash"])
    assert np.array_equal(r["node_steps"], gold["node_steps"]) and np.array_equal(r["tri_tests"], gold["tri_tests"])


@pytest.mark.parametrize("name", ["cornell-box2", "cornell-box2-transmissive", "teapot-trans", "dragon-spec"])
def test_full_size_primary_rays_digest(name, gpu_scene):
    """512x512 primary rays (the benchmark's image size): digests of directions, hit ids, distances
    and visit hashes against the committed ones."""
    import hashlib
    import json
    import os
    from conftest import ROOT
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "primary_digests.json")))[name]
    g = gpu_scene(name)
    o, d = g.camera_rays(512, 512)
    r = g.trace_rays(o, d)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert sha(d) == want["dir"] and sha(r["prim"]) == want["prim"] and sha(r["t"]) == want["t"]
    assert sha(r["vhash"]) == want["vhash"]
    assert int(r["node_steps"].sum()) == want["node_steps"] and int(r["tri_tests"].sum()) == want["tri_tests"]


def test_work_counters_match_oracle(gpu_scene, oracle_scenes):
    """Node expansions and triangle tests are a property of the rays, not of the implementation:
    the device-side counters must equal the oracle's for the same render."""
    import torch
    g = gpu_scene("cornell-box2-transmissive")
    w, h, spp, depth = 32, 32, 4, 6
    _, want = oracle_scenes("cornell-box2-transmissive").render(w, h, 0, spp, depth)
    for variant in (0, 1):
        g.set_option("extend_variant", variant)
        g.reset_stats()
        g.set_instrumentation(False, True)
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth))
        g.synchronize()
        st = g.stats()
        g.set_instrumentation(False, False)
        for k in ("paths", "segments", "hits", "node_steps", "tri_tests"):
            assert st[k] == want[k], (variant, k, st[k], want[k])
    g.set_option("extend_variant", 1)


def test_every_pipeline_variant_gives_the_same_image(gpu_scene):
    import torch
    g = gpu_scene("teapot-trans")
    w, h, spp, depth = 96, 64, 6, 8
    images = []
    try:
        for ev in (0, 1, 2, 3, 4):  # 2 = nearest-first with culling, 3 / 4 = binary32 inner nodes without / with culling: other visit sets, the same closest hits
            g.set_option("extend_variant", ev)
            acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
            g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth))
            g.synchronize()
            images.append(acc.cpu().numpy())
        # batches in flight on 1, 2 and 4 streams, with batches small enough that every lane is used
        for streams in (1, 2, 4):
            g.set_option("streams", streams)
            g.set_option("batch_paths", 2 * 96 * 64)
            acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
            g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth))
            g.synchronize()
            images.append(acc.cpu().numpy())
    finally:
        g.set_option("extend_variant", 1)
        g.set_option("streams", 2)
        g.set_option("batch_paths", 1 << 25)
    for im in images[1:]:
        assert np.array_equal(bits(images[0]), bits(im))


@pytest.mark.parametrize("name", ["cornell-box2-transmissive", "teapot-trans", "dragon-spec"])
def test_shared_origin_instance_of_the_first_bounce_changes_nothing(name, gpu_scene):
    """First rays of a pin-hole render go through the instance of the persistent extend kernel whose records already
    contain `plane - eye` and the triangles' plane terms for the eye (reference operations: src/gpu/GPUbb.cuh:33-34,
    src/gpu/GPUTrimesh.cuh:62).  With it and without it: the same accumulation buffer bit for bit and, on the reference
    walk, the same work counters -- on an image whose sides are not multiples of 8, so that slots without a path
    take the instance's short cut."""
    import torch
    g = gpu_scene(name)
    w, h, spp, depth = 61, 45, 5, 6
    try:
        for variant in (1, 2):
            for repeat in (2, 3):
                g.set_option("extend_variant", variant)
                g.set_option("pw_box_repeat", repeat)
                got = []
                for eye in (1, 0):
                    g.set_option("pw_eye", eye)
                    g.reset_stats()
                    g.set_instrumentation(False, True)
                    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
                    g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth))
                    g.synchronize()
                    st = g.stats()
                    g.set_instrumentation(False, False)
                    # and once more without the counting instance
                    acc2 = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
                    g.render_into(acc2.data_ptr(), g.params(w, h, 0, spp, depth))
                    g.synchronize()
                    assert np.array_equal(bits(acc.cpu().numpy()), bits(acc2.cpu().numpy()))
                    got.append((acc.cpu().numpy(), st))
                assert np.array_equal(bits(got[0][0]), bits(got[1][0])), (variant, repeat)
                assert got[0][0].any()
                if variant == 1:  # the culling walk's counts depend on when a wave runs its triangle passes
                    for k in ("paths", "segments", "hits", "node_steps", "tri_tests", "exact_node_steps"):
                        assert got[0][1][k] == got[1][1][k], (variant, repeat, k, got[0][1][k], got[1][1][k])
    finally:
        g.set_option("extend_variant", 1)
        g.set_option("pw_box_repeat", 0)
        g.set_option("pw_eye", 1)


@pytest.mark.parametrize("name", ["cornell-box2-transmissive", "teapot-trans", "dragon-spec", "3-dragon-spec"])
def test_fast_tree_returns_the_reference_hits_and_the_same_image(name, gpu, host_scenes, oracle_scenes, oracle):
    """The fast mode's tree (surface-area build over the same triangles, the reference tree's tie-break ranks in the
    triangle records): every extend variant returns, ray by ray, the triangle and the distance the oracle finds on
    the REFERENCE tree, also for tie-prone and axis-parallel rays; the image is the reference tree's bit for bit;
    the walk itself is the reference's algorithm on this tree (visit hashes and step counts equal the oracle's when it
    is given this tree), and it tests fewer boxes."""
    import torch
    hs = host_scenes(name)
    orc = oracle_scenes(name)
    rng = np.random.default_rng(41)
    boxes, _ = orc.nodes()
    lo, hi = boxes[0, :3], boxes[0, 3:]
    n = 15000
    o1 = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d1 = rng.normal(size=(n, 3))
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    o2, d2 = _tie_prone_rays(orc, rng, n)
    o3, d3 = orc.camera_rays(48, 48)
    o, d = np.concatenate([o1, o2, o3]), np.concatenate([d1, d2, d3])
    want = orc.trace(o, d)
    w, h, spp, depth = 64, 48, 4, 6
    gpu.upload(hs)
    ref_img = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    gpu.render_into(ref_img.data_ptr(), gpu.params(w, h, 0, spp, depth))
    gpu.synchronize()
    fast = hs.fast()
    assert fast.n_tris == hs.n_tris and fast.n_tree_nodes == 2 * hs.n_tris - 1 and fast.tree_height <= 32
    gpu.upload(fast)
    orc_fast = oracle.OracleScene(scene_path(name))
    orc_fast.set_tree(*hs.fast_tree())
    want_fast = orc_fast.trace(o, d)
    assert np.array_equal(bits(want_fast["t"]), bits(want["t"]))
    assert int(want_fast["node_steps"].sum()) < 0.9 * int(want["node_steps"].sum())
    try:
        for variant in (1, 0, 2, 3, 4):
            gpu.set_option("extend_variant", variant)
            got = gpu.extend_rays_observed(o, d)
            assert np.array_equal(got["prim"], want["prim"]), "hit triangles differ from the reference tree's (variant %d)" % variant
            assert np.array_equal(bits(got["t"]), bits(want["t"])), "hit distances differ (variant %d)" % variant
            if variant in (0, 1):
                assert np.array_equal(got["vhash"], want_fast["vhash"]), "visit sets on the fast tree differ from the oracle's on that tree"
            img = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
            gpu.render_into(img.data_ptr(), gpu.params(w, h, 0, spp, depth))
            gpu.synchronize()
            assert torch.equal(img.view(torch.int64), ref_img.view(torch.int64)), "image differs from the reference tree's (variant %d)" % variant
    finally:
        gpu.set_option("extend_variant", 1)


def test_cli_fast_mode_writes_the_same_image(tmp_path):
    """bin/ray -g --fast (surface-area tree with the reference's tie-break ranks; culling walk on the large scene)
    writes byte for byte the pixels of the default render."""
    import os
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    for name in ("cornell-box2-transmissive", "dragon-spec"):
        imgs = []
        for extra in ([], ["--fast"]):
            out = str(tmp_path / ("o%d.png" % len(imgs)))
            r = subprocess.run([exe, "-g", "-s", "6", "-r", "5", "-w", "96"] + extra + [scene_path(name), out],
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            imgs.append(np.asarray(Image.open(out).convert("RGB")))
        assert imgs[0].max() > 0 and np.array_equal(imgs[0], imgs[1])


def test_cli_renders_png_on_gpu(tmp_path, gpu_scene):
    """bin/ray -g with the reference's flags writes the image the library renders."""
    import os
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    out = str(tmp_path / "o.png")
    r = subprocess.run([exe, "-g", "-s", "4", "-r", "3", "-w", "64", scene_path("cornell-box2"), out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.asarray(Image.open(out).convert("RGB"))
    g = gpu_scene("cornell-box2")
    want = g.render_image(g.params(64, 64, 0, 4, 3))
    assert np.array_equal(got, want[::-1])  # PNG rows are top-down, the buffer is bottom-up


def test_cli_splits_samples_over_several_contexts(tmp_path):
    """bin/ray --devices a,b: one host thread and one context per listed GPU, each rendering its share of the
    samples; rehearsed with the same GPU listed three times (7 samples -> 3 + 2 + 2).  The sample set is the
    single-context one, only the last additions associate differently, so the 8-bit image may differ by one level
    in a few channels at most."""
    import os
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    base = [exe, "-g", "-s", "7", "-r", "5", "-w", "96", scene_path("cornell-box2-transmissive")]
    one, three = str(tmp_path / "one.png"), str(tmp_path / "three.png")
    r = subprocess.run(base + [one], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base[:2] + ["--devices", "0,0,0"] + base[2:] + [three], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    a = np.asarray(Image.open(one).convert("RGB")).astype(int)
    b = np.asarray(Image.open(three).convert("RGB")).astype(int)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1 and (a != b).mean() < 0.01
    r = subprocess.run(base[:2] + ["--devices", "0,99"] + base[2:] + [three], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "GPU 99" in (r.stdout + r.stderr)
    r = subprocess.run(base[:2] + ["--devices", "0,0", "--progressive", "2"] + base[2:] + [three], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_render_split_over_contexts_adds_the_sums_on_the_first_gpu(gpu_scene, host_scenes):
    """cpt_render_image_multi, rehearsed with three contexts on the one GPU (7 samples -> 3 + 2 + 2): the image is the
    tonemap of ((sum 0 + sum 1) + sum 2), each sum rendered by its context alone -- checked against those sums made one
    by one and added on the host in the same order.  Bad lists are refused."""
    import torch
    import cuda_path_tracer_amd as cpt
    name = "cornell-box2-transmissive"
    g = gpu_scene(name)
    others = [cpt.Renderer(0), cpt.Renderer(0)]
    try:
        for o in others:
            o.upload(host_scenes(name))
        w, h, depth = 61, 45, 5
        rgb, done = g.render_image_multi(others, g.params(w, h, 2, 9, depth))
        assert done == 7
        total = np.zeros((h, w, 3))
        for k, (a, b) in enumerate([(2, 5), (5, 7), (7, 9)]):
            acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
            g.render_into(acc.data_ptr(), g.params(w, h, a, b, depth))
            g.synchronize()
            total = acc.cpu().numpy() if k == 0 else total + acc.cpu().numpy()
        want = (255.0 * np.clip(total / 7, 0.0, 1.0)).astype(np.uint8)
        assert np.array_equal(rgb, want)
        # more contexts than samples: the last ones render nothing
        rgb2, done2 = g.render_image_multi(others, g.params(w, h, 0, 2, depth))
        assert done2 == 2 and np.array_equal(rgb2, g.render_image(g.params(w, h, 0, 2, depth)))
        with pytest.raises(cpt.CptError):
            g.render_image_multi([others[0], others[0]], g.params(w, h, 0, 4, depth))
        bare = cpt.Renderer(0)
        try:
            with pytest.raises(cpt.CptError):
                g.render_image_multi([bare], g.params(w, h, 0, 4, depth))
        finally:
            bare.close()
    finally:
        for o in others:
            o.close()


def test_progressive_and_resumed_renders_are_bit_identical(gpu_scene):
    """cpt_render_progressive: chunked, stopped-and-resumed and single-call renders hold the same
    radiance sums bit for bit and the same image; the progress callback sees every chunk."""
    import torch
    g = gpu_scene("cornell-box2-transmissive")
    w, h, spp, depth = 48, 40, 12, 5
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.render_into(acc.data_ptr(), g.params(w, h, 0, spp, depth))
    g.synchronize()
    want_acc = acc.cpu().numpy()
    want_rgb = g.render_image(g.params(w, h, 0, spp, depth))
    seen = []
    a1 = np.zeros((h, w, 3), np.float64)
    rgb1, done1 = g.render_progressive(g.params(w, h, 0, spp, depth), 5, accum=a1, on_progress=lambda s, im: seen.append(s))
    assert done1 == spp and seen == [5, 10, 12]
    assert np.array_equal(bits(a1), bits(want_acc)) and np.array_equal(rgb1, want_rgb)
    # stop after the second chunk, then resume from the saved sums
    a2 = np.zeros((h, w, 3), np.float64)
    _, done2 = g.render_progressive(g.params(w, h, 0, spp, depth), 4, accum=a2, on_progress=lambda s, im: s >= 8)
    assert done2 == 8
    rgb2, done3 = g.render_progressive(g.params(w, h, done2, spp, depth), 100, accum=a2, accum_samples=done2)
    assert done3 == spp - done2
    assert np.array_equal(bits(a2), bits(want_acc)) and np.array_equal(rgb2, want_rgb)


def test_cli_progressive_checkpoint_and_metrics(tmp_path):
    """bin/ray --progressive/--checkpoint/--metrics: a render stopped at 6 samples and resumed to 10
    writes the same PNG as a single 10-sample run, and the metrics file says what was done."""
    import json
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    scene = scene_path("cornell-box2-specular")
    base = [exe, "-g", "-r", "4", "-w", "72"]
    one = str(tmp_path / "one.png")
    r = subprocess.run(base + ["-s", "10", scene, one], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ck, two, met = str(tmp_path / "ck.bin"), str(tmp_path / "two.png"), str(tmp_path / "m.json")
    r = subprocess.run(base + ["-s", "6", "--progressive", "4", "--checkpoint", ck, scene, two], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "4 samples per pixel written" in r.stdout and "6 samples per pixel written" in r.stdout
    r = subprocess.run(base + ["-s", "10", "--checkpoint", ck, "--metrics", met, scene, two], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "resuming from" in r.stdout and "at 6 samples" in r.stdout
    assert open(one, "rb").read() == open(two, "rb").read()
    m = json.load(open(met))
    assert m["samples_in_image"] == 10 and m["width"] == 72 and m["paths"] == 72 * m["height"] * 4 and m["seconds"] > 0
    # a checkpoint of another render (different depth) is ignored, not mixed in
    r = subprocess.run([exe, "-g", "-r", "2", "-w", "72", "-s", "3", "--checkpoint", ck, scene, two], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ignoring" in r.stdout
    # ... and so is a checkpoint of another scene of the same size and depth, or of the same scene on the other kind of tree
    ck2 = str(tmp_path / "ck2.bin")
    r = subprocess.run(base + ["-s", "4", "--checkpoint", ck2, scene, two], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    other = scene_path("cornell-box2-transmissive")
    three = str(tmp_path / "three.png")
    r = subprocess.run(base + ["-s", "4", other, three], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    four = str(tmp_path / "four.png")
    r = subprocess.run(base + ["-s", "4", "--checkpoint", ck2, other, four], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ignoring" in r.stdout
    assert open(three, "rb").read() == open(four, "rb").read()  # nothing of the first scene was mixed in
    r = subprocess.run(base + ["-s", "4", "--checkpoint", ck2, other, four], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "resuming" in r.stdout  # the file was replaced by this scene's sums
    r = subprocess.run(base + ["-s", "4", "--device-bvh", "--checkpoint", ck2, other, four], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ignoring" in r.stdout


@pytest.mark.parametrize("scaling,launcher", [("weak", "torchrun"), ("strong", "self"), ("strong-tiles", "self")])
def test_bench_two_ranks_on_one_gpu(scaling, launcher):
    """bench.py --gpus 2, rehearsed on one GPU (both ranks share it; gloo stands in for RCCL, which refuses two ranks
    on one device): the N > 1 path -- sharding, the reduce, the max-over-ranks timing, rank 0's solo instrumentation
    steps, the final barrier -- runs to the end and prints one JSON line.  Under torch.distributed.run as the driver
    starts it, and as plain `python bench.py --gpus 2`, which starts the launcher itself; weak scaling (every rank
    adds the workload's sample count), strong scaling by samples, and strong scaling by tiles (a workload with
    fewer samples than ranks is not among the named ones, so that case uses the library call directly below)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CPT_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    tail = ["--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "cornell2", "--extra", "none", "--no-cpu-baseline", "--no-fast-modes",
            "--scaling", "weak" if scaling == "weak" else "strong"]
    if launcher == "torchrun":
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    if scaling == "strong-tiles":
        env["CPT_BENCH_CORNELL2_SPP"] = "1"  # one sample for two ranks: the image is split by tiles
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["rccl_ranks"] == 2
    if scaling == "weak":
        assert out["scaling"] == "weak" and out["config"]["paths_per_step"] == 2 * 512 * 512 * 16
    elif scaling == "strong":
        assert out["scaling"] == "strong" and out["config"]["paths_per_step"] == 512 * 512 * 16
    else:
        assert out["scaling"] == "strong" and out["config"]["paths_per_step"] == 512 * 512
    assert 0 < out["roofline"]["frac"] <= 1 and out["roofline"]["bound"] == "fp64_valu_issue"


def test_bench_full_node_shapes_on_two_ranks():
    """The line bench.py prints for N > 1: beside the weak-scaling headline, the fixed-total shapes of the full-node
    configurations (here at 1/8 of their width and 1/64 of their samples) dealt over the ranks, each with the slowest
    rank's render and reduce times and the number of ranks the reduce went through."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CPT_BENCH_BACKEND="gloo", CPT_BENCH_SHRINK="8")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "cornell2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["paths_per_step"] == 2 * 512 * 512 * 16
    assert out["render_ms"] > 0 and out["reduce_ms"] >= 0
    assert "synthetic" not in out["data"].split(",")[0] and "cornell-box2" in out["data"]
    want = {"cfg4": 128 * 128 * 64, "cfg5_cornell": 64 * 64 * 256, "cfg5_dragon": 256 * 256 * 128}
    assert set(out["workloads"]) == set(want)
    for name, paths in want.items():
        w = out["workloads"][name]
        assert w["scaling"] == "strong" and w["n_gpus"] == 2 and w["rccl_ranks"] == 2
        assert w["paths_per_step"] == paths and w["value"] > 0
        assert w["render_ms"] > 0 and w["reduce_ms"] >= 0
    rf = out["roofline"]
    assert rf["instr_per_node_step"] == 70 and rf["instr_per_triangle_test"] == 89
    assert rf["hbm_algorithmic_ratio_to_hbm_peak"] > 0


def test_bench_one_rank_through_rccl():
    """What one GPU can rehearse of the RCCL path: bench.py under the launcher with a single rank and
    CPT_BENCH_FORCE_DIST=1 brings the nccl (= RCCL) process group up on the device and sends the warm-up reduce, every
    step's fp64 radiance reduce, the timing all-reduce and the barriers through it."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CPT_BENCH_FORCE_DIST="1")
    env.pop("CPT_BENCH_BACKEND", None)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload",
           "cornell2", "--extra", "none", "--no-cpu-baseline", "--no-fast-modes"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["backend"] == "nccl" and out["config"]["rccl_ranks"] == 1 and out["value"] > 0


def test_bench_fails_when_a_rank_fails():
    """A rank that dies makes the self-launched bench exit non-zero and print no result line (here every rank
    refuses an option the library does not know)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, CPT_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "cornell2",
                        "--extra", "none", "--opt", "no_such_option=1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=200, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_deep_stacks_spill_past_the_lds_entries(tmp_path, gpu, oracle):
    """A row of 3000 triangles facing down the x axis, spaced geometrically: the SAH tree is 22 levels
    tall and a ray along the axis has up to 15 postponed siblings, more than the 12 stack entries the
    persistent extend kernel keeps in LDS, so its global-memory spill columns are used."""
    from cuda_path_tracer_amd import HostScene
    pts, faces = [], []
    x, dx = 0.0, 0.001
    for i in range(3000):
        pts += ["(%r,-1,-1)" % x, "(%r,1,-1)" % x, "(%r,0,1)" % x]
        faces.append("(%d,%d,%d)" % (3 * i, 3 * i + 1, 3 * i + 2))
        x += dx
        dx *= 1.003
    text = ("SBT-raytracer 1.0\ncamera { position = (-5,0,0); viewdir = (1,0,0); updir = (0,1,0); }\n"
            "polymesh { material = { diffuse = (0.5,0.5,0.5); transmissive = (0.5,0.5,0.5); index = 1.0; };\n"
            " points = (%s); faces = (%s); }\n" % (",".join(pts), ",".join(faces)))
    path = tmp_path / "cards.ray"
    path.write_text(text)
    orc = oracle.OracleScene(str(path))
    hs = HostScene(str(path))
    assert hs.desc.tree_height >= 20
    gpu.upload(hs)
    rng = np.random.default_rng(2)
    n = 4000
    o = np.column_stack([np.full(n, -5.0), rng.uniform(-0.9, 0.9, n), rng.uniform(-0.9, 0.9, n)])
    d = np.column_stack([np.ones(n), rng.normal(0, 0.02, n), rng.normal(0, 0.02, n)])
    o[n // 2:, 0] = x + 5.0  # half of them from the far side, against the face normals
    d[n // 2:, 0] = -1.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    want = _compare_trace(gpu, orc, o, d, bitmap=False)
    assert int(want["max_stack"].max()) > 12
    try:
        for variant in (1, 0, 2, 3, 4):
            gpu.set_option("extend_variant", variant)
            tri, t = gpu.extend_rays(o, d)
            assert np.array_equal(tri, want["prim"]) and np.array_equal(bits(t), bits(want["t"])), variant
    finally:
        gpu.set_option("extend_variant", 1)


def test_balanced_row_of_cards_needs_the_spill_columns_under_any_child_order(tmp_path, gpu, oracle):
    """The persistent extend kernel takes each node's children in the order that keeps its stack shortest, which brings
    the bundled scenes' needs (4 to 10 entries) under the 12 entries it holds in LDS.  The reference's nearly balanced tree
    over 2^15 cards in an evenly spaced row needs 13 whichever child goes first (worked out below from its topology), and
    rays along the row do pierce every box: the instance with the global-memory stack columns is the one that runs, for
    the reference walk too, and its hits and visit sets are the oracle's."""
    from cuda_path_tracer_amd import HostScene
    n_cards = 1 << 15
    pts, faces = [], []
    for i in range(n_cards):
        x = 0.01 * i
        pts += ["(%r,-1,-1)" % x, "(%r,1,-1)" % x, "(%r,0,1)" % x]
        faces.append("(%d,%d,%d)" % (3 * i, 3 * i + 1, 3 * i + 2))
    text = ("SBT-raytracer 1.0\ncamera { position = (-5,0,0); viewdir = (1,0,0); updir = (0,1,0); }\n"
            "polymesh { material = { diffuse = (0.5,0.5,0.5); transmissive = (0.5,0.5,0.5); index = 1.0; };\n"
            " points = (%s); faces = (%s); }\n" % (",".join(pts), ",".join(faces)))
    path = tmp_path / "row.ray"
    path.write_text(text)
    orc = oracle.OracleScene(str(path))
    hs = HostScene(str(path))
    topo = hs.tree()[1]
    need = np.zeros(len(topo), int)  # fewest stack entries a depth-first walk of the subtree can do with
    for k in range(len(topo) - 1, -1, -1):
        l, r, prim = topo[k]
        if prim < 0:
            nl = need[l] if topo[l, 2] < 0 else 0
            nr = need[r] if topo[r, 2] < 0 else 0
            need[k] = max(max(nl, nr), min(nl, nr) + 1) if (topo[l, 2] < 0 and topo[r, 2] < 0) else max(nl, nr)
    assert need[0] == 13
    gpu.upload(hs)
    rng = np.random.default_rng(4)
    n = 300
    o = np.column_stack([np.full(n, -5.0), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)])
    d = np.column_stack([np.ones(n), rng.normal(0, 0.001, n), rng.normal(0, 0.001, n)])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    want = orc.trace(o, d)
    assert int(want["max_stack"].max()) >= 13
    try:
        for variant in (1, 0, 2):
            gpu.set_option("extend_variant", variant)
            got = gpu.extend_rays_observed(o, d)
            assert np.array_equal(got["prim"], want["prim"]) and np.array_equal(bits(got["t"]), bits(want["t"])), variant
            if variant != 2:
                assert np.array_equal(got["vhash"], want["vhash"]), variant
    finally:
        gpu.set_option("extend_variant", 1)


def test_stream_handover_does_not_touch_the_previous_stream(gpu_scene):
    """cpt_set_stream orders the new stream behind the context's earlier work through an event of the context's own,
    not through the previous stream's handle: the previous stream may be gone by then (a torch stream that went out
    of scope).  Renders on three streams in a row, each handed over after the one before was dropped, add up to the
    one-stream render bit for bit."""
    import gc
    import torch
    g = gpu_scene("cornell-box2-specular")
    w, h, depth = 96, 96, 5
    want = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.set_stream(None)
    g.render_into(want.data_ptr(), g.params(w, h, 0, 9, depth))
    g.synchronize()
    got = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    try:
        for k in range(3):
            st = torch.cuda.Stream()
            g.set_stream(st.cuda_stream)
            g.render_into(got.data_ptr(), g.params(w, h, 3 * k, 3 * k + 3, depth))
            del st
            gc.collect()
    finally:
        g.set_stream(None)
    g.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(want.view(torch.int64), got.view(torch.int64))


def test_malformed_scenes_are_refused_at_upload(gpu, host_scenes):
    """The kernels index the scene arrays unchecked, so cpt_upload_scene must refuse a scene whose
    structure is broken, with a message, and leave the context usable."""
    import ctypes as C
    from cuda_path_tracer_amd import CptError, _capi
    hs = host_scenes("cornell-box2")
    base_nodes, base_tris, base_norm, base_mats = hs.nodes(), hs.tris(), hs.tri_normals(), hs.materials()

    class Shim:  # what Renderer.upload needs: .desc, with the arrays kept alive beside it
        pass

    def build(nodes, tris, **over):
        s = Shim()
        s.keep = (np.ascontiguousarray(nodes), np.ascontiguousarray(tris), np.ascontiguousarray(base_norm), np.ascontiguousarray(base_mats))
        d = _capi.SceneDesc()
        C.memmove(C.byref(d), C.byref(hs.desc), C.sizeof(d))
        d.nodes = C.cast(s.keep[0].ctypes.data, C.POINTER(_capi.Node))
        d.tris = C.cast(s.keep[1].ctypes.data, C.POINTER(_capi.Tri))
        d.tri_normals = C.cast(s.keep[2].ctypes.data, C.POINTER(_capi.TriNormals))
        d.materials = C.cast(s.keep[3].ctypes.data, C.POINTER(_capi.Material))
        for k, v in over.items():
            setattr(d, k, v)
        s.desc = d
        return s

    gpu.upload(build(base_nodes, base_tris))  # the untouched copy is accepted
    cases = []
    n = base_nodes.copy(); n["lref"][0] = len(n) + 5; cases.append((build(n, base_tris), "child record"))
    n = base_nodes.copy(); n["rref"][1] = 0; cases.append((build(n, base_tris), "not after its parent"))
    n = base_nodes.copy()
    leaves = [(k, f) for k in range(len(n)) for f in ("lref", "rref") if n[f][k] < 0]
    (k0, f0), (k1, f1) = leaves[0], leaves[1]
    n[f1][k1] = n[f0][k0]; cases.append((build(n, base_tris), "two leaves"))
    n = base_nodes.copy(); n["lbox"][2][1] = np.inf; cases.append((build(n, base_tris), "not finite"))
    n = base_nodes.copy(); n["rid"][0] = 10 ** 6; cases.append((build(n, base_tris), "canonical id"))
    t = base_tris.copy(); t["material"][3] = 99; cases.append((build(base_nodes, t), "bad material"))
    cases.append((build(base_nodes, base_tris, n_nodes=hs.n_nodes - 1), "n_nodes"))
    cases.append((build(base_nodes, base_tris, tree_height=64), "stack"))
    # a 40-triangle chain (every right child is the next node) is 40 levels tall whatever tree_height claims
    m = 40
    chain = np.zeros(m - 1, base_nodes.dtype)
    for k in range(m - 1):
        chain["lbox"][k] = chain["rbox"][k] = (0, 0, 0, 1, 1, 1)
        chain["lref"][k] = ~k
        chain["rref"][k] = k + 1 if k + 1 < m - 1 else ~(m - 1)
        chain["lid"][k], chain["rid"][k], chain["self_id"][k] = 2 * k + 1, 2 * k + 2, 2 * k
    tall = Shim()
    tall.keep = (chain, np.repeat(base_tris[:1], m), np.repeat(base_norm[:1], m), np.ascontiguousarray(base_mats))
    d = _capi.SceneDesc()
    C.memmove(C.byref(d), C.byref(hs.desc), C.sizeof(d))
    d.n_nodes, d.n_tris, d.n_tree_nodes, d.tree_height = m - 1, m, 2 * m - 1, 5
    d.nodes = C.cast(tall.keep[0].ctypes.data, C.POINTER(_capi.Node))
    d.tris = C.cast(tall.keep[1].ctypes.data, C.POINTER(_capi.Tri))
    d.tri_normals = C.cast(tall.keep[2].ctypes.data, C.POINTER(_capi.TriNormals))
    d.materials = C.cast(tall.keep[3].ctypes.data, C.POINTER(_capi.Material))
    tall.desc = d
    cases.append((tall, "tree height 40"))
    for scene, needle in cases:
        with pytest.raises(CptError) as e:
            gpu.upload(scene)
        assert needle in str(e.value), (needle, str(e.value))
    # still usable
    gpu.upload(hs)
    img = gpu.render_image(gpu.params(16, 16, 0, 2, 2))
    assert img.shape == (16, 16, 3)


def _reference_block_cases():
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    names = json.load(open(os.path.join(here, "golden", "reference_blocks.json")))
    cases = []
    for key, fname in sorted(names.items()):
        scene, depth, estimator = key.split("|")
        spp = int(fname.split("_")[-2])
        cases.append(pytest.param(key, scene, int(depth), estimator, spp, id=fname))
    return cases


@pytest.mark.parametrize("key,scene,depth,estimator,ref_spp", _reference_block_cases())
def test_converged_image_against_reference_renders_blockwise(key, scene, depth, estimator, ref_spp, gpu_scene):
    """Per-block RMSE against the reference's own golden renders, CPU and GPU estimator alike
    (tests/golden/reference_blocks.npz: mean RGB of the 64 x 64 blocks of 8 x 8 pixels of each 512 x 512
    render, made by make_reference_blocks.py), on the 0-255 scale, after setting aside the 1 % of blocks
    that differ most: the outline of the directly visible, saturated light sits a fraction of a pixel
    away in the golden files and dominates an untrimmed sum.

    GPU renders (the estimator rebuilt here): the reference image is noisy, so the yardstick is
    measured -- two independent renders of ours at ref_spp give the RMS block noise sigma at that sample
    count, and the reference must be no further from our 8192-spp image than 1.5 sigma + 0.25, i.e.
    indistinguishable from one more independent render.  CPU renders (the branching estimator, whose
    noise at equal spp is not ours): RMSE below 6, the same order as the 10/255 of the course's own
    image checker (reference: raycheck.py:229-232).  Both: block-mean difference below 1.
    A wrong lobe, refraction direction or light falloff moves whole regions by far more than that."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    ref = np.load(os.path.join(here, "golden", "reference_blocks.npz"))[key].astype(np.float64)
    g = gpu_scene(scene)

    def blocks(s0, s1):
        img = g.render_image(g.params(512, 512, s0, s1, depth)).astype(np.float64)
        return img.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3))

    def trimmed_rms(a, b):
        sq = ((a - b) ** 2).mean(axis=2).ravel()
        sq.sort()
        return float(np.sqrt(sq[:int(0.99 * sq.size)].mean()))
    hi = blocks(0, 8192)
    got = trimmed_rms(hi, ref)
    if estimator == "gpu":
        sigma = trimmed_rms(blocks(8192, 8192 + ref_spp), blocks(8192 + ref_spp, 8192 + 2 * ref_spp)) / np.sqrt(2.0)
        assert got < 1.5 * sigma + 0.25, (got, sigma)
    else:
        assert got < 6.0, got
    assert abs(hi.mean() - ref.mean()) < 1.0


@pytest.mark.parametrize("scene,depth,rms_limit", [("teapot-trans", 5, 1.5), ("cornell-box", 3, 3.0)])
def test_converged_image_against_reference_showcase_renders(scene, depth, rms_limit, gpu_scene):
    """The reference's 16 384-spp showcase renders (tests/golden/showcase_blocks.npz: 8x8-pixel block means of
    important_assets/Renders/teapot-trans-16384-5.png and cornell-box-16384.bmp, made by make_reference_blocks.py).
    The teapot is the only golden image of a smooth-shaded mesh with a dielectric lobe and caustics; its file name
    carries the depth (5).  The Cornell box's does not: depth 3 is the one that matches (the neighbouring depths are
    2 and 4 times further away), so for that image the depth is fitted, not given.  Trimmed RMSE of the block means
    (1 % of blocks set aside, 0-255 scale) and the overall level; the next depths up and down must fit worse."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    ref = np.load(os.path.join(here, "golden", "showcase_blocks.npz"))[scene].astype(np.float64)
    g = gpu_scene(scene)

    def fit(d, spp):
        img = g.render_image(g.params(512, 512, 0, spp, d)).astype(np.float64)
        got = img.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3))
        sq = ((got - ref) ** 2).mean(axis=2).ravel()
        sq.sort()
        return float(np.sqrt(sq[:int(0.99 * sq.size)].mean())), float(got.mean())
    rms, level = fit(depth, 8192)
    assert rms < rms_limit, rms
    assert abs(level - ref.mean()) < 0.5, (level, ref.mean())
    assert fit(depth - 1, 2048)[0] > 1.5 * rms and fit(depth + 1, 2048)[0] > 1.3 * rms


def test_switching_streams_is_a_fence(gpu_scene):
    """cpt_set_stream orders the new stream after what the context queued on the old one: two renders into one buffer,
    the second on another stream with no synchronisation in between, add up to the single render of both sample
    ranges (the streams, counters and scratch of the context are shared by the two)."""
    import torch
    g = gpu_scene("cornell-box2-transmissive")
    w, h, depth = 160, 120, 6
    want = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.render_into(want.data_ptr(), g.params(w, h, 0, 24, depth))
    g.synchronize()
    got = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        g.set_stream(a.cuda_stream)
        g.render_into(got.data_ptr(), g.params(w, h, 0, 13, depth))
        g.set_stream(b.cuda_stream)
        g.render_into(got.data_ptr(), g.params(w, h, 13, 24, depth))
        g.synchronize()  # waits for stream b, which was ordered after stream a by the switch
        a.synchronize()
    finally:
        g.set_stream(None)
    assert torch.equal(want.view(torch.int64), got.view(torch.int64))
