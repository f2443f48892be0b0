"""The BVH built on the GPU (cpt_build_scene): structural checks on the downloaded tree, and traversal /
render parity against the oracle walking that same tree.

The device build produces a different tree than the reference's host build, so it is compared with the
oracle in two steps: (1) the tree is a correct BVH over the scene's triangles -- every triangle in one
leaf, every box the exact bound of what is below it, numbering and tie-break ranks as the traversal
defines them; (2) the oracle's restatement of the reference traversal, given this tree, returns the
same hits, visit sets and step counts bit for bit, and the same radiance to 1e-9.
"""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

SCENES = ["cornell-box2-transmissive", "teapot-trans", "dragon-spec"]
MAX_HEIGHT = 34  # 32-entry traversal stacks


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def tree_table(nodes, n_tris):
    """Node records -> the oracle's table (row 0 = root; boxes; left row, right row, prim)."""
    n_rec = len(nodes)
    n = 2 * n_tris - 1
    boxes = np.zeros((n, 6))
    topo = np.full((n, 3), -1, np.int32)
    if n_rec == 0:
        topo[0, 2] = 0
        return boxes, topo
    row_of = {0: 0}
    next_row = 1
    root = nodes[0]
    boxes[0, :3] = np.minimum(root["lbox"][:3], root["rbox"][:3])
    boxes[0, 3:] = np.maximum(root["lbox"][3:], root["rbox"][3:])
    for k in range(n_rec):  # children come after their parents, so row_of[k] is known here
        r = row_of[k]
        for side, (ref, box) in enumerate([(nodes[k]["lref"], nodes[k]["lbox"]), (nodes[k]["rref"], nodes[k]["rbox"])]):
            row = next_row
            next_row += 1
            boxes[row] = box
            topo[r, side] = row
            if ref >= 0:
                assert ref > k and ref not in row_of
                row_of[int(ref)] = row
            else:
                topo[row, 2] = ~int(ref)
    assert next_row == n
    return boxes, topo


def check_structure(nodes, rank, height, tris):
    n_tris = len(tris)
    assert len(nodes) == n_tris - 1
    lo = np.minimum(np.minimum(tris["a"], tris["b"]), tris["c"])
    hi = np.maximum(np.maximum(tris["a"], tris["b"]), tris["c"])
    seen = np.zeros(n_tris, int)
    own = np.zeros((len(nodes), 6))  # a record's own box: the union of its child boxes
    levels = np.ones(len(nodes), int)
    for k in range(len(nodes) - 1, -1, -1):  # children first
        for ref, box in ((nodes[k]["lref"], nodes[k]["lbox"]), (nodes[k]["rref"], nodes[k]["rbox"])):
            if ref >= 0:
                assert k < ref < len(nodes)
                assert np.array_equal(bits(box), bits(own[ref])), "internal child box is not the union below it"
                levels[k] = max(levels[k], levels[ref] + 1)
            else:
                t = ~int(ref)
                seen[t] += 1
                assert np.array_equal(bits(box[:3]), bits(lo[t])) and np.array_equal(bits(box[3:]), bits(hi[t]))
        own[k, :3] = np.minimum(nodes[k]["lbox"][:3], nodes[k]["rbox"][:3])
        own[k, 3:] = np.maximum(nodes[k]["lbox"][3:], nodes[k]["rbox"][3:])
    assert (seen == 1).all(), "every triangle must be in exactly one leaf"
    assert height == levels[0] + 1 and height <= MAX_HEIGHT
    # numbering: preorder ids over all nodes, tie-break rank = 2 * expansion order of the parent + side
    next_id, order = 0, 0
    todo = [("node", 0, 0)]
    want_rank = np.zeros(n_tris, np.uint32)
    while todo:
        kind, k, expect = todo.pop()
        if kind == "leaf":
            assert expect == next_id
            next_id += 1
            continue
        nd = nodes[k]
        assert nd["self_id"] == next_id == expect
        next_id += 1
        push = []
        for side, (ref, cid) in enumerate([(nd["lref"], nd["lid"]), (nd["rref"], nd["rid"])]):
            if ref < 0:
                want_rank[~int(ref)] = 2 * order + side
                push.append(("leaf", -1, int(cid)))
            else:
                push.append(("node", int(ref), int(cid)))
        order += 1
        todo.extend(reversed(push))
    assert next_id == 2 * n_tris - 1
    assert np.array_equal(rank, want_rank)


def random_rays(boxes, rng, n):
    lo, hi = boxes[0, :3], boxes[0, 3:]
    o = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def corner_rays(boxes, rng, n):
    """Aimed at corners of node boxes, a third starting on a box plane, some axis-parallel."""
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
    d[-40:-20, 0] = 0.0
    d[-20:, :2] = 0.0
    d[-20:, 2] = -1.0
    return o, d


def compare_with_oracle_on_same_tree(gpu, orc, boxes, n_rays, seed):
    rng = np.random.default_rng(seed)
    for o, d in (random_rays(boxes, rng, n_rays), corner_rays(boxes, rng, n_rays)):
        want = orc.trace(o, d)
        got = gpu.trace_rays(o, d)
        for key in ("prim", "vhash", "node_steps", "tri_tests"):
            assert np.array_equal(got[key], want[key]), key
        assert np.array_equal(bits(got["t"]), bits(want["t"]))
        try:
            for variant in (1, 0, 2, 3, 4):
                gpu.set_option("extend_variant", variant)
                tri, t = gpu.extend_rays(o, d)
                assert np.array_equal(tri, want["prim"]) and np.array_equal(bits(t), bits(want["t"])), "variant %d" % variant
        finally:
            gpu.set_option("extend_variant", 1)


@pytest.mark.parametrize("name", SCENES)
def test_device_built_tree_structure_and_traversal(name, gpu, host_scenes, oracle):
    hs = host_scenes(name)
    gpu.build(hs)
    nodes, rank, height = gpu.download_tree()
    check_structure(nodes, rank, height, hs.tris())
    boxes, topo = tree_table(nodes, hs.n_tris)
    orc = oracle.OracleScene(scene_path(name))
    orc.set_tree(boxes, topo)
    assert orc.height == height
    compare_with_oracle_on_same_tree(gpu, orc, boxes, 20000, 11)
    # building twice gives the same tree (no dependence on the order in which threads arrive)
    gpu.build(hs)
    nodes2, rank2, height2 = gpu.download_tree()
    assert nodes.tobytes() == nodes2.tobytes() and np.array_equal(rank, rank2) and height == height2


@pytest.mark.parametrize("name", ["cornell-box2-transmissive", "dragon-spec"])
def test_render_on_device_built_tree(name, gpu, host_scenes, oracle):
    import torch
    hs = host_scenes(name)
    w = h = 48
    spp, depth = 4, 5

    def render():
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        gpu.render_into(acc.data_ptr(), gpu.params(w, h, 0, spp, depth))
        gpu.synchronize()
        return acc.cpu().numpy()

    gpu.upload(hs)
    on_host_tree = render()
    gpu.build(hs)
    on_device_tree = render()
    nodes, _, _ = gpu.download_tree()
    orc = oracle.OracleScene(scene_path(name))
    orc.set_tree(*tree_table(nodes, hs.n_tris))
    want, _ = orc.render(w, h, 0, spp, depth)
    bad = int((~np.isclose(on_device_tree, want, rtol=1e-9, atol=1e-12).all(axis=2)).sum())
    assert bad <= 1, "%d pixels differ from the oracle rendering with the same tree" % bad
    # The random numbers of a path do not depend on the tree, so the two trees give the same path wherever no
    # hit is a tie between triangles or grazes a box plane: almost every pixel agrees to rounding.
    same = np.isclose(on_device_tree, on_host_tree, rtol=1e-9, atol=1e-12).all(axis=2).mean()
    assert same > 0.99, "only %.4f of the pixels agree between the two trees" % same


def _mesh_text(tri_points):
    # (the .ray number syntax has no '+': an exponent is written e39, not e+39)
    pts = ",".join(("(%r,%r,%r)" % tuple(float(x) for x in p)).replace("e+", "e") for tri in tri_points for p in tri)
    faces = ",".join("(%d,%d,%d)" % (3 * k, 3 * k + 1, 3 * k + 2) for k in range(len(tri_points)))
    return ("SBT-raytracer 1.0\ncamera { position = (0.5,0.5,4); viewdir = (0,0,-1); updir = (0,1,0); }\n"
            "polymesh { material = { emissive = (0.5,0.25,1.0); diffuse = (0.5,0.5,0.5); }; points = (%s); faces = (%s); }\n" % (pts, faces))


def _cell_of_code(code):
    x = y = z = 0
    for b in range(10):
        z |= ((code >> (3 * b)) & 1) << b
        y |= ((code >> (3 * b + 1)) & 1) << b
        x |= ((code >> (3 * b + 2)) & 1) << b
    return x, y, z


def _small_tri(center, size, rng):
    """A triangle whose bounding box is centred at `center`."""
    c = np.asarray(center, float)
    return [c + size * np.array([-1.0, -1.0, 0.0]), c + size * np.array([1.0, -1.0, 0.0]), c + size * np.array([0.0, 1.0, 0.0])]


def test_small_and_degenerate_inputs(gpu, oracle, tmp_path):
    from cuda_path_tracer_amd import HostScene
    rng = np.random.default_rng(3)
    cases = {
        "one": [_small_tri((0.5, 0.5, 0.0), 0.3, rng)],
        "two": [_small_tri((0.3, 0.5, 0.0), 0.2, rng), _small_tri((0.7, 0.5, 0.1), 0.2, rng)],
        # all centres equal: every Morton code ties, the position tie-break must give a balanced tree
        "coincident": [_small_tri((0.5, 0.5, 0.0), 0.3, rng) for _ in range(300)],
        "collinear": [_small_tri((0.01 * k, 0.5, 0.0), 0.004, rng) for k in range(97)],
    }
    for label, tri_points in cases.items():
        text = _mesh_text(tri_points)
        hs = HostScene(text=text)
        assert hs.n_tris == len(tri_points), label
        gpu.build(hs)
        nodes, rank, height = gpu.download_tree()
        if hs.n_tris == 1:
            assert len(nodes) == 0 and height == 1
        else:
            check_structure(nodes, rank, height, hs.tris())
        if label == "coincident":
            assert height <= 10  # ceil(log2 300) + 1
        path = tmp_path / (label + ".ray")
        path.write_text(text)
        orc = oracle.OracleScene(str(path))
        boxes, topo = tree_table(nodes, hs.n_tris)
        if hs.n_tris == 1:  # the root is the leaf: its box is the triangle's bounds
            t = hs.tris()[0]
            pts = np.stack([t["a"], t["b"], t["c"]])
            boxes[0, :3], boxes[0, 3:] = pts.min(axis=0), pts.max(axis=0) + np.array([0.0, 0.0, 0.5])
        orc.set_tree(boxes, topo)
        compare_with_oracle_on_same_tree(gpu, orc, boxes, 3000, 5)


def test_tall_morton_chain_is_rebuilt_with_coarser_codes(gpu, oracle, tmp_path):
    """One triangle per Morton prefix length makes a 30-level chain under full-resolution codes, and 200
    coincident triangles at its end add eight more levels: 39 levels, more than the traversal stacks hold.
    The height bound in check_structure therefore shows that the builder fell back to coarser codes, and the
    rest shows that what it returned is still a correct tree."""
    from cuda_path_tracer_amd import HostScene
    rng = np.random.default_rng(9)
    cells = [_cell_of_code(1 << k) for k in range(30)] + [(0, 0, 0)] * 200 + [(1023, 1023, 1023)]
    tri_points = [_small_tri(((x + 0.5) / 1024.0, (y + 0.5) / 1024.0, (z + 0.5) / 1024.0), 1e-5, rng) for (x, y, z) in cells]
    text = _mesh_text(tri_points)
    hs = HostScene(text=text)
    gpu.build(hs)
    nodes, rank, height = gpu.download_tree()
    check_structure(nodes, rank, height, hs.tris())
    path = tmp_path / "chain.ray"
    path.write_text(text)
    orc = oracle.OracleScene(str(path))
    boxes, topo = tree_table(nodes, hs.n_tris)
    orc.set_tree(boxes, topo)
    compare_with_oracle_on_same_tree(gpu, orc, boxes, 3000, 6)


@pytest.mark.parametrize("name", ["cornell-box2-transmissive", "dragon-spec"])
def test_device_built_tree_with_reference_ranks_gives_the_reference_image(name, gpu, host_scenes, oracle_scenes):
    """cpt_build_scene with CPT_SCENE_KEEP_RANKS: the Morton tree built on the GPU, the triangle records keeping the reference
    tree's tie-break ranks.  The closest hit depends on a tree only through its leaves and the ranks, so hit triangles and
    distances are the oracle's on the reference tree (tie-prone rays included) and the image is the reference tree's bit
    for bit, under the reference walk and the culling walk."""
    import torch
    from test_gpu_parity import _tie_prone_rays
    hs = host_scenes(name)
    orc = oracle_scenes(name)
    rng = np.random.default_rng(53)
    boxes, _ = orc.nodes()
    o1, d1 = random_rays(boxes, rng, 8000)
    o2, d2 = _tie_prone_rays(orc, rng, 8000)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    want = orc.trace(o, d)
    w, h, spp, depth = 64, 48, 4, 6
    gpu.upload(hs)
    ref_img = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    gpu.render_into(ref_img.data_ptr(), gpu.params(w, h, 0, spp, depth))
    gpu.synchronize()
    gpu.build(hs, keep_ranks=True)
    try:
        for variant in (1, 2):
            gpu.set_option("extend_variant", variant)
            tri, t = gpu.extend_rays(o, d)
            assert np.array_equal(tri, want["prim"]) and np.array_equal(bits(t), bits(want["t"])), variant
            img = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
            gpu.render_into(img.data_ptr(), gpu.params(w, h, 0, spp, depth))
            gpu.synchronize()
            assert torch.equal(img.view(torch.int64), ref_img.view(torch.int64)), variant
    finally:
        gpu.set_option("extend_variant", 1)


def test_uploaded_tree_without_nested_boxes_runs_the_binary32_variants_as_exact_ones(gpu, host_scenes, oracle):
    """Extend variants 3 and 4 return variant 1's hits only on a tree whose child boxes lie inside their parents'
    boxes and whose leaf boxes are exactly their triangles' bounds.  A caller's tree with tightened leaf boxes (here:
    every leaf box shrunk to its lower half along x, so rays that pierce only the other half must miss the triangle)
    is valid for the reference's walk; the library accepts it, notices that the binary32 variants' assumptions do
    not hold (nor the culling variants': a hit may lie before the point where the ray enters the tightened box), and
    runs all of them as the reference walk: hits, distances and the very same node and triangle counts as the oracle
    walking that tree."""
    import ctypes as C
    from cuda_path_tracer_amd import _capi
    hs = host_scenes("teapot-trans")
    nodes = hs.nodes().copy()
    for side, ref in (("lbox", "lref"), ("rbox", "rref")):
        leaf = nodes[ref] < 0
        mid = 0.5 * (nodes[side][leaf, 0] + nodes[side][leaf, 3])
        b = nodes[side][leaf]
        b[:, 3] = mid
        nodes[side][leaf] = b

    class Shim:
        pass
    sh = Shim()
    sh.keep = np.ascontiguousarray(nodes)
    sh.desc = _capi.SceneDesc()
    C.memmove(C.byref(sh.desc), C.byref(hs.desc), C.sizeof(sh.desc))
    sh.desc.nodes = C.cast(sh.keep.ctypes.data, C.POINTER(_capi.Node))
    sh.n_tree_nodes = hs.n_tree_nodes
    gpu.upload(sh)
    boxes, topo = tree_table(nodes, hs.n_tris)
    orc = oracle.OracleScene(scene_path("teapot-trans"))
    orc.set_tree(boxes, topo)
    rng = np.random.default_rng(31)
    o, d = random_rays(boxes, rng, 20000)
    want = orc.trace(o, d)
    whole = oracle.OracleScene(scene_path("teapot-trans")).trace(o, d)
    assert (want["prim"] != whole["prim"]).sum() > 50, "the tightened boxes should change some hits"
    try:
        for variant in (1, 3, 4, 2):
            gpu.set_option("extend_variant", variant)
            gpu.reset_stats()
            gpu.set_instrumentation(False, True)
            tri, t = gpu.extend_rays(o, d)
            st = gpu.stats()
            assert np.array_equal(tri, want["prim"]) and np.array_equal(bits(t), bits(want["t"])), "variant %d" % variant
            if True:  # every variant has fallen back to the reference walk
                assert st["node_steps"] == int(want["node_steps"].sum()) and st["tri_tests"] == int(want["tri_tests"].sum())
    finally:
        gpu.set_option("extend_variant", 1)
        gpu.set_instrumentation(False, False)
        gpu.reset_stats()


def test_cli_with_device_built_tree(tmp_path, gpu, host_scenes):
    """bin/ray -g --device-bvh skips the host build and writes the image the library renders on its own tree."""
    import os
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    out = str(tmp_path / "o.png")
    r = subprocess.run([exe, "-g", "--device-bvh", "-s", "4", "-r", "3", "-w", "64", scene_path("teapot-trans"), out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.asarray(Image.open(out).convert("RGB"))
    gpu.build(host_scenes("teapot-trans"))
    want = gpu.render_image(gpu.params(64, 64, 0, 4, 3))
    assert np.array_equal(got, want[::-1])  # PNG rows are top-down, the buffer is bottom-up


def test_large_triangle_soup(gpu):
    """100 000 triangles through the C ABI: large enough for the sort's multi-block scan.  The tree is checked
    structurally, and rays aimed at triangle centres must hit at or before their target."""
    from cuda_path_tracer_amd import ArrayScene
    n = 100000
    rng = np.random.default_rng(2)
    centre = rng.uniform(-1, 1, (n, 1, 3))
    hs = ArrayScene(centre + rng.normal(size=(n, 3, 3)) * 0.01)
    gpu.build(hs)
    nodes, rank, height = gpu.download_tree()
    check_structure(nodes, rank, height, hs.tris())
    m = 5000
    pick = rng.integers(0, n, m)
    tris = hs.tris()
    target = (tris["a"][pick] + tris["b"][pick] + tris["c"][pick]) / 3.0
    o = target + rng.normal(size=(m, 3)) * 3.0
    d = target - o
    dist = np.linalg.norm(d, axis=1)
    d /= dist[:, None]
    tri, t = gpu.extend_rays(o, d)
    assert ((tri >= 0) & (t <= dist * (1 + 1e-9))).all()
    got = gpu.trace_rays(o, d)
    assert np.array_equal(got["prim"], tri) and np.array_equal(bits(got["t"]), bits(t))


def test_refit_keeps_the_tree_and_redoes_its_boxes(gpu, oracle, tmp_path):
    """cpt_refit_scene: the same triangles with moved vertices.  Topology, numbering and ranks stay, every box
    becomes the exact bound of the moved geometry below it, and the oracle walking that tree over the moved
    triangles agrees bit for bit -- for a tree that came from the host and for one built on the device."""
    import torch
    from cuda_path_tracer_amd import CptError, HostScene, Renderer
    rng = np.random.default_rng(21)
    n = 400
    centre = rng.uniform(0, 1, (n, 1, 3))
    before = centre + rng.normal(size=(n, 3, 3)) * 0.03
    after = before + 0.05 * np.sin(7.0 * before[:, :, [1, 2, 0]]) + rng.normal(size=(n, 1, 3)) * 0.01  # bend and jiggle
    text_a, text_b = _mesh_text(list(before)), _mesh_text(list(after))
    hs_a, hs_b = HostScene(text=text_a), HostScene(text=text_b)
    assert hs_a.n_tris == hs_b.n_tris == n
    path_b = tmp_path / "after.ray"
    path_b.write_text(text_b)
    for place in (gpu.upload, gpu.build):
        place(hs_a)
        nodes_a, rank_a, height_a = gpu.download_tree()
        gpu.refit(hs_b)
        nodes_b, rank_b, height_b = gpu.download_tree()
        for key in ("lref", "rref", "lid", "rid", "self_id"):
            assert np.array_equal(nodes_a[key], nodes_b[key])
        assert np.array_equal(rank_a, rank_b) and height_a == height_b
        assert not np.array_equal(nodes_a["lbox"], nodes_b["lbox"])
        check_structure(nodes_b, rank_b, height_b, hs_b.tris())
        orc = oracle.OracleScene(str(path_b))
        boxes, topo = tree_table(nodes_b, n)
        orc.set_tree(boxes, topo)
        compare_with_oracle_on_same_tree(gpu, orc, boxes, 4000, 8)
        w = h = 40
        acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
        gpu.render_into(acc.data_ptr(), gpu.params(w, h, 0, 3, 3))
        gpu.synchronize()
        want, _ = orc.render(w, h, 0, 3, 3)
        assert np.allclose(acc.cpu().numpy(), want, rtol=1e-9, atol=1e-12)
    with pytest.raises(CptError):
        gpu.refit(HostScene(text=_mesh_text(list(after[:-1]))))  # another triangle count
    fresh = Renderer(0)
    try:
        with pytest.raises(CptError):
            fresh.refit(hs_b)  # nothing to refit
    finally:
        fresh.close()


def test_binary32_inner_nodes_keep_the_hits_on_hostile_geometry(gpu, oracle, tmp_path):
    """Extend variants 3 and 4 test inner nodes in binary32 with error margins; their claim is that the hits are those of
    the exact walk whatever the geometry.  Scenes chosen against the margins: a triangle soup a few million units from
    the origin whose sizes span six orders of magnitude (the planes lose up to a quarter of a unit to binary32), and a
    stack of axis-aligned sheets (flat boxes, many coplanar) with rays that start on the sheets' planes, run inside
    them, have direction components of 1e-12, 1e-200 and exactly 0, or start 1e12 away.  Hit triangle and t bits must
    equal the oracle's and variant 1's; on the host-built (reference) tree and on the device-built one."""
    from cuda_path_tracer_amd import HostScene
    rng = np.random.default_rng(17)
    centre = np.array([3.0e6, -2.0e6, 4.0e5])
    soup = []
    for k in range(2500):
        size = 10.0 ** rng.uniform(-3, 3)
        c = centre + rng.normal(size=3) * 2000.0
        soup.append([c + rng.normal(size=3) * size for _ in range(3)])
    sheets = []
    for layer in range(12):
        z = 0.25 * layer
        for i in range(6):
            for j in range(6):
                x0, y0 = float(i), float(j)
                sheets.append([np.array([x0, y0, z]), np.array([x0 + 1, y0, z]), np.array([x0, y0 + 1, z])])
                sheets.append([np.array([x0 + 1, y0 + 1, z]), np.array([x0, y0 + 1, z]), np.array([x0 + 1, y0, z])])
    for label, tri_points in (("soup", soup), ("sheets", sheets)):
        text = _mesh_text(tri_points)
        path = tmp_path / (label + ".ray")
        path.write_text(text)
        hs = HostScene(text=text)
        orc = oracle.OracleScene(str(path))
        boxes, _ = orc.nodes()
        lo, hi = boxes[0, :3], boxes[0, 3:]
        o1, d1 = random_rays(boxes, rng, 6000)
        o2, d2 = corner_rays(boxes, rng, 6000)
        # rays that start on box planes and run along them, with tiny and zero direction components
        n = 3000
        o3 = rng.uniform(lo, hi, (n, 3))
        pick = boxes[rng.integers(0, len(boxes), n)]
        ax = rng.integers(0, 3, n)
        o3[np.arange(n), ax] = pick[np.arange(n), ax]
        d3 = rng.normal(size=(n, 3))
        tiny = np.array([0.0, 1e-12, -1e-12, 1e-200, -1e-200, 1e-300])
        d3[np.arange(n), ax] = tiny[rng.integers(0, len(tiny), n)]
        d3[: n // 2] /= np.linalg.norm(d3[: n // 2], axis=1, keepdims=True)
        # rays from very far away, aimed at the scene
        o4 = (lo + hi) / 2 + rng.normal(size=(500, 3)) * 1e12
        d4 = (rng.uniform(lo, hi, (500, 3)) - o4)
        d4 /= np.linalg.norm(d4, axis=1, keepdims=True)
        o = np.vstack([o1, o2, o3, o4])
        d = np.vstack([d1, d2, d3, d4])
        want = orc.trace(o, d)
        assert (want["prim"] >= 0).sum() > 200, label
        for build in ("host", "device"):
            (gpu.upload if build == "host" else gpu.build)(hs)
            try:
                hits = {}
                for variant in (1, 3, 4):
                    gpu.set_option("extend_variant", variant)
                    hits[variant] = gpu.extend_rays(o, d)
            finally:
                gpu.set_option("extend_variant", 1)
            if build == "host":  # the reference's tree: the oracle's hits
                assert np.array_equal(hits[1][0], want["prim"]) and np.array_equal(bits(hits[1][1]), bits(want["t"])), label
            for variant in (3, 4):
                assert np.array_equal(hits[variant][0], hits[1][0]), (label, build, variant)
                assert np.array_equal(bits(hits[variant][1]), bits(hits[1][1])), (label, build, variant)


def test_binary32_inner_nodes_with_planes_beyond_the_binary32_range(gpu, oracle, tmp_path):
    """Two triangles near 1e39 put box planes beyond the largest float.  The binary32 variants (3 and 4) then cannot
    constrain any axis and must leave every decision to the exact tests of the leaves: same hits as variant 1 and the
    oracle, also for rays with zero direction components (where 0 * inf would turn into a NaN that minimum and maximum
    drop silently)."""
    from cuda_path_tracer_amd import HostScene
    rng = np.random.default_rng(23)
    tris = []
    for i in range(5):
        for j in range(5):
            x0, y0, z = float(i), float(j), 0.1 * ((i + j) % 3)
            tris.append([np.array([x0, y0, z]), np.array([x0 + 1, y0, z]), np.array([x0, y0 + 1, z])])
            tris.append([np.array([x0 + 1, y0 + 1, z]), np.array([x0, y0 + 1, z]), np.array([x0 + 1, y0, z])])
    far = 1.0e39
    tris.append([np.array([far, far, far]), np.array([far * 1.5, far, far]), np.array([far, far * 1.5, far])])
    tris.append([np.array([-far, -far, -far]), np.array([-far * 1.5, -far, -far]), np.array([-far, -far * 1.5, -far])])
    text = _mesh_text(tris)
    path = tmp_path / "beyond.ray"
    path.write_text(text)
    hs = HostScene(text=text)
    orc = oracle.OracleScene(str(path))
    n = 4000
    o = np.column_stack([rng.uniform(-1, 6, n), rng.uniform(-1, 6, n), rng.uniform(1, 3, n)])
    d = np.column_stack([rng.normal(size=n) * 0.3, rng.normal(size=n) * 0.3, -np.ones(n)])
    d[:1000, 0] = 0.0
    d[500:1500, 1] = 0.0
    d[3000:] *= -1.0  # away from the sheets: misses
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    want = orc.trace(o, d)
    assert (want["prim"] >= 0).sum() > 1000 and (want["prim"] < 0).sum() > 500
    for build in ("host", "device"):
        (gpu.upload if build == "host" else gpu.build)(hs)
        try:
            hits = {}
            for variant in (1, 3, 4):
                gpu.set_option("extend_variant", variant)
                hits[variant] = gpu.extend_rays(o, d)
        finally:
            gpu.set_option("extend_variant", 1)
        if build == "host":
            assert np.array_equal(hits[1][0], want["prim"]) and np.array_equal(bits(hits[1][1]), bits(want["t"]))
        for variant in (3, 4):
            assert np.array_equal(hits[variant][0], hits[1][0]), (build, variant)
            assert np.array_equal(bits(hits[variant][1]), bits(hits[1][1])), (build, variant)


def test_scene_sizes_around_the_record_cache_capacity(gpu):
    """The persistent extend kernel runs another template instance when the whole scene fits its LDS record cache: in the
    exact variants 168 bytes per node and 112 per triangle in 79 872 bytes (67 584 with culling), which a device-built
    tree over n triangles (n - 1 nodes) fits up to n = 285 (241); in the binary32 ones 512 slots of 128 bytes, a node
    record taking half a slot, up to n = 341.  A scene beyond that keeps a prefix of each array there.  Triangle counts
    on both sides of every such boundary must give the hits of the one-ray-per-thread kernel, bit for bit."""
    from cuda_path_tracer_amd import HostScene
    rng = np.random.default_rng(41)
    for n in (240, 241, 242, 255, 256, 257, 284, 285, 286, 287, 340, 341, 342, 343, 600):
        tris = [_small_tri(rng.uniform(-1, 1, 3), 10.0 ** rng.uniform(-2, -0.5), rng) for _ in range(n)]
        gpu.build(HostScene(text=_mesh_text(tris)))
        o = rng.uniform(-1.3, 1.3, (6000, 3))
        d = rng.normal(size=(6000, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        try:
            hits = {}
            for variant in (0, 1, 2, 3, 4):
                gpu.set_option("extend_variant", variant)
                hits[variant] = gpu.extend_rays(o, d)
        finally:
            gpu.set_option("extend_variant", 1)
        assert (hits[0][0] >= 0).sum() > 300, n
        for variant in (1, 2, 3, 4):
            assert np.array_equal(hits[variant][0], hits[0][0]), (n, variant)
            assert np.array_equal(bits(hits[variant][1]), bits(hits[0][1])), (n, variant)
