"""CPU tests of the host scene library (libcpt_host.so): .ray reader, BVH build, flattening, image
writer -- against the oracle's independent restatement and against hand-written cases."""
import os

import numpy as np
import pytest

from conftest import scene_path

ALL = ["cornell-box", "cornell-box2", "cornell-box2-specular", "cornell-box2-transmissive", "teapot", "teapot-trans",
       "dragon-spec", "dragon-trans", "3-dragon-spec"]


@pytest.mark.parametrize("name", ALL)
def test_flat_scene_matches_oracle(name, host_scenes, oracle_scenes):
    hs, orc = host_scenes(name), oracle_scenes(name)
    boxes, topo = hs.tree()
    oboxes, otopo = orc.nodes()
    assert np.array_equal(topo, otopo), "tree topology differs from the reference build"
    assert boxes.tobytes() == oboxes.tobytes(), "node boxes differ bitwise"
    assert hs.tree_height == orc.height
    geom, vn, ids = orc.prims()
    t = hs.tris()
    assert np.hstack([t["a"], t["b"], t["c"], t["n"]]).tobytes() == geom.tobytes()
    assert np.array_equal(t["material"], ids[:, 0])
    assert np.array_equal(t["flags"] & 1, ids[:, 1]) and np.array_equal((t["flags"] >> 1) & 1, ids[:, 2])
    tn = hs.tri_normals()
    assert np.hstack([tn["na"], tn["nb"], tn["nc"]]).tobytes() == vn.tobytes()
    vals, flags = orc.materials()
    m = hs.materials()
    assert np.hstack([m["ke"], m["kd"], m["ks"], m["kr"], m["kt"], m["index"][:, None]]).tobytes() == vals.tobytes()
    assert np.array_equal(m["flags"], flags)
    assert hs.camera().tobytes() == orc.camera().tobytes()


@pytest.mark.parametrize("name", ["cornell-box2", "dragon-spec"])
def test_node_records(name, host_scenes):
    hs = host_scenes(name)
    nd = hs.nodes()
    boxes, topo = hs.tree()
    assert len(nd) == hs.n_tris - 1
    assert nd["self_id"][0] == 0
    seen_tris = []
    for k in range(len(nd)):
        me = nd["self_id"][k]
        assert topo[me, 0] == nd["lid"][k] and topo[me, 1] == nd["rid"][k]
        assert np.array_equal(nd["lbox"][k], boxes[nd["lid"][k]]) and np.array_equal(nd["rbox"][k], boxes[nd["rid"][k]])
        for ref, cid in ((nd["lref"][k], nd["lid"][k]), (nd["rref"][k], nd["rid"][k])):
            if ref >= 0:
                assert ref > k and nd["self_id"][ref] == cid  # children come after their parent
            else:
                assert topo[cid, 2] == ~ref
                seen_tris.append(~ref)
    assert sorted(seen_tris) == list(range(hs.n_tris))


def test_reader_grammar_details():
    from cuda_path_tracer_amd import HostScene
    text = """SBT-raytracer 1.0
    // line comment
    /* block
       comment */
    camera { position = (0,0,4); viewdir = (0,0,-1); updir = (0,1,0); fov = 45; aspectratio = 2 }
    point_light { position = (1,1,1); colour = (1,1,1); constant_attenuation_coeff = 0.25; }
    material = { diffuse = (0.5,0.5,0.5); name = "grey"; }
    sphere { material = { diffuse = (1,0,0); } }
    translate( 1, 2, 3,
      scale( 2,
        polymesh { name = "quad"; points = ((0,0,0),(1,0,0),(1,1,0),(0,1,0)); faces = ((0,1,2,3)); gennormals; } ) )
    polymesh { material = "grey"; points = ((0,0,0),(1,0,0),(1,0,0),(0,1,-1e-1)); faces = ((0,1,2),(0,1,3)); }
    """
    hs = HostScene(text=text)
    assert hs.aspect == 2.0 and hs.height_for_width(512) == 256
    assert hs.n_tris == 3          # the quad fans into two triangles; (0,1,2) of the second mesh is degenerate
    t = hs.tris()
    # translate(1,2,3) o scale(2) applied to the quad
    assert np.array_equal(t["a"][0], [1, 2, 3]) and np.array_equal(t["b"][0], [3, 2, 3]) and np.array_equal(t["c"][1], [1, 4, 3])
    assert np.allclose(t["n"][0], [0, 0, 1])
    assert (t["flags"][:2] & 2).all() and not (t["flags"][2] & 2)      # gennormals only on the first mesh
    m = hs.materials()
    assert np.array_equal(m["kd"][0], [0.5, 0.5, 0.5]) and np.array_equal(m["kd"][1], [0.5, 0.5, 0.5])
    assert any("sphere" in w for w in hs.warnings) and any("light" in w for w in hs.warnings)
    cam = hs.camera()
    assert np.allclose(cam[6:9], [2 * 2 * np.tan(np.radians(45) / 2), 0, 0], rtol=1e-9)  # u = x * height * aspect


@pytest.mark.parametrize("text,fragment", [
    ("camera {}", "SBT-raytracer"),
    ("SBT-raytracer 2.0", "version"),
    ("SBT-raytracer 1.0 polymesh { points = ((0,0,0)); faces = ((0,1,2)); }", "bad face"),
    ("SBT-raytracer 1.0 polymesh { points = ((0,0,0),(1,0,0),(0,1,0)); faces = ((0,1)); }", "at least 3"),
    ("SBT-raytracer 1.0 camera { viewdir = (0,0,-1); }", "updir"),
    ("SBT-raytracer 1.0 polymesh { material = { diffuse = map(\"x.png\"); } }", "texture"),
    ("SBT-raytracer 1.0 polymesh { points = ((0,0,0),(1,0,0),(0,1,0)); faces = ((0,1,2)); normals = ((0,0,1)); }", "normals"),
    ("SBT-raytracer 1.0 camera { position = (0,0,1); }", "no polymesh"),
    ("SBT-raytracer 1.0 /* never closed", "comment"),
])
def test_reader_errors(text, fragment):
    from cuda_path_tracer_amd import HostScene
    with pytest.raises(RuntimeError) as e:
        HostScene(text=text)
    assert fragment in str(e.value)


def test_missing_file():
    from cuda_path_tracer_amd import HostScene
    with pytest.raises(RuntimeError):
        HostScene("/nonexistent/scene.ray")


def test_image_writer_round_trip(tmp_path):
    from PIL import Image
    from cuda_path_tracer_amd import write_image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)  # row 0 = bottom
    for ext in ("png", "bmp"):
        p = str(tmp_path / ("x." + ext))
        write_image(p, img)
        back = np.asarray(Image.open(p).convert("RGB"))
        assert np.array_equal(back, img[::-1]), ext  # files are top-down
    with pytest.raises(RuntimeError):
        write_image(str(tmp_path / "x.jpg"), img)


def test_build_is_fast_and_deterministic(host_scenes):
    import time
    from cuda_path_tracer_amd import HostScene
    t0 = time.time()
    a = HostScene(scene_path("dragon-trans"))
    assert time.time() - t0 < 5.0
    b = host_scenes("dragon-trans")
    assert a.tree()[1].tobytes() == b.tree()[1].tobytes()


def test_fast_tree_is_a_tree_over_the_same_leaves(oracle):
    """The fast mode's host-built tree (surface-area build): every triangle in exactly one leaf with the exact bounds of
    its vertices, every inner box the exact union of its children's, preorder numbering, bounded height; the triangle
    records carry the REFERENCE tree's tie-break ranks; and the oracle walking that tree finds the same hit distances
    as on the reference tree from fewer box tests."""
    import numpy as np
    from conftest import scene_path
    from cuda_path_tracer_amd import HostScene
    for name in ["cornell-box2", "teapot-trans", "dragon-spec"]:
        hs = HostScene(scene_path(name))
        f = hs.fast()
        assert f.desc.flags == 1 and f.n_tris == hs.n_tris and f.n_nodes == hs.n_tris - 1
        boxes, topo = hs.fast_tree()
        n = len(boxes)
        assert n == 2 * hs.n_tris - 1 and f.tree_height <= 32
        tris = hs.tris()
        pts = np.stack([tris["a"], tris["b"], tris["c"]], 1)
        seen = np.zeros(hs.n_tris, int)
        for k in range(n - 1, -1, -1):  # children come after their parents in preorder
            l, r, prim = topo[k]
            if prim >= 0:
                assert l < 0 and r < 0
                seen[prim] += 1
                assert np.array_equal(boxes[k, :3], pts[prim].min(0)) and np.array_equal(boxes[k, 3:], pts[prim].max(0))
            else:
                assert l == k + 1 and r > l
                assert np.array_equal(boxes[k, :3], np.minimum(boxes[l, :3], boxes[r, :3]))
                assert np.array_equal(boxes[k, 3:], np.maximum(boxes[l, 3:], boxes[r, 3:]))
        assert (seen == 1).all()
        # ranks: those cpt_upload_scene would derive from the reference tree
        ref_boxes, ref_topo = hs.tree()
        rank = np.zeros(hs.n_tris, np.uint32)
        todo, order = [0], 0
        while todo:
            k = todo.pop()
            l, r, _ = ref_topo[k]
            if ref_topo[l, 2] >= 0:
                rank[ref_topo[l, 2]] = 2 * order
            if ref_topo[r, 2] >= 0:
                rank[ref_topo[r, 2]] = 2 * order + 1
            if ref_topo[r, 2] < 0:
                todo.append(r)
            if ref_topo[l, 2] < 0:
                todo.append(l)
            order += 1
        assert np.array_equal(tris["test_rank"], rank)
        orc = oracle.OracleScene(scene_path(name))
        o, d = orc.camera_rays(40, 40)
        a = orc.trace(o, d)
        orc.set_tree(boxes, topo)
        b = orc.trace(o, d)
        assert np.array_equal(a["t"].view(np.uint64), b["t"].view(np.uint64))
        assert b["node_steps"].sum() < a["node_steps"].sum()
