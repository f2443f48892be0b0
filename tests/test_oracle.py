

def test_oracle_accepts_a_tree_built_elsewhere(oracle):
    """set_tree with the oracle's own table reproduces its traversal results; a table that is not a tree over the
    scene's triangles is refused."""
    import numpy as np
    from conftest import scene_path
    a = oracle.OracleScene(scene_path("teapot"))
    b = oracle.OracleScene(scene_path("teapot"))
    boxes, topo = a.nodes()
    b.set_tree(boxes, topo)
    o, d = a.camera_rays(24, 24)
    ra, rb = a.trace(o, d), b.trace(o, d)
    for key in ("prim", "vhash", "node_steps", "tri_tests"):
        assert np.array_equal(ra[key], rb[key])
    assert np.array_equal(ra["t"].view(np.uint64), rb["t"].view(np.uint64))
    bad = topo.copy()
    leaf_rows = np.nonzero(topo[:, 2] >= 0)[0]
    bad[leaf_rows[0], 2] = bad[leaf_rows[1], 2]  # one triangle in two leaves
    import pytest
    with pytest.raises(ValueError):
        b.set_tree(boxes, bad)
    with pytest.raises(ValueError):
        b.set_tree(boxes[:-2], topo[:-2])
