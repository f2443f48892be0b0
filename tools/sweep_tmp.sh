python - <<'PY' || exit 1
import sys, numpy as np
sys.path.insert(0, '.')
from cuda_path_tracer_amd import HostScene, Renderer
from oracle import pyoracle
for name in ["cornell-box2-transmissive", "dragon-trans"]:
    path = "tests/golden/scenes/%s.ray" % name
    r = Renderer(0); r.upload(HostScene(path)); orc = pyoracle.OracleScene(path)
    rng = np.random.default_rng(3)
    boxes, _ = orc.nodes(); lo, hi = boxes[0, :3], boxes[0, 3:]
    n = 100000
    o = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    want = orc.trace(o, d)
    r.set_option("extend_variant", 2); r.set_instrumentation(False, True); r.reset_stats()
    tri, t = r.extend_rays(o, d); st = r.stats()
    print(name, "tri equal", np.array_equal(tri, want["prim"]), "mismatch", int((tri != want["prim"]).sum()), "t equal", np.array_equal(t.view(np.uint64), want["t"].view(np.uint64)),
          "node steps %d vs %d, tri tests %d vs %d" % (st["node_steps"], want["node_steps"].sum(), st["tri_tests"], want["tri_tests"].sum()))
    r.close()
PY
for v in 1 2; do for l in 8 32; do
echo "== variant $v leaf_go $l"
timeout -k 5 100 python tools/prof_render.py --spp 64 --depth 8 --repeat 2 --stats --opt extend_variant=$v --opt pw_leaf_go=$l | tail -2 || exit 1
timeout -k 5 100 python tools/prof_render.py --scene dragon-spec --width 1024 --spp 16 --depth 8 --repeat 2 --stats --opt extend_variant=$v --opt pw_leaf_go=$l | tail -2 || exit 1
done; done
