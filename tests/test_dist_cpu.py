"""Multi-GPU sharding logic on CPU: two gloo ranks each 'render' their share of the sample indices
(the oracle stands in for the GPU), the radiance sums are reduced to rank 0 and must equal the
single-process render of all samples."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def test_sample_range_partitions():
    from cuda_path_tracer_amd.dist import sample_range
    for world in (1, 2, 3, 8):
        for count in (0, 1, 7, 8, 1000):
            parts = [sample_range(r, world, 5, count) for r in range(world)]
            assert parts[0][0] == 5 and parts[-1][1] == 5 + count
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from cuda_path_tracer_amd.dist import sample_range, reduce_radiance
from oracle import pyoracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sc = pyoracle.OracleScene(os.path.join(%(root)r, "tests", "golden", "scenes", "cornell-box2-transmissive.ray"))
w, h, spp, depth = 20, 14, 7, 4
b, e = sample_range(rank, world, 0, spp)
acc, _ = sc.render(w, h, b, e, depth)
t = torch.from_numpy(acc)
reduce_radiance(t, dst=0)
if rank == 0:
    np.save(sys.argv[1], t.numpy())
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_reduce_equals_single_render(tmp_path, oracle_scenes):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    out = tmp_path / "sum.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29513", str(script), str(out)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    want, _ = oracle_scenes("cornell-box2-transmissive").render(20, 14, 0, 7, 4)
    # same sample set; only the order of the last addition differs
    assert np.allclose(got, want, rtol=1e-13, atol=1e-15)
    assert got.sum() > 0


def test_tile_range_partitions_the_tiles():
    from cuda_path_tracer_amd.dist import tile_range
    for (w, h, world) in [(512, 512, 8), (37, 23, 4), (8, 8, 3), (1000, 3, 7), (16, 8, 5)]:
        n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
        covered = []
        for r in range(world):
            share = tile_range(r, world, w, h)
            if share is None:  # more ranks than tiles: the rank renders nothing (a count of 0 would mean "whole image")
                assert r >= n_tiles
                continue
            b, c = share
            assert 0 <= b and c >= 1 and b + c <= n_tiles
            covered.extend(range(b, b + c))
        assert covered == list(range(n_tiles))  # contiguous, disjoint, complete, in rank order


def test_shard_weak_and_strong():
    from cuda_path_tracer_amd.dist import shard
    # weak: every rank adds spp new indices
    assert [shard(r, 4, 16, 64, 64, "weak")["sample_begin"] for r in range(4)] == [0, 16, 32, 48]
    # strong, enough samples: the samples are dealt out, whole image each
    parts = [shard(r, 3, 16, 64, 64) for r in range(3)]
    assert [(p["sample_begin"], p["sample_end"]) for p in parts] == [(0, 6), (6, 11), (11, 16)]
    assert all(p["tile_count"] == 0 for p in parts)
    # strong, fewer samples than ranks: by area, every rank all samples; ranks beyond the tiles get nothing
    parts = [shard(r, 4, 2, 16, 8) for r in range(4)]
    assert parts[2] is None and parts[3] is None
    assert [(p["tile_begin"], p["tile_count"], p["sample_begin"], p["sample_end"]) for p in parts[:2]] == [(0, 1, 0, 2), (1, 1, 0, 2)]
