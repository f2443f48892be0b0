"""cpt_render_noalloc (working memory from the library's static device array) against cpt_render.

Both entry points run the same kernels on the same paths and add a pixel's samples in sample order, so the
accumulation buffers must agree bit for bit -- the bar of every integer/bit test in this suite -- whatever batch
sizes the two pick.  Collected last (file name), so a failure here hides no other test.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(g, torch, w, h, s0, s1, depth, **kw):
    want = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    got = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.render_into(want.data_ptr(), g.params(w, h, s0, s1, depth, **kw))
    g.render_into_noalloc(got.data_ptr(), g.params(w, h, s0, s1, depth, **kw))
    g.synchronize()
    assert bool(torch.isfinite(want).all()) and float(want.sum()) > 0.0
    assert torch.equal(want.view(torch.int64), got.view(torch.int64))
    return got


def test_noalloc_render_equals_render_twice_in_a_row_second_call_smaller(gpu_scene):
    import torch
    g = gpu_scene("cornell-box2-transmissive")
    # first call: 512x512, 5 samples at depth 8: two samples fit the static array at a time, so three batches
    _both(g, torch, 512, 512, 0, 5, 8)
    # second call, smaller input, ragged sides, other sample range: the array still holds the first call's contents
    _both(g, torch, 77, 43, 3, 9, 5)
    # a tile window and the thin lens go through the same carving
    _both(g, torch, 200, 120, 0, 4, 6, tile_begin=7, tile_count=150, dof=True, focal_distance=8.1, aperture=0.6)


def test_noalloc_on_a_scene_that_spills_its_stack_and_misses_the_lds_cache(gpu_scene):
    import torch
    g = gpu_scene("dragon-spec")  # 10 005 triangles, tree height 24: spill columns and global record fetches in use
    _both(g, torch, 256, 256, 0, 6, 8)
    _both(g, torch, 64, 48, 2, 5, 3)


def test_noalloc_refuses_what_does_not_fit(gpu_scene):
    import torch
    from cuda_path_tracer_amd import CptError
    from cuda_path_tracer_amd._capi import NOALLOC_SCRATCH_BYTES
    g = gpu_scene("cornell-box2")
    w = h = 1024  # 1 048 576 path slots of 209 bytes per sample: beyond the 128 MiB array
    assert w * h * 209 > NOALLOC_SCRATCH_BYTES
    acc = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    with pytest.raises(CptError, match="static scratch"):
        g.render_into_noalloc(acc.data_ptr(), g.params(w, h, 0, 1, 3))
    g.synchronize()
    assert float(acc.abs().sum()) == 0.0  # refused before anything was launched
    with pytest.raises(CptError, match="wavefront"):
        g.render_into_noalloc(acc.data_ptr(), g.params(64, 64, 0, 1, 3, mode=1))
    # ... and the same image goes through as four tile windows, equal to cpt_render's
    n_tiles = (w // 8) * (h // 8)
    want = torch.zeros(h, w, 3, dtype=torch.float64, device="cuda")
    g.render_into(want.data_ptr(), g.params(w, h, 0, 2, 3))
    for q in range(4):
        g.render_into_noalloc(acc.data_ptr(), g.params(w, h, 0, 2, 3, tile_begin=q * n_tiles // 4, tile_count=n_tiles // 4))
    g.synchronize()
    assert torch.equal(want.view(torch.int64), acc.view(torch.int64))
