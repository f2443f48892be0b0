"""Multi-GPU sharding of one render: one process per GPU, samples-per-pixel split across ranks (or, when there
are fewer samples than ranks, the image's tiles), per-rank radiance sums combined with one reduce (RCCL over
xGMI when the tensors are on GPUs).

The reference is single-GPU; this has no counterpart there.  Because every random number is
addressed by (pixel, sample, bounce), rank r rendering samples [b_r, e_r) produces exactly the
paths a single GPU would have produced for those sample indices: the image does not depend on the
number of GPUs except for the order of the final fp64 additions.
"""


def sample_range(rank, world, first, count):
    """Contiguous share of sample indices [first, first+count) for `rank` of `world`: the first
    count % world ranks get one extra sample."""
    if world < 1 or not (0 <= rank < world) or count < 0:
        raise ValueError("bad rank/world/count")
    base, extra = divmod(count, world)
    begin = first + rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def tile_range(rank, world, width, height):
    """Share of the image's 8x8-pixel tiles for `rank` of `world`, as (tile_begin, tile_count) for
    Renderer.params: the way to split one render by area instead of by samples (fewer samples than GPUs, or a
    single sample to be finished sooner).  Tiles are numbered row by row; ranks get contiguous runs.
    A rank beyond the number of tiles has nothing to render and gets None (a tile_count of 0 would mean the
    whole image to cpt_render)."""
    n_tiles = ((width + 7) // 8) * ((height + 7) // 8)
    begin, end = sample_range(rank, world, 0, n_tiles)
    if end == begin:
        return None
    return begin, end - begin


def shard(rank, world, spp, width, height, scaling="strong"):
    """What `rank` of `world` renders of an image of `spp` samples per pixel, as a dict of Renderer.params
    keywords (sample_begin, sample_end, tile_begin, tile_count), or None when the rank has no share.
    "weak": every rank adds `spp` new sample indices (the image then holds world * spp samples).
    "strong": the spp samples are dealt out by sample_range; when there are fewer samples than ranks the image is
    split by area instead and every rank renders all samples of its tiles."""
    if scaling == "weak":
        return dict(sample_begin=rank * spp, sample_end=(rank + 1) * spp, tile_begin=0, tile_count=0)
    if scaling != "strong":
        raise ValueError("scaling must be 'weak' or 'strong'")
    if spp >= world:
        b, e = sample_range(rank, world, 0, spp)
        return dict(sample_begin=b, sample_end=e, tile_begin=0, tile_count=0)
    tiles = tile_range(rank, world, width, height)
    if tiles is None:
        return None
    return dict(sample_begin=0, sample_end=spp, tile_begin=tiles[0], tile_count=tiles[1])


def reduce_radiance(accum, dst=0, group=None, single_rank_too=False):
    """Sum every rank's accumulation tensor into rank `dst` (in place there).  One collective of
    w*h*3 float64 per render.  With one rank there is nothing to add and no collective is issued, unless
    single_rank_too asks for it (a one-GPU rehearsal of the RCCL call itself)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank_too):
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # gloo reduces host tensors only: stage through the host (rehearsals on boxes without RCCL peers)
            host = accum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dist.get_rank(group) == dst:  # gloo leaves the other ranks' buffers unspecified: theirs stay as they were
                accum.copy_(host)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum
