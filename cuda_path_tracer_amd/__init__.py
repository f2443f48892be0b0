"""MI355X-native path tracer behind the reference's GPU render path.

    from cuda_path_tracer_amd import HostScene, Renderer
    scene = HostScene("scene.ray")            # parse + reference-identical BVH + flat records (host, C++)
    r = Renderer(device=0); r.upload(scene)   # HIP library, gfx950
    rgb = r.render_image(r.params(512, 512, 0, 256, depth=5))

The compute path is libcpt_amd.so (hand-written HIP); importing this package without the built
libraries works, using it raises.
"""
from ._capi import MODE_MEGAKERNEL, MODE_WAVEFRONT, BuildMissing  # noqa: F401
from .render import CptError, Renderer  # noqa: F401
from .scene import ArrayScene, HostScene, write_image  # noqa: F401
