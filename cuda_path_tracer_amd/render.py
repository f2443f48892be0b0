"""GPU renderer: the Python face of libcpt_amd.so.

PyTorch is used only as plumbing (device buffers, streams); every pixel is computed by the HIP
library.  Nothing here computes on the CPU and nothing here may import the oracle.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import MODE_MEGAKERNEL, MODE_WAVEFRONT  # noqa: F401


class CptError(RuntimeError):
    pass


NODE_DTYPE = np.dtype([("lbox", np.float64, 6), ("rbox", np.float64, 6), ("lref", np.int32), ("rref", np.int32),
                       ("lid", np.int32), ("rid", np.int32), ("self_id", np.int32), ("pad_", np.int32, 3)])
assert NODE_DTYPE.itemsize == C.sizeof(_capi.Node)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Renderer:
    """One GPU context.  Mirrors the GPU half of the reference's RayTracer
    (reference: RayTracer::traceImageGPU src/path_tracer.cu:507-551,
    RayTracer::betterPathTracing src/improved_path_tracer.cu:277-347)."""

    def __init__(self, device=0):
        self._L = _capi.amd()
        h = C.c_void_p()
        rc = self._L.cpt_create(int(device), C.byref(h))
        if rc != 0:
            raise CptError("cpt_create(%d): %s" % (device, self._L.cpt_last_error(None).decode()))
        self._h = h
        self.device = int(device)
        self._scene = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.cpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise CptError("%s failed (%d): %s" % (what, rc, self._L.cpt_last_error(self._h).decode()))

    # --- configuration ------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        """Launch on this hipStream_t (an integer handle, e.g. torch.cuda.Stream().cuda_stream; 0 is the
        legacy default stream).  None goes back to the context's own stream."""
        ptr = C.c_void_p(-1) if hip_stream_ptr is None else C.c_void_p(hip_stream_ptr)
        self._check(self._L.cpt_set_stream(self._h, ptr), "cpt_set_stream")

    def set_option(self, key, value):
        self._check(self._L.cpt_set_option(self._h, key.encode(), int(value)), "cpt_set_option")

    def set_instrumentation(self, time_kernels=False, count_work=False):
        self._check(self._L.cpt_set_instrumentation(self._h, int(time_kernels), int(count_work)),
                    "cpt_set_instrumentation")

    def upload(self, host_scene):
        self._check(self._L.cpt_upload_scene(self._h, C.byref(host_scene.desc)), "cpt_upload_scene")
        self._scene = host_scene  # keeps the host arrays alive, and n_tree_nodes for bitmaps

    def build(self, host_scene, keep_ranks=False):
        """Upload triangles, materials and camera and build the BVH on the GPU (the host scene's own tree is not used).
        keep_ranks: the triangle records' tie-break ranks (the reference tree's, as HostScene fills them in) are kept, so
        that ties between equal hits fall as on the reference's tree."""
        desc = host_scene.desc
        if keep_ranks:
            desc = _capi.SceneDesc()
            C.memmove(C.byref(desc), C.byref(host_scene.desc), C.sizeof(desc))
            desc.flags |= 1  # CPT_SCENE_KEEP_RANKS
        self._check(self._L.cpt_build_scene(self._h, C.byref(desc)), "cpt_build_scene")
        self._scene = host_scene

    def refit(self, host_scene):
        """Same triangles with new vertices: recompute the boxes of the tree on the GPU, keep its topology."""
        self._check(self._L.cpt_refit_scene(self._h, C.byref(host_scene.desc)), "cpt_refit_scene")
        self._scene = host_scene

    def download_tree(self):
        """(nodes, test_rank, height) of the scene on the GPU: node records as a structured array of _capi.Node."""
        n, h = C.c_int32(0), C.c_int32(0)
        self._check(self._L.cpt_download_tree(self._h, None, 0, None, 0, C.byref(n), C.byref(h)), "cpt_download_tree")
        nodes = np.zeros(max(n.value, 1), dtype=NODE_DTYPE)
        rank = np.zeros(self._scene.desc.n_tris, np.uint32)
        self._check(self._L.cpt_download_tree(self._h, _vp(nodes), nodes.shape[0], _vp(rank), rank.shape[0], C.byref(n), C.byref(h)),
                    "cpt_download_tree")
        return nodes[:n.value], rank, h.value

    @staticmethod
    def params(width, height, sample_begin, sample_end, depth, mode=MODE_WAVEFRONT, seed=1984, dof=False,
               focal_distance=8.5, aperture=0.6, russian_roulette=False, tile_begin=0, tile_count=0):
        return _capi.RenderParams(int(width), int(height), int(sample_begin), int(sample_end), int(depth), int(mode),
                                  int(seed), int(bool(dof)), int(bool(russian_roulette)), float(focal_distance),
                                  float(aperture), int(tile_begin), int(tile_count))

    # --- rendering ------------------------------------------------------------------------------
    def render_into(self, accum_ptr, params):
        """Add samples [begin, end) to a device accumulation buffer (w*h*3 float64) given by address."""
        self._check(self._L.cpt_render(self._h, C.byref(params), C.c_void_p(accum_ptr)), "cpt_render")

    def render_into_noalloc(self, accum_ptr, params):
        """cpt_render_noalloc: as render_into, with all working memory taken from the library's static device array."""
        self._check(self._L.cpt_render_noalloc(self._h, C.byref(params), C.c_void_p(accum_ptr)), "cpt_render_noalloc")

    def tonemap_into(self, accum_ptr, width, height, total_spp, rgb_ptr):
        self._check(self._L.cpt_tonemap(self._h, C.c_void_p(accum_ptr), width, height, total_spp, C.c_void_p(rgb_ptr)),
                    "cpt_tonemap")

    def render_image(self, params):
        """Whole-image render to a host (h, w, 3) uint8 array, row 0 = bottom (traceImageGPU)."""
        rgb = np.zeros((params.height, params.width, 3), np.uint8)
        self._check(self._L.cpt_render_image(self._h, C.byref(params), _vp(rgb)), "cpt_render_image")
        return rgb

    def synchronize(self):
        self._check(self._L.cpt_synchronize(self._h), "cpt_synchronize")

    def render_image_multi(self, others, params):
        """cpt_render_image_multi: this renderer and `others` (each with the same scene in place, normally one per GPU)
        share the samples of one render; the radiance sums are added on this renderer's GPU.  Returns (rgb, samples)."""
        rgb = np.zeros((params.height, params.width, 3), np.uint8)
        handles = (C.c_void_p * (1 + len(others)))(self._h, *[o._h for o in others])
        done = C.c_int32(0)
        self._check(self._L.cpt_render_image_multi(handles, len(handles), C.byref(params), _vp(rgb), C.byref(done)),
                    "cpt_render_image_multi")
        return rgb, done.value

    def render_progressive(self, params, chunk_spp, accum=None, accum_samples=0, on_progress=None):
        """cpt_render_progressive: returns (rgb (h, w, 3) uint8, samples this call added).  `accum`, a
        float64 array (h, w, 3), carries the radiance sums between calls (resume / checkpoint);
        on_progress(samples_in_image, rgb_view) may return True to stop."""
        rgb = np.zeros((params.height, params.width, 3), np.uint8)
        if accum is not None:
            assert accum.dtype == np.float64 and accum.flags.c_contiguous and accum.shape == rgb.shape

        def cb(_user, samples, _ptr):
            return 1 if (on_progress is not None and on_progress(samples, rgb)) else 0
        fn = _capi.PROGRESS_FN(cb)
        done = C.c_int32(0)
        self._check(self._L.cpt_render_progressive(self._h, C.byref(params), int(chunk_spp),
                                                   accum.ctypes.data if accum is not None else None, int(accum_samples),
                                                   rgb.ctypes.data, fn, None, C.byref(done)), "cpt_render_progressive")
        return rgb, done.value

    def stats(self):
        s = _capi.Stats()
        self._check(self._L.cpt_get_stats(self._h, C.byref(s)), "cpt_get_stats")
        return {k: getattr(s, k) for k, _ in _capi.Stats._fields_}

    def reset_stats(self):
        self._check(self._L.cpt_reset_stats(self._h), "cpt_reset_stats")

    # --- parity hooks ---------------------------------------------------------------------------
    def extend_rays(self, origins, dirs):
        """Closest hits of explicit rays through the pipeline's own extend kernel: (tri, t)."""
        o = np.ascontiguousarray(origins, np.float64)
        d = np.ascontiguousarray(dirs, np.float64)
        n = o.shape[0]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float64)
        self._check(self._L.cpt_extend_rays(self._h, n, o.ctypes.data, d.ctypes.data, tri.ctypes.data, t.ctypes.data),
                    "cpt_extend_rays")
        return tri, t

    def extend_rays_observed(self, origins, dirs, bitmap=False):
        """extend_rays through the kernel's counting instance: dict with prim, t, vhash (order-free hash of the
        canonical ids of the nodes whose boxes each ray's walk tested) and, on request, one visit bitmap row per ray."""
        o = np.ascontiguousarray(origins, np.float64)
        d = np.ascontiguousarray(dirs, np.float64)
        n = o.shape[0]
        tri = np.empty(n, np.int32)
        t = np.empty(n, np.float64)
        vh = np.zeros(n, np.uint64)
        stride = (self._scene.n_tree_nodes + 7) // 8 if self._scene is not None else 0
        bm = np.zeros((n, stride), np.uint8) if bitmap else None
        self._check(self._L.cpt_extend_rays_observed(self._h, n, _vp(o), _vp(d), _vp(tri), _vp(t), _vp(vh), _vp(bm), stride),
                    "cpt_extend_rays_observed")
        return dict(prim=tri, t=t, vhash=vh, bitmap=bm)

    def trace_rays(self, origins, dirs, bitmap=False):
        o = np.ascontiguousarray(origins, np.float64)
        d = np.ascontiguousarray(dirs, np.float64)
        n = o.shape[0]
        tri = np.zeros(n, np.int32)
        t = np.zeros(n, np.float64)
        nrm = np.zeros((n, 3), np.float64)
        vh = np.zeros(n, np.uint64)
        cnt = np.zeros((n, 4), np.int32)
        stride = (self._scene.n_tree_nodes + 7) // 8 if self._scene is not None else 0
        bm = np.zeros((n, stride), np.uint8) if bitmap else None
        self._check(self._L.cpt_trace_rays(self._h, n, _vp(o), _vp(d), _vp(tri), _vp(t), _vp(nrm), _vp(vh), _vp(cnt),
                                           _vp(bm), stride), "cpt_trace_rays")
        return dict(prim=tri, t=t, normal=nrm, vhash=vh, node_steps=cnt[:, 0], box_tests=cnt[:, 1],
                    tri_tests=cnt[:, 2], max_stack=cnt[:, 3], bitmap=bm)

    def camera_rays(self, width, height, jx=0.5, jy=0.5):
        o = np.zeros((width * height, 3), np.float64)
        d = np.zeros((width * height, 3), np.float64)
        self._check(self._L.cpt_camera_rays(self._h, width, height, float(jx), float(jy), _vp(o), _vp(d)),
                    "cpt_camera_rays")
        return o, d

    def path_radiance(self, params, ijs):
        ijs = np.ascontiguousarray(ijs, np.int32)
        rad = np.zeros((ijs.shape[0], 3), np.float64)
        self._check(self._L.cpt_path_radiance(self._h, C.byref(params), ijs.shape[0], _vp(ijs), _vp(rad)),
                    "cpt_path_radiance")
        return rad

    def probe_math(self, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        b = None if b is None else np.ascontiguousarray(b, np.float64)
        s = np.zeros_like(a)
        q = np.zeros_like(a)
        self._check(self._L.cpt_probe_math(self._h, a.size, _vp(a), _vp(b), _vp(s), _vp(q)), "cpt_probe_math")
        return s, q

    def probe_rng(self, ctr, key):
        c = np.asarray(ctr, np.uint32)
        k = np.asarray(key, np.uint32)
        out = np.zeros(4, np.uint32)
        u = np.zeros(2, np.float64)
        self._check(self._L.cpt_probe_rng(self._h, _vp(c), _vp(k), _vp(out), _vp(u)), "cpt_probe_rng")
        return out, u
