"""ctypes declarations for libcpt_amd.so (include/cpt_amd.h) and libcpt_host.so (include/cpt_host.h).

There is no fallback: if a library is missing, loading raises with the command that builds it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class BuildMissing(RuntimeError):
    pass


def _load(name):
    path = os.path.join(_HERE, name)
    if name == "libcpt_amd.so" and os.environ.get("CPT_AMD_LIBRARY"):  # A/B builds of the GPU library (tuning experiments)
        path = os.path.abspath(os.environ["CPT_AMD_LIBRARY"])
    if not os.path.exists(path):
        raise BuildMissing(
            "%s is not built. Run `make -C %s` (or `python -c 'import __graft_entry__ as g; g.build()'` "
            "from the repository root). There is no CPU fallback for the GPU path." % (path, _HERE))
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


class Node(C.Structure):
    _fields_ = [("lbox", C.c_double * 6), ("rbox", C.c_double * 6), ("lref", C.c_int32), ("rref", C.c_int32),
                ("lid", C.c_int32), ("rid", C.c_int32), ("self_id", C.c_int32), ("pad_", C.c_int32 * 3)]


class Tri(C.Structure):
    _fields_ = [("a", C.c_double * 3), ("b", C.c_double * 3), ("c", C.c_double * 3), ("n", C.c_double * 3),
                ("material", C.c_int32), ("flags", C.c_uint32), ("test_rank", C.c_uint32), ("pad_", C.c_int32 * 5)]


class TriNormals(C.Structure):
    _fields_ = [("na", C.c_double * 3), ("nb", C.c_double * 3), ("nc", C.c_double * 3)]


class Material(C.Structure):
    _fields_ = [("ke", C.c_double * 3), ("kd", C.c_double * 3), ("ks", C.c_double * 3), ("kr", C.c_double * 3),
                ("kt", C.c_double * 3), ("index", C.c_double), ("flags", C.c_uint32), ("pad_", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("look", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_tris", C.c_int32), ("n_materials", C.c_int32),
                ("n_tree_nodes", C.c_int32), ("tree_height", C.c_int32), ("flags", C.c_uint32),
                ("nodes", C.POINTER(Node)), ("tris", C.POINTER(Tri)), ("tri_normals", C.POINTER(TriNormals)),
                ("materials", C.POINTER(Material)), ("camera", Camera)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
                ("depth", C.c_int32), ("mode", C.c_int32), ("seed", C.c_uint64), ("dof", C.c_int32),
                ("russian_roulette", C.c_int32), ("focal_distance", C.c_double), ("aperture", C.c_double),
                ("tile_begin", C.c_int32), ("tile_count", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("hits", C.c_uint64), ("node_steps", C.c_uint64),
                ("tri_tests", C.c_uint64), ("launches_generate", C.c_uint64), ("launches_extend", C.c_uint64),
                ("launches_shade", C.c_uint64), ("launches_mega", C.c_uint64), ("ms_generate", C.c_double),
                ("ms_extend", C.c_double), ("ms_shade", C.c_double), ("ms_mega", C.c_double), ("ms_total", C.c_double),
                ("iterations", C.c_uint64), ("exact_node_steps", C.c_uint64), ("box_passes", C.c_uint64),
                ("leaf_passes", C.c_uint64)]


assert C.sizeof(Node) == 128 and C.sizeof(Tri) == 128 and C.sizeof(TriNormals) == 72 and C.sizeof(Material) == 136

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p)

NOALLOC_SCRATCH_BYTES = 128 << 20  # CPT_NOALLOC_SCRATCH_BYTES
MODE_WAVEFRONT = 0
MODE_MEGAKERNEL = 1

# every symbol include/cpt_amd.h declares
AMD_SYMBOLS = [
    "cpt_create", "cpt_destroy", "cpt_last_error", "cpt_set_stream", "cpt_set_option", "cpt_set_instrumentation",
    "cpt_upload_scene", "cpt_build_scene", "cpt_refit_scene", "cpt_download_tree", "cpt_render", "cpt_render_noalloc", "cpt_tonemap", "cpt_render_image", "cpt_render_image_multi", "cpt_render_progressive", "cpt_synchronize", "cpt_get_stats",
    "cpt_reset_stats", "cpt_extend_rays", "cpt_extend_rays_observed", "cpt_trace_rays", "cpt_camera_rays", "cpt_path_radiance", "cpt_probe_math", "cpt_probe_rng",
]
# every symbol include/cpt_host.h declares
HOST_SYMBOLS = [
    "cpth_load", "cpth_load_text", "cpth_free", "cpth_describe", "cpth_aspect", "cpth_warning_count", "cpth_warning",
    "cpth_tree_nodes", "cpth_write_image", "cpth_describe_fast", "cpth_fast_tree_nodes",
]

_amd = None
_host = None


def amd():
    """libcpt_amd.so with argument types set."""
    global _amd
    if _amd is None:
        L = _load("libcpt_amd.so")
        vp, i32, dbl = C.c_void_p, C.c_int32, C.c_double
        L.cpt_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.cpt_destroy.argtypes = [vp]
        L.cpt_destroy.restype = None
        L.cpt_last_error.argtypes = [vp]
        L.cpt_last_error.restype = C.c_char_p
        L.cpt_set_stream.argtypes = [vp, vp]
        L.cpt_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
        L.cpt_set_instrumentation.argtypes = [vp, C.c_int, C.c_int]
        L.cpt_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
        L.cpt_build_scene.argtypes = [vp, C.POINTER(SceneDesc)]
        L.cpt_refit_scene.argtypes = [vp, C.POINTER(SceneDesc)]
        L.cpt_download_tree.argtypes = [vp, vp, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
        L.cpt_render.argtypes = [vp, C.POINTER(RenderParams), vp]
        L.cpt_render_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp]
        L.cpt_tonemap.argtypes = [vp, vp, i32, i32, i32, vp]
        L.cpt_render_image.argtypes = [vp, C.POINTER(RenderParams), vp]
        try:
            L.cpt_render_image_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(RenderParams), vp, C.POINTER(i32)]
        except AttributeError:  # only an older A/B build named by CPT_AMD_LIBRARY may lack an entry point
            if not os.environ.get("CPT_AMD_LIBRARY"):
                raise
        L.cpt_render_progressive.argtypes = [vp, C.POINTER(RenderParams), i32, vp, i32, vp, PROGRESS_FN, vp, C.POINTER(i32)]
        L.cpt_synchronize.argtypes = [vp]
        L.cpt_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.cpt_reset_stats.argtypes = [vp]
        L.cpt_extend_rays.argtypes = [vp, i32, vp, vp, vp, vp]
        try:
            L.cpt_extend_rays_observed.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32]
        except AttributeError:  # only an older A/B build named by CPT_AMD_LIBRARY may lack an entry point
            if not os.environ.get("CPT_AMD_LIBRARY"):
                raise
        L.cpt_trace_rays.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32]
        L.cpt_camera_rays.argtypes = [vp, i32, i32, dbl, dbl, vp, vp]
        L.cpt_path_radiance.argtypes = [vp, C.POINTER(RenderParams), i32, vp, vp]
        L.cpt_probe_math.argtypes = [vp, i32, vp, vp, vp, vp]
        L.cpt_probe_rng.argtypes = [vp, vp, vp, vp, vp]
        _amd = L
    return _amd


def host():
    """libcpt_host.so with argument types set."""
    global _host
    if _host is None:
        L = _load("libcpt_host.so")
        vp = C.c_void_p
        L.cpth_load.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
        L.cpth_load_text.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int]
        L.cpth_free.argtypes = [vp]
        L.cpth_free.restype = None
        L.cpth_describe.argtypes = [vp, C.POINTER(SceneDesc)]
        L.cpth_describe_fast.argtypes = [vp, C.POINTER(SceneDesc)]
        L.cpth_fast_tree_nodes.argtypes = [vp, vp, vp]
        L.cpth_aspect.argtypes = [vp]
        L.cpth_aspect.restype = C.c_double
        L.cpth_warning_count.argtypes = [vp]
        L.cpth_warning.argtypes = [vp, C.c_int]
        L.cpth_warning.restype = C.c_char_p
        L.cpth_tree_nodes.argtypes = [vp, vp, vp]
        L.cpth_write_image.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp, C.c_char_p, C.c_int]
        _host = L
    return _host
