"""Host-side scene: .ray file -> reference-identical BVH -> flat records (through libcpt_host.so)."""
import ctypes as C
import os

import numpy as np

from . import _capi


class HostScene:
    """A loaded, flattened scene.  Mirrors what RayTracer::loadScene + traceSetup leave behind in the
    reference (reference: src/RayTracer.cpp:554-645): a parsed scene with its BVH built."""

    def __init__(self, path=None, text=None):
        L = _capi.host()
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        if (path is None) == (text is None):
            raise ValueError("give exactly one of path / text")
        if path is not None:
            rc = L.cpth_load(os.fsencode(path), C.byref(h), err, 1024)
        else:
            rc = L.cpth_load_text(text.encode(), C.byref(h), err, 1024)
        if rc != 0:
            raise RuntimeError("scene load failed: %s" % err.value.decode())
        self._h = h
        self.desc = _capi.SceneDesc()
        L.cpth_describe(self._h, C.byref(self.desc))
        self.aspect = float(L.cpth_aspect(self._h))
        self.warnings = [L.cpth_warning(self._h, k).decode() for k in range(L.cpth_warning_count(self._h))]

    n_tris = property(lambda s: s.desc.n_tris)
    n_nodes = property(lambda s: s.desc.n_nodes)
    n_tree_nodes = property(lambda s: s.desc.n_tree_nodes)
    n_materials = property(lambda s: s.desc.n_materials)
    tree_height = property(lambda s: s.desc.tree_height)

    def close(self):
        if getattr(self, "_h", None):
            _capi.host().cpth_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fast(self):
        """The same scene over the fast mode's tree (full-sweep surface-area build; the triangle records keep the
        reference tree's tie-break ranks, so every ray's closest hit is unchanged): an object Renderer.upload takes."""
        if getattr(self, "_fast", None) is None:
            f = FastTreeScene()
            f.desc = _capi.SceneDesc()
            if _capi.host().cpth_describe_fast(self._h, C.byref(f.desc)) != 0:
                raise RuntimeError("the fast tree could not be built")
            f.parent = self  # keeps the arrays alive
            f.aspect = self.aspect
            self._fast = f
        return self._fast

    def fast_tree(self):
        """tree() of the fast mode's tree."""
        f = self.fast()
        n = f.n_tree_nodes
        boxes = np.zeros((n, 6), np.float64)
        topo = np.zeros((n, 3), np.int32)
        _capi.host().cpth_fast_tree_nodes(self._h, boxes.ctypes.data_as(C.c_void_p), topo.ctypes.data_as(C.c_void_p))
        return boxes, topo

    def height_for_width(self, width):
        """reference: src/ui/CommandLineUI.cpp:86"""
        return int(width / self.aspect + 0.5)

    # --- views for tests and inspection ---------------------------------------------------------
    def tree(self):
        n = self.n_tree_nodes
        boxes = np.zeros((n, 6), np.float64)
        topo = np.zeros((n, 3), np.int32)
        _capi.host().cpth_tree_nodes(self._h, boxes.ctypes.data_as(C.c_void_p), topo.ctypes.data_as(C.c_void_p))
        return boxes, topo

    def _records(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype)
        buf = C.cast(ptr, C.POINTER(C.c_uint8 * (n * dtype.itemsize))).contents
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    def tris(self):
        dt = np.dtype([("a", "f8", 3), ("b", "f8", 3), ("c", "f8", 3), ("n", "f8", 3), ("material", "i4"),
                       ("flags", "u4"), ("test_rank", "u4"), ("pad", "i4", 5)])
        return self._records(self.desc.tris, self.n_tris, dt)

    def tri_normals(self):
        dt = np.dtype([("na", "f8", 3), ("nb", "f8", 3), ("nc", "f8", 3)])
        return self._records(self.desc.tri_normals, self.n_tris, dt)

    def materials(self):
        dt = np.dtype([("ke", "f8", 3), ("kd", "f8", 3), ("ks", "f8", 3), ("kr", "f8", 3), ("kt", "f8", 3),
                       ("index", "f8"), ("flags", "u4"), ("pad", "u4")])
        return self._records(self.desc.materials, self.n_materials, dt)

    def nodes(self):
        dt = np.dtype([("lbox", "f8", 6), ("rbox", "f8", 6), ("lref", "i4"), ("rref", "i4"), ("lid", "i4"),
                       ("rid", "i4"), ("self_id", "i4"), ("pad", "i4", 3)])
        return self._records(self.desc.nodes, self.n_nodes, dt)

    def camera(self):
        c = self.desc.camera
        return np.array(list(c.eye) + list(c.look) + list(c.u) + list(c.v), np.float64)


class FastTreeScene:
    """HostScene.fast(): the description of the same scene over the fast mode's tree."""
    n_tris = property(lambda s: s.desc.n_tris)
    n_nodes = property(lambda s: s.desc.n_nodes)
    n_tree_nodes = property(lambda s: s.desc.n_tree_nodes)
    tree_height = property(lambda s: s.desc.tree_height)

    def height_for_width(self, width):
        return self.parent.height_for_width(width)


def write_image(path, rgb_bottom_up):
    """Write an (h, w, 3) uint8 array whose row 0 is the bottom of the image (.png or .bmp)."""
    a = np.ascontiguousarray(rgb_bottom_up, np.uint8)
    h, w, _ = a.shape
    err = C.create_string_buffer(512)
    if _capi.host().cpth_write_image(os.fsencode(path), w, h, a.ctypes.data_as(C.c_void_p), err, 512) != 0:
        raise RuntimeError(err.value.decode())


TRI_DTYPE = np.dtype([("a", "f8", 3), ("b", "f8", 3), ("c", "f8", 3), ("n", "f8", 3), ("material", "i4"),
                      ("flags", "u4"), ("test_rank", "u4"), ("pad", "i4", 5)])


class ArrayScene:
    """Triangles handed over as arrays, without a tree: for Renderer.build, which has the GPU build the BVH.

    vertices: (n, 3, 3) float64.  One diffuse grey material unless `material` (a _capi.Material) is given;
    triangles are two-sided when `two_sided`.  The camera looks down -z from `eye`."""

    def __init__(self, vertices, material=None, two_sided=True, eye=(0.0, 0.0, 5.0)):
        v = np.ascontiguousarray(vertices, np.float64)
        n = v.shape[0]
        self._tris = np.zeros(n, TRI_DTYPE)
        self._tris["a"], self._tris["b"], self._tris["c"] = v[:, 0], v[:, 1], v[:, 2]
        nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        self._tris["n"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
        self._tris["flags"] = 1 if two_sided else 0
        self._normals = np.zeros((n, 9), np.float64)
        if material is None:
            material = _capi.Material()
            for k in range(3):
                material.kd[k] = 0.5
            material.flags = 1  # diffuse
        self._material = material
        self.desc = _capi.SceneDesc()
        self.desc.n_tris, self.desc.n_materials = n, 1
        self.desc.n_tree_nodes = 2 * n - 1
        self.desc.tris = C.cast(self._tris.ctypes.data, C.POINTER(_capi.Tri))
        self.desc.tri_normals = C.cast(self._normals.ctypes.data, C.POINTER(_capi.TriNormals))
        self.desc.materials = C.pointer(self._material)
        cam = self.desc.camera
        for k in range(3):
            cam.eye[k] = eye[k]
            cam.look[k] = (0.0, 0.0, -1.0)[k]
            cam.u[k] = (1.0, 0.0, 0.0)[k]
            cam.v[k] = (0.0, 1.0, 0.0)[k]
        self.aspect = 1.0

    n_tris = property(lambda s: s.desc.n_tris)
    n_tree_nodes = property(lambda s: s.desc.n_tree_nodes)

    def tris(self):
        return self._tris
