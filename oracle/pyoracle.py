"""ctypes front end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under cuda_path_tracer_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so with the committed Makefile (g++, strict IEEE fp64)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_render_cpu_reference.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, path):
        L = lib()
        err = C.create_string_buffer(512)
        self._h = L.orc_load(os.fsencode(path), err, 512)
        if not self._h:
            raise RuntimeError("oracle: %s: %s" % (path, err.value.decode()))
        self._refresh_counts()

    def _refresh_counts(self):
        c = np.zeros(4, np.int32)
        lib().orc_counts(C.c_void_p(self._h), _p(c))
        self.n_prims, self.n_nodes, self.n_meshes, self.height = (int(x) for x in c)

    def close(self):
        if self._h:
            lib().orc_free(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def h(self):
        return C.c_void_p(self._h)

    def prims(self):
        geom = np.zeros((self.n_prims, 12), np.float64)
        vn = np.zeros((self.n_prims, 9), np.float64)
        ids = np.zeros((self.n_prims, 3), np.int32)
        lib().orc_get_prims(self.h, _p(geom), _p(vn), _p(ids))
        return geom, vn, ids

    def materials(self):
        vals = np.zeros((self.n_meshes, 16), np.float64)
        flags = np.zeros(self.n_meshes, np.int32)
        lib().orc_get_materials(self.h, _p(vals), _p(flags))
        return vals, flags

    def nodes(self):
        boxes = np.zeros((self.n_nodes, 6), np.float64)
        topo = np.zeros((self.n_nodes, 3), np.int32)
        lib().orc_get_nodes(self.h, _p(boxes), _p(topo))
        return boxes, topo

    def set_tree(self, boxes, topo):
        """Replace the tree by a table in the format of nodes() (row 0 = root); node ids become its preorder."""
        boxes = np.ascontiguousarray(boxes, np.float64)
        topo = np.ascontiguousarray(topo, np.int32)
        if not lib().orc_set_tree(self.h, C.c_int(boxes.shape[0]), _p(boxes), _p(topo)):
            raise ValueError("not a binary tree over this scene's triangles")
        self._refresh_counts()

    def camera(self):
        c = np.zeros(12, np.float64)
        lib().orc_get_camera(self.h, _p(c))
        return c

    def camera_rays(self, w, h, jx=0.5, jy=0.5):
        o = np.zeros((w * h, 3), np.float64)
        d = np.zeros((w * h, 3), np.float64)
        lib().orc_camera_rays(self.h, C.c_int(w), C.c_int(h), C.c_double(jx), C.c_double(jy), _p(o), _p(d))
        return o, d

    def trace(self, o, d, bitmap=False):
        o = np.ascontiguousarray(o, np.float64)
        d = np.ascontiguousarray(d, np.float64)
        n = o.shape[0]
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float64)
        nrm = np.zeros((n, 3), np.float64)
        vh = np.zeros(n, np.uint64)
        cnt = np.zeros((n, 4), np.int32)
        stride = (self.n_nodes + 7) // 8
        bm = np.zeros((n, stride), np.uint8) if bitmap else None
        lib().orc_trace(self.h, C.c_int(n), _p(o), _p(d), _p(prim), _p(t), _p(nrm), _p(vh), _p(cnt),
                        _p(bm) if bitmap else None, C.c_int(stride))
        return dict(prim=prim, t=t, normal=nrm, vhash=vh, node_steps=cnt[:, 0], box_tests=cnt[:, 1],
                    tri_tests=cnt[:, 2], max_stack=cnt[:, 3], bitmap=bm)

    def render(self, w, h, s0, s1, depth, seed=1984, dof=0, focal=8.5, aperture=0.6, rr=0, accum=None):
        if accum is None:
            accum = np.zeros((h, w, 3), np.float64)
        st = np.zeros(5, np.uint64)
        lib().orc_render(self.h, C.c_int(w), C.c_int(h), C.c_int(s0), C.c_int(s1), C.c_int(depth),
                         C.c_uint64(seed), C.c_int(dof), C.c_double(focal), C.c_double(aperture), C.c_int(rr),
                         _p(accum), _p(st))
        stats = dict(paths=int(st[0]), segments=int(st[1]), hits=int(st[2]), node_steps=int(st[3]),
                     tri_tests=int(st[4]))
        return accum, stats

    def paths(self, w, h, depth, ijs, seed=1984, dof=0, focal=8.5, aperture=0.6, rr=0):
        ijs = np.ascontiguousarray(ijs, np.int32)
        rad = np.zeros((ijs.shape[0], 3), np.float64)
        lib().orc_paths(self.h, C.c_int(w), C.c_int(h), C.c_int(depth), C.c_uint64(seed), C.c_int(dof),
                        C.c_double(focal), C.c_double(aperture), C.c_int(rr), C.c_int(ijs.shape[0]), _p(ijs), _p(rad))
        return rad

    def render_cpu_reference(self, w, h, spp, depth, threads, seed=1):
        rgb = np.zeros((h, w, 3), np.uint8)
        mean = np.zeros((h, w, 3), np.float64)
        secs = lib().orc_render_cpu_reference(self.h, C.c_int(w), C.c_int(h), C.c_int(spp), C.c_int(depth),
                                              C.c_int(threads), C.c_uint64(seed), _p(rgb), _p(mean))
        return rgb, mean, float(secs)


def tonemap(accum, spp):
    accum = np.ascontiguousarray(accum, np.float64)
    rgb = np.zeros(accum.shape, np.uint8)
    lib().orc_tonemap(_p(accum), C.c_int(accum.size // 3), C.c_int(spp), _p(rgb))
    return rgb


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().orc_philox(_p(c), _p(k), _p(out))
    return out


def u01(lo, hi):
    return float(lib().orc_u01(C.c_uint32(lo), C.c_uint32(hi)))
