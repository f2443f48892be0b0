"""The two shared libraries load without a GPU and export every symbol their headers declare."""
import ctypes
import os
import re

from conftest import ROOT


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpth?_[a-z_0-9]+)\s*\(", text)))


def test_device_library_exports_header():
    from cuda_path_tracer_amd import _capi
    names = _declared("cpt_amd.h")
    assert sorted(_capi.AMD_SYMBOLS) == names
    lib = ctypes.CDLL(os.path.join(ROOT, "cuda_path_tracer_amd", "libcpt_amd.so"))
    for n in names:
        assert hasattr(lib, n), n


def test_host_library_exports_header():
    from cuda_path_tracer_amd import _capi
    names = _declared("cpt_host.h")
    assert sorted(_capi.HOST_SYMBOLS) == names
    lib = ctypes.CDLL(os.path.join(ROOT, "cuda_path_tracer_amd", "libcpt_host.so"))
    for n in names:
        assert hasattr(lib, n), n


def test_record_sizes_match_header():
    from cuda_path_tracer_amd import _capi
    assert ctypes.sizeof(_capi.Node) == 128 and ctypes.sizeof(_capi.Tri) == 128
    assert ctypes.sizeof(_capi.TriNormals) == 72 and ctypes.sizeof(_capi.Material) == 136
    assert ctypes.sizeof(_capi.Camera) == 96


def test_product_does_not_reach_the_oracle():
    """Nothing that ships may import, link or execute the oracle."""
    pkg = os.path.join(ROOT, "cuda_path_tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in text and "liboracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_cli_refuses_cpu_rendering(tmp_path):
    """bin/ray without -g has no CPU renderer to fall back to."""
    import subprocess
    exe = os.path.join(ROOT, "cuda_path_tracer_amd", "bin", "ray")
    scene = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box2.ray")
    r = subprocess.run([exe, "-s", "1", "-w", "8", scene, str(tmp_path / "o.png")], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "no input" in r.stderr
