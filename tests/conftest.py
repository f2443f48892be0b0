import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(SCENES, name + ".ray")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with its committed Makefile."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def oracle_scenes(oracle):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = oracle.OracleScene(scene_path(name))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def host_scenes():
    from cuda_path_tracer_amd import HostScene
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = HostScene(scene_path(name))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def gpu():
    """One GPU context for the whole session; fails loudly when the HIP library is not built."""
    from cuda_path_tracer_amd import Renderer
    r = Renderer(0)
    yield r
    r.close()


@pytest.fixture(scope="session")
def gpu_scene(gpu, host_scenes):
    def use(name):
        hs = host_scenes(name)
        if getattr(gpu, "_scene", None) is not hs:  # also true after a test uploaded a scene of its own
            gpu.upload(hs)
        return gpu
    return use
