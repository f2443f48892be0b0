"""The C ABI called from C++, with the GPU library's host code under ASan + UBSan (tests/sanitize/capi_driver,
built on the CPU box by tests/sanitize/build_capi_driver.sh or __graft_entry__.build()).  Device code cannot be
sanitized on this pool; this covers the host half: argument checking, buffer carving, uploads, copy-backs."""
import os
import subprocess

import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu


def test_capi_host_code_is_clean_under_sanitizers():
    exe = os.path.join(ROOT, "tests", "sanitize", "capi_driver")
    assert os.path.exists(exe), "tests/sanitize/capi_driver is not built: run tests/sanitize/build_capi_driver.sh"
    # the HIP runtime keeps allocations until process exit and maps memory where ASan expects a shadow gap
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, scene_path("cornell-box2-transmissive"), scene_path("teapot-trans"), scene_path("dragon-spec")],
                       capture_output=True, text=True, timeout=300, env=env)
    # a sanitizer report starts with its error class and the stacks of the access, the free and the allocation: show
    # the head of stderr (and keep all of it in a file when the run has somewhere to put it)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if r.stderr and os.path.isdir(out_dir) and os.access(out_dir, os.W_OK):
        with open(os.path.join(out_dir, "asan_capi_driver_stderr.txt"), "w") as f:
            f.write(r.stderr)
    head = "\n".join(r.stderr.splitlines()[:80])
    assert r.returncode == 0, r.stdout[-500:] + "\n--- stderr, first 80 lines ---\n" + head
    assert "capi driver ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, head
