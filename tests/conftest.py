import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # test infrastructure: the CPU oracle (and oracle/_ref where the reference tree exists)
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so) or os.path.isdir("/root/reference/src/shaders"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = os.path.join(ROOT, "niagara_amd", "libniagara_vis.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "niagara_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "examples", "frame_driver")) or not os.path.exists(os.path.join(ROOT, "examples", "shard_driver")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _drop_scene_registrations(request):
    """nv_upload_meshlets / nv_upload_meshes register buffers by device pointer.  A context that is shared between tests
    must not carry one test's registration into the next: torch's allocator hands the same addresses out again."""
    if "ctx" in request.fixturenames:
        c = request.getfixturevalue("ctx")
        c.upload_meshlets(None, 0)
        c.upload_meshes(None, 0)
        c.upload_draws(None, 0)
    yield
