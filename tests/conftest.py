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
        try:
            subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "niagara_amd", "csrc")])
        except subprocess.CalledProcessError as e:
            raise pytest.UsageError("building niagara_amd/libniagara_vis.so failed (%s).  The inline-asm load rings of clustercull.hip are validated for ROCm "
                                    "7.2.0's hipcc only: with another compiler set NV_ALLOW_UNVALIDATED_HIPCC=1 to scan and build anyway (then run "
                                    "tests/test_plain_loads.py), or use libniagara_vis_plain.so, which builds with any hipcc" % e)
    if not os.path.exists(os.path.join(ROOT, "examples", "frame_driver")) or not os.path.exists(os.path.join(ROOT, "examples", "shard_driver")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


def _poison_torch_allocator():
    """GPU tests only: leave garbage in the blocks torch's caching allocator will hand out next.  A buffer from a fresh process is
    zero beyond the bytes the test wrote (new pages from the driver); in a long pytest process it is whatever an earlier test left there.
    A kernel that reads a little past a buffer, or a scratch word nobody initialised, therefore behaves differently from file to file
    — poisoning makes the 'long process' case the only case.  0xAB bytes: a huge index, a denormal-free negative float."""
    import torch
    if not torch.cuda.is_available() or os.environ.get("NV_TEST_POISON", "1") == "0":
        return
    big = [torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda") for n in (256 << 20, 64 << 20, 16 << 20, 4 << 20, 2 << 20)]
    small = [torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda") for n in (512, 2048, 8192, 65536, 262144) for _ in range(16)]
    torch.cuda.synchronize()
    del big, small  # back to the allocator's free lists, contents intact


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu:
        _poison_torch_allocator()
    yield
    if gpu:
        # the experiments build (NV_LIBRARY_PATH=niagara_amd/libniagara_vis_exp.so) keeps every library-owned block between canary
        # zones: none may have been written by the test's launches (context.hip nv_debug_check_scratch)
        from niagara_amd import _lib
        check = getattr(_lib.lib, "nv_debug_check_scratch", None)
        if check is not None:
            assert check() == 0, "a kernel wrote outside a library-owned block"


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
