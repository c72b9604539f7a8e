"""clustercull.hip issues its streaming loads from inline asm and waits with hand-counted vmcnt
values — correct only as long as hipcc neither copies nor reuses a slot register between issue and wait.  Besides the
ISA scan that is part of the build (tools/check_asm_hazards.py), the library is also built with ordinary loads and the
compiler's own waits (-DNV_PLAIN_LOADS -> niagara_amd/libniagara_vis_plain.so); this test runs the same passes through
both builds, in separate processes, and requires identical visible lists and visibility words.  (It would have caught
the round-2 hazard: a division scheduled into the registers of a drained ring's last loads.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAIN = os.path.join(ROOT, "niagara_amd", "libniagara_vis_plain.so")
PRODUCT = os.path.join(ROOT, "niagara_amd", "libniagara_vis.so")


EXPERIMENTS = os.path.join(ROOT, "niagara_amd", "libniagara_vis_exp.so")


def _run(lib, **extra):
    env = dict(os.environ, NV_LIBRARY_PATH=lib, **extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "plain_runner.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l and l[0].isdigit()]


def test_plain_build_exists_and_exports_the_abi():
    import ctypes
    assert os.path.exists(PLAIN), "make -C niagara_amd/csrc builds it next to the product"
    lib = ctypes.CDLL(PLAIN)
    for name in ("nv_create", "nv_clustercull", "nv_drawcull", "nv_depthreduce"):
        assert hasattr(lib, name)


@pytest.mark.gpu
def test_asm_rings_equal_plain_loads():
    a, b = _run(PRODUCT), _run(PLAIN)
    assert len(a) == 24 and a == b
    assert any(int(l.split()[5]) > 1000 for l in a)


@pytest.mark.gpu
def test_occlusion_stage_scan_fallback_equals_the_list():
    """The late pass's occlusion stage takes its commands from the list the cull kernel left; when a sub-list runs out of
    room the stage scans all commands instead.  The product sizes the sub-lists so that this cannot happen, so the fallback
    is exercised through the experiments build (the only one that reads the environment) with room for 4 entries."""
    a, b = _run(PRODUCT), _run(EXPERIMENTS, NV_HIZ_LIST_STRIDE="4")
    assert len(a) == 24 and a == b
