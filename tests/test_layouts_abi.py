"""Layouts and ABI surface: product dtypes == oracle dtypes == the reference's own structs (via oracle/_ref);
the shared library loads without a GPU and exports every symbol include/niagara_vis.h declares."""
import os
import re

import numpy as np
import pytest

import niagara_amd as N
import oracle
import oracle.ref as R
from niagara_amd import layouts as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_and_oracle_dtypes_agree():
    for name in ("MESHLET", "MESHDRAW", "MESHLOD", "MESH", "DRAWCMD", "TASKCMD", "CULLDATA"):
        a, b = getattr(L, name), getattr(oracle, name)
        assert a.itemsize == b.itemsize, name
        assert [(n, a.fields[n][1]) for n in a.names] == [(n, b.fields[n][1]) for n in b.names], name


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no reference tree and no prebuilt .so)")
def test_layouts_match_reference_structs():
    s = lambda i: int(R.lib().ref_sizeof(i))
    assert s(0) == L.MESHLET.itemsize == 24
    assert s(1) == L.MESHDRAW.itemsize == 48
    assert s(2) == L.MESHLOD.itemsize == 20
    assert s(3) == L.MESH.itemsize == 208
    assert s(4) == L.DRAWCMD.itemsize == 24
    assert s(5) == L.TASKCMD.itemsize == 20
    assert s(6) == 136 and L.CULLDATA.itemsize == 144  # alignas(16) on the host struct pads 136 -> 144
    assert s(7) == L.MESH.fields["lods"][1] == 48
    assert s(8) == L.CULLDATA.fields["P00"][1] == 64
    assert s(9) == L.CULLDATA.fields["frustum"][1] == 80
    assert s(10) == L.CULLDATA.fields["lodTarget"][1] == 96
    assert s(11) == L.CULLDATA.fields["drawCount"][1] == 108
    assert s(12) == L.CULLDATA.fields["cullingEnabled"][1] == 112
    assert s(13) == L.CULLDATA.fields["postPass"][1] == 132
    assert s(14) == L.MESHDRAW.fields["orientation"][1] == 16
    assert s(15) == L.MESHDRAW.fields["meshIndex"][1] == 32
    assert s(16) == L.MESHLET.fields["cone_axis"][1] == 8
    assert s(17) == L.MESHLET.fields["dataOffset"][1] == 12
    assert (s(18), s(19), s(20), s(21), s(22)) == (L.TASK_WGSIZE, L.TASK_WGLIMIT, L.CLUSTER_LIMIT, L.CLUSTER_TILE, 1)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "niagara_vis.h")).read()
    declared = set(re.findall(r"^(?:int|void|uint32_t|const char\*)\s+(nv_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(N.EXPORTS)
    for name in declared:
        assert hasattr(N.lib, name), name
    assert N.lib.nv_version().startswith(b"niagara_vis")


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "niagara_vis.h"\nint main(void){return sizeof(NvCullData)==144?0:1;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(tmp_path / "c")])
    subprocess.check_call([str(tmp_path / "c")])
    subprocess.check_call(["g++", "-std=c++11", "-x", "c++", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(tmp_path / "cc")])


def test_no_gpu_means_loud_failure(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    from niagara_amd import pipeline
    with pytest.raises(N.NvError):
        pipeline.Context()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "niagara_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text, f
