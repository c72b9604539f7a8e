"""The hazard recognizer of hipcc does not look inside inline asm (clustercull.hip, rules above SlotA): every asm statement
that reads an SGPR a VALU instruction produced just before it must bring its own s_nop.  tools/check_asm_hazards.py
compiles the file to gfx950 ISA and checks that; this test runs it and also feeds the scanner a violation."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_scanner_flags_an_unguarded_block():
    bad = "\tv_readlane_b32 s6, v80, 21\n\tv_readlane_b32 s7, v80, 22\n\t;;#ASMSTART\n\tglobal_load_dword v22, v20, s[6:7]\n\t;;#ASMEND\n"
    good = bad.replace("\tglobal_load_dword", "\ts_nop 4\n\tglobal_load_dword")
    weak = bad.replace("\tglobal_load_dword", "\ts_nop 1\n\tglobal_load_dword")  # enough for a VALU consumer, not for VMEM
    assert len(chk.scan(bad)[2]) == 1
    assert chk.scan(good)[1:] == (1, [])
    assert len(chk.scan(weak)[2]) == 1
    valu = "\tv_readlane_b32 s10, v39, s14\n\t;;#ASMSTART\n\ts_nop 1\n\tv_mov_b32 v65, s10\n\t;;#ASMEND\n"
    assert chk.scan(valu)[1:] == (1, [])
    assert len(chk.scan(valu.replace("\ts_nop 1\n", ""))[2]) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_clustercull_asm_blocks_are_guarded():
    assert chk.main() == 0
