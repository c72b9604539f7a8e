"""The hazard recognizer of hipcc does not look inside inline asm (clustercull.hip, rules above SlotA): every asm statement
that reads an SGPR a VALU instruction produced just before it must bring its own s_nop.  tools/check_asm_hazards.py
compiles the file to gfx950 ISA and checks that, (check 2) that no instruction touches a ring register between the
asm statement that issues its load and the asm statement that waits for it, and (check 3) that every hand-counted
vmcnt(N) has at least N younger vector-memory operations behind the load it names on every path; this test runs them on the
real file and feeds each scanner a violation."""
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
    assert len(chk.scan_nops(bad)[2]) == 1
    assert chk.scan_nops(good)[1:] == (1, [])
    assert len(chk.scan_nops(weak)[2]) == 1
    valu = "\tv_readlane_b32 s10, v39, s14\n\t;;#ASMSTART\n\ts_nop 1\n\tv_mov_b32 v65, s10\n\t;;#ASMEND\n"
    assert chk.scan_nops(valu)[1:] == (1, [])
    assert len(chk.scan_nops(valu.replace("\ts_nop 1\n", ""))[2]) == 1


def test_scanner_flags_a_touched_in_flight_register():
    """the hazard found in round 2: a computation scheduled between a ring's last (unused) loads and the drain, in the
    registers of those loads"""
    kernel = ["\t;;#ASMSTART", "\ts_nop 4", "\tglobal_load_dwordx2 v[0:1], v20, s[28:29]", "\t;;#ASMEND",
              "\tv_add_f32_e32 v5, v6, v7",
              "\tv_rcp_f32_e32 v1, v9",                       # writes v1 while its load is in flight
              "\t;;#ASMSTART", "\ts_waitcnt vmcnt(0) ; nv_ready all", "\t;;#ASMEND",
              "\tv_mul_f32_e32 v0, v0, v1", "\ts_endpgm"]
    found = chk.scan_inflight("k", kernel)
    assert len(found) == 1 and found[0][2] == [1]
    ok = [l for l in kernel if "v_rcp" not in l]
    assert chk.scan_inflight("k", ok) == []
    # across a branch: the register is still in flight on the taken path
    branchy = kernel[:4] + ["\ts_cbranch_scc1 .LBB0_2", "\t;;#ASMSTART", "\ts_waitcnt vmcnt(0) ; nv_ready v[0:1]", "\t;;#ASMEND",
                            ".LBB0_2:", "\tv_mov_b32_e32 v3, v0", "\ts_endpgm"]
    assert len(chk.scan_inflight("k", branchy)) >= 1


def test_scanner_flags_a_counted_wait_that_is_too_weak():
    """check 3: vmcnt(N) that names a slot needs N younger vector-memory operations behind the slot's load on EVERY path"""
    def ring(n_younger, count, extra=()):
        k = ["\t;;#ASMSTART", "\ts_nop 4", "\tglobal_load_dwordx2 v[0:1], v20, s[28:29]", "\t;;#ASMEND"]
        for i in range(n_younger):
            k += ["\t;;#ASMSTART", "\ts_nop 4", "\tglobal_load_dwordx2 v[%d:%d], v20, s[28:29]" % (2 + 2 * i, 3 + 2 * i), "\t;;#ASMEND"]
        k += list(extra)
        k += ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(%d) ; nv_ready v[0:1]" % count, "\t;;#ASMEND", "\tv_mul_f32_e32 v40, v0, v1",
              "\t;;#ASMSTART", "\ts_waitcnt vmcnt(0) ; nv_ready all", "\t;;#ASMEND", "\ts_endpgm"]
        return k

    assert chk.scan_counts("k", ring(3, 3)) == []
    assert chk.scan_counts("k", ring(3, 2)) == []                       # stricter than needed is fine
    weak = chk.scan_counts("k", ring(2, 3))                             # only two younger loads: the slot may be one of the three outstanding
    assert [(r, k, n) for _, _, r, k, n in weak] == [(0, 2, 3), (1, 2, 3)]
    # a compiler-issued store between issue and wait counts (vmcnt counts stores on gfx9) ...
    assert chk.scan_counts("k", ring(2, 3, extra=["\tglobal_store_dword v30, v31, s[4:5]"])) == []
    # ... but not when it sits on one side of a branch only: the weakest path decides
    branchy = ring(2, 3, extra=["\ts_cbranch_scc1 .LBB0_2", "\tglobal_store_dword v30, v31, s[4:5]", ".LBB0_2:"])
    assert len(chk.scan_counts("k", branchy)) == 2
    # a loop that reissues the slot: the count restarts at the reissue
    loop = [".LBB0_1:"] + ring(3, 3)[:-4] + ["\ts_cbranch_scc1 .LBB0_1"] + ring(3, 3)[-4:]
    assert chk.scan_counts("k", loop) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_clustercull_asm_blocks_are_guarded():
    assert chk.main() == 0


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_dpp_scans_refuse_a_non_gfx9_target(tmp_path):
    """VERDICT r5 item 8 / ADVICE r4: args.h's DPP scans (row_bcast:15 / 31, wave_shr:1) exist on GFX9-family wave64 parts only.  Compiled for
    another target (gfx1100) the header must stop the build with its own #error text — not compile silently into something else; for gfx950 the same
    translation unit compiles."""
    import subprocess
    src = tmp_path / "tu.hip"
    src.write_text('#include "%s"\n#include "%s"\n' % (os.path.join(ROOT, "niagara_amd", "csrc", "cullmath.h"), os.path.join(ROOT, "niagara_amd", "csrc", "args.h")))
    bad = subprocess.run(["hipcc", "--offload-arch=gfx1100", "-std=c++17", "-c", str(src), "-o", str(tmp_path / "bad.o")], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
    assert "args.h: the DPP scans are written for GFX9-family wave64 targets (gfx950)" in bad.stderr
    good = subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", "-c", str(src), "-o", str(tmp_path / "good.o")], capture_output=True, text=True, timeout=300)
    assert good.returncode == 0, good.stderr[-2000:]


def test_decide_ring_stays_in_flight():
    """drawcull.hip's decide kernel keeps four units of requests in flight behind waits the COMPILER counts; twice in round 6 hipcc placed them so that
    the ring drained (results right, kernel 1 us slower): tools/check_decide_ring.py finds a wait with a small count inside the walk outside the
    queue's drains.  On the real file with the validated compiler, and on the two regressions it was written after."""
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    tool = os.path.join(ROOT, "tools", "check_decide_ring.py")
    src = os.path.join(ROOT, "niagara_amd", "csrc", "drawcull.hip")
    r = subprocess.run([sys.executable, tool, "--hipcc", hipcc], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    text = open(src).read()
    gather = "\t\tif (RECORDS)\n\t\t\tmvo = a.draws[s_q1[wave][mine ? lane : 0u].z].meshletVisibilityOffset;"
    use = "\t\tif (RECORDS)\n\t\t\tasm volatile(\"\" ::\"v\"(mvo));"
    assert gather in text and use in text
    bad = text.replace(gather, "\t\tif (RECORDS && mine)\n\t\t\tmvo = a.draws[s_q1[wave][lane].z].meshletVisibilityOffset;").replace(use, "")
    copy = os.path.join(os.path.dirname(src), "_ring_regression.hip")
    try:
        open(copy, "w").write(bad)
        r = subprocess.run([sys.executable, tool, "--hipcc", hipcc, "--src", copy], capture_output=True, text=True)
        assert r.returncode == 1 and "vmcnt(1)" in r.stdout, r.stdout
    finally:
        os.remove(copy)
