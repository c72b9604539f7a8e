#!/usr/bin/env python3
"""check_decide_ring.py — ISA check of drawcull.hip's compiler-counted request ring (draw_decide_kernel).

The walk of a decide wave keeps DS_DEPTH units of requests in flight.  Nothing in it is inline asm, so hipcc places the waits — and twice in round 6 it
placed them so that the ring drained, with right results and a slower kernel: copies of in-flight registers on the loop's back edge (a slot re-requested
before the last use of its old value), and a `vmcnt(1)` in front of every unit (a request inside the walk whose use was conditional).  Both show in the
compiled text as waits with a small count INSIDE the walk's loop.  The drains of the survivor queue (conditional blocks of the loop) do wait for their own
requests with small counts; they are recognised by their guard (s_cmp_lt_u32 sN, 64 / s_cbranch_scc1 past the drain).  Everything else in the loop must
wait with vmcnt(N), N >= MIN_STEADY.

    python3 tools/check_decide_ring.py [--hipcc /opt/rocm/bin/hipcc] [--src copy.hip]      exit status 1 on a finding
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "niagara_amd", "csrc", "drawcull.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S"]
MIN_STEADY = 4      # a steady-state wait leaves at least the other slots' requests outstanding (3-4 loads per slot, DS_DEPTH = 4 slots)


def functions(isa):
    """(name, [lines]) per draw_decide_kernel instantiation, up to .Lfunc_end"""
    out, cur, name = [], None, None
    for ln in isa.split("\n"):
        m = re.match(r"^(_ZN2nv18draw_decide_kernel\w+):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                out.append((name, cur))
                cur = None
            else:
                cur.append(ln)
    return out


def walk_loop(lines):
    """[first, last] line index of the outermost loop that re-requests the ring (the last Depth=1 loop header of the function that contains global loads)"""
    best = None
    for i, ln in enumerate(lines):
        m = re.search(r"=>This Loop Header: Depth=1", ln)
        if not m:
            continue
        label = None
        for j in range(i, max(i - 4, -1), -1):
            mm = re.match(r"^(\.LBB\d+_\d+):", lines[j])
            if mm:
                label = mm.group(1)
                break
        if not label:
            continue
        tag = label[2:]  # BBn_m
        last = i
        for j in range(i, len(lines)):
            if ("Header=%s Depth=1" % tag) in lines[j] or ("Parent Loop %s Depth=1" % tag) in lines[j]:
                last = j
        # the block that holds the back edge runs to its terminator
        for j in range(last, min(last + 400, len(lines))):
            if re.match(r"\s*s_(c?branch\w*)\s+%s\b" % re.escape(label), lines[j]):
                last = j
                break
        body = lines[i:last + 1]
        if sum(1 for b in body if re.match(r"\s*global_load_dwordx4", b)) >= 2:
            best = (i, last)
    return best


def drain_regions(lines, first, last):
    """line ranges of the queue's drains inside the loop: `if (queued >= 64) drain(64)` compiles to s_cmp_lt_u32 sN, 64 ... s_cbranch_scc1 <past the drain>"""
    out = []
    for i in range(first, last + 1):
        if not re.match(r"\s*s_cmp_lt_u32\s+s\d+,\s*64\b", lines[i]):
            continue
        for j in range(i + 1, min(i + 12, last + 1)):
            m = re.match(r"\s*s_cbranch_scc1\s+(\.LBB\d+_\d+)", lines[j])
            if m:
                for k in range(j + 1, last + 1):
                    if lines[k].startswith(m.group(1) + ":"):
                        out.append((j, k))
                        break
                break
    return out


def scan(name, lines):
    loop = walk_loop(lines)
    if not loop:
        return None, []
    first, last = loop
    drains = drain_regions(lines, first, last)
    findings = []
    if len(drains) < 2:
        findings.append((first + 1, "fewer than two drains recognised inside the walk (s_cmp_lt_u32 sN, 64 / s_cbranch_scc1): the check no longer knows this code"))
    for i in range(first, last + 1):
        if any(a <= i <= b for a, b in drains):
            continue
        m = re.match(r"\s*s_waitcnt\s+vmcnt\((\d+)\)", lines[i])
        if m and int(m.group(1)) < MIN_STEADY:
            findings.append((i + 1, lines[i].strip()))
    return loop, findings


def main():
    hipcc, src = "/opt/rocm/bin/hipcc", SRC
    args = sys.argv[1:]
    while args[:1] and args[0] in ("--hipcc", "--src"):
        if args[0] == "--hipcc":
            hipcc = args[1]
        else:
            src = os.path.abspath(args[1])  # (a modified copy next to the original: the checker's own test)
        args = args[2:]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dc.s")
        subprocess.check_call([hipcc] + FLAGS + [src, "-o", out], cwd=os.path.dirname(SRC), stderr=subprocess.DEVNULL)
        isa = open(out).read()
    bad = 0
    funcs = functions(isa)
    checked = 0
    for name, lines in funcs:
        # <LATE, TASK, MESH_LDS, ...>: the kernels that gather the Mesh table from global memory inside the drain (no registered table, or more than 64 meshes)
        # wait for those gathers with small counts well past the window; the BASELINE paths all stage the table
        if re.match(r"_ZN2nv18draw_decide_kernelILb[01]ELb[01]ELb0E", name):
            continue
        checked += 1
        loop, findings = scan(name, lines)
        if loop is None:
            print("%s: no walk loop found" % name)
            bad += 1
            continue
        if findings:
            print("%s: %d wait(s) inside the walk (function lines %d-%d) that drain the ring:" % (name, len(findings), loop[0] + 1, loop[1] + 1))
            for no, text in findings[:8]:
                print("   +%d  %s" % (no, text))
        bad += len(findings)
    if not bad:
        print("decide ring: %d kernels (of %d; those with the Mesh table in LDS), every wait of the walk outside the queue's drains leaves >= %d requests outstanding" % (checked, len(funcs), MIN_STEADY))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
