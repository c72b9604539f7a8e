#!/usr/bin/env python3
"""check_asm_hazards.py — ISA scan of clustercull.hip's inline-asm load rings; part of the build (niagara_amd/csrc/Makefile).

The rings issue global loads from inline asm ("=&v" destinations) and wait for them later with hand-counted
`s_waitcnt vmcnt(N)` statements that name the slot's registers ("+v").  hipcc is told the destination already holds the
value at the issue statement, so nothing but its register allocation keeps it from copying, spilling or reading a slot
before the wait, and its hazard recognizer does not look inside inline asm.  Two checks over the compiled ISA of every
kernel in the file:

  1. s_nop guard.  On gfx950 an SGPR written by a VALU instruction (v_readlane_b32 / v_readfirstlane_b32 — also the
     reload of a spilled kernel-argument pointer) needs 2 wait states before a VALU instruction reads it and 5 before a
     VMEM instruction uses it as an address base: every asm block fed by such an SGPR within the preceding few
     instructions must begin with a sufficient s_nop.
  2. in-flight registers.  Forward dataflow over the kernel's control-flow graph: a VGPR written by an asm load is
     "in flight" until an asm wait names it (the wait statements print their registers as `; nv_ready v[..] ..`, a drain
     as `; nv_ready all`).  Any other instruction that reads or writes an in-flight VGPR (a v_mov the allocator inserted
     to split a live range, a spill, a compiler-scheduled use) is reported, and so is an in-flight register at s_endpgm.

  4. address spaces.  No flat_* instruction in any kernel of the file (an LDS pointer kept across a ring decays to a generic one: flat loads count on vmcnt
     and hipcc then waits for the whole ring per iteration — slow, not wrong).

  3. counted waits.  `s_waitcnt vmcnt(N)` returns once at most N vector-memory operations are outstanding, and they complete in
     order, so a wait that names a slot is sufficient only if AT LEAST N younger VMEM operations (loads, stores and atomics: all
     of them count on gfx9) were issued after the slot's load on EVERY path to the wait — with fewer, the slot's load may be among
     the N that are allowed to be outstanding.  Forward dataflow again: per in-flight VGPR the minimum over all paths of the
     number of VMEM instructions issued since its load; a wait `vmcnt(N) ; nv_ready regs` with a smaller minimum is reported
     (a wait that names nothing — the compiler's own, a ring's first-round wait — lands every register with at least its N behind it).
     (More younger operations than counted only make a wait stricter.)

The rings were validated on ONE compiler (VALIDATED_HIPCC below: register allocation and scheduling around the asm
statements are what the scan certifies, and they change with the compiler); the scan refuses any other unless
NV_ALLOW_UNVALIDATED_HIPCC=1, in which case it still scans and says so.

    python tools/check_asm_hazards.py [--hipcc PATH] [-D MACRO ...]     # exit code 1 and a listing if anything is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "niagara_amd", "csrc", "clustercull.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S"]
LOOKBACK = 6  # instructions
VALIDATED_HIPCC = "roc-7.2.0"  # substring of `hipcc --version` (AMD clang 22.0.0git, ROCm 7.2.0)


def compiler_is_validated(hipcc):
    try:
        text = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=120).stdout
    except OSError as e:
        print("cannot run %s: %s" % (hipcc, e))
        return False
    if VALIDATED_HIPCC in text:
        return True
    print("clustercull.hip's inline-asm load rings were validated with hipcc %s; this is:\n%s" % (VALIDATED_HIPCC, text.strip()))
    if os.environ.get("NV_ALLOW_UNVALIDATED_HIPCC") == "1":
        print("NV_ALLOW_UNVALIDATED_HIPCC=1: scanning anyway — re-run tests/test_plain_loads.py and the soak tools before trusting this build")
        return True
    print("refusing to build the asm rings with an unvalidated compiler (set NV_ALLOW_UNVALIDATED_HIPCC=1 to scan and build anyway, or "
          "build with -DNV_PLAIN_LOADS)")
    return False


def sgprs(text):
    regs = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def vgprs(text):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def scan_nops(isa):
    lines = isa.split("\n")
    problems, blocks, guarded = [], 0, 0
    i = 0
    while i < len(lines):
        if ";;#ASMSTART" not in lines[i]:
            i += 1
            continue
        start = i
        body = []
        i += 1
        while i < len(lines) and ";;#ASMEND" not in lines[i]:
            if lines[i].strip():
                body.append(lines[i].strip())
            i += 1
        blocks += 1
        reads = set()
        for ins in body:
            ops = ins.split(";")[0].split(None, 1)
            if len(ops) == 2 and not ins.startswith("s_nop") and not ins.startswith("s_waitcnt"):
                reads |= sgprs(ops[1])
        if not reads:
            continue
        prev = [x.strip() for x in lines[max(0, start - 40):start] if x.strip() and not x.strip().startswith((";", "."))][-LOOKBACK:]
        producers = [p for p in prev if re.match(r"v_read(first)?lane_b32 s(\d+)", p) and int(re.match(r"v_read(?:first)?lane_b32 s(\d+)", p).group(1)) in reads]
        if not producers:
            continue
        vmem = any(ins.startswith(("global_load", "global_store", "buffer_", "flat_")) for ins in body)
        need = 4 if vmem else 1
        m = re.match(r"s_nop (\d+)", body[0]) if body else None
        if m and int(m.group(1)) >= need:
            guarded += 1
        else:
            problems.append((start + 1, body[:2], producers[-1]))
    return blocks, guarded, problems


def kernels(isa):
    """(name, [lines]) per function of the .s file"""
    out, cur, name = [], None, None
    for ln in isa.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(ln)
            if ln.strip().startswith("s_endpgm"):
                out.append((name, cur))
                cur = None
    return out


def control_flow(lines):
    """(blocks, successors): blocks = label-delimited lists of (line number in kernel, instruction, kind), kind in asm / ins / ctl"""
    # ---- split into basic blocks
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    in_asm = False
    for no, raw in enumerate(lines):
        t = raw.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            nxt = m.group(1)
            blocks.setdefault(cur, [])
            blocks[cur].append((no, "__fall__ " + nxt, "ctl"))
            cur = nxt
            blocks[cur] = []
            continue
        if ";;#ASMSTART" in t:
            in_asm = True
            continue
        if ";;#ASMEND" in t:
            in_asm = False
            continue
        if not t or t.startswith((";", ".")):
            continue
        blocks[cur].append((no, t, "asm" if in_asm else "ins"))
        if not in_asm and re.match(r"s_(branch|cbranch_\w+|endpgm|setpc)", t):
            # a branch ends the block; what follows falls into an anonymous block
            nxt = "%s.after%d" % (cur, no)
            if not t.startswith(("s_branch", "s_endpgm", "s_setpc")):
                blocks[cur].append((no, "__fall__ " + nxt, "ctl"))
            cur = nxt
            blocks[cur] = []
    succ = {}
    for b, ins in blocks.items():
        s = []
        for _, t, kind in ins:
            if kind == "ctl":
                s.append(t.split()[1])
            elif kind == "ins":
                m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
                if m:
                    s.append(m.group(1))
        succ[b] = [x for x in s if x in blocks]
    return blocks, succ


VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store", "flat_atomic",
        "scratch_load", "scratch_store")
CAP = 64  # vmcnt is a 6-bit counter: more younger operations than that make no difference


def scan_counts(name, lines):
    """check 3: [(line number in kernel, wait statement, register, younger VMEM operations on the weakest path, N)]"""
    blocks, succ = control_flow(lines)
    state_in = {b: None for b in blocks}  # None = not reached yet; else {vgpr: min younger VMEM ops since its asm load}
    state_in["entry"] = {}
    problems = {}
    work = ["entry"]
    while work:
        b = work.pop(0)
        st = dict(state_in[b])
        for no, t, kind in blocks[b]:
            if kind == "ctl":
                continue
            body = t.split(";")[0].strip()
            if body.startswith(VMEM):
                for r in st:
                    st[r] = min(CAP, st[r] + 1)
                if kind == "asm" and body.startswith(("global_load", "buffer_load", "flat_load")) and not body.startswith("global_load_lds"):
                    # (LDS-DMA — prefetch_candidate — has no VGPR destination: its first operand is the address; it still counts as a
                    #  younger VMEM operation above)
                    for r in vgprs(body.split(None, 1)[1].split(",")[0]):
                        st[r] = 0
            elif "nv_ready" not in t and re.search(r"s_waitcnt\b.*vmcnt\((\d+)\)", body):
                # a wait that names nothing (the compiler's own, or a ring's first-round wait): whatever has at least that many
                # younger operations behind it on every path has landed
                n = int(re.search(r"vmcnt\((\d+)\)", body).group(1))
                for r in [r for r, k in st.items() if k >= n]:
                    del st[r]
            elif kind == "asm" and "nv_ready" in t:
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", body)
                ready = t.split("nv_ready", 1)[1]
                regs = list(st) if "all" in ready else [r for r in vgprs(ready) if r in st]
                for r in regs:
                    if m and st[r] < int(m.group(1)):
                        problems[(no, t, r)] = (st[r], int(m.group(1)))
                    del st[r]
        for nx in succ[b]:
            old = state_in[nx]
            if old is None:
                merged = dict(st)
            else:
                merged = dict(old)
                for r, v in st.items():
                    merged[r] = min(v, merged[r]) if r in merged else v
            if merged != old:
                state_in[nx] = merged
                if nx not in work:
                    work.append(nx)
    return [(no, t, r, k, n) for (no, t, r), (k, n) in sorted(problems.items())]


def scan_inflight(name, lines):
    """forward dataflow: blocks = label-delimited; returns a list of (line number in kernel, instruction, registers)"""
    blocks, succ = control_flow(lines)
    # ---- iterate
    state_in = {b: set() for b in blocks}
    problems = {}
    work = list(blocks)
    seen_once = set()
    while work:
        b = work.pop(0)
        st = set(state_in[b])
        for no, t, kind in blocks[b]:
            if kind == "ctl":
                continue
            body = t.split(";")[0]
            if kind == "asm":
                if body.startswith("global_load_lds"): # LDS-DMA: reads its address VGPR, writes LDS — nothing goes in flight
                    touched = vgprs(body) & st
                    if touched:
                        problems[(no, t)] = touched
                elif body.startswith(("global_load", "buffer_load", "flat_load")):
                    ops = body.split(None, 1)[1].split(",")
                    st |= vgprs(ops[0])
                    touched = vgprs(",".join(ops[1:])) & st
                    if touched:
                        problems[(no, t)] = touched
                elif "nv_ready" in t:
                    ready = t.split("nv_ready", 1)[1]
                    if "all" in ready:
                        st.clear()
                    else:
                        st -= vgprs(ready)
                else:
                    touched = vgprs(body) & st
                    if touched:
                        problems[(no, t)] = touched
                continue
            if body.startswith("s_endpgm") and st:
                problems[(no, t)] = set(st)
            touched = vgprs(body) & st
            if touched:
                problems[(no, t)] = touched
        for s in succ[b]:
            if not st <= state_in[s] or s not in seen_once:
                seen_once.add(s)
                if not st <= state_in[s]:
                    state_in[s] |= st
                    if s not in work:
                        work.append(s)
                elif s not in work and s not in seen_once:
                    work.append(s)
    return [(no, t, sorted(r)) for (no, t), r in sorted(problems.items())]


def main():
    defines = []
    hipcc = "/opt/rocm/bin/hipcc"
    args = sys.argv[1:]
    while args and args[0] in ("-D", "--hipcc"):
        if args[0] == "-D":
            defines.append("-D" + args[1])
        else:
            hipcc = args[1]
        args = args[2:]
    if not compiler_is_validated(hipcc):
        return 1
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "cc.s")
        subprocess.check_call([hipcc] + FLAGS + defines + [SRC, "-o", out], cwd=os.path.dirname(SRC), stderr=subprocess.DEVNULL)
        isa = open(out).read()
    blocks, guarded, problems = scan_nops(isa)
    print("inline-asm blocks: %d, of which fed by a VALU-written SGPR and guarded by s_nop: %d, unguarded: %d" % (blocks, guarded, len(problems)))
    for line, body, producer in problems:
        print("  ISA line %d: %s   <=   %s" % (line, " | ".join(body), producer))
    bad = len(problems)
    for name, lines in kernels(isa):
        found = scan_inflight(name, lines)
        if found:
            print("%s: %d instruction(s) touch a ring register whose load is still in flight" % (name, len(found)))
            for no, t, regs in found[:12]:
                print("   +%d  %s    [%s]" % (no, t, ", ".join("v%d" % r for r in regs)))
        bad += len(found)
    weak = 0
    waits = 0
    for name, lines in kernels(isa):
        waits += sum(1 for ln in lines if "nv_ready" in ln)
        found = scan_counts(name, lines)
        if found:
            print("%s: %d counted wait(s) that may return before the load they name has landed" % (name, len(found)))
            for no, t, r, k, n in found[:12]:
                print("   +%d  %s    v%d: only %d younger VMEM operation(s) on some path, the wait allows %d outstanding" % (no, t, r, k, n))
        weak += len(found)
    bad += weak
    # check 4 (round 6): no FLAT memory instruction in the file's kernels.  Every pointer here is global (kernel arguments) or LDS; an LDS pointer that is kept in a
    # register across an asm ring decays to a generic one, hipcc then emits flat loads — which count on vmcnt AND lgkmcnt — and waits for the whole ring in front
    # of every iteration (the packed walk lost a third of its speed that way before its table offsets were integers again); results stay right, so only this sees it
    flat = 0
    for name, lines in kernels(isa):
        hits = [ln.strip() for ln in lines if re.match(r"\s*flat_(load|store|atomic)", ln)]
        if hits:
            print("%s: %d flat memory instruction(s) — an LDS or global pointer lost its address space (first: %s)" % (name, len(hits), hits[0]))
        flat += len(hits)
    bad += flat
    if not flat:
        print("address spaces: no flat memory instruction in any kernel")
    if not bad:
        print("in-flight scan: no instruction touches a ring register between its issue and its wait (%d kernels)" % len(kernels(isa)))
        print("counted waits: every vmcnt(N) that names a slot has at least N younger vector-memory operations behind the slot's load on every path (%d waits, no exemptions)" % waits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
