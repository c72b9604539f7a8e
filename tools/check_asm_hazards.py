#!/usr/bin/env python3
"""check_asm_hazards.py — guards the one hazard class hipcc cannot see (clustercull.hip, rules above SlotA).

The hazard recognizer of the compiler does not look inside inline asm.  On gfx950 an SGPR written by a VALU instruction
(v_readlane_b32 / v_readfirstlane_b32 — also the reload of a spilled kernel-argument pointer) needs 2 wait states before
a VALU instruction reads it and 5 before a VMEM instruction uses it as an address base.  Every inline-asm statement that
reads SGPRs therefore starts with its own s_nop.  This script compiles the file to ISA and checks exactly that: for
every ;;#ASMSTART block that reads an SGPR produced by a VALU instruction within the preceding few instructions, the
block must begin with an s_nop that covers the consumer (>= 1 for VALU, >= 4 for VMEM).

    python tools/check_asm_hazards.py            # exit code 1 and a listing if a block is unguarded
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


def sgprs(text):
    regs = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def scan(isa):
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
            ops = ins.split(None, 1)
            if len(ops) == 2 and not ins.startswith("s_nop") and not ins.startswith("s_waitcnt"):
                # destination operands of loads are VGPRs; every sN / s[a:b] in the operand list is a read (s_mov m0 excepted: still a read)
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


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "cc.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [SRC, "-o", out], cwd=os.path.dirname(SRC), stderr=subprocess.DEVNULL)
        blocks, guarded, problems = scan(open(out).read())
    print("inline-asm blocks: %d, of which fed by a VALU-written SGPR and guarded by s_nop: %d, unguarded: %d" % (blocks, guarded, len(problems)))
    for line, body, producer in problems:
        print("  ISA line %d: %s   <=   %s" % (line, " | ".join(body), producer))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
