#!/usr/bin/env python3
"""ref_translate.py — rewrite one of the reference's GLSL compute shaders into compilable C++.

TEST INFRASTRUCTURE.  Reads the shader where it lies under the reference tree (never copied into this
repository), inlines its #includes, and applies purely syntactic rewrites so that the reference's own
statements compile against oracle/glsl_shim.h.  Output goes to oracle/_ref/ (git-ignored).

    python ref_translate.py /root/reference/src/shaders/drawcull.comp.glsl oracle/_ref/drawcull.gen.h

Rewrites (nothing semantic):
  * `#version` / `#extension` lines dropped; `#include "x"` inlined relative to the including file;
  * `layout(constant_id=N) const bool X = false;`  ->  `bool X = false;` (set by the runner);
  * `layout(local_size_x=...) in;` dropped (the runner owns the invocation loop);
  * storage blocks:  `buffer B { T a[]; };` -> `T* a;`   `buffer B { uint a; uint b; };` -> struct + pointer +
    `#define a (B_buf->a)`;
  * `layout(push_constant) uniform block { ... };` -> the members as globals;
  * `uniform texture2D/sampler/image2D name;` -> plain globals of the shim's types;
  * `out T name` parameters -> `T& name`;
  * float literals get an `f` suffix (GLSL literals are fp32);
  * the rvalue swizzles .xy .zw .xyz .xwzy -> member calls;
  * `void main()` -> `void shader_main()`.
  * mesh-shader declarations (meshlet.mesh.glsl): `layout(triangles, ...) out;` dropped, `layout(location=N) out [flat] T
    name[];` -> `T name[256];`, `taskPayloadSharedEXT` / `shared` qualifiers dropped (the runner serialises a workgroup).
`--limit math.h:49` keeps only the first 49 lines of that include (the cull helpers; the rest is shading).
`--define NAME=V` flips one of the reference's own `#define NAME ...` configuration switches (MESH_CULL, src/config.h:10-11).
`--once "stmt;"` guards ONE statement with `if (gl_LocalInvocationID.x == 0u)`: a workgroup-shared variable that every
invocation initialises in front of a barrier() (meshlet.task.glsl:64-65) is initialised once when the runner serialises
the invocations of a workgroup — the only way barrier() can be honoured without threads.
Also extracts line ranges of host C++ (PCG32, previousPow2, projection) with --lines.
"""
import os
import re
import sys


LIMITS = {}  # basename -> number of leading lines to keep (e.g. math.h:49 = the cull helpers only)
DEFINES = {}  # NAME -> value: rewrites the reference's own `#define NAME x` line
ONCE = []  # statements executed by the first invocation of a workgroup only (barrier emulation)


def inline_includes(path, seen=None):
    seen = seen or set()
    out = []
    base = os.path.dirname(path)
    lines = open(path).readlines()
    limit = LIMITS.get(os.path.basename(path))
    if limit:
        lines = lines[:limit]
    for line in lines:
        m = re.match(r'\s*#include\s+"([^"]+)"', line)
        if m:
            inc = os.path.normpath(os.path.join(base, m.group(1)))
            if inc not in seen:
                seen.add(inc)
                out.append(inline_includes(inc, seen))
            continue
        out.append(line)
    return "".join(out)


def translate(src):
    src = re.sub(r"^\s*#(version|extension)[^\n]*\n", "", src, flags=re.M)
    src = re.sub(r"layout\s*\(\s*constant_id\s*=\s*\d+\s*\)\s*const\s+bool\s+(\w+)\s*=\s*(\w+)\s*;", r"bool \1 = \2;", src)
    src = re.sub(r"layout\s*\(\s*local_size_x[^)]*\)\s*in\s*;", "", src)
    src = re.sub(r"layout\s*\(\s*triangles[^)]*\)\s*out\s*;", "", src)
    src = re.sub(r"layout\s*\(\s*location\s*=\s*\d+\s*\)\s*out\s+(?:flat\s+)?(\w+)\s+(\w+)\s*\[\s*\]\s*;", r"\1 \2[256];", src)
    src = re.sub(r"\btaskPayloadSharedEXT\s+", "", src)
    src = re.sub(r"^shared\s+", "", src, flags=re.M)
    for name, value in DEFINES.items():
        src, n = re.subn(r"^(\s*#define\s+%s)\s+\S+[^\n]*$" % re.escape(name), r"\1 %s" % value, src, flags=re.M)
        if n != 1:
            raise SystemExit("--define %s: expected exactly one #define in the reference, found %d" % (name, n))

    for stmt in ONCE:
        if src.count(stmt) != 1:
            raise SystemExit("--once %r: expected exactly one occurrence in the reference, found %d" % (stmt, src.count(stmt)))
        src = src.replace(stmt, "if (gl_LocalInvocationID.x == 0u) " + stmt)

    def block(m):
        name, body = m.group(1), m.group(2)
        members = [x.strip() for x in body.split(";") if x.strip()]
        if len(members) == 1 and members[0].endswith("[]"):
            ty, var = members[0][:-2].rsplit(None, 1)
            return "%s* %s;" % (ty, var)
        out = ["struct %s_t {" % name]
        defs = []
        for mem in members:
            ty, var = mem.rsplit(None, 1)
            out.append("\t%s %s_;" % (ty, var))
            defs.append("#define %s (%s_buf->%s_)" % (var, name, var))
        out.append("};")
        out.append("%s_t* %s_buf;" % (name, name))
        return "\n".join(out + defs)

    src = re.sub(r"layout\s*\(\s*binding\s*=\s*\d+\s*\)\s*(?:readonly\s+|writeonly\s+)?buffer\s+(\w+)\s*\{([^}]*)\}\s*;", block, src)
    src = re.sub(r"layout\s*\(\s*push_constant\s*\)\s*uniform\s+\w+\s*\{([^}]*)\}\s*;", lambda m: m.group(1).strip(), src)
    src = re.sub(r"layout\s*\([^)]*\)\s*uniform\s+(?:writeonly\s+|readonly\s+)?(texture2D|sampler|image2D)\s+(\w+)\s*;", r"\1 \2;", src)
    src = re.sub(r"\bout\s+(vec[234]|float|uint|int)\s+(\w+)", r"\1& \2", src)
    src = re.sub(r"(?<![\w.])(\d+\.\d*|\.\d+)(?![\w.])", r"\1f", src)
    src = re.sub(r"\.(xyz|xy|zw|xwzy)\b(?!\s*\()", r".\1()", src)
    src = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", src)
    return src


def main():
    args = sys.argv[1:]
    if args[0] == "--lines":
        # --lines FILE A-B[,C-D...] OUT : verbatim line ranges of host C++ (1-based, inclusive)
        path, ranges, out = args[1], args[2], args[3]
        lines = open(path).read().split("\n")
        chunks = []
        for r in ranges.split(","):
            a, b = r.split("-")
            chunks.append("\n".join(lines[int(a) - 1:int(b)]))
        open(out, "w").write("\n\n".join(chunks) + "\n")
        return
    while args[0] in ("--limit", "--define", "--once"):
        if args[0] == "--limit":
            name, n = args[1].split(":")
            LIMITS[name] = int(n)
        elif args[0] == "--once":
            ONCE.append(args[1])
        else:
            name, v = args[1].split("=")
            DEFINES[name] = v
        args = args[2:]
    path, out = args
    open(out, "w").write("// generated from %s by oracle/ref_translate.py — do not commit\n" % path + translate(inline_includes(path)))


if __name__ == "__main__":
    main()
