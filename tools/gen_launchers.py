#!/usr/bin/env python3
"""tools/gen_launchers.py -- rewrites the launcher section at the end of raytracer_amd/csrc/rt_kernels.hip and the
declarations at the end of rt_shared.h from the __global__ signatures in rt_kernels.hip (run after changing a kernel's
parameters).  A launcher is rtk_<name>( [template flags as bools,] grid, block, dynamic LDS bytes, stream, kernel arguments )."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERN = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_kernels.hip")
SHARED = os.path.join(ROOT, "raytracer_amd", "csrc", "rt_shared.h")
MARK_K = "// ---- launchers: all the host side (rtmi_host.cpp) sees of the kernels above"
MARK_H = "// ---- launchers (rt_kernels.hip)"

src = open(KERN).read()
body = src[:src.index(MARK_K)]
decl, defs = [], []
for m in re.finditer(r'(template <([^>]*)>\n)?__global__ (?:__launch_bounds__\([^)]*\) )?void (k_\w+)\(([^{]*?)\)\n\{', body, re.S):
    tmpl, name, params = m.group(2), m.group(3), " ".join(m.group(4).split())
    plist = [p.strip() for p in params.split(",")]
    names = ", ".join(re.findall(r"(\w+)$", p)[0] for p in plist)
    clean = ", ".join(p.replace("__restrict__ ", "") for p in plist)
    flags = [t.strip().split()[-1] for t in tmpl.split(",")] if tmpl else []
    sig = "void rt%s( %sdim3 grid, dim3 block, size_t lds, hipStream_t stream, %s )" % (name, "".join("bool %s, " % f.lower() for f in flags), clean)
    decl.append(sig + ";")
    if not flags:
        code = "\thipLaunchKernelGGL( %s, grid, block, lds, stream, %s );" % (name, names)
    else:
        rows = []
        for k in range(2 ** len(flags) - 1, -1, -1):
            vals = [(k >> (len(flags) - 1 - i)) & 1 for i in range(len(flags))]
            cond = " && ".join(("" if v else "!") + f.lower() for f, v in zip(flags, vals))
            rows.append("\t%sif (%s) hipLaunchKernelGGL( ( %s<%s> ), grid, block, lds, stream, %s );" % ("else " if rows else "", cond, name, ", ".join("true" if v else "false" for v in vals), names))
        code = "\n".join(rows)
    defs.append(sig + "\n{\n" + code + "\n}")
line = src[src.index(MARK_K):].split("\n")[0]
open(KERN, "w").write(body + line + "\n" + "\n".join(defs) + "\n")
h = open(SHARED).read()
hl = h[h.index(MARK_H):].split("\n")[0]
open(SHARED, "w").write(h[:h.index(MARK_H)] + hl + "\n" + "\n".join(decl) + "\n")
print("%d launchers" % len(defs))
