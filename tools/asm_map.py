#!/usr/bin/env python3
"""Where a kernel's scratch accesses, barriers, priority switches and transcendentals sit in its ISA.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -I include \
        -I reversible-raytracer_amd/csrc reversible-raytracer_amd/csrc/rrt_kernels.hip -o /tmp/k.s
    python tools/asm_map.py /tmp/k.s k_bwdILi4ELi0ELi1ELi1ELi1ELi0EE

The row loops are bracketed by s_setprio 0 / s_setprio 2: scratch traffic between them is in a hot loop.
"""
import sys

lines = open(sys.argv[1]).read().split("\n")
for name in sys.argv[2:]:
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    print(name, len(body), "lines")
    ev = []
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith("scratch_"):
            ev.append((i, "SCRATCH " + t.split()[0]))
        elif t.startswith("s_setprio"):
            ev.append((i, " ".join(t.split()[:2])))
        elif t.startswith(("v_log_f32", "v_exp_f32", "v_sqrt_f32", "v_rsq_f32")):
            ev.append((i, t.split()[0]))
        elif t.startswith("s_barrier"):
            ev.append((i, "barrier"))
        elif t.startswith(("v_readlane", "v_writelane")):
            ev.append((i, "lane-spill?"))
    out, last, cnt, first = [], None, 0, 0
    for i, e in ev:
        if e == last:
            cnt += 1
        else:
            if last:
                out.append((first, last, cnt))
            last, cnt, first = e, 1, i
    out.append((first, last, cnt))
    for o in out:
        print("  line %5d  %s x%d" % o)
