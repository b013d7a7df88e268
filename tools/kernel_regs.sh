#!/bin/bash
# register / scratch use of every kernel instantiation: tools/kernel_regs.sh [-DFLAG ...] | grep k_bwd
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -c -I $R/include -I $R/reversible-raytracer_amd/csrc "$@" \
  $R/reversible-raytracer_amd/csrc/rrt_kernels.hip -o /tmp/kernel_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m: cur = {"name": m.group(1)}; continue
    if cur is None: continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, l)
        if m: cur[key] = int(m.group(1))
    if "lds" in cur:
        print("%-70s vgpr %3d  vgpr-spill %3d  sgpr-spill %3d  scratch %4d  lds %6d" % (cur["name"], cur.get("vgpr", -1), cur.get("vspill", -1), cur.get("sspill", -1), cur.get("scratch", -1), cur["lds"]))
        cur = None
'
