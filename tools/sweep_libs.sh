#!/bin/bash
# usage: tools/sweep_libs.sh lib1.so lib2.so ...   (GPU box; bench.py with RRT_LIB set to each)
for l in "$@"; do
  echo -n "$l "
  RRT_LIB=$PWD/$l timeout -k 10 200 python bench.py --cpu-rows 0 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read())
print("ms/step %.4f" % d["ms_per_step"], {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})'
done
