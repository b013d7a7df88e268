#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... -- [bench.py args]   (GPU box; prints ms/step and kernel times per value)
var=$1; shift
vals=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  echo -n "$var=$v "
  env "$var=$v" timeout -k 10 200 python bench.py --cpu-rows 0 "$@" 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read())
print("ms/step %.4f" % d["ms_per_step"], {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})'
done
