#!/bin/bash
# usage: tools/sweep_env.sh REPS VAR v1 v2 ... -- [bench.py args]   (GPU box; RRT_LIB may be set; prints us/step and kernel times per value)
reps=$1; shift
var=$1; shift
vals=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq $reps); do
for v in "${vals[@]}"; do
  echo -n "$var=$v "
  env "$var=$v" timeout -k 10 200 python3 bench.py --cpu-rows 0 "$@" 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read())
print("us/step %.2f" % (d["ms_per_step"] * 1e3), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})'
done
done
