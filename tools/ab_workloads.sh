#!/bin/bash
# A/B of library variants on the secondary workloads (GPU box): bench lines for configs 3 and 5 and antialias_samples = 4.
# usage: tools/ab_workloads.sh OUTDIR lib ...      (lib: build/variants/librrt_<lib>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in "$@"; do
  for w in "--workload c3" "--workload c5" "--aa 4"; do
    echo "== $v $w" >> $OUT/ab.txt
    RRT_LIB=$R/build/variants/librrt_$v.so timeout -k 10 300 python3 bench.py $w --cpu-rows 0 >> $OUT/ab.txt 2>$OUT/err.txt || exit 1
  done
done
done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip(), end=" ")
    elif l.startswith("{"):
        d = json.loads(l); print("us/step %.2f" % (d["ms_per_step"]*1e3), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})
PY
