#!/bin/bash
# A/B of library variants (GPU box): headline step / k_bwd alone (tools/mall_probe.py) and the bench lines of configs 3 / 5.
# usage: tools/ab_all.sh OUTDIR REPS lib ...      (lib: build/variants/librrt_<lib>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
REPS=$1; shift
mkdir -p $OUT
cd $R
for rep in $(seq $REPS); do
for v in "$@"; do
  echo "== $v headline" >> $OUT/ab.txt
  RRT_LIB=$R/build/variants/librrt_$v.so timeout -k 10 200 python3 tools/mall_probe.py --rows 4096 >> $OUT/ab.txt 2>$OUT/err.txt || exit 1
  for w in c3 c5; do
    echo "== $v $w" >> $OUT/ab.txt
    RRT_LIB=$R/build/variants/librrt_$v.so timeout -k 10 300 python3 bench.py --workload $w --cpu-rows 0 >> $OUT/ab.txt 2>$OUT/err.txt || exit 1
  done
done
done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): name = l.strip()
    elif l.startswith("{"):
        d = json.loads(l)
        if "rows" in d: print("%-22s step %.2f k_bwd_alone %.2f bwd_call %.2f" % (name, d["step_loop_us"], d["k_bwd_alone_loop_us"], d["bwd_call_loop_us"]))
        else: print("%-22s us/step %.2f" % (name, d["ms_per_step"]*1e3), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})
PY
