#!/bin/bash
# A/B of library variants and tuning knobs (GPU box): bench + band probe per spec, then optionally the GPU tests.
# usage: tools/r3_batch2.sh OUTDIR test_spec|none spec ...      spec = lib[,VAR=value...]  (lib: build/variants/librrt_<lib>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
TESTV=$1; shift
mkdir -p $OUT
cd $R
run() {  # spec, command...
  local spec=$1; shift
  local lib=${spec%%,*}; local rest=${spec#*,}; [ "$rest" == "$spec" ] && rest=""
  env $(echo $rest | tr ',' ' ') RRT_LIB=$R/build/variants/librrt_$lib.so "$@"
}
for rep in 1 2; do
for v in "$@"; do
  echo "== bench $v" >> $OUT/sweep.txt
  run $v timeout -k 10 300 python3 bench.py --cpu-rows 0 >> $OUT/sweep.txt 2>$OUT/err_bench.txt || exit 1
done
done
python3 - <<PY
import json
for l in open("$OUT/sweep.txt"):
    if l.startswith("=="): print(l.strip(), end=" ")
    elif l.startswith("{"):
        d = json.loads(l); print("us/step %.2f" % (d["ms_per_step"]*1e3), {k: round(v, 1) for k, v in d["kernels"].items() if k.endswith("_us")})
PY
for v in "$@"; do
  echo "== $v" >> $OUT/probe.txt
  run $v timeout -k 10 300 python3 tools/mall_probe.py --rows 4096,512 >> $OUT/probe.txt 2>>$OUT/err_probe.txt || exit 1
done
cat $OUT/probe.txt
if [ "$TESTV" != "none" ]; then
  run $TESTV timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
  echo pytest rc=$?
  tail -15 $OUT/pytest.txt
fi
