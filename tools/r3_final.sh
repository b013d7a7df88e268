#!/bin/bash
# round-3 closing run (GPU box): rocprofv3 passes of the default bench, then the bench lines kept under profiles/
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_round.sh r03 > gpurun_out/prof_r03_log.txt 2>&1
python3 tools/make_counters_json.py gpurun_out/prof_r03 profiles/r03_counters.json > /dev/null
mkdir -p gpurun_out/r3fin
cp profiles/r03_counters.json gpurun_out/r3fin/
timeout -k 10 300 python3 bench.py > gpurun_out/r3fin/bench_default.json 2>gpurun_out/r3fin/err1.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-rows 0 > gpurun_out/r3fin/bench_steps20.json 2>gpurun_out/r3fin/err2.txt
for w in c2 c3 c5; do timeout -k 10 300 python3 bench.py --workload $w --cpu-rows 0 > gpurun_out/r3fin/bench_$w.json 2>>gpurun_out/r3fin/err3.txt; done
timeout -k 10 300 python3 bench.py --aa 4 --cpu-rows 0 > gpurun_out/r3fin/bench_aa4.json 2>>gpurun_out/r3fin/err3.txt
timeout -k 10 300 python3 bench.py --force-dist --cpu-rows 0 > gpurun_out/r3fin/bench_forcedist.json 2>>gpurun_out/r3fin/err3.txt
timeout -k 10 300 python3 tools/band_cost_model.py > gpurun_out/r3fin/band_cost_model.txt 2>>gpurun_out/r3fin/err3.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3fin/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], "us/step %.2f" % (d["ms_per_step"]*1e3), "value %.0f" % d["value"], {k: round(v, 1) for k, v in (d["kernels"] or {}).items() if k.endswith("_us")})
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r3fin/band_cost_model.txt | cut -c1-300
