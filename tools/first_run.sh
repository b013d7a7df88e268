#!/bin/bash
# usage: tools/first_run.sh SPINUP_MS   (GPU box, as the FIRST thing on a fresh box: is the first bench run slower?)
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-rows 0 --spinup-ms $1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i spinup $1: us/step %.2f' % (d['ms_per_step'] * 1e3), {k: round(v, 1) for k, v in d['kernels'].items() if k.endswith('_us')})"
done
