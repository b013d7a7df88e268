#!/bin/bash
mkdir -p gpurun_out/r3soak
python3 tools/soak_seeds.py 382:0 453:0 520:0 542:1 111:0 116:1 142:1 152:1 159:1 164:0 183:1 196:1 259:0 > gpurun_out/r3soak/recheck3.txt 2>gpurun_out/r3soak/recheck3.err
cat gpurun_out/r3soak/recheck3.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3soak/pytest3.txt 2>&1; echo pytest rc=$?; tail -2 gpurun_out/r3soak/pytest3.txt
