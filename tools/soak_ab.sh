#!/bin/bash
mkdir -p gpurun_out/r3soak
python3 tools/soak_seeds.py 259:0 159:1 111:0 > gpurun_out/r3soak/recheck2.txt 2>gpurun_out/r3soak/recheck2.err
cat gpurun_out/r3soak/recheck2.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3soak/pytest.txt 2>&1; echo pytest rc=$?; tail -2 gpurun_out/r3soak/pytest.txt
