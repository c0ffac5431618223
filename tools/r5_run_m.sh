#!/bin/bash
# round-5 GPU call M: the driver's sequence at the round's final code - full -m gpu suite, smoke, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r5m
timeout 1100 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r5m/pytest_all.log 2>&1
echo "full suite rc=$?" > gpurun_out/r5m/status.txt
tail -3 gpurun_out/r5m/pytest_all.log
timeout 120 python __graft_entry__.py smoke >> gpurun_out/r5m/status.txt 2>&1
timeout 300 python bench.py > gpurun_out/r5m/bench_default.json 2> gpurun_out/r5m/bench_default.err
echo "bench rc=$?" >> gpurun_out/r5m/status.txt
cat gpurun_out/r5m/status.txt
