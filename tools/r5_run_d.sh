#!/bin/bash
# round-5 GPU call D: the whole -m gpu suite as the driver runs it (with durations), then the slack audit of the loosened bounds
export TMPDIR=/tmp
mkdir -p gpurun_out/r5d
timeout 1300 python -m pytest tests -q -m gpu --durations=40 > gpurun_out/r5d/pytest_all.log 2>&1
echo "full suite rc=$?" > gpurun_out/r5d/status.txt
tail -3 gpurun_out/r5d/pytest_all.log
timeout 200 python tools/slack_audit.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r5d/slack_audit.log
timeout 120 python __graft_entry__.py smoke >> gpurun_out/r5d/status.txt 2>&1
