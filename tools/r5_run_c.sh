#!/bin/bash
# round-5 GPU call C: the round's committed profiles (tools/profile_round.sh r5), the widened fuzz, and the tests changed since call B
export TMPDIR=/tmp
mkdir -p gpurun_out/r5c
timeout 400 python -m pytest "tests/test_fullsize_gpu.py::test_cpd_bench_config_vs_oracle_dense_and_late[C1_rigid_100k]" tests/test_resid_gpu.py tests/test_filterreg_claim_gpu.py -q --durations=12 > gpurun_out/r5c/pytest_changed.log 2>&1
echo "changed tests rc=$?" > gpurun_out/r5c/status.txt
tail -2 gpurun_out/r5c/pytest_changed.log
timeout 1500 bash tools/profile_round.sh r5 > gpurun_out/r5c/profile_round.log 2>&1
echo "profile_round rc=$?" >> gpurun_out/r5c/status.txt
timeout 900 python tools/fuzz_fused.py 56 5 2 48000 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r5c/fuzz_fused.log
echo "fuzz rc=$?" >> gpurun_out/r5c/status.txt
tail -1 gpurun_out/r5c/fuzz_fused.log
