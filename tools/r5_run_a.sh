#!/bin/bash
# round-5 GPU call A: new single-sweep tests, C1 line with per-iteration table (three lower bounds of the fused regime), shard replays, full suite with durations
out=gpurun_out/r5a
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_resid_gpu.py tests/test_fused_gpu.py -q --durations=30 -x > $out/pytest_new.log 2>&1
echo "new tests rc=$?" > $out/status.txt
c1="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
for sc in 0.5 0.75 1.0 1.4; do
  PRG_FUSED_RCOL_SCALE=$sc timeout 300 $c1 --pairs-log $out/c1_pairs_scale$sc.log > $out/c1_line_scale$sc.json 2> $out/c1_line_scale$sc.err
done
PRG_RESID_SWEEP=0 timeout 300 $c1 --pairs-log $out/c1_pairs_resid_off.log > $out/c1_line_resid_off.json 2> /dev/null
timeout 300 python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" > $out/whole_registrations_100k.log
timeout 600 python tools/shard_window.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/shard_window_c1.log
timeout 900 python tools/shard_window.py 200000 20 affine 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/shard_window_c2.log
echo "measurements done" >> $out/status.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=80 > $out/pytest_all.log 2>&1
echo "full suite rc=$?" >> $out/status.txt
tail -5 $out/pytest_all.log
