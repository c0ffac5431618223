#!/bin/bash
# round-5 GPU call B: changed tests, C1 line + kernel trace at the final lower bound, shard replays with the fine grid
out=gpurun_out/r5b
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_resid_gpu.py tests/test_fused_gpu.py tests/test_lean_gpu.py tests/test_mfma_gpu.py tests/test_queue_engine_gpu.py tests/test_edge_gpu.py tests/test_dist_gpu.py "tests/test_fullsize_gpu.py::test_cpd_bench_config_vs_oracle_dense_and_late" tests/test_cpd_gpu.py -q --durations=25 > $out/pytest_changed.log 2>&1
echo "changed tests rc=$?" > $out/status.txt
tail -3 $out/pytest_changed.log
c1="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
timeout 300 $c1 --pairs-log $out/c1_pairs.log > $out/c1_line.json 2> $out/c1_line.err
PRG_FUSED_RCOL_SCALE=2.0 timeout 300 $c1 --pairs-log $out/c1_pairs_scale2.0.log > $out/c1_line_scale2.0.json 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt_c1 -o b -- $c1 > $out/c1_kt_line.json 2> $out/kt_c1.err
python tools/rocpd_summary.py $(ls $out/kt_c1/*.db $out/kt_c1/*/*.db 2>/dev/null | head -1) > $out/c1_kernel_trace.txt 2>&1
rm -rf $out/kt_c1
timeout 600 python tools/shard_window.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/shard_window_c1.log
PRG_MFMA_FINE_GRID=0 timeout 600 python tools/shard_window.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/shard_window_c1_default_grid.log
timeout 900 python tools/shard_window.py 200000 20 affine 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/shard_window_c2.log
echo "measurements done" >> $out/status.txt
