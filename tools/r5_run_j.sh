#!/bin/bash
# round-5 GPU call J: after the FilterReg memory-access changes - the driver's line again, C4's kernel trace and PMC passes
export TMPDIR=/tmp
out=gpurun_out/prof_r5
mkdir -p $out
sum() { python tools/rocpd_summary.py "$@"; }
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
python bench.py > $out/r5_bench_default_line_1gpu.json 2> $out/default_line.err
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats -d $out/c4_kt -o b -- $c4 > $out/c4_line.json 2> $out/c4_kt.err
sum $(db $out/c4_kt) > $out/r5_filterreg_500k_kernel_trace.txt
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $pass --kernel-trace -d $out/c4_$pass -o b -- $c4 > $out/c4_$pass.log 2>&1
done
sum --pmc $(db $out/c4_FETCH_SIZE) $(db $out/c4_WRITE_SIZE) > $out/r5_filterreg_500k_pmc.txt
rm -rf $out/c4_kt $out/c4_FETCH_SIZE $out/c4_WRITE_SIZE
timeout 300 python -m pytest tests/test_filterreg_gpu.py tests/test_filterreg_claim_gpu.py tests/test_feature_lattice_gpu.py tests/test_mstep_arrays_gpu.py tests/test_gauss_gpu.py "tests/test_fullsize_gpu.py::test_filterreg_c4_500k_vs_oracle" -q 2>&1 | tail -2
python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" > $out/r5_whole_registrations_100k.log
head -8 $out/r5_filterreg_500k_kernel_trace.txt
