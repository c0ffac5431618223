#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r2
# rocprofv3 kernel traces of the default bench line and PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes;
# never combined with system / runtime traces) for C1, C3 and C4; summaries land in gpurun_out/prof_<tag>/*.txt.
tag=${1:-r3}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
sum() { python tools/rocpd_summary.py "$@"; }
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }

# the driver's own command, un-profiled: the line tests/test_bench_line_schema.py checks
python bench.py > $out/${tag}_bench_default_line_1gpu.json 2> $out/default_line.err
rocprofv3 --kernel-trace --stats -d $out/kt_default -o b -- python bench.py --no-cpu-baseline > $out/bench_default_line_profiled.json 2> $out/kt_default.err
sum $(db $out/kt_default) > $out/${tag}_bench_default_kernel_trace.txt

c1="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
# C1 alone: the un-profiled line with its per-iteration (engine, pairs, ms) table, then the kernel trace of the same command -
# roofline.frac can be recomputed from these two committed files
$c1 --pairs-log $out/${tag}_c1_pairs_per_iteration.log > $out/${tag}_bench_c1_line.json 2> $out/c1_line.err
rocprofv3 --kernel-trace --stats -d $out/kt_c1 -o b -- $c1 > $out/c1_kt_line.json 2> $out/kt_c1.err
sum $(db $out/kt_c1) > $out/${tag}_bench_c1_kernel_trace.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace -d $out/c1_$name -o b -- $c1 > $out/c1_$name.log 2>&1
done
sum --pmc $(db $out/c1_FETCH_SIZE) $(db $out/c1_WRITE_SIZE) $(db $out/c1_SQ_INSTS_VALU) > $out/${tag}_bench_rigid100k_pmc.txt

c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats -d $out/c4_kt -o b -- $c4 > $out/c4_line.json 2> $out/c4_kt.err
sum $(db $out/c4_kt) > $out/${tag}_filterreg_500k_kernel_trace.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace -d $out/c4_$name -o b -- $c4 > $out/c4_$name.log 2>&1
done
sum --pmc $(db $out/c4_FETCH_SIZE) $(db $out/c4_WRITE_SIZE) $(db $out/c4_SQ_INSTS_VALU) > $out/${tag}_filterreg_500k_pmc.txt

c3="python bench.py --workload nonrigid_50k --steps 20 --warmup 2 --no-dense-compare"
rocprofv3 --kernel-trace --stats -d $out/c3_kt -o b -- $c3 > $out/c3_line.json 2> $out/c3_kt.err
sum $(db $out/c3_kt) > $out/${tag}_nonrigid_50k_kernel_trace.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace -d $out/c3_$name -o b -- $c3 > $out/c3_$name.log 2>&1
done
sum --pmc $(db $out/c3_FETCH_SIZE) $(db $out/c3_WRITE_SIZE) $(db $out/c3_SQ_INSTS_VALU) > $out/${tag}_nonrigid_50k_pmc.txt
# measured logs, no profiler: both engines and the switch along registrations (surface / volume / 10:1:1, sizes, shards), the
# 8-rank window of E-steps, whole registrations
for cfg in "100000 30 surface 1" "12000 30 surface 1" "30000 30 surface 1" "50000 30 surface 1" "250000 30 surface 1" "400000 26 surface 1" \
           "100000 120 volume 1" "100000 72 aniso 1" "100000 30 surface 8" "100000 30 surface 4" "100000 30 surface 2"; do
  set -- $cfg
  python tools/mfma_vs_valu.py $1 $2 $3 $4 2>&1 | grep -v "amdgpu.ids" > $out/${tag}_engine_switch_$3_$1_w$4.log
done
python tools/shard_window.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/${tag}_shard_window.log
python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" > $out/${tag}_whole_registrations_100k.log
ls -la $out/*.txt
# afterwards, locally: cp gpurun_out/prof_$tag/${tag}_* profiles/ && python tools/pmc_traffic_update.py $tag
# keep the merged output small: the databases stay on the box
rm -rf $out/kt_default $out/kt_c1 $out/c1_FETCH_SIZE $out/c1_WRITE_SIZE $out/c1_SQ_INSTS_VALU $out/c4_kt $out/c4_FETCH_SIZE $out/c4_WRITE_SIZE $out/c4_SQ_INSTS_VALU $out/c3_kt $out/c3_FETCH_SIZE $out/c3_WRITE_SIZE $out/c3_SQ_INSTS_VALU 2>/dev/null
