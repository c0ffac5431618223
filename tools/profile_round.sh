#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r2
# rocprofv3 kernel traces of the default bench line and PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes;
# never combined with system / runtime traces) for C1, C3 and C4; summaries land in gpurun_out/prof_<tag>/*.txt.
tag=${1:-r6}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
sum() { python tools/rocpd_summary.py "$@"; }
db() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }

# Ordered by importance: a round's GPU budget may end before the script does (set PROFILE_FULL=1 for every pass).
# 1. the driver's own command, un-profiled: the line tests/test_bench_line_schema.py checks
python bench.py > $out/${tag}_bench_default_line_1gpu.json 2> $out/default_line.err

c1="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
# 2. C1 alone: the un-profiled line with its per-iteration (engine, pairs, ms) table, then the kernel trace of the same command -
# roofline.frac can be recomputed from these two committed files
$c1 --pairs-log $out/${tag}_c1_pairs_per_iteration.log > $out/${tag}_bench_c1_line.json 2> $out/c1_line.err
PROBREG_BENCH_TWO_SWEEPS=1 $c1 --pairs-log $out/${tag}_c1_pairs_per_iteration_two_sweeps.log > $out/${tag}_bench_c1_line_two_sweeps.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $out/kt_c1 -o b -- $c1 > $out/c1_kt_line.json 2> $out/kt_c1.err
sum $(db $out/kt_c1) > $out/${tag}_bench_c1_kernel_trace.txt
pmc() {  # pmc <prefix> <command...> : FETCH_SIZE and WRITE_SIZE in separate passes (+ the SQ pass with PROFILE_FULL=1)
  local pre=$1; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --pmc $pass --kernel-trace -d $out/${pre}_$pass -o b -- "$@" > $out/${pre}_$pass.log 2>&1
  done
}
pmc c1 $c1
if [ -n "$PROFILE_FULL" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $out/c1_SQ_INSTS_VALU -o b -- $c1 > $out/c1_SQ.log 2>&1
fi
sum --pmc $(db $out/c1_FETCH_SIZE) $(db $out/c1_WRITE_SIZE) $(db $out/c1_SQ_INSTS_VALU) > $out/${tag}_bench_rigid100k_pmc.txt

# 3. measured logs, no profiler: whole registrations, the 8-rank window of E-steps, both engines and the switch along
# registrations (surface / volume / 10:1:1, sizes, shards)
python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" > $out/${tag}_whole_registrations_100k.log
python tools/shard_window.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/${tag}_shard_window_c1_allranks.log
configs=''   # [r6] the engine-switch A/B (tools/mfma_vs_valu.py) compares the TWO-sweep engines; RUN_SWITCH=1 takes it again
[ -n "$RUN_SWITCH" ] && configs='100000:30:surface:1 30000:30:surface:1 250000:30:surface:1 100000:120:volume:1 100000:72:aniso:1 100000:30:surface:8'   # (round 4: the engine-switch logs were taken on their own, profiles/r4_engine_switch_*)
[ -n "$PROFILE_FULL" ] && configs="$configs 12000:30:surface:1 50000:30:surface:1 400000:26:surface:1 100000:30:surface:4 100000:30:surface:2"
for cfg in $configs; do
  IFS=: read n its kind world <<< "$cfg"
  python tools/mfma_vs_valu.py $n $its $kind $world 2>&1 | grep -v "amdgpu.ids" > $out/${tag}_engine_switch_${kind}_${n}_w${world}.log
done

# 3a. which kernels one rank's iteration is made of (rank 3 of 8 / the one GPU; dense, mid, late), with the owner sweep and with round 5's engines
for spec in 8,3,3 8,3,9 8,3,19 1,0,19; do
  name=$(echo $spec | tr ',' '_')
  SHARD_TRACE=$spec rocprofv3 --kernel-trace --stats -d $out/st_$name -o b -- python tools/shard_window.py > $out/st_$name.log 2>&1
  (grep "^# rank" $out/st_$name.log; sum $(db $out/st_$name) | head -14) > $out/${tag}_shard_trace_$name.txt
  if [ $spec != 8,3,3 ]; then
    PRG_OWNER_SWEEP=0 SHARD_TRACE=$spec rocprofv3 --kernel-trace --stats -d $out/sq_$name -o b -- python tools/shard_window.py > $out/sq_$name.log 2>&1
    (grep "^# rank" $out/sq_$name.log; sum $(db $out/sq_$name) | head -14) > $out/${tag}_shard_trace_${name}_round5_engines.txt
  fi
  rm -rf $out/st_$name $out/sq_$name
done

# 3b. C2 (the 8-GPU configuration): HBM traffic of its two sweeps, and the shard replay the 6x projection rests on
c2="python bench.py --workload affine_200k --steps 20 --warmup 1 --no-cpu-baseline"
pmc c2 $c2
sum --pmc $(db $out/c2_FETCH_SIZE) $(db $out/c2_WRITE_SIZE) > $out/${tag}_affine_200k_pmc.txt
[ -z "$SKIP_SHARD_C2" ] && python tools/shard_window.py 200000 20 affine 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $out/${tag}_shard_window_c2_allranks.log

# 4. C3 and C4: kernel traces and HBM traffic
c3="python bench.py --workload nonrigid_50k --steps 20 --warmup 2 --no-dense-compare"
rocprofv3 --kernel-trace --stats -d $out/c3_kt -o b -- $c3 > $out/c3_line.json 2> $out/c3_kt.err
sum $(db $out/c3_kt) > $out/${tag}_nonrigid_50k_kernel_trace.txt
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats -d $out/c4_kt -o b -- $c4 > $out/c4_line.json 2> $out/c4_kt.err
sum $(db $out/c4_kt) > $out/${tag}_filterreg_500k_kernel_trace.txt
pmc c3 $c3
sum --pmc $(db $out/c3_FETCH_SIZE) $(db $out/c3_WRITE_SIZE) > $out/${tag}_nonrigid_50k_pmc.txt
pmc c4 $c4
sum --pmc $(db $out/c4_FETCH_SIZE) $(db $out/c4_WRITE_SIZE) > $out/${tag}_filterreg_500k_pmc.txt

# 5. the whole default command under the kernel trace (C1, C2, C3, C4 mixed)
if [ -n "$PROFILE_FULL" ]; then
  rocprofv3 --kernel-trace --stats -d $out/kt_default -o b -- python bench.py --no-cpu-baseline > $out/bench_default_line_profiled.json 2> $out/kt_default.err
  sum $(db $out/kt_default) > $out/${tag}_bench_default_kernel_trace.txt
fi
ls -la $out/*.txt
# afterwards, locally: cp gpurun_out/prof_$tag/${tag}_* profiles/ && python tools/pmc_traffic_update.py $tag
# keep the merged output small: the databases stay on the box
rm -rf $out/kt_default $out/kt_c1 $out/c2_FETCH_SIZE $out/c2_WRITE_SIZE $out/c1_FETCH_SIZE $out/c1_WRITE_SIZE $out/c1_SQ_INSTS_VALU $out/c4_kt $out/c4_FETCH_SIZE $out/c4_WRITE_SIZE $out/c3_kt $out/c3_FETCH_SIZE $out/c3_WRITE_SIZE 2>/dev/null
