#!/bin/bash
# The round's GPU calls, one parameterised script (replaces the one-shot tools/r5_run_*.sh):
#     gpurun --timeout S -- bash tools/gpu_session.sh <step> [args]
# Every step writes under gpurun_out/<tag>/ (merged back by gpurun); what is kept is copied into profiles/ by hand.
export TMPDIR=/tmp
step=${1:-help}; shift
tag=${TAG:-r6}
out=gpurun_out/${tag}_$step
mkdir -p $out
quiet='amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
c1="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads"
line() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        extra = "".join("  %s %.0f" % (k, d[k]) for k in ("dense_it_s", "late_it_s") if k in d)
        print("%-60s %9.1f it/s  %.4f ms%s" % (f, d["value"], d["ms_per_step"], extra))
        for k, v in (d.get("other_workloads") or {}).items():
            if isinstance(v, dict) and "value" in v:
                print("%-60s   %-16s %9.1f it/s" % ("", k, v["value"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
}
case $step in
  tests)      # the whole GPU suite with durations
    python -m pytest tests -m gpu -x -q --durations=25 "$@" 2>&1 | grep -v "$quiet" | tail -60 > $out/pytest_gpu.log; tail -5 $out/pytest_gpu.log ;;
  pytest)     # selected tests:  pytest tests/test_x.py ...
    python -m pytest -x -q "$@" 2>&1 | grep -v "$quiet" | tail -40 | tee $out/pytest.log ;;
  order_ab)   # spatial order of the plan's clouds (PRG_SPATIAL_ORDER) and the shard cut (PROBREG_SHARD_CUT): C1 window + shard replay
    for o in morton kd; do
      PRG_SPATIAL_ORDER=$o $c1 --pairs-log $out/c1_pairs_$o.log > $out/c1_$o.json 2> $out/c1_$o.err
    done
    line $out/c1_morton.json $out/c1_kd.json
    PRG_SPATIAL_ORDER=morton PROBREG_SHARD_CUT=morton python tools/shard_window.py 2>&1 | grep -v "$quiet" > $out/shard_window_c1_allranks_morton.log
    python tools/shard_window.py 2>&1 | grep -v "$quiet" > $out/shard_window_c1_allranks.log
    python tools/shard_window.py 200000 20 affine 2>&1 | grep -v "$quiet" > $out/shard_window_c2_allranks.log
    for f in $out/shard_window_c*.log; do tail -n 8 $f; done ;;
  shards)     # every rank's shard of 1 / 2 / 4 / 8 over the bench window, C1 and C2
    python tools/shard_window.py 2>&1 | grep -v "$quiet" > $out/shard_window_c1_allranks.log
    [ -z "$SKIP_C2" ] && python tools/shard_window.py 200000 20 affine 2>&1 | grep -v "$quiet" > $out/shard_window_c2_allranks.log
    for f in $out/shard_window_c*_allranks.log; do tail -n 8 $f; done ;;
  shard_trace) # per-kernel cost of ONE rank's iteration:  shard_trace "8,3,19" "8,3,3" ...  (world,rank,iteration)
    for spec in "$@"; do
      name=$(echo $spec | tr ',' '_')
      SHARD_TRACE=$spec rocprofv3 --kernel-trace --stats -d $out/kt_$name -o b -- python tools/shard_window.py $SHARD_ARGS > $out/trace_$name.log 2>&1
      python tools/rocpd_summary.py $(ls $out/kt_$name/*.db $out/kt_$name/*/*.db 2>/dev/null | head -1) > $out/shard_trace_$name.txt
      grep "^# rank" $out/trace_$name.log; head -16 $out/shard_trace_$name.txt
    done ;;
  bench)      # the driver's own command
    python bench.py "$@" > $out/bench_default_line_1gpu.json 2> $out/bench.err; line $out/bench_default_line_1gpu.json ;;
  c1)         # C1 alone, with env A/B:  c1 NAME=value ...   (each assignment is one run)
    $c1 --pairs-log $out/c1_pairs_base.log > $out/c1_base.json 2> /dev/null
    for kv in "$@"; do env $kv $c1 --pairs-log $out/c1_pairs_$kv.log > $out/c1_$kv.json 2> /dev/null; done
    line $out/c1_*.json ;;
  run)        # anything else:  run <command...>
    "$@" 2>&1 | grep -v "$quiet" | tee $out/run.log | tail -60 ;;
  *) echo "steps: tests | pytest <files> | order_ab | shards | bench [args] | c1 [ENV=v ...] | run <cmd>" ;;
esac
