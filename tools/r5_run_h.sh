#!/bin/bash
# round-5 GPU call H: same-box A/B of two library builds on C4 (PROBREG_HIP_LIB=tools/bin/libprobreg_hip_v0.so is the baseline), alternating
export TMPDIR=/tmp
out=gpurun_out/r5h
mkdir -p $out; rm -f $out/*.json
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
for rep in 1 2 3 4; do
  PROBREG_HIP_LIB=$PWD/tools/bin/libprobreg_hip_v0.so timeout 200 $c4 > $out/c4_v0_$rep.json 2> /dev/null
  timeout 200 $c4 > $out/c4_new_$rep.json 2> /dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/c4_kt -o b -- $c4 > $out/c4_kt_line.json 2> $out/c4_kt.err
python tools/rocpd_summary.py $(ls $out/c4_kt/*.db $out/c4_kt/*/*.db 2>/dev/null | head -1) > $out/c4_kernel_trace_new.txt 2>&1
rm -rf $out/c4_kt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5h/c4_*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.1f it/s %.4f ms frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
head -12 gpurun_out/r5h/c4_kernel_trace_new.txt
