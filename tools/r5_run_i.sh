#!/bin/bash
# round-5 GPU call I: splat table 512 x 5 against 256 x 8 on the same box (PRG_SPLAT_TABLE=0), parity tests
export TMPDIR=/tmp
out=gpurun_out/r5i
mkdir -p $out; rm -f $out/*.json
timeout 600 python -m pytest tests/test_filterreg_gpu.py tests/test_filterreg_claim_gpu.py tests/test_feature_lattice_gpu.py "tests/test_fullsize_gpu.py::test_filterreg_c4_500k_vs_oracle" -q > $out/pytest_fr.log 2>&1
echo "filterreg tests rc=$?" > $out/status.txt
tail -2 $out/pytest_fr.log
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
for rep in 1 2 3 4; do
  PRG_SPLAT_TABLE=0 timeout 200 $c4 > $out/c4_old_$rep.json 2> /dev/null
  timeout 200 $c4 > $out/c4_new_$rep.json 2> /dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/c4_kt -o b -- $c4 > $out/c4_kt_line.json 2> $out/c4_kt.err
python tools/rocpd_summary.py $(ls $out/c4_kt/*.db $out/c4_kt/*/*.db 2>/dev/null | head -1) > $out/c4_kernel_trace_new.txt 2>&1
rm -rf $out/c4_kt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5i/c4_*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.1f it/s %.4f ms frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
head -6 gpurun_out/r5i/c4_kernel_trace_new.txt
