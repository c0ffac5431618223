#!/bin/bash
# round-5 GPU call G: FilterReg embedding with 16-byte stores of a point's slots / weights - parity tests and C4, three runs
export TMPDIR=/tmp
out=gpurun_out/r5g
mkdir -p $out
timeout 600 python -m pytest tests/test_filterreg_gpu.py tests/test_filterreg_claim_gpu.py tests/test_feature_lattice_gpu.py "tests/test_fullsize_gpu.py::test_filterreg_c4_500k_vs_oracle" -q > $out/pytest_fr.log 2>&1
echo "filterreg tests rc=$?" > $out/status.txt
tail -2 $out/pytest_fr.log
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
for rep in 1 2 3; do timeout 200 $c4 > $out/c4_$rep.json 2> /dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5g/c4_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.1f it/s %.4f ms frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
