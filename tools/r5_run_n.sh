#!/bin/bash
# round-5 GPU call N (what was left of the budget): FilterReg embedding with the first attempt of all rounds interleaved, creating stage only
export TMPDIR=/tmp
out=gpurun_out/r5n
mkdir -p $out; rm -f $out/*.json
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
for rep in 1 2; do for v in 0 1; do
  PRG_EMBED_INTERLEAVE=$v timeout 60 $c4 > $out/c4_il${v}_$rep.json 2> /dev/null
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5n/c4_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.1f it/s %.4f ms" % (d["value"], d["ms_per_step"]))
PY
PRG_EMBED_INTERLEAVE=3 timeout 60 python -m pytest tests/test_filterreg_gpu.py "tests/test_fullsize_gpu.py::test_filterreg_c4_500k_vs_oracle" -q 2>&1 | tail -1
