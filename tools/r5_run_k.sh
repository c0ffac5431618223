#!/bin/bash
# round-5 GPU call K: FilterReg - which fraction of the points should create the vertices (stage 1 of a lattice build)? same box, alternating
export TMPDIR=/tmp
out=gpurun_out/r5k
mkdir -p $out; rm -f $out/*.json
c4="python bench.py --workload filterreg_500k --steps 20 --warmup 3"
for rep in 1 2 3; do for p in 16 8 4 32; do
  PRG_EMBED_PERIOD=$p timeout 200 $c4 > $out/c4_p${p}_$rep.json 2> /dev/null
done; done
PRG_EMBED_PERIOD=8 timeout 300 python -m pytest tests/test_filterreg_gpu.py tests/test_filterreg_claim_gpu.py "tests/test_fullsize_gpu.py::test_filterreg_c4_500k_vs_oracle" -q 2>&1 | tail -2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5k/c4_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.1f it/s %.4f ms" % (d["value"], d["ms_per_step"]))
PY
