mkdir -p gpurun_out/r3w
python -m pytest tests/test_mfma_gpu.py tests/test_queue_engine_gpu.py tests/test_cpd_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for cfg in "250000 30 surface 1" "400000 26 surface 1" "100000 30 surface 1"; do
  set -- $cfg
  python tools/mfma_vs_valu.py $1 $2 $3 $4 > gpurun_out/r3w/r3_engine_switch_$3_$1_w$4.log 2>&1
  tail -1 gpurun_out/r3w/r3_engine_switch_$3_$1_w$4.log
done
python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --warmup 3 > gpurun_out/r3w/bench_c1.json 2> gpurun_out/r3w/bench_c1.err
python -c "
import json;d=json.load(open('gpurun_out/r3w/bench_c1.json'));print(d['value'],d.get('dense_it_s'),d.get('late_it_s'),d['roofline']['frac'])"
