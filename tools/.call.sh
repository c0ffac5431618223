for m in 4 6; do
  echo "segment model $m"
  PRG_MFMA_SEG_MODEL=$m python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  C1 %.1f it/s dense %.0f late %.0f' % (d['value'], d['dense_it_s'], d['late_it_s']))
for k,v in d['other_workloads'].items(): print('  %s %.1f' % (k, v['value']))"
  PRG_MFMA_SEG_MODEL=$m SKIP_C2=1 TAG=r6m$m bash tools/gpu_session.sh shards > /dev/null 2>&1
  sed -n 6,8p gpurun_out/r6m${m}_shards/shard_window_c1_allranks.log; grep "^sum\|whole iterations (max over ranks), +  0" gpurun_out/r6m${m}_shards/shard_window_c1_allranks.log
done
bash tools/gpu_session.sh pytest tests/test_mfma_gpu.py tests/test_fused_gpu.py tests/test_lean_gpu.py "tests/test_fullsize_gpu.py::test_cpd_bench_config_vs_oracle_dense_and_late"
