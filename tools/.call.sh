mkdir -p gpurun_out/r6_ab
for cfg in aniso:100000:1:0 clusters:100000:1:0 volume:100000:1:0 surface:100000:1:0; do
  IFS=: read shape n world rank <<< "$cfg"
  its=24; [ $shape = volume ] && its=60; [ $shape = aniso ] && its=48
  python tools/single_sweep_ab.py $n $its $shape $world $rank 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_ab/single_sweep_ab_${shape}_${n}_w${world}r${rank}.log
  tail -1 gpurun_out/r6_ab/single_sweep_ab_${shape}_${n}_w${world}r${rank}.log | sed "s/^/$cfg /"
done
bash tools/gpu_session.sh pytest tests/test_fused_gpu.py tests/test_resid_gpu.py tests/test_mfma_gpu.py tests/test_lean_gpu.py tests/test_cpd_gpu.py
