mkdir -p gpurun_out/r6_sw
for cfg in 100000:30:surface:1 200000:30:surface:1 30000:30:surface:1 100000:72:aniso:1 100000:100:volume:1 100000:30:surface:8; do
  IFS=: read n its kind world <<< "$cfg"
  python tools/mfma_vs_valu.py $n $its $kind $world 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_sw/engine_switch_${kind}_${n}_w${world}.log
  echo "$cfg $(tail -1 gpurun_out/r6_sw/engine_switch_${kind}_${n}_w${world}.log)"
done
