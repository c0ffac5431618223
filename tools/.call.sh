python tools/fuzz_parity.py 160 11 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6_fuzz_parity.log | tail -6
python tools/fuzz_nonrigid.py 40 5 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6_fuzz_nonrigid.log | tail -4
