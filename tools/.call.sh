bash tools/gpu_session.sh tests
cp gpurun_out/r6_tests/pytest_gpu.log gpurun_out/r6_pytest_gpu_durations_final.log
python __graft_entry__.py smoke
bash tools/gpu_session.sh bench
