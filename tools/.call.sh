bash tools/gpu_session.sh tests
cp gpurun_out/r6_tests/pytest_gpu.log gpurun_out/r6_pytest_gpu_durations.log
