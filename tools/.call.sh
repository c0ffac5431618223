bash tools/profile_round.sh r6 > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
