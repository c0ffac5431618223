PRG_OWNER_CPL=1 bash tools/gpu_session.sh pytest tests/test_resid_gpu.py tests/test_edge_gpu.py "tests/test_fullsize_gpu.py::test_cpd_bench_config_vs_oracle_dense_and_late[C1_rigid_100k]" tests/test_world8_gpu.py::test_eight_ranks_on_one_gpu_match_the_unsharded_oracle
bash tools/gpu_session.sh c1 PRG_OWNER_CPL=1 "PRG_OWNER_CPL=1 PRG_OWNER_PLANES=1" "PRG_OWNER_CPL=1 PRG_OWNER_PLANES=3"
for spec in 1,0,12 1,0,19 8,3,7 8,3,9 8,3,19; do for c in 2 1; do echo "spec $spec cpl $c: $(PRG_OWNER_CPL=$c SHARD_TRACE=$spec,100 python tools/shard_window.py 2>&1 | grep '^# rank')"; done; done
