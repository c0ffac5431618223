bash tools/gpu_session.sh shards
