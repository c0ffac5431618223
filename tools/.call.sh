bash tools/gpu_session.sh pytest tests/test_spatial_order.py tests/test_cpd_gpu.py tests/test_resid_gpu.py -q
python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6_whole_registrations_100k.log
PRG_SPATIAL_ORDER=kd_host python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" | head -2
PRG_SPATIAL_ORDER=morton python tools/time_registration.py 2>&1 | grep -v "amdgpu.ids" | head -2
python - <<'PY'
import time, numpy as np
from probreg_amd import engine, synthetic
for n in (100000, 500000):
    p = synthetic.surface(n, seed=1).astype(np.float32)
    engine.spatial_order(p[:1000])
    for dev in (True, False):
        t0 = time.perf_counter(); engine.spatial_order(p, on_device=dev); dt = time.perf_counter() - t0
        print("spatial order of %d points, %s: %.2f ms (incl. upload / download)" % (n, "device" if dev else "host", dt * 1e3))
PY
for lib in sub1 sub4; do PROBREG_HIP_LIB=tools/bin/libprobreg_hip_$lib.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('owner $lib (old host order build): %.1f it/s late %.0f' % (d['value'], d['late_it_s']))"; done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('current: %.1f it/s late %.0f' % (d['value'], d['late_it_s']))"
