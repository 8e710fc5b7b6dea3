#!/bin/bash
# A/B of the multi-work-group FPS shapes (PM_FM_CFG: 0 = 1024 x 16 + 8 192, 1 = 512 x 50 + 10 176): FPS tests + known answers, then the depth2pc line
mkdir -p gpurun_out/r4fps
for c in ${FM_CFGS:-0 1}; do
  PM_FM_CFG=$c timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_pointops_kat.py -m gpu -q -k "fps or depth2pc or varlen" > gpurun_out/r4fps/tests_$c.log 2>&1
  tail -2 gpurun_out/r4fps/tests_$c.log
  PM_FM_CFG=$c timeout 600 python bench.py --workload depth2pc --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r4fps/bench_$c.json 2> gpurun_out/r4fps/bench_$c.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r4fps/bench_$c.json").read().strip().splitlines()[-1])
    print("cfg $c", d["ms_per_step"], d.get("roofline", {}).get("latency"))
except Exception as e:
    print("cfg $c failed", e)
PY
done
