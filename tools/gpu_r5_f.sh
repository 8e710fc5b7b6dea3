#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_learner.py tests/test_gpu_bookkeeping.py -m gpu -q -x > gpurun_out/f_tests.log 2>&1; echo rc=$? >> gpurun_out/f_tests.log; grep -v amdgpu.ids gpurun_out/f_tests.log | tail -6
timeout 300 python tools/time_sa.py 2>&1 | grep -v amdgpu.ids | tail -12
for i in 1 2; do timeout 600 python bench.py --workload vision_pn2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pn2.json 2> gpurun_out/bench_pn2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_pn2.json'))
print("vision_pn2", d["value"], d["ms_per_step"], {k:(round(v["mean_launch_ms"],3), round(v["frac"],3)) for k,v in d["roofline"]["kernels"].items()})
PY
done
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh gpurun_out/pmc_sa_f python tools/time_sa.py < /dev/null; grep -i "sa_bwd\|sa_fwd" gpurun_out/pmc_sa_f/summary.txt | head -12
