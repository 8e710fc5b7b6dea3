#!/bin/bash
# round 5, call D: the whole -m gpu suite (timed), then the PointNet++ bench line + its kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/parity_margins.jsonl
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "amdgpu.ids" gpurun_out/pytest_gpu.log | tail -40
python tools/margins_summary.py gpurun_out/parity_margins.jsonl gpurun_out/parity_margins.json | tail -5
timeout 600 python bench.py --workload vision_pn2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pn2.json 2> gpurun_out/bench_pn2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_pn2.json'))
print("vision_pn2", d["value"], d["ms_per_step"], {k:(round(v["mean_launch_ms"],3), round(v["frac"],3)) for k,v in d["roofline"]["kernels"].items()})
PY
bash tools/prof_cmd.sh r5d_pn2 python bench.py --workload vision_pn2 --steps 2 --warmup 1 --no-cpu-baseline --lean
head -40 gpurun_out/r5d_pn2/kernel_stats.csv | cut -c1-170
