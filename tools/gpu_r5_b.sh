#!/bin/bash
# round 5, call B: the race fix (stress reruns), the rewritten compact decoder backward (tests + bench line), ATen call sites
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
S=gpurun_out/stress_b.log; : > $S
timeout 400 python tools/stress_sparse_unet.py --reps 1000 --B 256 --mode fused --noise >> $S 2>&1
timeout 300 python tools/stress_sparse_unet.py --reps 1000 --B 256 --mode fused >> $S 2>&1
timeout 300 python tools/stress_sparse_unet.py --reps 200 --B 256 --dagger >> $S 2>&1
grep -v "amdgpu.ids\|load teacher\|update loss\|save ckpt" $S | tail -25; grep "update loss" $S | sort | uniq -c
timeout 900 python -m pytest tests/test_gpu_sparse_unet.py tests/test_gpu_kernels.py tests/test_gpu_learner.py -m gpu -q -x --durations=5 > gpurun_out/su.log 2>&1; echo rc=$? >> gpurun_out/su.log
grep -v amdgpu.ids gpurun_out/su.log | tail -12
timeout 600 python tools/aten_sites.py vision_pn2 > gpurun_out/aten_pn2.log 2>&1; grep -v amdgpu.ids gpurun_out/aten_pn2.log | tail -45
timeout 600 python tools/aten_sites.py sparse_unet > gpurun_out/aten_su.log 2>&1; grep -v amdgpu.ids gpurun_out/aten_su.log | tail -45
timeout 600 python bench.py --workload dagger --student sparse_unet --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_su.json 2> gpurun_out/bench_su.err; tail -c 1200 gpurun_out/bench_su.json
timeout 600 python bench.py --workload vision_pn2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pn2.json 2> gpurun_out/bench_pn2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_pn2.json'))
print("vision_pn2", d["value"], d["ms_per_step"], {k:(round(v["mean_launch_ms"],3), round(v["frac"],3)) for k,v in d["roofline"]["kernels"].items()})
PY
