#!/bin/bash
# round 5, call E: gathered GEMM with its neighbour indices through LDS-DMA slots -- tests, race hunt, bench lines (sparse, conv3d, state floor)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sparse_unet.py tests/test_gpu_kernels.py tests/test_gpu_learner.py tests/test_gpu_fuzz.py tests/test_gpu_wholeupdate.py -m gpu -q -x --durations=5 > gpurun_out/su.log 2>&1; echo rc=$? >> gpurun_out/su.log
grep -v amdgpu.ids gpurun_out/su.log | tail -12
S=gpurun_out/stress_e.log; : > $S
timeout 400 python tools/stress_sparse_unet.py --reps 1000 --B 256 --mode fused --noise >> $S 2>&1
timeout 300 python tools/stress_sparse_unet.py --reps 1000 --B 256 --mode fused >> $S 2>&1
timeout 300 python tools/stress_sparse_unet.py --reps 200 --B 256 --dagger >> $S 2>&1
grep -v "amdgpu.ids\|load teacher\|update loss\|save ckpt" $S | tail -12; grep "update loss" $S | sort | uniq -c
for i in 1 2; do timeout 600 python bench.py --workload dagger --student sparse_unet --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_su.json 2> gpurun_out/bench_su.err; python -c "
import json; d=json.load(open('gpurun_out/bench_su.json')); print('sparse_unet', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['fwd_mean_ms'], d['roofline']['bwd_mean_ms'])"; done
timeout 600 python bench.py --workload dagger --student conv3d --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c3d.json 2> gpurun_out/bench_c3d.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c3d.json')); print('conv3d', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --workload state --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_state.json 2> gpurun_out/bench_state.err; python -c "
import json; d=json.load(open('gpurun_out/bench_state.json')); print('state', d['value'], d['ms_per_step'], json.dumps({k:v for k,v in d['roofline'].get('floor_us_per_step',{}).items() if k!='note'}))"; tail -3 gpurun_out/bench_state.err
