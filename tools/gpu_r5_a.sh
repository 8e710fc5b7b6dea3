#!/bin/bash
# round 5, call A: whole-update parity tests, the DP / FPS tests touched this round, the SparseUNet race hunt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/parity_margins.jsonl
timeout 1500 python -m pytest tests/test_gpu_wholeupdate.py -m gpu -q -s --durations=5 > gpurun_out/whole.log 2>&1; echo rc=$? >> gpurun_out/whole.log
grep -v amdgpu.ids gpurun_out/whole.log | tail -15
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_learner.py -m gpu -q --durations=5 -k "dp or dagger_resume or bench or capture or varlen" > gpurun_out/dp.log 2>&1; echo rc=$? >> gpurun_out/dp.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "varlen or fps" >> gpurun_out/dp.log 2>&1; echo rc=$? >> gpurun_out/dp.log
grep -v amdgpu.ids gpurun_out/dp.log | tail -25
S=gpurun_out/stress.log; : > $S
timeout 600 python tools/stress_sparse_unet.py --reps 1000 --B 256 --mode both >> $S 2>&1
timeout 600 python tools/stress_sparse_unet.py --reps 1000 --B 256 --mode both --noise >> $S 2>&1
timeout 600 python tools/stress_sparse_unet.py --reps 300 --B 256 --mode both --perlaunch >> $S 2>&1
AMD_SERIALIZE_KERNEL=3 timeout 600 python tools/stress_sparse_unet.py --reps 300 --B 256 --mode both >> $S 2>&1
timeout 900 python tools/stress_sparse_unet.py --reps 200 --B 256 --dagger >> $S 2>&1
grep -v amdgpu.ids $S | tail -40
