#!/bin/bash
# round 4, call h: multi-work-group FPS with the streamed pass's first points requested ahead of the register / LDS passes (A/B)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4h; mkdir -p $out
for v in main fps_e1 fps_e2 fps_e2r14 fps_e4r12 main fps_e1 fps_e2; do
  if [ $v == main ]; then unset PARTMANIP_HIP_LIB; else export PARTMANIP_HIP_LIB=gpurun_ab/$v.so; fi
  python bench.py --workload depth2pc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', round(j['ms_per_step'],3), 'ms/call; fps launch', round(j['roofline']['mean_launch_ms'],3))" | tee -a $out/fps_early.txt
done
unset PARTMANIP_HIP_LIB
PARTMANIP_HIP_LIB=gpurun_ab/fps_e1.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_pointops_kat.py -m gpu -q -k "fps or depth" 2>&1 | tail -2
