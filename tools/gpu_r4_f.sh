#!/bin/bash
# round 4, call f: 2x2-line conv3d input-layer weight gradient (parity + A/B), streamed-pass unroll of the multi-work-group FPS (A/B)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_learner.py tests/test_gpu_run_loop.py -m gpu -q -k "conv3d or Conv3D or tsdf or dagger" 2>&1 | tail -4
for v in 1 0 1 0; do echo -n "PM_C1_WGRAD=$v  " >> $out/conv3d.txt; PM_C1_WGRAD=$v python tools/time_conv3d.py 2>&1 | tail -1 >> $out/conv3d.txt; done
cat $out/conv3d.txt
for v in main fps_u6 fps_u8 main fps_u6; do
  if [ $v == main ]; then unset PARTMANIP_HIP_LIB; else export PARTMANIP_HIP_LIB=gpurun_ab/$v.so; fi
  python bench.py --workload depth2pc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', round(j['ms_per_step'],3), 'ms/call; fps launch', round(j['roofline']['mean_launch_ms'],3))" | tee -a $out/fps_unroll.txt
done
unset PARTMANIP_HIP_LIB
python bench.py --workload dagger --student conv3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('dagger conv3d', j['value'], j['ms_per_step'], j['roofline'])"
