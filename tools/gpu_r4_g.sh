#!/bin/bash
# round 4, call g: two-stage LDS ring for the 128 x 128 GEMM tile (two work-groups per CU): parity + A/B
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4g; mkdir -p $out
PM_G2_NBUF=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_learner.py -m gpu -q -k "linear or gemm or pointnet2 or grouped" 2>&1 | tail -3
for nb in 3 2 3 2; do
  echo "PM_G2_NBUF=$nb" >> $out/gemm.txt
  PM_G2_NBUF=$nb python tools/time_gemm.py 131072x256x512 131072x288x256 524288x128x128 131072x512x256 2048x512x512 2>&1 | grep "^M=" >> $out/gemm.txt
done
cat $out/gemm.txt
for nb in 3 2 3 2; do
  PM_G2_NBUF=$nb python bench.py --workload vision_pn2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('pn2 NBUF=$nb', round(j['value']), round(j['ms_per_step'],1))"
done
