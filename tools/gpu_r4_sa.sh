#!/bin/bash
# round 4: duplicate-free set-abstraction kernels -- parity (packed vs dense, PointNet++ path vs the oracle), timing, bench line
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4sa; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_learner.py tests/test_gpu_fuzz.py -m gpu -q -x -k "sa_ or pointnet2 or PointNet2" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -25 $out/pytest.log
rm -f $out/time_sa.txt
for v in 1 0 1; do
  echo "== SA_PACKED=$v" >> $out/time_sa.txt
  SA_PACKED=$v timeout 300 python tools/time_sa.py 2>&1 | grep level >> $out/time_sa.txt
done
cat $out/time_sa.txt
timeout 600 python bench.py --workload vision_pn2 --steps 2 --warmup 1 --no-cpu-baseline > $out/line_vision_pn2.json 2> $out/line_vision_pn2.err
tail -3 $out/line_vision_pn2.err; cat $out/line_vision_pn2.json | head -c 1500
