#!/bin/bash
# round-4 (i): fused group-all level -- kernel tests, the level's timing, PointNet++ tests, the vision_pn2 line
out=gpurun_out/r4i; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "groupall" > $out/tests_ga.log 2>&1; tail -5 $out/tests_ga.log
timeout 300 python tools/time_groupall.py > $out/time_groupall.txt 2>&1; cat $out/time_groupall.txt | tail -3
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_fuzz.py -m gpu -q -x -k "pointnet2" > $out/tests_pn2.log 2>&1; tail -3 $out/tests_pn2.log
timeout 600 python bench.py --workload vision_pn2 --no-cpu-baseline > $out/line_vision_pn2.json 2> $out/line_vision_pn2.err
python - <<PY
import json
d = json.loads(open("$out/line_vision_pn2.json").read().strip().splitlines()[-1])
print("vision_pn2", round(d["value"], 1), round(d["ms_per_step"], 2))
PY
