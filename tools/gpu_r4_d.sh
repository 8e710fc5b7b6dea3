#!/bin/bash
# round 4, call d: plan-side staging tables + early loads in the packed SA kernels, aligned glue GEMMs; PMC passes for the traffic fields
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4d; mkdir -p $out
rm -f gpurun_out/parity_margins.jsonl
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_learner.py tests/test_gpu_fuzz.py -m gpu -q --durations=5 \
  -k "one_call or sa_ or pointnet2 or PointNet2 or first_graph_chunk" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -v amdgpu.ids $out/pytest.log | tail -15
python tools/margins_summary.py gpurun_out/parity_margins.jsonl $out/parity_margins.json > $out/margins.txt; grep -i "first 16\|first chunk" $out/margins.txt
for v in 1 0; do echo "== SA_PACKED=$v" >> $out/time_sa.txt; SA_PACKED=$v timeout 300 python tools/time_sa.py 2>&1 | grep level >> $out/time_sa.txt; done
cat $out/time_sa.txt
timeout 600 python bench.py --workload vision_pn2 --steps 3 --warmup 1 > $out/line_vision_pn2.json 2> $out/line_vision_pn2.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4d/line_vision_pn2.json"))
print("pn2", j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], j.get("cpu_baseline",{}).get("value"), j.get("cpu_baseline",{}).get("sample"))
print({k:(round(v["mean_launch_ms"],3), round(v["frac"],3)) for k,v in j["roofline"]["kernels"].items()}, j["roofline"]["levels"])
PY
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_sa python tools/time_sa.py < /dev/null
PMC_PASS_TIMEOUT=300 bash tools/pmc_run.sh $out/pmc_su2048 python tools/time_sparse_unet.py 2048 < /dev/null
PMC_PASS_TIMEOUT=400 bash tools/pmc_run.sh $out/pmc_state env PARTMANIP_GRAPHS=0 python bench.py --workload state --lean --no-cpu-baseline --steps 2 --warmup 0 < /dev/null
cat $out/pmc_sa/summary.txt | head; du -sh $out
