#!/bin/bash
# round 4, call e: packed-vs-dense test after the plan-table change, PointNet++ with actor || critic on two streams (A/B), trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4e; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "sa_packed" 2>&1 | tail -3
for ov in 0 1 0 1; do
  PARTMANIP_OVERLAP=$ov timeout 300 python bench.py --workload vision_pn2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('overlap=$ov', round(j['value']), round(j['ms_per_step'],1))"
done
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o p -- python bench.py --workload vision_pn2 --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_vision_pn2.json 2> $out/bench_vision_pn2.err < /dev/null
python tools/trace_summary.py $out/t/p_kernel_trace.csv $out/bench_vision_pn2_kernel_by_grid.csv 60 < /dev/null > /dev/null
cp $out/t/p_kernel_stats.csv $out/bench_vision_pn2_kernel_stats.csv; rm -rf $out/t
head -32 $out/bench_vision_pn2_kernel_stats.csv | cut -c1-150
