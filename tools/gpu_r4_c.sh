#!/bin/bash
# round 4, call c: tests added after call b, PointNet++ kernel trace, PMC passes for the new traffic fields
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4c; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_sparse_unet.py tests/test_pointops_kat.py tests/test_gpu_learner.py -m gpu -q --durations=8 \
  -k "one_call or degrades or hand_off or first_graph_chunk or full_size or known_answers or other_activations or nan or chained" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -v amdgpu.ids $out/pytest.log | tail -30
python tools/margins_summary.py gpurun_out/parity_margins.jsonl $out/parity_margins.json | grep -i "first 16\|first chunk" 
prof() {   # name, bench args...
  name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o p -- python bench.py "$@" > $out/$name.json 2> $out/$name.err < /dev/null
  python tools/trace_summary.py $out/$name/p_kernel_trace.csv $out/${name}_kernel_by_grid.csv 60 < /dev/null
  cp $out/$name/p_kernel_stats.csv $out/${name}_kernel_stats.csv
  rm -rf $out/$name
}
prof bench_vision_pn2 --workload vision_pn2 --steps 1 --warmup 1 --no-cpu-baseline
head -40 $out/bench_vision_pn2_kernel_stats.csv
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_sa python tools/time_sa.py < /dev/null
PMC_PASS_TIMEOUT=300 bash tools/pmc_run.sh $out/pmc_su2048 python tools/time_sparse_unet.py 2048 < /dev/null
PMC_PASS_TIMEOUT=400 bash tools/pmc_run.sh $out/pmc_state env PARTMANIP_GRAPHS=0 python bench.py --workload state --lean --no-cpu-baseline --steps 2 --warmup 0 < /dev/null
ls $out $out/pmc_sa | head -40; tail -3 $out/pmc_state/pmc_0.log
