#!/bin/bash
# kernel trace of the vision_pn2 line:  tools/gpu_prof_pn2.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/prof_pn2}; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python bench.py --workload vision_pn2 --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_vision_pn2.json 2> $out/bench_vision_pn2.err < /dev/null
python tools/trace_summary.py $out/p/p_kernel_trace.csv $out/bench_vision_pn2_kernel_by_grid.csv 60 < /dev/null
cp $out/p/p_kernel_stats.csv $out/bench_vision_pn2_kernel_stats.csv; rm -rf $out/p
