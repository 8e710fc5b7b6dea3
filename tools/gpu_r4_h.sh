#!/bin/bash
# round-4 (h): the reshaped multi-work-group FPS -- point-operator tests, depth2pc line, kernel trace, PMC pass (HBM traffic)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_pointops_kat.py tests/test_gpu_fuzz.py -m gpu -q -x > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 300 python bench.py --workload depth2pc > $out/line_depth2pc.json 2> $out/line_depth2pc.err
PM_FM_CFG=0 timeout 300 python bench.py --workload depth2pc --no-cpu-baseline > $out/line_depth2pc_cfg0.json 2>> $out/line_depth2pc.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python bench.py --workload depth2pc --no-cpu-baseline > $out/bench_depth2pc.json 2> $out/bench_depth2pc.err < /dev/null
python tools/trace_summary.py $out/prof/p_kernel_trace.csv $out/bench_depth2pc_kernel_by_grid.csv 40 < /dev/null
cp $out/prof/p_kernel_stats.csv $out/bench_depth2pc_kernel_stats.csv; rm -rf $out/prof
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_fps python bench.py --workload depth2pc --no-cpu-baseline < /dev/null
python tools/make_hbm_traffic.py r4h fps=$out/pmc_fps/summary.json > $out/hbm_traffic.txt 2>&1
cp profiles/hbm_traffic.json $out/hbm_traffic.json
python - <<PY
import json
for f in ("line_depth2pc", "line_depth2pc_cfg0"):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 3), round(d["roofline"]["latency"]["us_per_round"], 3), d["roofline"]["traffic"])
PY
