#!/bin/bash
# round-4 (k): compact decoder backward of the SparseUNet -- tests, line, kernel trace, PMC traffic at 2048 and 256 clouds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4k; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_sparse_unet.py -m gpu -q -x > $out/tests.log 2>&1; tail -2 $out/tests.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python bench.py --workload dagger --student sparse_unet --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_dagger_sparse_unet.json 2> $out/bench.err < /dev/null
python tools/trace_summary.py $out/p/p_kernel_trace.csv $out/bench_dagger_sparse_unet_kernel_by_grid.csv 50 < /dev/null
cp $out/p/p_kernel_stats.csv $out/bench_dagger_sparse_unet_kernel_stats.csv; rm -rf $out/p
PMC_PASS_TIMEOUT=300 bash tools/pmc_run.sh $out/pmc_su2048 python tools/time_sparse_unet.py 2048 < /dev/null
PMC_PASS_TIMEOUT=200 bash tools/pmc_run.sh $out/pmc_su python tools/time_sparse_unet.py 256 < /dev/null
python tools/make_hbm_traffic.py r4k su2048=$out/pmc_su2048/summary.json > $out/hbm_traffic.txt 2>&1
cp profiles/hbm_traffic.json $out/hbm_traffic.json
timeout 600 python bench.py --workload dagger --student sparse_unet > $out/line_dagger_sparse_unet.json 2>> $out/bench.err
python - <<PY
import json
d = json.loads(open("$out/line_dagger_sparse_unet.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["fwd_mean_ms"], r["bwd_mean_ms"], r["traffic"], r["algorithmic_bytes"], d.get("cpu_baseline", {}).get("value"))
PY
