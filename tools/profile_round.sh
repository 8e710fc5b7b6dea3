#!/bin/bash
# Round profile set (run ON the GPU box from the repo root, inside ONE gpurun call):  tools/profile_round.sh <tag>
# rocprofv3 --kernel-trace --stats summaries of the bench workloads (+ per-grid tables) and the PMC passes of the encoder
# kernels; everything lands under gpurun_out/<tag>/ -- copy what should be judged into profiles/.
tag=${1:-round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
prof() {   # name, bench args...
  name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o p -- python bench.py "$@" > $out/$name.json 2> $out/$name.err < /dev/null
  python tools/trace_summary.py $out/$name/p_kernel_trace.csv $out/${name}_kernel_by_grid.csv 40 < /dev/null
  cp $out/$name/p_kernel_stats.csv $out/${name}_kernel_stats.csv
  rm -rf $out/$name
}
prof bench_default --no-optional --no-secondary --steps 2 --warmup 1
prof bench_state --workload state --steps 2 --warmup 1 --no-cpu-baseline
prof bench_dagger_sparse_unet --workload dagger --student sparse_unet --steps 1 --warmup 1 --no-cpu-baseline
prof bench_vision_pn2 --workload vision_pn2 --steps 1 --warmup 1 --no-cpu-baseline
prof bench_depth2pc --workload depth2pc --no-cpu-baseline
# un-profiled bench lines (the numbers to quote: a profiled run clocks lower)
timeout 900 python bench.py > $out/line_default.json 2> $out/line_default.err < /dev/null
timeout 300 python bench.py --workload state > $out/line_state.json 2>> $out/line_default.err < /dev/null
timeout 400 python bench.py --workload vision_pn2 > $out/line_vision_pn2.json 2>> $out/line_default.err < /dev/null
timeout 400 python bench.py --workload dagger > $out/line_dagger_pointnet.json 2>> $out/line_default.err < /dev/null
timeout 400 python bench.py --workload dagger --student sparse_unet > $out/line_dagger_sparse_unet.json 2>> $out/line_default.err < /dev/null
timeout 300 python bench.py --workload dagger --student conv3d > $out/line_dagger_conv3d.json 2>> $out/line_default.err < /dev/null
timeout 300 python bench.py --workload depth2pc > $out/line_depth2pc.json 2>> $out/line_default.err < /dev/null
# PMC passes (separate runs, counters only) on short single-purpose commands
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_enc python tools/time_enc.py < /dev/null
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_lin python tools/time_gemm.py 2048x512x512 < /dev/null
PMC_PASS_TIMEOUT=200 bash tools/pmc_run.sh $out/pmc_su python tools/time_sparse_unet.py 256 < /dev/null
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_fps python bench.py --workload depth2pc --no-cpu-baseline < /dev/null
PMC_PASS_TIMEOUT=150 bash tools/pmc_run.sh $out/pmc_sa python tools/time_sa.py < /dev/null
PMC_PASS_TIMEOUT=300 bash tools/pmc_run.sh $out/pmc_su2048 python tools/time_sparse_unet.py 2048 < /dev/null
PMC_PASS_TIMEOUT=400 bash tools/pmc_run.sh $out/pmc_state env PARTMANIP_GRAPHS=0 python bench.py --workload state --lean --no-cpu-baseline --steps 2 --warmup 0 < /dev/null
python tools/make_hbm_traffic.py $tag $out/pmc_enc/summary.json $out/pmc_lin/summary.json $out/pmc_su/summary.json $out/pmc_fps/summary.json \
  sa=$out/pmc_sa/summary.json su2048=$out/pmc_su2048/summary.json state=$out/pmc_state/summary.json > $out/hbm_traffic.txt 2>&1
cp profiles/hbm_traffic.json $out/hbm_traffic.json
ls $out
