#!/bin/bash
# round 5, call C: where the PointNet++ step's tensor-library copies come from; kernel trace of the vision_pn2 step; the oracle's PN2 cost
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/aten_sites.py vision_pn2 > gpurun_out/aten_pn2.log 2>&1; grep -v amdgpu.ids gpurun_out/aten_pn2.log | tail -45
timeout 300 python gpurun_ab/dbg_pn2.py > gpurun_out/dbg_pn2.log 2>&1; grep -v amdgpu.ids gpurun_out/dbg_pn2.log | cut -c1-200 | tail -50
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pn2 -o pn2 -- python $GRAFT_REPO_ROOT/bench.py --workload vision_pn2 --steps 3 --warmup 1 --no-cpu-baseline --lean > $GRAFT_REPO_ROOT/gpurun_out/prof_pn2.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_pn2.err
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_pn2 | head; find gpurun_out/prof_pn2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/pn2_kernel_stats.csv; head -30 gpurun_out/pn2_kernel_stats.csv | cut -c1-160
find gpurun_out/prof_pn2 -type f ! -name "*stats.csv" -delete
