#!/bin/bash
# kernel-trace of a short command, summarised by kernel and grid:  tools/prof_cmd.sh <outdir-under-gpurun_out> <command...>
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -o p -- "$@" > $out/cmd.log 2>&1
python tools/trace_summary.py $out/raw/p_kernel_trace.csv $out/kernel_by_grid.csv 60 < /dev/null
cp $out/raw/p_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/raw
tail -3 $out/cmd.log
