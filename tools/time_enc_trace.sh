#!/bin/bash
# kernel-level view of the encoder-only command: rocprofv3 --kernel-trace --stats of tools/time_enc.py -> <out>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out.d -o p -- python tools/time_enc.py > $out.log 2>&1
cp $out.d/p_kernel_stats.csv ${out}_kernel_stats.csv; rm -rf $out.d
