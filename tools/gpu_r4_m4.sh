#!/bin/bash
out=gpurun_out/r4m; mkdir -p $out
S="524288x128x128 131072x288x256 131072x256x256 131072x256x512"
for t in 22 21 12 11; do
  echo "PM_G2_WTILE=$t" | tee -a $out/gemm_wtiles.txt
  PARTMANIP_HIP_LIB=gpurun_ab/g2tile.so PM_G2_WTILE=$t python tools/time_gemm.py $S 2>&1 | grep "^M=" | sed 's/fwd.*wgrad/wgrad/' | tee -a $out/gemm_wtiles.txt
done
