#!/bin/bash
out=gpurun_out/r4m; mkdir -p $out
S="2048x512x512 2048x53x512 131072x256x512 131072x128x128 8388608x96x32 2680950x192x64"
for t in 22 21 12; do
  echo "PM_G2_TILE=$t" | tee -a $out/gemm_tiles2.txt
  PARTMANIP_HIP_LIB=gpurun_ab/g2tile.so PM_G2_TILE=$t python tools/time_gemm.py $S 2>&1 | grep "^M=" | tee -a $out/gemm_tiles2.txt
done
