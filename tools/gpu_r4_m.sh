#!/bin/bash
# plain-GEMM tile / stage A/B at the PointNet++ glue shapes (gpurun_ab/g2tile.so: -DG2_TILE_ENV)
out=gpurun_out/r4m; mkdir -p $out
S="524288x128x128 131072x288x256 131072x256x256"
echo "default build" | tee $out/gemm_tiles.txt
python tools/time_gemm.py $S 2>&1 | grep "^M=" | tee -a $out/gemm_tiles.txt
for t in 22 12 21 11; do for nb in 3 2; do
  echo "PM_G2_TILE=$t PM_G2_NBUF=$nb" | tee -a $out/gemm_tiles.txt
  PARTMANIP_HIP_LIB=gpurun_ab/g2tile.so PM_G2_TILE=$t PM_G2_NBUF=$nb python tools/time_gemm.py $S 2>&1 | grep "^M=" | tee -a $out/gemm_tiles.txt
done; done
