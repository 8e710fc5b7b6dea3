#!/bin/bash
# round 4, call i: gathered GEMMs (cfg 5's two big convolutions) over tile shapes x LDS stages (A/B build with G2_TILE_ENV)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4i; mkdir -p $out
export PARTMANIP_HIP_LIB=gpurun_ab/g2env.so
for cfg in "- -" "11 2" "12 2" "21 2" "22 2" "22 3" "21 3"; do
  set -- $cfg
  echo "== PM_G2_TILE=$1 PM_G2_NBUF=$2" >> $out/sparse_conv.txt
  ( [ $1 != - ] && export PM_G2_TILE=$1 PM_G2_WTILE=$1; [ $2 != - ] && export PM_G2_NBUF=$2; timeout 300 python tools/time_sparse_conv.py 2>&1 | grep "gathered" >> $out/sparse_conv.txt )
done
cat $out/sparse_conv.txt
