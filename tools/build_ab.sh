#!/bin/bash
# build an A/B variant of one kernel file into gpurun_ab/<name>.so:  tools/build_ab.sh <name> <file.hip> <extra flags...>
set -e
name=$1; src=$2; shift 2
mkdir -p gpurun_ab/obj_$name
objs=""
for f in gae losses adam gemm_f32 gemm2_f32 pointnet_enc pointnet_enc_bf3 pointnet_enc_bf6 pointops sa_fused sa_groupall conv3d sparse_voxel; do
  if [ "$f.hip" == "$src" ]; then
    /opt/rocm/bin/hipcc -x hip -c partmanip_amd/csrc/$f.hip -o gpurun_ab/obj_$name/$f.o -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off "$@"
    objs="$objs gpurun_ab/obj_$name/$f.o"
  else
    objs="$objs partmanip_amd/lib/obj/$f.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_ab/$name.so $objs
echo gpurun_ab/$name.so
