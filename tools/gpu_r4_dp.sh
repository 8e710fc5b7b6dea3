cd "$GRAFT_REPO_ROOT"
export PARTMANIP_SHARE_GPU=1 PARTMANIP_DIST_BACKEND=gloo
for w in vision vision_pn2 state; do
  timeout 900 python bench.py --gpus 2 --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-optional 2> gpurun_out/dp_$w.err | python tools/dp_line.py $w || tail -8 gpurun_out/dp_$w.err
done
