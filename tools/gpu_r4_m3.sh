#!/bin/bash
out=gpurun_out/r4m; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -k "linear or gemm or mlp or groupall or sa_ or pointnet2" 2>&1 | tail -2
python tools/time_gemm.py 524288x128x128 131072x288x256 131072x256x512 2048x512x512 2>&1 | grep "^M="
for w in vision_pn2 state; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $out/line_$w.json
  python -c "
import json; d=json.loads(open('$out/line_$w.json').read()); print('$w', round(d['value'],1), round(d['ms_per_step'],2))"
done
timeout 600 python bench.py --workload dagger --student sparse_unet --no-cpu-baseline 2>/dev/null | tail -1 > $out/line_su.json
python -c "
import json; d=json.loads(open('$out/line_su.json').read()); print('sparse_unet', round(d['value'],1), round(d['ms_per_step'],2))"
