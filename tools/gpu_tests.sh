#!/bin/bash
# the whole -m gpu suite on the box + the observed-margin summary:  gpurun -- 'bash tools/gpu_tests.sh [pytest args]'
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/parity_margins.jsonl
timeout 2400 python -m pytest tests -m gpu -q --durations=12 "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "amdgpu.ids" gpurun_out/pytest_gpu.log | tail -60
python tools/margins_summary.py gpurun_out/parity_margins.jsonl gpurun_out/parity_margins.json | tail -25
