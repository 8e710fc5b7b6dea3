#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (new full-size cases), pm_tanh2 NaN-select A/B, default bench line with secondaries
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
rm -f gpurun_out/parity_margins.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -30 gpurun_out/r3a/pytest.log
for i in 1 2; do
  python tools/time_enc.py 2>&1 | tail -1 | sed 's/^/nan-select: /' >> gpurun_out/r3a/tanh_ab.txt
  PARTMANIP_HIP_LIB=gpurun_ab/tanhdrop.so python tools/time_enc.py 2>&1 | tail -1 | sed 's/^/drops-nan : /' >> gpurun_out/r3a/tanh_ab.txt
done
cat gpurun_out/r3a/tanh_ab.txt
( time timeout 900 python bench.py ) > gpurun_out/r3a/line_default.json 2> gpurun_out/r3a/line_default.err
tail -5 gpurun_out/r3a/line_default.err
python tools/margins_summary.py gpurun_out/parity_margins.jsonl gpurun_out/r3a/parity_margins.json | tail -20
