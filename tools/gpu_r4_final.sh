#!/bin/bash
# final check of the round: the whole -m gpu suite, smoke(), the default bench line with the driver's arguments
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4final; mkdir -p $out
rm -f gpurun_out/parity_margins.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -v amdgpu.ids $out/pytest.log | tail -12
python __graft_entry__.py smoke 2>&1 | tail -2
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/line_default_driver_args.json 2> $out/line_default_driver_args.err
tail -4 $out/line_default_driver_args.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4final/line_default_driver_args.json"))
print("headline", j["value"], j["roofline"]["frac"], j["ms_per_step"], "traffic", j["roofline"]["traffic"])
for k,v in j.get("secondary",{}).items():
    r=v.get("roofline") or {}
    print(k, v.get("value"), v.get("error"), "cpu", (v.get("cpu_baseline") or {}).get("value"), "frac", r.get("frac"), "traffic", r.get("traffic"))
PY
python tools/margins_summary.py gpurun_out/parity_margins.jsonl $out/parity_margins.json | tail -3
