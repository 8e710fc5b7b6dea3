#!/bin/bash
# round 4, call b: whole -m gpu suite, GAE envs-per-work-group A/B, default bench line with its four secondaries
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4b; mkdir -p $out
rm -f gpurun_out/parity_margins.jsonl
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -v amdgpu.ids $out/pytest.log | tail -45
for e in 16 32 64; do echo "PM_GAE_E=$e" >> $out/gae.txt; PM_GAE_E=$e python tools/time_gae.py 2>&1 | grep "us per" >> $out/gae.txt; done
cat $out/gae.txt
( time timeout 1200 python bench.py ) > $out/line_default.json 2> $out/line_default.err
tail -4 $out/line_default.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4b/line_default.json"))
print("headline", j["value"], j["roofline"]["frac"], j["ms_per_step"])
for k,v in j.get("secondary",{}).items():
    print(k, {a:(v.get(a) if not isinstance(v.get(a),dict) else "...") for a in ("value","unit","ms_per_step","error","gpu_over_cpu","wall_s_incl_setup_and_cpu_baseline")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"))
PY
python tools/margins_summary.py gpurun_out/parity_margins.jsonl $out/parity_margins.json | tail -5
