"""stdin: output of `bench.py --gpus N` (gloo prints its own lines) -> the data-parallel fields of the JSON line."""
import json, sys
name = sys.argv[1] if len(sys.argv) > 1 else ""
lines = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not lines:
    print(name, "NO JSON LINE")
    sys.exit(1)
j = json.loads(lines[0])
dp = j["config"].get("data_parallel", {})
print(name, "n_gpus", j["n_gpus"], "value", round(j["value"]), "ms", round(j["ms_per_step"], 1), "comm_ms", j.get("comm_ms_per_step"),
      "all-reduces/step", dp.get("all_reduces_per_step"), "equal", dp.get("param_checksum_equal"), dp.get("graph_mode"),
      dp.get("per_rank_ms_per_step"))
