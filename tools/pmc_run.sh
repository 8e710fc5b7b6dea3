#!/bin/bash
# PMC passes for a SHORT command (run ON the GPU box, from the repo root):  tools/pmc_run.sh <outdir> <command...>
# One rocprofv3 --pmc pass per counter group (separate runs, counters only: no trace domains), then
# tools/pmc_summary.py folds the per-launch CSVs into per-kernel means.  Counter collection serialises every
# dispatch: give it a command with tens of launches (tools/time_sa.py 512, tools/time_enc.py), never a full
# bench of a many-kernel workload.  FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE needs the x2 gfx950 correction
# (MI355X_MICROARCH.md, HBM section) -- applied by the reader, not here.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout ${PMC_PASS_TIMEOUT:-150} rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$i" -- "$@" > "$out/pmc_$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python tools/pmc_summary.py "$out" "$out/summary.json" > "$out/summary.txt"
rm -rf "$out"/pmc_[0-9]        # the raw per-dispatch CSVs are tens of MB per pass: gpurun merges at most 64 MiB back
