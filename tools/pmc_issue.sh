#!/bin/bash
# Issue-slot counters of a SHORT command (run ON the GPU box from the repo root):  tools/pmc_issue.sh <outdir> <command...>
# Three rocprofv3 --pmc passes (counters only, no trace domains): which instruction classes kept the SIMDs busy, and whether VALU and
# MFMA work overlapped (SQ_VALU_MFMA_COEXEC_CYCLES) -- the counters behind `roofline.floor_ms`.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  timeout ${PMC_PASS_TIMEOUT:-150} rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$i" -- "$@" > "$out/pmc_$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python tools/pmc_summary.py "$out" "$out/summary.json" > "$out/summary.txt"
rm -rf "$out"/pmc_[0-9]
