cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4m; mkdir -p $out
for i in 1 2; do
rm -f gpurun_out/parity_margins.jsonl
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_$i.log 2>&1; echo "pytest rc=$?" >> $out/pytest_$i.log
grep -v amdgpu.ids $out/pytest_$i.log | tail -6
python tools/margins_summary.py gpurun_out/parity_margins.jsonl $out/parity_margins_$i.json | grep -i "fused vs mat"
done
python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-60
