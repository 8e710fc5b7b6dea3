cd "$GRAFT_REPO_ROOT"
python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-80
for v in 1e30 nan; do
python tools/poison_run.py 120 $v tests/test_gpu_sparse_unet.py -m gpu -q -x 2>&1 | tail -6
done
