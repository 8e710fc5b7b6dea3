cd "$GRAFT_REPO_ROOT"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import sys, torch
sys.path.insert(0, '.')
import tests.test_gpu_sparse_unet as T
bad = 0
for i in range(25):
    try:
        T.test_full_size_forward_and_gradients_fused_vs_materialised_and_vs_restatement(256)
    except AssertionError as e:
        bad += 1
        print("rep", i, "FAILED:", str(e)[:200])
print("failures:", bad, "of 25")
PY
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_run_loop.py tests/test_gpu_sparse_unet.py -m gpu -q 2>&1 | tail -1; done
