#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python __graft_entry__.py smoke 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_sparse_unet.py -m gpu -q -k "full_size_forward" 2>&1 | tail -2; done
python __graft_entry__.py smoke 2>&1 | tail -2
