import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _record_parity_margins(request, monkeypatch):
    """GPU tests: every np.testing.assert_allclose also appends what it OBSERVED to gpurun_out/parity_margins.jsonl
    (tests/helpers.py; folded into profiles/parity_margins.json by tools/margins_summary.py)."""
    if "gpu" in request.keywords:
        import numpy as np
        from tests import helpers
        monkeypatch.setattr(np.testing, "assert_allclose", helpers.recording_allclose)
    yield
