import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    has_gpu = None
    for item in items:
        if "gpu" in item.keywords:
            if has_gpu is None:
                has_gpu = _has_gpu()
            if not has_gpu:
                item.add_marker(pytest.mark.skip(reason="no GPU visible"))
        if "reference" in item.keywords and not os.path.isdir("/root/reference"):
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))
