import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_torch_cache(request):
    """The at-scale GPU tests generate their reads with torch; its caching allocator keeps every block it ever took, and the
    library allocates beside it with hipMalloc -- after a few tests of 10^7 ... 10^8 reads the cache holds tens of GB the next
    test's device table cannot have (C4's device-table build failed with KMR_ERR_OOM behind the config-3 test).  Hand the cache
    back after every GPU test."""
    yield
    if "gpu" in request.keywords and "torch" in sys.modules:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
