import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference (oracle/_ref), i.e. the build container")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libaadref.so"))
    gpu = None
    for item in items:
        if "ref" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="oracle/_ref not built (reference sources absent)"))
        if "gpu" in item.keywords:
            if gpu is None:
                gpu = _has_gpu()
            if not gpu:
                item.add_marker(pytest.mark.skip(reason="no GPU in this environment"))


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build its C restatement (and, in the build container,
    the reference itself) once per session."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
