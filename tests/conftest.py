import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vectorsearch-rbac_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# PyTorch wheels bundle their own HIP runtime; when torch and libvsrbac share a process torch must load first so
# that both bind the same libamdhip64 ("Sharing a process with PyTorch").
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle("strict")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
