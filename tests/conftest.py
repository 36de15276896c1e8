"""Shared pytest configuration: the ``gpu`` marker and repo-root imports."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def weight_cache(tmp_path_factory):
    """Directory holding generated safetensors weight files for this test session."""
    d = os.environ.get("OPD_WEIGHT_CACHE")
    if d:
        os.makedirs(d, exist_ok=True)
        return d
    return str(tmp_path_factory.mktemp("weights"))
