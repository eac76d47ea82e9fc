import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name starts with a digit, hence importlib)."""
    return importlib.import_module("2d_object_detection_amd")


@pytest.fixture(scope="session")
def ops(pkg):
    return importlib.import_module("2d_object_detection_amd.ops")
