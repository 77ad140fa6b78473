import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("MS_GUARD_PAGES"):      # tests/guard_pages.py: every allocation of the package at the edge of its own hipMalloc'd block
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import guard_pages
        guard_pages.install()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
