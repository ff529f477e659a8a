import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def has_reference():
    return os.path.isdir('/root/reference/images')


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs a kernel must not hold the box until the outer limit: every gpu test gets a timeout
    (pytest-timeout, thread method: the process is terminated, a signal cannot interrupt a blocked HIP call)."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker('gpu') is not None and item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(180, method='thread'))
