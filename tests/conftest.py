import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (build container only)")


@pytest.fixture(scope="session")
def fir():
    import __graft_entry__ as ge

    return ge.load_package()


@pytest.fixture(scope="session")
def fir_audit():
    """The same package over libfir_amd_audit.so (-DFIR_AUDIT): the only build with the knobs that can change answers."""
    import __graft_entry__ as ge

    return ge.load_package(audit=True)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.load_oracle()
