import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure), built on demand."""
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def img256():
    from snesimage_amd.synth import synth_image
    return synth_image()


@pytest.fixture(scope="session")
def img256_alpha():
    from snesimage_amd.synth import synth_image
    return synth_image(0x5EED0001, variant=1)
