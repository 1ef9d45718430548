import os
import sys

import pytest

# the oracle's OpenMP loops are tiny in the tests: on a 128-core GPU host the default team size only adds overhead
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tiny_map():
    from covins_amd import synth
    return synth.make_map(synth.config_named("tiny"))


@pytest.fixture(scope="session")
def small_map():
    from covins_amd import synth
    return synth.make_map(synth.config_named("small"))
