import os
import sys

import pytest

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
