import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def free_port():
    """An unused TCP port on 127.0.0.1 for a torch.distributed rendezvous (no fixed port: parallel test runs collide)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def dbg_lib():
    """The test compares an alternative kernel form with the default one: it needs the switches of include/musehip_dbg.h, which only
    libmusehip_dbg.so exports.  Every library call of the test goes through that build; the production library is restored after."""
    from musediffusion_amd import _lib
    with _lib.debug_library() as handle:
        yield handle
