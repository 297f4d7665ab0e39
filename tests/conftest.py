import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))

EARTH_PPM = os.path.join(REPO, "rust-raytracing_amd", "assets", "earthmap.ppm")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = [int(x) for x in parts[1].split()]
    return np.frombuffer(parts[3], dtype=np.uint8, count=w * h * 3).reshape(h, w, 3)


@pytest.fixture(scope="session")
def earth_rgb():
    return read_ppm(EARTH_PPM)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def rtamd():
    import rtamd as m
    m.load_libraries()
    return m


@pytest.fixture(scope="session")
def gpu(rtamd):
    """One context for the whole GPU session (one process, one GPU)."""
    ctx = rtamd.Context(0)
    yield ctx
    ctx.close()
