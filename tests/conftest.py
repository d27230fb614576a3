import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ASSETS = os.path.join(ROOT, "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("RF_REFERENCE", "/root/reference")
STEMS = ("mnet-deconv-0517", "mnet25")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "model"))


needs_reference = pytest.mark.skipif(not has_reference(), reason="/root/reference is only mounted in the dev container")


@pytest.fixture(scope="session")
def built_lib():
    """The product library; built on demand in the dev container, shipped prebuilt to the GPU box."""
    import retinaface_amd
    if not os.path.exists(retinaface_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return retinaface_amd.load_library()


@pytest.fixture(scope="session")
def nets():
    from oracle.caffe_io import read_rfw
    return {s: read_rfw(os.path.join(ASSETS, s + ".rfw")) for s in STEMS}


@pytest.fixture(scope="session")
def oracles(nets):
    from oracle.pipeline import OracleDetector
    return {s: OracleDetector(n) for s, n in nets.items()}


@pytest.fixture(scope="session")
def base_frame():
    from retinaface_amd.frames import padded_base_frame
    return padded_base_frame()


@pytest.fixture(scope="session")
def crop448(base_frame):
    import numpy as np
    return np.ascontiguousarray(base_frame[30:478, 440:888])


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
