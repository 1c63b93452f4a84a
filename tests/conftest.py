import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch brings its own copy of the HIP runtime: let it initialise first so that the engine library (linked against
    # /opt/rocm) and torch.distributed share one runtime in this process (bench.py does the same)
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def cfg4_scene():
    """BASELINE config 4/5 (500 images x 5000 points, U = 18 014): built once per session (about 15 s)."""
    from bundle_adjustment_amd import scene
    return scene.config("cfg4")
