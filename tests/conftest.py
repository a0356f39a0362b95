import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    yield O
    O.set_math("glibc")


@pytest.fixture(scope="session")
def tum_depth():
    import synth
    return synth.load_png_u16(synth.TUM_PNG)


@pytest.fixture(scope="session")
def icl_depth():
    import synth
    return synth.load_png_u16(synth.ICL_PNG)
