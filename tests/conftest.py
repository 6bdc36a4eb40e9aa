import os
import sys

import pytest
import torch  # noqa: F401  before libptcore.so is loaded: one HIP runtime per process (pathtracer_amd/dist.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def small_sky():
    """Full-size synthetic sky (the lookup hard-codes 2500 x 1250, src/scene.rs:163-165)."""
    from pathtracer_amd import scenes
    return scenes.make_sky()


# A full-size C5 test spends minutes on the CPU side (10 M-triangle scene + oracle build) without printing;
# runners that take a silent process for a hung one get a line on the real stderr once a test runs over a minute.
_current = {"id": None, "t0": 0.0}


def _heartbeat():
    import time
    while True:
        time.sleep(30)
        if _current["id"] and time.time() - _current["t0"] > 60:
            print(f"[still running] {_current['id']} {time.time() - _current['t0']:.0f} s", file=sys.__stderr__, flush=True)


def pytest_sessionstart(session):
    import threading
    threading.Thread(target=_heartbeat, daemon=True).start()


def pytest_runtest_logstart(nodeid, location):
    import time
    _current["id"], _current["t0"] = nodeid, time.time()


def pytest_runtest_logfinish(nodeid, location):
    _current["id"] = None
