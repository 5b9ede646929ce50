import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: -m gpu tests whose wall time is mostly CPU-side KKT accounting of the device results (oracle/kkt_check.py) or a long "
                            "stress loop; `-m 'gpu and not slow'` is the quick validation tier (< 90 s of a GPU lease), the driver's `-m gpu` runs everything")


def pytest_collection_modifyitems(config, items):
    """The slow tier is derived, not hand-kept: a GPU test that calls the parity accounting (tests/_parity.py::account -> KKT checks of every device
    result that is not within 1e-4 of the oracle's) is slow; so are the ones marked explicitly."""
    import inspect
    for it in items:
        fn = getattr(it, "function", None)
        if fn is None or it.get_closest_marker("gpu") is None:
            continue
        try:
            src = inspect.getsource(fn)
        except (OSError, TypeError):
            continue
        if "account(" in src or "kkt_many" in src or "SLOW_TIER" in src:
            it.add_marker(pytest.mark.slow)


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as CO
    CO.build()
    return CO
