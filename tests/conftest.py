import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (compiled on first use with gcc; a few seconds)."""
    from oracle import oracle_api
    oracle_api.build()
    return oracle_api


@pytest.fixture(scope="session")
def ctx():
    """A device context of the HIP library; GPU tests only."""
    from sfm_amd import capi
    c = capi.Context(0)
    if os.environ.get("SFM_SIM_VARIANT"):  # run the suite against another similarity-kernel instance
        c.set_option("sim_variant", int(os.environ["SFM_SIM_VARIANT"]))
    yield c
    c.close()
