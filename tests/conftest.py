import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the library honours its diagnostic switches (LW_HIP_MSM_C, LW_HIP_SRS_FOLD_MIN, ...) only when the process was started
# with LW_HIP_TUNING=1; the parity tests sweep some of them
os.environ.setdefault("LW_HIP_TUNING", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)
