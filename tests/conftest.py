import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- GPU suite order ---------------------------------------------------------------------------------------------------------------
# The GPU suite is ~7 minutes on the build boxes, but a third of it is CPU time of the oracle (DenseNet121-3D epochs in fp32, two-rank
# worker processes) and varies with the host.  The oracle-heavy tests run LAST (every test is self-contained: order is free), so that a
# run cut short by its caller has the quick op-level evidence first.  No test is ever skipped by the clock: a parity test runs or fails.
# MMS_GPU_SUITE_BUDGET_S (opt-in, default 0 = off) makes an oracle-heavy test that would START after that many seconds FAIL with that
# reason -- for callers who prefer a red run to one killed at their own limit.
import time as _time

_SUITE_T0 = _time.monotonic()
_HEAVY = (          # most important first
    "test_epoch_and_validate_match_oracle_loops", "test_lockstep_epoch_matches_oracle_loops",
    "test_ddp_", "entry_point",
    "test_config1_simple_fusion_ct_stubbed", "test_gradient_error_vs_fp64", "test_config4_volume_shape_parity",
)


def _heavy_rank(item):
    for i, h in enumerate(_HEAVY):
        if h in item.nodeid:
            return i
    return -1


def pytest_collection_modifyitems(config, items):
    light = [it for it in items if _heavy_rank(it) < 0]
    heavy = sorted((it for it in items if _heavy_rank(it) >= 0), key=_heavy_rank)       # stable: file order inside a rank
    items[:] = light + heavy


def pytest_runtest_setup(item):
    budget = float(os.environ.get("MMS_GPU_SUITE_BUDGET_S", "0"))
    if budget > 0 and item.get_closest_marker("gpu") is not None and _heavy_rank(item) >= 0:
        elapsed = _time.monotonic() - _SUITE_T0
        if elapsed > budget:
            pytest.fail("GPU suite time budget exceeded: %.0f s elapsed > MMS_GPU_SUITE_BUDGET_S = %.0f s -- this parity test was NOT run"
                        % (elapsed, budget), pytrace=False)
