import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# the reference's own tolerances (test/evalGrad.jl:4-5)
RTOL = 1e-10
ATOL = 1e-14


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: full-length (32 386-step) DUPLICATE of a test that also runs shortened or on another parameter set; "
                                       "skipped unless --runslow / JQ_RUN_SLOW=1 (every kernel family keeps one full-length test in the default suite)")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="also run the tests marked slow")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow") or os.environ.get("JQ_RUN_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow duplicate: --runslow / JQ_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
        return json.load(f)


def case_inputs(case):
    """(params, info, pcof0, golden-or-None) for a named set-up of juqbox_jl_amd.cases."""
    import juqbox_jl_amd as jq
    params, info = jq.cases.BUILDERS[case]()
    golden = load_golden(info["golden"]) if info.get("golden") else None
    if golden is not None and "pcof0" in golden:
        pcof = np.array(golden["pcof0"])
    else:
        pcof = np.asarray(info["pcof0"])
    return params, info, pcof, golden


def reference_pass(value, ref):
    """The pass criterion of test/evalGrad.jl:43-67: abs diff < atol, or rel diff < rtol."""
    value = np.atleast_1d(np.asarray(value, dtype=np.float64))
    ref = np.atleast_1d(np.asarray(ref, dtype=np.float64))
    d = np.linalg.norm(value - ref)
    nrm = np.linalg.norm(ref)
    return d < ATOL or (nrm >= ATOL and d / nrm < RTOL)


@pytest.fixture(scope="session")
def jq():
    import juqbox_jl_amd
    return juqbox_jl_amd
