"""GPU (-m gpu): a fixed-seed slice of the randomised stress of scripts/fuzz_gpu.py (problem sizes 2..96, structures,
kernel-family overrides, chunkings, ensemble sizes, both integrators) against the CPU oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "scripts"))


def test_fixed_seed_fuzz_matches_oracle():
    import fuzz_gpu
    worst, n = fuzz_gpu.run(60, 11, verbose=False)
    # n = cases really compared with the oracle (drawn combinations without kernels are skipped, not counted)
    assert n == 60 and worst < 1e-9, (n, worst)
