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


@pytest.mark.parametrize("focus,n,seed", [("wfull_cq", 70, 4242), ("slab", 40, 1313), ("wfull", 40, 1414)])
def test_fixed_seed_fuzz_focus_modes(focus, n, seed):
    """Round 5: the focus modes of the fuzz -- full weights on the cooperative-quad kernels (the mode that found a miscompiled kernel
    object: draw 64 of this seed is that case), every draw forced onto the slab kernels, full weights in every draw."""
    import fuzz_gpu
    os.environ["FUZZ_FOCUS"] = focus
    try:
        worst, m = fuzz_gpu.run(n, seed, verbose=False)
    finally:
        os.environ.pop("FUZZ_FOCUS", None)
    assert m == n and worst < 1e-9, (focus, m, worst)
