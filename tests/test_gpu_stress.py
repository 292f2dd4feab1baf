"""-m gpu: a short run of tools/stress.py -- random sizes (tile / workgroup boundary cases), kernel variants,
persistent-kernel modes (register-resident tiles on/off, reducer workgroups on/off, 1-rank exchange) and
call patterns, each cross-checked against the per-stage launch chain."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_random_cases_persistent_kernel_vs_launch_chain():
    import stress
    n, worst = stress.run(budget=15.0, seed=7, max_exp=5.3, verbose=False)
    assert n >= 20 and worst <= 1e-9
