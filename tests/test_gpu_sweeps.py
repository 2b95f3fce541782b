"""Bounded, fixed-seed slices of the randomised parity sweeps (tools/stress_parity.py, tools/stress_matcher.py):
random image sizes, pyramid parameters, thresholds, lapping ranges, batch sizes and trig modes for the extractor;
random problem sizes over every matcher entry point.  About a minute on the GPU box; the tools run longer sweeps."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("seed", [7, 2026])
def test_extractor_random_configurations(seed):
    import stress_parity
    lines = []
    valid, bad = stress_parity.run(ncases=60, seed=seed, max_side=(700, 1000), log=lambda *a: lines.append(" ".join(map(str, a))))
    assert not bad, "\n".join(lines)
    assert valid >= 30, "\n".join(lines)  # (pyramids whose top level cannot hold one 35-px cell are refused, as documented)


@pytest.mark.parametrize("seed", [3, 99])
def test_matcher_random_problems(seed):
    import stress_matcher
    lines = []
    bad = stress_matcher.run(ncases=360, seed=seed, scale=1.0, log=lambda *a: lines.append(" ".join(map(str, a))))
    assert not bad, "\n".join(lines)
