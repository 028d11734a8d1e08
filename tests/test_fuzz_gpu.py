"""A seeded slice of tools/fuzz_parity.py in the GPU suite: 20 random configurations (dataset, group sizes, lengths,
padding, widths / heads, layer counts, projection depth, every ablation switch) of the HIP path against the CPU oracle
-- outputs, losses, matcher indices, the set of parameters with a gradient and the gradients."""
import os
import random
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_seeded_fuzz_slice_against_the_oracle():
    from fuzz_parity import fuzz_case
    rng = random.Random(20261003)
    bad = []
    for case in range(20):
        tag, status = fuzz_case(rng, case)
        if status != "ok":
            bad.append(tag + " -> " + status)
    assert not bad, "\n".join(bad)
