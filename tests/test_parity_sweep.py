"""GPU: a reduced tools/parity_sweep.py under pytest, so that the driver runs it -- random worlds (stand-in graphs with random numbers of haplotypes, mutation
densities, large gaps, identical haplotypes, k-mer merge lengths; small Graph M worlds with allele-rich gene windows), random read lengths, insert sizes, clips and
numbers of secondary alignments (48 worlds): extended chains column by column, pair records and the work counters (DP calls, iterations, cells) equal the oracle's; some of the
worlds once more as unpaired reads in long-read mode (alignOneLongRead, mapper/processBAM.cpp:3618-3838).  The path: mapper/processBAM.cpp:3129-3616,
mapper/aligner/extensionAligner.cpp:186-1556."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEEDS = list(range(9000, 9048))          # seeds with s % 3 == 2 are Graph M worlds (16 of the 48); even seeds of the others add the unpaired pass (16 more)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_world_is_bit_exact(pkg, oracle, seed):
    from tools import parity_sweep
    st = parity_sweep.sweep_world(pkg, seed, 500)
    if st is None:
        pytest.skip("the generator produced a record the reference asserts on (the oracle raised)")
    assert st.n_errors == 0 and st.n_dp_calls > 0
