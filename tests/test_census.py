"""CPU: the tie-margin census in small (the full 2 x 1e6-step run is tools/decode_census.py ->
profiles/r02_decode_census.json).  The reference's CRF arithmetic (ont-seqdist-cuda 0.0.4) cannot be run here, so
label parity with it is bounded instead: implementations that differ from the contract only in ROUNDING (libm instead of
the polynomial exp/log; seqdist's softmax-normalised posteriors) must agree with it except on near-tied time steps."""
import numpy as np

import oracle
from conftest import random_scores


def test_rounding_variants_agree_with_the_contract():
    flips = steps = near = 0
    for nb, seed in ((5, 11), (6, 12)):
        sc = random_scores(1000, 24, nb, seed=seed)
        contract = oracle.decode(sc, nb, 3)["labels"]
        twin = oracle.decode_logdomain(sc, nb, 3, libm=False, want=("gap",))
        assert np.array_equal(contract, twin["labels"])          # the same arithmetic written twice
        for other in (oracle.decode_logdomain(sc, nb, 3, libm=True)["labels"],
                      oracle.decode_logdomain(sc, nb, 3, softmax=True)["labels"]):
            diff = contract != other
            # every disagreement sits on a step whose label-relevant max-marginal margin is a few ulps of the path sums
            assert np.all(twin["gap"][diff] < 4e-3)
            flips += int(diff.sum())
        steps += contract.size
        near += int((twin["gap"] < 1e-3).sum())
    assert flips <= max(2, steps // 20000), (flips, steps)      # full run: 2..4 per 1.02e6 steps
    assert near < steps // 500                                   # full run: 3.1e-4 of the steps


def test_float64_grade_evaluation_is_a_different_rounding_family():
    """The scaled-probability evaluation (float64-grade accuracy) is NOT what a log-domain fp32 implementation such as
    seqdist computes: it disagrees on ~4e-4 of the steps of flat random posteriors, which is why the contract stays in
    the log domain."""
    sc = random_scores(2000, 16, 6, seed=2026)
    a = oracle.decode(sc, 6, 3)["labels"]
    b = oracle.decode_scaled(sc, 6, 3)["labels"]
    assert 0 < (a != b).mean() < 5e-3
