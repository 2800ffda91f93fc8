"""GPU: a fixed-seed slice of the randomised parity sweep (tests/fuzz_all.py): random architectures (depth, uniform / ragged
/ odd widths, 1..16 inputs, 1..4 outputs, tanh / relu / identity, bias on / off), row subsets and weight scales through
whichever kernel family the dispatcher picks, SSE / gradient / predictions against the oracle (the reference's torch
float64 module + autograd).  Bars: 1e-11 on SSE and predictions, 1e-10 of max |g| on gradients."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 4])
def test_random_architectures_match_the_oracle(seed):
    import fuzz_all
    nfail, worst = fuzz_all.run(ncases=40, seed=seed, verbose=False)
    assert nfail == 0, worst


def test_random_residual_networks_match_the_oracle():
    import fuzz_all
    nfail, worst = fuzz_all.run_rnet(ncases=30, seed=5, verbose=False)
    assert nfail == 0, worst


def test_random_sampler_settings_reproduce_the_oracle_chains():
    """AMCMC chains bit for bit, HMC / MALA acceptance indices (tests/fuzz_all.py: run_mcmc)."""
    import fuzz_all
    assert fuzz_all.run_mcmc(ncases=3, seed=9, verbose=False) == 0


def test_random_elbo_estimates_and_gradients_match_the_oracle():
    import fuzz_all
    nfail, worst = fuzz_all.run_vi(ncases=25, seed=6, verbose=False)
    assert nfail == 0, worst


def test_non_finite_values_at_random_places_follow_the_reference():
    """NaN / Inf / huge / denormal weights, inputs and targets on random networks (tanh / relu / identity, padded twins,
    int8-slice and fused kernels): NaN / +Inf / -Inf pattern of SSE, predictions and gradient as torch's."""
    import fuzz_all
    assert fuzz_all.run_exceptional(ncases=80, seed=4, verbose=False) == 0


def test_random_training_loops_match_the_oracle():
    import fuzz_all
    assert fuzz_all.run_fit(ncases=12, seed=8, verbose=False) == 0


def test_device_resident_samplers_on_random_networks():
    import fuzz_all
    assert fuzz_all.run_device(ncases=12, seed=3, verbose=False) == 0


def test_random_vi_fits_match_the_oracle_loop():
    import fuzz_all
    assert fuzz_all.run_vifit(ncases=5, seed=2, verbose=False) == 0


def test_random_batched_ensemble_fits_match_member_after_member_oracle_loops():
    import fuzz_all
    assert fuzz_all.run_ens(ncases=10, seed=5, verbose=False) == 0
