"""GPU: every sampler option beyond the default path, free-running against the REFERENCE's recorded traces
(tests/golden/options.npz: np_bnn itself on the same seeds).  Both ways the product advances a chain are held to them:
``MCMC.mh_step`` (one fused device evaluation per proposal) and ``MCMC.run_steps`` (the device-resident chains,
npbnn_chain_run / npbnn_chain_run_general, with mh_step inside where no device chain covers an option).

The device computes row terms in float32, so a decision may flip where |logPost' - logPost - log u| falls below that noise,
after which two correct chains part ways; the bar is the one of test_hip_sampler.py: the same accept / reject sequence AND
state (log-likelihood and sigma 2e-6 relative; prior 1e-9; slopes, indicators, prior scales as the host keeps them) over at least 150 calls - and
a state that differs while every decision was the same is an error wherever it happens."""
import os

import numpy as np
import pytest

import cases
import npbnn_amd as bn
import option_traces as ot

pytestmark = pytest.mark.gpu

RTOL_LIK = 2e-6
COMMON = 150


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "options.npz"))


@pytest.mark.parametrize("advance", ["mh_step", "run_steps"])
@pytest.mark.parametrize("name", list(cases.OPTION_TRACES))
def test_free_running_chain_follows_the_reference_under_every_option(name, advance, golden):
    g = ot.load(golden, name)
    _, bnn, mcmc = cases.option_chain(bn, name)
    ot.check_initial_state(name, g, bnn, mcmc, rtol_lik=RTOL_LIK, rtol_stats=2e-3)
    n_calls = len(g["states"])
    n = ot.follow(name, g, bnn, mcmc, rtol_lik=RTOL_LIK, advance=advance, stats_every=50, rtol_stats=2e-3, chunk=40,
                  rtol_state=2e-6)     # (an empirical sigma comes from float32 residuals)
    assert n >= min(COMMON, n_calls), "%s via %s: left the reference's accept / reject sequence after %d calls" % (name, advance, n)
    if n == n_calls:
        ot.check_final_state(name, g, bnn, mcmc, rtol_stats=2e-3)


@pytest.mark.parametrize("name", ["slopes", "feature_ind", "weight_ind", "fixed_normal", "normalized", "sigma", "hyper2"])
def test_device_chain_is_really_used_for_the_option(name, golden):
    """The run_steps leg above must not pass by quietly falling back to mh_step: these options have a device chain."""
    _, bnn, mcmc = cases.option_chain(bn, name)
    cfg = cases.OPTION_TRACES[name]
    if name == "feature_ind":
        mcmc.run_steps(bnn, 11)            # (the batch that straddles adapt_stop is cut; iterations from 11 on are one side)
    before = mcmc._device_iterations
    mcmc.run_steps(bnn, 40)
    assert mcmc._device_iterations - before == 40, (name, cfg.get("update_function"))
