"""First layers whose width is not a multiple of 16, three or more output tiles (NetMeta::l0_rows, csrc/npbnn_common.hip.h): the
fp16-split image holds ceil(width / tiles) rows per tile instead of 16 - the reference's DEFAULT network [50, 5] (np_bnn/BNN_env.py:20)
on 256 features then takes 57 KB instead of 70 KB per candidate and two candidates share a read of X.  Against the float64 oracle:
plain evaluations and predictions, and a device chain whose accepted proposals have patched layer 0, the bias column and layer 1 of
that layout entry by entry."""
import contextlib
import io

import numpy as np
import pytest

import cases
import oracle as orc
import npbnn_amd as bn

pytestmark = pytest.mark.gpu


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


@pytest.mark.parametrize("hidden,fun,bias", [([50, 5], "ReLU", 1), ([50, 5], "tanh", 2), ([33, 8], "tanh", 3), ([70, 3], "swish", -1),
                                             ([45], "tanh", 2), ([61, 16, 4], "ReLU", 2)])
def test_plain_evaluation_against_the_oracle(hidden, fun, bias):
    from npbnn_amd import _capi as capi
    rs = np.random.default_rng(len(hidden) * 100 + hidden[0])
    n, f, c = 1203, 256, 7
    x = rs.standard_normal((n, f))
    lab = rs.integers(0, c, n)
    w = [rs.normal(0, 0.5 / np.sqrt(s[1] / 8 + 1), s) for s in cases.layer_shapes(f, hidden, c, bias)]
    act = orc.Act(fun)
    y64 = orc.forward(x, w, act, orc.out_softmax)
    want = orc.lik_categorical(y64, lab, np.arange(n))
    ctx = bn.HipContext(0)
    ctx.set_data(x)
    ctx.set_labels(lab)
    ctx.set_arch_from_weights(w, f, {"ReLU": 0, "swish": 2, "tanh": 3}[fun], 0, 0, 0)
    got = ctx.predict(w)
    assert np.max(np.abs(got - y64)) <= 2e-5
    np.testing.assert_allclose(ctx.eval(w)["loglik"], want, rtol=5e-6)
    assert ctx.l0_mode() == "f16-split"
    if hidden == [50, 5]:
        assert ctx.info(capi.INFO_MAX_CANDIDATES) == 2          # (one, with 16 rows per tile: a 70-KB image)
    ctx.close()


@pytest.mark.parametrize("hidden,fun", [([50, 5], "ReLU"), ([40, 6], "tanh")])
def test_chain_on_the_compact_image_stays_on_the_true_weights(hidden, fun):
    """Every accepted proposal is committed to the device's weight image entry by entry, at the positions the host worked out for
    this layout; after a few hundred accepts the chain's log-likelihood must still be the one of its float64 weights - by a fresh
    evaluation on the device and by the oracle - and the chain must be the mh_step loop's."""
    rs = np.random.default_rng(3)
    n, f, c = 5000, 256, 6
    x = rs.standard_normal((n, f)).astype(np.float32)
    proj = rs.standard_normal((f, c)) / np.sqrt(f)
    y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    chains = []
    for _ in range(2):
        np.random.seed(1234)
        bnn = quiet(bn.npBNN, dat, n_nodes=hidden, actFun=bn.ActFun(fun=fun), use_bias_node=2, prior_f=1, p_scale=1)
        chains.append((bnn, bn.MCMC(bnn, update_f=[0.01, 0.05, 0.1], update_ws=[0.02, 0.05, 0.05])))
    (bnn_a, mcmc_a), (bnn_b, mcmc_b) = chains
    for _ in range(4):
        mcmc_b.run_steps(bnn_b, 100)
    assert mcmc_b._device_iterations == 400 and mcmc_b._device_accepted > 60
    assert mcmc_b._device_passes < 400                          # (more than one candidate per pass)
    fresh = mcmc_b._backend.evaluate(bnn_b._w_layers, None)["loglik"]
    np.testing.assert_allclose(mcmc_b._logLik, fresh, rtol=1e-12)
    act = orc.Act(fun)
    want = orc.lik_categorical(orc.forward(x.astype(np.float64), bnn_b._w_layers, act, orc.out_softmax), y, np.arange(n))
    np.testing.assert_allclose(mcmc_b._logLik, want, rtol=2e-6)
    for _ in range(400):
        mcmc_a.mh_step(bnn_a)
    assert mcmc_a._last_accepted_mem == mcmc_b._last_accepted_mem
    for wa, wb in zip(bnn_a._w_layers, bnn_b._w_layers):
        np.testing.assert_array_equal(wa, wb)
