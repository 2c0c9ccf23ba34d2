"""GPU tests of the reference's helper call surface (activation / output / likelihood / accuracy callables and the
free forward functions) served by the stand-alone device operators, against the golden vectors of the reference."""
import os

import numpy as np
import pytest

import cases
import npbnn_amd as bn
import oracle as orc

pytestmark = pytest.mark.gpu


def test_helper_callables_against_reference_golden(golden_dir):
    grid = np.load(os.path.join(golden_dir, "grid.npz"))
    case = [c for c in cases.grid_cases() if c["name"] == "f64_h32x8_c5_tanh_b2"][0]
    inp = cases.grid_inputs(case)
    k = case["name"]
    act = bn.ActFun(fun="tanh")
    x, w, lab = inp["x"], inp["weights"], inp["labels"]
    sid = np.arange(len(x))
    y = bn.RunPredict(x, w, act, bn.SoftMax)
    np.testing.assert_allclose(y[:16], grid[k + "/y_head"], atol=2e-5)
    z = bn.RunPredict(x, w, act, bn.RegressTransform)
    np.testing.assert_allclose(z[:16], grid[k + "/z_head"], rtol=2e-5, atol=2e-5)
    h0 = bn.RunHiddenLayer(x + 0, w[0], act, 0)
    np.testing.assert_allclose(h0[:16], grid[k + "/h0_head"], atol=2e-5)
    # likelihood / accuracy helpers on the float64 prediction matrix of the oracle: float64 device kernels
    y64 = orc.forward(x, w, orc.Act("tanh"), orc.out_softmax)
    lik = grid[k + "/lik"]
    np.testing.assert_allclose(bn.calc_likelihood(y64, lab, sid), lik[0], rtol=1e-12)
    np.testing.assert_allclose(bn.calc_likelihood(y64, lab, sid, instance_weight=inp["inst_w"]), lik[1], rtol=1e-12)
    np.testing.assert_allclose(bn.calc_likelihood(y64, lab, sid, class_weight=inp["class_w"]), lik[2], rtol=1e-12)
    np.testing.assert_allclose(bn.calc_likelihood(y64, lab, sid, lik_temp=0.5), lik[3], rtol=1e-12)
    with pytest.raises(Exception):
        bn.calc_likelihood(y64, lab, sid, class_weight=inp["class_w"], instance_weight=inp["inst_w"])
    assert bn.CalcAccuracy(y64, lab) == grid[k + "/acc"]
    np.testing.assert_array_equal(bn.CalcLabelAccuracy(y64, lab), grid[k + "/label_acc"])
    np.testing.assert_array_equal(bn.CalcLabelFreq(y64), grid[k + "/label_freq"])
    np.testing.assert_array_equal(bn.CalcAccuracy(np.stack([y64, y64]), lab), [grid[k + "/acc"]] * 2)
    # elementwise helpers
    zz = np.random.default_rng(0).normal(0, 2, (37, 9))
    np.testing.assert_allclose(bn.SoftMax(zz), orc.out_softmax(zz), rtol=1e-13)
    np.testing.assert_allclose(bn.SoftPlus(zz), orc.softplus(zz), rtol=1e-13)
    np.testing.assert_allclose(bn.tanh_f(zz + 0, 0), orc.activate(zz + 0, orc.Act("tanh"), 0), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(bn.swish_f(zz + 0, 0), orc.activate(zz + 0, orc.Act("swish"), 0), rtol=1e-13)
    a = zz + 0
    assert bn.relu_f(a, 0) is a and a.min() == 0.0                       # in place, like the reference
    np.testing.assert_allclose(bn.ActFun("genReLU", prm=np.array([0.1, 0.2])).eval(zz + 0, 1),
                               orc.activate(zz + 0, orc.Act("genReLU", prm=np.array([0.1, 0.2])), 1), rtol=1e-13)
    np.testing.assert_allclose(bn.MatrixMultiplicationD(x, w[0]), orc.dense(x, w[0]), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(bn.MatrixMultiplication(x, w[0]), orc.dense(x, w[0]), rtol=2e-5, atol=2e-5)
    ind = (np.random.default_rng(1).random(w[0].shape) < 0.7).astype(float)
    np.testing.assert_allclose(bn.RunPredictInd(x, w, ind, act, bn.SoftMax),
                               orc.forward(x, w, orc.Act("tanh"), orc.out_softmax, indicators=ind), atol=2e-5)


def test_regression_and_count_helpers(golden_dir):
    g = np.load(os.path.join(golden_dir, "regression.npz"))
    inp = cases.regression_inputs()
    y, t = g["y"], inp["targets"]
    np.testing.assert_allclose(bn.calc_likelihood_regression(y, t, None, sig2=1), g["lik_sig1"], rtol=1e-12)
    np.testing.assert_allclose(bn.calc_likelihood_regression(y, t, None, sig2=inp["sig_vec"]), g["lik_sigvec"], rtol=1e-12)
    np.testing.assert_allclose(bn.calc_likelihood_regression(y, t, None, sig2=g["emp_sigma"], lik_temp=0.7), g["lik_emp_temp"], rtol=1e-12)
    np.testing.assert_allclose(bn.CalcAccuracyRegression(y, t), g["mse"], rtol=1e-12)
    np.testing.assert_allclose(bn.CalcLabelAccuracyRegression(y, t), g["mse_col"], rtol=1e-12)
    inp2 = cases.regression_inputs(seed=12, double_out=True)
    np.testing.assert_allclose(bn.calc_likelihood_regression_error(g["y_err"], inp2["targets"], None), g["lik_err"], rtol=1e-12)
    z = orc.forward_logits(inp2["x"], inp2["weights"], orc.Act("tanh"))
    np.testing.assert_allclose(bn.RegressTransformError(z + 0), g["y_err"], rtol=1e-12, atol=1e-14)
    c = np.load(os.path.join(golden_dir, "counts.npz"))
    a = cases.count_inputs(seed=23, n_out=1, k=1)
    np.testing.assert_allclose(bn.poi_likelihood(c["poi_z"], a["counts"]), c["poi"], rtol=1e-11)
    np.testing.assert_allclose(bn.poi_acc(c["poi_z"], a["counts"]), c["poi_acc"], rtol=1e-12)
    b = cases.count_inputs(seed=24, n_out=2, k=1)
    np.testing.assert_allclose(bn.negbin_likelihood(c["nb_z"], b["counts"]), c["nb"], rtol=1e-9)
    np.testing.assert_allclose(bn.negbin_likelihood_base10(c["nb_z"], b["counts"]), c["nb10"], rtol=1e-9)
    np.testing.assert_allclose(bn.negbin_acc(c["nb_z"], b["counts"]), c["nb_acc"], rtol=1e-12)
    np.testing.assert_allclose(bn.negbin_acc_base10(c["nb_z"], b["counts"]), c["nb10_acc"], rtol=1e-12)
    d = cases.count_inputs(seed=25, n_out=4, k=2)
    np.testing.assert_allclose(bn.negbin_likelihood2d(c["nb2d_z"], d["counts"]), c["nb2d"], rtol=1e-9)
    np.testing.assert_allclose(bn.negbin2d_acc(c["nb2d_z"], d["counts"]), c["nb2d_acc"], rtol=1e-12)


def test_custom_callables_take_the_slow_path():
    """User output function + likelihood + accuracy callables (estimation_mode='custom', as in the reference's
    test_BNNexpectation.py): the device returns the last layer's values, the callables run on the host."""
    import contextlib, io
    rs = np.random.default_rng(3)
    n, f = 400, 12
    x = rs.standard_normal((n, f))
    counts = rs.poisson(4.0, (n, 1)).astype(float)
    dat = dict(data=x, labels=counts, test_data=np.zeros((0, f)), test_labels=np.zeros(0))

    def my_out(z):
        return np.clip(z, -5, 5)

    def my_lik(prediction, true_values, sample_id=None, class_weight=None, instance_weight=None, lik_temp=1, sig2=0):
        return float(np.sum(true_values[:, 0] * prediction[:, 0] - np.exp(prediction[:, 0])))

    np.random.seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=[6, 3], estimation_mode="custom", size_output=1, output_act_fun=my_out,
                       actFun=bn.ActFun(fun="swish"), use_bias_node=2)
    mcmc = bn.MCMC(bnn, likelihood_f=my_lik, n_iteration=100)
    y64 = my_out(orc.forward_logits(x, bnn._w_layers, orc.Act("swish")))
    np.testing.assert_allclose(mcmc._logLik, my_lik(y64, counts), rtol=1e-5)
    for _ in range(30):
        mcmc.mh_step(bnn)
    mcmc.run_steps(bnn, 20)                       # falls back to mh_step: custom likelihood
    assert mcmc._current_iteration == 50 and mcmc._accuracy == 1.0
    y64 = my_out(orc.forward_logits(x, bnn._w_layers, orc.Act("swish")))
    np.testing.assert_allclose(mcmc._logLik, my_lik(y64, counts), rtol=1e-5)
    # the fused plug-in likelihood gives the same chain law: check its value on the same weights
    with contextlib.redirect_stdout(io.StringIO()):
        bnn2 = bn.npBNN(dat, n_nodes=[6, 3], estimation_mode="custom", size_output=1, actFun=bn.ActFun(fun="swish"),
                        use_bias_node=2, init_weights=[w + 0 for w in bnn._w_layers])
    m2 = bn.MCMC(bnn2, likelihood_f=bn.poi_likelihood, accuracy_f=bn.poi_acc, n_iteration=100)
    z64 = orc.forward_logits(x, bnn2._w_layers, orc.Act("swish"))
    np.testing.assert_allclose(m2._logLik, orc.lik_poisson(z64, counts), rtol=2e-6)
    m2.run_steps(bnn2, 40)                        # device-resident chain with the fused Poisson likelihood
    z64 = orc.forward_logits(x, bnn2._w_layers, orc.Act("swish"))
    np.testing.assert_allclose(m2._logLik, orc.lik_poisson(z64, counts), rtol=2e-6)
    np.testing.assert_allclose(m2._accuracy, np.mean((np.exp(z64[:, 0]) - counts[:, 0]) ** 2), rtol=1e-4)
