"""Checkpoints in upstream's format (npbnn_amd/export.py) on CPU.  The committed file tests/golden/export_upstream.pkl was
written by this package and then OPENED AND CONTINUED BY np_bnn ITSELF when the fixtures were generated (make_golden.py G10:
np_bnn's mh_step went on from it for 120 iterations and landed call for call on np_bnn's own uninterrupted trace).  Here:
the exporter still writes that file's content, the file names nothing of this package, and this package reads it back -
objects, restart and continuation."""
import contextlib
import io
import os
import pickletools

import numpy as np
import pytest

import cases
import npbnn_amd as bn
import option_traces as ot
from oracle_backend import OracleChainBackend, serve_from_oracle

FIXTURE = "export_upstream.pkl"


def _export_again(tmp_path):
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    dat, bnn, mcmc = cases.option_chain(bn, cases.EXPORT_CASE)
    logger = bn.postLogger(bnn, filename="EXPORT", wdir=str(tmp_path), export="upstream")
    with contextlib.redirect_stdout(io.StringIO()):
        bn.run_mcmc(bnn, mcmc, logger)
    return dat, bnn, mcmc, logger


def _globals_named(path):
    names = set()
    strings = []
    with open(path, "rb") as fh:
        for op, arg, _ in pickletools.genops(fh.read()):
            if op.name == "GLOBAL":
                names.add(tuple(arg.split(" ")))
            elif op.name in ("SHORT_BINUNICODE", "BINUNICODE", "UNICODE"):
                strings.append(arg)
            elif op.name == "STACK_GLOBAL":
                names.add((strings[-2], strings[-1]))
    return names


def test_export_names_only_upstream_modules(tmp_path):
    _, _, _, logger = _export_again(tmp_path)
    names = _globals_named(logger._pklfile)
    assert not [n for n in names if n[0].startswith("npbnn_amd")], names
    upstream = {n for n in names if n[0].startswith("np_bnn")}
    assert {("np_bnn.BNN_env", "npBNN"), ("np_bnn.BNN_env", "MCMC"), ("np_bnn.BNN_env", "postLogger"), ("np_bnn.BNN_lib", "ActFun"),
            ("np_bnn.BNN_lib", "SoftMax"), ("np_bnn.BNN_lib", "leaky_relu_f"), ("np_bnn.BNN_lib", "calc_likelihood"),
            ("np_bnn.BNN_lib", "CalcAccuracy"), ("np_bnn.BNN_lib", "CalcLabelAccuracy"), ("np_bnn.BNN_mcmc", "UpdateNormal")} <= upstream
    # the committed file - the one np_bnn itself opened and continued - names the same things
    assert _globals_named(os.path.join(os.path.dirname(cases.__file__), FIXTURE)) == names


def test_exporter_still_writes_what_upstream_accepted(tmp_path, golden_dir):
    """Object by object, attribute by attribute: a fresh export against the committed one (both read by this package)."""
    _, _, _, logger = _export_again(tmp_path)
    fresh = bn.load_obj(logger._pklfile)
    kept = bn.load_obj(os.path.join(golden_dir, FIXTURE))
    for a, b in zip(fresh, kept):
        assert type(a) is type(b)
        da, db = dict(vars(a)), dict(vars(b))
        for skip in ("_logfile", "_w_file", "_pklfile", "_bnn", "_backend", "_gen", "_act_fun", "_lazy"):
            da.pop(skip, None), db.pop(skip, None)
        assert sorted(da) == sorted(db)
        for key in da:
            if callable(da[key]):
                assert da[key] is db[key], key
            else:
                np.testing.assert_equal(da[key], db[key], err_msg=key)
    assert fresh[1]._gen.bit_generator.state == kept[1]._gen.bit_generator.state
    for name in ("_y", "_y_test", "_accuracy", "_test_accuracy", "_label_acc", "_label_freq"):
        np.testing.assert_allclose(getattr(fresh[1], name), getattr(kept[1], name), rtol=1e-12)
    np.testing.assert_array_equal(fresh[0]._act_fun._acc_prm, kept[0]._act_fun._acc_prm)
    assert fresh[0]._act_fun.activate is kept[0]._act_fun.activate is bn.leaky_relu_f


def test_this_package_continues_from_an_upstream_format_checkpoint(golden_dir):
    """load_obj of the committed upstream-format file gives this package's objects; the chain goes on from them (served by the
    oracle stand-in) along np_bnn's own uninterrupted trace - the mirror image of what np_bnn did with the same file."""
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    bnn, mcmc, logger = bn.load_obj(os.path.join(golden_dir, FIXTURE))
    assert isinstance(bnn, bn.npBNN) and isinstance(mcmc, bn.MCMC) and isinstance(logger, bn.postLogger)
    assert mcmc._current_iteration == cases.EXPORT_SWITCH and len(logger._post_weight_samples) == 6
    states = np.load(os.path.join(golden_dir, "options.npz"))["%s/states" % cases.EXPORT_CASE]
    np.testing.assert_allclose(cases.option_state(bnn, mcmc), states[cases.EXPORT_SWITCH - 1], rtol=1e-10)
    for it in range(cases.EXPORT_SWITCH, cases.OPTION_TRACES[cases.EXPORT_CASE]["steps"]):
        mcmc.mh_step(bnn)
        np.testing.assert_allclose(cases.option_state(bnn, mcmc), states[it], rtol=1e-10, err_msg="iteration %d" % it)


def test_restart_from_the_last_posterior_sample(golden_dir):
    """npBNN(pickle_file=...) (np_bnn/BNN_env.py:117-120,129-131) on the upstream-format file, against what np_bnn's own
    constructor read from it."""
    g = np.load(os.path.join(golden_dir, "export.npz"))
    cfg = cases.OPTION_TRACES[cases.EXPORT_CASE]
    dat = cases.option_data(cfg)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=cfg["n_nodes"], use_bias_node=cfg["bias"], pickle_file=os.path.join(golden_dir, FIXTURE),
                       actFun=bn.ActFun(fun="genReLU", prm=np.zeros(2), trainable=True))
    np.testing.assert_array_equal(bnn._w_layers[0], g["restart_w0"])
    np.testing.assert_array_equal(np.asarray(bnn._act_fun._prm, dtype=float), g["restart_alphas"])


def test_export_refuses_a_model_without_its_data(tmp_path):
    from npbnn_amd.export import save_upstream
    serve_from_oracle(lambda b: OracleChainBackend(b, 0))
    _, bnn, mcmc = cases.option_chain(bn, cases.EXPORT_CASE)
    bnn._data = bn.DetachedMatrix("data", bnn._data)
    with pytest.raises(ValueError, match="attach the data"):
        save_upstream([bnn], str(tmp_path / "x.pkl"))
