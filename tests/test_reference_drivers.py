"""The reference's OWN driver scripts, unchanged, against this package (CPU, build container only: skipped where
/root/reference is absent - the GPU box - and never part of `-m gpu`).

BASELINE.json's north_star asks that ``bnn_classify.py`` run unchanged against the new backend.  ``tests/run_reference_driver.py``
executes the script from where it lies with ``np_bnn`` resolving to ``npbnn_amd`` (the device seams served by the float64 oracle:
there is no GPU here); the same script runs beside it under the real np_bnn.  The script never seeds numpy's global stream (its
``rseed`` only reaches ``get_data``), so both runs get ``np.random.seed`` first - nothing else is touched.  Held against each
other: the 1000-row log file of the 10 000-iteration run, every prediction file, the per-sample probabilities, the feature-importance
table.  (Driver data - the example tables - are copied into the scratch directories; the scripts write next to them.)"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "bnn_classify.py")), reason="the reference repository is not mounted here")
SEED = 4321


def _scratch(tmp_path, name):
    d = tmp_path / name
    (d / "example_files").mkdir(parents=True)
    for f in ("data_features.txt", "data_labels.txt", "unlabeled_data.txt", "data_features_reg.txt", "data_lab_reg.txt"):
        shutil.copy(os.path.join(REF, "example_files", f), str(d / "example_files" / f))
    return d


# bnn_regress.py plots with seaborn, which this image does not have: both runs get a module of that name whose regplot returns the
# current axes (test infrastructure only - nothing of the script or of either package is touched), and matplotlib's Agg backend
_SEABORN_STUB = ("import types; _sb = types.ModuleType('seaborn'); "
                 "_sb.regplot = lambda *a, **k: __import__('matplotlib.pyplot').pyplot.gca(); sys.modules['seaborn'] = _sb; ")


def _run_both(tmp_path, script, threads="2", stub_seaborn=False):
    """The script under np_bnn and under npbnn_amd, side by side; returns their scratch directories."""
    env = dict(os.environ, OMP_NUM_THREADS=threads, OPENBLAS_NUM_THREADS=threads, MKL_NUM_THREADS=threads, MPLBACKEND="Agg")
    env.pop("PYTHONPATH", None)
    ref_dir, our_dir = _scratch(tmp_path, "reference"), _scratch(tmp_path, "ours")
    seeded = "import sys, runpy, numpy as np; np.random.seed(%d); " % SEED
    if stub_seaborn:
        seeded += _SEABORN_STUB
    ref = subprocess.Popen([sys.executable, "-c", seeded + "sys.path.insert(0, %r); runpy.run_path(%r, run_name='__main__')"
                            % (REF, os.path.join(REF, script))], cwd=str(ref_dir), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    ours = subprocess.Popen([sys.executable, "-c", seeded + "sys.argv = ['run_reference_driver.py', %r]; runpy.run_path(%r, run_name='__main__')"
                             % (os.path.join(REF, script), os.path.join(HERE, "run_reference_driver.py"))],
                            cwd=str(our_dir), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out_ref, out_ours = ref.communicate(timeout=1500)[0].decode(), ours.communicate(timeout=1500)[0].decode()
    assert ref.returncode == 0, out_ref[-2000:]
    assert ours.returncode == 0, out_ours[-3000:]
    return ref_dir, our_dir


@needs_reference
def test_bnn_classify_runs_unchanged_and_writes_the_references_files(tmp_path):
    ref_dir, our_dir = _run_both(tmp_path, "bnn_classify.py")
    stem = "BNN_cv0_l5_5"
    log_ref, log_ours = str(ref_dir / (stem + ".log")), str(our_dir / (stem + ".log"))
    assert open(log_ref).readline() == open(log_ours).readline()                  # the header row, column for column
    a, b = np.loadtxt(log_ref, skiprows=1), np.loadtxt(log_ours, skiprows=1)
    assert a.shape == b.shape == (1000, 19)
    np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-12)                          # 10 000 iterations, one row per ten
    for prefix in ("data_features_", "all_data_", "unlabeled_data_"):
        table = prefix + stem + "_pred_mean_pr.txt"
        assert np.array_equal(np.genfromtxt(str(ref_dir / table), dtype=str), np.genfromtxt(str(our_dir / table), dtype=str)), table
        per_sample = prefix + stem + "_pred_pr.npy"
        np.testing.assert_allclose(np.load(str(our_dir / per_sample)), np.load(str(ref_dir / per_sample)), rtol=0, atol=1e-12)
    import pandas as pd
    fi_ref, fi_ours = (pd.read_csv(str(d / "example_files" / "feature_imp.csv")) for d in (ref_dir, our_dir))
    assert list(fi_ref.columns) == list(fi_ours.columns) and fi_ref.shape == fi_ours.shape == (3, 7)
    assert list(fi_ref["feature_name"]) == list(fi_ours["feature_name"])
    np.testing.assert_allclose(fi_ours.iloc[:, 3:].to_numpy(float), fi_ref.iloc[:, 3:].to_numpy(float), rtol=1e-9)
    # the accuracy the reference reports for the held-out rows is the one this package reports (its file adds TP / FP figures)
    acc_ref = float(open(str(ref_dir / ("data_features_" + stem + "_accuracy.txt"))).read().split()[2])
    acc_ours = float(open(str(our_dir / ("data_features_" + stem + "_accuracy.txt"))).read().split()[2])
    assert acc_ref == acc_ours and acc_ours > 0.9


@needs_reference
def test_block_bnns_runs_unchanged(tmp_path):
    """block_bnns.py (the layouts of BASELINE.json config 5): get_data in regression mode, three masked models."""
    _run_both(tmp_path, "block_bnns.py")


@needs_reference
def test_bnn_regress_runs_unchanged_and_writes_the_references_files(tmp_path):
    """bnn_regress.py (BASELINE.json config 4's driver: regression with the empirical error, 20 000 iterations, then every posterior
    sample through RunPredict): the 200-row log file and the stored posterior samples of both runs."""
    import pickle
    ref_dir, our_dir = _run_both(tmp_path, "bnn_regress.py", stub_seaborn=True)
    stem = "testM_l6_4"
    log_ref, log_ours = str(ref_dir / (stem + ".log")), str(our_dir / (stem + ".log"))
    assert open(log_ref).readline() == open(log_ours).readline()
    a, b = np.loadtxt(log_ref, skiprows=1), np.loadtxt(log_ours, skiprows=1)
    assert a.shape == b.shape == (200, 18)
    np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-12)
    sys.path.insert(0, REF)
    try:
        with open(str(ref_dir / (stem + ".pkl")), "rb") as fh:
            ref_samples = pickle.load(fh)[2]._post_weight_samples
    finally:
        sys.path.remove(REF)
        for name in [k for k in sys.modules if k == "np_bnn" or k.startswith("np_bnn.")]:
            del sys.modules[name]
    import npbnn_amd as bn
    ours = bn.load_obj(str(our_dir / (stem + ".pkl")))[2]._post_weight_samples
    assert len(ours) == len(ref_samples) == 100
    for so, sr in zip(ours, ref_samples):
        assert so["mcmc_it"] == sr["mcmc_it"]
        for wo, wr in zip(so["weights"], sr["weights"]):
            np.testing.assert_allclose(wo, wr, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(np.ones(2) * so["error_prm"], np.ones(2) * sr["error_prm"], rtol=1e-10)


@needs_reference
@pytest.mark.skipif(not os.environ.get("NPBNN_SLOW_TESTS"), reason="two minutes of CPU (20 000 iterations x 4 chains twice over): NPBNN_SLOW_TESTS=1")
def test_bnn_runner_mc3_runs_unchanged_and_writes_the_references_files(tmp_path):
    """bnn_runner_MC3.py (BASELINE.json config 3's call sequence; it seeds numpy itself): the cold chain's log file, the swap
    messages it prints, the prediction files.  Last run in the build container (round 4): 200 log rows, 21 swaps, per-sample
    probabilities - all equal to the run under np_bnn, max relative difference 0.0."""
    ref_dir, our_dir = _run_both(tmp_path, "bnn_runner_MC3.py", threads="1")
    stem = "BNNMC3_l5_5"
    a, b = np.loadtxt(str(ref_dir / (stem + ".log")), skiprows=1), np.loadtxt(str(our_dir / (stem + ".log")), skiprows=1)
    assert a.shape == b.shape == (200, 19)
    np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.load(str(our_dir / (stem + "_pred_pr.npy"))), np.load(str(ref_dir / (stem + "_pred_pr.npy"))), rtol=0, atol=1e-12)
