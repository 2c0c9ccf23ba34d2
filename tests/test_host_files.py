"""get_data / randomize_data (the host shim every reference driver calls first, np_bnn/BNN_files.py:10-99,190-257) against the
reference's own output on seeded example tables (tests/golden/split.npz, group G8 of make_golden.py): same rows in the same
order in the training and test sets, same instance names, same label coding."""
import contextlib
import io
import os

import numpy as np
import pytest

import cases
import npbnn_amd as bn

KEYS = ("data", "labels", "test_data", "test_labels", "id_data", "id_test_data", "label_dict", "feature_names")


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    return cases.write_split_tables(str(tmp_path_factory.mktemp("split")))


@pytest.mark.parametrize("name,kw", cases.SPLIT_CASES, ids=[c[0] for c in cases.SPLIT_CASES])
def test_get_data_splits_as_the_reference(name, kw, tables, golden_dir):
    g = np.load(os.path.join(golden_dir, "split.npz"))
    f_x, f_lab, f_y = tables
    d = quiet(bn.get_data, f_x, f_y if kw.get("label_mode") == "regression" else f_lab, **kw)
    assert d["file_name"] == "split_features"
    for key in KEYS:
        want = g["%s_%s" % (name, key)]
        got = np.asarray(d[key])
        if want.dtype.kind == "U":
            assert [str(v) for v in got.ravel()] == [str(v) for v in want.ravel()], key
        else:
            assert got.shape == want.shape, key
            np.testing.assert_array_equal(got, want, err_msg=key)


def test_get_data_without_labels(tables, golden_dir):
    g = np.load(os.path.join(golden_dir, "split.npz"))
    d = quiet(bn.get_data, tables[0], header=1, instance_id=1)
    np.testing.assert_array_equal(d["data"], g["unlabelled_data"])
    assert list(d["id_data"]) == list(g["unlabelled_id_data"])
    assert d["labels"] == [] and d["test_data"] == []


def test_get_data_from_arrays_and_the_driver_recipe(tables):
    """from_file=False takes data frames / arrays; and the dictionary feeds npBNN directly, as bnn_classify.py:15-37 does."""
    import pandas as pd
    rs = np.random.default_rng(1)
    x = rs.standard_normal((60, 4))
    lab = rs.integers(0, 3, 60)
    d = quiet(bn.get_data, pd.DataFrame(x, columns=list("abcd")), pd.DataFrame(lab.astype(str)), from_file=False, testsize=0.1, seed=4)
    assert d["file_name"] == "bnn" and list(d["feature_names"]) == list("abcd")
    assert len(d["labels"]) + len(d["test_labels"]) >= 60 - 6 and set(np.unique(d["labels"])) <= {0, 1, 2}
    again = quiet(bn.get_data, pd.DataFrame(x, columns=list("abcd")), pd.DataFrame(lab.astype(str)), from_file=False, testsize=0.1, seed=4)
    np.testing.assert_array_equal(d["data"], again["data"])
    x2, l2, xt, lt, _, _ = bn.randomize_data(x, lab, testsize=0.25, all_class_in_testset=0, rs=np.random.default_rng(0))
    assert x2.shape == (45, 4) and xt.shape == (15, 4) and len(l2) == 45 and len(lt) == 15
