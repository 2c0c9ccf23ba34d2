"""Data loading / splitting and persistence helpers: the host shim every reference driver calls first
(``bn.get_data``, np_bnn/BNN_files.py:10-99; ``randomize_data`` :190-257; ``load_obj`` :260-267; ``SaveObject``
np_bnn/BNN_lib.py:241-243).

None of this is on the accelerated path - it runs once per job on the host - but the train / test split decides which rows
the chain sees, so it consumes the seeded numpy ``Generator`` draw for draw as upstream does (tests/golden/split.npz pins the
split indices of a seeded example against the reference).  The structure is this package's own: a table reader, a label
encoder, a split planner that returns row indices, and ``get_data`` assembling upstream's dictionary from them.
"""
import os
import pickle

import numpy as np


# ---- persistence ---------------------------------------------------------------------------------
class DetachedMatrix:
    """What a light checkpoint holds in place of a feature matrix: shape and dtype, and the name it has in the side file
    ``<checkpoint>_data.npz`` (written once per run by ``postLogger``; the reference pickles the matrices with every posterior
    sample, np_bnn/BNN_env.py:655-658).  ``load_obj`` puts the matrix back when the side file is there."""

    def __init__(self, key, array):
        self.key, self.shape, self.dtype = key, tuple(np.shape(array)), str(np.asarray(array).dtype)

    def __len__(self):
        return self.shape[0] if self.shape else 0

    def __repr__(self):
        return "DetachedMatrix(%r, shape=%s, dtype=%s)" % (self.key, self.shape, self.dtype)


def data_side_file(checkpoint_name):
    stem = checkpoint_name[:-4] if checkpoint_name.endswith(".pkl") else checkpoint_name
    return stem + "_data.npz"


def attach_data(obj, source):
    """Put feature matrices back into the model of a light checkpoint: ``source`` is the side file's path, an open ``np.load``
    result or a dictionary with the keys ``data`` / ``test_data`` (what ``npBNN`` was built from)."""
    if isinstance(source, (str, os.PathLike)):
        source = np.load(source)
    for name, value in list(vars(obj).items()):
        if isinstance(value, DetachedMatrix):
            key = value.key if value.key in source else value.key.lstrip("_")
            matrix = np.asarray(source[key])
            if tuple(matrix.shape) != value.shape:
                raise ValueError("%s has shape %s; the checkpoint was written with %s" % (key, matrix.shape, value.shape))
            setattr(obj, name, matrix)
    return obj


def load_obj(file_name, dat=None):
    """The pickled object(s) of ``file_name`` (np_bnn/BNN_files.py:260-267).  Checkpoints written by ``postLogger`` keep their
    feature matrices in a side file; they are re-attached from it - or from ``dat``, the dictionary the model was built from.
    A checkpoint in upstream's format (np_bnn's own, or written here with ``export="upstream"``) comes back as objects of this
    package (npbnn_amd/export.py)."""
    with open(file_name, 'rb') as f:
        raw = f.read()
    from . import export
    if export.names_upstream(raw):        # written by np_bnn itself, or by postLogger(export="upstream") / save_upstream
        return export.loads_as_this_package(raw)
    obj = pickle.loads(raw)
    models = [o for o in (obj if isinstance(obj, (list, tuple)) else [obj])
              if any(isinstance(v, DetachedMatrix) for v in getattr(o, "__dict__", {}).values())]
    if models:
        side = data_side_file(file_name)
        source = dat if dat is not None else (np.load(side) if os.path.exists(side) else None)
        if source is None:
            print("load_obj: %s holds no feature matrices and %s is missing; pass dat= or call attach_data / update_data"
                  % (file_name, side))
        else:
            for m in models:
                attach_data(m, source)
    return obj


def SaveObject(obj, filename):
    with open(filename, 'wb') as output:
        pickle.dump(obj, output, pickle.HIGHEST_PROTOCOL)


# ---- reading tables ------------------------------------------------------------------------------
def _read_table(path, header, with_ids):
    """(values float [rows, cols], instance names or [], column names) of a .npy matrix or a whitespace-separated text
    table whose first ``header`` lines are skipped and whose first column, with ``with_ids``, names the instances."""
    names = []
    try:
        values = np.load(path, allow_pickle=True)
    except Exception:
        if with_ids:
            cells = np.genfromtxt(path, skip_header=int(header), dtype=str)
            names, values = cells[:, 0].astype(str), cells[:, 1:].astype(float)
        else:
            values = np.loadtxt(path, skiprows=int(header))
    if header:
        with open(path) as fh:
            columns = np.array(fh.readline().split()[1:])
    else:
        columns = np.array(["feature_%s" % i for i in range(values.shape[1])])
    return values, names, columns


def _frame_table(frame, with_ids):
    import pandas as pd
    frame = pd.DataFrame(frame)
    if with_ids:
        return frame.values[:, 1:], frame.values[:, 0].astype(str), np.array(frame.columns[1:])
    return frame.values, [], np.array(frame.columns)


def _read_labels(source, header, with_ids, classification):
    """Label table from a data frame / array, or - when that fails - from a text file (first column = instance names when
    ``with_ids``)."""
    import pandas as pd
    try:
        values = pd.DataFrame(source).values
        if with_ids:
            return values[:, 1:]
        return values.astype(str).flatten() if classification else values
    except Exception:
        cells = np.loadtxt(source, skiprows=int(header), dtype=str)
        return cells[:, 1:] if with_ids else cells


def turn_labels_to_numeric(labels, label_file, save_to_file=False):
    """Class names -> 0 .. C-1 in the sorted order of the names (np_bnn/BNN_files.py:300-310)."""
    labels = np.asarray(labels)
    classes = np.unique(labels)
    numeric = np.zeros(len(labels), dtype=int)
    for code, name in enumerate(classes):
        numeric[(labels == name).flatten()] = code
    if save_to_file:
        np.savetxt(label_file.replace('.txt', '_numerical.txt'), numeric, fmt='%i')
    return numeric


# ---- train / test split ---------------------------------------------------------------------------
def _split_plan(labels, testsize, all_class_in_testset, randomize, cv, rs):
    """Row indices (order of the shuffled table, training rows, test rows) of upstream's split (np_bnn/BNN_files.py:190-257).
    Draws from ``rs`` exactly what upstream draws, in its order: one permutation of the rows (only when shuffling AND a test
    share is asked for), then - for the stratified split - one sample WITH replacement per class, classes in sorted order."""
    n = len(labels)
    shuffle = bool(randomize) and bool(testsize)
    order = rs.choice(range(n), n, replace=False) if shuffle else np.arange(n)
    shuffled = labels[order]
    n_test = int(testsize * n)
    everything = np.arange(n)
    if cv > -1 and testsize:                                   # fold `cv` of consecutive blocks of the shuffled rows
        first = n_test * cv
        test = np.arange(first, min(first + n_test, n))
        return order, np.delete(everything, test), test
    if randomize and all_class_in_testset and testsize:        # every class sends max(1, share) draws to the test set
        picks = []
        for value in np.unique(shuffled):
            members = np.where(shuffled == value)[0]
            picks.extend(rs.choice(members, max(1, int(testsize * len(members)))))
        test = np.array(picks)
        # (a multi-column label table is searched entry-wise, as np.where does; training rows are counted over .size as upstream)
        train = np.array([z for z in range(shuffled.size) if z not in test])
        return order, train, test
    if n_test == 0:
        return order, everything, None
    return order, everything[:-n_test], everything[-n_test:]


def randomize_data(tot_x, tot_labels, testsize=0.1, all_class_in_testset=1, inst_id=[], randomize=True, cv=-1, rs=None):
    """Shuffle and split a labelled table; returns ``(x, labels, x_test, labels_test, ids, ids_test)``
    (np_bnn/BNN_files.py:190-257)."""
    order, train, test = _split_plan(tot_labels, testsize, all_class_in_testset, randomize, cv, rs)
    x_all, lab_all = tot_x[order], tot_labels[order]
    ids_all = inst_id[order] if len(inst_id) else []
    take = lambda table, rows: table[rows] if len(table) else []      # noqa: E731
    if test is None:
        return x_all, lab_all, [], [], take(ids_all, train), []
    return x_all[train], lab_all[train], x_all[test], lab_all[test], take(ids_all, train), take(ids_all, test)


def get_data(f, l=None, testsize=0.1, batch_training=0, seed=1234, all_class_in_testset=1,
             instance_id=0, header=0, feature_indx=None, randomize_order=True, from_file=True,
             label_mode="classification", cv=-1):
    """Features (and labels) from files, arrays or data frames into the dictionary ``npBNN`` takes
    (np_bnn/BNN_files.py:10-99): keys ``data, labels, label_dict, test_data, test_labels, id_data, id_test_data, file_name,
    feature_names``."""
    rs = np.random.default_rng(seed)
    if from_file:
        values, names, columns = _read_table(f, header, instance_id)
        stem = os.path.splitext(os.path.basename(f))[0]
    else:
        values, names, columns = _frame_table(f, instance_id)
        stem = 'bnn'
    if feature_indx is not None:
        keep = np.array(feature_indx)
        values, columns = values[:, keep], columns[keep]
    out = {'file_name': stem, 'feature_names': columns}
    if l is None:                                              # unlabelled matrix (prediction input)
        out.update(data=np.array(values).astype(float), labels=[], label_dict=[], test_data=[], test_labels=[],
                   id_data=names, id_test_data=[])
        return out
    classification = label_mode == "classification"
    raw = _read_labels(l, header, instance_id, classification)
    if classification:
        coded = turn_labels_to_numeric(raw, l)
    else:
        coded = raw.reshape((raw.shape[0], 1)) if raw.ndim == 1 else raw
    x, labels, x_test, labels_test, ids, ids_test = randomize_data(values, coded, testsize=testsize,
                                                                  all_class_in_testset=all_class_in_testset, inst_id=names,
                                                                  randomize=randomize_order, cv=cv, rs=rs)
    if batch_training:
        rows = rs.integers(0, len(labels), batch_training)
        x, labels = x[rows], labels[rows]
    if label_mode == "regression":
        labels = labels.astype(float)
        if testsize:
            labels_test = labels_test.astype(float)
    out.update(data=np.array(x).astype(float), labels=labels, label_dict=np.unique(raw),
               test_data=np.array(x_test).astype(float), test_labels=labels_test, id_data=ids, id_test_data=ids_test)
    return out
