"""Posterior prediction from stored weight samples: ``get_posterior_cat_prob``, ``sample_from_categorical``,
``predictBNN`` (reference: np_bnn/BNN_lib.py:352-501, 682-713).

The reference runs ``RunPredict`` once per stored sample - a fresh copy of the feature matrix and a full pass over
it every time.  Here the matrix is uploaded once and the samples go through ``npbnn_predict_sets``: up to three
weight sets per streaming read of X, all layers fused.  Shuffling of feature columns (feature importance) and the
posterior-predictive resampling draw from numpy's global stream exactly as the reference does."""
import os

import numpy as np

from . import _capi as capi
from .files import load_obj
from .layers import ActFun, output_kind
from .likelihoods import CalcAccuracy


class _SamplePredictor:
    """One device context for a set of stored samples: the feature matrix can be replaced (column shuffles of
    feature_importance) while network description and packed weight sets stay."""

    def __init__(self, n_features, post_samples, actFun, output_act_fun):
        from .backend import HipContext, pack_weights
        self._weights = [s["weights"] for s in post_samples]
        self._packed = np.stack([pack_weights(w) for w in self._weights])
        self._kind = output_kind(output_act_fun)
        self._out_fn = output_act_fun
        self._act = actFun
        self._n_features = n_features
        self._slopes = None
        if actFun._function == "genReLU":
            n_hidden = len(self._weights[0]) - 1
            self._slopes = [np.asarray(s["alphas"], dtype=float)[:n_hidden] for s in post_samples]
        self._ctx = HipContext()
        self._arch_set = False

    def predict(self, features):
        """[n_samples, n_rows, n_out] predictions of every stored sample on ``features``."""
        ctx = self._ctx
        ctx.set_data(features)
        if not self._arch_set:
            ctx.set_arch_from_weights(self._weights[0], self._n_features, self._act.device_kind(),
                                      capi.OUT_IDENTITY if self._kind is None else self._kind, capi.LIK_NONE)
            self._arch_set = True
        y = ctx.predict_sets(list(self._packed), act_prm_sets=self._slopes, apply_out_fn=self._kind is not None)
        if self._kind is None and self._out_fn is not None:      # custom output callable: host side, sample by sample
            y = np.array([self._out_fn(yi) for yi in y])
        return y

    def close(self):
        self._ctx.close()


def _predict_samples(features, post_samples, actFun, output_act_fun):
    pred = _SamplePredictor(features.shape[1], post_samples, actFun, output_act_fun)
    try:
        return pred.predict(features)
    finally:
        pred.close()


def sample_from_categorical(posterior_weights=None, post_prob_file=None, verbose=False):
    """Posterior-predictive resampling (np_bnn/BNN_lib.py:682-713): one class drawn per (instance, stored sample) from that
    sample's class probabilities; the point estimate of an instance is the class frequency among its draws.  The uniforms
    come from numpy's global stream in the upstream order (instance by instance, one per sample), so a seeded run
    reproduces upstream's draws; the arithmetic is done for all instances at once."""
    if posterior_weights is None:
        if not post_prob_file:
            raise ValueError("sample_from_categorical needs posterior_weights or post_prob_file")
        posterior_weights = np.load(post_prob_file)
    probs = np.asarray(posterior_weights)
    n_samples, n_instances, n_classes = probs.shape
    cdf = np.cumsum(np.transpose(probs, (1, 0, 2)), axis=2)               # [instance, sample, class]
    u = np.random.random((n_instances, n_samples))
    # the class drawn is the one whose cumulative probability exceeds u by the least (ties and a cdf that never gets there
    # fall to the first such class, as upstream's argmin over the masked differences does)
    excess = cdf - u[:, :, None]
    excess[excess < 0] = 1
    draws = np.argmin(excess, axis=2)                                      # [instance, sample]
    one_hot = draws[:, :, None] == np.arange(n_classes)[None, None, :]
    return {'predictions': one_hot.sum(axis=1) / n_samples,
            'class_counts': one_hot.sum(axis=0).astype(float),
            'post_predictions': draws.astype(float)}


def _shuffled_copy(features, columns, independently):
    """``features`` with the given column(s) permuted between the rows: every column on its own, or the block as a whole
    (rows stay intact inside the block).  Permutations from numpy's global stream (np_bnn/BNN_lib.py:366-372)."""
    out = np.array(features, dtype=np.float64, copy=True)
    if not columns:
        return out
    if independently and type(columns) == list:
        for col in columns:
            out[:, col] = np.random.permutation(out[:, col])
    else:
        out[:, columns] = np.random.permutation(out[:, columns])
    return out


def _summarise(probs, mode):
    """[sample, instance, class] probabilities -> [instance, class]: 0 share of samples voting for the class, 1 mean
    probability, 2 posterior-predictive resampling (np_bnn/BNN_lib.py:382-395)."""
    if mode == 0:
        votes = np.argmax(probs, axis=2)                                   # [sample, instance]
        return (votes[:, :, None] == np.arange(probs.shape[2])).mean(axis=0)
    if mode == 1:
        return np.mean(probs, axis=0)
    if mode == 2:
        return sample_from_categorical(posterior_weights=probs)['predictions']
    return None


def get_posterior_cat_prob(pred_features, post_samples=None, feature_index_to_shuffle=None, post_summary_mode=0,
                           unlink_features_within_block=False, actFun=None, output_act_fun=None, _predictor=None):
    """Predictions of every stored posterior sample and their summary (np_bnn/BNN_lib.py:352-397).  Returns
    ``(per-sample predictions [sample, instance, output], summary [instance, output])``."""
    if len(pred_features) == 0:
        print("Data not found.")
        return 0
    features = _shuffled_copy(pred_features, feature_index_to_shuffle, unlink_features_within_block)
    act = ActFun() if actFun is None else actFun
    if len(post_samples):
        act.reset_prm(post_samples[-1]['alphas'])          # (upstream leaves the last sample's slopes installed)
    probs = _predictor.predict(features) if _predictor is not None else _predict_samples(features, post_samples, act, output_act_fun)
    return probs, _summarise(probs, post_summary_mode)


def get_posterior_est(pkl_file):
    """Predictions of every stored posterior sample of a checkpoint on its own training and test matrices, and their means
    over the samples (np_bnn/BNN_lib.py:715-748; the regression drivers read the estimated parameters from it).  Keys as
    upstream: ``post_est`` / ``post_est_test`` [sample, row, output], ``prm_mean`` / ``prm_mean_test`` [row, output],
    ``error_prm`` (the samples' error parameters, or an empty list when the model has none)."""
    model, _, logger = load_obj(pkl_file)
    samples = logger._post_weight_samples
    act = model._act_fun
    if len(samples):
        act.reset_prm(samples[-1]['alphas'])          # (upstream leaves the last sample's slopes installed)

    def on(matrix):
        if len(matrix) == 0:
            return np.zeros((len(samples), 0, model._size_output))
        return _predict_samples(np.asarray(matrix, dtype=np.float64), samples, act, model._output_act_fun)

    est, est_test = on(model._data), on(model._test_data)
    return {'prm_mean': np.mean(est, axis=0), 'post_est': est,
            'prm_mean_test': np.mean(est_test, axis=0), 'post_est_test': est_test,
            'error_prm': [s['error_prm'] for s in samples] if (len(samples) and 'error_prm' in samples[0]) else []}


def _confusion_table(true_labels, predicted, n_classes):
    table = np.zeros((n_classes, n_classes), dtype=int)
    np.add.at(table, (np.asarray(true_labels, dtype=int), np.asarray(predicted, dtype=int)), 1)
    return table


def predictBNN(predict_features, pickle_file, test_labels=[], instance_id=[], post_summary_mode=0, fname="", wd="",
               verbose=1):
    """Posterior predictions for a feature matrix from a checkpoint ``[bnn, mcmc, logger]`` (np_bnn/BNN_lib.py:404-501,
    without its Bayes-factor and threshold extras).  Files, next to the checkpoint or in ``wd``:
    ``<fname_><checkpoint>_pred_pr.npy`` (every sample's predictions), ``..._pred_mean_pr.txt`` (the summary, with the
    instance names in front when given), ``..._accuracy.txt`` when labels are given."""
    model, _, logger = load_obj(pickle_file)
    per_sample, summary = get_posterior_cat_prob(predict_features, logger._post_weight_samples,
                                                 post_summary_mode=post_summary_mode, actFun=model._act_fun,
                                                 output_act_fun=model._output_act_fun)
    stem = os.path.join(wd if wd else os.path.dirname(pickle_file),
                        (fname + "_" if fname else "") + os.path.splitext(os.path.basename(pickle_file))[0])
    result = {'post_prob_predictions': summary, 'mean_accuracy': np.nan, 'confusion_matrix': np.nan}
    if len(test_labels):
        result['mean_accuracy'] = np.mean(CalcAccuracy(summary, test_labels))
        result['confusion_matrix'] = _confusion_table(test_labels, np.argmax(summary, axis=1), summary.shape[1])
        with open(stem + '_accuracy.txt', 'w') as fh:
            fh.write("Mean accuracy: %s" % result['mean_accuracy'])
        if verbose:
            print("Accuracy:", result['mean_accuracy'])
            print("Confusion matrix:\n", result['confusion_matrix'])
    if len(instance_id):
        names = np.asarray(instance_id).reshape(-1, 1)
        np.savetxt(stem + '_pred_mean_pr.txt', np.hstack((names, np.round(summary, 4).astype(str))), fmt='%s', delimiter='\t')
    else:
        np.savetxt(stem + '_pred_mean_pr.txt', summary, fmt='%.3f')
    np.save(stem + '_pred_pr.npy', per_sample)
    if verbose:
        print("Predictions saved in files:\n    %s\n    %s\n" % (stem + '_pred_pr.npy', stem + '_pred_mean_pr.txt'))
    return result


def _feature_blocks(feature_blocks, names):
    """(column lists, block names): the caller's dict {name: columns} or list of column lists; one block per column when
    none is given (np_bnn/BNN_lib.py:521-533)."""
    if isinstance(feature_blocks, dict):
        if feature_blocks:
            return list(feature_blocks.values()), list(feature_blocks.keys())
        return [[i] for i in range(len(names))], list(names)
    return list(feature_blocks), ['block_%d' % i for i in range(len(feature_blocks))]


def feature_importance(input_features, weights_pkl=None, weights_posterior=None, true_labels=[], fname_stem='',
                       feature_names=[], verbose=False, post_summary_mode=0, n_permutations=100, feature_blocks=dict(),
                       write_to_file=True, predictions_outdir='', unlink_features_within_block=True, actFun=None,
                       output_act_fun=None):
    """Permutation importance (np_bnn/BNN_lib.py:504-597): how much accuracy is lost when a block of feature columns is
    shuffled between the instances, ``n_permutations`` shuffles per block (numpy's global stream, upstream's order).  The
    stored samples stay packed on the device; a shuffle costs one upload of the matrix and one ``npbnn_predict_sets``.
    Returns a data frame, most important block first, with upstream's column names; written to
    ``<fname_stem_>feature_importance.txt`` unless ``write_to_file`` is off."""
    import pandas as pd
    features = np.asarray(input_features)
    names = feature_names if len(feature_names) else np.arange(features.shape[1]).astype(str)
    blocks, block_names = _feature_blocks(feature_blocks, names)
    if weights_pkl:
        model, _, logger = load_obj(weights_pkl)
        weights_posterior, actFun, output_act_fun = logger._post_weight_samples, model._act_fun, model._output_act_fun
    act = ActFun() if actFun is None else actFun

    predictor = _SamplePredictor(features.shape[1], weights_posterior, act, output_act_fun)
    try:
        def accuracy(shuffle=None):
            summary = get_posterior_cat_prob(features, weights_posterior, feature_index_to_shuffle=shuffle,
                                             post_summary_mode=post_summary_mode, actFun=act, output_act_fun=output_act_fun,
                                             unlink_features_within_block=unlink_features_within_block, _predictor=predictor)[1]
            return CalcAccuracy(summary, true_labels)

        baseline = accuracy()
        if verbose:
            print("Reference accuracy (mean):", np.mean(baseline))
        shuffled = np.array([[accuracy(block) for _ in range(n_permutations)] for block in blocks])    # [block, permutation]
    finally:
        predictor.close()

    loss = baseline - shuffled
    table = pd.DataFrame({'feature_block_index': np.arange(len(blocks)).astype(str), 'feature_name': [str(n) for n in block_names],
                          'delta_acc_mean': loss.mean(axis=1), 'delta_acc_std': loss.std(axis=1),
                          'acc_with_feature_randomized_mean': shuffled.mean(axis=1),
                          'acc_with_feature_randomized_std': shuffled.std(axis=1)})
    table = table.sort_values('delta_acc_mean', ascending=False)
    if write_to_file:
        out_dir = predictions_outdir if predictions_outdir else os.path.dirname(weights_pkl)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, (fname_stem + "_" if fname_stem else "") + 'feature_importance.txt')
        table.to_csv(path, sep='\t', index=False, header=True, float_format='%.6f')
        print("Output saved in: %s" % path)
    return table
