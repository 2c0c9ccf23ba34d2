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
from .layers import ActFun, SoftMax, output_kind
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
    """One categorical draw per instance and posterior sample; point estimate = class frequencies of the draws
    (reference: BNN_lib.py:682-713)."""
    if posterior_weights is not None:
        pass
    elif post_prob_file:
        posterior_weights = np.load(post_prob_file)
    else:
        print("Input pickle file or posterior weights required.")
    n_post_samples, n_instances, n_classes = posterior_weights.shape
    res = np.zeros((n_instances, n_post_samples))
    point_estimates = np.zeros((n_instances, n_classes))
    for j in range(n_instances):
        if j % 1000 == 0 and verbose is True:
            print(j)
        p = np.cumsum(posterior_weights[:, j, :], axis=1)
        r = np.random.random(len(p))
        q = p - r.reshape(len(r), 1)
        q[q < 0] = 1
        classification = np.argmin(q, axis=1)
        res[j, :] = classification
        counts = np.bincount(classification, minlength=n_classes)
        point_estimates[j, :] = counts / np.sum(counts)
    class_counts = np.zeros((n_post_samples, n_classes))
    for i in range(res.shape[1]):
        class_counts[i] = np.bincount(res[:, i].astype(int), minlength=n_classes)
    return {'predictions': point_estimates, 'class_counts': class_counts, 'post_predictions': res}


def get_posterior_cat_prob(pred_features, post_samples=None, feature_index_to_shuffle=None, post_summary_mode=0,
                           unlink_features_within_block=False, actFun=None, output_act_fun=None, _predictor=None):
    """Class probabilities of every posterior sample and their summary (reference: BNN_lib.py:352-397):
    mode 0 frequency of the arg-max class over the samples, 1 mean probabilities, 2 posterior-predictive resampling."""
    if len(pred_features) == 0:
        print("Data not found.")
        return 0
    predict_features = np.array(pred_features, dtype=np.float64, copy=True)
    if feature_index_to_shuffle:     # permute the given feature column(s) between the instances
        if unlink_features_within_block and type(feature_index_to_shuffle) == list:
            for feature_index in feature_index_to_shuffle:
                predict_features[:, feature_index] = np.random.permutation(predict_features[:, feature_index])
        else:
            predict_features[:, feature_index_to_shuffle] = np.random.permutation(predict_features[:, feature_index_to_shuffle])
    if actFun is None:
        actFun = ActFun()
    if len(post_samples):
        actFun.reset_prm(post_samples[-1]['alphas'])          # the reference leaves the last sample's slopes installed
    if _predictor is not None:
        post_softmax_probs = _predictor.predict(predict_features)
    else:
        post_softmax_probs = _predict_samples(predict_features, post_samples, actFun, output_act_fun)
    if post_summary_mode == 0:
        class_call_posterior = np.argmax(post_softmax_probs, axis=2).T
        n_posterior_samples, n_instances, n_classes = post_softmax_probs.shape
        posterior_prob_classes = np.zeros([n_instances, n_classes])
        for c in range(n_classes):
            posterior_prob_classes[:, c] = np.sum(class_call_posterior == c, axis=1)
        posterior_prob_classes = posterior_prob_classes / n_posterior_samples
    elif post_summary_mode == 1:
        posterior_prob_classes = np.mean(post_softmax_probs, axis=0)
    elif post_summary_mode == 2:
        posterior_prob_classes = sample_from_categorical(posterior_weights=post_softmax_probs)['predictions']
    return post_softmax_probs, posterior_prob_classes


def predictBNN(predict_features, pickle_file, test_labels=[], instance_id=[], post_summary_mode=0, fname="", wd="",
               verbose=1):
    """Posterior predictions for new data from a saved run ``[bnn, mcmc, logger]`` (reference: BNN_lib.py:404-501;
    the Bayes-factor and posterior-threshold extras of the reference are not carried over).  Writes
    ``<name>_pred_pr.npy`` (all samples) and ``<name>_pred_mean_pr.txt`` (summary) next to the pickle or into ``wd``."""
    bnn_obj, mcmc_obj, logger_obj = load_obj(pickle_file)
    post_samples = logger_obj._post_weight_samples
    out_name = os.path.basename(os.path.splitext(pickle_file)[0])
    predictions_outdir = wd if wd != "" else os.path.dirname(pickle_file)
    post_softmax_probs, post_prob_predictions = get_posterior_cat_prob(
        predict_features, post_samples, post_summary_mode=post_summary_mode, actFun=bnn_obj._act_fun,
        output_act_fun=bnn_obj._output_act_fun)
    if fname != "":
        fname = fname + "_"
    out_file_post_pr = os.path.join(predictions_outdir, fname + out_name + '_pred_pr.npy')
    out_file_mean_pr = os.path.join(predictions_outdir, fname + out_name + '_pred_mean_pr.txt')
    if len(test_labels) > 0:
        mean_accuracy = np.mean(CalcAccuracy(post_prob_predictions, test_labels))
        n_classes = post_prob_predictions.shape[1]
        cm_out = np.zeros((n_classes, n_classes), dtype=int)
        np.add.at(cm_out, (np.asarray(test_labels, dtype=int), np.argmax(post_prob_predictions, axis=1)), 1)
        if verbose:
            print("Accuracy:", mean_accuracy)
            print("Confusion matrix:\n", cm_out)
        with open(os.path.join(predictions_outdir, fname + out_name + '_accuracy.txt'), 'w') as outf:
            outf.writelines("Mean accuracy: %s" % mean_accuracy)
    else:
        mean_accuracy = np.nan
        cm_out = np.nan
    if len(instance_id):
        instance_id = np.asarray(instance_id)
        post_prob_predictions_id = np.hstack((instance_id.reshape(len(instance_id), 1),
                                              np.round(post_prob_predictions, 4).astype(str)))
        np.savetxt(out_file_mean_pr, post_prob_predictions_id, fmt='%s', delimiter='\t')
    else:
        np.savetxt(out_file_mean_pr, post_prob_predictions, fmt='%.3f')
    np.save(out_file_post_pr, post_softmax_probs)
    if verbose:
        print("Predictions saved in files:")
        print('   ', out_file_post_pr)
        print('   ', out_file_mean_pr, "\n")
    return {'post_prob_predictions': post_prob_predictions, 'mean_accuracy': mean_accuracy, 'confusion_matrix': cm_out}


def feature_importance(input_features, weights_pkl=None, weights_posterior=None, true_labels=[], fname_stem='',
                       feature_names=[], verbose=False, post_summary_mode=0, n_permutations=100, feature_blocks=dict(),
                       write_to_file=True, predictions_outdir='', unlink_features_within_block=True, actFun=None,
                       output_act_fun=None):
    """Accuracy lost when a feature (or block of features) is shuffled between the instances, ``n_permutations`` times
    per block (reference: BNN_lib.py:504-597).  Every permutation is one upload of the shuffled matrix and one
    ``npbnn_predict_sets`` over all stored samples; the permutations themselves come from numpy's global stream as in
    the reference.  Returns the reference's data frame, sorted by decreasing mean accuracy loss."""
    import pandas as pd
    features = np.asarray(input_features)
    feature_indices = np.arange(features.shape[1])
    if len(feature_names) == 0:
        feature_names = feature_indices.astype(str)
    if type(feature_blocks) is dict:
        if len(feature_blocks.keys()) > 0:
            selected_features = list(feature_blocks.values())
            feature_block_names = list(feature_blocks.keys())
        else:
            selected_features = [[i] for i in feature_indices]
            feature_block_names = [i for i in feature_names]
    else:
        selected_features = feature_blocks
        feature_block_names = ['block_' + str(i) for i in range(len(feature_blocks))]
    if weights_pkl:
        bnn_obj, mcmc_obj, logger_obj = load_obj(weights_pkl)
        weights_posterior = logger_obj._post_weight_samples
        actFun = bnn_obj._act_fun
        output_act_fun = bnn_obj._output_act_fun
    if actFun is None:
        actFun = ActFun()
    predictor = _SamplePredictor(features.shape[1], weights_posterior, actFun, output_act_fun)
    try:
        _, post_prob_predictions = get_posterior_cat_prob(features, weights_posterior, post_summary_mode=post_summary_mode,
                                                          actFun=actFun, output_act_fun=output_act_fun, _predictor=predictor)
        ref_accuracy = CalcAccuracy(post_prob_predictions, true_labels)
        if verbose:
            print("Reference accuracy (mean):", np.mean(ref_accuracy))
        accuracies_wo_feature = []
        for block_id, feature_block in enumerate(selected_features):
            if verbose:
                print('Processing feature block %i', block_id + 1)
            n_accuracies = []
            for _ in np.arange(n_permutations):
                _, post_prob_predictions = get_posterior_cat_prob(
                    features, weights_posterior, feature_index_to_shuffle=feature_block, post_summary_mode=post_summary_mode,
                    unlink_features_within_block=unlink_features_within_block, actFun=actFun, output_act_fun=output_act_fun,
                    _predictor=predictor)
                n_accuracies.append(CalcAccuracy(post_prob_predictions, true_labels))
            accuracies_wo_feature.append(n_accuracies)
    finally:
        predictor.close()
    accuracies_wo_feature = np.array(accuracies_wo_feature)
    delta_accs = ref_accuracy - accuracies_wo_feature
    df = pd.DataFrame(np.array([np.arange(0, len(selected_features)), feature_block_names,
                                np.mean(delta_accs, axis=1), np.std(delta_accs, axis=1),
                                np.mean(accuracies_wo_feature, axis=1), np.std(accuracies_wo_feature, axis=1)]).T,
                      columns=['feature_block_index', 'feature_name', 'delta_acc_mean', 'delta_acc_std',
                               'acc_with_feature_randomized_mean', 'acc_with_feature_randomized_std'])
    df.iloc[:, 2:] = df.iloc[:, 2:].astype(float)
    df_sorted = df.sort_values('delta_acc_mean', ascending=False)
    df_sorted['delta_acc_mean'] = pd.to_numeric(df_sorted['delta_acc_mean'])
    df_sorted['acc_with_feature_randomized_mean'] = pd.to_numeric(df_sorted['acc_with_feature_randomized_mean'])
    if write_to_file:
        if predictions_outdir == "":
            predictions_outdir = os.path.dirname(weights_pkl)
        if not os.path.exists(predictions_outdir) and predictions_outdir != "":
            os.makedirs(predictions_outdir)
        if fname_stem != "":
            fname_stem = fname_stem + "_"
        out_name = os.path.join(predictions_outdir, fname_stem + 'feature_importance.txt')
        df_sorted.to_csv(out_name, sep='\t', index=False, header=True, float_format='%.6f')
        print("Output saved in: %s" % out_name)
    return df_sorted
