"""Partial dependence of the posterior prediction on a focal feature (reference: np_bnn/BNN_pdp.py:14-108).

For every step of the focal feature's gradient the whole data matrix is evaluated under every stored weight sample; the
reference does that with one ``RunPredict`` per (step, sample).  Here each step is one upload of the modified matrix
and one ``npbnn_predict_sets`` over all samples."""
import numpy as np

from .files import load_obj
from .model import data_transform_obj
from .posterior import _SamplePredictor


def get_feature_summary(data, focal_features):
    """Per focal feature: 1 for binary/ordinal (consecutive integers) else 0, its minimum and maximum
    (reference: BNN_pdp.py:14-27)."""
    num_features = len(focal_features)
    feature_summary = np.zeros((3, num_features))
    for i in range(num_features):
        values = np.unique(data[:, focal_features[i]])
        feature_summary[1, i] = np.nanmin(values)
        feature_summary[2, i] = np.nanmax(values)
        values_range = np.arange(feature_summary[1, i], feature_summary[2, i] + 1)
        feature_summary[0, i] = np.all(np.isin(values, values_range))
    return feature_summary


def make_pdp_features(data, focal_features, steps_continuous=100):
    """The gradient of feature values along which the partial dependence is computed (reference: BNN_pdp.py:30-46)."""
    feature_summary = get_feature_summary(data, focal_features)
    if np.sum(feature_summary[0, :] == 0) and len(focal_features) == 1:        # single continuous feature
        pdp_feat = np.linspace(feature_summary[1, 0], feature_summary[2, 0], num=steps_continuous).reshape(steps_continuous, 1)
    elif feature_summary[0, 0] == 1 and len(focal_features) == 1:              # ordinal or binary
        M = int(feature_summary[2, 0])
        pdp_feat = np.linspace(feature_summary[1, 0], M, num=M + 1).reshape((M + 1, 1))
    else:                                                                      # one-hot encoded
        pdp_feat = np.eye(feature_summary.shape[1])
    return pdp_feat


def get_pdp(data, focal_features, estimation_mode, size_output, actFun, output_act_fun, weights, alphas, data_transform):
    """Mean and 95 % interval of the prediction (cumulative class probabilities for classification) with the focal
    feature(s) set to every value of their gradient (reference: BNN_pdp.py:49-84)."""
    pdp_features = make_pdp_features(data, focal_features)
    num_pdp_steps = pdp_features.shape[0]
    pdp = np.zeros((num_pdp_steps, size_output, 3))
    samples = [dict(weights=w, alphas=a) for w, a in zip(weights, alphas)]
    if len(alphas):
        actFun.reset_prm(alphas[-1])                  # the reference leaves the last sample's slopes installed
    predictor = _SamplePredictor(np.asarray(data).shape[1], samples, actFun, output_act_fun)
    try:
        for n in range(num_pdp_steps):
            feat = np.copy(data)
            feat[:, focal_features] = pdp_features[n, :]
            if data_transform is not None:
                feat = data_transform.transform(feat)
            pred = predictor.predict(feat)
            if estimation_mode == 'classification':
                pred = np.cumsum(pred, axis=2)
            pdp[n, :, 0] = np.mean(pred, axis=(0, 1))
            probs_quantiles = np.quantile(np.mean(pred, axis=0), q=(0.025, 0.975), axis=0)
            pdp[n, :, 1] = probs_quantiles[0, :]
            pdp[n, :, 2] = probs_quantiles[1, :]
    finally:
        predictor.close()
    return {'feature': pdp_features, 'pdp': pdp}


def pdp(pickle_file, pdp_features):
    """Partial dependence for each entry of ``pdp_features`` (lists of focal feature indices) from a saved run
    (reference: BNN_pdp.py:87-108)."""
    bnn_obj, mcmc_obj, logger_obj = load_obj(pickle_file)
    post_samples = logger_obj._post_weight_samples
    post_weights = [s['weights'] for s in post_samples]
    post_alphas = [s['alphas'] for s in post_samples]
    data_transform = None
    if bnn_obj._feature_indicators is not None:
        data_transform = data_transform_obj(bnn_obj._feature_indicators, bnn_obj._feature_means)
    return [get_pdp(bnn_obj._data, p, bnn_obj._estimation_mode, bnn_obj._size_output, bnn_obj._act_fun,
                    bnn_obj._output_act_fun, post_weights, post_alphas, data_transform) for p in pdp_features]
