"""``postLogger`` — TSV log of the sampled states and the posterior weight samples.

Same files and layouts as the reference logger (np_bnn/BNN_env.py:553-658, header built in
np_bnn/BNN_files.py:110-187): ``<name>_l<nodes>.log`` (tab separated; any tool that reads upstream's
log reads this one), optional ``..._W.log`` with every weight, and ``<name>_l<nodes>.pkl`` holding
``[bnn_obj, mcmc_obj, logger]`` with the same ``_post_weight_samples`` dictionaries.  By default the pickle
holds objects of THIS package (``npbnn_amd.model.npBNN`` ...) without the feature matrices; with
``postLogger(export="upstream")`` it is written in upstream's own format (npbnn_amd/export.py): ``np_bnn.load_obj``
opens it as np_bnn objects and ``np_bnn.predictBNN`` / ``get_posterior_est`` / ``npBNN(pickle_file=...)`` consume it.
``load_obj`` of this package reads both.
"""
import csv
import os

import numpy as np

from .files import DetachedMatrix, SaveObject, data_side_file


def init_output_files(bnn_obj, filename="bnn", sample_from_prior=0, outpath="", add_prms=None,
                      continue_logfile=False, log_all_weights=0):
    """Create the log file(s) with their header row; returns (logfile, weights file or None, pickle file)."""
    outdir = os.path.dirname(filename)
    if len(outdir) > 0 and not os.path.exists(outdir):
        os.makedirs(outdir)
    outname = "%s_l%s" % (filename, "_".join(map(str, bnn_obj._n_nodes)))
    logfile_name = os.path.join(outpath, outname + ".log")
    w_file_name = os.path.join(outpath, outname + "_W.log") if log_all_weights else None

    head = ["it", "posterior", "likelihood", "prior"]
    if bnn_obj._estimation_mode == "classification":
        head += ["accuracy", "test_accuracy"] + ["acc_C%s" % i for i in range(bnn_obj._n_output_prm)]
    elif bnn_obj._estimation_mode == "custom":
        head += ["MSE", "test_MSE"]
    else:
        head += ["MSE", "test_MSE"] + ["MSE_prm%s" % i for i in range(bnn_obj._n_output_prm)]
    head_w = ["it"]
    for i in range(bnn_obj._n_layers):
        head += ["mean_w%s" % i, "std_w%s" % i]
        if bnn_obj._hyper_p:
            head.append("prior_std_w%s" % i if bnn_obj._hyper_p == 1 else "mean_prior_std_w%s" % i)
        if log_all_weights:
            head_w += ["w_%s_%s" % (i, j) for j in range(bnn_obj._w_layers[i].size)]
    if bnn_obj._freq_indicator:
        head.append("mean_ind")
    if add_prms:
        head = head + add_prms
    if bnn_obj._act_fun._trainable:
        head += ["alpha_%s" % i for i in range(bnn_obj._n_layers - 1)]
    if len(bnn_obj._error_prm):
        head += ["sig_%s" % i for i in range(len(bnn_obj._error_prm))]
    if bnn_obj._feature_indicators is not None:
        head += ['feature_ind_%s' % i for i in range(bnn_obj._n_features)]
    head += ["acc_prob", "mcmc_id"]

    if not continue_logfile:
        with open(logfile_name, "w", newline='') as f:
            csv.writer(f, delimiter='\t').writerow(head)
    if log_all_weights:
        with open(w_file_name, "w", newline='') as f:
            csv.writer(f, delimiter='\t').writerow(head_w)
    return logfile_name, w_file_name, os.path.join(outpath, outname + ".pkl")


class postLogger():
    def __init__(self, bnn_obj, filename="BNN", wdir="", sample_from_prior=0, add_prms=None,
                 continue_logfile=False, log_all_weights=0, pickle_data=False, export=None):
        """``pickle_data``: True writes the feature matrices into the checkpoint with every posterior sample, as the reference
        does (np_bnn/BNN_env.py:655-658; > 200 MB per sample at 100k x 256); the default writes them ONCE into
        ``<checkpoint>_data.npz`` and keeps the checkpoint itself to weights, sampler state and posterior samples -
        ``load_obj`` puts them back together.  ``export="upstream"``: the checkpoint in upstream's own pickle format (with
        the feature matrices, as upstream writes it), readable by np_bnn's tools without this package."""
        if export not in (None, "upstream"):
            raise ValueError("export=%r; expected None or 'upstream'" % (export,))
        self._export = export
        self._logfile, self._w_file, self._pklfile = init_output_files(
            bnn_obj, filename, sample_from_prior, outpath=wdir, add_prms=add_prms,
            continue_logfile=continue_logfile, log_all_weights=log_all_weights)
        self._log_all_weights = log_all_weights
        self._post_weight_samples = []
        self._estimation_mode = bnn_obj._estimation_mode
        self._pickle_data = bool(pickle_data)
        self._side_key = None

    _DETACHED = ("_data", "_test_data")

    def _light_views(self, bnn_obj, mcmc_obj):
        """(model, sampler) as the checkpoint stores them: the feature matrices replaced by place holders (their side file written
        when the matrices are not the ones it holds) and the sampler without its prediction matrices (functions of the weights:
        computed on demand after loading, as for a live chain)."""
        big = {name: getattr(bnn_obj, name) for name in self._DETACHED if np.size(getattr(bnn_obj, name, ())) > 0}
        if not big:
            return bnn_obj, mcmc_obj
        key = tuple((name, id(a), np.shape(a)) for name, a in big.items())
        side = data_side_file(self._pklfile)
        if key != self._side_key or not os.path.exists(side):
            np.savez(side, **{name.lstrip("_"): np.asarray(a) for name, a in big.items()})
            self._side_key = key
        bnn_v = bnn_obj.__class__.__new__(bnn_obj.__class__)
        bnn_v.__dict__.update(bnn_obj.__dict__)
        for name, a in big.items():
            setattr(bnn_v, name, DetachedMatrix(name.lstrip("_"), a))
        mcmc_v = mcmc_obj
        if hasattr(mcmc_obj, "_light_view"):
            mcmc_v = mcmc_obj._light_view(bnn_v)
        return bnn_v, mcmc_v

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_side_key"] = None        # (object identities mean nothing in another process)
        return state

    def update_post_weight_samples(self, row):
        self._post_weight_samples += [row]

    def replace_post_weight_samples(self, post_weight_samples):
        self._post_weight_samples = post_weight_samples

    def control_weight_sample_length(self, maxlength):
        if len(self._post_weight_samples) > maxlength:
            self._post_weight_samples = self._post_weight_samples[-maxlength:]

    def log_sample(self, bnn_obj, mcmc_obj, add_prms=None):
        """Append one row to the TSV log (reference: BNN_env.py:585-620)."""
        row = [mcmc_obj._current_iteration, mcmc_obj._logPost, mcmc_obj._logLik, mcmc_obj._logPrior,
               mcmc_obj._accuracy, mcmc_obj._test_accuracy]
        if self._estimation_mode != "custom":
            row += list(mcmc_obj._label_acc)
        for i in range(bnn_obj._n_layers):
            row += [np.mean(bnn_obj._w_layers[i]), np.std(bnn_obj._w_layers[i])]
            if bnn_obj._hyper_p:
                row.append(bnn_obj._prior_scale[i] if bnn_obj._hyper_p == 1 else np.mean(bnn_obj._prior_scale[i]))
        if bnn_obj._freq_indicator > 0:
            row.append(np.mean(bnn_obj._indicators))
        if add_prms:
            row = row + add_prms
        if bnn_obj._act_fun._trainable:
            row += list(bnn_obj._act_fun._acc_prm)
        if self._estimation_mode == "regression":
            row += list(bnn_obj._error_prm)
        if bnn_obj._feature_indicators is not None:
            row += list(bnn_obj._feature_indicators)
        row += [mcmc_obj._acceptance_rate, mcmc_obj._mcmc_id]
        with open(self._logfile, "a", newline='') as f:
            csv.writer(f, delimiter='\t').writerow(row)
            f.flush()

    def log_weights(self, bnn_obj, mcmc_obj, add_prms=None, add_obj=None, save_pickle=True):
        """Keep the posterior weight sample (ring of n_post_samples) and rewrite the pickle, or append every
        weight to the ``_W.log`` file (reference: BNN_env.py:622-658).  ``save_pickle=False`` skips the rewrite of the
        pickle file - for callers that log several samples in a row and know the next call overwrites it anyway."""
        if self._log_all_weights:
            row = [mcmc_obj._current_iteration] + list((bnn_obj._w_layers[0] * bnn_obj._indicators[0]).flatten())
            for i in range(1, bnn_obj._n_layers):
                row += list(bnn_obj._w_layers[i].flatten())
            with open(self._w_file, "a", newline='') as f:
                csv.writer(f, delimiter='\t').writerow(row)
                f.flush()
        else:
            if bnn_obj._freq_indicator:
                weights = [bnn_obj._w_layers[0] * bnn_obj._indicators] + list(bnn_obj._w_layers[1:])
            else:
                weights = bnn_obj._w_layers
            sample = {'weights': weights, 'alphas': list(bnn_obj._act_fun._acc_prm),
                      'mcmc_it': mcmc_obj._current_iteration}
            if len(bnn_obj._error_prm):
                sample['error_prm'] = list(bnn_obj._error_prm)
            if add_prms:
                sample['additional_prm'] = list(add_prms)
            self.update_post_weight_samples(sample)
            self.control_weight_sample_length(mcmc_obj._n_post_samples)
        if save_pickle and getattr(self, "_export", None) == "upstream":
            from .export import save_upstream
            save_upstream([bnn_obj, mcmc_obj, self] + ([add_obj] if add_obj else []), self._pklfile)
        elif save_pickle:
            if not getattr(self, "_pickle_data", True):
                bnn_obj, mcmc_obj = self._light_views(bnn_obj, mcmc_obj)
            objs = [bnn_obj, mcmc_obj, self] + ([add_obj] if add_obj else [])
            SaveObject(objs, self._pklfile)
