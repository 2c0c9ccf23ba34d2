"""TEST-ONLY evaluation backend: serves the product's sampler from the CPU oracle so that the host
logic (proposal stream, prior, accept/reject, adaptation, MC3 swaps) can be exercised without a GPU.
Lives under tests/ on purpose - the product ships the HIP backend only."""
import numpy as np

import oracle as orc


class OracleBackend:
    fused_likelihood = True

    def __init__(self, bnn, out_kind=0):
        self.bnn = bnn
        self.out_kind = out_kind
        self.n_eval = 0

    def _act(self, slopes):
        fun = self.bnn._act_fun._function
        trainable = self.bnn._act_fun._trainable
        return orc.Act(fun, prm=slopes if slopes is not None else np.zeros(8), trainable=trainable)

    def _data(self, which):
        return (self.bnn._data, self.bnn._labels) if which == 0 else (self.bnn._test_data, self.bnn._test_labels)

    def _out_fn(self):
        return {0: orc.out_softmax, 1: orc.out_identity, 2: orc.out_regress_error}[self.out_kind]

    def predict(self, weights, slopes=None, col_override=None, which=0, apply_out_fn=True):
        x, _ = self._data(which)
        co = None
        if col_override is not None:
            co = ((~np.isnan(col_override)).astype(int) * 0 + np.isnan(col_override).astype(int), np.nan_to_num(col_override))
        z = orc.forward_logits(x, weights, self._act(slopes), col_override=co)
        return self._out_fn()(z) if apply_out_fn else z

    def evaluate(self, weights, slopes=None, col_override=None, lik_temp=1.0, sigma=None, which=0, want_confusion=False):
        self.n_eval += 1
        x, lab = self._data(which)
        y = self.predict(weights, slopes, col_override, which)
        out = dict(confusion=None, sigma=None, sum_r=None, sum_r2=None, n_rows=len(x))
        if self.bnn._estimation_mode == "classification":
            iw = self.bnn._instance_weights if which == 0 else None
            cw = self.bnn._class_w if which == 0 else []
            with np.errstate(divide="ignore"):
                out["loglik"] = orc.lik_categorical(y, lab, np.arange(len(x)), class_weight=cw, instance_weight=iw,
                                                    lik_temp=lik_temp)
            if want_confusion:
                out["confusion"] = orc.confusion_counts(y, lab)
        else:
            r = lab - y[:, :lab.shape[1]]
            sig = np.std(y - lab, axis=0) if sigma is None else np.asarray(sigma, dtype=float)
            out["loglik"] = orc.lik_gaussian(y, lab, None, lik_temp=lik_temp, sig2=sig)
            out["sigma"] = sig
            out["sum_r"] = r.sum(axis=0)
            out["sum_r2"] = (r * r).sum(axis=0)
        return out


class OracleChainBackend(OracleBackend):
    """Adds a float64 numpy stand-in for npbnn_chain_run so that MCMC.run_steps (batching, pre-draw pipeline,
    bookkeeping) can be tested on CPU against the plain mh_step loop."""

    def run_chain(self, weights, idx, delta, cnt, log_u, prior_kind, prior_scale, w_bound, temperature, lik_temp,
                  cur_loglik, cur_logprior, cur_sigma=None, sigma=None, mask=None, n_candidates=0, schedule=0):
        shapes = [w.shape for w in weights]
        cur = np.concatenate([np.asarray(w, dtype=float).ravel() for w in weights])
        m = None if mask is None else np.concatenate([np.asarray(x, dtype=float).ravel() for x in mask])

        def unpack(v):
            out, off = [], 0
            for s in shapes:
                n = int(np.prod(s))
                out.append(v[off:off + n].reshape(s))
                off += n
            return out

        ll, lp = cur_loglik, cur_logprior
        sig = cur_sigma
        K = len(cnt)
        acc = np.zeros(K, dtype=np.uint8)
        llp, lpp = np.zeros(K), np.zeros(K)
        n_acc = 0
        for t in range(K):
            prop = cur.copy()
            sel = idx[t, :cnt[t]]
            ok = sel >= 0
            v = cur[sel[ok]] + delta[t, :cnt[t]][ok]
            v = np.where(v > w_bound, w_bound - (v - w_bound), v)
            v = np.where(v < -w_bound, -w_bound + (-w_bound - v), v)
            prop[sel[ok]] = v
            if m is not None:
                prop = prop * m
            wl = unpack(prop)
            r = self.evaluate(wl, lik_temp=lik_temp, sigma=sigma)
            p = orc.log_prior(wl, prior_kind, prior_scale)
            llp[t], lpp[t] = r["loglik"], p
            if ((r["loglik"] + p) - (ll + lp)) * temperature >= log_u[t]:
                cur, ll, lp, acc[t] = prop, r["loglik"], p, 1
                n_acc += 1
                if r["sigma"] is not None:
                    sig = r["sigma"]
        return cur, acc, llp, lpp, dict(loglik=ll, logprior=lp, sigma=sig, n_accepted=n_acc, n_passes=K, n_candidates=1)
