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

    # -- rows split over ranks (npbnn_amd/rowshard.py): the stand-in's chain asks the wrapper for the sums over all rows --
    _sharded = None

    @property
    def lik_kind(self):
        return 0 if self.bnn._estimation_mode == "classification" else 1        # (capi.LIK_CATEGORICAL / LIK_GAUSS)

    @property
    def n_targets(self):
        return 0 if self.bnn._estimation_mode == "classification" else self.bnn._labels.shape[1]

    def train_rows(self):
        return len(self.bnn._data)

    def set_row_shard(self, sharded):
        self._sharded = sharded

    def refresh_row_weights(self, bnn):
        pass                              # (the stand-in reads bnn._class_w at every evaluation)

    def _chain_evaluate(self, *a, **k):
        return (self._sharded.evaluate if self._sharded is not None else self.evaluate)(*a, **k)

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

    _ROWWISE = {2: orc.lik_gaussian_error, 3: orc.lik_poisson, 4: orc.lik_negbin, 5: orc.lik_negbin2d, 6: orc.lik_negbin_base10}

    def evaluate(self, weights, slopes=None, col_override=None, lik_temp=1.0, sigma=None, which=0, want_confusion=False):
        self.n_eval += 1
        x, lab = self._data(which)
        y = self.predict(weights, slopes, col_override, which)
        out = dict(confusion=None, sigma=None, sum_r=None, sum_r2=None, n_rows=len(x))
        from npbnn_amd.likelihoods import likelihood_kind
        kind = likelihood_kind(getattr(self, "_lik_f", None))          # (capi.LIK_*; set by npbnn_amd.sampler.get_backend)
        if self.bnn._estimation_mode == "classification":
            iw = self.bnn._instance_weights if which == 0 else None
            cw = self.bnn._class_w if which == 0 else []
            with np.errstate(divide="ignore"):
                out["loglik"] = orc.lik_categorical(y, lab, np.arange(len(x)), class_weight=cw, instance_weight=iw,
                                                    lik_temp=lik_temp)
            if want_confusion:
                out["confusion"] = orc.confusion_counts(y, lab)
        elif kind in self._ROWWISE:
            out["loglik"] = self._ROWWISE[kind](y, lab, None, lik_temp=lik_temp)
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
                  cur_loglik, cur_logprior, cur_sigma=None, sigma=None, mask=None, n_candidates=0, schedule=0, sigma_mult=None,
                  hastings=None, fixed_slopes=None):
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
            h = 0.0
            if sigma_mult is not None:             # sigma' = current sigma * pre-drawn factors (BNN_env.py:435-442)
                r = self._chain_evaluate(wl, slopes=fixed_slopes, lik_temp=lik_temp, sigma=np.asarray(sig, dtype=float) * sigma_mult[t])
                h = hastings[t]
            else:
                r = self._chain_evaluate(wl, slopes=fixed_slopes, lik_temp=lik_temp, sigma=sigma)
            p = orc.log_prior(wl, prior_kind, prior_scale)
            llp[t], lpp[t] = r["loglik"], p
            if ((r["loglik"] + p) - (ll + lp)) * temperature + h >= log_u[t]:
                cur, ll, lp, acc[t] = prop, r["loglik"], p, 1
                n_acc += 1
                if r["sigma"] is not None:
                    sig = r["sigma"]
        return cur, acc, llp, lpp, dict(loglik=ll, logprior=lp, sigma=sig, n_accepted=n_acc, n_passes=K, n_candidates=1)


    def run_chain_general(self, weights, draws, log_u, mask=None, indicators=None, feature_indicators=None, feature_means=None,
                          prior_ind1=0.5, has_indicator_prior=False, prior_kind=1, prior_scale=None, w_bound=np.inf, temperature=1.0,
                          lik_temp=1.0, cur_loglik=0.0, cur_logprior=0.0, cur_sigma=None, sigma=None, n_candidates=0, schedule=0,
                          sigma_mult=None, hastings=None, fixed_slopes=None):
        """numpy stand-in for npbnn_chain_run_general: every iteration builds the full candidate from the pre-drawn numbers."""
        import scipy.stats
        shapes = [w.shape for w in weights]
        sizes = [int(np.prod(sh)) for sh in shapes]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        cur = np.concatenate([np.asarray(w, dtype=float).ravel() for w in weights])
        m = None if mask is None else np.concatenate([np.asarray(x, dtype=float).ravel() for x in mask])
        ind = None if indicators is None else np.asarray(indicators, dtype=float).ravel().copy()
        find = None if feature_indicators is None else np.asarray(feature_indicators, dtype=float).copy()
        kind = draws["kind"]
        K = len(log_u)
        acc = np.zeros(K, dtype=np.uint8)
        llp, lpp = np.zeros(K), np.zeros(K)
        ll, lp, sig, n_acc = cur_loglik, cur_logprior, cur_sigma, 0

        def layers(v):
            return [v[offs[i]:offs[i + 1]].reshape(shapes[i]) for i in range(len(shapes))]

        for t in range(K):
            prop = cur.copy()
            n = draws["cnt"][t]
            i_t, v_t = draws["idx"][t, :n], draws["val"][t, :n]
            new = v_t if kind == 2 else cur[i_t] + v_t
            if kind != 3:
                new = np.where(new > w_bound, w_bound - (new - w_bound), new)
                new = np.where(new < -w_bound, -w_bound + (-w_bound - new), new)
            prop[i_t] = new
            if kind == 3:
                for li in range(len(shapes)):
                    if (draws["layer_mask"][t] >> li) & 1:
                        lay = prop[offs[li]:offs[li + 1]].reshape(shapes[li])
                        prop[offs[li]:offs[li + 1]] = (lay / np.sum(lay)).ravel()
            if m is not None:
                prop = prop * m
            ind_p, find_p = ind, find
            if ind is not None:
                fl = draws["ind_pos"][draws["ind_ptr"][t]:draws["ind_ptr"][t + 1]]
                ind_p = ind.copy()
                ind_p[fl] = np.abs(ind[fl] - 1)
            override = None
            if find is not None:
                fl = draws["find_pos"][draws["find_ptr"][t]:draws["find_ptr"][t + 1]]
                find_p = find.copy()
                find_p[fl] = np.abs(find[fl] - 1)
                if draws["find_use"][t]:
                    override = np.where(find_p == 0, feature_means, np.nan)
            wl = layers(prop)
            fw = [wl[0] * ind_p.reshape(shapes[0])] + wl[1:] if ind is not None else wl
            h = 0.0
            if kind == 2:
                hn = draws["h_cnt"][t]
                hi, hv, hf = draws["h_idx"][t, :hn], draws["h_val"][t, :hn], draws["h_fac"][t, :hn]
                dd = np.sqrt(0.5 / hf)
                h = np.sum(scipy.stats.norm.logpdf(cur[hi], 0, dd) - scipy.stats.norm.logpdf(hv, 0, dd))
            if sigma_mult is not None:
                r = self.evaluate(fw, slopes=fixed_slopes, col_override=override, lik_temp=lik_temp, sigma=np.asarray(sig, dtype=float) * sigma_mult[t])
                h += hastings[t]
            else:
                r = self.evaluate(fw, slopes=fixed_slopes, col_override=override, lik_temp=lik_temp, sigma=sigma)
            p = orc.log_prior(wl, prior_kind, prior_scale)
            if has_indicator_prior and ind is not None:
                n_on = np.sum(ind_p)
                p = p + (n_on * np.log(prior_ind1) + (ind_p.size - n_on) * np.log(1 - prior_ind1))
            llp[t], lpp[t] = r["loglik"], p
            if ((r["loglik"] + p) - (ll + lp)) * temperature + h >= log_u[t]:
                cur, ll, lp, acc[t], ind, find = prop, r["loglik"], p, 1, ind_p, find_p
                n_acc += 1
                if r["sigma"] is not None:
                    sig = r["sigma"]
        return cur, ind, find, acc, llp, lpp, dict(loglik=ll, logprior=lp, sigma=sig, n_accepted=n_acc, n_passes=K, n_candidates=1)


class OracleExchangeBackend(OracleChainBackend):
    """Adds a numpy stand-in for npbnn_chains_run_exchange (swap intervals with the temperature swaps between them) so
    that the exchange driver (npbnn_amd/exchange.py) and MC3's logging from saved cold-chain states run on CPU, over gloo
    too.  Semantics of the device entry point: records = [logPost, temperature, finished flag, iterations done] per chain and
    exchange; a chain that falls short of an interval (``starve = {(chain_id, interval): iterations it manages}``) makes
    every chain stop at that exchange."""
    starve = {}
    exchange_slack = 1.5
    exchange_slack_floor = 1.15

    def exchange_job(self, weights, chain_id, idx, delta, cnt, log_u, mask, cfg):
        return dict(be=self, chain_id=chain_id, weights=weights, idx=idx, delta=delta, cnt=cnt, log_u=log_u, mask=mask, cfg=cfg)

    @staticmethod
    def run_exchange(jobs, n_chains, seg_len, n_seg, swap_j, swap_k, swap_logu, comm=None, launch_slack=1.25, want_cold_w=True):
        world = 1 if comm is None else comm.world_size
        rank = 0 if comm is None else comm.rank
        per_rank = len(jobs)
        assert n_chains == world * per_rank
        K = seg_len * n_seg
        st = []
        for job in jobs:
            cfg = dict(job["cfg"])
            st.append(dict(w=[np.array(w, dtype=float) for w in job["weights"]], ll=cfg["cur_loglik"], lp=cfg["cur_logprior"],
                           temp=cfg["temperature"], t=0, acc=np.zeros(K, dtype=np.uint8), llp=np.zeros(K), lpp=np.zeros(K), n_acc=0,
                           sigma=cfg.get("cur_sigma"), state=np.zeros((n_seg, 20)),
                           cold=np.zeros((n_seg, sum(w.size for w in job["weights"]))) if want_cold_w else None, cfg=cfg))
        records = np.zeros((n_seg, n_chains, 4))
        done, poisoned = 0, False
        for s in range(n_seg):
            mine = np.zeros((per_rank, 4))
            for q, (job, c) in enumerate(zip(jobs, st)):
                end = (s + 1) * seg_len
                if not poisoned:
                    upto = min(end, s * seg_len + OracleExchangeBackend.starve.get((job["chain_id"], s), seg_len))
                    if upto > c["t"]:
                        a, b = c["t"], upto
                        kw = dict(c["cfg"])
                        kw.update(temperature=c["temp"], cur_loglik=c["ll"], cur_logprior=c["lp"], cur_sigma=c["sigma"])
                        kw.pop("n_candidates", None), kw.pop("schedule", None)
                        if kw.get("sigma_mult") is not None:
                            kw["sigma_mult"], kw["hastings"] = kw["sigma_mult"][a:b], kw["hastings"][a:b]
                        cur, acc, llp, lpp, res = job["be"].run_chain(c["w"], idx=job["idx"][a:b], delta=job["delta"][a:b],
                                                                      cnt=job["cnt"][a:b], log_u=job["log_u"][a:b],
                                                                      mask=job["mask"], **kw)
                        c["acc"][a:b], c["llp"][a:b], c["lpp"][a:b] = acc, llp, lpp
                        if res["n_accepted"] > 0:
                            off, layers = 0, []
                            for w in c["w"]:
                                layers.append(cur[off:off + w.size].reshape(w.shape))
                                off += w.size
                            c["w"], c["ll"], c["lp"], c["sigma"] = layers, res["loglik"], res["logprior"], res["sigma"]
                            c["n_acc"] += res["n_accepted"]
                        c["t"] = b
                mine[q] = (c["ll"] + c["lp"], c["temp"], 1.0 if (not poisoned and c["t"] >= end) else 0.0, c["t"])
            allv = mine.reshape(1, per_rank, 4) if world == 1 else comm.allgather_f64(mine.ravel()).reshape(world, per_rank, 4)
            for i in range(n_chains):
                records[s, i] = allv[i % world, i // world]
            if poisoned or not np.all(records[s, :, 2] == 1.0):
                poisoned = True
                continue
            j, k = int(swap_j[s]), int(swap_k[s])
            pj, tj, pk, tk = records[s, j, 0], records[s, j, 1], records[s, k, 0], records[s, k, 1]
            if j != k and (pk - pj) * tj + (pj - pk) * tk >= swap_logu[s]:
                for job, c in zip(jobs, st):
                    if job["chain_id"] == j:
                        c["temp"] = tk
                    elif job["chain_id"] == k:
                        c["temp"] = tj
            for c in st:
                c["state"][s, :4] = (c["ll"], c["lp"], c["temp"], c["t"])
                if c["sigma"] is not None:
                    sg = np.ravel(np.asarray(c["sigma"], dtype=float))
                    c["state"][s, 4:4 + len(sg)] = sg
                if c["cold"] is not None and c["temp"] == 1.0:
                    c["cold"][s] = np.concatenate([w.ravel() for w in c["w"]])
            done = s + 1
        outs = []
        for c in st:
            outs.append(dict(w=np.concatenate([w.ravel() for w in c["w"]]), accepted=c["acc"], loglik_prop=c["llp"],
                             logprior_prop=c["lpp"], state=c["state"], cold_w=c["cold"],
                             result=dict(loglik=c["ll"], logprior=c["lp"], sigma=c["sigma"], n_accepted=c["n_acc"], n_passes=c["t"],
                                         n_candidates=1, n_void_passes=0, schedule=1, temperature=c["temp"],
                                         iterations_done=c["t"], overflow=0)))
        return outs, records, done


def serve_from_oracle(factory):
    """Point the package's backend seam (npbnn_amd.sampler._make_backend) at ``factory(bnn)`` - an oracle-backed stand-in - so
    that samplers built from now on run without a GPU.  tests/conftest.py puts the real builder back after every test."""
    from npbnn_amd import sampler
    sampler._make_backend = lambda bnn, likelihood_f: factory(bnn)
