"""Helpers for bench.py: the BASELINE.json workloads (synthetic data, model and sampler through the reference's own call
sequence, the matching oracle chain for the CPU baseline, the parity read-out) and the exchange self-check."""
import contextlib
import io

import numpy as np

import npbnn_amd as bn


def _quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


class Workload:
    """One BASELINE.json configuration: SURVEY.md 8(d) gives inputs (``np.random.default_rng(0)``), model
    (``np.random.seed(1234)`` before npBNN) and the algorithmic bytes per proposal (X in float32 + labels / targets)."""
    config = 0
    moving_update_f = None      # proposal sizes at which about a quarter of the proposals is accepted (bench.py's moving_chain leg)

    def kernel_name(self, ctx, cand):
        return "eval_kernel<MT0=%d,MTI=1,%s,D=%d,LK=%s,%s>" % (self.mt0, "fp16-split" if ctx.l0_mode() == "f16-split" else "f32", cand,
                                                              self.lik_class, "fast" if ctx.info(bn._capi.INFO_FAST_TAILS) else "general")


class Config2(Workload):
    config, mt0, lik_class = 2, 2, "categorical"
    n, f, c, hidden = 100_000, 256, 10, [32, 8]
    short = "config 2 (100k x 256, [32,8])"
    moving_update_f = [0.004] * 3
    description = "config 2: 100k x 256 features, 10 classes, hidden [32,8], tanh, bias 2"

    def __init__(self):
        rs = np.random.default_rng(0)
        self.x = rs.standard_normal((self.n, self.f))
        self.y = rs.integers(0, self.c, self.n)
        self.bytes_per_proposal = 4.0 * self.n * self.f + 4.0 * self.n

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, rows=None, **sampler_kw):
        """np.random.seed(1234); npBNN(n_nodes=[32,8], tanh, bias nodes in input+hidden layers, N(0,1) prior); MCMC defaults
        (update_f 0.05 -> update_n [411,13,4]).  rows = (lo, hi): this rank's share of the rows (with MCMC's row_comm=)."""
        lo, hi = rows if rows is not None else (0, self.n)
        x32 = self.x[lo:hi].astype(np.float32)
        dat = dict(data=x32, labels=self.y[lo:hi], test_data=np.zeros((0, self.f)), test_labels=np.zeros(0))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat, n_nodes=self.hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        mcmc = bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **sampler_kw)
        return bnn, mcmc

    def oracle_chain(self, orc):
        np.random.seed(1234)
        return orc.make_chain(self.x, self.y, self.hidden, act=orc.Act("tanh"), use_bias_node=2, prior_kind=1, p_scale=1)

    def parity(self, bnn, mcmc):
        """State of the timed chain against the float64 oracle: relative error of the device log-likelihood of the chain's current
        weights and max abs error of its class probabilities on a row sample."""
        import oracle as orc
        w = [np.array(v, dtype=np.float64) for v in bnn._w_layers]
        xs = self.x.astype(np.float32).astype(np.float64)
        pred = orc.forward(xs, w, orc.Act("tanh"), orc.out_softmax)
        ll = orc.lik_categorical(pred, self.y, np.arange(self.n))
        dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"]
        rows = np.arange(0, self.n, 97)
        y_dev = np.asarray(mcmc._y)[rows]
        return dict(loglik_oracle=float(ll), loglik_device=float(dev), loglik_rel_err=float(abs(dev - ll) / abs(ll)),
                    chain_loglik_rel_err=float(abs(mcmc._logLik - ll) / abs(ll)),
                    prediction_max_abs_err=float(np.max(np.abs(y_dev - pred[rows]))), tolerance="1e-4 relative (BASELINE.json)")


class Config2LogNormal(Config2):
    """Config 2's shapes and model on heavy-tailed features: log-normal columns, sigma = 3 (largest entry ~1e5 x the median).  Round 4's
    `auto` sent such data to the float32 layer 0 at half the speed; the columns' fp16 scales are moved up instead."""
    config = 12
    short = "config 2 on log-normal features"
    moving_update_f = None
    description = "config 2's shapes and model on log-normal features (exp(3 z), z ~ N(0,1)): 100k x 256, 10 classes, hidden [32,8], tanh, bias 2"

    def __init__(self):
        rs = np.random.default_rng(0)
        self.x = np.exp(3.0 * rs.standard_normal((self.n, self.f)))
        self.y = rs.integers(0, self.c, self.n)
        self.bytes_per_proposal = 4.0 * self.n * self.f + 4.0 * self.n

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, rows=None, **sampler_kw):
        """Config 2's build with the first-layer weights in units of the columns - divided by each column's mean |value|, as they would
        be after training on such data (the reference's initialiser knows nothing of the features' scale) - before the sampler
        evaluates them."""
        x32 = self.x.astype(np.float32)
        dat = dict(data=x32, labels=self.y, test_data=np.zeros((0, self.f)), test_labels=np.zeros(0))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat, n_nodes=self.hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        bnn._w_layers[0][:, 1:] /= np.abs(x32).mean(axis=0, dtype=np.float64)
        return bnn, bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **sampler_kw)


class _Regression(Workload):
    lik_class = "gaussian"

    def _targets(self, rs, x, k):
        """tanh-MLP teacher + 0.5 * noise."""
        w1 = rs.standard_normal((x.shape[1], 8)) / np.sqrt(x.shape[1])
        w2 = rs.standard_normal((8, k))
        return np.tanh(x @ w1) @ w2 + 0.5 * rs.standard_normal((x.shape[0], k))

    def parity(self, bnn, mcmc):
        import oracle as orc
        w = [np.array(v, dtype=np.float64) for v in bnn._w_layers]
        xs = self.x.astype(np.float32).astype(np.float64)
        pred = orc.forward(xs, w, orc.Act(self.act), orc.out_identity)
        if bnn._empirical_error:
            ll = orc.closed_gaussian_empirical(pred, self.y)[0]
            dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"]
        else:
            sig = np.ones(self.k) * bnn._error_prm
            ll = orc.lik_gaussian(pred, self.y, sig2=sig)
            dev = mcmc._backend.evaluate(bnn._w_layers, None, sigma=sig)["loglik"]
        rows = np.arange(0, self.n, 997)
        y_dev = np.asarray(mcmc._y)[rows]
        return dict(loglik_oracle=float(ll), loglik_device=float(dev), loglik_rel_err=float(abs(dev - ll) / abs(ll)),
                    chain_loglik_rel_err=float(abs(mcmc._logLik - ll) / abs(ll)),
                    prediction_max_abs_err=float(np.max(np.abs(y_dev - pred[rows]))), tolerance="1e-4 relative (BASELINE.json)")


class Config4(_Regression):
    config, mt0, act = 4, 1, "tanh"
    n, f, k, hidden = 1_000_000, 64, 2, [16, 4]
    short = "config 4 (1M x 64 regression, [16,4])"
    description = "config 4: 1M x 64 features, 2 Gaussian targets, hidden [16,4], tanh, bias 2, empirical sigma (bnn_regress.py settings)"
    sampler = dict(update_ws=[0.025, 0.025, 0.05], update_f=[0.005, 0.005, 0.05], n_iteration=20000, adapt_f=0.3, estimate_error=False)

    def __init__(self):
        rs = np.random.default_rng(0)
        self.x = rs.standard_normal((self.n, self.f))
        self.y = self._targets(rs, self.x, self.k)
        self.bytes_per_proposal = 4.0 * self.n * self.f + 4.0 * self.n * self.k

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, rows=None, **sampler_kw):
        """bnn_regress.py:33-52: npBNN(regression, tanh, p_scale 1, bias 2, empirical_error); MCMC(update_ws [.025,.025,.05],
        update_f [.005,.005,.05], adapt_f .3, estimate_error False).  rows = (lo, hi): this rank's share of the rows (row_comm=)."""
        lo, hi = rows if rows is not None else (0, self.n)
        dat = dict(data=self.x[lo:hi].astype(np.float32), labels=self.y[lo:hi], test_data=np.zeros((0, self.f)),
                   test_labels=np.zeros((0, self.k)))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat, n_nodes=self.hidden, estimation_mode="regression", actFun=bn.ActFun(fun="tanh"), p_scale=1,
                     use_bias_node=2, empirical_error=True)
        mcmc = bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **dict(self.sampler, **sampler_kw))
        return bnn, mcmc

    def oracle_chain(self, orc):
        np.random.seed(1234)
        return orc.make_chain(self.x, self.y, self.hidden, act=orc.Act("tanh"), use_bias_node=2, prior_kind=1, p_scale=1,
                              mode="regression", empirical_error=True, **self.sampler)


class Config5(_Regression):
    config, mt0, act = 5, 2, "ReLU"
    n, f, k, hidden, blocks = 50_000, 512, 1, [32, 8], 8
    short = "config 5 (50k x 512 block layers, [32,8])"
    moving_update_f = [0.01] * 3
    description = ("config 5: 50k x 512 features, layer 0 in 8 blocks of 64 inputs x 4 nodes (create_mask), hidden [32,8], ReLU, "
                   "bias -1, 1 Gaussian target (block_bnns.py layout)")

    def __init__(self):
        rs = np.random.default_rng(0)
        self.x = rs.standard_normal((self.n, self.f))
        self.y = self._targets(rs, self.x, self.k)
        self.bytes_per_proposal = 4.0 * self.n * self.f + 4.0 * self.n * self.k

    def _mask_args(self):
        return [list(np.repeat(np.arange(self.blocks), self.f // self.blocks)), [], []], [[4] * self.blocks, [], []]

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, **sampler_kw):
        """block_bnns.py:32-43: npBNN(regression, default ReLU, p_scale 1, bias on the last layer) + create_mask / apply_mask;
        MCMC defaults."""
        dat = dict(data=self.x.astype(np.float32), labels=self.y, test_data=np.zeros((0, self.f)), test_labels=np.zeros((0, self.k)))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat, n_nodes=self.hidden, estimation_mode="regression", p_scale=1, use_bias_node=-1)
        idx, per = self._mask_args()
        _quiet(bnn.apply_mask, bn.create_mask(bnn._w_layers, indx_input_list=idx, nodes_per_feature_list=per))
        mcmc = bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **sampler_kw)
        return bnn, mcmc

    def oracle_chain(self, orc):
        np.random.seed(1234)
        w = orc.init_weights(self.hidden, self.f, self.k, init_std=0.1, bias_node=-1)
        idx, per = self._mask_args()
        return orc.make_chain(self.x, self.y, self.hidden, act=orc.Act("ReLU"), use_bias_node=-1, prior_kind=1, p_scale=1,
                              mode="regression", init_w=w, mask=orc.block_mask(w, idx, per))


class DefaultNetwork(Config2):
    """Config 2's data under the reference's DEFAULT model: npBNN(n_nodes=[50, 5]), ActFun() = ReLU, use_bias_node=1
    (np_bnn/BNN_env.py:20-23), MCMC defaults.  Layer 0 has 50 nodes = four 16-unit output tiles (MT0 = 4): one 70-KB weight image
    per candidate, so ONE candidate per pass fits a compute unit's LDS beside the X rings."""
    config, mt0 = 0, 4
    hidden = [50, 5]
    short = "default network (100k x 256, [50,5])"
    moving_update_f = None
    description = "config-2 data, the reference's default network: hidden [50,5], ReLU, bias 1"

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, rows=None, **sampler_kw):
        lo, hi = rows if rows is not None else (0, self.n)
        dat = dict(data=self.x[lo:hi].astype(np.float32), labels=self.y[lo:hi], test_data=np.zeros((0, self.f)), test_labels=np.zeros(0))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat)
        mcmc = bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **sampler_kw)
        return bnn, mcmc

    def oracle_chain(self, orc):
        np.random.seed(1234)
        return orc.make_chain(self.x, self.y, self.hidden, act=orc.Act("ReLU"), use_bias_node=1, prior_kind=1, p_scale=1)

    def parity(self, bnn, mcmc):
        import oracle as orc
        w = [np.array(v, dtype=np.float64) for v in bnn._w_layers]
        xs = self.x.astype(np.float32).astype(np.float64)
        pred = orc.forward(xs, w, orc.Act("ReLU"), orc.out_softmax)
        ll = orc.lik_categorical(pred, self.y, np.arange(self.n))
        dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"]
        return dict(loglik_oracle=float(ll), loglik_device=float(dev), loglik_rel_err=float(abs(dev - ll) / abs(ll)),
                    chain_loglik_rel_err=float(abs(mcmc._logLik - ll) / abs(ll)), tolerance="1e-4 relative (BASELINE.json)")


class WideNetwork(Workload):
    """A network the LDS of a compute unit cannot hold (the weight-streamed path, npbnn_amd/csrc/npbnn_wide.hip.h): 20 000 x 4 096
    features, hidden [256, 64], 10 classes, tanh, bias 2 - layer 0 is a real contraction (42 GF per proposal against 328 MB of X)."""
    config = 9
    n, f, c, hidden = 20_000, 4096, 10, [256, 64]
    short = "wide network (20k x 4096, [256,64])"
    description = "weight-streamed path: 20k x 4096 features, 10 classes, hidden [256,64], tanh, bias 2"

    def __init__(self):
        rs = np.random.default_rng(0)
        self.x = rs.standard_normal((self.n, self.f)).astype(np.float32)
        self.y = rs.integers(0, self.c, self.n)
        self.bytes_per_proposal = 4.0 * self.n * self.f + 4.0 * self.n
        self.flops_layer0 = 2.0 * self.n * self.f * self.hidden[0]
        self.flops = self.flops_layer0 + 2.0 * self.n * (self.hidden[0] * self.hidden[1] + self.hidden[1] * self.c)

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, **sampler_kw):
        dat = dict(data=self.x, labels=self.y, test_data=np.zeros((0, self.f)), test_labels=np.zeros(0))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat, n_nodes=self.hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
        return bnn, bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **sampler_kw)

    def oracle_chain(self, orc):
        np.random.seed(1234)
        return orc.make_chain(self.x.astype(np.float64), self.y, self.hidden, act=orc.Act("tanh"), use_bias_node=2, prior_kind=1, p_scale=1)

    def parity(self, bnn, mcmc):
        import oracle as orc
        w = [np.array(v, dtype=np.float64) for v in bnn._w_layers]
        pred = orc.forward(self.x.astype(np.float64), w, orc.Act("tanh"), orc.out_softmax)
        ll = orc.lik_categorical(pred, self.y, np.arange(self.n))
        dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"]
        dev_pred = mcmc._backend.predict(bnn._w_layers)
        return dict(loglik_oracle=float(ll), loglik_device=float(dev), loglik_rel_err=float(abs(dev - ll) / abs(ll)),
                    chain_loglik_rel_err=float(abs(mcmc._logLik - ll) / abs(ll)), pred_max_abs_err=float(np.max(np.abs(dev_pred - pred))),
                    tolerance="1e-4 relative (BASELINE.json)")


class DefaultNetworkManyFeatures(Workload):
    """The reference's DEFAULT model (npBNN(n_nodes=[50, 5]), ReLU, bias 1; np_bnn/BNN_env.py:20-23) on 100k x 1024 features: its
    weight image (52 x 1024 x 4 B) no longer fits a compute unit's LDS - the weight-streamed path with its fused passes (the first
    layer's product takes the narrow end of the network and the likelihood along, up to three candidates per read of X)."""
    config = 8
    n, f, c, hidden = 100_000, 1024, 10, [50, 5]
    short = "default network on 1024 features (100k x 1024, [50,5])"
    description = "weight-streamed path, fused passes: 100k x 1024 features, 10 classes, the reference's default network [50,5], ReLU, bias 1"

    def __init__(self):
        rs = np.random.default_rng(0)
        self.x = rs.standard_normal((self.n, self.f)).astype(np.float32)
        self.y = rs.integers(0, self.c, self.n)
        self.bytes_per_proposal = 4.0 * self.n * self.f + 4.0 * self.n
        self.flops_layer0 = 2.0 * self.n * self.f * self.hidden[0]

    def build(self, mcmc_id=0, temperature=1.0, randomize_seed=False, **sampler_kw):
        dat = dict(data=self.x, labels=self.y, test_data=np.zeros((0, self.f)), test_labels=np.zeros(0))
        np.random.seed(1234)
        bnn = _quiet(bn.npBNN, dat)
        return bnn, bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed, **sampler_kw)

    def oracle_chain(self, orc):
        np.random.seed(1234)
        return orc.make_chain(self.x.astype(np.float64), self.y, self.hidden, act=orc.Act("ReLU"), use_bias_node=1, prior_kind=1, p_scale=1)

    def parity(self, bnn, mcmc):
        import oracle as orc
        w = [np.array(v, dtype=np.float64) for v in bnn._w_layers]
        pred = orc.forward(self.x.astype(np.float64), w, orc.Act("ReLU"), orc.out_softmax)
        ll = orc.lik_categorical(pred, self.y, np.arange(self.n))
        dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"]
        return dict(loglik_oracle=float(ll), loglik_device=float(dev), loglik_rel_err=float(abs(dev - ll) / abs(ll)),
                    chain_loglik_rel_err=float(abs(mcmc._logLik - ll) / abs(ll)), tolerance="1e-4 relative (BASELINE.json)")


def workload(config):
    return {2: Config2, 4: Config4, 5: Config5, 0: DefaultNetwork, 9: WideNetwork, 8: DefaultNetworkManyFeatures, 12: Config2LogNormal}[config]()


def build_config2(x, y, hidden, mcmc_id=0, temperature=1.0, randomize_seed=False):
    """BASELINE.json config 2 on given arrays (tools/time_rows_sweep.py): np.random.seed(1234); npBNN(n_nodes=[32,8], tanh, bias nodes in
    input+hidden layers, N(0,1) prior); MCMC defaults (update_f 0.05 -> update_n [411,13,4])."""
    dat = dict(data=x, labels=y, test_data=np.zeros((0, x.shape[1])), test_labels=np.zeros(0))
    np.random.seed(1234)
    bnn = _quiet(bn.npBNN, dat, n_nodes=hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
    mcmc = bn.MCMC(bnn, temperature=temperature, mcmc_id=mcmc_id, randomize_seed=randomize_seed)
    return bnn, mcmc
def exchange_self_check(chains, chain_ids, n_chains, comm, seg_len, make_swaps, rank=0, n_intervals=4):
    """The same ``n_intervals`` swap intervals from the same state on the interval-by-interval path and on the device exchange
    path of THIS machine; the chains are put back where they started.  Returns (every rank agrees the two paths gave the same
    chains, ranks that do not).  Weights and temperatures are compared exactly; the log-posterior to rounding (the two paths may
    pick different launch geometries for a batch, which changes the summation order of the log-likelihood in its last bits)."""
    import copy
    from npbnn_amd import exchange as ex
    keep = ("_logLik", "_logPrior", "_logPost", "_temperature", "_current_iteration", "_last_accepted_mem",
            "_acceptance_rate", "_last_accepted", "_gen")
    saved = []
    for bnn, mcmc in chains:
        mcmc._cancel_speculation()
        saved.append(({k: copy.deepcopy(getattr(mcmc, k)) for k in keep}, [w.copy() for w in bnn._w_layers]))
    outcome = []
    for use_device in (False, True):
        try:
            ex.advance_intervals(chains, chain_ids, n_chains, n_intervals, seg_len, make_swaps(), 0, comm=comm, batch=n_intervals,
                                 device=use_device)
            outcome.append([(np.concatenate([w.ravel() for w in bnn._w_layers]), mcmc._logPost, mcmc._temperature)
                            for bnn, mcmc in chains])
        except Exception as e:           # noqa: BLE001 - any failure of the device path means: use the other one
            print("[rank %d] exchange self-check (%s path) failed: %s" % (rank, "device" if use_device else "host", e), flush=True)
            outcome.append(None)
        for (bnn, mcmc), (state, weights) in zip(chains, saved):
            mcmc._cancel_speculation()
            for k, v in state.items():
                setattr(mcmc, k, copy.deepcopy(v))
            bnn.reset_weights([w.copy() for w in weights])
            mcmc._invalidate()
    same = outcome[0] is not None and outcome[1] is not None
    if same:
        for (wa, pa, ta), (wb, pb, tb) in zip(outcome[0], outcome[1]):
            same = same and np.array_equal(wa, wb) and ta == tb and abs(pa - pb) <= 1e-9 * abs(pa)
    if comm is None or comm.world_size == 1:
        return bool(same), ([] if same else [rank])
    agree = comm.allgather_f64(np.array([1.0 if same else 0.0]))
    return bool(np.all(agree[:, 0] == 1.0)), np.nonzero(agree[:, 0] != 1.0)[0].tolist()
