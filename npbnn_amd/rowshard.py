"""One chain over several GPUs: the rows of the training matrix split over the ranks of a communicator.

The reference evaluates a proposal's likelihood as ONE sum over all rows (np_bnn/BNN_lib.py:100-143, called from
MCMC.mh_step, np_bnn/BNN_env.py:467-491) in one process.  Here every rank holds a contiguous share of the rows as its
ordinary training set (``shard_rows``), all ranks run the same chain from the same random streams, and whatever the
likelihood needs from all rows - the log-likelihood sum, the residual moments behind the empirical sigma of a regression,
the confusion counts behind the accuracies - is gathered from every rank and added in RANK ORDER on every rank: the same
floating-point result everywhere, hence the same accept decisions, hence chains that stay identical without further
communication.

  * device-resident batches (``MCMC.run_steps``): npbnn_set_row_shard makes npbnn_chain_run gather the per-pass records
    between the evaluation kernel and the step kernel - ncclAllGather on the chain's stream when the communicator is an
    ``RcclComm``, through the host with any other communicator (the rehearsal form: ``SocketComm`` between processes that
    share one GPU or, in the CPU test-suite, the oracle stand-in);
  * single evaluations (``MCMC.__init__``, ``mh_step``, the lazily computed statistics): ``RowShardedBackend.evaluate``
    adds the local sums across ranks on the host.

What a sharded sampler does not run on the device: the general chain (indicators, other proposal functions) and MC3's
exchange run - those fall back to ``mh_step``, which is sharded through ``evaluate``.  Predictions (``_y``, ``predict``) are
the LOCAL rows' predictions.
"""

import numpy as np

from . import _capi as capi

_LOG_SQRT_2PI = 0.9189385332046727418


def shard_bounds(n_rows, rank, world):
    """Rows [lo, hi) of rank ``rank``: contiguous shares that differ by at most one row."""
    base, extra = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(dat, rank, world):
    """The rank's share of a data dictionary (``get_data`` layout, np_bnn/BNN_files.py:31-160): its rows of ``data``,
    ``labels`` and, when present, of the test set and the instance identifiers; everything else as it is."""
    out = dict(dat)
    for rows, keys in ((len(dat["data"]), ("data", "labels", "id_data", "instance_id")),
                       (len(dat.get("test_data", [])), ("test_data", "test_labels", "id_test_data"))):
        if rows == 0:
            continue
        lo, hi = shard_bounds(rows, rank, world)
        for k in keys:
            v = dat.get(k)
            if v is not None and np.ndim(v) >= 1 and len(v) == rows:
                out[k] = v[lo:hi]
    return out


def _rank_order_sum(gathered):
    tot = np.array(gathered[0], dtype=np.float64, copy=True)
    for r in range(1, len(gathered)):
        tot += gathered[r]
    return tot


class RowShardedBackend:
    """Wraps a model's device context (``HipBackend``; in the CPU test-suite the oracle stand-in): same interface, sums over
    ALL rows."""
    row_sharded = True

    def __init__(self, inner, comm, bnn=None):
        self._inner = inner
        self._comm = comm
        self.rank, self.world = int(comm.rank), int(comm.world_size)
        n_local = float(inner.train_rows())
        counts = np.asarray(comm.allgather_f64(np.array([n_local]))).reshape(self.world)
        self.rows_per_rank = [int(c) for c in counts]
        self.n_rows_total = int(sum(self.rows_per_rank))
        self._callback = None
        if bnn is not None:
            self._agree_on_the_model(bnn)
        inner.set_row_shard(self)

    def _agree_on_the_model(self, bnn):
        """What the reference derives from the label vector of ALL rows must not come from a share of them: the number of classes
        (np_bnn/BNN_env.py:77-78 sizes the output layer by the labels it sees) and the balanced class weights (BNN_env.py:98-102)."""
        sizes = np.asarray(self._comm.allgather_f64(np.array([float(bnn._size_output), float(bnn._n_features)]))).reshape(self.world, 2)
        if np.any(sizes != sizes[0]):
            raise ValueError("the ranks' shares describe different models (outputs, features per rank: %s): every share of a "
                             "classification set must hold every class" % sizes.astype(int).tolist())
        if bnn._estimation_mode == "classification" and len(bnn._class_w):
            local = np.bincount(np.asarray(bnn._labels, dtype=np.int64), minlength=int(bnn._size_output)).astype(np.float64)
            counts = _rank_order_sum(np.asarray(self._comm.allgather_f64(local)).reshape(self.world, -1))
            inv = counts.max() / counts
            bnn._class_w = inv / inv.mean()
            self._inner.refresh_row_weights(bnn)

    def __getattr__(self, name):            # everything that is not about sums over rows: the wrapped context's
        return getattr(self._inner, name)

    # -- the gather the C library calls between a pass and its step when the communicator is not RCCL --------------
    def gather_callback(self):
        if self._callback is None:
            comm, world = self._comm, self.world

            def gather(_user, send, recv, count):
                try:
                    mine = np.ctypeslib.as_array(send, shape=(count,))
                    allv = np.asarray(comm.allgather_f64(mine), dtype=np.float64).reshape(world, count)
                    np.ctypeslib.as_array(recv, shape=(world * count,))[:] = allv.ravel()
                    return 0
                except Exception:           # (the library reports NPBNN_E_COMM; an exception cannot cross the C frames)
                    return 1
            self._callback = capi.GATHER_FN(gather)
        return self._callback

    def rccl_handle(self):
        """The communicator's npbnn_comm* when it is an RcclComm on this context's GPU, else None."""
        return getattr(self._comm, "_comm", None) if type(self._comm).__name__ == "RcclComm" else None

    # -- single evaluations ------------------------------------------------------------------------------------------------
    def evaluate(self, weights, slopes=None, col_override=None, lik_temp=1.0, sigma=None, which=capi.TRAIN, want_confusion=False):
        inner = self._inner
        gaussian = getattr(inner, "lik_kind", None) == capi.LIK_GAUSS
        if gaussian:
            # local residual moments (any sigma will do for them), then sigma and the log-likelihood from the totals exactly as
            # the device forms them from its own (loglik_from_totals, npbnn_chain.hip.h): sum_j -N (log sqrt(2 pi) + log s_j) - S2_j / (2 s_j^2)
            k = int(inner.n_targets)
            r = inner.evaluate(weights, slopes=slopes, col_override=col_override, lik_temp=1.0,
                               sigma=np.ones(k) if sigma is None else sigma, which=which)
            vec = np.concatenate([[float(r["n_rows"])], np.asarray(r["sum_r"], dtype=float), np.asarray(r["sum_r2"], dtype=float)])
            tot = _rank_order_sum(self._comm.allgather_f64(vec))
            n, s1, s2 = tot[0], tot[1:1 + k], tot[1 + k:1 + 2 * k]
            if sigma is None:
                mean = s1 / n
                sg = np.sqrt(s2 / n - mean * mean)
            else:
                sg = np.broadcast_to(np.asarray(sigma, dtype=float), (k,)).copy()
            ll = 0.0
            for j in range(k):
                ll += -n * (_LOG_SQRT_2PI + np.log(sg[j])) - s2[j] / (2.0 * sg[j] * sg[j])
            return dict(loglik=float(lik_temp * ll), sigma=sg, sum_r=s1, sum_r2=s2, n_rows=int(n), confusion=None)
        r = inner.evaluate(weights, slopes=slopes, col_override=col_override, lik_temp=lik_temp, sigma=sigma, which=which,
                           want_confusion=want_confusion)
        parts = [np.array([float(r["loglik"]), float(r["n_rows"])])]
        conf = r.get("confusion")
        if conf is not None:
            parts.append(np.asarray(conf, dtype=np.float64).ravel())
        tot = _rank_order_sum(self._comm.allgather_f64(np.concatenate(parts)))
        out = dict(r)
        out["loglik"], out["n_rows"] = float(tot[0]), int(tot[1])
        if conf is not None:
            out["confusion"] = np.rint(tot[2:]).astype(np.int64).reshape(np.shape(conf))
        return out

    # -- device-resident batches --------------------------------------------------------------------------------------------
    def run_chain(self, weights, **kw):
        return self._inner.run_chain(weights, **kw)

    def _no(self, *a, **k):
        raise NotImplementedError("a row-sharded sampler runs plain device batches only (MCMC.run_steps falls back to mh_step "
                                  "for the general chain; MC3 exchange runs need one chain per rank instead)")

    run_chain_general = exchange_job = _no
    run_exchange = run_batched = staticmethod(_no)
