"""Helpers for bench.py: the config-2 model and the timed Metropolis-Hastings loop."""
import contextlib
import io

import numpy as np

import npbnn_amd as bn


def build_config2(x, y, hidden, mcmc_id=0, temperature=1.0, randomize_seed=False):
    """BASELINE.json config 2 through the reference's own call sequence: np.random.seed(1234); npBNN(n_nodes=[32,8],
    tanh, bias nodes in input+hidden layers, N(0,1) prior); MCMC defaults (update_f 0.05 -> update_n [411,13,4])."""
    dat = dict(data=x, labels=y, test_data=np.zeros((0, x.shape[1])), test_labels=np.zeros(0))
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dat, n_nodes=hidden, actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
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
