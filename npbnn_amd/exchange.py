"""Exchange runs: the chains of an MC3 run advance several swap intervals per device call.

The reference's MC3 loop (np_bnn/BNN_mc3.py:94-112) alternates ``swap_frequency`` iterations of every chain with one
temperature-swap proposal.  The swap reads two scalars per chain, so here the whole alternation is enqueued on the GPU
streams (``npbnn_chains_run_exchange``): segment, record, all-gather (RCCL between processes), decision, next segment -
with no host round trip in between.  This module holds the host side:

    SwapProposals         the (j, k, log u) of the swaps, pre-drawn in the reference's order from ``np.random``
    exchange_ready        can these chains take the device path for the next n_seg intervals?
    run_exchange          n_seg intervals of every local chain; falls back to the interval-by-interval path for the
                          interval in which some chain was given too few launches (rare; every rank sees it in the records)
"""
import numpy as np


class SwapProposals:
    """Swap proposals in the order MC3.run_mcmc draws them (BNN_mc3.py:99,110): per swap one
    ``choice(range(n_chains), 2, replace=False)`` then one ``random()``, from ``np.random`` (or the given RandomState).
    Drawn ahead in blocks and served by swap index, so that the device path (which needs them up front) and the
    interval-by-interval path consume the same stream."""

    def __init__(self, n_chains, rs=None):
        self.n_chains = int(n_chains)
        self._rs = rs if rs is not None else np.random
        self._first = 0
        self._j, self._k, self._logu = [], [], []

    def _extend(self, upto):
        while self._first + len(self._j) < upto:
            j, k = self._rs.choice(range(self.n_chains), 2, replace=False)
            u = self._rs.random() if hasattr(self._rs, "random") else self._rs.random_sample()
            self._j.append(int(j))
            self._k.append(int(k))
            self._logu.append(float(np.log(u)))

    def get(self, first, n=1):
        """(j, k, log u) arrays of swaps first .. first+n-1."""
        if first < self._first:
            raise ValueError("swap %d was already released" % first)
        self._extend(first + n)
        a = first - self._first
        return (np.array(self._j[a:a + n], dtype=np.int32), np.array(self._k[a:a + n], dtype=np.int32),
                np.array(self._logu[a:a + n], dtype=np.float64))

    def release(self, upto):
        """Forget the swaps before ``upto`` (they have been applied)."""
        n = min(max(0, upto - self._first), len(self._j))
        del self._j[:n], self._k[:n], self._logu[:n]
        self._first += n


def swap_decision(log_post_j, log_post_k, temp_j, temp_k, log_u):
    """The accept test of a temperature swap (BNN_mc3.py:102-110)."""
    r = (log_post_k - log_post_j) * temp_j + (log_post_j - log_post_k) * temp_k
    return r, bool(r >= log_u)


def exchange_ready(chains, n_iterations):
    """True when every (bnn, mcmc) of ``chains`` can run its next ``n_iterations`` as one device batch with the
    proposal settings fixed: a device backend with the exchange entry point, the default sampler path, no adaptation
    point inside, per-iteration reseeding (MC3 chains, BNN_env.py:384) so that the draws are a function of the
    iteration alone."""
    for bnn, mcmc in chains:
        if mcmc._backend is None:
            from .sampler import get_backend
            mcmc._backend = get_backend(bnn, mcmc._likelihood_f)
        be = mcmc._backend
        if not hasattr(be, "exchange_job"):
            return False
        if not mcmc._randomize_seed or not mcmc._device_loop_ok(bnn, n_iterations):
            return False
        if not mcmc._plain_device_batches(bnn):
            return False
        boundary = mcmc._next_adapt_boundary()
        if boundary is not None and boundary < mcmc._current_iteration + n_iterations:
            return False
    return True


def run_exchange(chains, chain_ids, n_chains, n_seg, seg_len, swaps, first_swap, comm=None, launch_slack=None, want_cold_w=True):
    """Advance every local chain by up to ``n_seg`` swap intervals on the device.

    chains      [(bnn, mcmc)] of this process, ``chain_ids`` their global ids (chain i on rank i % world, in id order)
    swaps       SwapProposals; swap number ``first_swap`` follows the first interval
    comm        communicator (``RcclComm`` / ``LocalComm``); its native handle carries the all-gather
    Returns (segments_done, records, outs): ``records[s, i] = (logPost, temperature before swap s, done flag, iterations)`` of
    every chain, ``outs[q]`` the per-interval state / cold-chain weights of local chain q.  ``segments_done < n_seg`` only when a
    chain fell short of an interval: every chain then stands somewhere inside interval ``segments_done`` (possibly at its end)
    and the caller completes it the slow way."""
    K = int(n_seg) * int(seg_len)
    sj, sk, su = swaps.get(first_swap, n_seg)
    jobs = []
    backend_cls = None
    for (bnn, mcmc), cid in zip(chains, chain_ids):
        mcmc._bnn = bnn
        it = mcmc._current_iteration
        idx, delta, cnt, log_u, smult, hast = mcmc._claim_draw(bnn, it, K).result()[:6]
        # the draws of the probable next call, made while the GPU runs this one
        mcmc._speculation = mcmc._submit_draw(bnn, it + K, K, rewindable=True)
        job = mcmc._backend.exchange_job(bnn._w_layers, chain_id=cid, idx=idx, delta=delta, cnt=cnt, log_u=log_u, mask=bnn._mask,
                                         cfg=mcmc._device_chain_cfg(bnn, smult, hast))
        jobs.append(job)
        backend_cls = type(mcmc._backend)
    slack = launch_slack if launch_slack is not None else getattr(backend_cls, "exchange_slack", 1.5)
    handle = getattr(comm, "_comm", None) if comm is not None else None
    outs, records, done = backend_cls.run_exchange(jobs, n_chains, seg_len, n_seg, sj, sk, su, comm=handle, launch_slack=slack,
                                                   want_cold_w=want_cold_w)
    if launch_slack is None:           # launches per interval relative to the expected number: creep down while every interval
        # gets through (idle launches cost time), jump up when one fell short (that costs a lot more)
        backend_cls.exchange_slack = min(4.0, slack * 1.5) if done < n_seg else max(backend_cls.exchange_slack_floor, slack * 0.97)
    for (bnn, mcmc), out in zip(chains, outs):
        res = out["result"]
        k = int(res["iterations_done"])
        if k < K:
            mcmc._cancel_speculation()
        mcmc._absorb_device_batch(bnn, k, out["w"], out["accepted"], res)
        if k > 0:
            mcmc._temperature = res["temperature"]
    return done, records, outs


def _batchable(chains, n_iterations):
    """Can these chains (of one process) share their passes over the data for the next ``n_iterations``?  Replicas of one model
    on one device (shared resident matrix), the device chain applies to each, no adaptation point inside."""
    if len(chains) < 2:
        return False
    first = None
    for bnn, mcmc in chains:
        if mcmc._backend is None:
            from .sampler import get_backend
            mcmc._backend = get_backend(bnn, mcmc._likelihood_f)
        be = mcmc._backend
        if not hasattr(be, "run_batched") or not mcmc._device_loop_ok(bnn, n_iterations) or not mcmc._plain_device_batches(bnn):
            return False
        boundary = mcmc._next_adapt_boundary()
        if boundary is not None and boundary < mcmc._current_iteration + n_iterations:
            return False
        if bnn._estimation_mode == "regression" and mcmc._current_iteration <= mcmc._estimate_error < mcmc._current_iteration + n_iterations:
            return False
        root = be.data_shared_with if getattr(be, "data_shared_with", None) is not None else be
        key = (id(root), tuple(w.shape for w in bnn._w_layers), id(bnn._mask) if bnn._mask is not None else None)
        if first is None:
            first = key
        elif key[0] != first[0] or key[1] != first[1] or (key[2] is None) != (first[2] is None):
            return False
    return True


def run_steps_batched(chains, n_iterations):
    """``n_iterations`` iterations of every chain of ``chains`` ([(bnn, mcmc)], all in this process) - what
    ``for bnn, mcmc in chains: mcmc.run_steps(bnn, n_iterations)`` does - with the chains of one model sharing their passes over
    the feature matrix in groups of up to three (``npbnn_chains_run_batched``): one proposal per chain per streaming read of X.
    Chains that cannot take part (a sampler setting the device chain does not cover, an adaptation point inside the span,
    matrices of their own) advance on their own."""
    todo = list(chains)
    while todo:
        be0 = todo[0][1]._backend
        size = getattr(be0, "group_size", 1) if be0 is not None else 1
        group = todo[:size]
        while len(group) >= 2 and not _batchable(group, n_iterations):
            group = group[:-1]
        if len(group) < 2:
            bnn, mcmc = todo.pop(0)
            mcmc.run_steps(bnn, n_iterations)
            continue
        del todo[:len(group)]
        K = int(n_iterations)
        jobs = []
        for bnn, mcmc in group:
            mcmc._bnn = bnn
            mcmc._adapt(bnn)
            it = mcmc._current_iteration
            idx, delta, cnt, log_u, smult, hast = mcmc._claim_draw(bnn, it, K).result()[:6]
            mcmc._speculation = mcmc._submit_draw(bnn, it + K, K, rewindable=True)      # the probable next call's draws, meanwhile
            jobs.append(mcmc._backend.exchange_job(bnn._w_layers, chain_id=len(jobs), idx=idx, delta=delta, cnt=cnt, log_u=log_u,
                                                   mask=bnn._mask, cfg=mcmc._device_chain_cfg(bnn, smult, hast)))
        try:
            outs = type(group[0][1]._backend).run_batched(jobs, K)
        except Exception:
            for _, mcmc in group:
                mcmc._cancel_speculation()
            raise
        for (bnn, mcmc), out in zip(group, outs):
            mcmc._absorb_device_batch(bnn, K, out["w"], out["accepted"], out["result"])


def gather_scalars(chains, chain_ids, n_chains, comm):
    """[logPost, temperature] of every chain of the run, on every rank (chain i lives on rank i % world)."""
    world = 1 if comm is None else comm.world_size
    per_rank = (n_chains + world - 1) // world
    mine = np.full((per_rank, 2), np.nan)
    for (_, mcmc), cid in zip(chains, chain_ids):
        mine[cid // world] = (mcmc._logPost, mcmc._temperature)
    if comm is None or world == 1:
        allv = mine.reshape(1, per_rank, 2)
    else:
        allv = comm.allgather_f64(mine.ravel()).reshape(world, per_rank, 2)
    out = np.empty((n_chains, 2))
    for i in range(n_chains):
        out[i] = allv[i % world, i // world]
    return out


def host_swap(chains, chain_ids, n_chains, swaps, swap_index, comm=None):
    """One temperature-swap proposal decided on the host from an all-gather of [logPost, temperature]
    (BNN_mc3.py:98-112); every rank reaches the same decision from the same pre-drawn proposal.
    Returns (scalars after the swap, (j, k, r, log u, accepted))."""
    scal = gather_scalars(chains, chain_ids, n_chains, comm)
    sj, sk, su = swaps.get(swap_index, 1)
    j, k, log_u = int(sj[0]), int(sk[0]), float(su[0])
    temp_j, temp_k = scal[j, 1] + 0, scal[k, 1] + 0
    r, accepted = swap_decision(scal[j, 0], scal[k, 0], temp_j, temp_k, log_u)
    if accepted:
        for (_, mcmc), cid in zip(chains, chain_ids):
            if cid == j:
                mcmc.reset_temperature(temp_k)
            elif cid == k:
                mcmc.reset_temperature(temp_j)
        scal[j, 1], scal[k, 1] = temp_k, temp_j
    return scal, (j, k, float(r), log_u, accepted)


GROUP_PASS_ACCEPTANCE = 0.10     # advance_intervals(group_passes="auto"): share the passes above this acceptance rate


def advance_intervals(chains, chain_ids, n_chains, n_intervals, seg_len, swaps, first_swap, comm=None, batch=20, device=True,
                      on_interval=None, group_passes=False):
    """``n_intervals`` rounds of [seg_len iterations of every chain, one swap proposal] - MC3.run_mcmc's loop body
    (BNN_mc3.py:94-112) - in device batches of up to ``batch`` intervals where the chains allow it, interval by interval
    otherwise.  ``on_interval(index, info)`` is called after every swap with ``info`` = dict(scalars=[n_chains, 2] logPost /
    temperature after the swap, swap=(j, k, r, log u, accepted), cold=per local chain None or dict(w, loglik, logprior) when the
    interval ran on the device and that chain is the cold one afterwards - ``iterations`` into the device batch that started at
    ``iteration0`` with the acceptance memory ``mem_before``; ``last_of_batch`` marks the last interval of a device batch).
    ``group_passes``: on the interval-by-interval path the chains of this process advance through group passes
    (:func:`run_steps_batched`) instead of one after the other - the same chains, their log-likelihoods equal to rounding rather
    than to the bit (a group's sums come from a different number of workgroup partials).  ``"auto"``: group passes, on that path,
    for a batch whose chains have been accepting more than ``GROUP_PASS_ACCEPTANCE`` of their proposals - a chain that moves wastes
    its own speculative candidates, a group pass has one useful candidate per chain whatever the acceptance rate (config-2 shapes,
    3 chains on one MI355X: 61 k against 45-56 k it/s aggregate at 18-25 % acceptance, 72 k against 82 k at 3 %;
    tools/time_group_pass.py) - and device batches otherwise."""
    world = 1 if comm is None else comm.world_size
    done = 0
    while done < n_intervals:
        n = min(int(batch), n_intervals - done)
        grouped = group_passes
        if group_passes == "auto":
            grouped = len(chains) >= 2 and float(np.mean([m._acceptance_rate for _, m in chains])) > GROUP_PASS_ACCEPTANCE
        ok = (bool(device) and not (group_passes == "auto" and grouped) and n >= 2 and n_chains == world * len(chains) and exchange_ready(chains, n * seg_len)
              and (world == 1 or getattr(comm, "_comm", None) is not None))        # (several ranks: the native RCCL handle)
        if world > 1 and bool(device):      # every rank must take the same path (without `device` - the same argument on every rank - it is
            # the interval-by-interval one everywhere: nothing to agree on, and no collective of its own per interval for it)
            ok = bool(np.all(comm.allgather_f64(np.array([1.0 if ok else 0.0]))[:, 0] == 1.0))
        if not ok:
            if grouped:                         # the local chains share their passes over the data (run_steps_batched)
                run_steps_batched(chains, seg_len)
            else:
                for bnn, mcmc in chains:
                    mcmc.run_steps(bnn, seg_len)
            scal, swap = host_swap(chains, chain_ids, n_chains, swaps, first_swap + done, comm)
            if on_interval is not None:
                on_interval(done, dict(scalars=scal, swap=swap, cold=None))
            done += 1
            swaps.release(first_swap + done)
            continue
        it0 = [mcmc._current_iteration for _, mcmc in chains]
        mem0 = [list(mcmc._last_accepted_mem) for _, mcmc in chains]
        n_done, records, outs = run_exchange(chains, chain_ids, n_chains, n, seg_len, swaps, first_swap + done, comm=comm,
                                             want_cold_w=on_interval is not None)
        sj, sk, su = swaps.get(first_swap + done, n)
        for s in range(n_done):
            if on_interval is None:
                continue
            scal = records[s, :, :2].copy()
            j, k = int(sj[s]), int(sk[s])
            r, accepted = swap_decision(scal[j, 0], scal[k, 0], scal[j, 1], scal[k, 1], float(su[s]))
            if accepted:
                scal[j, 1], scal[k, 1] = scal[k, 1], scal[j, 1]
            cold = []
            for q, out in enumerate(outs):
                st = out["state"][s]
                cold.append(dict(w=out["cold_w"][s], loglik=st[0], logprior=st[1], accepted=out["accepted"], sigma=st[4:],
                                 iterations=(s + 1) * seg_len, iteration0=it0[q], mem_before=mem0[q])
                            if st[2] == 1.0 and out["cold_w"] is not None else None)
            on_interval(done + s, dict(scalars=scal, swap=(j, k, float(r), float(su[s]), accepted), cold=cold,
                                       last_of_batch=s == n_done - 1))
        done += n_done
        if n_done < n:                 # some chain fell short inside interval n_done: finish that one the slow way
            for (bnn, mcmc), start in zip(chains, it0):
                rest = start + (n_done + 1) * seg_len - mcmc._current_iteration
                if rest > 0:
                    mcmc.run_steps(bnn, rest)
            scal, swap = host_swap(chains, chain_ids, n_chains, swaps, first_swap + done, comm)
            if on_interval is not None:
                on_interval(done, dict(scalars=scal, swap=swap, cold=None))
            done += 1
        swaps.release(first_swap + done)
    return done
