"""Shared by the CPU and GPU tests: run a sampler through the schedule of one reference-generated option trace
(tests/golden/options.npz, cases.OPTION_TRACES) and hold it to the reference's recorded state after every call."""
import numpy as np

import cases

# columns of cases.option_state that every case has
LOGLIK, LOGPRIOR, LOGPOST, ACCEPTED, ACC_RATE, ITERATION = range(6)


class Divergence(AssertionError):
    pass


def load(golden, name):
    return {k.split("/", 1)[1]: golden[k] for k in golden.files if k.startswith(name + "/")}


def check_initial_state(name, g, bnn, mcmc, rtol_lik, rtol_stats=None):
    for i, w in enumerate(bnn._w_layers):
        np.testing.assert_array_equal(w, g["w0_%d" % i])
    np.testing.assert_array_equal(mcmc._update_n, g["update_n"])
    state = cases.option_state(bnn, mcmc)
    np.testing.assert_allclose(state[LOGLIK], g["init"][LOGLIK], rtol=rtol_lik)
    np.testing.assert_allclose(state[LOGPRIOR], g["init"][LOGPRIOR], rtol=1e-11)
    np.testing.assert_allclose(state[ITERATION + 1:], g["init"][ITERATION + 1:], rtol=1e-12)
    if rtol_stats is not None:
        np.testing.assert_allclose([mcmc._accuracy, mcmc._test_accuracy], g["init_acc"], rtol=rtol_stats, atol=rtol_stats)


def follow(name, g, bnn, mcmc, rtol_lik, advance="mh_step", stats_every=0, rtol_stats=1e-9, chunk=None, rtol_state=1e-9):
    """Run the trace's schedule.  ``advance``: "mh_step" (one call per iteration) or "run_steps" (one device batch per block of
    iterations between two gibbs steps, cut into calls of ``chunk``).  Returns the number of calls (iterations and gibbs steps)
    over which the sampler reproduced the reference's accept / reject sequence and state; raises Divergence where a state
    differs although every decision up to there was the same."""
    cfg = cases.OPTION_TRACES[name]
    states = g["states"]
    done = 0

    def compare(n_calls):
        """State after ``n_calls`` calls against the recorded one; False if the accept sequences have parted."""
        want = states[n_calls - 1]
        got = cases.option_state(bnn, mcmc)
        if got[ITERATION] != want[ITERATION]:
            raise Divergence("%s: iteration counter %r, reference %r" % (name, got[ITERATION], want[ITERATION]))
        if got[ACCEPTED] != want[ACCEPTED] or got[ACC_RATE] != want[ACC_RATE]:
            return False
        try:
            np.testing.assert_allclose(got[LOGLIK], want[LOGLIK], rtol=rtol_lik)
            np.testing.assert_allclose(got[LOGPRIOR], want[LOGPRIOR], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(got[ITERATION + 1:], want[ITERATION + 1:], rtol=rtol_state, atol=1e-12)
        except AssertionError as e:
            raise Divergence("%s: state after call %d differs with the same decisions\n%s" % (name, n_calls, e))
        if stats_every and (n_calls % stats_every == 0 or n_calls == len(states)):
            np.testing.assert_allclose([mcmc._accuracy, mcmc._test_accuracy], g["stats"][n_calls - 1], rtol=rtol_stats, atol=rtol_stats)
        return True

    for what, n in cases.option_schedule(cfg):
        if what == "gibbs":
            mcmc.gibbs_step(bnn)
            if not compare(done + 1):
                return done
            done += 1
        elif advance == "mh_step":
            for _ in range(n):
                mcmc.mh_step(bnn)
                if not compare(done + 1):
                    return done
                done += 1
        else:
            left = n
            while left > 0:
                k = min(left, chunk or 50, 100)        # (the sampler remembers the last 100 flags)
                mcmc.run_steps(bnn, k)
                got_flags = [int(a) for a in mcmc._last_accepted_mem[-k:]]
                want_flags = [int(a) for a in states[done:done + k, ACCEPTED]]
                if got_flags != want_flags:
                    return done + next(i for i in range(k) if got_flags[i] != want_flags[i])
                if not compare(done + k):
                    return done
                done += k
                left -= k
    return done


def check_final_state(name, g, bnn, mcmc, rtol_stats=1e-9, stats=True):
    for i, w in enumerate(bnn._w_layers):
        np.testing.assert_array_equal(w, g["wfinal_%d" % i])          # float64 on the host: bit-equal
    np.testing.assert_array_equal(np.asarray(bnn._indicators, dtype=np.int8), g["final_indicators"])
    if bnn._hyper_p:
        for i, sc in enumerate(bnn._prior_scale):
            np.testing.assert_allclose(sc, g["final_prior_scale_%d" % i], rtol=1e-12)
    if stats:
        np.testing.assert_allclose(mcmc._label_acc, g["final_label_acc"], rtol=rtol_stats, atol=rtol_stats)
