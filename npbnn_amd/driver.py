"""Single-chain driver loop (reference: np_bnn/BNN_mcmc.py:153-170)."""
import numpy as np


def _next_event(mcmc):
    """Iterations until the driver next has to look at the chain (print, sample or stop)."""
    it = mcmc._current_iteration
    stops = [mcmc._n_iterations - it] if mcmc._n_iterations > it else []
    for f in (mcmc._print_f, mcmc._sampling_f):
        if f and f > 0:
            stops.append(f - it % f)
    if it == 0:
        stops.append(1)          # the reference also prints after the very first iteration
    return max(1, int(min(stops))) if stops else 1


def run_mcmc(bnn, mcmc, logger):
    """Run the chain until ``_n_iterations``, printing every ``_print_f`` iterations and logging a posterior
    sample every ``_sampling_f`` iterations.  Between two such events the chain advances with
    ``mcmc.run_steps`` (device-resident where the sampler's settings allow, else one ``mh_step`` at a time):
    the same iterations as the reference's ``while True: mcmc.mh_step(bnn)`` loop."""
    while True:
        mcmc.run_steps(bnn, _next_event(mcmc))
        it = mcmc._current_iteration
        if it % mcmc._print_f == 0 or it == 1:
            print(it, np.round([mcmc._logLik, mcmc._accuracy, mcmc._test_accuracy, mcmc._acceptance_rate], 3), flush=True)
            if bnn._estimation_mode == "regression":
                print(bnn._error_prm)
            if bnn._feature_indicators is not None:
                print(bnn._feature_indicators)
        if it % mcmc._sampling_f == 0:
            logger.log_sample(bnn, mcmc)
            logger.log_weights(bnn, mcmc)
        if it == mcmc._n_iterations:
            break
