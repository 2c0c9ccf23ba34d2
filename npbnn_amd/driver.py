"""Single-chain driver loop (reference: np_bnn/BNN_mcmc.py:153-170)."""
import numpy as np


def run_mcmc(bnn, mcmc, logger):
    """Run ``mcmc.mh_step`` until ``_n_iterations``, printing every ``_print_f`` iterations and
    logging a posterior sample every ``_sampling_f`` iterations."""
    while True:
        mcmc.mh_step(bnn)
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
