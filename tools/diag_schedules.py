import contextlib, io, os, sys, warnings
import numpy as np
sys.path.insert(0, '/root/repo')
import npbnn_amd as bn
rs = np.random.default_rng(0)
n, f, c = 100_000, 256, 10
x = rs.standard_normal((n, f)).astype(np.float32)
y = rs.integers(0, c, n)
uf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.004
chains = []
for sched in (1, 2):
    np.random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        bnn = bn.npBNN(dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0)), n_nodes=[32, 8],
                       actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
    m = bn.MCMC(bnn, update_f=[uf] * 3)
    m.device_schedule = sched
    chains.append((bnn, m))
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 250
for it in range(0, 10000, chunk):
    for b, m in chains:
        m.run_steps(b, chunk)
    (b1, m1), (b2, m2) = chains
    d = dict(ll=m1._logLik == m2._logLik, lp=m1._logPrior == m2._logPrior, acc=m1._device_accepted == m2._device_accepted,
             mem=m1._last_accepted_mem == m2._last_accepted_mem, w=all(np.array_equal(u, v) for u, v in zip(b1._w_layers, b2._w_layers)))
    if not all(d.values()):
        print("after", it + chunk, d, m1._logLik, m2._logLik, m1._logPrior, m2._logPrior, m1._device_accepted, m2._device_accepted)
        print("max |dw|", max(np.max(np.abs(u - v)) for u, v in zip(b1._w_layers, b2._w_layers)))
        break
else:
    print("identical through 10000")
