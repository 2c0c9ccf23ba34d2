import os, sys
sys.path.insert(0, '/root/repo')
from bench_support import workload
wl = workload(2)
for uf, sched in ((None, 2), ([0.004]*3, 2), ([0.004]*3, 1)):
    kw = {} if uf is None else dict(update_f=uf)
    bnn, m = wl.build(**kw)
    m.device_schedule = sched
    m.run_steps(bnn, 2000)
    os.environ["NPBNN_STEP_STAMPS"] = "1"
    m.run_steps(bnn, 1000)
    del os.environ["NPBNN_STEP_STAMPS"]
    print("update_f", uf, "schedule", sched, "acceptance", m._acceptance_rate, "its/pass", m._device_iterations / m._device_passes, flush=True)
    m._backend.close()
