#!/usr/bin/env python3
"""Benchmark of the npBNN MCMC hot path on MI355X.

metric : MCMC iterations/sec, a full forward pass + likelihood per proposal
         (BASELINE.json).  One "step" = one Metropolis-Hastings iteration of one chain
         per GPU on BASELINE.json config 2: synthetic 100k x 256 features, 10 classes,
         hidden [32, 8], tanh, bias nodes in input+hidden layers.
N GPUs : one independent chain per GPU (MC3 layout, config 3), weak scaling; the only
         exchange is the temperature-swap all-gather every `swap_frequency` iterations.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS, N_FEATURES, N_CLASSES, HIDDEN = 100_000, 256, 10, [32, 8]
HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
# HBM bytes per launch of the 3-candidate pass kernel measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
# on this workload (profiles/r01_pass3_pmc_*.csv): (2 x 50701 KiB [gfx950 reports half of a wide streaming read] + 24 KiB)
MEASURED_TRAFFIC_BYTES = (2 * 50701.0 + 24.0) * 1024


def synthetic_config2():
    rs = np.random.default_rng(0)
    x = rs.standard_normal((N_ROWS, N_FEATURES))
    y = rs.integers(0, N_CLASSES, N_ROWS)
    return x, y


def cpu_baseline(x, y, budget_s=15.0):
    """The oracle's op-for-op numpy float64 restatement of MCMC.mh_step timed on the host
    cores (reported baseline, not the target).  Bounded sample: as many iterations as fit
    in ~budget_s seconds (at least 5)."""
    import oracle as orc
    np.random.seed(1234)
    st = orc.make_chain(x, y, HIDDEN, act=orc.Act("tanh"), use_bias_node=2, prior_kind=1, p_scale=1)
    orc.mh_step(st)
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.mh_step(st)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 5) or n >= 2000:
            break
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    return dict(value=n / el, unit="MCMC iterations/s", cores=int(blas_threads), kind="port",
                sample="%d mh_step iterations of config 2 (100k x 256, [32,8]) with the numpy float64 oracle, "
                       "%d BLAS threads of %d host CPUs" % (n, blas_threads, os.cpu_count()))


def parity_readout(x, y, bnn, mcmc):
    """State of the timed chain against the float64 oracle (same leg as the CPU baseline: rank 0, one GPU): relative error
    of the device log-likelihood of the chain's current weights and max abs error of its class probabilities on a row sample."""
    import oracle as orc
    w = [np.array(v, dtype=np.float64) for v in bnn._w_layers]
    xs = x.astype(np.float32).astype(np.float64)
    pred = orc.forward(xs, w, orc.Act("tanh"), orc.out_softmax)
    ll = orc.lik_categorical(pred, y, np.arange(len(y)))
    dev = mcmc._backend.evaluate(bnn._w_layers, None)["loglik"] if hasattr(mcmc._backend, "evaluate") else float("nan")
    rows = np.arange(0, len(y), 97)
    y_dev = np.asarray(mcmc._y)[rows]
    return dict(loglik_oracle=float(ll), loglik_device=float(dev), loglik_rel_err=float(abs(dev - ll) / abs(ll)),
                chain_loglik_rel_err=float(abs(mcmc._logLik - ll) / abs(ll)),
                prediction_max_abs_err=float(np.max(np.abs(y_dev - pred[rows]))), tolerance="1e-4 relative (BASELINE.json)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # NPBNN_BENCH_DIST_BACKEND=gloo: rehearsal of the multi-rank flow on a box with fewer GPUs than ranks (ranks share GPUs, the
    # exchange goes through the host path over gloo); the driver's runs use the default, one GPU per rank over RCCL
    dist_backend = os.environ.get("NPBNN_BENCH_DIST_BACKEND", "nccl")
    device_index = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        if dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            device_index = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group(dist_backend)

    from bench_support import build_config2

    x, y = synthetic_config2()
    # one chain per GPU, MC3 layout (config 3): chain r has mcmc_id r, temperature linspace(0.8, 1, world)[r]
    temps = [1.0] if world == 1 else list(np.linspace(0.8, 1.0, world))
    if world > 1:
        os.environ["NPBNN_DEVICE"] = str(device_index)
    bnn, mcmc = build_config2(x.astype(np.float32), y, HIDDEN, mcmc_id=rank, temperature=temps[rank],
                              randomize_seed=world > 1)
    comm = None
    swap_frequency = 100
    if world > 1:
        import torch
        from npbnn_amd.comm import RcclComm, TorchDistComm

        # the launcher's process group carries the 128-byte RCCL unique id; every rank takes part in every collective
        # below whatever happens, so a failure anywhere cannot leave the others waiting
        box = [None]
        on_gpu = dist_backend == "nccl"
        try_native = on_gpu or bool(os.environ.get("NPBNN_BENCH_TRY_RCCL"))      # (rehearsal: ranks sharing a GPU, if RCCL accepts that)
        if rank == 0 and try_native:
            try:
                box[0] = RcclComm.make_unique_id()
            except Exception as e:
                print("[rank 0] cannot create an RCCL unique id (%s)" % e, flush=True)
        dist.broadcast_object_list(box, src=0, device=torch.device("cuda", local_rank) if on_gpu else None)
        ok = 1 if box[0] else 0
        if ok:
            try:
                comm = RcclComm(rank=rank, world_size=world, device=device_index, uid=box[0])
            except Exception as e:
                print("[rank %d] native RCCL communicator unavailable (%s)" % (rank, e), flush=True)
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            comm_kind = "rccl (C ABI)"
        else:                                       # every rank falls back together; torch.distributed "nccl" is RCCL over xGMI too
            comm = TorchDistComm()
            comm_kind = "rccl (torch.distributed)" if on_gpu else "%s (torch.distributed)" % dist_backend
    else:
        comm_kind = "none"

    # the swap proposals (which two chains, the uniform of the accept test: BNN_mc3.py:99,110) come from a stream every rank
    # seeds identically, so the records [logPost, temperature] of the chains are the whole exchange.  Default: the exchange
    # run - batches of swap intervals enqueued on the GPU stream, records all-gathered in place by RCCL on that stream, the
    # decision applied by a kernel (npbnn_chains_run_exchange); checked below against the interval-by-interval path (one
    # device batch per interval, all-gather and decision on the host), which takes over if the check fails.
    from npbnn_amd import exchange as ex
    swaps = ex.SwapProposals(max(world, 2), np.random.RandomState(4321))
    chains, ids = [(bnn, mcmc)], [rank]
    state = {"swap": 0, "device": world > 1 and getattr(comm, "_comm", None) is not None and not os.environ.get("NPBNN_BENCH_HOST_SWAPS")}
    exchange_path = "none"

    def advance(n):
        """n iterations of every chain; with several chains, a temperature-swap exchange every swap_frequency."""
        if world == 1:
            mcmc.run_steps(bnn, n)
            return
        whole, rest = divmod(n, swap_frequency)
        if whole:
            ex.advance_intervals(chains, ids, world, whole, swap_frequency, swaps, state["swap"], comm=comm, batch=20,
                                 device=state["device"])
            state["swap"] += whole
        if rest:
            mcmc.run_steps(bnn, rest)

    if world > 1 and state["device"]:
        # self-check of the device exchange path on this machine: the same 4 intervals both ways from the same state
        from bench_support import exchange_self_check
        mcmc.run_steps(bnn, swap_frequency)
        state["device"], bad_ranks = exchange_self_check(chains, ids, world, comm, swap_frequency,
                                                         lambda: ex.SwapProposals(world, np.random.RandomState(99)), rank=rank)
        if rank == 0 and not state["device"]:
            print("[bench] device exchange path disagrees with the host path on ranks %s: using the host path" % bad_ranks, flush=True)
    if world > 1:
        exchange_path = ("device: swap intervals in batches of 20 on the stream, records all-gathered in place, decision by a kernel"
                         if state["device"] else "host: one device batch per interval, all-gather and decision on the host")

    advance(args.warmup)
    accepted_before = mcmc._device_accepted

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    advance(args.steps)
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        ctx = mcmc._backend.ctx
        # dominant kernel: the evaluation kernel of a chain pass.  One launch streams X once and evaluates `cand`
        # proposals against it (speculative Metropolis-Hastings: iteration t and t+1.. assuming the earlier ones are
        # rejected; the step kernel stops at the first accept, so the chain is the sequential one).
        ms_kernel, cand = ctx.time_pass(bnn._w_layers, n_candidates=mcmc.n_candidates, iters=50)
        ms_single, _ = ctx.time_eval(bnn._w_layers, iters=50)
        bytes_per_proposal = 4.0 * N_ROWS * N_FEATURES + 4.0 * N_ROWS       # X in float32 + labels
        alg_bytes = bytes_per_proposal * cand
        achieved = alg_bytes / (ms_kernel * 1e-3)
        its_per_pass = mcmc._device_iterations / max(1, mcmc._device_passes)
        line = {
            "metric": "MCMC iterations/sec (full fwd+lik per proposal)",
            "value": world * args.steps / el,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config 2: 100k x 256 features, 10 classes, hidden [32,8], tanh, bias 2; "
                                   "one chain per GPU", "chains": world, "swap_frequency": swap_frequency if world > 1 else None,
                       "swap_exchange": comm_kind, "swap_exchange_path": exchange_path, "layer0": ctx.l0_mode(),
                       "loop": "device-resident chain (npbnn_chain_run), proposals pre-drawn on the host",
                       "candidates_per_pass": cand, "iterations_per_pass": its_per_pass},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK,
                         "traffic": MEASURED_TRAFFIC_BYTES,
                         "traffic_source": "profiles/r01_pass3_pmc_FETCH_SIZE.csv + r01_pass3_pmc_WRITE_SIZE.csv (rocprofv3 --pmc, "
                                           "separate passes): one streaming read of X per launch, whatever the number of candidates",
                         "kernel": "eval_kernel<MT0=2,MTI=1,%s,D=%d,LK=categorical>" % ("fp16-split" if ctx.l0_mode() == "f16-split" else "f32", cand),
                         "kernel_ms": ms_kernel, "proposals_per_launch": cand, "bytes_per_proposal": bytes_per_proposal,
                         "algorithmic_bytes": alg_bytes,
                         "note": "algorithmic bytes = reference bytes per proposal evaluation (SURVEY 8d) x proposals evaluated per "
                                 "launch; the kernel reads X once for all of them, so frac can exceed the HBM share it actually uses "
                                 "(physical: traffic / kernel time)",
                         "physical_GBps": MEASURED_TRAFFIC_BYTES / (ms_kernel * 1e-3) / 1e9,
                         "single_candidate_kernel_ms": ms_single,
                         "single_candidate_frac": bytes_per_proposal / (ms_single * 1e-3) / HBM_PEAK},
            "accept_rate": float(mcmc._device_accepted - accepted_before) / max(1, args.steps),      # rank 0, timed region
            "accept_rate_last_100": float(mcmc._acceptance_rate),
            "loglik": float(mcmc._logLik),
        }
        passes, voids = max(1, mcmc._device_passes), mcmc._device_void_passes
        used = getattr(mcmc, "_device_schedule_used", 0)
        line["config"]["schedule"] = {
            1: "serial: evaluate a pass, decide it, evaluate the next",
            2: "overlapped: the launch that evaluates pass L also decides pass L-1 (one workgroup); a pass overtaken by an accept is "
               "dropped and re-evaluated",
            3: "overlapped, launches alternating between two streams: the workgroups of launch L+1 take the compute units over as "
               "launch L drains, device-side flags (agent-scope release / acquire) order what a kernel boundary used to; a pass "
               "overtaken by an accept is dropped and re-evaluated"}.get(used, "auto")
        line["config"]["void_pass_fraction"] = voids / (passes + voids)
        line["roofline"]["useful_iterations_per_launch"] = mcmc._device_iterations / (passes + voids)
        if cand != 3:
            line["roofline"]["traffic"] = None      # the PMC figure was collected on the 3-candidate kernel
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(x, y)
            line["parity"] = parity_readout(x, y, bnn, mcmc)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    mcmc._backend.close()


if __name__ == "__main__":
    main()
