#!/usr/bin/env python3
"""Benchmark of the npBNN MCMC hot path on MI355X.

metric  MCMC iterations/sec with a full forward pass + likelihood per proposal (BASELINE.json).
step    one chain dispatch = ITERATIONS_PER_STEP (100) Metropolis-Hastings iterations of the chain through
        ``MCMC.run_steps`` - the unit the reference's MC3 hands to a worker between two swap proposals
        (``run_single_mcmc``, np_bnn/BNN_mc3.py:80-85, swap_frequency = 100) - with the chain's state back on the host at
        the end of every step.  With the driver's ``--steps 20 --warmup 5`` that is BASELINE.md's protocol: 500 warm-up
        and 2 000 timed iterations, starting from the freshly initialised chain.  ``value`` = iterations per second.
config  BASELINE.json config 2 by default (synthetic 100k x 256, 10 classes, hidden [32, 8], tanh, bias nodes in input and
        hidden layers) - the line's ``value``; the same run then times config 4 (1M x 64 regression, [16, 4], empirical sigma)
        and config 5 (50k x 512, layer 0 in 8 blocks of 64 inputs x 4 nodes, [32, 8]) the same way and reports them under
        ``other_configs`` (``--only`` skips that; ``--config 4`` / ``--config 5`` make one of them the line's subject).
N GPUs  ``--gpus N``: N chains, one per GPU, MC3 layout (config 3; weak scaling); every step ends with the temperature-swap
        proposal (np_bnn/BNN_mc3.py:98-112), whose only exchange is an all-gather of [logPost, temperature] per chain over
        RCCL.  Launched by ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) - or by this script
        itself: without WORLD_SIZE it starts the N rank processes before anything touches a GPU and relays rank 0's line.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|4|5|0] [--only] [--no-cpu-baseline]
"""
import argparse
import csv
import gc
import glob
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERATIONS_PER_STEP = 100      # swap_frequency of MC3 (np_bnn/BNN_mc3.py:17) = iterations per chain dispatch
HBM_PEAK = 8.0e12              # B/s, MI355X HBM3E (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5, 0), help="BASELINE.json config; 0 = config 2's data under the reference's default network [50,5]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="the named config alone: without the other single-GPU configurations that the "
                                                         "default run (config 2) adds under other_configs")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the N ranks ourselves (fresh processes, before any GPU call in this one)
# ------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Parent of a multi-GPU run: one child per GPU with the environment torch.distributed.run would give it, rendezvous on
    127.0.0.1 (npbnn_amd.launch.spawn_ranks).  This process never touches a GPU and never imports torch; it relays rank 0's JSON
    line and exits with the first failing child's status - the moment a rank fails, the others are killed."""
    from npbnn_amd.launch import spawn_ranks
    status, out0, _ = spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    sys.stdout.write(out0)
    sys.stdout.flush()
    return status


# ------------------------------------------------------------------------------------------------------------
# measured HBM traffic of the dominant kernel: read from the rocprofv3 --pmc passes committed under profiles/
# ------------------------------------------------------------------------------------------------------------
def measured_traffic(config, cand):
    """Bytes per launch of the pass kernel with ``cand`` candidates on this config, from the FETCH_SIZE / WRITE_SIZE passes under
    profiles/ (newest round first): mean counter value (KiB) over the kernel's dispatches; FETCH_SIZE x 2 = the gfx950
    correction for wide streaming reads (MI355X_MICROARCH.md, HBM).  (bytes, source) or (None, reason)."""
    fetch = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg%d_pass%d_pmc_FETCH_SIZE.csv" % (config, cand))), reverse=True)
    if config == 2 and not fetch:
        fetch = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pass%d_pmc_FETCH_SIZE.csv" % cand)), reverse=True)
    if not fetch:
        return None, "no profiles/*_cfg%d_pass%d_pmc_FETCH_SIZE.csv" % (config, cand)
    write = fetch[0].replace("FETCH_SIZE", "WRITE_SIZE")

    def mean_kib(path, counter):
        """mean over the dispatches of eval_kernel<MT0, MTI, F16, D = cand, ...> (the run also holds the single evaluation of
        MCMC.__init__, a D = 1 build)"""
        vals = []
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"eval_kernel<\d+, \d+, \w+, (\d+),", row.get("Kernel_Name", ""))
                if row.get("Counter_Name") == counter and m and int(m.group(1)) == cand:
                    vals.append(float(row["Counter_Value"]))
        return sum(vals) / len(vals) if vals else None

    f = mean_kib(fetch[0], "FETCH_SIZE")
    w = mean_kib(write, "WRITE_SIZE") if os.path.exists(write) else 0.0
    if f is None:
        return None, "no eval_kernel rows in %s" % os.path.basename(fetch[0])
    return (2.0 * f + (w or 0.0)) * 1024.0, "%s (FETCH_SIZE x 2, gfx950 wide-read correction) + %s" % (
        os.path.relpath(fetch[0], ROOT), os.path.relpath(write, ROOT))


# ------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, budget_s=15.0):
    """The oracle's op-for-op numpy float64 restatement of MCMC.mh_step timed on the host cores (a reported baseline, not the
    target): as many iterations of the same workload as fit in ~budget_s seconds (at least 5)."""
    import numpy as np
    import oracle as orc
    st = wl.oracle_chain(orc)
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
    del np
    return dict(value=n / el, unit="MCMC iterations/s", cores=int(blas_threads), kind="port",
                sample="%d mh_step iterations of %s with the numpy float64 oracle, %d BLAS threads of %d host CPUs"
                       % (n, wl.short, blas_threads, os.cpu_count()))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        if rank == 0:
            print("[bench] --gpus %d but the launcher started %d ranks; measuring %d" % (args.gpus, world, world), file=sys.stderr)
    # What carries the swap exchange and the timing barrier between the ranks.  Default "rccl": the communicator of the C ABI
    # (npbnn_comm_*), its unique id handed out over a TCP socket on MASTER_ADDR - no torch anywhere in the process, so the only
    # librccl mapped is the one libnpbnn_hip.so was built against.  "socket": plain TCP (rehearsal of the rank flow on fewer GPUs
    # than ranks).  "gloo" / "nccl": a torch.distributed process group (kept for comparison; imports torch).
    dist_backend = os.environ.get("NPBNN_BENCH_DIST_BACKEND", "rccl")
    device_index = local_rank
    if world > 1:
        from npbnn_amd import _capi as capi
        import ctypes
        n_dev = ctypes.c_int(0)
        capi.load_library().npbnn_device_count(ctypes.byref(n_dev))
        if dist_backend in ("rccl", "nccl") and not os.environ.get("NPBNN_BENCH_RCCL_FAULT"):
            if n_dev.value < world:
                sys.exit("[rank %d] --gpus %d over RCCL needs %d GPUs, %d visible (NPBNN_BENCH_DIST_BACKEND=socket rehearses the "
                         "rank flow on fewer)" % (rank, world, world, n_dev.value))
        else:
            device_index = local_rank % max(1, n_dev.value)
        os.environ["NPBNN_DEVICE"] = str(device_index)

    from bench_support import workload
    wl = workload(args.config)
    temps = [1.0] if world == 1 else list(np.linspace(0.8, 1.0, world))          # MC3 defaults (np_bnn/BNN_mc3.py:46-51)
    bnn, mcmc = wl.build(mcmc_id=rank, temperature=temps[rank], randomize_seed=world > 1)

    if os.environ.get("NPBNN_BENCH_SCHEDULE"):            # A/B of the chain schedules on one box (npbnn_chain_cfg.schedule)
        mcmc.device_schedule = int(os.environ["NPBNN_BENCH_SCHEDULE"])
    comm, comm_kind, nranks_seen = None, "none", 1
    if world > 1:
        comm, comm_kind = make_comm(dist_backend, rank, world, local_rank, device_index)
        nranks_seen = int(comm.world_size)

    from npbnn_amd import exchange as ex
    swaps = ex.SwapProposals(max(world, 2), np.random.RandomState(4321))      # the same stream on every rank
    chains, ids = [(bnn, mcmc)], [rank]
    swap_no = [0]
    # NPBNN_BENCH_DEVICE_SWAPS=1: swap intervals in device batches, records all-gathered in place by RCCL on the stream
    # (npbnn_chains_run_exchange).  Not the default for several ranks: that path has only ever met a one-rank communicator
    # on hardware; the default keeps every collective a plain host-side call between two device batches.
    device_swaps = world > 1 and bool(os.environ.get("NPBNN_BENCH_DEVICE_SWAPS")) and getattr(comm, "_comm", None) is not None

    def advance(n_steps):
        """n_steps chain dispatches; with several chains each one ends with the swap proposal."""
        if world == 1:
            for _ in range(n_steps):
                mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
            return
        ex.advance_intervals(chains, ids, world, n_steps, ITERATIONS_PER_STEP, swaps, swap_no[0], comm=comm, batch=20,
                             device=device_swaps)
        swap_no[0] += n_steps

    def sync():
        """barrier + device synchronisation (the driver's contract for both ends of the timed region)"""
        if comm is not None:
            comm.barrier()
        from npbnn_amd import _capi as capi
        capi.check(capi.load_library(), None, capi.load_library().npbnn_device_synchronize(device_index))
        if comm is not None:
            comm.barrier()

    spin_up(mcmc, bnn)
    advance(args.warmup)
    book = dict(acc=mcmc._device_accepted, passes=mcmc._device_passes, voids=mcmc._device_void_passes, its=mcmc._device_iterations)
    sync()
    t0 = time.perf_counter()
    advance(args.steps)
    sync()
    el = time.perf_counter() - t0
    per_rank = None
    if comm is not None:
        # the slowest rank's clock prices the job; every rank's own clock, rate and acceptance go into the line beside it, so that
        # the first run on real multi-GPU hardware can be read rank by rank
        mine = np.array([el, float(mcmc._device_accepted) / max(1, mcmc._device_iterations), float(mcmc._device_schedule_used)])
        all_ranks = comm.allgather_f64(mine)
        el = float(np.max(all_ranks[:, 0]))
        per_rank = [{"rank": r, "seconds": float(all_ranks[r, 0]), "value": args.steps * ITERATIONS_PER_STEP / float(all_ranks[r, 0]),
                     "accept_rate": float(all_ranks[r, 1]), "schedule": int(all_ranks[r, 2])} for r in range(all_ranks.shape[0])]

    line = None
    if rank == 0:
        its = args.steps * ITERATIONS_PER_STEP
        line = report(args, wl, bnn, mcmc, world, el, its, book, comm_kind, nranks_seen, device_swaps)
        if per_rank is not None:
            line["per_rank"] = per_rank
    if world > 1 and not os.environ.get("NPBNN_BENCH_NO_ROW_SHARD"):
        # after the timed region: the other way several GPUs serve this path - ONE chain, its rows split over the ranks.  Nothing in
        # here may cost the headline line: every failure is caught and reported inside it, on every rank
        leg, why = _with_deadline(lambda: row_sharded_chain(comm, rank, world, device_index),
                                  float(os.environ.get("NPBNN_BENCH_LEG_DEADLINE", "240")))
        if leg is None:
            leg = {"error": why}
        if line is not None:
            line["row_sharded_chain"] = leg
        if why is not None and why.startswith("not back"):
            # a rank is still inside the leg (a collective that never completed): the line goes out, and the process ends without
            # the closing barrier that would wait for that rank
            if line is not None:
                print(json.dumps(line), flush=True)
            sys.stderr.write("[bench rank %d] row-sharded leg: %s; ending without the closing barrier\n" % (rank, why))
            sys.stderr.flush()
            # A rank is wedged: a launcher or CI that goes by exit codes must see it (NPBNN_BENCH_STRICT=1: exit code 3).  By default
            # the process still ends with 0 - the headline measurement of the timed region is complete and printed, the line itself
            # carries the failed leg ("row_sharded_chain": {"error": ...}), and a record that is discarded over a secondary leg
            # is worth less than one that says what happened.
            os._exit(3 if os.environ.get("NPBNN_BENCH_STRICT") else 0)
    if line is not None:
        print(json.dumps(line), flush=True)
    if comm is not None:
        try:
            comm.barrier()
            comm.close()
        except Exception as e:      # noqa: BLE001 - (a rank that failed in the leg above: the line is out, end quietly)
            print("[bench rank %d] closing the communicator: %s" % (rank, e), file=sys.stderr)
    try:
        mcmc._backend.close()
    except Exception:       # noqa: BLE001 - (already closed by the report)
        pass


def row_sharded_chain(comm, rank, world, device_index, config=4, n_steps=10):
    """ONE chain of BASELINE config 4 (1M rows) with its rows split over the ranks (MCMC(row_comm=...), npbnn_set_row_shard): every
    rank evaluates its share, the per-pass records are all-gathered (RCCL on the chain's stream when the communicator is the C
    ABI's, else through the host) and every rank takes the same decision.  Dispatches of 100 iterations, the slowest rank's clock."""
    import numpy as np
    from bench_support import workload
    from npbnn_amd import _capi as capi
    from npbnn_amd.rowshard import shard_bounds
    wl = workload(config)
    lo, hi = shard_bounds(wl.n, rank, world)
    bnn, mcmc = wl.build(rows=(lo, hi), row_comm=comm)
    lib = capi.load_library()
    spin_up(mcmc, bnn)
    for _ in range(3):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    book = dict(passes=mcmc._device_passes, its=mcmc._device_iterations, acc=mcmc._device_accepted)
    comm.barrier()
    capi.check(lib, None, lib.npbnn_device_synchronize(device_index))
    t0 = time.perf_counter()
    for _ in range(n_steps):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    capi.check(lib, None, lib.npbnn_device_synchronize(device_index))
    el = float(np.max(comm.allgather_f64(np.array([time.perf_counter() - t0]))))
    state = comm.allgather_f64(np.concatenate([[float(mcmc._logLik)], np.concatenate([w.ravel() for w in bnn._w_layers])]))
    done = mcmc._device_iterations - book["its"]
    be = mcmc._backend
    out = {"workload": wl.description + "; ONE chain, rows split over %d ranks" % world, "value": n_steps * ITERATIONS_PER_STEP / el,
           "unit": "iterations/s", "ms_per_step": 1e3 * el / n_steps, "ranks": world, "rows_per_rank": be.rows_per_rank,
           "schedule": int(mcmc._device_schedule_used), "iterations_per_pass": done / max(1, mcmc._device_passes - book["passes"]),
           "accept_rate": float(mcmc._device_accepted - book["acc"]) / max(1, done),
           "gather": "ncclAllGather of the per-pass records on the chain's stream" if be.rccl_handle() else "through the host (communicator without a device handle)",
           "loglik": float(mcmc._logLik), "ranks_hold_the_same_chain": bool(np.all(state == state[0])),
           "note": "kernel boundaries: pass on this rank's rows, record, gather, step (npbnn_chain_run on a context with npbnn_set_row_shard)"}
    be.close()
    return out


def _with_deadline(fn, seconds):
    """``fn()`` on a helper thread: (result, None), or (None, reason) when it raised or is not back within ``seconds`` (a collective
    that never completes must not take the line with it; the thread is left behind, the caller ends the process by ``os._exit``)."""
    import threading
    box = {}

    def run():
        try:
            box["out"] = fn()
        except BaseException as e:      # noqa: BLE001
            box["err"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return None, "not back within %.0f s" % seconds
    if "err" in box:
        return None, box["err"]
    return box["out"], None


def make_comm(dist_backend, rank, world, local_rank, device_index):
    """The communicator of the swap exchange (see main).  A rank that cannot build it fails loudly; the launcher ends the others."""
    from npbnn_amd.comm import RcclComm, SocketComm
    if dist_backend == "rccl":
        # The RCCL communicator has to come up on EVERY rank or on none: the ranks tell each other over a TCP channel how theirs
        # went (built and one all-gather through, within the deadline) and, if any of them failed, all carry the swap exchange
        # over that channel instead - a few dozen bytes per swap interval, not a data path - and the line says so.
        import numpy as np
        ctl = SocketComm(rank=rank, world_size=world, timeout=400.0)

        def bring_up():
            if os.environ.get("NPBNN_BENCH_RCCL_FAULT"):          # (rehearsal of the fallback on a box with fewer GPUs than ranks)
                raise RuntimeError("NPBNN_BENCH_RCCL_FAULT is set")
            c = RcclComm(rank=rank, world_size=world, device=device_index)
            c.barrier()
            return c
        comm, why = _with_deadline(bring_up, float(os.environ.get("NPBNN_BENCH_RCCL_DEADLINE", "150")))
        ok = ctl.allgather_f64(np.array([0.0 if comm is None else 1.0]))[:, 0]
        if bool(np.all(ok == 1.0)):
            ctl.close()
            return comm, comm.describe()
        if comm is None:
            print("[bench rank %d] RCCL communicator: %s" % (rank, why), file=sys.stderr)
        else:
            comm._comm = None          # (left as it is: tearing down a communicator whose peers never joined may not return)
        failed = [int(r) for r in np.nonzero(ok != 1.0)[0]]
        return ctl, ("tcp sockets through rank 0, %d ranks - the RCCL communicator did not come up on rank(s) %s%s"
                     % (world, failed, (": " + why) if comm is None else ""))
    if dist_backend == "socket":
        return SocketComm(rank=rank, world_size=world), "tcp sockets through rank 0 (rehearsal; no RCCL), %d ranks" % world
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    from torch_dist_comm import TorchDistComm
    if dist_backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(dist_backend)
    return TorchDistComm(), "%s (torch.distributed), %d ranks" % ("rccl" if dist_backend == "nccl" else dist_backend, world)


SCHEDULES = {1: "serial: evaluate a pass, decide it, evaluate the next",
             2: "overlapped: the launch that evaluates pass L also decides pass L-1 (one workgroup); a pass overtaken by an "
                "accept is dropped and re-evaluated",
             3: "overlapped, launches alternating between two streams (opt-in)",
             4: "overlapped, persistent: one launch per batch round, its workgroups loop over the passes (a workgroup that is "
                "through with pass L starts pass L+1), device-side flags order what kernel boundaries used to; bounded waits, "
                "falls back to (2) on a time-out",
             5: "persistent launch with the decision between the passes: the step workgroup prepares the next pass for every outcome "
                "of the pass in flight, then decides it and raises a flag; no pass is evaluated in vain"}


def profiled_kernel_ms(config, cand):
    """Mean duration (ms) of the pass kernel with ``cand`` candidates in the newest rocprofv3 --kernel-trace --stats summary under
    profiles/ for this config, or (None, reason)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg%d_pass%d_kernel_stats.csv" % (config, cand))), reverse=True)
    for path in files:
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"eval_kernel<\d+, \d+, \w+, (\d+),", row.get("Name", "") or row.get("Kernel_Name", ""))
                if m and int(m.group(1)) == cand:
                    for key in ("AverageNs", "Average", "AverageDuration"):
                        if row.get(key):
                            return float(row[key]) * 1e-6, os.path.relpath(path, ROOT)
    return None, "no profiles/*_cfg%d_pass%d_kernel_stats.csv" % (config, cand)


def served_from(config, cand):
    """Where the pass kernel's bytes come from, from the newest profiles/r*_dram_vs_mall.csv: the same columns and
    network timed at row counts below and above what the 256-MiB Infinity Cache holds (tools/dram_vs_mall.py).  No counter rocprofv3
    exposes here separates the two - TCC_EA0_RDREQ_DRAM counts every request that leaves the L2 - so the clock is asked instead."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dram_vs_mall.csv")), reverse=True)
    cols = {2: "256", 0: "256", 4: "64"}.get(config)
    if not files or cols is None:
        return None
    rows = []
    with open(files[0], newline="") as fh:
        for r in csv.DictReader(fh):
            if r["columns"] == cols and int(r["candidates"]) == (1 if cand == 1 else 3):
                rows.append(r)
    if not rows:
        return None
    fits = [r for r in rows if r["fits_infinity_cache"] == "yes" and float(r["MB_of_X"]) >= 100.0]
    beyond = [r for r in rows if r["fits_infinity_cache"] == "no"]
    out = {"source": os.path.relpath(files[0], ROOT), "candidates": 1 if cand == 1 else 3,
           "us_per_100MB_by_MB_of_X": {r["MB_of_X"]: float(r["us_per_100MB"]) for r in rows},
           "TBps_beyond_the_infinity_cache": max(float(r["TB_per_s"]) for r in beyond) if beyond else None,
           "infinity_cache_lds_dma_TBps_measured": 10.4, "infinity_cache_source": "profiles/r05_microbench_ingest.txt (16 waves per CU, 96 MB region)"}
    if fits and beyond:
        a, b = float(fits[0]["us_per_100MB"]), min(float(r["us_per_100MB"]) for r in beyond)
        out["verdict"] = ("X of this configuration fits the Infinity Cache, so its bytes may be served from there (FETCH_SIZE counts those reads too); "
                          "the same kernel takes %.1f us per 100 MB on it and %.1f us per 100 MB on matrices 2-8 x larger, which HBM must serve: "
                          "it is not bound by where the bytes come from, and the HBM peak is the roof that applies in either case" % (a, b))
    return out


def kernel_roofline(config, wl, ctx, bnn, mcmc, useful=None):
    """The dominant kernel of a chain - the pass kernel: one launch streams X once and evaluates ``cand`` proposals against it -
    timed live with HIP events on the chain's stream (npbnn_time_pass), priced with the HBM bytes the PMC passes under profiles/
    recorded for it."""
    ms_kernel, cand = ctx.time_pass(bnn._w_layers, n_candidates=mcmc.n_candidates, iters=200)
    ms_single, _ = ctx.time_eval(bnn._w_layers, iters=200)
    l0 = ctx.l0_mode()
    alg = wl.bytes_per_proposal
    traffic, traffic_src = measured_traffic(config, cand)
    physical = traffic if traffic is not None else alg          # (no counter file: one read of X is the model)
    achieved = physical / (ms_kernel * 1e-3)
    ms_prof, prof_src = profiled_kernel_ms(config, cand)
    out = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
           "frac": achieved / HBM_PEAK,
           "frac_live": achieved / HBM_PEAK,
           "frac_profiles": (physical / (ms_prof * 1e-3) / HBM_PEAK) if ms_prof else None,
           "kernel_ms_profiles": ms_prof, "kernel_ms_profiles_source": prof_src,
           "traffic": traffic, "traffic_source": traffic_src,
           "traffic_note": "a RECORDED measurement (separate rocprofv3 --pmc passes of the round the file name carries), not re-measured "
                           "by this run; kernel_ms is measured live",
           "kernel": wl.kernel_name(ctx, cand), "kernel_ms": ms_kernel,
           "note": "achieved = HBM bytes one launch moves (the PMC traffic; one streaming read of X whatever the number of "
                   "candidates) / its mean duration: the share of the HBM peak the kernel really uses.  The reference reads "
                   "X once PER PROPOSAL; that byte model (SURVEY 8d) is under algorithmic_equivalent",
           "algorithmic_equivalent": {"proposals_per_launch": cand, "bytes_per_proposal": alg,
                                      "GBps": alg * cand / (ms_kernel * 1e-3) / 1e9,
                                      "x_hbm_peak": alg * cand / (ms_kernel * 1e-3) / HBM_PEAK},
           "single_candidate_kernel_ms": ms_single,
           "single_candidate_frac": alg / (ms_single * 1e-3) / HBM_PEAK}
    sf = served_from(config, cand)
    if sf is not None:
        out["served_from"] = sf
    if useful is not None:
        out["useful_iterations_per_launch"] = useful
    if l0 == "f16-split":        # the same launches with layer 0 on float32 matrix cores (bit-exact float32 products)
        try:
            ctx.set_l0_precision("f32")
            ms32, cand32 = ctx.time_pass(bnn._w_layers, n_candidates=mcmc.n_candidates, iters=100)
            ms32_single, _ = ctx.time_eval(bnn._w_layers, iters=100)
            out["f32_path"] = {"kernel_ms": ms32, "candidates": cand32, "frac": physical / (ms32 * 1e-3) / HBM_PEAK,
                               "single_candidate_kernel_ms": ms32_single, "single_candidate_frac": alg / (ms32_single * 1e-3) / HBM_PEAK,
                               "note": "NPBNN_L0=f32: v_mfma_f32_16x16x4_f32 for layer 0 instead of three fp16 products per term"}
        finally:
            ctx.set_l0_precision("auto")
            ctx.time_eval(bnn._w_layers, iters=1)      # (back on the default layout before the chain goes on)
    return out, cand


def _moving_summary(mv, wl, roof, bytes_per_proposal=None):
    if not mv:
        return None
    bpp = wl.bytes_per_proposal if wl is not None else bytes_per_proposal
    out = {"value": mv["value"], "unit": "iterations/s", "one_call_of_4000": mv.get("one_call_of_4000"), "accept_rate": mv.get("accept_rate"),
           "iterations_per_pass": mv.get("iterations_per_pass"), "schedule": mv.get("schedule")}
    if bpp and roof:
        kernel_ms = roof.get("kernel_ms")
        out["roofline"] = {"bound": "hbm", "kernel": roof.get("kernel"), "kernel_ms": kernel_ms, "frac": roof.get("frac"),
                           "one_read_per_iteration_it_per_s": HBM_PEAK / bpp,
                           "value_over_one_read_per_iteration": mv["value"] / (HBM_PEAK / bpp)}
    return out


def _drop_the_last_leg():
    """The objects of the leg before (a model with its 100-260 MB feature matrix, its chain) sit in reference cycles: left to the
    cyclic collector they are torn down whenever its counters say so, i.e. inside the next leg's dispatches.  Collect between the
    legs instead.  (Round 3 blamed the collector for the one 80 ms dispatch early in config 5's leg; round 4 found the device's clock
    change after the host-only parity check - see spin_up - but an untimely collection is no better.)"""
    gc.collect()


def spin_up(mcmc, bnn, seconds=0.25):
    """Keep the GPU busy for a quarter of a second before a leg's warm-up steps, with launches of the leg's own pass kernel that
    touch nothing of the chain (npbnn_time_pass).  After seconds of host-only work (data generation, the CPU baseline, a parity
    check) the device has clocked down, and the clock change that follows the first new launches stalls ONE dispatch by 70-80 ms
    somewhere in the first ten - seen as config 5 at 18-20 k instead of 62 k it/s whenever it ran behind config 4's parity
    check (NPBNN_CHAIN_TIMING=1: the stall is inside the library's wait for the stream) and never when it ran first.  The
    driver's W warm-up steps (5 x 1 ms) are too short to absorb it; the timed region is untouched."""
    be = mcmc._backend
    ctx = getattr(be, "ctx", None) or getattr(getattr(be, "_inner", None), "ctx", None)
    if ctx is None:
        return
    try:
        ms, _ = ctx.time_pass(bnn._w_layers, n_candidates=mcmc.n_candidates, iters=50)
        ctx.time_pass(bnn._w_layers, n_candidates=mcmc.n_candidates, iters=max(50, int(seconds * 1e3 / max(ms, 1e-3))))
    except Exception as e:      # noqa: BLE001 - a courtesy to the clocks, never a reason to lose the measurement
        print("[bench] spin-up skipped: %s" % e, file=sys.stderr)


def moving_chain(wl):
    """A chain that moves: the same model with proposals small enough that about a quarter of them is accepted (the reference
    drivers adapt towards 0.2-0.4: adapt_f / adapt_fM, np_bnn/BNN_env.py:392-413)."""
    if wl.moving_update_f is None:
        return None
    _drop_the_last_leg()
    bnn_q, mcmc_q = wl.build(update_f=list(wl.moving_update_f))
    spin_up(mcmc_q, bnn_q)
    mcmc_q.run_steps(bnn_q, 2000)
    t0 = time.perf_counter()
    for _ in range(20):
        mcmc_q.run_steps(bnn_q, ITERATIONS_PER_STEP)
    el4 = time.perf_counter() - t0
    t0 = time.perf_counter()
    mcmc_q.run_steps(bnn_q, 4000)
    el5 = time.perf_counter() - t0
    from npbnn_amd import _capi as capi
    ctx = mcmc_q._backend.ctx
    out = {"value": 20 * ITERATIONS_PER_STEP / el4, "unit": "iterations/s", "one_call_of_4000": 4000 / el5,
           "us_per_dispatch_of_100": 1e6 * el4 / 20, "us_fixed_per_dispatch": 1e6 * (el4 / 20 - ITERATIONS_PER_STEP * el5 / 4000),
           "auto_turn_us": {"overlapped": ctx.info(capi.INFO_TURN_NS_OVERLAPPED) / 1e3,
                            "decision_between_passes": ctx.info(capi.INFO_TURN_NS_BETWEEN) / 1e3},
           "auto_us_per_iteration": {"overlapped": ctx.info(capi.INFO_IT_NS_OVERLAPPED) / 1e3,
                                     "decision_between_passes": ctx.info(capi.INFO_IT_NS_BETWEEN) / 1e3},
           "accept_rate_last_100": float(mcmc_q._acceptance_rate),
           "accept_rate": float(mcmc_q._device_accepted) / max(1, mcmc_q._device_iterations),
           "schedule": int(mcmc_q._device_schedule_used),
           "iterations_per_pass": mcmc_q._device_iterations / max(1, mcmc_q._device_passes),
           "note": "20 dispatches of %d iterations after 2000 of warm-up, then one call of 4000; update_f = %s"
                   % (ITERATIONS_PER_STEP, list(wl.moving_update_f))}
    mcmc_q._backend.close()
    return out


def other_config(args, config):
    """BASELINE.json's other single-GPU configurations, timed the way the headline is (same step, same warm-up, from the freshly
    initialised chain) so that the driver's run covers them."""
    from bench_support import workload
    _drop_the_last_leg()
    wl = workload(config)
    bnn, mcmc = wl.build()
    spin_up(mcmc, bnn)
    for _ in range(args.warmup):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    book = dict(acc=mcmc._device_accepted, passes=mcmc._device_passes, voids=mcmc._device_void_passes, its=mcmc._device_iterations)
    t0 = time.perf_counter()
    per_step = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
        per_step.append(time.perf_counter() - t1)
    el = time.perf_counter() - t0
    passes = max(1, mcmc._device_passes - book["passes"])
    voids = mcmc._device_void_passes - book["voids"]
    done = mcmc._device_iterations - book["its"]
    roof, cand = kernel_roofline(config, wl, mcmc._backend.ctx, bnn, mcmc, useful=done / (passes + voids))
    acc_rate = float(mcmc._device_accepted - book["acc"]) / max(1, done)
    t0 = time.perf_counter()
    mcmc.run_steps(bnn, 4000)          # (before the parity check: the oracle's BLAS threads keep the host cores busy for a while after it,
    one_call = 4000 / (time.perf_counter() - t0)      #  and config 5's proposals cost the one pre-draw thread 10 us per iteration)
    out = {"workload": wl.description, "bytes_per_proposal": wl.bytes_per_proposal, "value": args.steps * ITERATIONS_PER_STEP / el, "unit": "iterations/s",
           "ms_per_step": 1e3 * el / args.steps, "roofline_it_per_s_one_read_per_proposal": HBM_PEAK / wl.bytes_per_proposal,
           "accept_rate": acc_rate, "iterations_per_pass": done / passes,
           "schedule": int(mcmc._device_schedule_used), "candidates_per_pass": cand, "layer0": mcmc._backend.ctx.l0_mode(),
           "fp16_columns_with_moved_scale": list(mcmc._backend.ctx.f16_moved_columns()),
           "roofline": roof, "parity": wl.parity(bnn, mcmc), "one_call_of_4000": one_call,
           "ms_per_step_median": 1e3 * sorted(per_step)[len(per_step) // 2], "ms_slowest_step": 1e3 * max(per_step),
           "slowest_step_index": per_step.index(max(per_step))}
    mcmc._backend.close()
    mv = moving_chain(wl)
    if mv is not None:
        out["moving_chain"] = mv
    return out


MFMA_F16_PEAK = 2.5e15      # dense fp16 matrix-core peak of an MI355X (MI355X_MICROARCH.md); the fp16-split path spends 3 products per term
MFMA_F32_PEAK = 157.3e12


def wide_config(args):
    """The weight-streamed path (networks the LDS cannot hold): the chain the way the headline is timed, and the first layer's
    product - a real contraction - against the matrix cores' roofline."""
    from bench_support import workload
    _drop_the_last_leg()
    wl = workload(9)
    bnn, mcmc = wl.build()
    ctx = mcmc._backend.ctx
    ms0, ms_pass, geo = ctx.time_wide(bnn._w_layers, iters=100)          # (also the spin-up)
    ms0, ms_pass, geo = ctx.time_wide(bnn._w_layers, iters=200)
    for _ in range(args.warmup):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    el = time.perf_counter() - t0
    f16 = ctx.l0_mode() == "f16-split"
    products = 3 if f16 else 1
    peak = MFMA_F16_PEAK if f16 else MFMA_F32_PEAK
    achieved = products * wl.flops_layer0 / (ms0 * 1e-3)
    recorded = recorded_wide_counters()
    out = {"workload": wl.description, "path": "weight-streamed" if ctx.is_wide() else "resident", "layer0": ctx.l0_mode(),
           "value": args.steps * ITERATIONS_PER_STEP / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / args.steps,
           "value_note": "the reference's default proposal perturbs 5 %% of every layer: %d of this network's %d weights per iteration - the "
                         "chain is paced by the host's numpy-identical pre-draw of those entries and by the step's walk over them, not by "
                         "the pass (pass_us)" % (int(sum(mcmc._update_n)), int(bnn._n_params)),
           "pass_us": 1e3 * ms_pass, "layer0_product_us": 1e3 * ms0, "layer0_block": geo,
           "roofline": {"bound": "mfma", "kernel": "wide_gemm_kernel<8,4,2,4,%s> (+ wide_reduce_kernel)" % ("fp16-split" if f16 else "f32"),
                        "kernel_ms": ms0, "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                        "algorithmic_TFLOPs": wl.flops_layer0 / (ms0 * 1e-3) / 1e12,
                        "note": "achieved = matrix-core flops of the first layer's product (2 N F H0, x 3 on the fp16-split path: "
                                "wh.xh + wl.xh + wh.xl) / its duration (HIP events around back-to-back launches, npbnn_time_wide); peak = dense "
                                "fp16 matrix-core rate.  The same kernel against HBM: hbm_frac (one read of X)",
                        "hbm_frac": wl.bytes_per_proposal / (ms0 * 1e-3) / HBM_PEAK,
                        "traffic": recorded.get("traffic"), "mfma_busy": recorded.get("mfma_busy"), "recorded_from": recorded.get("source")},
           "accept_rate": float(mcmc._device_accepted) / max(1, mcmc._device_iterations),
           "schedule": int(mcmc._device_schedule_used), "parity": wl.parity(bnn, mcmc)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, budget_s=5.0)
    mcmc._backend.close()
    return out


def wide_fused_config(args):
    """The reference's default network on a thousand features: the weight-streamed path's fused passes (one launch per pass, up to three
    candidates per read of X).  Bound by the intake of X: priced against the HBM peak like the resident kernels."""
    from bench_support import workload
    _drop_the_last_leg()
    wl = workload(8)
    bnn, mcmc = wl.build()
    ctx = mcmc._backend.ctx
    spin_up(mcmc, bnn)
    for _ in range(args.warmup):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    book = dict(passes=mcmc._device_passes, its=mcmc._device_iterations, acc=mcmc._device_accepted)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
    el = time.perf_counter() - t0
    done = mcmc._device_iterations - book["its"]
    passes = max(1, mcmc._device_passes - book["passes"])
    ms_pass, cand = ctx.time_pass(bnn._w_layers, n_candidates=mcmc.n_candidates, iters=200)
    ms_one, _ = ctx.time_pass(bnn._w_layers, n_candidates=1, iters=200)
    t0 = time.perf_counter()
    mcmc.run_steps(bnn, 2000)
    one_call = 2000 / (time.perf_counter() - t0)
    out = {"workload": wl.description, "path": "weight-streamed" if ctx.is_wide() else "resident", "layer0": ctx.l0_mode(),
           "value": args.steps * ITERATIONS_PER_STEP / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / args.steps, "one_call_of_2000": one_call,
           "candidates_per_pass": cand, "iterations_per_pass": done / passes, "accept_rate": float(mcmc._device_accepted - book["acc"]) / max(1, done),
           "schedule": int(mcmc._device_schedule_used),
           "roofline": {"bound": "hbm", "kernel": "wide_gemm_kernel<2,4,8,1,%s,D=%d> (fused pass: product + narrow layers + likelihood)" % ("fp16-split" if ctx.l0_mode() == "f16-split" else "f32", cand),
                        "kernel_ms": ms_pass, "achieved": wl.bytes_per_proposal / (ms_pass * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": wl.bytes_per_proposal / (ms_pass * 1e-3) / HBM_PEAK, "traffic": recorded_fused_counters(cand).get("traffic"),
                        "recorded": recorded_fused_counters(cand),
                        "single_candidate_kernel_ms": ms_one, "single_candidate_frac": wl.bytes_per_proposal / (ms_one * 1e-3) / HBM_PEAK,
                        "note": "achieved = one read of X (algorithmic bytes of a proposal) / the pass's duration: %d candidates share that read" % cand,
                        "roofline_it_per_s_one_read_per_proposal": HBM_PEAK / wl.bytes_per_proposal},
           "parity": wl.parity(bnn, mcmc)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, budget_s=5.0)
    mcmc._backend.close()
    return out


def recorded_fused_counters(cand):
    """Traffic and matrix-core occupancy of the fused pass with ``cand`` candidates from the newest profiles/r*_widefused_pass<cand>_counters.json
    (tools/collect_wide_profiles.sh; a recorded measurement, not re-measured by this run)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_widefused_pass%d_counters.json" % cand)), reverse=True)
    if not files:
        return {}
    with open(files[0]) as fh:
        rec = json.load(fh)
    return {"traffic": rec.get("traffic"), "mfma_busy": rec.get("mfma_busy"), "l2_hit_rate": rec.get("l2_hit_rate"),
            "kernel_ms_profiles": (rec.get("kernel_ns_rocprof") or 0.0) * 1e-6 or None, "source": os.path.relpath(files[0], ROOT)}


def recorded_wide_counters():
    """What the rocprofv3 --pmc passes under profiles/ recorded for the first layer's product of the wide leg (a RECORDED measurement of
    the round the file name carries): HBM bytes per launch and the share of its cycles the matrix cores were busy."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_wide_layer0_counters.json")), reverse=True)
    for path in files:
        try:
            with open(path) as fh:
                d = json.load(fh)
            d["source"] = os.path.relpath(path, ROOT)
            return d
        except Exception:       # noqa: BLE001
            continue
    return {}


def group_pass_leg():
    """Three chains of config 2's model on ONE GPU sharing their passes over the data (group passes, npbnn_chains_run_batched: one
    proposal per chain per streaming read of X), at the default proposal size and at one a quarter of whose
    proposals is accepted; aggregate iterations/s over the three chains, beside one such chain alone."""
    import numpy as np
    import npbnn_amd as bn
    from npbnn_amd import exchange as ex
    from bench_support import _quiet
    _drop_the_last_leg()
    rs = np.random.default_rng(0)
    n, f, c = 100_000, 256, 10
    x = rs.standard_normal((n, f)).astype(np.float32)
    proj = rs.standard_normal((f, c)) / np.sqrt(f)
    y = np.argmax(x @ proj + 0.5 * rs.standard_normal((n, c)), axis=1)          # learnable labels: the proposal size sets the rate
    dat = dict(data=x, labels=y, test_data=np.zeros((0, f)), test_labels=np.zeros(0))
    out = {"workload": "config-2 shapes (100k x 256, [32,8], tanh), learnable labels; chains of one model sharing the resident matrix",
           "note": "20 rounds of %d iterations per chain after 1500 of warm-up; value = aggregate over the chains" % ITERATIONS_PER_STEP}
    for uf in (0.05, 0.002):
        row = {}
        for n_chains in (1, 3):
            chains = []
            for i in range(n_chains):
                np.random.seed(1234 + i)
                bnn = _quiet(bn.npBNN, dat, n_nodes=[32, 8], actFun=bn.ActFun(fun="tanh"), use_bias_node=2, prior_f=1, p_scale=1)
                chains.append((bnn, bn.MCMC(bnn, update_f=[uf] * 3, mcmc_id=i, randomize_seed=True)))
            for bnn, m in chains:
                m.run_steps(bnn, 1500)
            for rep in range(2):
                t0 = time.perf_counter()
                for _ in range(20):
                    if n_chains == 1:
                        chains[0][1].run_steps(chains[0][0], ITERATIONS_PER_STEP)
                    else:
                        ex.run_steps_batched(chains, ITERATIONS_PER_STEP)
                el = time.perf_counter() - t0
            row["one chain" if n_chains == 1 else "group pass of 3"] = {
                "value": n_chains * 20 * ITERATIONS_PER_STEP / el, "unit": "iterations/s (aggregate)",
                "accept_rate": float(np.mean([m._acceptance_rate for _, m in chains]))}
            for bnn, m in chains:
                m._backend.close()
            del chains
            _drop_the_last_leg()
        out["update_f %.3f" % uf] = row
    return out


def report(args, wl, bnn, mcmc, world, el, its, book, comm_kind, nranks_seen, device_swaps):
    from npbnn_amd import _capi as capi
    ctx = mcmc._backend.ctx
    passes = max(1, mcmc._device_passes - book["passes"])
    voids = mcmc._device_void_passes - book["voids"]
    done = mcmc._device_iterations - book["its"]
    used = getattr(mcmc, "_device_schedule_used", 0)
    accept_rate = float(mcmc._device_accepted - book["acc"]) / max(1, done)
    accept_last = float(mcmc._acceptance_rate)
    loglik = float(mcmc._logLik)
    roof, cand = kernel_roofline(args.config, wl, ctx, bnn, mcmc, useful=done / (passes + voids))
    line = {
        "metric": "MCMC iterations/sec (full fwd+lik per proposal)",
        "value": world * its / el,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16x3-split/f32" if ctx.l0_mode() == "f16-split" else "f32",
        "dtype_note": "layer 0 (and layer 1 of tanh networks) as three fp16 matrix-core products per term on hi/lo pairs of the float32 "
                      "operands (~22 significant bits), float32 accumulation; later layers and the likelihood terms float32; every "
                      "cross-row sum float64.  roofline.f32_path: the same kernel with float32 matrix cores",
        "data": "synthetic",
        "config": {"workload": wl.description + "; one chain per GPU",
                   "step": "%d Metropolis-Hastings iterations of every chain (one MC3 chain dispatch, np_bnn/BNN_mc3.py:80-85), state "
                           "back on the host after each; timed from the freshly initialised chain" % ITERATIONS_PER_STEP,
                   "iterations_per_step": ITERATIONS_PER_STEP, "timed_iterations_per_chain": its,
                   "chains": world, "swap_frequency": ITERATIONS_PER_STEP if world > 1 else None,
                   "swap_exchange": comm_kind, "swap_exchange_nranks": nranks_seen,
                   "swap_exchange_path": ("none" if world == 1 else
                                          "device: swap intervals in batches on the stream, records all-gathered in place" if device_swaps
                                          else "host: one device batch per interval, all-gather of [logPost, temperature] and decision on the host"),
                   "layer0": ctx.l0_mode(), "fast_tails": bool(ctx.info(capi.INFO_FAST_TAILS)),
                   "loop": "device-resident chain (npbnn_chain_run), proposals pre-drawn on the host",
                   "candidates_per_pass": cand, "iterations_per_pass": done / passes, "schedule": SCHEDULES.get(used, "host loop"),
                   "void_pass_fraction": voids / (passes + voids)},
        "roofline": roof,
        "accept_rate": accept_rate,          # rank 0, timed region
        "accept_rate_last_100": accept_last,
        "loglik": loglik,
        "schedule_timeouts": int(ctx.sync_fallbacks),
        "torch_imported": "torch" in sys.modules,
    }
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(wl)
        line["parity"] = wl.parity(bnn, mcmc)
        # the same chain once it has settled (acceptance of the default proposal falls to a few per cent): steady-state rate
        n_more = 40
        mcmc.run_steps(bnn, 3000)
        t0 = time.perf_counter()
        for _ in range(n_more):
            mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
        el2 = time.perf_counter() - t0
        line["settled_chain"] = {"value": n_more * ITERATIONS_PER_STEP / el2, "unit": "iterations/s",
                                 "after_iterations": int(mcmc._current_iteration), "accept_rate_last_100": float(mcmc._acceptance_rate),
                                 "note": "same dispatches of %d iterations, measured after 3000 further iterations" % ITERATIONS_PER_STEP}
        # what a dispatch costs beyond its iterations: the same chain in ONE call of 10 000 iterations against the dispatches above
        c0 = ctx.seconds_in_chain_run
        t0 = time.perf_counter()
        for _ in range(n_more):
            mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
        el2b = time.perf_counter() - t0
        inside = ctx.seconds_in_chain_run - c0
        t0 = time.perf_counter()
        mcmc.run_steps(bnn, 10000)
        el3 = time.perf_counter() - t0
        line["call_cost"] = {"long_call_iterations_per_s": 10000 / el3, "us_per_iteration_in_a_long_call": 1e2 * el3,
                             "us_per_dispatch_of_%d" % ITERATIONS_PER_STEP: 1e6 * el2 / n_more,
                             "us_fixed_per_dispatch": 1e6 * el2 / n_more - ITERATIONS_PER_STEP * 1e2 * el3,
                             "us_of_python_per_dispatch": 1e6 * (el2b - inside) / n_more,
                             "note": "upload of state and draws, first step kernel (full prior re-sum), result copy, synchronisation, host Python"}
        # a pass INSIDE the persistent launch (the form the chain runs): wall time of the long call's 512-iteration batches over their
        # turns - tiles, sums, the hand-over to the step and back; no launch of its own, so no ramp-up or ramp-down
        turn_us = ctx.info(capi.INFO_TURN_NS_OVERLAPPED) / 1e3
        if turn_us > 0 and roof.get("traffic") and int(mcmc._device_schedule_used) == 4:
            roof["pass_inside_the_persistent_launch"] = {
                "us": turn_us, "frac": roof["traffic"] / (turn_us * 1e-6) / 1e9 / roof["peak"],
                "note": "a batch's wall time over its turns (decided and void passes), running mean of the library (NPBNN_INFO_TURN_NS_OVERLAPPED) "
                        "after the long call; the same HBM bytes per pass as a launch of the pass kernel alone, which is what frac is on"}
        if "f32_path" in roof:          # the same chain with layer 0 on float32 matrix cores: a rate, not only kernel times
            try:
                ctx.set_l0_precision("f32")
                for _ in range(args.warmup):
                    mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
                el32 = time.perf_counter() - t0
                roof["f32_path"].update({"value": args.steps * ITERATIONS_PER_STEP / el32, "unit": "iterations/s",
                                         "ms_per_step": 1e3 * el32 / args.steps, "layer0": ctx.l0_mode(),
                                         "value_note": "%d dispatches of %d iterations of the settled chain (compare settled_chain)"
                                                       % (args.steps, ITERATIONS_PER_STEP)})
            finally:
                ctx.set_l0_precision("auto")
                mcmc.run_steps(bnn, ITERATIONS_PER_STEP)
        mv = moving_chain(wl)
        if mv is not None:
            line["moving_chain"] = mv
        if args.config == 2 and not args.only:          # the other single-GPU configurations of BASELINE.json, in the same run
            mcmc._backend.close()
            line["other_configs"] = {}
            for cfg in (4, 5, 0):         # (0: config 2's data under the reference's default network, hidden [50, 5])
                name = "config %d" % cfg if cfg else "default network"
                try:
                    line["other_configs"][name] = other_config(args, cfg)
                except Exception as e:        # noqa: BLE001 - the headline must not be lost to a side measurement
                    line["other_configs"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:       # heavy-tailed features stay on the fp16 pair
                leg = other_config(args, 12)
                leg.pop("moving_chain", None)
                line["other_configs"]["config 2 on log-normal features"] = leg
            except Exception as e:        # noqa: BLE001
                line["other_configs"]["config 2 on log-normal features"] = {"error": "%s: %s" % (type(e).__name__, e)}
            for name, leg in (("wide", lambda: wide_config(args)), ("default network on 1024 features", lambda: wide_fused_config(args)),
                              ("group_pass", group_pass_leg)):
                try:
                    line["other_configs"][name] = leg()
                except Exception as e:        # noqa: BLE001
                    line["other_configs"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
            # the chains that move, side by side at the top of the line: about a quarter of the proposals accepted
            # - the regime the reference's drivers adapt to (bnn_classify.py:61-69, bnn_regress.py:44-51)
            c4 = line["other_configs"].get("config 4") or {}
            mv4 = c4.get("moving_chain")
            if mv4 is None and "value" in c4:       # (config 4's chain moves by itself: its default proposals are accepted 4 times in 10)
                mv4 = {"value": c4["value"], "one_call_of_4000": c4.get("one_call_of_4000"), "accept_rate": c4.get("accept_rate"),
                       "iterations_per_pass": c4.get("iterations_per_pass"), "schedule": c4.get("schedule")}
            roof4 = (line["other_configs"].get("config 4") or {}).get("roofline") or {}
            line["moving_chains"] = {
                "config 2": _moving_summary(line.get("moving_chain"), wl, roof),
                "config 4": _moving_summary(mv4, None, roof4, bytes_per_proposal=(line["other_configs"].get("config 4") or {}).get("bytes_per_proposal")),
                "note": "value: 20 dispatches of %d iterations; roofline: the pass kernel these chains run (same launch as the headline's: "
                        "its candidates share one read of X) and what a DECIDED iteration costs in HBM time - bytes of one read of X over "
                        "the iterations a pass decides on average" % ITERATIONS_PER_STEP}
    else:
        line["cpu_baseline"] = None
    return line


if __name__ == "__main__":
    main()
