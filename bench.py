#!/usr/bin/env python
"""bench.py -- beta updates/s of the CAVI iteration on MI355X (BASELINE.json metric).

Workload (default, N = 1): BASELINE config C3, identity link, n = 1000, p = 50 000,
d = 10 000, fp64, synthetic data of the reference's simulation shape
(R/locus.R:152-187), default hyper-parameters / initial values
(R/set_hyper_init.R:321-419, 750-874).  It fits one GPU (X 0.4 GB, state 12 GB).

A "step" is ONE full CAVI iteration of locus_core_ (R/locus_core.R:49-211): hyper-parameter
updates, the coreLoop sweep over all p covariates in a fresh random order, rowSums /
entropy reductions, the packed all-reduce and the ELBO.  value = p * d / step time.
With --gpus N the d responses are sharded over N ranks (X replicated, one RCCL
all-reduce of p + 2 doubles per step): total work is fixed => "scaling": "strong".

The JSON line also carries
  roofline     : the sweep (one call of locus_hip_sweep_dev = block packing + the matrix-core
                 sweep launches that together visit all p x d_loc pairs) against the roofline
                 that binds it: the dense fp64 MFMA peak (78.6 TFLOP/s) with the ALGORITHMIC
                 flops of SURVEY.md §8(d) (4 n + 60 per update).  `traffic` is the PMC-measured
                 HBM bytes per sweep from profiles/ (null if no profile of this shape is
                 committed).
  roofline_hbm : SURVEY.md §8(d)'s HBM figure, algorithmic bytes 16 n + 8 n / d_loc + 32 per
                 update against 8 TB/s.  The residual lives in registers for the whole sweep, so
                 this fraction exceeds 1: the kernel is not HBM-bound.
  cpu_baseline : the C oracle (literal restatement of src/coreLoop.cpp, gcc -O2, 1 thread)
                 timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == matrix peak
ELBO_ROUNDING_REL = 1e-12  # a converged ELBO may still move by this much of its magnitude (rounding)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--p", type=int, default=50000)
    ap.add_argument("--d", type=int, default=10000)
    ap.add_argument("--p0", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--to-tol", type=float, default=0.1,
                    help="after the timed steps keep iterating until |dELBO| < TOL (default 0.1 = "
                         "R/locus.R:243) and report the secondary metric 'time to ELBO tol' "
                         "(0 = skip)")
    ap.add_argument("--cpu-covariates", type=int, default=0,
                    help="covariates of the CPU sample (0 = sized for ~5 s per leg)")
    return ap.parse_args()


def make_data(n, p, d, p0, k_lo, k_hi, seed=123):
    """X (n x p, standardised) and this rank's slice of the centred Y, per SURVEY §8(d)."""
    rng = np.random.default_rng(seed)
    X = rng.binomial(2, 0.25, size=(p, n)).astype(np.float32)  # [p][n]: column j contiguous
    sd = X.std(axis=1, ddof=1)
    for j in np.nonzero(sd == 0)[0]:
        X[j, rng.integers(n)] += 1.0
    X = X.astype(np.float64)
    X -= X.mean(axis=1, keepdims=True)
    X /= X.std(axis=1, ddof=1, keepdims=True)
    d0 = int(0.8 * d)
    ind_p0 = np.sort(rng.choice(p, size=p0, replace=False))
    ind_d0 = np.sort(rng.choice(d, size=d0, replace=False))
    beta = np.where(rng.random((p0, d0)) < 0.2, rng.normal(size=(p0, d0)), 0.0)
    Y = rng.normal(scale=0.5, size=(n, d))
    Y[:, ind_d0] += X[ind_p0].T @ beta
    Y -= Y.mean(axis=0)
    var = Y.var(axis=0, ddof=1)
    return X, np.ascontiguousarray(Y[:, k_lo:k_hi].T), float(np.median(var))


def cpu_baseline(n, d, ncov, seed=7):
    """Oracle coreLoop (oracle/locus_oracle.c) on `ncov` covariates x all d responses: the
    reference's single-threaded path, plus a best-effort all-cores line (the responses are
    independent, so each thread runs the same routine on its own slice of the columns)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    c_oracle.build()
    from locus_amd.simulate import sweep_inputs
    if ncov <= 0:
        # ~6 n d flop and ~5 passes over n x d doubles per covariate; aim at ~5 s per leg
        probe = sweep_inputs(n, 4, d, seed=seed)
        t0 = time.perf_counter()
        _run_oracle(c_oracle, probe)
        per = (time.perf_counter() - t0) / 4
        ncov = int(max(4, min(4096, 5.0 / max(per, 1e-6))))
    inp = sweep_inputs(n, ncov, d, seed=seed)
    t0 = time.perf_counter()
    _run_oracle(c_oracle, inp)
    dt = time.perf_counter() - t0

    # all cores of this job's CPU share (OMP_NUM_THREADS on the GPU box), same sample again
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or os.cpu_count(),
                       os.cpu_count(), d))
    inp = sweep_inputs(n, ncov, d, seed=seed)
    cuts = np.linspace(0, d, cores + 1).astype(int)

    def part(i):
        k0, k1 = cuts[i], cuts[i + 1]
        a = dict(inp)
        for key in ("Y", "gam_vb", "beta_vb", "mat_x_m1", "mu_beta_vb"):
            a[key] = inp[key][:, k0:k1]          # column slices of Fortran arrays: in place
        for key in ("log_tau_vb", "sig2_beta_vb", "tau_vb"):
            a[key] = inp[key][k0:k1]
        _run_oracle(c_oracle, a)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(part, range(cores)))
    dt_all = time.perf_counter() - t0
    return dict(value=ncov * d / dt, unit="beta updates/s", cores=1, kind="port",
                sample=f"oracle coreLoop (gcc -O2, 1 thread) on {ncov} covariates x all {d} "
                       f"responses, n={n} ({dt:.1f} s; cost per covariate does not depend on p); "
                       f"host has {os.cpu_count()} logical CPUs",
                all_cores=dict(value=ncov * d / dt_all, cores=cores,
                               note=f"same sample, responses split over {cores} threads "
                                    f"({dt_all:.1f} s)"))


def _run_oracle(c_oracle, a):
    c_oracle.coreLoop(a["X"], a["Y"], a["gam_vb"], a["log_om_vb"], a["log_1_min_om_vb"],
                      a["log_sig2_inv_vb"], a["log_tau_vb"], a["beta_vb"], a["mat_x_m1"],
                      a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"], 1.0)


def committed_traffic(n, p, d_loc):
    """HBM bytes per sweep launch measured with rocprofv3 --pmc (profiles/traffic_*.json)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.startswith("traffic_") and f.endswith(".json"):
                try:
                    for rec in json.load(open(os.path.join(pdir, f))):
                        if ((rec.get("n"), rec.get("p"), rec.get("d_loc")) == (n, p, d_loc)
                                and "hbm_bytes_per_launch" in rec):
                            best = rec["hbm_bytes_per_launch"]
                except Exception:
                    pass
    return best


def main():
    # LOCUS_BENCH_WATCHDOG=<seconds>: dump every thread's Python stack and exit if the run is still
    # going after that long (a rank stuck in a collective or a kernel names itself instead of hanging)
    if os.environ.get("LOCUS_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LOCUS_BENCH_WATCHDOG"]), exit=True)
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # one rank per GPU; LOCUS_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the
    # multi-rank path on a one-GPU box (RCCL itself refuses two ranks on one device)
    backend = os.environ.get("LOCUS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    from locus_amd import device as dv
    from locus_amd.comm import Comm, DistComm
    from locus_amd.locus_core import FusedTail, IdentityEngine
    dv.require_gpu()
    # LOCUS_BENCH_FORCE_DIST=1 (launched through torch.distributed.run with ONE rank): the process
    # group, DistComm and the packed all-reduce of every step run as they do for N > 1 -- how a
    # one-GPU box executes the RCCL path
    force_dist = os.environ.get("LOCUS_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        comm = DistComm(force_collectives=force_dist)
    else:
        comm = Comm()

    n, p, d, p0 = args.n, args.p, args.d, args.p0
    k_lo, k_hi = comm.shard(d)
    d_loc = k_hi - k_lo

    # ---- synthetic inputs (same on every rank; Y sliced) -----------------------------
    Xpn, Yt, med_var = make_data(n, p, d, p0, k_lo, k_hi)
    X_dev = torch.from_numpy(Xpn).to(device).t()          # n x p, column-major storage
    px = dv.PackedX(X_dev)
    del X_dev, Xpn
    Y_dev = torch.from_numpy(Yt).to(device).t()           # n x d_loc, column-major storage
    del Yt
    # default hyper-parameters / initial values (R/set_hyper_init.R:328-363, 765-804)
    b = d * (p - p0) / p0
    hyper = dict(a=np.ones(p), b=np.full(p, b), eta=np.full(d_loc, 1.0 / med_var),
                 kappa=np.ones(d_loc), nu=1.0)
    hyper["lambda"] = 1e-2
    tau0 = np.full(d_loc, 1.0 / med_var)
    rng = np.random.default_rng(123 + 7919 * rank)
    sig2_beta0 = 1.0 / rng.gamma(2.0, 1e-2 * tau0)
    gen = torch.Generator(device=device)
    gen.manual_seed(123 + rank)
    state = torch.empty((3, p, d_loc), dtype=torch.float64, device=device)
    state[0].uniform_(0, 1, generator=gen)                # gam ~ Beta(1, b) by inversion
    state[0] = -torch.expm1(torch.log1p(-state[0]) / b)
    state[1].normal_(0, 1, generator=gen)                 # mu ~ N(0, 1)
    host = FusedTail(n, p, d, hyper, tau0, sig2_beta0, comm, dv=dv)
    eng = IdentityEngine(px, Y_dev, state, host, device)
    eng.initialise()

    perm_rng = np.random.default_rng(1)                   # same stream on every rank
    ev = []
    nxt = [torch.from_numpy(perm_rng.permutation(p).astype(np.float64)).to(device)]

    def draw_next():                                      # on the host while the sweep runs
        nxt.append(torch.from_numpy(perm_rng.permutation(p).astype(np.float64)).to(device))

    def step():                                           # every step sweeps in a fresh random order
        return eng.iterate(nxt.pop(0), 1.0, with_elbo=True, after_launch=draw_next, c_next=1.0)

    # kernel timing: HIP events around the sweep launch, on the stream it is launched on
    orig_identity = eng.sweeper.identity

    def timed_identity(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_identity(*a, **k)
        e1.record()
        ev.append((e0, e1))
    eng.sweeper.identity = timed_identity

    lbs = []
    for _ in range(args.warmup):
        lbs.append(step())
    comm.barrier()
    torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lbs.append(step())
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        import torch.distributed as dist
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in ev])) if ev else float("nan")

    # secondary metric of BASELINE.json: wall time from the default initial values to |dELBO| < tol
    # = all iterations so far (at the measured step time) + the ones still needed, timed here
    to_tol = None
    eng.sweeper.identity = orig_identity
    if args.to_tol > 0:
        def first_converged():
            for i in range(1, len(lbs)):
                if abs(lbs[i] - lbs[i - 1]) < args.to_tol:
                    return i + 1
            return None
        n_before = len(lbs)
        t1 = time.perf_counter()
        while first_converged() is None and len(lbs) < 500:
            lbs.append(step())
        torch.cuda.synchronize()
        comm.barrier()
        extra = time.perf_counter() - t1
        n_it = first_converged()
        per = dt / args.steps
        to_tol = {"tol": args.to_tol, "iterations": n_it if n_it is not None else len(lbs),
                  "converged": n_it is not None,
                  "seconds": (min(n_it or len(lbs), n_before) * per + extra),
                  "note": "iterations from the default initial values until |dELBO| < tol, at the "
                          "measured time per iteration"}

    if rank == 0:
        conv_tol = args.to_tol if args.to_tol > 0 else 0.1      # R/locus.R:243
        n_conv = next((i + 1 for i in range(1, len(lbs)) if abs(lbs[i] - lbs[i - 1]) < conv_tol),
                      len(lbs))
        # beyond convergence the ELBO may only move by its own rounding: bounded, and part of the flag
        dec_after = float(max(0.0, -np.diff(lbs[n_conv - 1:]).min())) if len(lbs) > n_conv else 0.0
        dec_bound = ELBO_ROUNDING_REL * abs(lbs[-1])
        ms_per_step = dt / args.steps * 1e3
        value = p * d / (dt / args.steps)
        upd_launch = p * d_loc
        bytes_per_update = 16.0 * n + 8.0 * n / d_loc + 32.0
        flops_per_update = 4.0 * n + 60.0
        ach_gbs = upd_launch * bytes_per_update / (kern_ms * 1e-3) / 1e9
        ach_tf = upd_launch * flops_per_update / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "beta updates/sec (p*d per CAVI sweep)", "value": value,
            "unit": "beta updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3 identity-link CAVI iteration n={n} p={p} d={d} p0_av={p0}"
                                   f" (d sharded over {world} GPU(s), X replicated)",
                       "n": n, "p": p, "d": d, "d_per_gpu": d_loc,
                       "parallelism": f"d-shard x{world}",
                       "collectives": (f"{backend} (forced, 1 rank)" if force_dist and world == 1
                                       else backend if use_dist else "none"),
                       "elbo_last": lbs[-1],
                       # the reference checks lb_new + 1.49e-8 >= lb_old while it iterates, i.e. up
                       # to |dELBO| < tol (R/locus_core.R:202-205); the timed steps go on far beyond
                       # that point, where an ELBO of ~1e7 only moves by its own rounding
                       "elbo_monotone": bool(np.all(np.diff(lbs[:n_conv]) > -1.49e-8)
                                             and dec_after <= dec_bound),
                       "elbo_iterations_checked": n_conv,
                       "elbo_max_decrease_after_convergence": dec_after,
                       "elbo_decrease_bound_after_convergence": dec_bound},
            # the roofline that BINDS the sweep: it runs on the fp64 matrix cores with the residual
            # register-resident, SURVEY §8(d)'s flops per update against the dense fp64 MFMA peak
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS,
                         "traffic": committed_traffic(n, p, d_loc),
                         "kernel": "coreLoop sweep = block packing + matrix-core sweep launches "
                                   "(mfma_sweep_kernel<2,KT>: 32 responses per CU; the rest, and "
                                   "every shard of <= 16 responses per CU, mfma_sweep1_kernel: one "
                                   "tile of 16 responses per cluster of 1-2 CUs)",
                         "kernel_ms": kern_ms,
                         "flops_per_update": flops_per_update, "updates_per_launch": upd_launch,
                         "note": "traffic = PMC-measured HBM bytes of one sweep (profiles/)"},
            # SURVEY §8(d)'s HBM figure (algorithmic bytes of the reference's streaming
            # formulation): the residual never leaves the registers, so this exceeds 1
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                             "bytes_per_update": bytes_per_update,
                             "note": "algorithmic bytes 16 n + 8 n / d_loc + 32 per update; "
                                     "frac > 1 because the kernel is not HBM-bound"},
        }
        if to_tol is not None:
            out["time_to_elbo_tol"] = to_tol
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, d, args.cpu_covariates)
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
