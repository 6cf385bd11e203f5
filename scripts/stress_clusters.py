"""Repeated full-length sweeps in a cluster geometry against the one-workgroup-per-tile kernel on
the same inputs: any disagreement beyond rounding is a protocol error of the cluster exchange.
usage: python scripts/stress_clusters.py n p d reps geom [geom ...]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from locus_amd import device as dv


def main():
    n, p, d, reps = map(int, sys.argv[1:5])
    geoms = sys.argv[5:]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.bernoulli(torch.full((p, n), 0.25, device=dev, dtype=torch.float64), generator=g) + \
        torch.bernoulli(torch.full((p, n), 0.25, device=dev, dtype=torch.float64), generator=g)
    X = (X - X.mean(dim=1, keepdim=True)) / X.std(dim=1, keepdim=True).clamp_min(1e-3)
    px = dv.PackedX(X.t())
    del X
    state0 = torch.empty((3, p, d), device=dev, dtype=torch.float64)
    state0[0].uniform_(0, 1, generator=g); state0[0].pow_(8.0)
    state0[1].normal_(0, 0.3, generator=g)
    state0[2] = state0[0] * state0[1]
    R0 = torch.randn((d, n), device=dev, dtype=torch.float64, generator=g) * 0.7
    lo = torch.full((p,), -3.0, device=dev, dtype=torch.float64)
    l1 = torch.full((p,), -0.05, device=dev, dtype=torch.float64)
    tau = torch.full((d,), 2.0, device=dev, dtype=torch.float64)
    s2k = 1.0 / ((n - 1 + 0.7) * tau)
    cons = (s2k * tau, 1.0 / (2 * s2k), -(tau.log() + np.log(0.7) + s2k.log()) / 2, s2k)
    perms = [torch.randperm(p, device=dev, generator=g).to(torch.float64) for _ in range(3)]

    def run(cfg):
        sw = dv.Sweeper(px, d, dev, cfg=cfg)
        st, R = state0.clone(), R0.clone().t()
        stats = torch.empty((4, d), device=dev, dtype=torch.float64)
        for pm in perms:
            sw.identity(pm, lo, l1, *cons, 1.0, R, st, stats)
            sw.status()
        return st, R, stats
    ref = run((-1, 1, 1, 0))
    bad = 0
    for gs in geoms:
        cfg = tuple(map(int, gs.split(",")))
        first = None
        for rep in range(reps):
            got = run(cfg)
            errs = [float(((a - b).abs() / (b.abs() + 1e-3)).max()) for a, b in zip(got, ref)]
            same = first is None or all(torch.equal(a, b) for a, b in zip(got, first))
            first = first or got
            ok = max(errs) < 1e-8
            bad += (not ok) + (not same)
            if not ok or not same or rep == 0:
                print(f"geom {gs} rep {rep}: max rel err state {errs[0]:.2e} R {errs[1]:.2e} stats {errs[2]:.2e}"
                      f" {'OK' if ok else 'WRONG'} {'' if same else 'NOT REPRODUCIBLE'}", flush=True)
    print("disagreements:", bad)
    return 1 if bad else 0


sys.exit(main())
