"""Wall-clock per CAVI iteration of the drivers at BASELINE sizes (one GPU), with the ELBO trace.
usage: python scripts/perf_drivers.py {c2|c4|c5|z} [iterations | tol]
       "tol": iterate to |dELBO| < 0.1 (R/locus.R:243) and report the time to convergence."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from locus_amd.prepare import add_intercept_, auto_set_hyper_, auto_set_init_
from locus_amd.struct import set_struct


def data(n, p, d, p0, link, seed=123):
    rng = np.random.default_rng(seed)
    X = rng.binomial(2, 0.25, size=(n, p)).astype(np.float64)
    X = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
    act = rng.choice(p, p0, replace=False)
    beta = np.where(rng.random((p0, d)) < 0.2, rng.normal(size=(p0, d)), 0.0)
    Y = X[:, act] @ beta + rng.normal(scale=0.5, size=(n, d))
    if link == "logit":
        Y = (Y > 0).astype(np.float64) - 0.5
    else:
        Y -= Y.mean(axis=0)
    return np.asfortranarray(X), np.asfortranarray(Y)


def main():
    which = sys.argv[1]
    to_tol = len(sys.argv) > 2 and sys.argv[2] == "tol"
    iters = 500 if to_tol else (int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    n, p, d, link = dict(c2=(1000, 20000, 1000, "identity"), c4=(2000, 20000, 500, "logit"),
                         c5=(1000, 50000, 2000, "identity"), z=(1000, 20000, 1000, "identity"))[which]
    p0 = 100
    t0 = time.perf_counter()
    X, Y = data(n, p, d, p0, link)
    hy = auto_set_hyper_(Y, p, p0, q=3 if which == "z" else None, link=link, struct=(which == "c5"))
    ini = auto_set_init_(Y, p, p0, q=3 if which == "z" else None, user_seed=1, link=link)
    print(f"{which}: n={n} p={p} d={d} link={link}; data + defaults {time.perf_counter() - t0:.1f} s",
          flush=True)
    trace, marks = [], []

    class T(list):
        def append(self, v):
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
            super().append(v)
    trace = T()
    kw = dict(tol=0.1 if to_tol else 1e-30, maxit=iters, trace=trace)
    t0 = time.perf_counter()
    if which == "c2":
        from locus_amd.locus_core import locus_core_
        locus_core_(Y, X, hy, ini["gam_vb"], ini["mu_beta_vb"], ini["sig2_beta_vb"], ini["tau_vb"], **kw)
    elif which == "z":
        from locus_amd.locus_z_core import locus_z_core_
        rng = np.random.default_rng(2)
        Z = rng.normal(size=(n, 3)); Z = np.asfortranarray((Z - Z.mean(axis=0)) / Z.std(axis=0, ddof=1))
        locus_z_core_(Y, X, Z, hy, ini["gam_vb"], ini["alpha_vb"], ini["mu_beta_vb"],
                      ini["sig2_alpha_vb"], ini["sig2_beta_vb"], ini["tau_vb"], **kw)
    elif which == "c5":
        from locus_amd.locus_struct_core import locus_struct_core_
        ls = set_struct(n, p, np.arange(1, p + 1, 100))
        locus_struct_core_(Y, X, hy, ini["gam_vb"], ini["mu_beta_vb"], ini["sig2_beta_vb"],
                           ini["tau_vb"], ls, **kw)
    else:
        from locus_amd.locus_bin_core import locus_logit_core_
        Z = add_intercept_(n, d, hy, ini, link)
        locus_logit_core_(Y, X, Z, hy, ini["gam_vb"], ini["alpha_vb"], ini["mu_beta_vb"],
                          ini["sig2_alpha_vb"], ini["sig2_beta_vb"], **kw)
    total = time.perf_counter() - t0
    per = np.diff(marks) * 1e3
    lbs = np.array([v[1] for v in trace])
    print(f"  set-up + first iteration {1e3 * (marks[0] - t0):.0f} ms; later iterations "
          f"{np.round(per, 1).tolist()} ms; total {total:.2f} s")
    if to_tol:
        print(f"  to |dELBO| < 0.1: {len(lbs)} iterations, {marks[-1] - t0:.2f} s from the call "
              f"(driver set-up included)")
    print(f"  ELBO {lbs[0]:.6e} -> {lbs[-1]:.6e}, monotone: {bool(np.all(np.diff(lbs) > -1e-6 * abs(lbs[-1])))}")


main()
