"""Wall-clock breakdown of one iteration of the structured driver at C5 (one GPU).
usage: python scripts/struct_breakdown.py [iterations]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")


class pd:  # same synthetic data as scripts/perf_drivers.py
    @staticmethod
    def data(n, p, d, p0, link, seed=123):
        rng = np.random.default_rng(seed)
        X = rng.binomial(2, 0.25, size=(n, p)).astype(np.float64)
        X = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
        act = rng.choice(p, p0, replace=False)
        beta = np.where(rng.random((p0, d)) < 0.2, rng.normal(size=(p0, d)), 0.0)
        Y = X[:, act] @ beta + rng.normal(scale=0.5, size=(n, d))
        Y -= Y.mean(axis=0)
        return np.asfortranarray(X), np.asfortranarray(Y)

from locus_amd.prepare import auto_set_hyper_, auto_set_init_
from locus_amd.struct import set_struct
from locus_amd import locus_struct_core as lsc, locus_core as lc, device as dv

n, p, d, p0 = 1000, 50000, 2000, 100
X, Y = pd.data(n, p, d, p0, "identity")
hy = auto_set_hyper_(Y, p, p0, struct=True)
ini = auto_set_init_(Y, p, p0, user_seed=1)
T = {}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
for name in ("begin_iteration", "end_iteration"):
    timed(lsc.StructHost, name)
timed(dv.Sweeper, "identity"); timed(dv, "gamma_rows"); timed(lc.IdentityEngine, "_up")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ls = set_struct(n, p, np.arange(1, p + 1, 100))
t0 = time.perf_counter()
lsc.locus_struct_core_(Y, X, hy, ini["gam_vb"], ini["mu_beta_vb"], ini["sig2_beta_vb"], ini["tau_vb"], ls,
                       tol=1e-30, maxit=iters)
print(f"total {time.perf_counter() - t0:.2f} s; per iteration (ms): " +
      " ".join(f"{k}={v / iters * 1e3:.2f}" for k, v in T.items()))
