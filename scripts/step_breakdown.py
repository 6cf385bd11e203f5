"""Wall-clock breakdown of one CAVI iteration of the identity driver (bench workload by default).
usage: python scripts/step_breakdown.py [n p d]
env: STEP_HOST=host uses the numpy host side (IdentityHost) instead of the device-resident tail;
     STEP_SYNC=0 times whole iterations only (no synchronisation around the parts)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from locus_amd import device as dv
from locus_amd.comm import Comm
from locus_amd.locus_core import FusedTail, IdentityEngine, IdentityHost, IdentityTail
HostCls = {'host': IdentityHost, 'tail': IdentityTail}.get(os.environ.get('STEP_HOST'), FusedTail)
SYNC = os.environ.get('STEP_SYNC', '1') != '0'

n, p, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 50000, 10000)
dev = torch.device("cuda:0")
Xpn, Yt, med = bench.make_data(n, p, d, 100, 0, d)
px = dv.PackedX(torch.from_numpy(Xpn).to(dev).t())
Y = torch.from_numpy(Yt).to(dev).t()
hyper = dict(a=np.ones(p), b=np.full(p, d * (p - 100) / 100), eta=np.full(d, 1 / med),
             kappa=np.ones(d), nu=1.0)
hyper["lambda"] = 1e-2
tau0 = np.full(d, 1 / med)
state = torch.empty((3, p, d), dtype=torch.float64, device=dev)
state[0].uniform_(0, 1e-3); state[1].normal_(0, 1)
host = HostCls(n, p, d, hyper, tau0, 1.0 / np.random.default_rng(1).gamma(2.0, 1e-2 * tau0), Comm(),
               **(dict(dv=dv) if HostCls is FusedTail else {}))
eng = IdentityEngine(px, Y, state, host, dev)
eng.initialise()
T = {}
def timed(obj, name, sync=True):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        if sync: torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
if SYNC:
    timed(host, "begin_iteration"); timed(host, "end_iteration"); timed(eng.sweeper, "identity")
    timed(dv, "gamma_rows"); timed(eng, "_up")
rng = np.random.default_rng(1)
K = 6
for it in range(K + 1):
    if it == 1:
        T.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    if it == 0:
        nxt = [rng.permutation(p).astype(np.float64)]
    def draw():  # overlaps the sweep, as in locus_core_
        tp = time.perf_counter(); nxt.append(rng.permutation(p).astype(np.float64)); T["perm"] = T.get("perm", 0) + time.perf_counter() - tp
    kw = dict(c_next=1.0) if HostCls is not IdentityHost else {}
    eng.iterate(nxt.pop(0), 1.0, with_elbo=True, after_launch=draw, **kw)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / K
print(f"{HostCls.__name__} n={n} p={p} d={d}: total {tot*1e3:.3f} ms/iteration; " + " ".join(f"{k}={v/K*1e3:.2f}" for k, v in T.items()))
