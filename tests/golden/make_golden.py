"""Generates the committed fixtures of tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path and cannot run here (no R / Rcpp /
Eigen), so these vectors pin the ORACLE's behaviour (regression guard) -- they are not
outputs of the reference.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import INOUT, call_variant, inputs_for  # noqa: E402
from oracle import c_oracle, ref_driver  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
c_oracle.build()
for variant in ["identity", "z", "struct", "probit", "logit", "batch", "zbatch", "probitbatch"]:
    inp = inputs_for(variant, 40, 12, 3, seed=17, c=0.8 if variant in ("identity", "z") else 1.0)
    out = call_variant(c_oracle, variant, inp)
    blob = {"in_" + k: np.asarray(v) for k, v in inp.items()}
    blob.update({"out_" + k: out[k] for k in INOUT})
    np.savez_compressed(os.path.join(HERE, f"sweep_{variant}.npz"), **blob)

from locus_amd.prepare import auto_set_hyper_, auto_set_init_  # noqa: E402
from locus_amd.simulate import simulate  # noqa: E402
sim = simulate(200, 500, 20, 15, seed=123)
hy = auto_set_hyper_(sim["Y"], 500, 15)
ini = auto_set_init_(sim["Y"], 500, 15, user_seed=123)
rng = np.random.default_rng(1)
tr = []
r = ref_driver.locus_core_(sim["Y"], sim["X"], hy, ini["gam_vb"], ini["mu_beta_vb"],
                           ini["sig2_beta_vb"], ini["tau_vb"], tol=0.1, maxit=100,
                           perm_fn=lambda it: rng.permutation(500).astype(np.float64), trace=tr)
np.savez_compressed(os.path.join(HERE, "driver_identity_c1.npz"),
                    elbo_trace=np.array([x[1] for x in tr]), it=r["it"],
                    selected=np.argwhere(r["gam_vb"] > 0.5))

# the restated drivers (oracle/ref_driver.py) on the small problems of tests/test_drivers_cpu.py:
# every link / variant, 3 iterations
import test_drivers_cpu as T  # noqa: E402
blob = {}
for kind in T.KINDS:
    res = T._run(kind, "ref")
    for k in T.GOLDEN_KEYS:
        blob[f"{kind.replace('-', '_')}__{k}"] = np.asarray(res[k])
np.savez_compressed(os.path.join(HERE, "drivers_small.npz"), **blob)
print("golden fixtures written")
