"""Per-output error of a forced sweep geometry against the C oracle on a small case.
usage: python scripts/debug_t2.py [cfg] [n p d]"""
import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
cfg = sys.argv[1] if len(sys.argv) > 1 else "-1,2,8,1"
os.environ["LOCUS_HIP_SWEEP_CFG"] = cfg
n, p, d = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (107, 53, 71)
from helpers import call_variant, inputs_for
from oracle import c_oracle
import locus_amd
c_oracle.build()
inp = inputs_for("identity", n, p, d, seed=21)
ref = call_variant(c_oracle, "identity", inp)
got = call_variant(locus_amd.core, "identity", inp)
for k in ref:
    r, g = np.asarray(ref[k]), np.asarray(got[k])
    err = np.abs(r - g)
    print(k, r.shape, "max abs err %.3e" % err.max(), "at", np.unravel_index(err.argmax(), err.shape))
    if err.max() > 1e-9 and r.ndim == 2:
        bad = err > 1e-9
        print("   bad rows:", np.unique(np.nonzero(bad)[0])[:20], " bad cols:", np.unique(np.nonzero(bad)[1])[:40])
