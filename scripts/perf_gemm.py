"""Times the fp64 MFMA GEMM (locus_hip_gemm_dev) on the contraction shapes of the path and checks it
against torch.matmul.  usage: python scripts/perf_gemm.py"""
import sys
import torch
sys.path.insert(0, ".")
from locus_amd import device as dv

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
PEAK = 78.6


def run(name, m, n, k, a_kfast, b_kfast, sq=False):
    # A is m x k: k-fast => row-major [m][k] (sa_m = k, sa_k = 1) else column-major (sa_m = 1, sa_k = m)
    A = torch.randn((m, k) if a_kfast else (k, m), device=dev, dtype=torch.float64, generator=g)
    B = torch.randn((n, k) if b_kfast else (k, n), device=dev, dtype=torch.float64, generator=g)
    C = torch.empty((n, m), device=dev, dtype=torch.float64)  # column-major m x n
    sa = (k, 1) if a_kfast else (1, m)
    sb = (1, k) if b_kfast else (n, 1)
    def call():
        dv.gemm(m, n, k, A, sa[0], sa[1], B, sb[0], sb[1], C, 1, m, sqA=sq)
    call(); torch.cuda.synchronize()
    Am = A if a_kfast else A.t()
    Bm = B.t() if b_kfast else B
    ref = ((Am * Am) if sq else Am) @ Bm
    err = ((C.t() - ref).abs().max() / ref.abs().max()).item()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts)
    tf = 2.0 * m * n * k / (t * 1e-3) / 1e12
    print(f"{name:34s} m={m:6d} n={n:6d} k={k:6d} {t:8.3f} ms  {tf:6.2f} TFLOP/s  {100*tf/PEAK:5.1f}% of MFMA-f64 peak  rel.err {err:.1e}", flush=True)


run("mat_x_m1 = X beta (C3)", 1000, 10000, 50000, False, False)
run("coreBatch X'R (C2)", 20000, 1000, 1000, True, True)
run("coreBatch X beta (C2)", 1000, 1000, 20000, False, False)
run("logit xwx = (X^2)'(2psi) (C4)", 20000, 500, 2000, True, True, sq=True)
run("logit Q = X^2 D (C4)", 2000, 500, 20000, False, False, sq=True)
