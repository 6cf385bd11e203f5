"""Where the wall-clock of an end-to-end locus() call goes (host preparation, defaults, upload, CAVI
iterations, download).  usage: python scripts/perf_locus_e2e.py [n p d]"""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
import locus_amd


def main():
    n, p, d = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 50000, 2500)
    rng = np.random.default_rng(123)
    t0 = time.perf_counter()
    X = rng.binomial(2, 0.25, size=(n, p)).astype(np.float64)
    act = rng.choice(p, 100, replace=False)
    beta = np.where(rng.random((100, d)) < 0.2, rng.normal(size=(100, d)), 0.0)
    Y = X[:, act] @ beta + rng.normal(scale=0.5, size=(n, d))
    print(f"simulate n={n} p={p} d={d}: {time.perf_counter() - t0:.1f} s", flush=True)
    import torch
    torch.zeros(1, device="cuda")
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    vb = locus_amd.locus(Y, X, p0_av=100, user_seed=1, verbose=False)
    pr.disable()
    print(f"locus(): {time.perf_counter() - t0:.2f} s, {vb['it']} iterations, converged {vb['converged']}")
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


main()
