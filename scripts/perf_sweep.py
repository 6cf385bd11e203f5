"""Times the identity (or logit) sweep kernel for explicit geometries on one GPU.
usage: python scripts/perf_sweep.py n p d [logit] cfg1 cfg2 ...   (cfg = NPL,C,S,G or auto)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from locus_amd import device as dv, _lib

def main():
    n, p, d = map(int, sys.argv[1:4])
    rest = sys.argv[4:]
    logit = False
    if rest and rest[0] == "logit":
        logit = True; rest = rest[1:]
    nsteps = p
    cfgs = rest or ["auto"]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.bernoulli(torch.full((p, n), 0.25, device=dev, dtype=torch.float64), generator=g) + \
        torch.bernoulli(torch.full((p, n), 0.25, device=dev, dtype=torch.float64), generator=g)
    X = (X - X.mean(dim=1, keepdim=True)) / X.std(dim=1, keepdim=True).clamp_min(1e-3)
    Xc = X.t()  # n x p column-major storage
    px = dv.PackedX(Xc)
    del X, Xc
    state = torch.empty((3, p, d), device=dev, dtype=torch.float64)
    state[0].uniform_(0, 1, generator=g); state[0].pow_(8.0)
    state[1].normal_(0, 0.3, generator=g)
    state[2] = state[0] * state[1]
    R0 = torch.randn((d, n), device=dev, dtype=torch.float64, generator=g) * 0.7
    lo = torch.full((p,), -3.0, device=dev, dtype=torch.float64)
    l1 = torch.full((p,), -0.05, device=dev, dtype=torch.float64)
    tau = torch.full((d,), 2.0, device=dev, dtype=torch.float64)
    s2k = 1.0 / ((n - 1 + 0.7) * tau)
    scale = s2k * tau; hinv = 1.0 / (2 * s2k); cst = -(tau.log() + np.log(0.7) + s2k.log()) / 2
    stats = torch.empty((4, d), device=dev, dtype=torch.float64)
    if logit:
        W2 = (torch.rand((d, n), device=dev, dtype=torch.float64, generator=g) * 0.2 + 0.2).t()
        import os as _os
        s2 = torch.full((p, d), 1.0 / (n * 0.3), device=dev, dtype=torch.float64)
        xwx = torch.full((p, d), n * 0.3, device=dev, dtype=torch.float64)
        ls2 = None if _os.environ.get('PERF_NO_LS2') else 0.5 * torch.log(s2)  # the driver tabulates it
    for cs in cfgs:
        cfg = None if cs == "auto" else tuple(map(int, cs.split(",")))
        sw = dv.Sweeper(px, d, dev, cfg=cfg)
        times = []
        for it in range(4):
            perm = torch.randperm(p, device=dev, generator=g).to(torch.float64)
            R = R0.clone().t()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if logit:
                sw.logit(perm, lo, l1, s2, xwx, W2, 0.1, R, state, stats, nsteps, ls2=ls2)
            else:
                sw.identity(perm, lo, l1, scale, hinv, cst, s2k, 1.0, R, state, stats, nsteps)
            e1.record(); torch.cuda.synchronize()
            sw.status()
            times.append(e0.elapsed_time(e1))
        t = min(times[1:])
        cfg_used = (_lib.ctypes.c_int32 * 4)()
        _lib.lib().locus_hip_sweep_config(n, d, int(logit), cfg_used)
        upd = nsteps * d / (t * 1e-3)
        if "diag" in _lib.LIB_PATH and cfg and cfg[0] == -3:
            import ctypes
            fn = ctypes.CDLL(_lib.LIB_PATH).locus_hip_diag_read_la_stamps
            nwg = min(512, (d + max(cfg[1], 1) - 1) // max(cfg[1], 1))
            buf = (ctypes.c_ulonglong * (24 * nwg))()
            fn(buf, 24 * nwg)
            a = np.array(buf, dtype=np.float64).reshape(nwg, 3, 8) / nsteps
            for ro, nm, names in ((0, "vector wave 0", ["busy", "wait", "->update", "->reduce", "busy2", "wait2",
                                                       "busy3", "wait3"]),
                                  (2, "wave 6 (duties)", ["busy", "wait", "->stored", "->prepared"]),
                                  (1, "chain", ["busy", "wait", "->U", "->delta"])):
                med = np.median(a[:, ro, :len(names)], axis=0)
                print(f"   ticks/step {nm:14s}: " + " ".join(f"{k}={v:.0f}" for k, v in zip(names, med)))
        elif "diag" in _lib.LIB_PATH and cfg and cfg[0] < 0 and (cfg[1] == 1 or (cfg[1] == 2 and cfg[2] in (2, 3))):
            import ctypes
            import os
            t2 = cfg[1] == 2
            cs_used = cfg[2] if cfg[2] in (2, 3) else int(os.environ.get("LOCUS_HIP_MFMA_CS", "1"))
            fn = getattr(ctypes.CDLL(_lib.LIB_PATH),
                         f"locus_hip_diag_read_mfma1_stamps_c{cs_used}" + ("t2" if t2 else ""))
            nwg = min(4096, cs_used * ((d + 15) // (32 if t2 else 16) + 8))
            buf = (ctypes.c_ulonglong * (32 * nwg))()
            fn(buf, 32 * nwg)
            a = np.array(buf, dtype=np.float64).reshape(nwg, 4, 8)
            member = (np.arange(nwg) >> 3) % cs_used      # c_mem of the kernel (members 8 workgroups apart)
            keep = a[:, 2, 1] > 0
            member = member[keep]
            a = a[keep] / ((nsteps + 15) // 16)
            if cs_used > 1 and os.environ.get("PERF_BY_MEMBER"):
                for cm in range(cs_used):
                    am = a[member == cm]
                    print(f"   member {cm} ({len(am)} workgroups), medians:")
                    for ro, nm in ((0, "matrix A"), (1, "matrix B"), (2, "chain"), (3, "helper")):
                        print(f"      {nm:9s} " + " ".join(f"{v:6.0f}" for v in np.median(am[:, ro, :7], axis=0))
                              + f"  total={np.median(am[:, ro, :7].sum(axis=1)):.0f}")
            roles = {0: ("matrix wave 0", ["B1-wait", "mfma", "gfin-wait", "corr(l>=2)", "follow+write"]),
                     1: ("matrix wave 5", ["B1-wait", "mfma"]),
                     2: ("chain", ["B1-wait", "run", "-", "next-G"]),
                     3: ("helper", ["B1-wait", "sums", "stores", "granule-wait", "requests", "load-wait+handover", "-", "refetches"])}
            for ro, (nm, names) in roles.items():
                med = np.median(a[:, ro, :len(names)], axis=0)
                print(f"   ticks/block {nm:14s}: " + " ".join(f"{k}={v:.0f}" for k, v in zip(names, med))
                      + f" total={med.sum():.0f}")
                if ro == 3:
                    for pc in (10, 90):
                        qv = np.percentile(a[:, ro, :len(names)], pc, axis=0)
                        print(f"      helper p{pc}: " + " ".join(f"{k}={v:.0f}" for k, v in zip(names, qv)))
        elif "diag" in _lib.LIB_PATH and cfg and cfg[0] < 0:
            import ctypes
            fn = getattr(ctypes.CDLL(_lib.LIB_PATH), f"locus_hip_diag_read_mfma_stamps_{cfg[1]}")
            nwg = min(4096, (d + 16 * cfg[1] - 1) // (16 * cfg[1]))
            MWV = 6 if cfg[1] == 1 else 7
            buf = (ctypes.c_ulonglong * (8 * nwg))()
            fn(buf, 8 * nwg)
            a = np.array(buf, dtype=np.float64).reshape(nwg, 8) / ((nsteps + 15) // 16 + 2)
            names = ["w0:B1-wait", "w0:bop+xcorr", "w0:upd0+B2", "w0:update", "w0:dot", "-",
                     "chain:run", "chain:wait"]
            # "end": partial-G and staging stores after the stream; on wave 3 also the cross-block
            # correction that follows the chain wave
            names = ["B1-wait", "bop", "end", "update", "dot"]
            for wv in range(MWV):
                aw = a[wv::MWV]
                if len(aw):
                    print(f"   ticks/iteration wave {wv}: " +
                          " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, np.median(aw[:, :5], axis=0))) +
                          f" total={np.median(aw[:, :5].sum(axis=1)):.0f}")
            print(f"   chain wave: run={np.median(a[:, 6]):.0f} wait={np.median(a[:, 7]):.0f}")
        elif "diag" in _lib.LIB_PATH:
            import ctypes
            c = cfg or tuple(cfg_used)
            f = getattr(_lib.lib(), f"locus_hip_diag_read_stamps_{c[0]}_{int(logit)}")
            nwg = min(4096, (d + c[1] * c[3] - 1) // (c[1] * c[3]))
            buf = (ctypes.c_ulonglong * (8 * nwg))()
            f(buf, 8 * nwg)
            a = np.array(buf, dtype=np.float64).reshape(nwg, 8)[:, :7] / nsteps
            names = ["dot", "reduce", "barrier", "elem", "store/pref", "rank1", "xstage"]
            print("   cycles/step (median over WGs, s_memtime ticks): " +
                  " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, np.median(a, axis=0))) +
                  f" total={np.median(a.sum(axis=1)):.0f}")
        print(f"cfg={cs:12s} auto={list(cfg_used)} ms={t:9.3f} (all {[round(x,2) for x in times]}) "
              f"updates/s={upd:.3e} fp64_TF={(upd*(4*n))/1e12:.2f} algGB/s={upd*(16*n+8*n/d+32)/1e9:.0f}",
              flush=True)

main()
