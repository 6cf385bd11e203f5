"""Turns gpurun_out/prof_<cfg> (scripts/profile_configs.sh) into profiles/<tag>_<cfg>_driver_kernel_stats.csv
and appends the per-config HBM traffic of one sweep to profiles/traffic_<tag>.json.
usage: python scripts/summarise_configs.py r02 c2 c4 c5"""
import csv, glob, json, os, shutil, sys

tag, cfgs = sys.argv[1], sys.argv[2:]
SHAPES = dict(c2=(1000, 20000, 1000), c4=(2000, 20000, 500), c5=(1000, 50000, 2000))
SWEEP = ("sweep_kernel", "sweep1_kernel", "pack_blocks", "build_steps_kernel", "logit_la_kernel",
         "refresh_lod_kernel")
path = f"profiles/traffic_{tag}.json"
traffic = json.load(open(path)) if os.path.exists(path) else []
traffic = [t for t in traffic if t.get("config") not in cfgs]


def counter(src, dirname, name):
    f = max(glob.glob(f"{src}/{dirname}/**/*_counter_collection.csv", recursive=True),
            key=os.path.getmtime)
    rows = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows


for cfg in cfgs:
    src = f"gpurun_out/prof_{cfg}"
    stats = max(glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
    shutil.copy(stats, f"profiles/{tag}_{cfg}_driver_kernel_stats.csv")
    fetch, write = counter(src, "fetch", "FETCH_SIZE"), counter(src, "write", "WRITE_SIZE")
    n, p, d = SHAPES[cfg]
    nsweeps = max([len(v) for k, v in fetch.items() if "build_steps_kernel" in k] + [1])
    if any("refresh_lod_kernel" in k for k in fetch):
        # operands packed ahead (round 3): build_steps runs in the pack-ahead call AND in the sweep;
        # k sweeps = 1 (first, plain) + 2 (k - 1) + 1 (the last pack-ahead, never swept) = 2 k dispatches
        nsweeps = max(1, nsweeps // 2)
    seq = [k for k in fetch if any(s in k for s in SWEEP)]
    # median dispatch of every kernel x its dispatches per sweep: the first dispatches of a
    # process read cold (up to 6x the steady-state misses) and would dominate a mean over 4 sweeps
    med = lambda v: sorted(v)[(len(v) - 1) // 2]  # lower median
    rd = sum(med(fetch[k]) * len(fetch[k]) for k in seq) * 2 * 1024 / nsweeps
    wr = sum(med(write.get(k, [0])) * len(write.get(k, [0])) for k in seq) * 1024 / nsweeps
    traffic.append(dict(config=cfg, n=n, p=p, d_loc=d, sweeps=nsweeps,
                        kernel="sweep sequence: " + "; ".join(
                            f"{k[:60]} x{len(fetch[k]) / nsweeps:g}" for k in seq),
                        hbm_bytes_per_sweep=rd + wr, hbm_read_bytes=rd, hbm_write_bytes=wr,
                        algorithmic_bytes_per_sweep=p * d * (16.0 * n + 8.0 * n / d + 32),
                        correction="2 x FETCH_SIZE (gfx950, 16-B coalesced loads) + WRITE_SIZE, "
                                   "summed over the kernels of one sweep"))
    print(cfg, f"sweeps={nsweeps} read={rd:.3e} write={wr:.3e}")
json.dump(traffic, open(path, "w"), indent=1)
