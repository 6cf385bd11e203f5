"""Turns gpurun_out/prof (scripts/profile_bench.sh) into the committed profiles/ files.
usage: python scripts/summarise_profile.py r01 n p d_loc"""
import csv, glob, json, os, shutil, sys

tag, n, p, d_loc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
src = "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)
stats = max(glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
shutil.copy(stats, f"profiles/{tag}_bench_kernel_stats.csv")

def counter(dirname, name):
    f = max(glob.glob(f"{src}/{dirname}/**/*_counter_collection.csv", recursive=True),
            key=os.path.getmtime)
    rows = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows

fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1",
         "# values in KiB per dispatch as reported, median over the dispatches (the first ones read cold); hbm_read applies the gfx950 correction of",
         "# MI355X_MICROARCH.md §HBM (FETCH_SIZE reports 1/2 of wide coalesced reads) = 2 x FETCH_SIZE",
         "kernel,calls,FETCH_SIZE_KiB_median,WRITE_SIZE_KiB_median,hbm_read_bytes,hbm_write_bytes"]
traffic = []
# one sweep = the kernels locus_hip_sweep_dev launches; build_steps_kernel runs once per sweep
nsweeps = max([len(v) for k, v in fetch.items() if "build_steps_kernel" in k] + [1])
if any("refresh_lod_kernel" in k for k in fetch):
    # operands packed ahead (round 3): build_steps runs in the pack-ahead call AND in the sweep;
    # k sweeps = 1 (first, plain) + 2 (k - 1) + 1 (the last pack-ahead, never swept) = 2 k dispatches
    nsweeps = max(1, nsweeps // 2)
SWEEP = ("sweep_kernel", "sweep1_kernel", "pack_blocks", "build_steps_kernel", "logit_la_kernel",
         "refresh_lod_kernel")
seq = [k for k in fetch if any(s in k for s in SWEEP)]
if seq:
    # median dispatch of every kernel x its dispatches per sweep: the first dispatches of a
    # process read cold (up to 6x the steady-state misses) and would dominate a mean over 4 sweeps
    med = lambda v: sorted(v)[(len(v) - 1) // 2]  # lower median
    rd = sum(med(fetch[k]) * len(fetch[k]) for k in seq) * 2 * 1024 / nsweeps
    wr = sum(med(write.get(k, [0])) * len(write.get(k, [0])) for k in seq) * 1024 / nsweeps
    traffic.append(dict(config="bench", n=n, p=p, d_loc=d_loc, kernel="sweep sequence: " + "; ".join(
                            f"{k[:60]} x{len(fetch[k]) / nsweeps:g}" for k in seq),
                        hbm_bytes_per_launch=rd + wr, hbm_read_bytes=rd, hbm_write_bytes=wr,
                        correction="2 x FETCH_SIZE (gfx950, 16-B coalesced loads) + WRITE_SIZE, "
                                   "summed over the kernels of one sweep"))
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):      # one row per kernel, largest reader first
    fs = sorted(fetch[k])[(len(fetch[k]) - 1) // 2]                # median dispatch (see above)
    ws = sorted(write.get(k, [0]))[(len(write.get(k, [0])) - 1) // 2]
    lines.append(f"\"{k[:80]}\",{len(fetch[k])},{fs:.1f},{ws:.1f},{2*fs*1024:.4e},{ws*1024:.4e}")
    if any(s in k for s in SWEEP[:2] + SWEEP[4:]):
        traffic.append(dict(config="bench", n=n, p=p, d_loc=d_loc, kernel=k, per="dispatch of this kernel",
                            hbm_bytes_per_dispatch=2 * fs * 1024 + ws * 1024,
                            fetch_size_kib=fs, write_size_kib=ws,
                            correction="2 x FETCH_SIZE (gfx950, 16-B coalesced loads) + WRITE_SIZE"))
open(f"profiles/{tag}_bench_hbm_counters.csv", "w").write("\n".join(lines) + "\n")
tpath = f"profiles/traffic_{tag}.json"   # the per-config records of summarise_configs.py stay
keep = [t for t in (json.load(open(tpath)) if os.path.exists(tpath) else []) if t.get("config") != "bench"]
json.dump(traffic + keep, open(tpath, "w"), indent=1)
print("\n".join(lines[:12]))
print(traffic)

# matrix-core utilisation (own --pmc pass): SQ_VALU_MFMA_BUSY_CYCLES counts SIMD-cycles with an MFMA
# in flight, SQ_BUSY_CU_CYCLES counts CU-cycles with waves resident => util = MFMA / (4 SIMDs x CU)
mf = glob.glob(f"{src}/mfma/**/*_counter_collection.csv", recursive=True)
if mf:
    import collections
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(mf, key=os.path.getmtime))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = ["# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE, bench.py --steps 3 --warmup 1",
           "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): share of the SIMD-cycles of the",
           "# busy CUs in which a matrix-core instruction executes (fp64 MFMA peak 78.6 TFLOP/s = 100 %)",
           "kernel,calls,SQ_VALU_MFMA_BUSY_CYCLES_avg,SQ_BUSY_CU_CYCLES_avg,GRBM_GUI_ACTIVE_avg,mfma_util"]
    for k, v in acc.items():
        m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])
        if sum(m) == 0:
            continue
        avg = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else 0.0
        out.append(f"\"{k[:80]}\",{len(m)},{avg('SQ_VALU_MFMA_BUSY_CYCLES'):.4e},"
                   f"{avg('SQ_BUSY_CU_CYCLES'):.4e},{avg('GRBM_GUI_ACTIVE'):.4e},"
                   f"{avg('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * avg('SQ_BUSY_CU_CYCLES')):.3f}")
    open(f"profiles/{tag}_bench_mfma_util.csv", "w").write("\n".join(out) + "\n")
    print("\n".join(out[3:]))
