// Host side of the sweep: geometry choice and launch dispatch.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace locus {

#define LOCUS_DECL(npl, l)                                                                    \
    int launch_sweep_npl##npl##_l##l(const SweepParams&, int, dim3, dim3, size_t, hipStream_t);
LOCUS_DECL(2, 0)
LOCUS_DECL(2, 1)
LOCUS_DECL(4, 0)
LOCUS_DECL(4, 1)
LOCUS_DECL(8, 0)
LOCUS_DECL(8, 1)
LOCUS_DECL(16, 0)
LOCUS_DECL(16, 1)
#undef LOCUS_DECL
// "wide" NPL = 16 instantiations: 4 staging pieces per thread, 2048 < n <= 4096
int launch_sweep_wide_npl16_l0(const SweepParams&, int, dim3, dim3, size_t, hipStream_t);
int launch_sweep_wide_npl16_l1(const SweepParams&, int, dim3, dim3, size_t, hipStream_t);

static const int kCs[] = {1, 2, 4, 5, 8, 10, 16};

// Largest C instantiated for (NPL, logit): 2*NPL*C*(logit?2:1) residual VGPRs must
// leave room for the rest of the step inside 256 (sweep_inst.inc has the same table).
static int max_c(int npl, bool logit)
{
    const int eff = npl * (logit ? 2 : 1);
    if (eff <= 4) return 16;
    if (eff <= 8) return 10;
    if (eff <= 16) return 5;
    return 2;
}

static int vgpr_estimate(int npl, int c, bool logit, bool wide = false)
{
    // keep in step with sweep_depth() / sweep_max_threads() of sweep_kernel.h
    const int regs = 2 * npl * c * (logit ? 2 : 1);
    if (wide) return regs + 16 * ((regs <= 64 && !logit) ? 4 : 2) + 90;
    const int depth = (logit && regs > 48) ? (regs <= 112 ? 4 : 2)
                                           : (regs <= 64 ? 6 : (regs <= 112 ? 4 : 2));
    return regs + 8 * depth + 90;
}

size_t sweep_lds_bytes(const SweepConfig& cfg)
{
    const size_t npad = (size_t)64 * cfg.NPL * cfg.S;
    const size_t CP = 16;
    return sizeof(double) * (3 * npad + 2 * (size_t)cfg.G * cfg.S * CP + 2 * (size_t)cfg.G * CP * 4);
}

// Rough cycle model of one step for one wave (issue cycles) -- only used to rank
// geometries; measured numbers are in DESIGN.md.
static double step_issue_cycles(int npl, int c, int s, bool logit)
{
    const int ng = (c + 3) / 4;
    double fma = (logit ? 3.0 : 2.0) * npl * c * 4.0;  // fp64 FMA: 4 cycles per wave
    double reduce = ng * 60.0;
    double elem = logit ? 420.0 : 300.0;
    double bcast = 8.0 * c;
    double xchg = (s > 1) ? 60.0 + 8.0 * s : 10.0;
    double lds = npl * 4.0;  // two passes of ds_read_b128 over the slice
    return fma + reduce + elem + bcast + xchg + lds + 60.0;
}

static double step_latency_cycles(int npl, int c, int s, bool logit)
{
    (void)c;
    double chain = 120.0 /*reduce*/ + (logit ? 650.0 : 500.0) /*element-wise*/ + 40.0 + 8.0 * npl;
    if (s > 1) chain += 200.0;
    return chain;
}

int choose_sweep_config(int n, int d, bool logit, SweepConfig* out)
{
    if (n < 1 || d < 1) {
        set_error("sweep: n and d must be positive");
        return LOCUS_HIP_EINVAL;
    }
    if (const char* env = getenv("LOCUS_HIP_SWEEP_CFG")) {  // "NPL,C,S,G" (tuning / tests)
        SweepConfig c{};
        if (sscanf(env, "%d,%d,%d,%d", &c.NPL, &c.C, &c.S, &c.G) == 4 &&
            (c.NPL == -2 || (c.NPL == -3 ? logit : (c.NPL < 0 ? !logit : (c.NPL * c.S * 64 >= n))))) {
            *out = c;
            return 0;
        }
    }
    // identity family: the matrix-core (blocked, look-ahead) sweep wherever it covers n; measured
    // 1.6-2.6x faster than the vector kernel at n = 1000 for every d.
    // LOCUS_HIP_SWEEP_VECTOR=1 keeps the vector kernel (comparisons, tests).
    if (!logit && mfma_sweep_covers(n)) {
        const char* v = getenv("LOCUS_HIP_SWEEP_VECTOR");
        if (!(v && v[0] == '1')) {
            *out = SweepConfig{-1, 0, 0, 0};
            return 0;
        }
    }
    const int units = (n + 63) / 64;  // 64-row units a column needs
    const int num_cu = device_cus();
    // logit family, latency regime (at most one workgroup of 4 responses per CU): the look-ahead
    // kernel with a dedicated chain wave.  LOCUS_HIP_SWEEP_VECTOR=1 keeps the vector kernel.
    const int la_cr = logit ? logit_la_max_cr(n) : 0;
    if (la_cr > 0 && (d + la_cr - 1) / la_cr <= num_cu) {
        const char* v = getenv("LOCUS_HIP_SWEEP_VECTOR");
        if (!(v && v[0] == '1')) {
            *out = SweepConfig{-3, 0, 0, 0};  // C and G from (n, d) at launch
            return 0;
        }
    }
    double best = 1e300;
    SweepConfig bc{0, 0, 0, 0};
    for (int npl : {2, 4, 8, 16}) {
        for (int s : {1, 2, 4, 8, 16}) {
            if (npl * s < units) continue;
            if (npl * s >= 2 * units && npl * s > 2) continue;  // more than 2x row padding
            for (int c : kCs) {
                if (c > max_c(npl, logit)) continue;
                // n > 2048: only the wide NPL = 16 instantiations (4 staging pieces per thread)
                const bool wide = units > 32;
                if (wide && (npl != 16 || c > (logit ? 1 : 4))) continue;
                const int vg = vgpr_estimate(npl, c, logit, wide);
                const int max_thr = vg <= 128 ? 1024 : 512;
                const int gmin = wide ? std::max(1, 8 / s)                     // 512 threads
                                      : std::max(1, (npl + 3) / 4);  // <= 2 staging pieces per thread
                for (int g : {1, 2, 4, 8, 16}) {
                    if (g < gmin) continue;
                    const int thr = 64 * s * g;
                    if (thr > max_thr) continue;
                    SweepConfig cfg{npl, c, s, g};
                    const size_t lds = sweep_lds_bytes(cfg);
                    if (lds > 160 * 1024) continue;
                    // workgroups per CU: VGPRs (512 per lane per SIMD), LDS, 32 waves
                    const int waves_wg = s * g;
                    const int wps_reg = vg <= 64 ? 8 : vg <= 128 ? 4 : vg <= 168 ? 3 : vg <= 256 ? 2 : 1;
                    const int wg_reg = std::max(1, (wps_reg * 4) / waves_wg);
                    const int wg_lds = (int)((160 * 1024) / lds);
                    const int wg_cu = std::max(1, std::min({wg_reg, wg_lds, 32 / waves_wg, 8}));
                    const long n_wg = ((long)d + (long)g * c - 1) / ((long)g * c);
                    const long slots = (long)num_cu * wg_cu;
                    const long rounds = (n_wg + slots - 1) / slots;
                    // resident waves per SIMD in the fullest round
                    const long wg_here = std::min<long>(n_wg, slots);
                    const double wg_per_cu = std::ceil((double)wg_here / num_cu);
                    const double waves_simd = std::max(1.0, wg_per_cu * waves_wg / 4.0);
                    const double issue = waves_simd * step_issue_cycles(npl, c, s, logit);
                    const double lat = step_latency_cycles(npl, c, s, logit);
                    const double t = rounds * std::max(issue, lat);
                    if (t < best) {
                        best = t;
                        bc = cfg;
                    }
                }
            }
        }
    }
    if (bc.NPL == 0) bc = SweepConfig{-2, 0, 0, 0};  // n > 4096: the streaming kernel
    *out = bc;
    return 0;
}

int launch_sweep(const SweepParams& prm_in, const SweepConfig& cfg, bool logit, hipStream_t stream)
{
    SweepParams prm = prm_in;
    prm.S = cfg.S;
    prm.G = cfg.G;
    prm.npad = 64 * cfg.NPL * cfg.S;
    if (prm.ld < prm.n || (prm.ld & 127) != 0 || prm.ld > prm.npad) {
        set_error("sweep: inconsistent geometry (need n <= ld <= 64*NPL*S, ld % 128 == 0)");
        return LOCUS_HIP_EINVAL;
    }
    const int thr = 64 * cfg.S * cfg.G;
    const int pieces = (prm.ld / 2 + thr - 1) / thr;  // 16-byte pieces of a column per thread
    const bool wide = pieces > 2;
    if (pieces > 4 || (wide && cfg.NPL != 16)) {
        set_error("sweep: workgroup too small to stage an X column (G >= NPL/4; NPL = 16: G >= NPL/8)");
        return LOCUS_HIP_EINVAL;
    }
    const long n_wg = ((long)prm.d + (long)cfg.G * cfg.C - 1) / ((long)cfg.G * cfg.C);
    dim3 grid((unsigned)n_wg), block((unsigned)thr);
    const size_t lds = sweep_lds_bytes(cfg);
    if (wide)
        return logit ? launch_sweep_wide_npl16_l1(prm, cfg.C, grid, block, lds, stream)
                     : launch_sweep_wide_npl16_l0(prm, cfg.C, grid, block, lds, stream);
#define LOCUS_CASE(npl)                                                                        \
    case npl:                                                                                  \
        return logit ? launch_sweep_npl##npl##_l1(prm, cfg.C, grid, block, lds, stream)       \
                     : launch_sweep_npl##npl##_l0(prm, cfg.C, grid, block, lds, stream);
    switch (cfg.NPL) {
        LOCUS_CASE(2)
        LOCUS_CASE(4)
        LOCUS_CASE(8)
        LOCUS_CASE(16)
        default:
            set_error("sweep: unsupported rows-per-lane NPL");
            return LOCUS_HIP_EUNSUPPORTED;
    }
#undef LOCUS_CASE
}

}  // namespace locus
