// Logit sweep for few resident responses (latency regime): look-ahead kernel with a dedicated
// chain wave.  Replaces src/coreLoop.cpp:115-131 (coreLogitLoop) like sweep_kernel<.., LOGIT>,
// which stays the kernel of the throughput regime (d_loc beyond one workgroup per CU).
//
// The per-covariate recurrence of the logit family is
//     U_s  = X_s' R^(s)                      R: residual weighted by W2 = 2 psi, after steps < s
//     m, gam, beta_new = f(U_s, state_s)     (element-wise, one response per lane)
//     R^(s+1) = R^(s) + (X_s o W2) delta_s   delta_s = beta_old - beta_new
// sweep_kernel runs these three phases back to back in every wave: a step costs the dot product,
// the wave reduction, a barrier, the element-wise chain and the rank-1 update in sequence
// (~2500 cycles at the C4 shape).  Here the row work runs ONE STEP AHEAD of the chain:
//     U_{s+1} = X_{s+1}' R^(s) + (X_{s+1}' (X_s o W2)) delta_s = p_{s+1} + c_{s+1} delta_s
// so while the chain wave evaluates step s, the six vector waves apply delta_{s-1}, form
// X_s o W2 and reduce p_{s+1}, c_{s+1} over their rows.  One barrier per step; the step time is
// the longer of the chain (LDS read of 2 x 6 partials, ~60 fp64 instructions) and the vector side
// (4 fp64 instructions per row register and response, one transposing wave reduction, the column
// prefetch), instead of their sum.  A wave issues one fp64 instruction per ~8 cycles whatever
// their dependences: a role's time is its instruction count.  The reduction is 21 instructions
// per vector wave, or 9 with the last four levels done once on the chain wave (R16, where the
// vector waves bind).
//
// Geometry: workgroup = 8 waves = 6 vector waves + the state wave (6) + the chain wave (7), CR in
// {1, 2, 4} responses per workgroup (one workgroup per CU: d_loc <= CR * CUs), lane l of vector
// wave v holds NPL rows of the wave's slice (consecutive rows in pairs: 16-byte loads) of R, W2 and
// the current X o W2 in VGPRs for the whole sweep.
// X columns are read straight from the packed matrix (scalar base + 32-bit lane offsets, coalesced
// 1 KB rows per wave) through a register ring D steps ahead; per-(j, k) state (beta_old, sig2_beta,
// xwx, log sig2_beta / 2) through a ring of the state wave, which also stores the results.
#include "sweep_kernel.h"

#include <cstddef>
#include <type_traits>
#include <utility>

namespace locus {
namespace {

template <typename F, int... I>
__device__ __forceinline__ void lfor_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void lfor(F&& f)
{
    if constexpr (N > 0) lfor_impl(f, std::make_integer_sequence<int, N>{});
}

#ifdef LOCUS_STAMPS
#define LA_NOW(t_)                                                                       \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
// busy: barrier release -> barrier arrival; wait: time inside the barrier
#define LA_BARRIER()                                                                     \
    do {                                                                                 \
        unsigned long long ta_, tb_;                                                     \
        LA_NOW(ta_);                                                                     \
        __syncthreads();                                                                 \
        LA_NOW(tb_);                                                                     \
        t_busy += ta_ - t_last;                                                          \
        t_wait += tb_ - ta_;                                                             \
        t_last = tb_;                                                                    \
    } while (0)
#define LA_MARK(k_)                                                                      \
    do {                                                                                 \
        unsigned long long tm_;                                                          \
        LA_NOW(tm_);                                                                     \
        t_mark[k_] += tm_ - t_last;                                                      \
    } while (0)
#else
#define LA_BARRIER() __syncthreads()
#define LA_MARK(k_) do { } while (0)
#endif

constexpr int kLaVec = 6;  // vector waves per workgroup (wave 6: state duties, wave 7: chain)

constexpr int kLaUnroll = 8;  // steps per loop trip of every role (ring slots are static)

template <int NPL, int CR>
constexpr int la_depth()
{
    // r, w, xw (3 CR) and a ring of D columns, NPL doubles each, must leave ~70 of the 256
    // VGPRs to the reductions and addresses
    return ((3 * CR + 8) * NPL <= 84) ? 8 : (((3 * CR + 4) * NPL <= 84) ? 4 : 2);
}

template <int CR>
struct LaShared {
    double* part;  // [parity][value 0..2CR-1][wave slot 0..7]  (slots 6, 7 stay zero)
    double* p16;   // R16: [parity][wave 0..5][value slot 0..NV-1][16]: totals over 4 lanes, the
                   // chain wave adds up the six waves and the 16 entries of a slot
    double* dl;    // [parity][c]           delta of the last step
    double* prep;  // [S % 4][c]{xb, a0, s2, bo}
    double* res;   // [s % 8][c]{gam, mu}
};

// ---- vector waves 0..5: rows of R, W2, X o W2 -------------------------------------------------
// Value slots of p16: CR <= 2 -> {p_0, p_1, c_0, c_1} (c sits 32 lanes above p on the chain wave),
// CR = 4 -> {p_0..p_3, c_0..c_3}
template <int CR>
constexpr int la_slots() { return CR <= 2 ? 4 : 8; }

// first two levels of reduce4: lane l holds the sum over lanes l % 16 + {0, 16, 32, 48} of the value
// its DPP row ends up with
__device__ __forceinline__ double reduce4_rows(double a0, double a1, double a2, double a3)
{
    return swap16_add(swap32_add(a0, a1), swap32_add(a2, a3));
}

// lanes < 32: the value of lane + 32 (v_permlane32_swap exchanges lanes 32..63 of its first operand
// with lanes 0..31 of its second and returns both)
__device__ __forceinline__ double upper_half(double v)
{
    const u32x2 lo = __builtin_amdgcn_permlane32_swap((u32)__double2loint(v), (u32)__double2loint(v),
                                                      false, false);
    const u32x2 hi = __builtin_amdgcn_permlane32_swap((u32)__double2hiint(v), (u32)__double2hiint(v),
                                                      false, false);
    return mk_double(hi.y, lo.y);
}

template <int NPL, int CR, int R16>
__device__ __forceinline__ void la_vector_role(const SweepParams& P, const LaShared<CR>& L,
                                               const int wave, const int rowbase, const int lane,
                                               const int k0, const int nsteps_pad)
{
    constexpr int D = la_depth<NPL, CR>();  // X ring depth (steps ahead)
    constexpr int NK = 2 * CR;
    constexpr int UN = kLaUnroll;
    const int n = P.n, d = P.d, ld = P.ld;
    // Row of register i of this lane: consecutive rows in pairs, so that a column is fetched with
    // 16-byte loads (half the load instructions; the dot products do not care which lane holds
    // which row).  An odd NPL keeps one 8-byte load for its last register.
    auto row_of = [&](int i) {
        return (NPL % 2 == 1 && i == NPL - 1) ? rowbase + (NPL - 1) * 64 + lane
                                              : rowbase + (i >> 1) * 128 + 2 * lane + (i & 1);
    };
    double r[CR][NPL], w[CR][NPL], xw[CR][NPL];
#pragma unroll
    for (int cc = 0; cc < CR; ++cc) {
        const int kk = k0 + cc;
        const double* Rk = P.R + (size_t)kk * n;
        const double* Wk = P.W2 + (size_t)kk * n;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int row = row_of(i);
            const bool ok = (kk < d) && (row < n);
            r[cc][i] = ok ? Rk[row] : 0.0;
            w[cc][i] = ok ? Wk[row] : 0.0;
        }
    }
    // value index whose total this lane's DPP row receives from reduce4 (either orientation
    // of the permlane swaps)
    const int val = (int)(reduce4(0.0, 1.0 / 64.0, 2.0 / 64.0, 3.0 / 64.0) + 0.5);
    const bool pw = (lane & 15) == 0;
    constexpr int NV = la_slots<CR>();

    // byte offsets of this lane's rows inside a packed column; rows past the column's padding
    // (the workgroup may cover more than ld rows) are clamped: r = w = 0 there, so whatever
    // finite value is read contributes nothing
    unsigned boff[NPL];   // of a pair: the even register's entry (ld is even: pairs never straddle it)
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int row = row_of(i);
        const bool paired = !(NPL % 2 == 1 && i == NPL - 1);
        boff[i] = 8u * (unsigned)(row < ld ? row : (paired ? ld - 2 + (i & 1) : ld - 1));
    }
    auto load_col = [&](int j, double (&x)[NPL]) {
        const char* col = reinterpret_cast<const char*>(
            P.Xp + (size_t)__builtin_amdgcn_readfirstlane(j) * ld);  // uniform: scalar base
        // scalar base + 32-bit lane offset: the empty asm keeps the offsets 32 bits wide inside
        // the loop (hoisted, they become 64-bit pairs and every load a 64-bit vector add)
#pragma unroll
        for (int i = 0; i + 1 < NPL; i += 2) {
            asm volatile("" : "+v"(boff[i]));
            const double2 v = *reinterpret_cast<const double2*>(col + boff[i]);
            x[i] = v.x;
            x[i + 1] = v.y;
        }
        if constexpr (NPL % 2 == 1) {
            asm volatile("" : "+v"(boff[NPL - 1]));
            x[NPL - 1] = *reinterpret_cast<const double*>(col + boff[NPL - 1]);
        }
    };
    // step records through VECTOR loads (lane-uniform address): a scalar load would share
    // lgkmcnt with the LDS traffic of the step, and every LDS wait would then also wait for a
    // K$ miss; vmcnt retires in order behind the ring, several steps later
    unsigned vz = 0;
    asm volatile("" : "+v"(vz));
    const StepRec* const sg = P.steps;
    // (uniform record address in SGPRs + a 32-bit zero the compiler cannot see through: a vector
    // load with no address arithmetic on the vector ALU)
    auto rec_j = [&](int step) {
        asm volatile("" : "+v"(vz));
        return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(sg + step) + vz +
                                             offsetof(StepRec, j));
    };
    // wave totals of p_c = sum_i x_i r_ci and c_c = sum_i x_i xw_ci (this wave's rows) -> part[par]
    auto reduce_write = [&](const double (&pv)[CR], const double (&cv)[CR], int par) {
        if constexpr (R16) {
            // two levels here (9 instructions), the other four on the chain wave: every lane
            // stores its row's value, no lane mask
            double* pb = L.p16 + ((par * kLaVec + wave) * NV + val) * 16 + (lane & 15);
            if constexpr (CR == 1) {
                pb[0] = reduce4_rows(pv[0], 0.0, cv[0], 0.0);
            } else if constexpr (CR == 2) {
                pb[0] = reduce4_rows(pv[0], pv[1], cv[0], cv[1]);
            } else {
                pb[0] = reduce4_rows(pv[0], pv[1], pv[2], pv[3]);
                pb[4 * 16] = reduce4_rows(cv[0], cv[1], cv[2], cv[3]);
            }
            return;
        }
        double* pb = L.part + par * NK * 8 + wave;
        if constexpr (CR == 1) {
            const double u = reduce4(pv[0], cv[0], 0.0, 0.0);
            if (pw && val < 2) pb[val * 8] = u;
        } else if constexpr (CR == 2) {
            const double u = reduce4(pv[0], pv[1], cv[0], cv[1]);
            if (pw) pb[val * 8] = u;
        } else {
            const double u = reduce4(pv[0], pv[1], pv[2], pv[3]);
            const double v = reduce4(cv[0], cv[1], cv[2], cv[3]);
            if (pw) {
                pb[val * 8] = u;
                pb[(4 + val) * 8] = v;
            }
        }
    };

    // xr[v % D]: column of step v, for v = s .. s + D - 1;  jv[v % D]: index of step v + D
    double xr[D][NPL];
    int jv[D];
    lfor<D>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        load_col(rec_j(u), xr[u]);
        jv[u] = rec_j(u + D);
    });
#pragma unroll
    for (int cc = 0; cc < CR; ++cc)
#pragma unroll
        for (int i = 0; i < NPL; ++i) xw[cc][i] = 0.0;
    {   // p_0 (c_0 = 0: no step before the first)
        double pv[CR], cv[CR];
#pragma unroll
        for (int cc = 0; cc < CR; ++cc) {
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < NPL; ++i) a = fma(xr[0][i], r[cc][i], a);
            pv[cc] = a;
            cv[cc] = 0.0;
        }
        reduce_write(pv, cv, 0);
    }
    __syncthreads();
#ifdef LOCUS_STAMPS
    unsigned long long t_busy = 0, t_wait = 0, t_last, t_mark[6] = {0, 0, 0, 0, 0, 0};
    LA_NOW(t_last);
#endif

    for (int s0 = 0; s0 < nsteps_pad; s0 += UN) {
        lfor<UN>([&](auto uc) {
            constexpr int uu = decltype(uc)::value;
            constexpr int u = uu % D;
            const int s = s0 + uu;
            constexpr int un = (u + 1) % D;  // slot of X_{s+1}
            // The delta of the last step arrives through LDS right after the barrier: what does
            // not depend on it goes first and covers the read's latency.
            double dlt[CR];
#pragma unroll
            for (int cc = 0; cc < CR; ++cc) dlt[cc] = L.dl[((s + 1) & 1) * CR + cc];
            // (b) xn = X_s o W2;  c = X_{s+1}' xn
            double pv[CR], cv[CR], xn[CR][NPL];
#pragma unroll
            for (int cc = 0; cc < CR; ++cc) {
                double ca = 0.0;
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    xn[cc][i] = xr[u][i] * w[cc][i];
                    ca = fma(xr[un][i], xn[cc][i], ca);
                }
                cv[cc] = ca;
            }
            __builtin_amdgcn_sched_barrier(0);
            // (a) R += (X_{s-1} o W2) delta_{s-1};  p = X_{s+1}' R
#pragma unroll
            for (int cc = 0; cc < CR; ++cc) {
                double pa = 0.0;
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    r[cc][i] = fma(xw[cc][i], dlt[cc], r[cc][i]);
                    pa = fma(xr[un][i], r[cc][i], pa);
                    xw[cc][i] = xn[cc][i];
                }
                pv[cc] = pa;
            }
            LA_MARK(0);
            reduce_write(pv, cv, (s + 1) & 1);
            LA_MARK(1);
            // slot u is free: column of step s + D, index of step s + 2 D
            load_col(jv[u], xr[u]);
            jv[u] = rec_j(s + 2 * D);
            LA_BARRIER();
        });
    }
#ifdef LOCUS_STAMPS
    if (wave == 0 && lane == 0 && blockIdx.x < 512) {
        const int o = blockIdx.x * 24;
        g_sweep_stamps[o + 0] = t_busy;
        g_sweep_stamps[o + 1] = t_wait;
        for (int q = 0; q < 2; ++q) g_sweep_stamps[o + 2 + q] = t_mark[q];
    }
    if ((wave == 2 || wave == 3) && lane == 0 && blockIdx.x < 512) {  // the waves beside wave 6 / the chain
        const int o = blockIdx.x * 24 + 4 + 2 * (wave - 2);
        g_sweep_stamps[o + 0] = t_busy;
        g_sweep_stamps[o + 1] = t_wait;
    }
#endif
    // last rank-1 update (the loop's final barrier published the delta of its last step)
    {
        double dlt[CR];
#pragma unroll
        for (int cc = 0; cc < CR; ++cc) dlt[cc] = L.dl[((nsteps_pad - 1) & 1) * CR + cc];
#pragma unroll
        for (int cc = 0; cc < CR; ++cc)
#pragma unroll
            for (int i = 0; i < NPL; ++i) r[cc][i] = fma(xw[cc][i], dlt[cc], r[cc][i]);
    }
    // ---- epilogue: residual back to memory, sum of squares ----
    double ss[CR];
#pragma unroll
    for (int cc = 0; cc < CR; ++cc) {
        const int kk = k0 + cc;
        double* Rk = P.R + (size_t)kk * n;
        double a0 = 0.0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int row = row_of(i);
            if ((kk < d) && (row < n)) Rk[row] = r[cc][i];
            a0 = fma(r[cc][i], r[cc][i], a0);
        }
        ss[cc] = a0;
    }
    __syncthreads();  // every wave is past its last read of `part`
    {
        double* pb = L.part + wave;
        if constexpr (CR == 1) {
            const double u = reduce4(ss[0], 0.0, 0.0, 0.0);
            if (pw && val < 1) pb[val * 8] = u;
        } else if constexpr (CR == 2) {
            const double u = reduce4(ss[0], ss[1], 0.0, 0.0);
            if (pw && val < 2) pb[val * 8] = u;
        } else {
            const double u = reduce4(ss[0], ss[1], ss[2], ss[3]);
            if (pw) pb[val * 8] = u;
        }
    }
    __syncthreads();
}

// ---- wave 6: every global access of the per-response state -----------------------------------
// Every step t it
//   1. stores gam / mu / beta of step t - 1 (left in `res` by the chain) and adds them to the
//      running statistics,
//   2. fetches the state (beta_old, sig2_beta, xwx, log sig2_beta / 2) of step t + 7 and the
//      record of step t + 15,
//   3. prepares the operands of step t + 2 that do not depend on U (prep[(t + 2) % 4]).
// Lane l < 4 CR serves array a = l / CR of response c = l % CR:
//   a = 0: beta_vb (old value; stores gam_vb)      a = 1: sig2_beta_vb (stores mu_beta_vb)
//   a = 2: xwx (stores the new beta_vb)            a = 3: log(sig2_beta_vb) / 2
// so that the four state loads of a step are ONE load instruction, the three stores one store,
// and the three statistics one accumulator per lane.  Ring slot v % 8 <-> step v, v = t .. t + 7.
template <int CR>
__device__ __forceinline__ void la_helper_role(const SweepParams& P, const LaShared<CR>& L,
                                               const int lane, const int k0, const int nsteps,
                                               const int nsteps_pad)
{
    constexpr int UN = kLaUnroll;
    const int d = P.d;
    const int ha = (lane < 4 * CR) ? lane / CR : 0;
    const int c = lane % CR;
    const int k = k0 + c;
    const bool kvalid = (lane < 4 * CR) && (k < d);
    const int ks = (k < d) ? k : d - 1;  // safe column for speculative loads
    const size_t pd = (size_t)P.p * d;
    const bool has_ls2 = P.ls2 != nullptr;  // uniform
    const double* const arr_base =
        (ha == 0 ? P.state + 2 * pd : ha == 1 ? P.s2 : ha == 2 ? P.xwx : (has_ls2 ? P.ls2 : P.s2)) + ks;
    double* const store_base = P.state + (size_t)(ha < 3 ? ha : 0) * pd + ks;
    // slot of this lane's value in prep[S % 4][c]{xb, a0, s2, bo}
    const int pslot = ha == 0 ? 3 : ha == 1 ? 2 : ha == 2 ? 0 : 1;
    int vz = 0;
    asm volatile("" : "+v"(vz));  // step records through vector loads (see la_vector_role)
    const StepRec* const sg = P.steps;
    double h_val[UN], h_lod[UN], n_lod[UN];
    int h_j[UN], n_j[UN];
    double acc = 0.0;
    auto prepare = [&](int S, double v, double lod) {
        // beta_old of the same response sits 2 CR lanes below the xwx lane (same DPP row)
        const int blo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + 2 * CR, 0xF, 0xF, true);
        const int bhi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + 2 * CR, 0xF, 0xF, true);
        const double xb = v * __hiloint2double(bhi, blo);                      // xwx * beta_old
        // a0 = lod - log_sig2_inv / 2 - log(sig2_beta) / 2                 (coreLoop.cpp:123-125)
        const double hl = has_ls2 ? v : 0.5 * log_pos_normal(ha == 3 ? v : 1.0);
        const double a0 = (lod - P.half_lsi) - hl;
        const double out = ha < 2 ? v : (ha == 2 ? xb : a0);
        if (lane < 4 * CR) L.prep[((S & 3) * CR + c) * 4 + pslot] = out;
    };
    // results of step q ({gam, mu} read from `res` by the caller, early): one store and one
    // statistics update per lane
    auto store_step = [&](int q, int jq, double myval, double2 gmm) {
        if (kvalid && ha < 3 && q >= 0 && q < nsteps) {
            const double gm = gmm.x, m = gmm.y;
            const double bn = m * gm;                                         // coreLoop.cpp:127
            store_base[(size_t)jq * d] = ha == 0 ? gm : (ha == 1 ? m : bn);
            // sum gam | sum gam (mu^2 + sig2_beta) | sum beta^2   (myval = sig2_beta for a = 1)
            acc += ha == 0 ? gm : (ha == 1 ? gm * fma(m, m, myval) : bn * bn);
        }
    };
    lfor<UN>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        // slot 7 <-> step -1: filled with step 7 in iteration 0
        if constexpr (u < UN - 1) {
            h_lod[u] = sg[u + vz].lod;
            h_j[u] = sg[u + vz].j;
            h_val[u] = arr_base[(size_t)h_j[u] * d];
        } else {
            h_lod[u] = 0.0;
            h_j[u] = 0;
            h_val[u] = 1.0;
        }
        // record ring: slot (t - 1) % 8 holds, in iteration t, the record of step t + 7
        n_lod[(u + UN - 1) % UN] = sg[u + UN - 1 + vz].lod;
        n_j[(u + UN - 1) % UN] = sg[u + UN - 1 + vz].j;
    });
    prepare(0, h_val[0], h_lod[0]);
    prepare(1, h_val[1], h_lod[1]);
    // Every load of the prologue has landed before the loop is entered: the waits inside the
    // loop are then those of the steady state (a load still in flight at the loop's entry would
    // make the compiler's vmcnt bookkeeping wait for the youngest loads in every trip's first steps).
#pragma unroll
    for (int u = 0; u < UN; ++u)
        asm volatile("" : "+v"(h_val[u]), "+v"(h_lod[u]), "+v"(h_j[u]), "+v"(n_lod[u]), "+v"(n_j[u]));
    __syncthreads();
#ifdef LOCUS_STAMPS
    unsigned long long t_busy = 0, t_wait = 0, t_last, t_mark[6] = {0, 0, 0, 0, 0, 0};
    LA_NOW(t_last);
#endif

    for (int s0 = 0; s0 < nsteps_pad; s0 += UN) {
        lfor<UN>([&](auto uc) {
            constexpr int uu = decltype(uc)::value;
            const int s = s0 + uu;
            constexpr int up = (uu + UN - 1) % UN;  // slot of step s - 1
            // {gam, mu} of step s - 1: read now, used last ((s - 1) & 7 = 7 for s = 0: unused)
            const double2 gmm =
                *reinterpret_cast<const double2*>(L.res + (((s - 1) & 7) * CR + c) * 2);
            const int j_prev = h_j[up];
            const double v_prev = h_val[up];
            // 1. operands of step s + 2
            constexpr int u2 = (uu + 2) % UN;
            prepare(s + 2, h_val[u2], h_lod[u2]);
            LA_MARK(0);
            // 2. results of step s - 1: the last use of slot `up` as step s - 1
            store_step(s - 1, j_prev, v_prev, gmm);
            // 3. the slot now belongs to step s + 7; record of step s + 15.  The old contents are
            // dead before the loads are issued and the hand-over of the record is a real move:
            // otherwise the compiler lets every load of a trip land in a fresh register and
            // copies them into place at the end of the trip, behind s_waitcnt vmcnt(0) -- the
            // whole prefetch ring drained once per 8 steps.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("v_mov_b32 %0, %1" : "=v"(h_j[up]) : "v"(n_j[up]));
            {
                int lo_ = __double2loint(n_lod[up]), hi_ = __double2hiint(n_lod[up]), lo2, hi2;
                asm volatile("v_mov_b32 %0, %1" : "=v"(lo2) : "v"(lo_));
                asm volatile("v_mov_b32 %0, %1" : "=v"(hi2) : "v"(hi_));
                h_lod[up] = __hiloint2double(hi2, lo2);
            }
            __builtin_amdgcn_sched_barrier(0);
            h_val[up] = arr_base[(size_t)h_j[up] * d];
            n_lod[up] = sg[s + 2 * UN - 1 + vz].lod;
            n_j[up] = sg[s + 2 * UN - 1 + vz].j;
            LA_MARK(1);
            LA_BARRIER();
        });
    }
#ifdef LOCUS_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        const int o = blockIdx.x * 24 + 16;
        g_sweep_stamps[o + 0] = t_busy;
        g_sweep_stamps[o + 1] = t_wait;
        for (int q = 0; q < 6; ++q) g_sweep_stamps[o + 2 + q] = t_mark[q];
    }
#endif
    // results of the last step (only when no pad step followed it)
    store_step(nsteps_pad - 1, h_j[UN - 1], h_val[UN - 1],
               *reinterpret_cast<const double2*>(L.res + (((nsteps_pad - 1) & 7) * CR + c) * 2));
    __syncthreads();  // vector waves: residual stored, part[0] free
    __syncthreads();  // sums of squares in part[0][c][0..5]
    if (P.st != nullptr && kvalid) {
        double out = acc;
        if (ha == 3) {  // sum_i R_ik^2
            out = 0.0;
            for (int v = 0; v < kLaVec; ++v) out += L.part[c * 8 + v];
        }
        P.st[(size_t)ha * d + k] = out;
    }
}

// ---- chain wave ------------------------------------------------------------------------------
// No global access at all: operands come prepared through `prep`, results leave through `dl`
// (delta, read by every vector wave) and `res` (gam, mu for wave 6).
// R16 = 0: one response per lane (lanes < CR), the vector waves deliver wave totals.
// R16 = 1: the vector waves stop their reduction at 16 entries per value (reduce4_rows) and this
// wave finishes it: lane l adds up the six waves' entry l % 16 of value slot l / 16 and the four
// DPP levels give every lane of row l / 16 that value's total; response c then lives in row c
// (all 16 lanes compute it, lane 16 c writes).  12 cross-lane instructions leave each of the six
// vector waves for 19 here (5 adds over the waves instead of 10: the four values are summed side
// by side) -- where the vector waves bind (n large), not where the chain does.
template <int CR, int R16>
__device__ __forceinline__ void la_chain_role(const SweepParams& P, const LaShared<CR>& L,
                                              const int lane, const int k0, const int nsteps,
                                              const int nsteps_pad)
{
    constexpr int NK = 2 * CR;
    constexpr int NV = la_slots<CR>();
    const int row = lane >> 4;
    const int c = R16 ? (row < CR ? row : 0) : (lane < CR ? lane : 0);
    const bool mine = R16 ? (row < CR) : (lane < CR);          // lanes that run response c
    const bool writer = R16 ? (mine && (lane & 15) == 0) : mine;
    const bool valid = mine && (k0 + c < P.d);
    double dprev = 0.0;
    // the chain is the critical path of every step: win the issue arbitration against the vector
    // wave that shares this SIMD
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();
#ifdef LOCUS_STAMPS
    unsigned long long t_busy = 0, t_wait = 0, t_last, t_mark[6] = {0, 0, 0, 0, 0, 0};
    LA_NOW(t_last);
#endif
    for (int s = 0; s < nsteps_pad; ++s) {
        const bool act = valid && (s < nsteps);
        const double2* qp = reinterpret_cast<const double2*>(L.prep + ((s & 3) * CR + c) * 4);
        // U = sum_w p_w + (sum_w c_w) delta_{s-1}
        double U;
        double2 q0, q1;
        if constexpr (R16) {
            const double* src = L.p16 + (s & 1) * (kLaVec * NV * 16) + lane;   // slot row, entry lane % 16
            double t[kLaVec], t2[kLaVec];
#pragma unroll
            for (int w = 0; w < kLaVec; ++w) t[w] = src[w * NV * 16];
            if constexpr (CR == 4) {
#pragma unroll
                for (int w = 0; w < kLaVec; ++w) t2[w] = src[w * NV * 16 + 64];
            }
            q0 = qp[0];
            q1 = qp[1];
            const double ps = row_allreduce(((t[0] + t[1]) + (t[2] + t[3])) + (t[4] + t[5]));
            double cs;
            if constexpr (CR == 4)
                cs = row_allreduce(((t2[0] + t2[1]) + (t2[2] + t2[3])) + (t2[4] + t2[5]));
            else
                cs = upper_half(ps);                    // rows 2, 3 hold c_0, c_1
            U = fma(cs, dprev, ps);
        } else {
            const double2* pp = reinterpret_cast<const double2*>(L.part + ((s & 1) * NK + c) * 8);
            const double2* cp = reinterpret_cast<const double2*>(L.part + ((s & 1) * NK + CR + c) * 8);
            const double2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
            const double2 p0 = pp[0], p1 = pp[1], p2 = pp[2];
            q0 = qp[0];
            q1 = qp[1];
            const double cs = ((c0.x + c0.y) + (c1.x + c1.y)) + (c2.x + c2.y);
            const double ps = ((p0.x + p0.y) + (p1.x + p1.y)) + (p2.x + p2.y);
            U = fma(cs, dprev, ps);
        }
        LA_MARK(0);
        const double bo = q1.y;
        const double dot = q0.x + U;   // xwx * beta_old + U   (coreLoop.cpp:121)
        const double m = q1.x * dot;
        // :123-125  m^2 / (2 sig2_beta) = m * dot / 2  (m = sig2_beta * dot): no reciprocal
        const double a = fma(-0.5 * m, dot, q0.y);
        const double e = exp_nonpos(-fabs(a));
        const double gm = div_1to2((a >= 0.0) ? e : 1.0, 1.0 + e);
        const double bn = m * gm;                                // :127
        const double delta = act ? (bo - bn) : 0.0;
        if (writer) {
            L.dl[(s & 1) * CR + c] = delta;
            *reinterpret_cast<double2*>(L.res + ((s & 7) * CR + c) * 2) = make_double2(gm, m);
        }
        dprev = delta;
        LA_MARK(1);
        LA_BARRIER();
    }
#ifdef LOCUS_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        g_sweep_stamps[blockIdx.x * 24 + 8] = t_busy;
        g_sweep_stamps[blockIdx.x * 24 + 9] = t_wait;
        for (int q = 0; q < 6; ++q) g_sweep_stamps[blockIdx.x * 24 + 10 + q] = t_mark[q];
    }
#endif
    __syncthreads();  // vector waves: residual stored, part[0] free
    __syncthreads();  // sums of squares in part[0][c][0..5]
}

// A wave issues at most one fp64 instruction per ~8 cycles and a SIMD one per 4: the step is
// bounded by the longest wave and by the fullest SIMD.  Waves 0, 1, 4, 5 share SIMDs 0 and 1 two by
// two; wave 2 shares SIMD 2 with the state wave, wave 3 shares SIMD 3 with the chain wave.  Uneven
// deals of the rows over the waves were measured at n = 2000, d = 500 (sweep, ms; rows per lane of
// waves 0..5):  (6,6,6,6,6,6) 7.04   (5,5,6,6,5,5) 7.04, with twice the L2 misses
// (5,5,7,5,5,5) 7.17   (5,5,8,4,5,5) 7.25  -- the longest wave binds; the deal stays even.
template <int NPL, int CR, int R16>
__global__ void __launch_bounds__(512) logit_la_kernel(const SweepParams P)
{
    constexpr int NK = 2 * CR;
    constexpr int NV = la_slots<CR>();
    __shared__ __align__(16) double part[2 * NK * 8];
    __shared__ __align__(16) double p16[R16 ? 2 * kLaVec * NV * 16 : 2];
    __shared__ double dl[2 * CR];
    __shared__ __align__(16) double prep[4 * CR * 4];
    __shared__ __align__(16) double res[8 * CR * 2];

    if (*P.err != 0) return;  // prologue found a bad shuffled_ind: uniform exit

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int k0 = blockIdx.x * CR;
    const int nsteps = P.nsteps;
    // The loops run whole trips of 8 steps: steps nsteps .. nsteps_pad - 1 are no-ops (delta = 0,
    // nothing stored).  A branch around a step would make the compiler's vmcnt bookkeeping
    // assume the shortest path and wait for the youngest prefetches instead of the oldest.
    const int nsteps_pad = (nsteps + kLaUnroll - 1) / kLaUnroll * kLaUnroll;

    if (tid < 2 * NK * 8) part[tid] = 0.0;
    if (tid < 2 * CR) dl[tid] = 0.0;
    if constexpr (R16) {
        for (int i = tid; i < 2 * kLaVec * NV * 16; i += 512) p16[i] = 0.0;   // slots no wave writes (CR = 1)
    }
    __syncthreads();

    const LaShared<CR> L{part, p16, dl, prep, res};
    if (wave < kLaVec)
        la_vector_role<NPL, CR, R16>(P, L, wave, wave * NPL * 64, lane, k0, nsteps_pad);
    else if (wave == kLaVec)
        la_helper_role<CR>(P, L, lane, k0, nsteps, nsteps_pad);
    else
        la_chain_role<CR, R16>(P, L, lane, k0, nsteps, nsteps_pad);
}

template <int NPL, int R16>
int launch_npl(const SweepParams& prm, int cr, dim3 grid, hipStream_t st)
{
    switch (cr) {
        case 1: hipLaunchKernelGGL((logit_la_kernel<NPL, 1, R16>), grid, dim3(512), 0, st, prm); break;
        case 2: hipLaunchKernelGGL((logit_la_kernel<NPL, 2, R16>), grid, dim3(512), 0, st, prm); break;
        case 4:
            if constexpr (NPL <= 6) {
                hipLaunchKernelGGL((logit_la_kernel<NPL, 4, R16>), grid, dim3(512), 0, st, prm);
                break;
            } else {
                set_error("logit look-ahead sweep: 4 responses per workgroup need n <= 2304");
                return LOCUS_HIP_EUNSUPPORTED;
            }
        default:
            set_error("logit look-ahead sweep: responses per workgroup must be 1, 2 or 4");
            return LOCUS_HIP_EINVAL;
    }
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int NPL>
int launch_red(const SweepParams& prm, int cr, int r16, dim3 grid, hipStream_t st)
{
    return r16 ? launch_npl<NPL, 1>(prm, cr, grid, st) : launch_npl<NPL, 0>(prm, cr, grid, st);
}

}  // namespace

#ifdef LOCUS_STAMPS
extern "C" int locus_hip_diag_read_la_stamps(unsigned long long* host, int count)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sweep_stamps),
                                    sizeof(unsigned long long) * (size_t)count);
}
#endif

// Smallest instantiated NPL (1..6, 8) whose workgroup covers n rows (six waves of 64 NPL rows).
static int logit_la_npl_for(int n)
{
    for (int npl : {1, 2, 3, 4, 5, 6, 8})
        if (64 * kLaVec * npl >= n) return npl;
    return 0;
}

// Largest CR instantiated for n rows (0: n not covered).
int logit_la_max_cr(int n)
{
    const int npl = logit_la_npl_for(n);
    return npl == 0 ? 0 : (npl <= 6 ? 4 : 2);
}

// Who finishes the reduction of the partial sums (1: the chain wave), as measured for every
// geometry at p = 20 000 (scripts/la_red_sweep.sh): the chain wave wherever the
// vector waves bind the step -- CR = 4 always (two reductions per step: -1 to -17 %), CR = 2 from
// 3 rows per lane (-5 to -13 %; +1 to +7 % below), CR = 1 from 5 (-1 to -4 %; +3 to +7 % below).
static int logit_la_r16_for(int npl, int cr)
{
    return cr == 4 || (cr == 2 && npl >= 3) || (cr == 1 && npl >= 5);
}

// cfg = {-3, CR, RED, NPL}: CR = 0 -> smallest of 1, 2, 4 with one workgroup per CU (4 beyond
// that); NPL = 0 -> smallest instantiated NPL that covers n; RED = 0 -> the reduction is finished
// where it was measured faster, 1 -> by the vector waves, 2 -> by the chain wave.
int run_logit_la_sweep(const SweepParams& prm, const SweepConfig& cfg, hipStream_t st)
{
    int npl = cfg.G;
    if (npl == 0) npl = logit_la_npl_for(prm.n);
    if (cfg.S < 0 || cfg.S > 2) {
        set_error("logit look-ahead sweep: the reduction mode (cfg[2]) must be 0, 1 or 2");
        return LOCUS_HIP_EINVAL;
    }
    if (npl < 1 || npl > 8 || 64 * kLaVec * npl < prm.n) {
        set_error("logit look-ahead sweep: rows per lane do not cover n (n <= 3072)");
        return LOCUS_HIP_EINVAL;
    }
    int cr = cfg.C;
    if (cr == 0) {
        const int cus = device_cus();
        cr = prm.d <= cus ? 1 : (prm.d <= 2 * cus ? 2 : 4);
        if (cr > logit_la_max_cr(prm.n)) cr = logit_la_max_cr(prm.n);
    }
    const int r16 = cfg.S == 0 ? logit_la_r16_for(npl, cr) : cfg.S - 1;
    if (prm.ld < prm.n || (prm.ld & 127) != 0 || (reinterpret_cast<uintptr_t>(prm.Xp) & 15) != 0) {
        set_error("logit look-ahead sweep: packed X needs ld >= n, ld % 128 == 0, 16-byte alignment");
        return LOCUS_HIP_EINVAL;
    }
    const dim3 grid((unsigned)((prm.d + cr - 1) / cr));
    switch (npl) {
        case 1: return launch_red<1>(prm, cr, r16, grid, st);
        case 2: return launch_red<2>(prm, cr, r16, grid, st);
        case 3: return launch_red<3>(prm, cr, r16, grid, st);
        case 4: return launch_red<4>(prm, cr, r16, grid, st);
        case 5: return launch_red<5>(prm, cr, r16, grid, st);
        case 6: return launch_red<6>(prm, cr, r16, grid, st);
        case 8: return launch_red<8>(prm, cr, r16, grid, st);
        default:
            set_error("logit look-ahead sweep: unsupported rows per lane (1..6, 8)");
            return LOCUS_HIP_EUNSUPPORTED;
    }
}

}  // namespace locus
