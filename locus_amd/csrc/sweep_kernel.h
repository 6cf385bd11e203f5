// CAVI sweep kernel for gfx950 (MI355X) -- register-resident residual.
//
// Replaces the per-covariate loops of the reference:
//   src/coreLoop.cpp:36-52 (coreLoop), :78-94 (coreZLoop), :115-131 (coreLogitLoop),
//   :154-169 (coreProbitLoop), src/coreStructLoop.cpp:34-50 (coreStructLoop).
//
// Design:
//  * responses are independent inside a sweep, so a wave owns C response columns
//    of the residual R = Y - mat_x_m1 (- mat_z_mu) for the WHOLE sweep, in VGPRs:
//    lane l of row-slice s holds rows  s*64*NPL + t*128 + 2l + {0,1},  t < NPL/2;
//  * per covariate j (strictly sequential): X[,j] is staged once per workgroup in
//    LDS (3-deep ring, prefetched two steps ahead), every wave reads its slice,
//    forms C partial dot products, reduces them across the 64 lanes with
//    v_permlane32_swap / v_permlane16_swap (two "transposing" levels that leave
//    response c's sum in DPP row r(c)) and four DPP levels inside the rows,
//    then ONE pass of the element-wise spike-and-slab update runs with the
//    responses spread over the lanes, and the rank-1 update is applied from
//    SGPR-broadcast deltas;
//  * the reference's remove / dot / add-back triple is folded algebraically:
//       X_j'(Y - (M - X_j b_old)) = X_j' R + (X_j'X_j) b_old,   R += X_j (b_old - b_new)
//    (the same identity the reference uses in src/coreBatch.cpp:32-33);
//  * fused epilogue: column sums of gam, m2_beta, beta^2 and ||R_k||^2, i.e. the
//    reductions update_vb.R:220,229-237 and elbo.R need.
#pragma once
#include "common.h"
#include "fastmath.h"

namespace locus {

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double mk_double(u32 hi, u32 lo)
{
    return __hiloint2double((int)hi, (int)lo);
}

// After the call: lanes 0..31 hold a[l] + a[l+32], lanes 32..63 hold b[l-32] + b[l]
// (or the two roles exchanged -- the kernel calibrates the mapping at run time).
__device__ __forceinline__ double swap32_add(double a, double b)
{
    u32x2 lo = __builtin_amdgcn_permlane32_swap((u32)__double2loint(a), (u32)__double2loint(b),
                                                false, false);
    u32x2 hi = __builtin_amdgcn_permlane32_swap((u32)__double2hiint(a), (u32)__double2hiint(b),
                                                false, false);
    return mk_double(hi.x, lo.x) + mk_double(hi.y, lo.y);
}

// Same across pairs of 16-lane rows: even rows get the pair-sum of a, odd rows of b.
__device__ __forceinline__ double swap16_add(double a, double b)
{
    u32x2 lo = __builtin_amdgcn_permlane16_swap((u32)__double2loint(a), (u32)__double2loint(b),
                                                false, false);
    u32x2 hi = __builtin_amdgcn_permlane16_swap((u32)__double2hiint(a), (u32)__double2hiint(b),
                                                false, false);
    return mk_double(hi.x, lo.x) + mk_double(hi.y, lo.y);
}

template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return v + __hiloint2double(hi, lo);
}

// Sum over the 16 lanes of each DPP row; every lane of the row gets the total.
__device__ __forceinline__ double row_allreduce(double v)
{
    v = dpp_add<0xB1>(v);   // quad_perm:[1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}

// Reduce up to four per-lane values over the wave.  DPP row r ends up holding the
// total of value rowval(r) in all of its 16 lanes.
__device__ __forceinline__ double reduce4(double a0, double a1, double a2, double a3)
{
    double t01 = swap32_add(a0, a1);
    double t23 = swap32_add(a2, a3);
    double u = swap16_add(t01, t23);
    return row_allreduce(u);
}

__device__ __forceinline__ double bcast_lane(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int C>
struct Geo {
    static constexpr int NG = (C + 3) / 4;                       // groups of four responses
    static constexpr int NGP = NG <= 1 ? 1 : (NG <= 2 ? 2 : 4);  // padded to a power of two
    static constexpr int SUBW = 16 / NGP;                        // lanes of a row per group
};

// Reduce C per-lane partial sums; lane l receives the total of "its" response
// (group sub = (l & 15) / SUBW, value = the one its DPP row ends up with).
template <int C>
__device__ __forceinline__ double reduce_merge(const double (&v)[C], int sub)
{
    constexpr int NG = Geo<C>::NG;
    double U = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double a0 = v[4 * g];
        const double a1 = (4 * g + 1 < C) ? v[(4 * g + 1 < C) ? 4 * g + 1 : 0] : 0.0;
        const double a2 = (4 * g + 2 < C) ? v[(4 * g + 2 < C) ? 4 * g + 2 : 0] : 0.0;
        const double a3 = (4 * g + 3 < C) ? v[(4 * g + 3 < C) ? 4 * g + 3 : 0] : 0.0;
        const double u = reduce4(a0, a1, a2, a3);
        U = (g == 0 || sub == g) ? u : U;
    }
    return U;
}

constexpr int kMaxPieces = 2;  // 16-byte X pieces a thread stages per column (G >= NPL/4);
                               // the "wide" instantiations (PCS = 4) cover 2048 < n <= 4096
constexpr int kStepPad = 12;   // nsteps is padded to a multiple of every prefetch depth
constexpr int kStepTail = 16;  // readable step records past nsteps_pad

// Prefetch depth D (= unroll factor of the main loop).  Every global load of the loop is a
// prefetch for step i + D and the loads are consumed in the order they were issued, because
// vmcnt retires in order: a wait for a young load would also wait for every older one, so a
// short-distance load behind a long-distance HBM miss would serialise the sweep on HBM
// latency.  Small geometries (latency regime) have the registers for a deep ring.
template <int NPL, int C, bool LOGIT, int PCS = kMaxPieces>
constexpr int sweep_depth()
{
    constexpr int regs = 2 * NPL * C * (LOGIT ? 2 : 1);
    // logit keeps four per-pair rings (beta_old, sig2_beta, xwx, log sig2_beta / 2) per slot
    if (PCS > kMaxPieces) return (regs <= 64 && !LOGIT) ? 4 : 2;  // 4 PCS registers per ring slot
    if (LOGIT && regs > 48) return regs <= 112 ? 4 : 2;
    return regs <= 64 ? 6 : (regs <= 112 ? 4 : 2);
}

// Every instantiation is built for <= 256 VGPRs (two waves per SIMD): arithmetic
// VGPRs stop at 256 on gfx950, anything above lives in AGPRs behind
// v_accvgpr moves, which costs more than it buys for an FMA-bound loop.
template <int NPL, int C, bool LOGIT, int PCS = kMaxPieces>
constexpr int sweep_max_threads()
{
    constexpr int est = 2 * NPL * C * (LOGIT ? 2 : 1) +
                        4 * PCS * sweep_depth<NPL, C, LOGIT, PCS>() + 90;
    return est <= 128 ? 1024 : 512;
}

// Step records are read-only for the kernel and uniform per step: reading them through the
// constant address space makes them scalar (s_load, lgkmcnt) instead of vector loads that
// would queue behind the state prefetches in vmcnt order.
typedef const StepRec __attribute__((address_space(4))) * StepPtr;

// Diagnostic build only (-DLOCUS_STAMPS, `make diag`): per-phase cycle sums of wave 0 of every
// workgroup (s_memtime), written to g_sweep_stamps.  No stamp executes in the product build.
#ifdef LOCUS_STAMPS
constexpr int kStampSlots = 8;
__device__ unsigned long long g_sweep_stamps[kStampSlots * 4096];
#define LOCUS_STAMP(k)                                                                   \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        tacc[k] += t_ - tprev;                                                           \
        tprev = t_;                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define LOCUS_STAMP(k) do { } while (0)
#endif

template <int NPL, int C, bool LOGIT, int PCS = kMaxPieces>
__global__ void __launch_bounds__((sweep_max_threads<NPL, C, LOGIT, PCS>()))
sweep_kernel(const SweepParams P)
{
    static_assert(NPL % 2 == 0 && NPL >= 2, "NPL must be even");
    static_assert(C >= 1 && C <= 16, "1..16 responses per wave");
    constexpr int NG = Geo<C>::NG;
    constexpr int SUBW = Geo<C>::SUBW;
    constexpr int CP = 16;  // slots per wave in the cross-wave partial buffer

    extern __shared__ __align__(16) double smem[];
    double* const xring = smem;                      // 3 * npad
    double* const part = smem + 3 * (size_t)P.npad;  // 2 * G * S * CP
    // per-response constants {scale, hinv, cst, sig2_beta} and running statistics
    // {sum gam, sum m2_beta, sum beta^2, -} live in LDS, not in VGPRs: the residual
    // needs 2*NPL*C of the 256 arithmetic VGPRs
    double* const cons = part + 2 * (size_t)P.G * P.S * CP;  // G * CP * 4
    double* const accs = cons + (size_t)P.G * CP * 4;        // G * CP * 4

    if (*P.err != 0) return;  // prologue found a bad shuffled_ind: uniform exit

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int S = P.S;
    const int g = wave / S;      // column group of this wave
    const int s = wave - g * S;  // row slice of this wave
    const int n = P.n, d = P.d;
    const int k0 = (blockIdx.x * P.G + g) * C;
    const int row0 = s * 64 * NPL;
    const int sub = (lane & 15) / SUBW;
    const int rowid = lane >> 4;

    // ---- calibrate the lane roles of the permlane swaps at run time ----------
    // f32: lanes >= 32 receive the SECOND operand's sum; f16: odd rows receive the
    // SECOND operand's sum (true for the documented semantics; the code is correct
    // for either orientation).
    bool f32, f16;
    {
        const double u = reduce4(0.0, 1.0 / 64.0, 2.0 / 64.0, 3.0 / 64.0);
        const int val = (int)(u + 0.5);  // value index held by this lane's row
        const unsigned long long m1 = __ballot((rowid == 2) && (val & 1));
        const unsigned long long m2 = __ballot((rowid == 1) && (val & 2));
        f32 = (m1 != 0);
        f16 = (m2 != 0);
    }
    // value index (0..3) whose total ends up in DPP row `rowid`
    const int vrow = (((rowid >> 1) != 0) == f32 ? 1 : 0) + 2 * (((rowid & 1) != 0) == f16 ? 1 : 0);
    int cl = sub * 4 + vrow;  // response inside the wave owned by this lane
    if (sub >= NG || cl >= C) cl = C;
    const int kl = k0 + cl;
    const bool cvalid = (cl < C) && (kl < d);
    const bool pwriter = (cl < C) && ((lane & 15) % SUBW == 0);
    const bool writer = cvalid && pwriter && (s == 0);
    // first lane that owns response c (uniform, a few SALU ops)
    auto lane_of = [&](int c) -> int {
        const int v = c & 3;
        const int hi = ((v & 1) != 0) == f32 ? 1 : 0;
        const int odd = ((v & 2) != 0) == f16 ? 1 : 0;
        return 32 * hi + 16 * odd + (c >> 2) * SUBW;
    };

    // ---- residual (and logit weights) into registers ------------------------
    double r[C][NPL];
    double w[LOGIT ? C : 1][LOGIT ? NPL : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = k0 + c;
        const double* Rk = P.R + (size_t)k * n;
#pragma unroll
        for (int t = 0; t < NPL / 2; ++t) {
            const int row = row0 + t * 128 + 2 * lane;
            const bool ok0 = (k < d) && (row < n);
            const bool ok1 = (k < d) && (row + 1 < n);
            r[c][2 * t] = ok0 ? Rk[row] : 0.0;
            r[c][2 * t + 1] = ok1 ? Rk[row + 1] : 0.0;
            if constexpr (LOGIT) {
                const double* Wk = P.W2 + (size_t)k * n;
                w[c][2 * t] = ok0 ? Wk[row] : 0.0;
                w[c][2 * t + 1] = ok1 ? Wk[row + 1] : 0.0;
            }
        }
    }

    // ---- per-response constants and statistics accumulators (LDS) -------------
    double* const mycons = cons + ((size_t)g * CP + (cl < C ? cl : 0)) * 4;
    double* const myaccs = accs + ((size_t)g * CP + (cl < C ? cl : 0)) * 4;
    if (pwriter && s == 0) {
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
        if (!LOGIT && cvalid) {
            c0 = P.scale[kl];
            c1 = P.hinv[kl];
            c2 = P.cst[kl];
            c3 = P.s2k[kl];
        }
        mycons[0] = c0;
        mycons[1] = c1;
        mycons[2] = c2;
        mycons[3] = c3;
        myaccs[0] = myaccs[1] = myaccs[2] = myaccs[3] = 0.0;
    }
    const size_t pd = (size_t)P.p * d;
    const int kls = cvalid ? kl : 0;  // safe column for speculative state loads

    constexpr int D = sweep_depth<NPL, C, LOGIT, PCS>();
    const StepPtr steps = (StepPtr)(unsigned long long)P.steps;

    // ---- X staging: zero the ring once, stage columns of steps 0 and 1 -------
    const int npieces = P.ld >> 1;  // 16-byte pieces per column
    for (int e = tid; e < 3 * P.npad; e += nthr) xring[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int pre = 0; pre < 2; ++pre) {
        const double2* src = reinterpret_cast<const double2*>(P.Xp + (size_t)steps[pre].j * P.ld);
        double2* dst = reinterpret_cast<double2*>(xring + pre * P.npad);
        for (int e = tid; e < npieces; e += nthr) dst[e] = src[e];
    }
    // ---- prefetch rings: X pieces of steps 2..D+1, per-response state of steps 0..D-1 ----
    double2 xq[D][PCS];
    double bo_ring[D];
    double s2_ring[LOGIT ? D : 1];
    double xw_ring[LOGIT ? D : 1];
    double lg_ring[LOGIT ? D : 1];   // log(sig2_beta_vb) / 2 when the caller tabulated it
    const bool has_ls2 = LOGIT && P.ls2 != nullptr;   // uniform
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const size_t off = (size_t)steps[u].j * d + kls;
        bo_ring[u] = P.state[2 * pd + off];
        if constexpr (LOGIT) {
            s2_ring[u] = P.s2[off];
            xw_ring[u] = P.xwx[off];
            lg_ring[u] = has_ls2 ? P.ls2[off] : 0.0;
        }
        // slot (u + 2) % D holds the column of step u + 2
        const double2* src =
            reinterpret_cast<const double2*>(P.Xp + (size_t)steps[u + 2].j * P.ld);
#pragma unroll
        for (int q = 0; q < PCS; ++q) {
            const int e = tid + q * nthr;
            xq[(u + 2) % D][q] = (e < npieces) ? src[e] : make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    int slot = 0;  // ring slot of the current step's column (i % 3)
    double nx_lod = steps[0].lod, nx_xtx = steps[0].xtx;
    int nx_j = steps[0].j, nx_jd = steps[D].j, nx_jx = steps[2 + D].j;
#ifdef LOCUS_STAMPS
    unsigned long long tacc[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
    for (int i0 = 0; i0 < P.nsteps_pad; i0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int i = i0 + u;
            const bool act = i < P.nsteps;  // pad steps are no-ops
            // step records are read one step ahead (scalar loads issued here are consumed in
            // step i + 1): an in-order wave would otherwise sit on their latency at every use
            const double st_lod = nx_lod, st_xtx = nx_xtx;
            const int st_j = nx_j, st_jd = nx_jd, st_jx = nx_jx;
            nx_lod = steps[i + 1].lod;
            nx_xtx = steps[i + 1].xtx;
            nx_j = steps[i + 1].j;
            nx_jd = steps[i + 1 + D].j;
            nx_jx = steps[i + 3 + D].j;

            // (2)+(3) partial dot products X_j' R_c over this wave's row slice
            const double2* xs =
                reinterpret_cast<const double2*>(xring + slot * P.npad + row0) + lane;
            // C >= 3 interleaved chains already cover the FMA latency; fewer
            // responses get a second accumulator per response
            constexpr bool kTwoAcc = (C < 3);
            double pd0[C], pd1[kTwoAcc ? C : 1];
#pragma unroll
            for (int c = 0; c < C; ++c) pd0[c] = 0.0;
            if constexpr (kTwoAcc) {
#pragma unroll
                for (int c = 0; c < C; ++c) pd1[c] = 0.0;
            }
#pragma unroll
            for (int t = 0; t < NPL / 2; ++t) {
                const double2 xv = xs[t * 64];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    pd0[c] = fma(xv.x, r[c][2 * t], pd0[c]);
                    if constexpr (kTwoAcc)
                        pd1[c] = fma(xv.y, r[c][2 * t + 1], pd1[c]);
                    else
                        pd0[c] = fma(xv.y, r[c][2 * t + 1], pd0[c]);
                }
            }
            if constexpr (kTwoAcc) {
#pragma unroll
                for (int c = 0; c < C; ++c) pd0[c] += pd1[c];
            }
            LOCUS_STAMP(0);  // LDS read of X + dot products
            double U = reduce_merge<C>(pd0, sub);
            LOCUS_STAMP(1);  // wave reduction

            // (4) cross-wave (row-slice) reduction through LDS; the barrier also
            //     publishes the X column staged at the end of the previous step
            if (S > 1) {
                double* pb = part + (size_t)(i & 1) * P.G * S * CP + (size_t)(g * S) * CP;
                if (pwriter) pb[s * CP + cl] = U;
                __syncthreads();
                if (cl < C) {
                    double tot = 0.0;
                    for (int ss = 0; ss < S; ++ss) tot += pb[ss * CP + cl];
                    U = tot;
                }
            } else {
                __syncthreads();
            }

            LOCUS_STAMP(2);  // barrier (+ cross-wave partials)
            // (5) element-wise update, one response per lane
            const double bo = bo_ring[u];
            double m, a, s2v;
            if constexpr (LOGIT) {
                s2v = s2_ring[u];
                const double dot = fma(xw_ring[u], bo, U);
                m = s2v * dot;                                                // coreLoop.cpp:121
                // :123-125, m^2 / (2 s2) + log(s2) / 2 without the library's division and log:
                // the latency regime pays 8 issue cycles per instruction (sig2_beta_vb is a
                // positive normal number)
                const double hl = has_ls2 ? lg_ring[u] : 0.5 * log_pos_normal(s2v);
                a = st_lod - P.half_lsi - m * m * (0.5 * rcp_pos_normal(s2v)) - hl;
            } else {
                const double2 k01 = *reinterpret_cast<const double2*>(mycons);
                const double2 k23 = *reinterpret_cast<const double2*>(mycons + 2);
                s2v = k23.y;
                const double dot = fma(st_xtx, bo, U);
                m = k01.x * dot;                                      // coreLoop.cpp:42-43
                a = P.cfac * ((st_lod - m * m * k01.y) + k23.x);      // coreLoop.cpp:45-46
            }
            // exp(-log(1 + exp(a))) = 1 / (1 + exp(a)), evaluated without overflow
            const double e = exp_nonpos(-fabs(a));
            const double gm = div_1to2((a >= 0.0) ? e : 1.0, 1.0 + e);
            const double bn = m * gm;                            // coreLoop.cpp:48
            const double delta = (act && cvalid) ? (bo - bn) : 0.0;
            LOCUS_STAMP(3);  // element-wise update
            if (act && writer) {
                double* o = P.state + (size_t)st_j * d + kl;
                o[0] = gm;
                o[pd] = m;
                o[2 * pd] = bn;
            }
            if (act && writer) {  // only this lane ever touches its accumulators
                double2 a01 = *reinterpret_cast<double2*>(myaccs);
                double a2 = myaccs[2];
                a01.x += gm;
                a01.y = fma(gm, fma(m, m, s2v), a01.y);
                a2 = fma(bn, bn, a2);
                *reinterpret_cast<double2*>(myaccs) = a01;
                myaccs[2] = a2;
            }
            // refill the ring slot just consumed with the state of step i + D
            {
                const size_t off = (size_t)st_jd * d + kls;
                bo_ring[u] = P.state[2 * pd + off];
                if constexpr (LOGIT) {
                    s2_ring[u] = P.s2[off];
                    xw_ring[u] = P.xwx[off];
                    if (has_ls2) lg_ring[u] = P.ls2[off];
                }
            }

            LOCUS_STAMP(4);  // stores, statistics, state prefetch
            // (6) rank-1 update of the residual columns: R_c += X_j (b_old - b_new)
            double dc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) dc[c] = bcast_lane(delta, lane_of(c));
#pragma unroll
            for (int t = 0; t < NPL / 2; ++t) {
                const double2 xv = xs[t * 64];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if constexpr (LOGIT) {
                        r[c][2 * t] = fma(xv.x * w[c][2 * t], dc[c], r[c][2 * t]);
                        r[c][2 * t + 1] = fma(xv.y * w[c][2 * t + 1], dc[c], r[c][2 * t + 1]);
                    } else {
                        r[c][2 * t] = fma(xv.x, dc[c], r[c][2 * t]);
                        r[c][2 * t + 1] = fma(xv.y, dc[c], r[c][2 * t + 1]);
                    }
                }
            }

            LOCUS_STAMP(5);  // broadcast + rank-1 update
            // (7) stage column i+2 (loaded D steps ago) into the LDS slot last read at step
            //     i-1, then reuse its registers for the column of step i + 2 + D
            {
                int wslot = slot + 2;
                if (wslot >= 3) wslot -= 3;
                double2* dst = reinterpret_cast<double2*>(xring + wslot * P.npad);
                const double2* src =
                    reinterpret_cast<const double2*>(P.Xp + (size_t)st_jx * P.ld);
#pragma unroll
                for (int q = 0; q < PCS; ++q) {
                    const int e = tid + q * nthr;
                    if (e < npieces) dst[e] = xq[(u + 2) % D][q];
                    xq[(u + 2) % D][q] = (e < npieces) ? src[e] : make_double2(0.0, 0.0);
                }
            }
            slot = (slot == 2) ? 0 : slot + 1;
            LOCUS_STAMP(6);  // X staging (ds_write + next global loads)
            __builtin_amdgcn_sched_barrier(0);  // keep the unrolled steps apart (VGPR budget)
        }
    }

#ifdef LOCUS_STAMPS
    if (tid == 0 && blockIdx.x < 4096) {
        for (int k = 0; k < kStampSlots; ++k) g_sweep_stamps[blockIdx.x * kStampSlots + k] = tacc[k];
    }
#endif
    // ---- epilogue: residual back to memory, fused statistics -------------------
    // The lane id and the bounds are laundered through empty asm so that the
    // compiler does not keep the C * NPL addresses and masks of the prologue
    // loads alive in registers across the whole sweep.
    int lane_e = lane, n_e = n, d_e = d;
    asm volatile("" : "+v"(lane_e));
    asm volatile("" : "+s"(n_e), "+s"(d_e));
    double ss[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = k0 + c;
        double* Rk = P.R + (size_t)k * n_e;
        double a0 = 0.0;
#pragma unroll
        for (int t = 0; t < NPL / 2; ++t) {
            const int row = row0 + t * 128 + 2 * lane_e;
            if ((k < d_e) && (row < n_e)) Rk[row] = r[c][2 * t];
            if ((k < d_e) && (row + 1 < n_e)) Rk[row + 1] = r[c][2 * t + 1];
            a0 = fma(r[c][2 * t], r[c][2 * t], a0);
            a0 = fma(r[c][2 * t + 1], r[c][2 * t + 1], a0);
        }
        ss[c] = a0;
    }
    if (P.st != nullptr) {
        double U = reduce_merge<C>(ss, sub);
        __syncthreads();  // all waves are past their last read of `part`
        if (S > 1) {
            double* pb = part + (size_t)(g * S) * CP;
            if (pwriter) pb[s * CP + cl] = U;
            __syncthreads();
            if (cl < C) {
                double tot = 0.0;
                for (int s2i = 0; s2i < S; ++s2i) tot += pb[s2i * CP + cl];
                U = tot;
            }
        }
        if (writer) {
            P.st[kl] = myaccs[0];
            P.st[(size_t)d + kl] = myaccs[1];
            P.st[2 * (size_t)d + kl] = myaccs[2];
            P.st[3 * (size_t)d + kl] = U;
        }
    }
}

}  // namespace locus
