// CAVI sweep on the fp64 matrix cores of gfx950 -- blocked coordinate ascent.
//
// Same update as sweep_kernel.h (src/coreLoop.cpp:36-52, coreStructLoop.cpp:34-50,
// coreZLoop / coreProbitLoop), restated so that the O(n) work per (covariate, response)
// pair runs as v_mfma_f64_16x16x4_f64 instead of vector FMAs + cross-lane reductions
// (measured on MI355X: MFMA fp64 sustains 77.6 TFLOP/s, v_fma_f64 about 60).
//
// The sweep order is cut into blocks of B = 16 consecutive covariates j_0..j_15.  With R the
// residual at the start of a block and d_s = beta_old - beta_new of step s,
//     X_t' R^(t) = X_t' (R + sum_{s<t} X_s d_s) = G_t + sum_{s<t} A_ts d_s,
//     G = X_J' R  (16 x responses),     A = X_J' X_J  (16 x 16, the block's Gram matrix),
// and after the block R += X_J d.  G and the rank-16 update are dense contractions over the
// n rows (the matrix cores); the 16 sequential element-wise updates of a block touch only
// O(16) numbers per response and run with one response per lane.  The algebra is the same
// folding the reference uses in src/coreBatch.cpp:32-33; only the summation order differs.
//
// Work decomposition: a workgroup of S waves owns 16*T responses (T tiles) for the whole
// sweep; wave s owns rows [s*16*KW4, (s+1)*16*KW4) of those columns IN REGISTERS, laid out as
// the B operand of the dot MFMAs == the C/D operand of the update MFMAs:
//     lane l, tile tau, k-step g (row 4g + (l>>4), response 16 tau + (l&15))  ->  rv[tau][g>>2][g&3].
// X arrives from global memory already in MFMA operand order (two block-packed copies built
// once per sweep by pack_blocks_kernel: XBd for G, XBu for the update), one coalesced 512-byte
// load per MFMA operand, prefetched through a register ring; X never touches LDS.
// Per block: dot MFMAs -> partial G to LDS -> ONE barrier -> every wave runs the 16-step
// chain redundantly (lane = response) -> deltas through LDS -> update MFMAs.
#pragma once
#include <utility>

#include "common.h"
#include "fastmath.h"

namespace locus {

typedef double d4 __attribute__((ext_vector_type(4)));

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>).  Register arrays
// indexed inside must see true constants (a `#pragma unroll` loop whose trip count depends on
// an outer unrolled index was compiled to movrel loops).
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f)
{
    if constexpr (N > 0) sfor_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kBlk = 16;  // covariates per block (M of the MFMA)

struct MfmaSweepParams {
    int n, p, d;
    int nblk;            // blocks of 16 steps (last one padded with inactive steps)
    int nsteps;          // active steps
    int rows_pad;        // 16 * KW4 * S  (>= n)
    const StepRec* steps;  // [16 * nblk + 16]
    const int* err;
    const double* XB;    // [nblk][S][KW][64][2]: per block and wave, the KW dot operands
                         // (A of G = X_J' R) then the KW update operands (A of R += X_J delta),
                         // consecutive operands paired per lane (one 16-byte load = 2 operands)
    const double* gram;  // [nblk][16][16]              X_J'X_J
    const int* jidx;     // [16 * nblk + 16]            covariate index of every step
    const double* scale; // per-response constants, as SweepParams
    const double* hinv;
    const double* cst;
    const double* s2k;
    double cfac;
    double* R;           // n x d column-major
    double* state;       // [3][p][d]
    double* st;          // [4][d] or nullptr
};

typedef const StepRec __attribute__((address_space(4))) * StepPtrM;

#ifndef LOCUS_RING
#define LOCUS_RING 8
#endif
constexpr int kRing = LOCUS_RING;  // X operands in flight per wave

// Diagnostic build only (-DLOCUS_STAMPS, `make diag`): per-phase cycle sums of wave 0 of every
// workgroup.  No stamp executes in the product build.
#ifdef LOCUS_STAMPS
__device__ unsigned long long g_mfma_stamps[8 * 4096];
#define MSTAMP(k)                                                                        \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        tacc[k] += t_ - tprev;                                                           \
        tprev = t_;                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif

// One update MFMA slot: row tile `i` of k-slice `kk` (covariates 4 kk .. 4 kk + 3 of the block),
// operand number KW + kk * KW4 + i of the wave's stream for this block.
#define LOCUS_UPD_SLOT(kk, i)                                                                  \
    do {                                                                                       \
        constexpr int o_ = KW + (kk) * KW4 + (i);      /* operand index in the block stream */  \
        constexpr int sl_ = (o_ % NR) >> 1;                                                    \
        const double a_ = (o_ & 1) ? ring[sl_].y : ring[sl_].x;                                \
        if constexpr ((o_ & 1) != 0) {                                                         \
            if constexpr (o_ + NR - 1 == 2 * KW) xp += blk_skip;                               \
            if (o_ + NR - 1 < 2 * KW || more) {                                                \
                ring[sl_] = *xp;                                                               \
                xp += 64;                                                                      \
            }                                                                                  \
        }                                                                                      \
        sfor<T>([&](auto tc_) {                                                                \
            constexpr int tau_ = decltype(tc_)::value;                                         \
            rv[tau_][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, bop[tau_], rv[tau_][i], 0,  \
                                                               0, 0);                          \
        });                                                                                    \
    } while (0)

template <int T, int KW4, int S>
__global__ void __launch_bounds__(64 * S)
mfma_sweep_kernel(const MfmaSweepParams P)
{
    static_assert(T >= 1 && T <= 2, "1 or 2 response tiles per workgroup");
    static_assert(S == 4 || S == 8, "4 or 8 waves per workgroup");
    constexpr int KW = 4 * KW4;          // k-steps (4 rows each) per wave
    constexpr int NR = (KW < kRing) ? KW : kRing;  // operands in flight (even)
    constexpr int NR2 = NR / 2;
    constexpr int NBW = 16 / S;          // beta_old loads per wave and block
    static_assert((2 * KW) % NR == 0 && NR % 2 == 0, "ring must divide the block stream");

    extern __shared__ __align__(16) double smem[];
    double* const part = smem;                         // [2][S][T][4][64]  partial G
    double* const bol = part + 2 * S * T * 256;        // [2][16][64]       beta_old of the block
    double* const gram = bol + 2 * 16 * 64;            // [2][256]          X_J'X_J
    double* const scl = gram + 2 * 256;                // [2][16][2]        lod, xtx of the steps
    double* const dl = scl + 2 * 32;                   // [16][64]          deltas of the block

    if (*P.err != 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int s = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, kept scalar
    const int q = lane >> 4, r = lane & 15;
    const int n = P.n, d = P.d;
    const int k0 = blockIdx.x * 16 * T;
    const int row0 = s * 16 * KW4;
    const size_t pd = (size_t)P.p * d;
    typedef const int __attribute__((address_space(4))) * JPtr;
    const JPtr jidx = (JPtr)(unsigned long long)P.jidx;

    // chain role of this lane: response k0 + 16 q + r (q < T); every wave runs the whole
    // chain, wave s stores the results of the steps t = s (mod S)
    const int qc = (q < T) ? q : 0;
    const int kc = k0 + 16 * qc + r;
    const bool cvalid = (q < T) && (kc < d);
    const int kcs = cvalid ? kc : 0;

    // ---- residual into registers: B operand of the dot MFMAs, C/D of the update MFMAs --
    d4 rv[T][KW4];
#pragma unroll
    for (int tau = 0; tau < T; ++tau) {
        const int k = k0 + 16 * tau + r;
        const double* Rk = P.R + (size_t)(k < d ? k : 0) * n;
#pragma unroll
        for (int g = 0; g < KW; ++g) {
            const int row = row0 + 4 * g + q;
            rv[tau][g >> 2][g & 3] = (k < d && row < n) ? Rk[row] : 0.0;
        }
    }

    double c_scale = 0.0, c_hinv = 0.0, c_cst = 0.0, c_s2 = 0.0;
    if (cvalid) {
        c_scale = P.scale[kc];
        c_hinv = P.hinv[kc];
        c_cst = P.cst[kc];
        c_s2 = P.s2k[kc];
    }
    const double cfac = P.cfac;
    double acc_g = 0.0, acc_m2 = 0.0, acc_b2 = 0.0;

    // This wave's X operands form ONE stream in consumption order (KW dot operands, then
    // the KW update operands k-slice by k-slice), two operands per 16-byte load.
    const double2* xp = reinterpret_cast<const double2*>(P.XB) + ((size_t)s * KW) * 64 + lane;
    constexpr size_t blk_skip = (size_t)(S - 1) * KW * 64;  // other waves' share, in pairs

    // ---- block 0: beta_old, Gram and step scalars straight to LDS (parity 0) ---------
    {
        int j0[NBW];
#pragma unroll
        for (int u = 0; u < NBW; ++u) j0[u] = jidx[s + u * S];
#pragma unroll
        for (int u = 0; u < NBW; ++u)
            bol[(s + u * S) * 64 + lane] = P.state[2 * pd + (size_t)j0[u] * d + kcs];
    }
    if (tid < 256) gram[tid] = P.gram[tid];
    if (tid < 16) {
        scl[2 * tid] = P.steps[tid].lod;
        scl[2 * tid + 1] = P.steps[tid].xtx;
    }

    double2 ring[NR2];
#pragma unroll
    for (int i = 0; i < NR2; ++i) {
        ring[i] = *xp;
        xp += 64;
    }

#ifdef LOCUS_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
    for (int b = 0; b < P.nblk; ++b) {
        const int par = b & 1;
        const bool more = (b + 1 < P.nblk);

        // prefetch for block b + 1 (rows past the last block exist: pack pads jidx / steps)
        int jn[NBW];
#pragma unroll
        for (int u = 0; u < NBW; ++u) jn[u] = jidx[16 * (b + 1) + s + u * S];
        double bo_nx[NBW];
#pragma unroll
        for (int u = 0; u < NBW; ++u) bo_nx[u] = P.state[2 * pd + (size_t)jn[u] * d + kcs];
        const double gr_nx = (tid < 256) ? P.gram[(size_t)(b + 1) * 256 + tid] : 0.0;
        double lod_nx = 0.0, xtx_nx = 0.0;
        if (tid < 16) {
            lod_nx = P.steps[16 * (b + 1) + tid].lod;
            xtx_nx = P.steps[16 * (b + 1) + tid].xtx;
        }
        MSTAMP(6);  // issue of next block's operand loads

        // ---- (1) G partial = X_J' R over this wave's rows ---------------------------
        d4 acc[T];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) acc[tau] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KW; ++g) {
            const double a = (g & 1) ? ring[(g % NR) >> 1].y : ring[(g % NR) >> 1].x;
            if (g & 1) {  // both operands of the pair consumed: request pair (g + NR) / 2
                ring[(g % NR) >> 1] = *xp;
                xp += 64;
            }
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
                acc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rv[tau][g >> 2][g & 3], acc[tau],
                                                                0, 0, 0);
        }
        MSTAMP(0);  // dot MFMAs

        // ---- (2) exchange partial G; one barrier per block ----------------------------
        {
            double* pb = part + ((par * S + s) * T) * 256 + lane;
#pragma unroll
            for (int tau = 0; tau < T; ++tau) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pb[tau * 256 + i * 64] = acc[tau][i];
            }
        }
        MSTAMP(1);  // partial G to LDS
        __syncthreads();
        MSTAMP(2);  // barrier
        // next block's beta_old / Gram / scalars: every wave is past its reads of that parity
#pragma unroll
        for (int u = 0; u < NBW; ++u) bol[((par ^ 1) * 16 + s + u * S) * 64 + lane] = bo_nx[u];
        if (tid < 256) gram[(par ^ 1) * 256 + tid] = gr_nx;
        if (tid < 16) {
            scl[(par ^ 1) * 32 + 2 * tid] = lod_nx;
            scl[(par ^ 1) * 32 + 2 * tid + 1] = xtx_nx;
        }
        MSTAMP(3);  // next block's operands to LDS

        // ---- (3) the 16 sequential updates of the block, one response per lane, with the
        //          rank-16 update of the residual riding in their shadow ------------------
        // The operands of step t + 1 (beta_old, lod, xtx, partial G) are requested at the top
        // of step t so that their LDS latency hides behind the exp / division chain.
        // Once steps 4 kk .. 4 kk + 3 are done, the deltas of k-slice kk are known and its
        // KW4 * T update MFMAs are issued between the instructions of steps 4 kk + 4 ..
        // 4 kk + 7 (different pipe, independent registers); only the last k-slice runs
        // after the chain.  The scheduling barrier at the end of each step bounds how far
        // loads are hoisted (register pressure).
        __builtin_amdgcn_sched_barrier(0);
        double corr[kBlk];  // sum_{s<t} A_ts delta_s
#pragma unroll
        for (int t = 0; t < kBlk; ++t) corr[t] = 0.0;
        const double* pr = part + ((par * S) * T + qc) * 256 + r;
        const double* gm_l = gram + par * 256;
        const double* sc_l = scl + par * 32;
        const double* bo_l = bol + (par * 16) * 64 + lane;
        double bo_c = bo_l[0];
        double2 sc_c = *reinterpret_cast<const double2*>(sc_l);
        double gp_c[S];
#pragma unroll
        for (int ss = 0; ss < S; ++ss) gp_c[ss] = pr[ss * T * 256];
        int jc[kBlk];
#pragma unroll
        for (int t = 0; t < kBlk; ++t) jc[t] = jidx[16 * b + t];
        double bop[T];
        __builtin_amdgcn_sched_barrier(0);
        sfor<kBlk>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const bool act = 16 * b + t < P.nsteps;
            // requests for step t + 1 and for the end of this step
            double bo_n = 0.0;
            double2 sc_n = make_double2(0.0, 0.0);
            double gp_n[S];
            sfor<S>([&](auto sc) { gp_n[decltype(sc)::value] = 0.0; });
            if constexpr (t + 1 < kBlk) {
                bo_n = bo_l[(t + 1) * 64];
                sc_n = *reinterpret_cast<const double2*>(sc_l + 2 * (t + 1));
                sfor<S>([&](auto sc) {
                    constexpr int ss = decltype(sc)::value;
                    gp_n[ss] = pr[ss * T * 256 + ((t + 1) >> 2) * 64 + 16 * ((t + 1) & 3)];
                });
            }
            double gcol[kBlk];
            sfor<kBlk - 1 - t>([&](auto c2) {
                constexpr int t2 = t + 1 + decltype(c2)::value;
                gcol[t2] = gm_l[t2 * 16 + t];
            });
            // B operands of the k-slice completed four steps ago
            if constexpr (t >= 4 && (t & 3) == 0) {
                sfor<T>([&](auto tc2) {
                    constexpr int tau = decltype(tc2)::value;
                    bop[tau] = dl[(4 * (t / 4 - 1) + q) * 64 + 16 * tau + r];
                });
            }

            double u = gp_c[0];
            sfor<S - 1>([&](auto sc) { u += gp_c[decltype(sc)::value + 1]; });
            u += corr[t];
            const double dot = fma(sc_c.y, bo_c, u);
            const double m = c_scale * dot;                              // coreLoop.cpp:42-43
            const double a = cfac * ((sc_c.x - m * m * c_hinv) + c_cst);  // :45-46
            const double e = exp_nonpos(-fabs(a));
            const double gmv = div_1to2((a >= 0.0) ? e : 1.0, 1.0 + e);
            const double bn = m * gmv;                                   // :48
            const double delta = (act && cvalid) ? (bo_c - bn) : 0.0;
            sfor<kBlk - 1 - t>([&](auto c2) {
                constexpr int t2 = t + 1 + decltype(c2)::value;
                corr[t2] = fma(gcol[t2], delta, corr[t2]);
            });
            dl[t * 64 + lane] = delta;
            if (act && cvalid && (t % S) == s) {
                double* o = P.state + (size_t)jc[t] * d + kc;
                o[0] = gmv;
                o[pd] = m;
                o[2 * pd] = bn;
                acc_g += gmv;
                acc_m2 = fma(gmv, fma(m, m, c_s2), acc_m2);
                acc_b2 = fma(bn, bn, acc_b2);
            }
            if constexpr (t >= 4) {
                // row tiles of k-slice kk handled in this step: i with 4 i / KW4 == t & 3
                constexpr int kk = t / 4 - 1;
                sfor<KW4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((4 * i) / KW4 == (t & 3)) LOCUS_UPD_SLOT(kk, i);
                });
            }
            bo_c = bo_n;
            sc_c = sc_n;
            sfor<S>([&](auto sc) { gp_c[decltype(sc)::value] = gp_n[decltype(sc)::value]; });
            __builtin_amdgcn_sched_barrier(0);
        });
        MSTAMP(4);  // chain (+ 3/4 of the update MFMAs)

        // ---- (4) last k-slice of R += X_J delta ------------------------------------------
#pragma unroll
        for (int tau = 0; tau < T; ++tau) bop[tau] = dl[(12 + q) * 64 + 16 * tau + r];
        sfor<KW4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            LOCUS_UPD_SLOT(3, i);
        });
        MSTAMP(5);  // update MFMAs
    }

#ifdef LOCUS_STAMPS
    if (tid == 0 && blockIdx.x < 4096)
        for (int k = 0; k < 8; ++k) g_mfma_stamps[blockIdx.x * 8 + k] = tacc[k];
#endif
    // ---- epilogue: residual back to memory, fused statistics ---------------------------
    double ss2[T];
#pragma unroll
    for (int tau = 0; tau < T; ++tau) {
        const int k = k0 + 16 * tau + r;
        double* Rk = P.R + (size_t)(k < d ? k : 0) * n;
        double a0 = 0.0;
#pragma unroll
        for (int g = 0; g < KW; ++g) {
            const int row = row0 + 4 * g + q;
            const double v = rv[tau][g >> 2][g & 3];
            if (k < d && row < n) Rk[row] = v;
            a0 = fma(v, v, a0);
        }
        ss2[tau] = a0;
    }
    if (P.st != nullptr) {
        __syncthreads();  // all waves are done with `part`
        // part[s][tau][lane] <- partial sum of squares; the column statistics were accumulated
        // by the wave that stored the step
#pragma unroll
        for (int tau = 0; tau < T; ++tau) part[(s * T + tau) * 64 + lane] = ss2[tau];
        double* sacc = part + S * T * 64;  // [S][3][64]
        sacc[(s * 3 + 0) * 64 + lane] = acc_g;
        sacc[(s * 3 + 1) * 64 + lane] = acc_m2;
        sacc[(s * 3 + 2) * 64 + lane] = acc_b2;
        __syncthreads();
        if (cvalid && s == 0) {
            double tot = 0.0, sg = 0.0, sm = 0.0, sb = 0.0;
#pragma unroll
            for (int ss = 0; ss < S; ++ss) {
                const double* pp = part + (ss * T + qc) * 64 + r;
                tot += (pp[0] + pp[16]) + (pp[32] + pp[48]);
                sg += sacc[(ss * 3 + 0) * 64 + lane];
                sm += sacc[(ss * 3 + 1) * 64 + lane];
                sb += sacc[(ss * 3 + 2) * 64 + lane];
            }
            P.st[kc] = sg;
            P.st[(size_t)d + kc] = sm;
            P.st[2 * (size_t)d + kc] = sb;
            P.st[3 * (size_t)d + kc] = tot;
        }
    }
}
#undef LOCUS_UPD_SLOT

}  // namespace locus
