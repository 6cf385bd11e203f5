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
#include "common.h"

namespace locus {

typedef double d4 __attribute__((ext_vector_type(4)));

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

// exp(x) for x <= 0, to < 1 ulp of truncation error and a few ulp of rounding: k = rint(x log2 e),
// r = x - k ln 2 (two-term Cody-Waite), degree-12 Taylor polynomial in Horner form, ldexp.
// 20 instructions with no branches and no table; at one wave per SIMD every fp64 VALU
// instruction costs 8 issue cycles, so the instruction COUNT of the chain is what matters.
__device__ __forceinline__ double exp_nonpos(double x)
{
    x = fmax(x, -800.0);  // exp underflows to 0 below -745.2; keeps k in int range
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;               // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);           // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);           // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);           // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);           // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);           // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);           // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);           // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);           // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// num / den for den in [1, 2], num in [0, 1]: v_rcp_f64 + two Newton steps + one residual
// correction (no scaling / fix-up needed in this range).
__device__ __forceinline__ double div_1to2(double num, double den)
{
    double rcp = __builtin_amdgcn_rcp(den);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    rcp = fma(fma(-den, rcp, 1.0), rcp, rcp);
    const double qv = num * rcp;
    return fma(fma(-den, qv, num), rcp, qv);
}

// Register homes.  In a kernel that uses AGPRs the compiler selects the MFMA form whose
// accumulator (C/D) operands are AGPRs; A/B operands may be VGPRs or AGPRs.  A wave can hold
// 256 AGPRs = two tiles of 16 responses x 256 rows as accumulators.  The third tile (T = 3)
// therefore lives in VGPRs as plain doubles: it is a B operand in the dot phase as it is,
// and its rank-16 update is accumulated into a zeroed AGPR tile and added with vector adds
// (1/16 of the flops of its MFMAs).
template <int T, int KW4, int S>
__global__ void __launch_bounds__(64 * S)
mfma_sweep_kernel(const MfmaSweepParams P)
{
    static_assert(T >= 1 && T <= 3, "1..3 response tiles per workgroup");
    static_assert(S == 4 || S == 8, "4 or 8 waves per workgroup");
    constexpr int KW = 4 * KW4;          // k-steps (4 rows each) per wave
    constexpr int NR = (KW < kRing) ? KW : kRing;  // operands in flight (even)
    constexpr int NR2 = NR / 2;
    constexpr int NBW = 16 / S;          // beta_old loads per wave and block

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
    const StepPtrM steps = (StepPtrM)(unsigned long long)P.steps;

    // chain role of this lane: response k0 + 16 q + r (q < T)
    const int qc = (q < T) ? q : 0;
    const int kc = k0 + 16 * qc + r;
    const bool cvalid = (q < T) && (kc < d);
    const int kcs = cvalid ? kc : 0;
    // every wave runs the whole chain; wave s writes the results of the steps t = s (mod S)
    const bool writer = cvalid;

    // ---- residual into registers (B-operand order) ---------------------------------
    constexpr int TA = (T < 2) ? T : 2;   // tiles held as MFMA accumulators
    d4 rv[TA][KW4];
    double rw[(T > 2) ? KW : 1];          // third tile, plain registers
#pragma unroll
    for (int tau = 0; tau < T; ++tau) {
        const int k = k0 + 16 * tau + r;
        const double* Rk = P.R + (size_t)(k < d ? k : 0) * n;
#pragma unroll
        for (int g = 0; g < KW; ++g) {
            const int row = row0 + 4 * g + q;
            const double v = (k < d && row < n) ? Rk[row] : 0.0;
            if (tau < 2)
                rv[tau < 2 ? tau : 0][g >> 2][g & 3] = v;
            else
                rw[(T > 2) ? g : 0] = v;
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

    // This wave's X operands form ONE stream in consumption order: 2 KW operands of 64
    // doubles per block, then a jump over the other waves' share.  A single running
    // pointer keeps the address arithmetic out of the SGPR file.
    const double2* xp = reinterpret_cast<const double2*>(P.XB) + ((size_t)s * KW) * 64 + lane;
    constexpr size_t blk_skip = (size_t)(S - 1) * KW * 64;  // in operand pairs (double2)

    // ---- block 0: beta_old, Gram and step scalars straight to LDS (parity 0) ---------
#pragma unroll
    for (int u = 0; u < NBW; ++u) {
        const int t = s + u * S;
        bol[t * 64 + lane] = P.state[2 * pd + (size_t)steps[t].j * d + kcs];
    }
    if (tid < 256) gram[tid] = P.gram[tid];
    if (tid < 16) {
        scl[2 * tid] = P.steps[tid].lod;
        scl[2 * tid + 1] = P.steps[tid].xtx;
    }

    // X ring: first operands of block 0's dot phase
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

        // prefetch for block b + 1: beta_old (this wave's share), Gram matrix, step scalars
        double bo_nx[NBW];
        double gr_nx = 0.0, lod_nx = 0.0, xtx_nx = 0.0;
#pragma unroll
        for (int u = 0; u < NBW; ++u) {
            const int t = s + u * S;
            bo_nx[u] = 0.0;
            if (more) bo_nx[u] = P.state[2 * pd + (size_t)steps[16 * (b + 1) + t].j * d + kcs];
        }
        if (tid < 256 && more) gr_nx = P.gram[(size_t)(b + 1) * 256 + tid];
        if (tid < 16 && more) {
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
            for (int tau = 0; tau < T; ++tau) {
                const double bv = (tau < 2) ? rv[tau < 2 ? tau : 0][g >> 2][g & 3]
                                            : rw[(T > 2) ? g : 0];
                acc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[tau], 0, 0, 0);
            }
        }

        MSTAMP(0);  // prefetch issue + dot MFMAs
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
        if (more) {
#pragma unroll
            for (int u = 0; u < NBW; ++u) bol[((par ^ 1) * 16 + s + u * S) * 64 + lane] = bo_nx[u];
            if (tid < 256) gram[(par ^ 1) * 256 + tid] = gr_nx;
            if (tid < 16) {
                scl[(par ^ 1) * 32 + 2 * tid] = lod_nx;
                scl[(par ^ 1) * 32 + 2 * tid + 1] = xtx_nx;
            }
        }

        // ---- (3) the 16 sequential updates of the block, one response per lane --------
        // Software pipelined by hand: the operands of step t + 1 (beta_old, lod, xtx, Gram
        // column, covariate index) are requested at the top of step t, so their LDS / scalar
        // cache latency hides behind the exp / division chain of step t.  The scheduling
        // barrier at the end of each step keeps later steps' loads from being hoisted
        // further (register pressure).
        MSTAMP(3);  // next block's operands to LDS
        __builtin_amdgcn_sched_barrier(0);
        // G_t arrives as S partial sums in LDS; corr[t] collects sum_{s<t} A_ts delta_s
        double corr[kBlk];
#pragma unroll
        for (int t = 0; t < kBlk; ++t) corr[t] = 0.0;
        const double* pr = part + ((par * S) * T + qc) * 256 + r;
        const double* gm_l = gram + par * 256;
        const double* sc_l = scl + par * 32;
        const double* bo_l = bol + (par * 16) * 64 + lane;
        double bo_c = bo_l[0];
        double2 sc_c = *reinterpret_cast<const double2*>(sc_l);
        int j_c = steps[16 * b].j;
        double gp_c[S];
#pragma unroll
        for (int ss = 0; ss < S; ++ss) gp_c[ss] = pr[ss * T * 256];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kBlk; ++t) {
            const int i = 16 * b + t;
            const bool act = i < P.nsteps;
            // requests for step t + 1 and for the end of this step
            double bo_n = 0.0;
            double2 sc_n = make_double2(0.0, 0.0);
            int j_n = 0;
            double gp_n[S];
#pragma unroll
            for (int ss = 0; ss < S; ++ss) gp_n[ss] = 0.0;
            if (t + 1 < kBlk) {
                bo_n = bo_l[(t + 1) * 64];
                sc_n = *reinterpret_cast<const double2*>(sc_l + 2 * (t + 1));
                j_n = steps[i + 1].j;
#pragma unroll
                for (int ss = 0; ss < S; ++ss)
                    gp_n[ss] = pr[ss * T * 256 + ((t + 1) >> 2) * 64 + 16 * ((t + 1) & 3)];
            }
            double gcol[kBlk];
#pragma unroll
            for (int t2 = t + 1; t2 < kBlk; ++t2) gcol[t2] = gm_l[t2 * 16 + t];

            double u = gp_c[0];
#pragma unroll
            for (int ss = 1; ss < S; ++ss) u += gp_c[ss];
            u += corr[t];
            const double dot = fma(sc_c.y, bo_c, u);
            const double m = c_scale * dot;                              // coreLoop.cpp:42-43
            const double a = cfac * ((sc_c.x - m * m * c_hinv) + c_cst);  // :45-46
            const double e = exp_nonpos(-fabs(a));
            const double gmv = div_1to2((a >= 0.0) ? e : 1.0, 1.0 + e);
            const double bn = m * gmv;                                   // :48
            const double delta = (act && cvalid) ? (bo_c - bn) : 0.0;
#pragma unroll
            for (int t2 = t + 1; t2 < kBlk; ++t2) corr[t2] = fma(gcol[t2], delta, corr[t2]);
            dl[t * 64 + lane] = delta;
            if (act && writer && (t % S) == s) {
                double* o = P.state + (size_t)j_c * d + kc;
                o[0] = gmv;
                o[pd] = m;
                o[2 * pd] = bn;
                acc_g += gmv;
                acc_m2 = fma(gmv, fma(m, m, c_s2), acc_m2);
                acc_b2 = fma(bn, bn, acc_b2);
            }
            bo_c = bo_n;
            sc_c = sc_n;
            j_c = j_n;
#pragma unroll
            for (int ss = 0; ss < S; ++ss) gp_c[ss] = gp_n[ss];
            __builtin_amdgcn_sched_barrier(0);
        }

        MSTAMP(4);  // chain
        // ---- (4) R += X_J delta on the matrix cores ------------------------------------
        double bop[T][4];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bop[tau][kk] = dl[(4 * kk + q) * 64 + 16 * tau + r];
        }
        d4 upd = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KW; ++g) {  // g = 4 * tile + kk
            const int slot = ((g + KW) % NR) >> 1;
            const double a = (g & 1) ? ring[slot].y : ring[slot].x;
            if (g & 1) {
                if (g + NR - 1 == KW) xp += blk_skip;  // the stream continues in the next block
                if (g + NR - 1 < KW || more) {
                    ring[slot] = *xp;
                    xp += 64;
                }
            }
#pragma unroll
            for (int tau = 0; tau < TA; ++tau)
                rv[tau][g >> 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop[tau][g & 3],
                                                                       rv[tau][g >> 2], 0, 0, 0);
            if constexpr (T > 2) {
                if ((g & 3) == 0) upd = d4{0.0, 0.0, 0.0, 0.0};
                upd = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop[2][g & 3], upd, 0, 0, 0);
                if ((g & 3) == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rw[g - 3 + e] += upd[e];
                }
            }
        }
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
            const double v = (tau < 2) ? rv[tau < 2 ? tau : 0][g >> 2][g & 3] : rw[(T > 2) ? g : 0];
            if (k < d && row < n) Rk[row] = v;
            a0 = fma(v, v, a0);
        }
        ss2[tau] = a0;
    }
    if (P.st != nullptr) {
        __syncthreads();  // all waves are done with `part`
        // part[s][tau][lane] <- partial sum of squares; lane (q, r) then adds S * 4 values
#pragma unroll
        for (int tau = 0; tau < T; ++tau) part[(s * T + tau) * 64 + lane] = ss2[tau];
        // the column statistics were accumulated by the wave that stored the step
        double* sacc = part + S * T * 64;  // [S][3][64]
        sacc[(s * 3 + 0) * 64 + lane] = acc_g;
        sacc[(s * 3 + 1) * 64 + lane] = acc_m2;
        sacc[(s * 3 + 2) * 64 + lane] = acc_b2;
        __syncthreads();
        if (writer && s == 0) {
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

}  // namespace locus
