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
    const double* XB;    // [nblk][S][2][KW][64]: per block and wave, the KW dot operands
                         // (A of G = X_J' R) then the KW update operands (A of R += X_J delta)
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

template <int T, int KW4>
constexpr int mfma_max_threads()
{
    return (T * KW4 * 8 > 128) ? 256 : 512;
}

constexpr int kRing = 8;  // X operands in flight per wave

template <int T, int KW4>
__global__ void __launch_bounds__((mfma_max_threads<T, KW4>()))
mfma_sweep_kernel(const MfmaSweepParams P)
{
    static_assert(T >= 1 && T <= 3, "1..3 response tiles per workgroup");
    constexpr int KW = 4 * KW4;          // k-steps (4 rows each) per wave
    constexpr int NR = (KW < kRing) ? KW : kRing;

    extern __shared__ __align__(16) double smem[];
    const int S = blockDim.x >> 6;
    double* const part = smem;                                  // [2][S][T][4][64]
    double* const bol = part + 2 * (size_t)S * T * 256;         // [2][16][64]
    double* const gram = bol + 2 * 16 * 64;                     // [2][256]
    double* const dl = gram + 2 * 256;                          // [16][64]

    if (*P.err != 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int s = tid >> 6;
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
    const bool writer = cvalid && (s == 0);

    // ---- residual into registers (B-operand order) ---------------------------------
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
    double acc_g = 0.0, acc_m2 = 0.0, acc_b2 = 0.0;

    // This wave's X operands form ONE stream in consumption order: 2 KW operands of 64
    // doubles per block, then a jump over the other waves' share.  A single running
    // pointer keeps the address arithmetic out of the SGPR file.
    const double* xp = P.XB + ((size_t)s * 2 * KW) * 64 + lane;
    const size_t blk_skip = (size_t)(S - 1) * 2 * KW * 64;
    const int nb_per_wave = 16 / S;                        // beta_old loads per wave per block (S >= 4)

    // ---- block 0: beta_old and Gram straight to LDS (parity 0) -----------------------
    for (int u = 0; u < nb_per_wave; ++u) {
        const int t = s + u * S;
        bol[t * 64 + lane] = P.state[2 * pd + (size_t)steps[t].j * d + kcs];
    }
    for (int e = tid; e < 256; e += blockDim.x) gram[e] = P.gram[e];

    // X ring: first operands of block 0's dot phase
    double ring[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        ring[i] = *xp;
        xp += 64;
    }

    for (int b = 0; b < P.nblk; ++b) {
        const int par = b & 1;
        const bool more = (b + 1 < P.nblk);

        // prefetch for block b + 1: beta_old (this wave's share) and the Gram matrix
        double bo_nx[4];
        double gr_nx = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = s + u * S;
            bo_nx[u] = 0.0;
            if (u < nb_per_wave && more)
                bo_nx[u] = P.state[2 * pd + (size_t)steps[16 * (b + 1) + t].j * d + kcs];
        }
        if (tid < 256 && more) gr_nx = P.gram[(size_t)(b + 1) * 256 + tid];

        // ---- (1) G partial = X_J' R over this wave's rows ---------------------------
        d4 acc[T];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) acc[tau] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KW; ++g) {
            const double a = ring[g % NR];
            ring[g % NR] = *xp;  // operand g + NR of this block's stream
            xp += 64;
            if ((g & 7) == 7) asm volatile("" : "+v"(xp));
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
                acc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rv[tau][g >> 2][g & 3], acc[tau],
                                                                0, 0, 0);
        }

        // ---- (2) exchange partial G; one barrier per block ----------------------------
        {
            double* pb = part + ((size_t)(par * S + s) * T) * 256 + lane;
#pragma unroll
            for (int tau = 0; tau < T; ++tau) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pb[tau * 256 + i * 64] = acc[tau][i];
            }
        }
        __syncthreads();
        // next block's beta_old / Gram: every wave is past its reads of that parity
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = s + u * S;
                if (u < nb_per_wave) bol[((par ^ 1) * 16 + t) * 64 + lane] = bo_nx[u];
            }
            if (tid < 256) gram[(par ^ 1) * 256 + tid] = gr_nx;
        }

        // ---- (3) the 16 sequential updates of the block, one response per lane --------
        __builtin_amdgcn_sched_barrier(0);
        double U[kBlk];
        {
            const double* pr = part + ((size_t)(par * S) * T + qc) * 256 + r;
#pragma unroll
            for (int t = 0; t < kBlk; ++t) {
                double u = 0.0;
                for (int ss = 0; ss < S; ++ss)
                    u += pr[(size_t)ss * T * 256 + (t >> 2) * 64 + 16 * (t & 3)];
                U[t] = u;
            }
        }
        const double* gm_l = gram + par * 256;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kBlk; ++t) {
            const int i = 16 * b + t;
            const bool act = i < P.nsteps;
            const double lod = steps[i].lod, xtx = steps[i].xtx;
            const int j = steps[i].j;
            const double bo = bol[(par * 16 + t) * 64 + lane];
            const double dot = fma(xtx, bo, U[t]);
            const double m = c_scale * dot;                            // coreLoop.cpp:42-43
            const double a = P.cfac * ((lod - m * m * c_hinv) + c_cst);  // :45-46
            const double e = exp(-fabs(a));
            const double gmv = ((a >= 0.0) ? e : 1.0) / (1.0 + e);
            const double bn = m * gmv;                                 // :48
            const double delta = (act && cvalid) ? (bo - bn) : 0.0;
            if (act && writer) {
                double* o = P.state + (size_t)j * d + kc;
                o[0] = gmv;
                o[pd] = m;
                o[2 * pd] = bn;
                acc_g += gmv;
                acc_m2 = fma(gmv, fma(m, m, c_s2), acc_m2);
                acc_b2 = fma(bn, bn, acc_b2);
            }
            dl[t * 64 + lane] = delta;
#pragma unroll
            for (int t2 = t + 1; t2 < kBlk; ++t2) U[t2] = fma(gm_l[t2 * 16 + t], delta, U[t2]);
            // keep the steps apart: everything a later step loads (Gram column, beta_old, step
            // record) would otherwise be hoisted to the top of the chain and spill
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- (4) R += X_J delta on the matrix cores ------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        double bop[T][4];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bop[tau][kk] = dl[(4 * kk + q) * 64 + 16 * tau + r];
        }
#pragma unroll
        for (int g = 0; g < KW; ++g) {  // g = 4 * tile + kk
            const int slot = (g + KW) % NR;
            const double a = ring[slot];
            if (g + NR == KW) xp += blk_skip;  // the stream continues in the next block
            if (g + NR < KW || more) {
                ring[slot] = *xp;
                xp += 64;
            }
            if ((g & 7) == 7) asm volatile("" : "+v"(xp));
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
                rv[tau][g >> 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop[tau][g & 3],
                                                                       rv[tau][g >> 2], 0, 0, 0);
        }
    }

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
        // part[s][tau][lane] <- partial sum of squares; lane (q, r) then adds S * 4 values
#pragma unroll
        for (int tau = 0; tau < T; ++tau) part[((size_t)s * T + tau) * 64 + lane] = ss2[tau];
        __syncthreads();
        if (writer) {
            double tot = 0.0;
            for (int ss = 0; ss < S; ++ss) {
                const double* pp = part + ((size_t)ss * T + qc) * 64 + r;
                tot += (pp[0] + pp[16]) + (pp[32] + pp[48]);
            }
            P.st[kc] = acc_g;
            P.st[(size_t)d + kc] = acc_m2;
            P.st[2 * (size_t)d + kc] = acc_b2;
            P.st[3 * (size_t)d + kc] = tot;
        }
    }
}

}  // namespace locus
