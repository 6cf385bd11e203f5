// CAVI sweep on the fp64 matrix cores of gfx950 -- blocked coordinate ascent with look-ahead.
//
// Same update as sweep_kernel.h (src/coreLoop.cpp:36-52, coreStructLoop.cpp:34-50,
// coreZLoop / coreProbitLoop), restated so that the O(n) work per (covariate, response)
// pair runs as v_mfma_f64_16x16x4_f64 and overlaps with the sequential element-wise chain.
//
// The sweep order is cut into blocks of 16 consecutive covariates.  With R the residual before
// block b, d_s = beta_old - beta_new of step s and A_b = X_b' X_b (16 x 16),
//     X_t' R^(t) = G_t + sum_{s<t} A_b[t][s] d_s,      G = X_b' R,      R += X_b d  afterwards
// (the folding of src/coreBatch.cpp:32-33, re-associated).  Look-ahead: G of block b is formed
// from the residual that has seen the updates of blocks <= b - 2 only,
//     G_b = X_b' (R_{b-2} applied) + (X_b' X_{b-1}) d_{b-1},
// so the dense work of iteration `it` (apply block it-2, project on block it) does not depend on
// the chain of block it-1 that runs at the same time.
//
// A workgroup owns 16 T responses for the whole sweep and has two roles:
//   waves 0..6 (matrix role): wave w keeps rows [16 KT w, 16 KT (w + 1)) of those columns as MFMA
//       accumulators (C/D layout of R += X_b d == B-operand layout of G = X_b' R) and streams its
//       X operands from global memory in consumption order (pack_blocks_kernel); never touches
//       the element-wise math;
//   wave 7 (chain role): one response per lane; sums the 7 partial G tiles and the cross-block
//       correction from LDS, runs the 16 sequential updates of the block, writes gam / mu /
//       beta and the deltas (LDS) that the matrix waves consume one iteration later.
// The cross-block correction of the NEXT block, (X_{b+1}' X_b) d_b, is formed by one matrix wave
// (with 7 matrix waves: the one that shares its SIMD with the chain wave only and idles most):
// it follows the chain's progress word in LDS and issues 4 T MFMAs, one group as soon as each
// group of four deltas is known.  ONE barrier per iteration publishes deltas, partial G,
// cross correction, beta_old and step scalars.
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
// Matrix-role waves per workgroup.  Measured on MI355X: an fp64 MFMA stream on the SIMD of the
// chain wave slows the chain by 1.5-1.9x.  T = 2 is bound by the matrix work, so it spreads it
// over all four SIMDs (7 matrix waves; the chain wave, id 7, shares SIMD 3 with matrix wave 3).
// T = 1 is bound by the chain: 6 matrix waves on SIMDs 0..2 and the chain wave (id 3) alone on
// SIMD 3 (waves of a workgroup go to the SIMDs round-robin).
__host__ __device__ constexpr int mfma_waves(int T) { return T == 1 ? 6 : 7; }

// Order of the 4 KT operands of one part (update or dot) in a wave's stream: row tiles are
// interleaved in pairs so that consecutive MFMAs never accumulate into the same registers.
// o -> (row tile i, sub-step c); c = covariate slice (update) or 4-row k-step (dot).
__host__ __device__ constexpr int op_tile(int o, int KT)
{
    const int ip = o >> 3, rem = o & 7;
    return (2 * ip + 1 < KT) ? 2 * ip + (rem & 1) : 2 * ip;
}
__host__ __device__ constexpr int op_sub(int o, int KT)
{
    const int ip = o >> 3, rem = o & 7;
    return (2 * ip + 1 < KT) ? (rem >> 1) : rem;
}
// inverse: (i, c) -> o
__host__ __device__ constexpr int op_index(int i, int c, int KT)
{
    const int ip = i >> 1;
    return (2 * ip + 1 < KT) ? 8 * ip + 2 * c + (i & 1) : 8 * ip + c;
}

// 16-row tiles per matrix wave (KT) the library instantiates, per T: fine steps so that the row
// padding 16 KT MW - n stays small for any n (n = 1500 needs KT = 16 at T = 1, not 21).  Values
// whose 4 KT has no divisor near the ring cap (13, 17, 19, 20) are left out.  One list for the
// host-side choice (mfma_sweep.hip) and the dispatch (mfma_sweep_inst.inc).
#define LOCUS_MFMA_KTS_T1(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(14) X(16) X(18) X(21)
#define LOCUS_MFMA_KTS_T2(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
// mfma_sweep1_kernel with the rows cut over 2 workgroups (12 matrix waves): n <= 2112
#define LOCUS_MFMA1_KTS_C2(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
// ... over 3 workgroups (18 matrix waves): KT = 7 covers the n <= 2016 of the matrix-core sweep
#define LOCUS_MFMA1_KTS_C3(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// two tiles (32 responses) per cluster of 2, 3 or 4 workgroups: 16 KT residual registers per wave
#define LOCUS_MFMA1_KTS_T2(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

struct MfmaSweepParams {
    int n, p, d;
    int k_begin, k_end;  // responses [k_begin, k_end) of the d columns are swept by this launch
    int nblk;            // blocks of 16 steps (last one padded with inactive steps)
    int nsteps;          // active steps
    const StepRec* steps;  // [16 * nblk + 16]
    const int* err;
    const double* XS;    // [MW][nblk + 2][8 KT][64]: per wave and iteration `it`, the 4 KT update
                         // operands of block it - 2 then the 4 KT dot operands of block it,
                         // consecutive operands paired per lane (one 16-byte load = 2 operands)
    const double* cg;    // [nblk + 2][16][16]: cg[b][t][t'] = X_t' X_t' of block b for t' > t, else 0
    const double* axo;   // [nblk + 2][64][4]: X_{it-1}' X_{it-2} as the 4 A operands (per lane) of a
                         // 16x16x16 product
    const double* sc4;   // [16 * (nblk + 2)][4]: lod, xtx, bits of (long long) j * d, 0 of every step
    const int* jidx;     // [16 * (nblk + 2)]  covariate index of every step (tail: 0)
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
typedef const double __attribute__((address_space(4))) * CDblPtr;
typedef const int __attribute__((address_space(4))) * CIntPtr;

// X operand pairs in flight per matrix wave (must divide 4 KT)
template <int T, int KT>
constexpr int ring_pairs()
{
    // largest divisor of 4 KT within the register budget (T = 2 keeps 16 KT residual registers)
#if defined(LOCUS_RING_T1) && defined(LOCUS_RING_T2)
    constexpr int cap = (T == 1) ? LOCUS_RING_T1 : LOCUS_RING_T2;
#else
    // KT = 21 at T = 1: 168 residual registers.  T = 2: 9 pairs while the 16 KT residual registers
    // leave room (KT = 9, n = 1000: 8.80 -> 8.69 ms for 16 000 steps; 12 gave 8.77), else 6
    constexpr int cap = (T == 1) ? (KT > 18 ? 7 : 12) : (KT <= 9 ? 9 : 6);
#endif
    int best = 1;
    for (int r = 1; r <= cap; ++r)
        if ((4 * KT) % r == 0) best = r;
    return best;
}

// Diagnostic build only (-DLOCUS_STAMPS, `make diag`): per-phase cycle sums of wave 0 (slots
// 0..4) and of the chain wave (slots 5..7) of every workgroup.  No stamp executes in the
// product build.
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

template <int T, int KT>
__global__ void __launch_bounds__(64 * (mfma_waves(T) + 1))
mfma_sweep_kernel(const MfmaSweepParams P)
{
    static_assert(T >= 1 && T <= 2, "1 or 2 response tiles per workgroup");
    constexpr int NOP = 8 * KT;        // X operands per wave and iteration
    constexpr int NPAIR = NOP / 2;     // 16-byte loads per wave and iteration
    constexpr int NR2 = ring_pairs<T, KT>();
    static_assert(NPAIR % NR2 == 0, "ring must divide the iteration stream");
    constexpr int NACC = (T == 1) ? 2 : 1;  // dot accumulators per response tile
    constexpr int MW = mfma_waves(T);       // matrix waves
    constexpr int CW = (MW == 7) ? 7 : 3;   // id of the chain wave
    constexpr int XW = (MW == 7) ? 3 : 0;   // matrix wave (logical index) forming the cross correction
    constexpr int NSL = MW + 1;             // partial-G slots: matrix waves + cross correction
    constexpr int NTHR = 64 * (MW + 1);

    extern __shared__ __align__(16) double smem[];
    double* const part = smem;                          // [2][NSL][T][4][64]  partial G
    double* const dl = part + 2 * NSL * T * 256;     // [2][16][64]  deltas of a block
    double* const bol = dl + 2 * 16 * 64;               // [2][16][64]  beta_old of a block
    double* const gram = bol + 2 * 16 * 64;             // [2][16][16]  cg of a block (by columns)
    double* const scl = gram + 2 * 256;                 // [2][16][4]   lod, xtx, row offset, -
    volatile int* const prog = reinterpret_cast<volatile int*>(scl + 2 * 64);  // steps the chain has done

    if (*P.err != 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, kept scalar
    const int q = lane >> 4, r = lane & 15;
    const int n = P.n, d = P.d;
    const int k0 = P.k_begin + blockIdx.x * 16 * T;
    const int kend = P.k_end;
    const size_t pd = (size_t)P.p * d;
    const int nit = P.nblk + 2;

    // response of this lane in the "one response per lane" role (chain, beta_old staging)
    const int kc = k0 + lane;
    const bool cvalid = (lane < 16 * T) && (kc < kend);
    const int kcs = cvalid ? kc : 0;

    for (int e = tid; e < 2 * 16 * 64; e += NTHR) dl[e] = 0.0;  // deltas of blocks -2, -1
    for (int e = tid; e < T * 256; e += NTHR)
        part[(MW * T) * 256 + e] = 0.0;  // cross correction of block 0
    if (tid == 0) *prog = 0;
    __syncthreads();

#ifdef LOCUS_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif

    if (w != CW) {
        const int m = (w > CW) ? w - 1 : w;  // logical matrix-wave index
        // =========================== matrix role ==========================================
        const int row0 = m * 16 * KT;
        const CIntPtr jidx = (CIntPtr)(unsigned long long)P.jidx;

        // residual tiles: B operand of the dot MFMAs, C/D of the update MFMAs
        d4 rv[T][KT];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
            const int k = k0 + 16 * tau + r;
            const double* Rk = P.R + (size_t)(k < kend ? k : 0) * n;
#pragma unroll
            for (int i = 0; i < KT; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = row0 + 16 * i + 4 * g + q;
                    rv[tau][i][g] = (k < kend && row < n) ? Rk[row] : 0.0;
                }
            }
        }

        // this wave's X operands: ONE contiguous stream in consumption order
        const double2* xp =
            reinterpret_cast<const double2*>(P.XS) + ((size_t)m * nit) * NPAIR * 64 + lane;
        double2 ring[NR2];
#pragma unroll
        for (int i = 0; i < NR2; ++i) {
            ring[i] = *xp;
            xp += 64;
        }
        // Staging jobs, one shape for every wave (two 16-byte loads per lane in the middle of the
        // iteration, consumed at its end, nothing carried across iterations):
        //   wave XW: A operands of the cross-block product X_{it+1}' X_it   (axo[it + 1], in registers)
        //   wave 1 : Gram columns of block `it`                              (cg  -> LDS `gram`)
        //   wave 2 : step scalars {lod, xtx, row offset} of block `it`       (sc4 -> LDS `scl`)
        // The others load the Gram columns too and drop them (uniform code, no per-role registers).
        const double* stg_src = P.cg + 4 * lane;
        size_t stg_stride = 256;
        if (m == XW) {
            stg_src = P.axo + 256 + 4 * lane;
        } else if (m == 2) {
            stg_src = P.sc4 + 4 * (lane & 15);
            stg_stride = 64;
        }

        for (int it = 0; it < nit; ++it) {
            const int par = it & 1;
            // covariates of the rows of block `it` whose beta_old this wave stages: m, m + MW, m + 2 MW
            int jn[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) jn[u] = jidx[16 * it + ((m + MW * u) & 15)];

            __syncthreads();  // B1: deltas of block it - 2 are in dl[par]
            MSTAMP(0);
            double bop[4][T];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
                    bop[c][tau] = dl[(par * 16 + 4 * c + q) * 64 + 16 * tau + r];

            MSTAMP(1);

            // one slot of the stream: operand o of this iteration
            d4 acc[T][NACC];
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[tau][a] = d4{0.0, 0.0, 0.0, 0.0};

            auto slot = [&](auto oc) {
                constexpr int o = decltype(oc)::value;          // 0 .. NOP - 1
                constexpr int sl = (o >> 1) % NR2;
                const double a = (o & 1) ? ring[sl].y : ring[sl].x;
                if constexpr ((o & 1) != 0) {  // pair consumed: request the pair NR2 loads ahead
                    ring[sl] = *xp;
                    xp += 64;
                }
                if constexpr (o < 4 * KT) {    // R += X_{it-2} d_{it-2}
                    constexpr int i = op_tile(o, KT), c = op_sub(o, KT);
                    sfor<T>([&](auto tc) {
                        constexpr int tau = decltype(tc)::value;
                        rv[tau][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop[c][tau], rv[tau][i],
                                                                          0, 0, 0);
                    });
                } else {                       // G_it partial = X_it' R over this wave's rows
                    constexpr int i = op_tile(o - 4 * KT, KT), g = op_sub(o - 4 * KT, KT);
                    sfor<T>([&](auto tc) {
                        constexpr int tau = decltype(tc)::value;
                        constexpr int a2 = (NACC == 2) ? (i & 1) : 0;
                        acc[tau][a2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rv[tau][i][g],
                                                                            acc[tau][a2], 0, 0, 0);
                    });
                }
            };
            sfor<4 * KT>(slot);
            MSTAMP(3);
            // staging loads of block `it` (latency hidden behind the dot MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            double bo_ld[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) bo_ld[u] = P.state[2 * pd + (size_t)jn[u] * d + kcs];
            const double2 sg0 = *reinterpret_cast<const double2*>(stg_src + (size_t)it * stg_stride);
            const double2 sg1 = *reinterpret_cast<const double2*>(stg_src + (size_t)it * stg_stride + 2);
            __builtin_amdgcn_sched_barrier(0);
            sfor<4 * KT>([&](auto oc) {
                slot(std::integral_constant<int, 4 * KT + decltype(oc)::value>{});
            });
            MSTAMP(4);
            {
                double* pb = part + ((par * NSL + m) * T) * 256 + lane;
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        double v = acc[tau][0][g];
                        if constexpr (NACC == 2) v += acc[tau][1][g];
                        pb[tau * 256 + g * 64] = v;
                    }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = m + MW * u;
                if (t < 16) bol[(par * 16 + t) * 64 + lane] = bo_ld[u];
            }
            if (m == 1) {
                double2* gd = reinterpret_cast<double2*>(gram + par * 256) + 2 * lane;
                gd[0] = sg0;
                gd[1] = sg1;
            } else if (m == 2) {
                double2* sd = reinterpret_cast<double2*>(scl + (par * 16 + (lane & 15)) * 4);
                sd[0] = sg0;
                sd[1] = sg1;
            } else if (m == XW) {
                // cross correction of block `it` from the deltas of block it - 1, which the chain
                // wave is producing right now: follow its progress word
                if (it >= 1 && it <= P.nblk) {
                    const double ax[4] = {sg0.x, sg0.y, sg1.x, sg1.y};
                    d4 xc[T];
#pragma unroll
                    for (int tau = 0; tau < T; ++tau) xc[tau] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int target = 16 * (it - 1) + 4 * c + 4;
                        while (*prog < target) __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int tau = 0; tau < T; ++tau) {
                            const double bq = dl[((par ^ 1) * 16 + 4 * c + q) * 64 + 16 * tau + r];
                            xc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[c], bq, xc[tau], 0, 0, 0);
                        }
                    }
                    double* pb = part + ((par * NSL + MW) * T) * 256 + lane;
#pragma unroll
                    for (int tau = 0; tau < T; ++tau)
#pragma unroll
                        for (int g = 0; g < 4; ++g) pb[tau * 256 + g * 64] = xc[tau][g];
                }
            }
            MSTAMP(2);   // (diagnostic builds: staging stores + cross correction, "upd0+B2" column)
        }

#ifdef LOCUS_STAMPS
        if (lane == 0 && m == (int)(blockIdx.x % MW) && blockIdx.x < 4096)
            for (int k = 0; k < 5; ++k) g_mfma_stamps[blockIdx.x * 8 + k] = tacc[k];
#endif
        // residual back to memory; partial sums of squares for the fused statistics
        __syncthreads();  // E1: the chain wave is done with `part`
        // lane id and bounds are laundered so that the compiler does not keep the masks and
        // addresses of the prologue loads alive (in SGPRs / VGPRs) across the whole sweep
        int lane_e = lane, n_e = n, d_e = kend;
        asm volatile("" : "+v"(lane_e));
        asm volatile("" : "+s"(n_e), "+s"(d_e));
        const int q_e = lane_e >> 4, r_e = lane_e & 15;
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
            const int k = k0 + 16 * tau + r_e;
            double* Rk = P.R + (size_t)(k < d_e ? k : 0) * n_e;
            double a0 = 0.0;
#pragma unroll
            for (int i = 0; i < KT; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = row0 + 16 * i + 4 * g + q_e;
                    const double v = rv[tau][i][g];
                    if (k < d_e && row < n_e) Rk[row] = v;
                    a0 = fma(v, v, a0);
                }
            }
            part[(m * T + tau) * 64 + lane] = a0;
        }
        __syncthreads();  // E2
    } else {
        // =========================== chain role ===========================================
        // The chain shares its SIMD with matrix wave XW: its instructions go first when both are
        // ready (block 20.1k -> 19.1k cycles).  Running the two one after the other instead (cross
        // correction first, XW's stream after the chain) was measured slower: the chain alone
        // takes 11.6k cycles per block here, and 11.6k + 9.2k of MFMA exceeds the 19.7k of the
        // other three SIMDs, while interleaved the chain's LDS and store instructions overlap
        // with the MFMAs.  The short-dependent-path step of mfma_sweep1_kernel.h (Estrin
        // exponential, next dot as one FMA after the division) was tried here too: parity green
        // but 1 % slower (block 20.1k cycles) -- next to an MFMA stream the chain pays per
        // instruction, not per dependent level, and that form has a few more of them.
        __builtin_amdgcn_s_setprio(3);
        double c_scale = 0.0, c_hinv = 0.0, c_cst = 0.0, c_s2 = 0.0;
        if (cvalid) {
            c_scale = P.scale[kc];
            c_hinv = P.hinv[kc];
            c_cst = P.cst[kc];
            c_s2 = P.s2k[kc];
        }
        const double cfac = P.cfac;
        double acc_g = 0.0, acc_m2 = 0.0, acc_b2 = 0.0;
        const int qc = (q < T) ? q : 0;

        for (int it = 0; it < nit; ++it) {
            __syncthreads();  // B1
            MSTAMP(7);
            const int b = it - 1;  // block whose chain runs now
            if (b < 0 || b >= P.nblk) continue;
            const int par = b & 1;
            const double* pr = part + ((par * NSL) * T + qc) * 256 + r;
            const double* bo_l = bol + (par * 16) * 64 + lane;
            double* dl_w = dl + (par * 16) * 64 + lane;
            const double* gm_l = gram + par * 256;
            const double* sc_l = scl + par * 64;

            double corr[kBlk];  // sum_{s<t} A[t][s] delta_s
#pragma unroll
            for (int t = 0; t < kBlk; ++t) corr[t] = 0.0;
            double bo_c = bo_l[0];
            double2 sc_c = *reinterpret_cast<const double2*>(sc_l);
            long long off_c = __double_as_longlong(sc_l[2]);
            double ps_c;  // sum of the partial G of the current step over the slots
            {
                double g0[8];
                g0[7] = 0.0;
#pragma unroll
                for (int ss = 0; ss < NSL; ++ss) g0[ss] = pr[ss * T * 256];
                ps_c = ((g0[0] + g0[1]) + (g0[2] + g0[3])) + ((g0[4] + g0[5]) + (g0[6] + g0[7]));
            }
            __builtin_amdgcn_sched_barrier(0);
            sfor<kBlk>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const bool act = 16 * b + t < P.nsteps;
                // requests for step t + 1
                double bo_n = 0.0;
                double2 sc_n = make_double2(0.0, 0.0);
                long long off_n = 0;
                double gp_n[8];
                sfor<8>([&](auto sc) { gp_n[decltype(sc)::value] = 0.0; });
                if constexpr (t + 1 < kBlk) {
                    bo_n = bo_l[(t + 1) * 64];
                    sc_n = *reinterpret_cast<const double2*>(sc_l + 4 * (t + 1));
                    off_n = __double_as_longlong(sc_l[4 * (t + 1) + 2]);
                    sfor<NSL>([&](auto sc) {
                        constexpr int ss = decltype(sc)::value;
                        gp_n[ss] = pr[ss * T * 256 + ((t + 1) >> 2) * 64 + 16 * ((t + 1) & 3)];
                    });
                }
                // column t of the block's Gram matrix: needed after the exp / division chain
                double gcol[kBlk];
                sfor<kBlk - 1 - t>([&](auto c2) {
                    constexpr int t2 = t + 1 + decltype(c2)::value;
                    gcol[t2] = gm_l[t * 16 + t2];
                });
                const double lod = sc_c.x, xtx = sc_c.y;

                const double u = ps_c + corr[t];
                const double dot = fma(xtx, bo_c, u);
                const double m = c_scale * dot;                            // coreLoop.cpp:42-43
                const double a = cfac * ((lod - m * m * c_hinv) + c_cst);  // :45-46
                const double e = exp_nonpos(-fabs(a));
                const double gmv = div_1to2((a >= 0.0) ? e : 1.0, 1.0 + e);
                const double bn = m * gmv;                                 // :48
                const double delta = (act && cvalid) ? (bo_c - bn) : 0.0;
                sfor<kBlk - 1 - t>([&](auto c2) {
                    constexpr int t2 = t + 1 + decltype(c2)::value;
                    corr[t2] = fma(gcol[t2], delta, corr[t2]);
                });
                dl_w[t * 64] = delta;
                if constexpr ((t & 3) == 3) {
                    // LDS executes a wave's accesses in order: a compiler barrier is enough to
                    // publish the four deltas before the progress word
                    asm volatile("" ::: "memory");
                    *prog = 16 * b + t + 1;
                }
                if (act && cvalid) {
                    double* o = P.state + off_c + kc;
                    o[0] = gmv;
                    o[pd] = m;
                    o[2 * pd] = bn;
                    acc_g += gmv;
                    acc_m2 = fma(gmv, fma(m, m, c_s2), acc_m2);
                    acc_b2 = fma(bn, bn, acc_b2);
                }
                bo_c = bo_n;
                sc_c = sc_n;
                off_c = off_n;
                ps_c = ((gp_n[0] + gp_n[1]) + (gp_n[2] + gp_n[3])) +
                       ((gp_n[4] + gp_n[5]) + (gp_n[6] + gp_n[7]));
                __builtin_amdgcn_sched_barrier(0);
            });
            MSTAMP(6);
        }
#ifdef LOCUS_STAMPS
        if (lane == 0 && blockIdx.x < 4096)
            for (int k = 5; k < 8; ++k) g_mfma_stamps[blockIdx.x * 8 + k] = tacc[k];
#endif
        __syncthreads();  // E1
        __syncthreads();  // E2: partial sums of squares of the matrix waves are in `part`
        if (P.st != nullptr && cvalid) {
            double tot = 0.0;
#pragma unroll
            for (int ss = 0; ss < MW; ++ss) {
                const double* pp = part + (ss * T + qc) * 64 + r;
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
