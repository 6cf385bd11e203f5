// Matrix-core CAVI sweep for ONE tile of 16 T responses (T = 1, 2) per cluster of CS workgroups
// (coreLoop / coreZLoop / coreStructLoop / coreProbitLoop, src/coreLoop.cpp:36-52,
// coreStructLoop.cpp:34-50) -- the latency regime: d_loc <= 16 responses per CU (C2, C5, the
// remainder of C3 on one GPU, every multi-GPU shard).
//
// Same blocked look-ahead restatement as mfma_sweep_kernel.h (blocks of 16 covariates,
//     X_t' R^(t) = G_t + sum_{s<t} A_b[t][s] d_s,   G_b = X_b' R_(b-L) + sum_{l<L} (X_b' X_{b-l}) d_{b-l}),
// re-cut so that the sequential element-wise chain carries nothing but the chain:
//
//   waves 0,1,2,4,5,6 (matrix role, SIMDs 0..2): rows of the residual tile as MFMA accumulators,
//       X operands streamed from global memory in consumption order.  Matrix waves 1 .. L-2 also
//       form the cross-block corrections of the known lags (l >= 2) of the block the chain runs
//       next; with one workgroup per tile matrix wave 0 adds them to the reduced G and follows
//       the chain's progress word for lag 1;
//   wave 3 (chain role, with the helper on SIMD 3, which runs no MFMA stream): per step ONE LDS
//       read for gx and beta_old, the step's scalars as an LDS broadcast, its share of the Gram
//       column, the update (21 dependent instructions: gam = 1 / (1 + exp(a)) with a table-driven
//       exponential; everything else in their latency gaps), three LDS writes (delta, mu, gamma).
//       Where the chain is what a block waits for, the wave's 64 lanes are 4 (T = 2: 2) groups
//       that run the same chain and share the in-block corrections (mfma1_lane_groups()).  No
//       global memory access, no partial-sum tree, no address arithmetic.  In a cluster it also
//       finishes gx of its next block right after its 16th step (lag 1: four MFMAs on the deltas
//       it just wrote);
//   wave 7 (helper role): everything else -- sums the matrix waves' partial G tiles, exchanges
//       them with the other workgroups of the cluster, stages beta_old, scalars, Gram columns
//       and cross-Gram operands one block ahead, writes gam / mu / beta of the block the chain
//       finished one iteration ago to global memory and accumulates the fused column statistics.
//
// Row-split clusters (CS = 2, 3): the 16 responses' rows are cut over CS workgroups on CS CUs.
// Each member projects / updates its own rows, publishes its partial G tile (2 KB per block)
// through global memory as tagged 16-byte granules {block + 1, one double, block + 1} written and
// read as single sc1 (agent-scope) accesses (a granule is valid when both tags match: no flag, no
// fence in steady state), and EVERY member runs the same chain on the same sums (added in member
// order, so the chains are bit-identical and no delta ever has to travel).  The tag is what makes
// the exchange safe when the members sit on different XCDs: an agent-scope load is served by the
// reader's own L2, which other XCDs' stores never refresh, so a slot can read as what it held
// sixteen blocks ago until the line is evicted -- stale data carries a stale tag and is fetched
// again.  (Round 3 tried whole-double granules validated against an "empty" bit pattern, half
// the loads and registers: right on a quiet GPU, where blocks b and b + 8 share an XCD; wrong
// once in a while when other kernels were in flight at launch -- a stale line passes that test.)  The look-ahead is
// deepened to L = 5 blocks so that a partial tile has two block times (~6 us) to arrive.  All
// waits are bounded: a partner that does not show up raises the error word instead of hanging
// the GPU.
#pragma once
#include "mfma_sweep_kernel.h"

namespace locus {

struct Mfma1Params {
    int n, p, d;
    int k_begin, k_end;   // responses [k_begin, k_end) are swept by this launch
    int ntiles;           // tiles of 16 responses = clusters
    int nblk, nsteps;     // blocks of 16 steps, active steps
    int* err;             // bits 0, 1: bad permutation (prologue); bit 2 (value 4): a cluster
                          // partner timed out -- the members stop at the next block, store
                          // nothing from then on and leave R as it was before the sweep
    const double* XS;     // [CS * 6][nit][8 KT][64] operand streams (pack_blocks_kernel), nit = nblk + L
    const double* cg;     // [nit][16][16]  column t of block b's Gram matrix (X_t2' X_t for t2 > t, else 0),
                          //               entry t2 at [t][t2 % NQ][t2 / NQ], NQ = mfma1_lane_groups()
    const double* ax;     // [L - 1][nit + 1][64][4]  A operands of (X_b' X_{b-l}), l = 1 .. L-1
    const double* sc2;    // [16 nit][4]   lod, xtx, g[t][t+1] (Gram entry below the diagonal), 0 of every
                          //               step (inactive steps: zeros)
    const int* jidx;      // [16 nit]      covariate of every step (inactive: 0)
    const double* scale;  // per-response constants, as SweepParams
    const double* hinv;
    const double* cst;
    const double* s2k;
    double cfac;
    double* R;            // n x d column-major
    double* state;        // [3][p][d]
    double* st;           // [4][d] or nullptr
    // CS > 1: exchange of the members' partial G tiles, [ntiles][kXTile] 8-byte words per cluster:
    // kXRing slots x CS members x T x 256 granules of 16 bytes {tag, double, tag}, tag = block + 1
    // (zeroed by the host before every launch: a granule is valid when both tags match), then
    // CS x T x 64 doubles for the members' sums of squares of the residual at the end of the sweep
    unsigned long long* xbuf;
    int* xflag;           // CS > 1: [ntiles][CS][32] end-of-sweep flags of the members
    int spread;           // test hook (LOCUS_HIP_TEST_SPREAD_MEMBERS): members of a cluster are
                          // CONSECUTIVE workgroups, i.e. on different XCDs under round-robin
                          // placement -- the exchange must not depend on sharing an L2
    int drop_member;      // test hook (LOCUS_HIP_TEST_DROP_MEMBER): this member of every cluster
                          // returns at entry, as a workgroup that was never dispatched; -1 = none
};

constexpr int kXRing = 16;  // exchange slots per member; >= 2 L + 1 blocks can be in flight
typedef unsigned int xgran __attribute__((ext_vector_type(4)));  // {tag, low, high, tag}
constexpr int kXAuxSc1 = 16;  // cache-policy operand of the raw buffer intrinsics on gfx94x/95x: sc1
__host__ __device__ constexpr size_t xbuf_tile_words(int CS, int T = 1)
{
    return ((size_t)kXRing * CS * 512 + CS * 64) * T;
}

__host__ __device__ constexpr int mfma1_lag(int CS) { return CS == 1 ? 3 : 5; }

template <int KT, int T = 1>
constexpr int ring1_pairs()
{
    // X operand pairs in flight per matrix wave (must divide 4 KT); the deep tiles keep fewer so
    // that the residual (8 KT registers) and the ring fit 256 registers without spilling
#if defined(LOCUS_RING1_OVR)      // experiments: -D'LOCUS_RING1_OVR(KT,T)=((KT)==3?6:8)'
    constexpr int cap = LOCUS_RING1_OVR(KT, T);
#elif defined(LOCUS_RING1_CAP)
    constexpr int cap = LOCUS_RING1_CAP;
#else
    // (KT <= 4 runs chain-bound with one workgroup per tile: a deeper ring only delays the
    // helper's loads; KT = 5, 6 are bound by the operand fetch, where 12 pairs in flight per wave
    // took the n = 1000 shard from 10.9 to 10.5 ms)
    // (T = 2 keeps 16 KT residual registers: the ring shrinks as the tile deepens.  Deeper is not
    // better: at T = 2, CS = 3, n = 1000 -- KT = 4 and, with the odd deal, 3 -- rings of 8 and 6 pairs
    // run the d_loc = 2500 shard in 12.5 ms, 8 and 12 in 15.8, 16 and 6 in 13.9, 4 and 6 in 12.8)
    constexpr int cap = (T == 2) ? (KT <= 6 ? 8 : 6) : ((KT > 14 || KT <= 4) ? 6 : 12);
#endif
    int best = 1;
    for (int r = 1; r <= cap; ++r)
        if ((4 * KT) % r == 0) best = r;
    return best;
}

__device__ __forceinline__ void lds_publish(int* word, int v)
{
    // LDS executes one wave's accesses in order: the data stores before this one are visible to
    // whoever sees the new value
    asm volatile("" ::: "memory");
    __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_peek(int* word)
{
    return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Diagnostic build only (-DLOCUS_STAMPS): per-phase cycle sums, [workgroup][role][8]; role 0 =
// matrix wave 0, 1 = matrix wave 5, 2 = chain, 3 = helper.  No stamp executes in the product build.
#ifdef LOCUS_STAMPS
__device__ unsigned long long g_mfma1_stamps[4096 * 4 * 8];
#define M1STAMP(k)                                                                       \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        unsigned long long t_;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        tacc[k] += t_ - tprev;                                                           \
        tprev = t_;                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#define M1STAMP_FLUSH(role)                                                              \
    do {                                                                                 \
        if (lane == 0 && blockIdx.x < 4096)                                              \
            for (int k_ = 0; k_ < 8; ++k_)                                               \
                g_mfma1_stamps[(blockIdx.x * 4 + (role)) * 8 + k_] = tacc[k_];           \
    } while (0)
#else
#define M1STAMP(k) do { } while (0)
#define M1STAMP_FLUSH(role) do { } while (0)
#endif

// Rows per matrix wave.  The six matrix waves of a workgroup sit two to a SIMD (waves m and m + 3
// of the matrix role); what bounds a block is the work of a SIMD, so the rows are dealt in 16-row
// tiles per SIMD: waves 0..2 take KT tiles each, waves 3..5 take KT - ODD.  With ODD = 1 a SIMD
// carries 2 KT - 1 tiles -- n = 1000 on three workgroups is 63 tiles = 9 SIMDs x 7 exactly, where
// an even deal (8 per SIMD) makes every SIMD multiply 14 % of padding.
__host__ __device__ constexpr int mfma1_member_tiles(int KT, int ODD) { return 3 * KT + 3 * (KT - ODD); }
__host__ __device__ constexpr int mfma1_wave_tile0(int m, int KT, int ODD)
{
    return m < 3 ? m * KT : 3 * KT + (m - 3) * (KT - ODD);
}

// Lane groups of the chain wave (see the chain role): 64 / (16 T) groups that run the same chain and
// share the in-block corrections where the chain wave is what a block waits for, one group where
// the matrix waves are (there the groups' wider LDS traffic only delays the matrix waves' operand
// reads: two-member clusters at n = 1000, d = 2000 took 10.4 ms with groups, 10.1 without).
// Thresholds: where the cost model of the launch plan (mfma_sweep.hip) puts the crossover, in
// half-tiles per matrix wave (2 kt = 2 KT - ODD).
__host__ __device__ constexpr int mfma1_lane_groups(int KT, int CS, int T, int ODD)
{
    const int kt2 = 2 * KT - ODD;
    if (T == 2) return kt2 <= 7 ? 2 : 1;
    return kt2 <= (CS == 1 ? 14 : (CS == 2 ? 8 : 10)) ? 4 : 1;
}

template <int KT, int CS, int T = 1, int ODD = 0>
__global__ void __launch_bounds__(512) mfma_sweep1_kernel(const Mfma1Params P)
{
    static_assert(ODD == 0 || (ODD == 1 && KT >= 2), "the short waves keep at least one tile");
    static_assert(T == 1 || T == 2, "one or two tiles of 16 responses per cluster");
    static_assert(T == 1 || CS > 1, "two tiles on ONE workgroup is mfma_sweep_kernel<2, KT>");
    constexpr int L = mfma1_lag(CS);
    constexpr int MW = 6;
    constexpr int TPM = mfma1_member_tiles(KT, ODD);   // 16-row tiles of a workgroup
    constexpr int TW = 16 * T;          // responses of a cluster = row length of the [t][k] LDS arrays
    constexpr int TB = 256 * T;         // doubles of one 16 x 16 T block of them
    constexpr int DR = 8;       // delta ring, blocks (power of two, >= L)
    constexpr int CHAIN_W = 3, HELP_W = 7;

    // (k = response within the cluster, 0 .. 16 T - 1 = 16 tau + r; "D layout" = [tau][g][lane])
    __shared__ __align__(16) double part[2 * MW * TB];   // partial G per matrix wave   [par][m][tau][g][lane]
    __shared__ __align__(16) double gfin[TB];            // reduced G of the next chain block, D layout
    // chain starting values and beta_old of a block, [par][{gx, beta_old}][t][k]: one array, so that
    // the chain reads gx and beta_old of a step with ONE ds_read2st64_b64
    __shared__ __align__(16) double gxbo[4 * TB];
    double* const gx = gxbo;           // gx of parity par:       gx + par * 2 TB
    double* const bol = gxbo + TB;     // beta_old of parity par: bol + par * 2 TB
    __shared__ __align__(16) double dl[DR * TB];         // deltas                      [blk & 7][t][k]
    __shared__ __align__(16) double mg[2 * 2 * TB];      // mu, gamma of a chain block  [par][2][t][k]
    __shared__ __align__(16) double gram[2 * 256];       // Gram columns of a chain block [par][t][t2]
    __shared__ __align__(16) double scl[2 * 64];         // lod, xtx, g[t][t+1], - of its steps [par][t][4]
    // CS > 1: A operands of the lag-1 cross-block correction of a block, staged by the helper [par][lane][4]
    __shared__ __align__(16) double axl[(CS > 1) ? 2 * 256 : 2];
    // CS > 1: corrections of the lags 2 .. L-1 (matrix waves 1 .. L-2), D layout, and their flags
    __shared__ __align__(16) double cx[(CS > 1) ? (L - 2) * TB : 2];
    __shared__ int cxf[4];
    __shared__ int jl[8 * 16];  // covariate of the steps of the blocks around the chain [blk & 7][t]
    __shared__ int prog;        // steps the chain has completed
    __shared__ int gfin_ready;  // iteration whose gfin is complete
    // CS > 1: 0, or the iteration at whose top every wave leaves the sweep (a partner timed out).
    // Written by the helper during iteration it as it + 2 and read by every wave behind the
    // barriers of it + 1 and it + 2: all waves see it at the same iteration.
    __shared__ int dead;
    __shared__ __align__(16) double etab[64];   // 2^(j/64): the chain wave's exponential (fastmath.h)
    // what the chain wave's lane groups 1 .. NQ-1 "store" (same values as group 0, never read): a
    // region as long as the offsets of a step's three result stores reach
    __shared__ double junk[2 * TB + 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int n = P.n, d = P.d;
    const int nit = P.nblk + L;

    // cluster g = tile of 16 responses, member c = slice of the rows; members of a cluster have
    // equal blockIdx % 8 (same XCD under round-robin placement: speed only, never correctness)
    const int bi = blockIdx.x;
    const int g_cl = P.spread ? bi / CS : (bi / (8 * CS)) * 8 + (bi & 7);
    const int c_mem = P.spread ? bi % CS : (bi >> 3) % CS;
    if (g_cl >= P.ntiles || *P.err != 0) return;
    if (CS > 1 && c_mem == P.drop_member) return;
    const int k0 = P.k_begin + TW * g_cl;
    const int kend = P.k_end;
    const size_t pd = (size_t)P.p * d;

    for (int e = tid; e < DR * TB; e += 512) dl[e] = 0.0;
    for (int e = tid; e < 4 * TB; e += 512) gxbo[e] = 0.0;
    for (int e = tid; e < 4 * TB; e += 512) mg[e] = 0.0;
    for (int e = tid; e < 512; e += 512) gram[e] = 0.0;
    if (tid < 128) scl[tid] = 0.0;
    if (tid < 128) jl[tid] = 0;
    if (tid == 0) { prog = 0; gfin_ready = -1; dead = 0; }
    if (tid < 4) cxf[tid] = -1;
    if (tid < 64) etab[tid] = kExp2Tab64[tid];
    __syncthreads();
#ifdef LOCUS_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif

    if (w != CHAIN_W && w != HELP_W) {
        // =========================== matrix role ==========================================
        const int m = (w > CHAIN_W) ? w - 1 : w;
        // the role is written once for a wave of KTW tiles and instantiated for KT and, with ODD,
        // for KT - 1 (waves 3..5)
        auto matrix_role = [&](auto ktw_c) {
        constexpr int KTW = decltype(ktw_c)::value;
        constexpr int NOP = 8 * KTW, NPAIR = NOP / 2, NR2 = ring1_pairs<KTW, T>();
        static_assert(NPAIR % NR2 == 0, "ring must divide the iteration stream");
        const int tile0 = c_mem * TPM + mfma1_wave_tile0(m, KT, ODD);
        const int row0 = tile0 * 16;

        d4 rv[T][KTW];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
            const int k = k0 + 16 * tau + r;
            const double* Rk = P.R + (size_t)(k < kend ? k : 0) * n;
#pragma unroll
            for (int i = 0; i < KTW; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = row0 + 16 * i + 4 * g + q;
                    rv[tau][i][g] = (k < kend && row < n) ? Rk[row] : 0.0;
                }
        }
        // the waves' streams follow one another: a tile contributes 8 operands = 4 pairs per iteration
        const double2* xp = reinterpret_cast<const double2*>(P.XS) + (size_t)tile0 * nit * 4 * 64 + lane;
        double2 ring[NR2];
#pragma unroll
        for (int i = 0; i < NR2; ++i) {
            ring[i] = *xp;
            xp += 64;
        }
        const size_t ax_lag = (size_t)(nit + 1) * 256;   // doubles per lag in P.ax
        // clusters: matrix waves 1 .. L-2 form the cross-block correction of one known lag each
        // (lag m + 1); its A operands come straight from global memory, requested one iteration
        // before they are used (they used to travel through the helper and LDS: 24 registers, six
        // loads and six LDS stores per block on the SIMD that the chain needs)
        double2 axn[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
        auto ax_request = [&](int bxn) {
            if constexpr (CS > 1) {
                if (m >= 1 && m <= L - 2 && bxn >= 0 && bxn < P.nblk) {
                    const double2* ap = reinterpret_cast<const double2*>(
                        P.ax + (size_t)m * ax_lag + (size_t)bxn * 256 + 4 * lane);
                    axn[0] = ap[0];
                    axn[1] = ap[1];
                }
            }
        };
        ax_request(2 - L);

        for (int it = 0; it < nit; ++it) {
            const int par = it & 1;
            __syncthreads();  // B1
            if constexpr (CS > 1) {
                const int dd_ = lds_peek(&dead);
                if (dd_ != 0 && it >= dd_) break;
            }
            M1STAMP(0);
            double bop[4][T];
            {
                const double* dp = dl + ((it - L) & (DR - 1)) * TB + q * TW + r;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int tau = 0; tau < T; ++tau) bop[c][tau] = dp[4 * TW * c + 16 * tau];
            }
            // Cross-block corrections of block bx, the one the chain runs in the NEXT iteration:
            //   gx = reduced G (helper) + sum_l (X_bx' X_{bx-l}) d_{bx-l},  l = 1 .. L-1.
            // One workgroup per tile (CS = 1): matrix wave 0 does all of them at the end of its
            // iteration.  Clusters: the lags >= 2, whose deltas are known, are spread over matrix waves
            // 1 .. L-2 (one lag each, right after the barrier, A operands staged in LDS by the
            // helper); wave 0 adds them up and is left with lag 1, which follows the chain.
            const int bx = it - L + 2;
            const bool bx_ok = bx >= 0 && bx < P.nblk;
            const bool xw = (CS == 1) && (m == 0) && bx_ok;  // clusters: the chain wave does it
            double2 axr[(CS > 1) ? 1 : L - 1][2];
            // clusters: this wave's lag (m + 1, for m = 1 .. L-2) of the cross-block corrections
            const bool do_corr = (CS > 1) && bx_ok && m >= 1 && m <= L - 2;
            double cdl[4][T];
            d4 ccx[T];
#pragma unroll
            for (int tau = 0; tau < T; ++tau) {
                ccx[tau] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int c = 0; c < 4; ++c) cdl[c][tau] = 0.0;
            }
            if constexpr (CS > 1) {
                // (requests only; the MFMAs and the result go into the operand stream below, a
                // quarter and a half of the way through, so that neither the LDS latency nor the
                // latency of four dependent MFMAs is paid at the top of the iteration)
                axr[0][0] = axn[0];
                axr[0][1] = axn[1];
                ax_request(bx + 1);
                if (do_corr) {
                    const double* dp = dl + ((bx - (m + 1)) & (DR - 1)) * TB + q * TW + r;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int tau = 0; tau < T; ++tau) cdl[c][tau] = dp[4 * TW * c + 16 * tau];
                }
            } else if (xw) {  // used at the end of this iteration only
#pragma unroll
                for (int l = 1; l < L; ++l) {
                    const double2* ap = reinterpret_cast<const double2*>(
                        P.ax + (size_t)(l - 1) * ax_lag + (size_t)bx * 256 + 4 * lane);
                    axr[l - 1][0] = ap[0];
                    axr[l - 1][1] = ap[1];
                }
            }
            // dot accumulators: two per tile at T = 1 (consecutive MFMAs never share registers);
            // with two tiles the tiles alternate, one each is enough
            constexpr int NACC = (T == 1) ? 2 : 1;
            d4 acc[T][NACC];
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
#pragma unroll
                for (int a2 = 0; a2 < NACC; ++a2) acc[tau][a2] = d4{0.0, 0.0, 0.0, 0.0};
            auto slot = [&](auto oc) {
                constexpr int o = decltype(oc)::value;
                if constexpr (CS > 1 && o == NOP / 4) {
                    if (do_corr) {
                        const double axv[4] = {axr[0][0].x, axr[0][0].y, axr[0][1].x, axr[0][1].y};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int tau = 0; tau < T; ++tau)
                                ccx[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(axv[c], cdl[c][tau],
                                                                                ccx[tau], 0, 0, 0);
                    }
                }
                if constexpr (CS > 1 && o == NOP / 2) {
                    if (do_corr) {
                        double* co = cx + (m - 1) * TB + lane;
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
#pragma unroll
                            for (int g = 0; g < 4; ++g) co[256 * tau + 64 * g] = ccx[tau][g];
                        lds_publish(&cxf[m - 1], it);
                    }
                }
                constexpr int sl = (o >> 1) % NR2;
                const double a = (o & 1) ? ring[sl].y : ring[sl].x;
                if constexpr ((o & 1) != 0) {
                    ring[sl] = *xp;
                    xp += 64;
                }
                if constexpr (o < 4 * KTW) {  // R += X_{it-L} d_{it-L}
                    constexpr int i = op_tile(o, KTW), c = op_sub(o, KTW);
                    sfor<T>([&](auto tc) {
                        constexpr int tau = decltype(tc)::value;
                        rv[tau][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bop[c][tau], rv[tau][i],
                                                                          0, 0, 0);
                    });
                } else {                     // partial G_it = X_it' R over this wave's rows
                    constexpr int i = op_tile(o - 4 * KTW, KTW), g = op_sub(o - 4 * KTW, KTW);
                    sfor<T>([&](auto tc) {
                        constexpr int tau = decltype(tc)::value;
                        constexpr int a2 = (NACC == 2) ? (i & 1) : 0;
                        acc[tau][a2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rv[tau][i][g],
                                                                            acc[tau][a2], 0, 0, 0);
                    });
                }
            };
            sfor<NOP>(slot);
            M1STAMP(1);
            {
                double* pb = part + (par * MW + m) * TB + lane;
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        double v = acc[tau][0][g];
                        if constexpr (NACC == 2) v += acc[tau][1][g];
                        pb[256 * tau + g * 64] = v;
                    }
            }
            if constexpr (CS == 1) if (xw) {   // (T == 1 here: strides of 256 / 16)
                d4 xc;
                const double* dp = dl + ((bx - 1) & (DR - 1)) * 256 + q * 16 + r;
                const double axv[4] = {axr[0][0].x, axr[0][0].y, axr[0][1].x, axr[0][1].y};
                {
                for (int spin = 0; lds_peek(&gfin_ready) < it && spin < (1 << 22); ++spin)
                    __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                M1STAMP(2);
#pragma unroll
                for (int g = 0; g < 4; ++g) xc[g] = gfin[g * 64 + lane];
                if constexpr (CS > 1) {
#pragma unroll
                    for (int l = 2; l < L; ++l) {   // fixed order: identical sums in every member
                        for (int spin = 0; lds_peek(&cxf[l - 2]) < it && spin < (1 << 22); ++spin)
                            __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int g = 0; g < 4; ++g) xc[g] += cx[(l - 2) * 256 + g * 64 + lane];
                    }
                } else {
#pragma unroll
                    for (int l = L - 1; l >= 2; --l) {
                        const double* dp2 = dl + ((bx - l) & (DR - 1)) * 256 + q * 16 + r;
                        const double axv2[4] = {axr[l - 1][0].x, axr[l - 1][0].y, axr[l - 1][1].x,
                                                axr[l - 1][1].y};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            xc = __builtin_amdgcn_mfma_f64_16x16x4f64(axv2[c], dp2[64 * c], xc, 0, 0, 0);
                    }
                }
                M1STAMP(3);
                // lag 1: the chain is producing these deltas right now
                if (lds_peek(&prog) >= 16 * bx) {
                    // the chain has already finished block bx - 1 (this wave's matrix work took
                    // longer): one batch of LDS reads, four MFMAs back to back
                    asm volatile("" ::: "memory");
                    double b4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) b4[c] = dp[64 * c];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        xc = __builtin_amdgcn_mfma_f64_16x16x4f64(axv[c], b4[c], xc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int target = 16 * (bx - 1) + 4 * c + 4;
                        for (int spin = 0; lds_peek(&prog) < target && spin < (1 << 22); ++spin)
                            __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
                        xc = __builtin_amdgcn_mfma_f64_16x16x4f64(axv[c], dp[64 * c], xc, 0, 0, 0);
                    }
                }
                }
                double* go = gx + (bx & 1) * 512 + q * 16 + r;
#pragma unroll
                for (int g = 0; g < 4; ++g) go[64 * g] = xc[g];
                M1STAMP(4);
            }
        }
#ifdef LOCUS_STAMPS
#ifndef LOCUS_STAMP_MA
#define LOCUS_STAMP_MA 0   // matrix waves whose stamps are kept (roles 0 and 1)
#define LOCUS_STAMP_MB 5
#endif
        if (m == LOCUS_STAMP_MA) M1STAMP_FLUSH(0);
        if (m == LOCUS_STAMP_MB) M1STAMP_FLUSH(1);
#endif

        __syncthreads();  // E1: everybody is done with `part`
        const bool alive = (CS == 1) || lds_peek(&dead) == 0;
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
            for (int i = 0; i < KTW; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = row0 + 16 * i + 4 * g + q_e;
                    const double v = rv[tau][i][g];
                    if (alive && k < d_e && row < n_e) Rk[row] = v;
                    a0 = fma(v, v, a0);
                }
            part[(m * T + tau) * 64 + lane] = a0;
        }
        __syncthreads();  // E2
        };  // matrix_role
        if constexpr (ODD == 1) {
            if (m < 3) matrix_role(std::integral_constant<int, KT>{});
            else matrix_role(std::integral_constant<int, KT - 1>{});
        } else {
            matrix_role(std::integral_constant<int, KT>{});
        }
    } else if (w == CHAIN_W) {
        // =========================== chain role ===========================================
        // The wave's 64 lanes are NQ = 64 / (16 T) groups of 16 T lanes.  Every group runs the same
        // chain for the same 16 T responses (lane & (TW - 1)) -- same instructions, so it costs
        // nothing -- and keeps ITS share of the in-block corrections: group q owns corr[t2] for
        // t2 = NQ i + q.  One FMA then applies a delta to NQ later steps at once (38 per block
        // instead of 120 at T = 1: every fp64 instruction of this wave costs ~5.7 cycles, on the
        // dependent path or off it), and the one correction a step needs from another group comes
        // through a ds_bpermute, which is not a vector-ALU instruction.
        // (NQ = 1 where the matrix waves bind, mfma1_lane_groups(): the plain form, lanes >= 16 T idle)
        constexpr int NQ = mfma1_lane_groups(KT, CS, T, ODD), NI = kBlk / NQ;
        const int rr = lane & (TW - 1), qq = (NQ == 1) ? 0 : lane / TW;
        const int kc = k0 + rr;
        const bool cvalid = kc < kend;
        double c_scale = 0.0, c_hinv = 0.0, c_cst = 0.0;
        if (cvalid) {
            c_scale = P.scale[kc];
            c_hinv = P.hinv[kc];
            c_cst = P.cst[kc];
        }
        const double cfac = P.cfac;

        for (int it = 0; it < nit; ++it) {
            __syncthreads();  // B1
            if constexpr (CS > 1) {
                const int dd_ = lds_peek(&dead);
                if (dd_ != 0 && it >= dd_) break;
            }
            M1STAMP(0);
            const int b = it - L + 1;  // block whose chain runs now
            if (b >= 0 && b < P.nblk && (NQ > 1 || lane < TW)) {
                const double* gx_l = gx + (b & 1) * 2 * TB + rr;
                const double* bo_l = bol + (b & 1) * 2 * TB + rr;
                // results: group 0 writes them, the other groups write the same values to a spare
                // slot of their own (no lane mask inside the step)
                double* dl_w = (qq == 0) ? dl + (b & (DR - 1)) * TB + rr : junk + lane;
                double* mg_w = (qq == 0) ? mg + (b & 1) * 2 * TB + rr : junk + lane;
                const double* sc_b = scl + (b & 1) * 64;      // wave-uniform addresses: broadcast reads
                const double* cg_l = gram + (b & 1) * 256 + qq * NI;   // this group's entries of a column

                // The dependent path of a step is cut to: dot -> dot^2 -> a -> exp -> 1 + e -> divide
                // -> dot of the NEXT step (one FMA).  With
                //     a_t = cB_t - cA dot_t^2,            cA = cfac scale^2 hinv,  cB_t = cfac (lod_t + cst)
                //     dot_{t+1} = pre_{t+1} - (g m_t) gam_t,
                //     pre_{t+1} = u_{t+1} + xtx_{t+1} bo_{t+1} + sum_{s<t} g[s][t+1] d_s + g[t][t+1] bo_t,
                // (g = column t of the block's Gram matrix, d_s = bo_s - m_s gam_s) everything but
                // the last FMA is known before gam_t is: m, beta, delta, the corrections of the
                // later steps and the LDS traffic of step t are issued inside step t + 1, in the
                // latency gaps of its chain (one scheduling region per step, fenced right after
                // the next dot).  coreLoop.cpp:42-48, re-associated (rounding-level differences).
                const double cA = cfac * c_scale * c_scale * c_hinv;
                const double cC = cfac * c_cst;
                // corr[i] = sum over the steps s whose delta is applied of g[s][t2] d_s, t2 = NQ i + qq
                double corr[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) corr[i] = 0.0;
                double bo_c = bo_l[0];
                double2 sc_c = *reinterpret_cast<const double2*>(sc_b);
                double sd_c = sc_b[2];                 // g[t][t+1] of the current step
                double dot = fma(sc_c.y, bo_c, gx_l[0]);
                // step t - 1, finished but not yet written / applied
                double m_p = 0.0, gam_p = 0.0, bo_p = 0.0;
                // this group's entries of the Gram columns of the steps t (even / odd): column t is
                // requested in step t and applied, with delta_t, during step t + 1
                double gc0[NI], gc1[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) gc0[i] = gc1[i] = 0.0;
                // The compiler's scheduler puts the whole dependent chain first and everything
                // else behind it (on the path); so the step is written as the chain's instructions
                // one by one, each fenced together with the off-path work that fills its latency.
#define LOCUS_FENCE() __builtin_amdgcn_sched_barrier(0)
                LOCUS_FENCE();
                sfor<kBlk>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    double (&gc)[NI] = (t & 1) ? gc1 : gc0;    // column t
                    double (&gp)[NI] = (t & 1) ? gc0 : gc1;    // column t - 1
                    // column s updates corr[t2] for t2 >= s + 2 (t2 = s + 1 goes through `pre`):
                    // entries i >= (s + 2) / NQ of a lane (what lies below that in entry i is dead
                    // or zero: harmless)
                    constexpr int ILO = (t + 2) / NQ;          // first entry of column t that is used
                    constexpr int IPL = (t + 1) / NQ;          // ... of column t - 1
                    double delta = 0.0, m = 0.0, bo_n = 0.0, gx_n = 0.0, base = 0.0, pre = 0.0, gm = 0.0;
                    double sd_n = 0.0, cnx = 0.0;
                    double2 sc_n = make_double2(0.0, 0.0);
                    using std::integral_constant;
                    const double lod = sc_c.x;
                    const double cB = fma(cfac, lod, cC);

                    const double dd = dot * dot;
                    if constexpr (t + 1 < kBlk) {   // requests for step t + 1
#pragma unroll
                        for (int i = ILO; i < NI; ++i) gc[i] = cg_l[t * 16 + i];
                        bo_n = bo_l[(t + 1) * TW];
                        gx_n = gx_l[(t + 1) * TW];
                        sc_n = *reinterpret_cast<const double2*>(sc_b + 4 * (t + 1));
                        sd_n = sc_b[4 * (t + 1) + 2];
                    }
                    LOCUS_FENCE();
                    const double a = fma(-cA, dd, cB);                         // coreLoop.cpp:42-46
                    if constexpr (t > 0) delta = fma(-m_p, gam_p, bo_p);       // :48, beta_old - beta_new
                    LOCUS_FENCE();
                    // gam = 1 / (1 + exp(a)), a clamped to where neither exp nor the reciprocal leaves
                    // the normal range (gam = 1 below -708, 2e-306 instead of less above 705).  The
                    // clamp is v_max / v_min, which drop a NaN: `one`, the numerator, carries it
                    // instead (a NaN log-odds must come out as a NaN gam, src/utils.cpp:3-12).
                    double x = __builtin_fmax(a, -708.0);
                    // delta of step t - 1 into the corrections of the steps >= t + 1, NQ of them per FMA
                    if constexpr (t > 0) {
#pragma unroll
                        for (int i = IPL; i < NI; ++i) corr[i] = fma(gp[i], delta, corr[i]);
                    }
                    m = c_scale * dot;
                    LOCUS_FENCE();
                    x = __builtin_fmin(x, 705.0);
                    const double one = (a == a) ? 1.0 : a;
                    // corr[t + 1], complete now, from the group that owns it to every lane
                    if constexpr (t > 0 && t + 1 < kBlk) {
                        constexpr int QS = (t + 1) % NQ, IS = (t + 1) / NQ;
                        if constexpr (NQ == 1) {
                            cnx = corr[IS];
                        } else {
                            const int src = (QS * TW + rr) * 4;
                            const int lo_ = __builtin_amdgcn_ds_bpermute(src, __double2loint(corr[IS]));
                            const int hi_ = __builtin_amdgcn_ds_bpermute(src, __double2hiint(corr[IS]));
                            cnx = __hiloint2double(hi_, lo_);
                        }
                    }
                    LOCUS_FENCE();
                    // N = round(64 x / ln 2) sits in the low word of x (64 / ln 2) + 1.5 2^52
                    const double kf0 = fma(x, 92.33248261689366, 6755399441055744.0);
                    if constexpr (t > 0) {
                        dl_w[(t - 1) * TW] = delta;
                        mg_w[(t - 1) * TW] = m_p;
                        mg_w[TB + (t - 1) * TW] = gam_p;
                        if constexpr (CS == 1 && (t & 3) == 0) lds_publish(&prog, 16 * b + t);
                    }
                    LOCUS_FENCE();
                    const double nf = kf0 - 6755399441055744.0;
                    const int nn = __double2loint(kf0);
                    const double tj = etab[nn & 63];                            // 2^(j/64)
                    LOCUS_FENCE();
                    double r = fma(nf, -6.93147180369123816490e-01 / 64, x);
                    const int kx = nn >> 6;
                    LOCUS_FENCE();
                    r = fma(nf, -1.90821492927058770002e-10 / 64, r);
                    LOCUS_FENCE();
                    const double r2 = r * r;
                    const double pa = fma(r, 1.66666666666666666667e-01, 0.5);
                    const double pb = fma(r, 8.33333333333333333333e-03, 4.16666666666666666667e-02);
                    LOCUS_FENCE();
                    const double pc = fma(pb, r2, pa);
                    LOCUS_FENCE();
                    const double pd_ = fma(pc, r2, r);                          // exp(r) - 1
                    LOCUS_FENCE();
                    const double pe = fma(tj, pd_, tj);
                    if constexpr (t + 1 < kBlk) base = fma(sc_n.y, bo_n, gx_n);
                    LOCUS_FENCE();
                    const double e = __builtin_ldexp(pe, kx);
                    LOCUS_FENCE();
                    const double den = 1.0 + e;
                    if constexpr (t + 1 < kBlk) {
                        pre = fma(sd_c, bo_c, cnx);
                        gm = sd_c * m;
                    }
                    LOCUS_FENCE();
                    // the hardware reciprocal (2^-24.4), ONE Newton step and the quotient's own
                    // correction: 0.5 ulp (scripts/microbench/div_accuracy.hip)
                    double rcp = __builtin_amdgcn_rcp(den);
                    if constexpr (t + 1 < kBlk) pre += base;
                    LOCUS_FENCE();
                    double tt = fma(-den, rcp, 1.0);
                    LOCUS_FENCE();
                    rcp = fma(tt, rcp, rcp);
                    LOCUS_FENCE();
                    const double qv = one * rcp;
                    LOCUS_FENCE();
                    tt = fma(-den, qv, one);
                    LOCUS_FENCE();
                    const double gmv = fma(tt, rcp, qv);
                    LOCUS_FENCE();
                    dot = fma(-gm, gmv, pre);                                  // dot of step t + 1
                    m_p = m;
                    gam_p = gmv;
                    bo_p = bo_c;
                    bo_c = bo_n;
                    sc_c = sc_n;
                    sd_c = sd_n;
                    LOCUS_FENCE();
                });
#undef LOCUS_FENCE
                {
                    const double delta = fma(-m_p, gam_p, bo_p);
                    dl_w[15 * TW] = delta;
                    mg_w[15 * TW] = m_p;
                    mg_w[TB + 15 * TW] = gam_p;
                    if constexpr (CS == 1) lds_publish(&prog, 16 * b + 16);
                }
            }
            M1STAMP(1);
            if constexpr (CS > 1) {
                // Cross-block corrections of the block this wave runs NEXT (all 64 lanes):
                //   gx = reduced G (helper) + sum_{l >= 2} (X_bx' X_{bx-l}) d_{bx-l} (matrix waves 1 ..
                //        L-2, published long ago) + (X_bx' X_{bx-1}) d_{bx-1} (just produced here).
                // On matrix wave 0, where this used to live, its ~60 VALU instructions crawled
                // behind the fp64 MFMA stream of the other wave of the SIMD (an MFMA holds the VALU
                // for 64 cycles) and ended every iteration ~2000 cycles after the chain; SIMD 3 runs
                // no such stream, and nobody has to follow the chain's progress any more.
                const int bx = b + 1;
                if (bx >= 0 && bx < P.nblk) {
                    int fl[L - 1];
                    fl[0] = lds_peek(&gfin_ready);
#pragma unroll
                    for (int l = 2; l < L; ++l) fl[l - 1] = lds_peek(&cxf[l - 2]);
                    asm volatile("" ::: "memory");  // flags first: the LDS serves a wave in order
                    const double2* ap = reinterpret_cast<const double2*>(axl + (bx & 1) * 256) + 2 * lane;
                    const double2 a01 = ap[0], a23 = ap[1];
                    const double* dp = dl + ((bx - 1) & (DR - 1)) * TB + q * TW + r;
                    double b4[4][T], gq[T][4], cq[L - 2][T][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int tau = 0; tau < T; ++tau) b4[c][tau] = dp[4 * TW * c + 16 * tau];
#pragma unroll
                    for (int tau = 0; tau < T; ++tau)
#pragma unroll
                        for (int g = 0; g < 4; ++g) gq[tau][g] = gfin[256 * tau + g * 64 + lane];
#pragma unroll
                    for (int l = 2; l < L; ++l)
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                cq[l - 2][tau][g] = cx[(l - 2) * TB + 256 * tau + g * 64 + lane];
                    asm volatile("" ::: "memory");
                    bool ok = fl[0] >= it;
#pragma unroll
                    for (int l = 2; l < L; ++l) ok = ok && (fl[l - 1] >= it);
                    if (!ok) {  // something was late: wait for it, read again
                        for (int spin = 0; lds_peek(&gfin_ready) < it && spin < (1 << 22); ++spin)
                            __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int l = 2; l < L; ++l)
                            for (int spin = 0; lds_peek(&cxf[l - 2]) < it && spin < (1 << 22); ++spin)
                                __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
#pragma unroll
                            for (int g = 0; g < 4; ++g) gq[tau][g] = gfin[256 * tau + g * 64 + lane];
#pragma unroll
                        for (int l = 2; l < L; ++l)
#pragma unroll
                            for (int tau = 0; tau < T; ++tau)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    cq[l - 2][tau][g] = cx[(l - 2) * TB + 256 * tau + g * 64 + lane];
                    }
                    d4 xc[T];
#pragma unroll
                    for (int tau = 0; tau < T; ++tau) {
                        xc[tau] = d4{gq[tau][0], gq[tau][1], gq[tau][2], gq[tau][3]};
#pragma unroll
                        for (int l = 2; l < L; ++l)   // fixed order: identical sums in every member
#pragma unroll
                            for (int g = 0; g < 4; ++g) xc[tau][g] += cq[l - 2][tau][g];
                    }
                    const double axv[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
                            xc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(axv[c], b4[c][tau], xc[tau],
                                                                           0, 0, 0);
                    // D layout (t = 4 g + q, response 16 tau + r) -> gx[t][k]
                    double* go = gx + (bx & 1) * 2 * TB + q * TW + r;
#pragma unroll
                    for (int tau = 0; tau < T; ++tau)
#pragma unroll
                        for (int g = 0; g < 4; ++g) go[4 * TW * g + 16 * tau] = xc[tau][g];
                }
            }
            M1STAMP(3);
        }
        M1STAMP_FLUSH(2);
        __syncthreads();  // E1
        __syncthreads();  // E2
    } else {
        // =========================== helper role ==========================================
        // This wave shares SIMD 3 with the chain wave, whose ~72 VALU instructions per step keep the
        // vector ALU 80 % busy; the older wave wins the issue arbitration, so the helper's ~300
        // VALU instructions per block crawled behind it (7900 cycles per block at CS = 3, where
        // the helper is the last to reach the barrier).  With three members it goes first from
        // the barrier until its requests are issued (reduced G, flags, loads), then steps back:
        // 10.9 -> 9.84 ms at n = 1000, d_loc = 1250.  CS = 1, 2 are bound elsewhere and lose
        // when the chain is delayed (n = 200, CS = 1: 3.31 -> 3.6 ms), so they keep the default.
        if constexpr (CS == 3) __builtin_amdgcn_s_setprio(2);
        // lane (q, r) serves response 16 tau + r of the cluster, tau = 0 .. T-1
        int kh[T];
        bool kvalid[T];
        double c_s2[T], acc_g[T], acc_m2[T], acc_b2[T];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
            kh[tau] = k0 + 16 * tau + r;
            kvalid[tau] = kh[tau] < kend;
            c_s2[tau] = kvalid[tau] ? P.s2k[kh[tau]] : 0.0;
            acc_g[tau] = acc_m2[tau] = acc_b2[tau] = 0.0;
        }
        // Member 0 leads: it adds up the members' column statistics at the end and writes them.
        // The rows of gam / mu / beta are written by the members in turn, block bs by member
        // bs % CS (every member runs the same chain and holds the same results): on one member
        // these stores -- 24 T scattered stores and ~150 instructions per block, issued in the
        // gaps the chain wave leaves -- made its helper the slowest wave of the whole cluster
        // (4080 of 10 290 cycles per block at T = 2, CS = 3, the partners waiting for its granules).
        const bool leader = (c_mem == 0);
        unsigned long long* xb_tile = nullptr;
        int* xf_tile = nullptr;
        if constexpr (CS > 1) {
            xb_tile = P.xbuf + (size_t)g_cl * xbuf_tile_words(CS, T);
            xf_tile = P.xflag + (size_t)g_cl * CS * 32;
        }
        constexpr int NGV = (CS > 1) ? CS * 4 * T : 1;   // granules per lane and block

        // CS > 1: the members' partial tiles of the block whose reduced G is due in the NEXT
        // iteration, as 16-byte granules {tag, low half, high half, tag} of one double each:
        // fetched one iteration ahead, validated (and in the rare case of a late partner
        // re-fetched) when they are due.  A granule is one 16-byte sc1 store and one 16-byte sc1
        // load; the tag on BOTH sides of the data means that a granule seen half old, half new
        // (8-byte tearing; not observed on gfx950) fails the check like a stale one.  Round 2's
        // 8-byte granules {tag, half a double} cost this wave two loads, two compares and three
        // integer instructions of re-assembly per double -- on the SIMD it shares with the chain.
        // Addresses: the cluster's exchange area is a buffer resource, the lane offset a
        // loop-invariant register, everything else scalar (no vector address arithmetic).
        // (index (cm * T + tau) * 4 + g: member cm, tile tau, accumulator register g of the lane)
        xgran gv[NGV];
#pragma unroll
        for (int e = 0; e < NGV; ++e) gv[e] = xgran{0u, 0u, 0u, 0u};
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)xb_tile, 0, 0x7fffffff, 0x00020000 /* 32-bit data format, gfx9 */);
        const int xvoff = lane * 16;
        constexpr int kXBlockBytes = CS * T * 4 * 1024;   // one ring slot: CS members x T tiles x 4 x 64 granules
        auto fetch_granules = [&](int blk) {
            const int sbase = (blk % kXRing) * kXBlockBytes;
#pragma unroll
            for (int e = 0; e < NGV; ++e)
                gv[e] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff + (e & 3) * 1024,
                                                              sbase + (e >> 2) * 4096, kXAuxSc1);
        };

        for (int it = 0; it < nit; ++it) {
            __syncthreads();  // B1
            if constexpr (CS > 1) {
                const int dd_ = lds_peek(&dead);
                if (dd_ != 0 && it >= dd_) break;
            }
            if constexpr (CS == 3) __builtin_amdgcn_s_setprio(2);  // see the top of this role
            M1STAMP(0);
            const int bn_blk = it - L + 2;   // block the chain runs in the next iteration
            const int bs = it - L;           // block the chain finished in the last iteration
            double pg[T][4];
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
#pragma unroll
                for (int g = 0; g < 4; ++g) pg[tau][g] = 0.0;
            // ---- reduced G of block bn_blk ---------------------------------------------------------
            if constexpr (CS == 1) {
                if (it >= 1) {   // (T == 1)
                    const double* pp = part + (((it - 1) & 1) * MW) * 256 + lane;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double* p2 = pp + g * 64;
                        pg[0][g] = ((p2[0] + p2[256]) + (p2[512] + p2[768])) + (p2[1024] + p2[1280]);
                        gfin[g * 64 + lane] = pg[0][g];
                    }
                }
            } else {
                double sg[T][4];
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
#pragma unroll
                    for (int g = 0; g < 4; ++g) sg[tau][g] = 0.0;
                if (bn_blk >= 0 && bn_blk < P.nblk) {
                    const unsigned tag = (unsigned)bn_blk + 1u;
                    // The granules were fetched one iteration ago and the wait of that iteration's
                    // hand-over covered them: the check below must not wait for anything.  (Written
                    // as one loop -- check, break, sleep, refetch -- the compiler put a vmcnt(0) in
                    // front of the FIRST check as well, because the registers may come from the
                    // refetch: ~1800 cycles per block waiting for the previous block's result stores.)
                    auto tags_ok = [&]() {
                        // every tag equals `tag` <=> their minimum and maximum do.  One v_min3 and one
                        // v_max3 per granule, vector registers only: as compares, each of them wrote
                        // a scalar register pair that a scalar AND then waited for -- ~85 cycles per
                        // granule of dependent latency on this wave (2060 per block at T = 2, CS = 3).
                        unsigned mn[2] = {tag, tag}, mx[2] = {tag, tag};
#pragma unroll
                        for (int e = 0; e < NGV; ++e) {
                            const unsigned a_ = gv[e].x, b_ = gv[e].w;
                            const unsigned lo_ = a_ < b_ ? a_ : b_, hi_ = a_ < b_ ? b_ : a_;
                            mn[e & 1] = lo_ < mn[e & 1] ? lo_ : mn[e & 1];
                            mx[e & 1] = hi_ > mx[e & 1] ? hi_ : mx[e & 1];
                        }
                        const bool bad = mn[0] != tag || mn[1] != tag || mx[0] != tag || mx[1] != tag;
                        return __builtin_amdgcn_ballot_w64(bad) == 0ull;
                    };
                    if (!tags_ok()) {  // a partner is late: poll
                        for (int spin = 0;; ++spin) {
                            if (spin > (1 << 19)) {  // ~0.5 s: the partner is not running
                                if (lane == 0) atomicOr(P.err, 4);
                                if (lds_peek(&dead) == 0) lds_publish(&dead, it + 2);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(4);
#ifdef LOCUS_STAMPS
                            tacc[7] += 1;  // refetches
#endif
                            asm volatile("" ::: "memory");   // a fresh fetch every pass
                            fetch_granules(bn_blk);
                            // pin: the wait for these loads is issued HERE, on the slow path, and
                            // the registers leave it with nothing pending
#pragma unroll
                            for (int e = 0; e < NGV; ++e) asm volatile("" : "+v"(gv[e]));
                            if (tags_ok()) break;
                        }
                    }
                    M1STAMP(3);
#pragma unroll
                    for (int cm = 0; cm < CS; ++cm)   // member order: identical sums in every member
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const xgran v = gv[(cm * T + tau) * 4 + g];
                                const double dv = __hiloint2double((int)v.z, (int)v.y);
                                sg[tau][g] = (cm == 0) ? dv : sg[tau][g] + dv;
                            }
                }
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
#pragma unroll
                    for (int g = 0; g < 4; ++g) gfin[256 * tau + g * 64 + lane] = sg[tau][g];
            }
            M1STAMP(1);
            lds_publish(&gfin_ready, it);

            // ---- requests: everything this iteration needs from global memory, at once -----------
            double bo_v[T][4];
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
#pragma unroll
                for (int rd = 0; rd < 4; ++rd) bo_v[tau][rd] = 0.0;
            double2 gr_v[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
            double2 sc_v = make_double2(0.0, 0.0);
            int jv = 0;
            const int jn_blk = bn_blk + 1;   // its covariates are staged one more iteration ahead
            if (lane < 16 && jn_blk >= 0 && jn_blk < P.nblk) jv = P.jidx[16 * jn_blk + lane];
            double2 ax_v[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
            if constexpr (CS > 1) {  // and the A operands of its lag-1 correction (the chain wave's)
                if (jn_blk >= 0 && jn_blk < P.nblk) {
                    const double2* ap =
                        reinterpret_cast<const double2*>(P.ax + (size_t)jn_blk * 256 + 4 * lane);
                    ax_v[0] = ap[0];
                    ax_v[1] = ap[1];
                }
            }
            if (bn_blk >= 0 && bn_blk < P.nblk) {
                const double2* gsrc =
                    reinterpret_cast<const double2*>(P.cg + (size_t)256 * bn_blk) + 2 * lane;
                gr_v[0] = gsrc[0];
                gr_v[1] = gsrc[1];
                if (lane < 32)
                    sc_v = reinterpret_cast<const double2*>(P.sc2 + (size_t)64 * bn_blk)[lane];
                {
                    const int* jp = jl + (bn_blk & 7) * 16 + q;
#pragma unroll
                    for (int rd = 0; rd < 4; ++rd)
                        if (16 * bn_blk + 4 * rd + q < P.nsteps) {
                            const double* row = P.state + 2 * pd + (size_t)jp[4 * rd] * d;
#pragma unroll
                            for (int tau = 0; tau < T; ++tau)
                                if (kvalid[tau]) bo_v[tau][rd] = row[kh[tau]];
                        }
                }
            }
            if constexpr (CS > 1) {
                // ---- granules of the block due next iteration (published a whole iteration ago) ---
                const int bg = bn_blk + 1;
                if (bg >= 0 && bg < P.nblk) fetch_granules(bg);
                // ---- this member's partial tile of block it - 1: reduce (published below) --------
                if (it >= 1 && it - 1 < P.nblk) {
                    const double* pp = part + (((it - 1) & 1) * MW) * TB + lane;
#pragma unroll
                    for (int tau = 0; tau < T; ++tau)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const double* p2 = pp + 256 * tau + g * 64;
                            pg[tau][g] = ((p2[0] + p2[TB]) + (p2[2 * TB] + p2[3 * TB])) +
                                         (p2[4 * TB] + p2[5 * TB]);
                        }
                }
            }
            // everything the chain and the memory system wait for is out: back behind the chain
            // (hand-over and result stores have until the barrier; 10.08 -> 9.84 ms)
            if constexpr (CS == 3) __builtin_amdgcn_s_setprio(0);
            M1STAMP(4);
            // ---- hand over what was requested: the ONE wait for global memory of this iteration.
            // Every store of the iteration comes after it (the memory counter is in order: a wait
            // for these loads behind the stores would also wait for the stores' write-through).
            __builtin_amdgcn_sched_barrier(0);
            if (bn_blk >= 0 && bn_blk < P.nblk) {
                double* bp = bol + (bn_blk & 1) * 2 * TB + q * TW + r;
#pragma unroll
                for (int rd = 0; rd < 4; ++rd)
#pragma unroll
                    for (int tau = 0; tau < T; ++tau) bp[4 * TW * rd + 16 * tau] = bo_v[tau][rd];
                double2* gp = reinterpret_cast<double2*>(gram + (bn_blk & 1) * 256) + 2 * lane;
                gp[0] = gr_v[0];
                gp[1] = gr_v[1];
                if (lane < 32) reinterpret_cast<double2*>(scl + (bn_blk & 1) * 64)[lane] = sc_v;
            }
            if (lane < 16) jl[(jn_blk & 7) * 16 + lane] = jv;
            if constexpr (CS > 1) {
                if (jn_blk >= 0 && jn_blk < P.nblk) {
                    double2* ap = reinterpret_cast<double2*>(axl + (jn_blk & 1) * 256) + 2 * lane;
                    ap[0] = ax_v[0];
                    ap[1] = ax_v[1];
                }
                // the granules just fetched are complete as well: pin them before the stores
#pragma unroll
                for (int e = 0; e < NGV; ++e) asm volatile("" : "+v"(gv[e]));
            }
            __builtin_amdgcn_sched_barrier(0);
            M1STAMP(5);
            if constexpr (CS > 1) {
                // ---- publish this member's partial tile of block it - 1 ---------------------------
                if (it >= 1 && it - 1 < P.nblk) {
                    const unsigned tagw = (unsigned)it;  // block + 1
                    const int sbase = ((it - 1) % kXRing) * kXBlockBytes + c_mem * T * 4096;
#pragma unroll
                    for (int tau = 0; tau < T; ++tau)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const xgran v = xgran{tagw, (unsigned)__double2loint(pg[tau][g]),
                                                  (unsigned)__double2hiint(pg[tau][g]), tagw};
                            __builtin_amdgcn_raw_buffer_store_b128(v, xrs, xvoff + g * 1024,
                                                                   sbase + tau * 4096, kXAuxSc1);
                            // A 16-byte store reads its data registers AFTER it has issued, and the
                            // compiler reuses them for the next granule in the very next
                            // instruction (it counts on no hazard when the buffer offset is a
                            // scalar register): the granule then carried the NEXT granule's halves
                            // under valid tags.  Found with a checksum in place of the second tag
                            // (no granule ever validated); two wait states after every such store.
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("s_nop 1");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
            // ---- results of the block the chain finished last iteration -----------------------
            if (bs >= 0 && bs < P.nblk && (CS == 1 || (bs % CS == c_mem && lds_peek(&dead) == 0))) {
                const double* mp = mg + (bs & 1) * 2 * TB + q * TW + r;
                const int* jp = jl + (bs & 7) * 16 + q;
#pragma unroll
                for (int rd = 0; rd < 4; ++rd) {
                    const int s = 16 * bs + 4 * rd + q;
                    if (s < P.nsteps) {
                        double* orow = P.state + (size_t)jp[4 * rd] * d;
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
                            if (kvalid[tau]) {
                                const double mv = mp[4 * TW * rd + 16 * tau];
                                const double gv2 = mp[TB + 4 * TW * rd + 16 * tau];
                                const double bv = mv * gv2;
                                double* o = orow + kh[tau];
                                o[0] = gv2;
                                o[pd] = mv;
                                o[2 * pd] = bv;
                                acc_g[tau] += gv2;
                                acc_m2[tau] = fma(gv2, fma(mv, mv, c_s2[tau]), acc_m2[tau]);
                                acc_b2[tau] = fma(bv, bv, acc_b2[tau]);
                            }
                    }
                }
            }
            M1STAMP(2);
        }
        M1STAMP_FLUSH(3);
        __syncthreads();  // E1
        const bool alive = (CS == 1) || lds_peek(&dead) == 0;
        // column statistics: sum the four step classes of a response through LDS (gfin is free)
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
            gfin[256 * tau + lane] = acc_g[tau];
            gfin[256 * tau + 64 + lane] = acc_m2[tau];
            gfin[256 * tau + 128 + lane] = acc_b2[tau];
        }
        __syncthreads();  // E2: the matrix waves' sums of squares are in `part`
        // per-response totals of this workgroup: [0] sum of squares of its residual rows, [1..3] the
        // column statistics of the blocks it stored (four step classes per response, summed here)
        double tot[T][4];
#pragma unroll
        for (int tau = 0; tau < T; ++tau) {
#pragma unroll
            for (int v = 0; v < 4; ++v) tot[tau][v] = 0.0;
            if (lane < 16) {
#pragma unroll
                for (int ss = 0; ss < MW; ++ss) {
                    const double* pp = part + (ss * T + tau) * 64 + lane;
                    tot[tau][0] += (pp[0] + pp[16]) + (pp[32] + pp[48]);
                }
                const double* sp = gfin + 256 * tau + lane;
#pragma unroll
                for (int v = 0; v < 3; ++v)
                    tot[tau][v + 1] = (sp[64 * v] + sp[64 * v + 16]) + (sp[64 * v + 32] + sp[64 * v + 48]);
            }
        }
        if (CS > 1 && alive) {
            // the members' totals, added in member order by the leader
            double* rs_base = reinterpret_cast<double*>(xb_tile + (size_t)kXRing * CS * T * 512);
            if (lane < 16) {
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        __hip_atomic_store(rs_base + (c_mem * T + tau) * 64 + 16 * v + lane, tot[tau][v],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(xf_tile + c_mem * 32, nit + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (leader) {
#pragma unroll
                for (int tau = 0; tau < T; ++tau)
#pragma unroll
                    for (int v = 0; v < 4; ++v) tot[tau][v] = 0.0;
#pragma unroll
                for (int cm = 0; cm < CS; ++cm) {
                    int spin = 0;
                    while (__hip_atomic_load(xf_tile + cm * 32, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT) < nit + 1) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spin > (1 << 21)) {
                            if (lane == 0) atomicOr(P.err, 4);
                            break;
                        }
                    }
                    if (lane < 16) {
#pragma unroll
                        for (int tau = 0; tau < T; ++tau)
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                tot[tau][v] += __hip_atomic_load(
                                    rs_base + (cm * T + tau) * 64 + 16 * v + lane, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        if (alive && P.st != nullptr && leader && lane < 16) {
#pragma unroll
            for (int tau = 0; tau < T; ++tau)
                if (kvalid[tau]) {
                    const int k = kh[tau];
                    P.st[k] = tot[tau][1];
                    P.st[(size_t)d + k] = tot[tau][2];
                    P.st[2 * (size_t)d + k] = tot[tau][3];
                    P.st[3 * (size_t)d + k] = tot[tau][0];
                }
        }
    }
}

}  // namespace locus
