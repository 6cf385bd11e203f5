// Host side of the matrix-core sweep: per-sweep block packing of X and launch dispatch.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "mfma_sweep1_kernel.h"
#include "mfma_sweep_kernel.h"

namespace locus {

int launch_mfma_sweep_t2(const MfmaSweepParams&, int, dim3, dim3, size_t, hipStream_t);
int launch_mfma_sweep1_c1(const Mfma1Params&, int KT, int odd, dim3 grid, hipStream_t);
int launch_mfma_sweep1_c2(const Mfma1Params&, int KT, int odd, dim3 grid, hipStream_t);
int launch_mfma_sweep1_c3(const Mfma1Params&, int KT, int odd, dim3 grid, hipStream_t);
int launch_mfma_sweep1_c2t2(const Mfma1Params&, int KT, int odd, dim3 grid, hipStream_t);
int launch_mfma_sweep1_c3t2(const Mfma1Params&, int KT, int odd, dim3 grid, hipStream_t);

// ---------------------------------------------------------------------------
// Block packing for mfma_sweep1_kernel (look-ahead of L blocks, NW = CS * 6 row groups).  One
// workgroup per iteration slot b = 0 .. nblk + L - 1.  As pack_blocks_kernel, plus: the update
// operands of block b go to iteration b + L; (lod, xtx) pairs and the Gram columns are laid out
// for scalar loads; the cross-block Gram matrices X_b' X_{b-l}, l = 1 .. L-1, are written as the
// four A operands (per lane) of a 16x16x16 product at ax[l - 1][b].
// ---------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256)
pack_blocks1_kernel(const double* __restrict__ Xp, const StepRec* __restrict__ steps,
                    double* __restrict__ XS, double* __restrict__ cg, double* __restrict__ ax,
                    double* __restrict__ sc2, int* __restrict__ jidx, int ld, int KT, int nblk, int NW,
                    int nsteps, int ODD, int NQ)
{
    extern __shared__ double pk_smem[];
    double (*xt)[130] = reinterpret_cast<double (*)[130]>(pk_smem);               // block b
    double (*xl)[130] = reinterpret_cast<double (*)[130]>(pk_smem + 16 * 130);    // blocks b-1 .. b-L+1
    int* js = reinterpret_cast<int*>(pk_smem + (size_t)L * 16 * 130);             // [L][16]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nit = nblk + L;
    const bool has = b < nblk;
    if (tid < 16 * L) {
        const int l = tid >> 4, t = tid & 15, bb = b - l;
        // steps beyond the sweep (padding of the last block) get a zero column: every operand of
        // their update is then 0 and so is their delta (the chain has no "inactive" select)
        js[tid] = (bb >= 0 && 16 * bb + t < nsteps) ? steps[16 * bb + t].j : -1;
    }
    if (tid < 16) {
        const bool act = 16 * b + tid < nsteps;
        const StepRec st = act ? steps[16 * b + tid] : StepRec{0.0, 0.0, 0, 0};
        jidx[16 * b + tid] = st.j;
        sc2[(size_t)(16 * b + tid) * 4] = st.lod;
        sc2[(size_t)(16 * b + tid) * 4 + 1] = st.xtx;
        sc2[(size_t)(16 * b + tid) * 4 + 3] = 0.0;
    }
    // rows are dealt to the matrix waves in 16-row tiles: per workgroup, waves 0..2 take KT tiles
    // each and waves 3..5 take KT - ODD (mfma_sweep1_kernel.h); the waves' operand streams follow
    // one another, 8 operands of 64 doubles per tile and iteration
    const int TPM = mfma1_member_tiles(KT, ODD);
    const int ntile = (NW / 6) * TPM;
    const int rows_pad = ntile * 16;
    const int gt = tid >> 4, gs = tid & 15;
    // Gram tiles X_b' X_{b-l}, l = 0 .. L-1, on the matrix cores: wave w contracts rows
    // 32 w .. 32 w + 31 of every 128-row chunk (8 k-steps of v_mfma_f64_16x16x4_f64 per lag),
    // the four partial tiles are added through LDS at the end
    typedef double pk_d4 __attribute__((ext_vector_type(4)));
    const int wv4 = tid >> 6, ln4 = tid & 63;
    pk_d4 macc[L];
#pragma unroll
    for (int l = 0; l < L; ++l) macc[l] = (pk_d4){0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    for (int c0 = 0; c0 < rows_pad; c0 += 128) {
        // all 8 L loads of this chunk are requested before the first one is used: a workgroup
        // alone on its CU (83 KB of LDS at L = 5) must not pay one memory latency per load
        double v[L][8];
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                const int col = e >> 7, rr = e & 127, row = c0 + rr;
                const int j = js[16 * l + col];
                v[l][i] = (j >= 0 && row < ld) ? Xp[(size_t)j * ld + row] : 0.0;
            }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                const int col = e >> 7, rr = e & 127;
                if (l == 0) xt[col][rr] = v[l][i];
                else xl[16 * (l - 1) + col][rr] = v[l][i];
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = tid + 256 * i;
            const int ln = o & 63;
            const int sub = (o >> 6) & 3;
            const int gl = o >> 8;
            const int gtile = (c0 >> 4) + gl;
            if (gtile < ntile) {
                const int mem = gtile / TPM, tm = gtile - mem * TPM;
                const bool lo_w = tm >= 3 * KT;                   // one of the waves 3..5
                const int ktw = lo_w ? KT - ODD : KT;
                const int mw = lo_w ? 3 + (tm - 3 * KT) / ktw : tm / ktw;
                const int tile0 = mem * TPM + mfma1_wave_tile0(mw, KT, ODD);   // first tile of the wave
                const int ti = gtile - tile0;
                const int NOP = 8 * ktw;
                double* xw = XS + (size_t)tile0 * nit * 512;      // the wave's stream
                const int oi = op_index(ti, sub, ktw);
                const double vd = xt[ln & 15][16 * gl + 4 * sub + (ln >> 4)];
                const double vu = xt[4 * sub + (ln >> 4)][16 * gl + (ln & 15)];
                {
                    const int ot = 4 * ktw + oi;
                    xw[((size_t)b * NOP + (ot & ~1)) * 64 + ln * 2 + (ot & 1)] = vd;
                }
                if (has) xw[((size_t)(b + L) * NOP + (oi & ~1)) * 64 + ln * 2 + (oi & 1)] = vu;
                if (b < L) xw[((size_t)b * NOP + (oi & ~1)) * 64 + ln * 2 + (oi & 1)] = 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // A[i = t][k = row] = X_b[row][t], B[k = row][j = s] = X_{b-l}[row][s]: both operands
            // are read as [column = lane & 15][row = 4 (8 w + q) + (lane >> 4)]
            const int rr = 4 * (8 * wv4 + q) + (ln4 >> 4);
            const double a = xt[ln4 & 15][rr];
            macc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, macc[0], 0, 0, 0);
#pragma unroll
            for (int l = 1; l < L; ++l)
                macc[l] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xl[16 * (l - 1) + (ln4 & 15)][rr],
                                                               macc[l], 0, 0, 0);
        }
    }
    // D[t = (lane >> 4) + 4 r][s = lane & 15]: partial tiles of the four waves -> LDS -> one
    // (t, s) pair per thread
    __syncthreads();
    double* red = pk_smem;  // [wave][l][r][lane], 4 L 256 doubles (the X tiles are dead)
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wv4 * L + l) * 4 + r) * 64 + ln4] = macc[l][r];
    __syncthreads();
    double gsum[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) a += red[((w * L + l) * 4 + (gt >> 2)) * 64 + (gt & 3) * 16 + gs];
        gsum[l] = a;
    }
    // Column t = gs of the block's Gram matrix, strictly below the diagonal, in the order the chain
    // wave's lanes read it (mfma_sweep1_kernel.h): its 64 lanes are NQ groups (mfma1_lane_groups: 1,
    // or 64 / (16 T)) that run the same 16 T responses, group q keeps the corrections of the steps
    // t' = NQ i + q, so entry t' sits at [t][q = t' % NQ][i = t' / NQ].  The entry right below the diagonal, which every lane
    // needs, also goes to the step's scalars {lod, xtx, g[t][t+1], 0}.
    cg[(size_t)b * 256 + gs * 16 + (gt % NQ) * (16 / NQ) + gt / NQ] = (gt > gs) ? gsum[0] : 0.0;
    if (gt == ((gs + 1) & 15)) sc2[(size_t)(16 * b + gs) * 4 + 2] = (gs < 15) ? gsum[0] : 0.0;
    // (X_b' X_{b-l})[t' = gt][s = gs]: A operand c = gs >> 2 of lane (gs & 3) * 16 + gt
#pragma unroll
    for (int l = 1; l < L; ++l)
        ax[((size_t)(l - 1) * (nit + 1) + b) * 256 + ((gs & 3) * 16 + gt) * 4 + (gs >> 2)] = gsum[l];
}

// ---------------------------------------------------------------------------
// One workgroup per iteration slot b = 0 .. nblk + 1 of the sweep.  For b < nblk it gathers the
// 16 columns of block b (128 rows at a time through LDS) and writes them in the two MFMA operand
// orders
//   dot operand (i, g):    l -> X[16 i + 4 g + (l >> 4)][j_{l & 15}]        (A of G = X_b' R)
//   update operand (i, c): l -> X[16 i + (l & 15)][j_{4 c + (l >> 4)}]      (A of R += X_b delta)
// into the per-wave streams XS[wave][it][8 KT][64] (mfma_sweep_kernel.h): the dot operands of
// block b belong to iteration it = b, its update operands to iteration it = b + 2.  Parts that no
// block fills (update of it = 0, 1; dot of it >= nblk) are zeroed.  Also: the block's strictly
// lower Gram matrix by columns, cg[b][t][t'] = X_t' . X_t (t' > t), and the cross-block Gram matrix
// X_{b+1}' X_b in A-operand order at axo[b + 2].  Rows >= n are zero (Xp is zero padded to ld).
// Bandwidth-bound: reads 2 p ld, writes 2 p rows_pad doubles per sweep.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_blocks_kernel(const double* __restrict__ Xp, const StepRec* __restrict__ steps,
                   double* __restrict__ XS, double* __restrict__ cg, double* __restrict__ axo,
                   double* __restrict__ sc4, int* __restrict__ jidx, int ld, int KT, int nblk, int MW,
                   int d)
{
    __shared__ double pk_lds[2 * 16 * 130];   // the two X tiles; reused for the Gram reduction
    double (*xt)[130] = reinterpret_cast<double (*)[130]>(pk_lds);
    double (*xn)[130] = reinterpret_cast<double (*)[130]>(pk_lds + 16 * 130);
    __shared__ int js[16], jn[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nit = nblk + 2;
    const bool has = b < nblk, has_next = b + 1 < nblk;
    if (tid < 16) {
        js[tid] = has ? steps[16 * b + tid].j : 0;
        jn[tid] = has_next ? steps[16 * (b + 1) + tid].j : 0;
        jidx[16 * b + tid] = js[tid];
        double* sc = sc4 + (size_t)(16 * b + tid) * 4;
        sc[0] = has ? steps[16 * b + tid].lod : 0.0;
        sc[1] = has ? steps[16 * b + tid].xtx : 0.0;
        sc[2] = __longlong_as_double((long long)js[tid] * d);
        sc[3] = 0.0;
    }
    const int rows_pad = MW * 16 * KT;
    const int NOP = 8 * KT;
    const int gt = tid >> 4, gs = tid & 15;
    // Gram tiles X_b' X_b and X_{b+1}' X_b on the matrix cores, as in pack_blocks1_kernel: wave w
    // contracts rows 32 w .. 32 w + 31 of every 128-row chunk, the four partial tiles are added
    // through LDS at the end (0.49 -> 0.47 ms at p = 50 000: the kernel is bound by its 8-byte
    // operand stores, two interleaved streams per 16-byte slot, not by this)
    typedef double pk_d4 __attribute__((ext_vector_type(4)));
    const int wv4 = tid >> 6, ln4 = tid & 63;
    pk_d4 m0 = (pk_d4){0.0, 0.0, 0.0, 0.0}, m1 = m0;
    for (int c0 = 0; c0 < rows_pad; c0 += 128) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            const int col = e >> 7, rr = e & 127, row = c0 + rr;
            xt[col][rr] = (has && row < ld) ? Xp[(size_t)js[col] * ld + row] : 0.0;
            xn[col][rr] = (has_next && row < ld) ? Xp[(size_t)jn[col] * ld + row] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = tid + 256 * i;
            const int ln = o & 63;
            const int sub = (o >> 6) & 3;  // k-step (dot) or covariate slice (update)
            const int gl = o >> 8;         // 16-row tile inside the chunk
            const int gtile = (c0 >> 4) + gl;
            if (gtile < MW * KT) {
                const int wv = gtile / KT, ti = gtile - wv * KT;
                const int oi = op_index(ti, sub, KT);
                const double vd = xt[ln & 15][16 * gl + 4 * sub + (ln >> 4)];
                const double vu = xt[4 * sub + (ln >> 4)][16 * gl + (ln & 15)];
                // dot part of iteration b: operands 4 KT .. 8 KT - 1
                {
                    const int ot = 4 * KT + oi;
                    XS[(((size_t)wv * nit + b) * NOP + (ot & ~1)) * 64 + ln * 2 + (ot & 1)] = vd;
                }
                // update part of iteration b + 2: operands 0 .. 4 KT - 1
                if (has) {
                    XS[(((size_t)wv * nit + b + 2) * NOP + (oi & ~1)) * 64 + ln * 2 + (oi & 1)] = vu;
                }
                if (b < 2) {  // nobody fills the update part of iterations 0 and 1
                    XS[(((size_t)wv * nit + b) * NOP + (oi & ~1)) * 64 + ln * 2 + (oi & 1)] = 0.0;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // A[i = t][k = row] and B[k = row][j = s] are both read as
            // [column = lane & 15][row = 4 (8 w + q) + (lane >> 4)]
            const int rr = 4 * (8 * wv4 + q) + (ln4 >> 4);
            const double a = xt[ln4 & 15][rr], an = xn[ln4 & 15][rr];
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, m0, 0, 0, 0);    // X_b' X_b
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(an, a, m1, 0, 0, 0);   // X_{b+1}' X_b
        }
    }
    // D[t = (lane >> 4) + 4 r][s = lane & 15]: partial tiles of the four waves -> LDS -> one
    // (t, s) pair per thread
    __syncthreads();
    double* red = pk_lds;  // [wave][product][r][lane], 2048 doubles (the X tiles are dead)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[((wv4 * 2 + 0) * 4 + r) * 64 + ln4] = m0[r];
        red[((wv4 * 2 + 1) * 4 + r) * 64 + ln4] = m1[r];
    }
    __syncthreads();
    double gacc = 0.0, xacc = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        gacc += red[((w * 2 + 0) * 4 + (gt >> 2)) * 64 + (gt & 3) * 16 + gs];
        xacc += red[((w * 2 + 1) * 4 + (gt >> 2)) * 64 + (gt & 3) * 16 + gs];
    }
    // column gs of the strictly lower triangle: cg[b][t = gs][t' = gt]
    cg[(size_t)b * 256 + gs * 16 + gt] = (gt > gs) ? gacc : 0.0;
    // cross Gram (X_{b+1}' X_b)[t' = gt][t = gs] as A operand c = gs >> 2, lane (gs & 3) * 16 + gt,
    // consumed in iteration b + 2
    if (b + 2 < nit) axo[((size_t)(b + 2) * 64 + (gs & 3) * 16 + gt) * 4 + (gs >> 2)] = xacc;
    if (b < 2) axo[((size_t)b * 64 + (gs & 3) * 16 + gt) * 4 + (gs >> 2)] = 0.0;
}

// row tiles of 16 per matrix wave for n rows: the instantiated KT values
int mfma_kt_for(int64_t n, int T)
{
#define LOCUS_KT_ITEM(k) k,
    static const int kts2[] = {LOCUS_MFMA_KTS_T2(LOCUS_KT_ITEM) 0};  // T = 2 (7 matrix waves)
    static const int kts1[] = {LOCUS_MFMA_KTS_T1(LOCUS_KT_ITEM) 0};  // T = 1 (6 matrix waves)
#undef LOCUS_KT_ITEM
    for (const int* kt = (T == 2) ? kts2 : kts1; *kt; ++kt)
        if ((int64_t)mfma_waves(T) * 16 * *kt >= n) return *kt;
    return 0;
}

// bytes of the block-packed operands for a sweep over p covariates of n rows
// 16-row tiles per matrix wave of mfma_sweep1_kernel for n rows cut over CS workgroups
// (T = tiles of 16 responses per cluster: 1, or 2 with CS = 2, 3)
// Even deal: every matrix wave takes KT tiles.  mfma1_tiling_for() below prefers the odd deal
// (waves 3..5 one tile fewer) when that covers n with fewer tiles per SIMD.
int mfma1_kt_for(int64_t n, int CS, int T = 1)
{
#define LOCUS_KT_ITEM(k) k,
    static const int kts1[] = {LOCUS_MFMA_KTS_T1(LOCUS_KT_ITEM) 0};
    static const int kts2[] = {LOCUS_MFMA1_KTS_C2(LOCUS_KT_ITEM) 0};
    static const int kts3[] = {LOCUS_MFMA1_KTS_C3(LOCUS_KT_ITEM) 0};
    static const int ktsT2[] = {LOCUS_MFMA1_KTS_T2(LOCUS_KT_ITEM) 0};
#undef LOCUS_KT_ITEM
    if (T == 2 && (CS < 2 || CS > 3)) return 0;
    if (T == 1 && (CS < 1 || CS > 3)) return 0;
    if (n > 2016) return 0;  // the matrix-core sweep's limit (mfma_kt_for)
    for (const int* kt = (T == 2) ? ktsT2 : (CS == 1) ? kts1 : (CS == 2 ? kts2 : kts3); *kt; ++kt)
        if ((int64_t)CS * 6 * 16 * *kt >= n) return *kt;
    return 0;
}

// Row tiling of mfma_sweep1_kernel for n rows: {KT, ODD}, KT = 0 when not covered.  A SIMD (waves
// m and m + 3) carries 2 KT - ODD tiles; the smallest such count that covers n wins.
struct Tiling1 { int KT, ODD; };
static Tiling1 mfma1_tiling_for(int64_t n, int CS, int T = 1)
{
    const int kt_even = mfma1_kt_for(n, CS, T);
    if (kt_even == 0) return {0, 0};
    // odd deal with the same KT: 3 CS (2 KT - 1) tiles.  Measured (n = 1000, p = 50 000, round 3): one
    // workgroup per tile 17.6 -> 16.8 ms, two 11.2 -> 10.0, three 9.67 -> 9.37; two tiles on two
    // 19.95 -> 18.4, on three 13.8 -> 12.5 (once the short waves' operand ring was cut from 12
    // pairs to 6: with 12 the odd deal lost, 15.8).
    const char* force = getenv("LOCUS_HIP_MFMA_EVEN");   // "1": always even, "0": odd wherever it fits
    const bool want_odd = force ? force[0] != '1' : true;
    if (want_odd && kt_even >= 2 && (int64_t)3 * CS * (2 * kt_even - 1) * 16 >= n) return {kt_even, 1};
    return {kt_even, 0};
}
static int tiling_rows(const Tiling1& t, int CS) { return CS * mfma1_member_tiles(t.KT, t.ODD) * 16; }

// doubles of the mfma_sweep1_kernel layout for `nblk` blocks (largest look-ahead, both CS)
static size_t mfma1_layout_doubles(int64_t n, size_t nblk, int cus)
{
    const size_t nit = nblk + 5;
    size_t rows_pad = 0;
    for (int T = 1; T <= 2; ++T)
        for (int CS = 1; CS <= 3; ++CS)
            rows_pad = std::max(rows_pad, (size_t)CS * 6 * 16 * mfma1_kt_for(n, CS, T));
    return nit * rows_pad * 32 + 2048      // XS (+ read-ahead slack)
           + nit * 256                     // cg
           + 4 * (nit + 1) * 256           // ax, lags 1..4
           + 16 * nit * 4 + 256            // sc2
           + 8 * nit + 256                 // jidx (ints)
           + (size_t)cus * (xbuf_tile_words(2, 2) / 2) + 256   // xbuf: <= cus members in flight
           + (size_t)cus * 16 + 256;       // xflag (32 ints per member)
}

size_t mfma_workspace_bytes(int64_t n, int64_t p)
{
    const size_t nit = ((size_t)p + 15) / 16 + 2;
    const int kt1 = mfma_kt_for(n, 1), kt2 = mfma_kt_for(n, 2);
    if (kt1 == 0) return 1024;  // the matrix-core sweep does not cover this n
    const size_t rows_pad = std::max((size_t)mfma_waves(1) * 16 * kt1, (size_t)mfma_waves(2) * 16 * kt2);
    const size_t old_layout = sizeof(double) * (nit * rows_pad * 32 + 2 * nit * 256 + nit * 64) +
                              sizeof(int) * 16 * nit + 64 * 1024;
    const size_t new_layout = sizeof(double) * mfma1_layout_doubles(n, ((size_t)p + 15) / 16, device_cus()) + 4096;
    return std::max(old_layout, new_layout);
}

size_t mfma_lds_bytes(int T)
{
    return sizeof(double) * (2 * (size_t)(mfma_waves(T) + 1) * T * 256 + 2 * 16 * 64 + 2 * 16 * 64 + 2 * 256 + 2 * 64 + 8);
}

// CUs of the current device as the runtime reports them (cached per device)
int real_cus()
{
    constexpr int kMaxDev = 64;
    static std::atomic<int> cache[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int cus = cache[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
        cache[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

int device_cus()
{
    // LOCUS_HIP_MFMA_CUS overrides the CU count the launch plan assumes (tests: forces the
    // multi-launch plan at small d)
    if (const char* e = getenv("LOCUS_HIP_MFMA_CUS")) {
        const int v = atoi(e);
        if (v > 0) return v;
    }
    return real_cus();
}

// CUs a row-split cluster launch may count on: the members of a cluster wait for each other, so
// all of them must be resident at once -- never more than the device really has, whatever the
// override says.
static int cluster_cus() { return std::min(device_cus(), real_cus()); }

// true when the matrix-core sweep covers n rows (identity family only)
bool mfma_sweep_covers(int n) { return mfma_kt_for(n, 1) > 0; }

// What a pack_and_launch* call does (bit mask).  A sweep whose operands were packed ahead
// (locus_hip_sweep_prepack_dev: everything that depends on X and the sweep order only) skips kPack
// and only refreshes the one thing that depends on the iteration: the log-odds of the steps.
enum : int { kPack = 1, kLaunch = 2, kRefresh = 4 };

// sc[i * stride] = steps[i].lod: the prior log-odds l1_j - lo_j of every step, in the layout the
// sweep kernels read (stride 4 in both)
__global__ void __launch_bounds__(256) refresh_lod_kernel(const StepRec* __restrict__ steps,
                                                          double* __restrict__ sc, int stride, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) sc[(size_t)i * stride] = steps[i].lod;
}

static int pack_and_launch(const SweepParams& prm, int T, int KT, int k_begin, int k_end, int what,
                           void* xb_ws, size_t xb_bytes, hipStream_t st)
{
    const int MW = mfma_waves(T);
    const int rows_pad = MW * 16 * KT;
    if (KT <= 0 || rows_pad < prm.n) {
        set_error("mfma sweep: geometry does not cover n rows");
        return LOCUS_HIP_EINVAL;
    }
    const int nblk = (prm.nsteps + 15) / 16;
    const size_t nit = (size_t)nblk + 2;
    const size_t xs = nit * rows_pad * 32;  // doubles of the packed operand streams
    if (sizeof(double) * (xs + 2 * nit * 256 + nit * 64) + sizeof(int) * 16 * nit + 64 * 1024 > xb_bytes) {
        set_error("mfma sweep: workspace too small");
        return LOCUS_HIP_EINVAL;
    }
    char* base = static_cast<char*>(xb_ws);
    base += (256 - ((uintptr_t)base & 255)) & 255;
    double* XS = reinterpret_cast<double*>(base);
    double* cgp = XS + xs + 2048;  // 16 KB of slack: the operand ring reads ahead of the stream
    double* axo = cgp + nit * 256;
    double* sc4 = axo + (nit + 1) * 256;  // one iteration of slack: staged one block ahead
    int* jidx = reinterpret_cast<int*>(sc4 + nit * 64 + 256);

    if (what & kPack) {
        hipLaunchKernelGGL(pack_blocks_kernel, dim3((unsigned)nit), dim3(256), 0, st, prm.Xp,
                           prm.steps, XS, cgp, axo, sc4, jidx, prm.ld, KT, nblk, MW, prm.d);
        LOCUS_HIP_CHECK(hipGetLastError());
    }
    if (what & kRefresh) {
        hipLaunchKernelGGL(refresh_lod_kernel, dim3((unsigned)((16 * nblk + 255) / 256)), dim3(256), 0, st,
                           prm.steps, sc4, 4, 16 * nblk);
        LOCUS_HIP_CHECK(hipGetLastError());
    }
    if (!(what & kLaunch)) return 0;

    MfmaSweepParams mp;
    mp.n = prm.n; mp.p = prm.p; mp.d = prm.d;
    mp.k_begin = k_begin; mp.k_end = k_end;
    mp.nblk = nblk; mp.nsteps = prm.nsteps;
    mp.steps = prm.steps; mp.err = prm.err;
    mp.XS = XS; mp.cg = cgp; mp.axo = axo; mp.sc4 = sc4; mp.jidx = jidx;
    mp.scale = prm.scale; mp.hinv = prm.hinv; mp.cst = prm.cst; mp.s2k = prm.s2k;
    mp.cfac = prm.cfac;
    mp.R = prm.R; mp.state = prm.state; mp.st = prm.st;

    const long n_wg = ((long)(k_end - k_begin) + 16L * T - 1) / (16L * T);
    dim3 grid((unsigned)n_wg), block((unsigned)(64 * (MW + 1)));
    const size_t lds = mfma_lds_bytes(T);
    if (T != 2) {
        set_error("mfma sweep: mfma_sweep_kernel is instantiated for T = 2 only");
        return LOCUS_HIP_EINVAL;
    }
    return launch_mfma_sweep_t2(mp, KT, grid, block, lds, st);
}

// One launch sequence of mfma_sweep1_kernel over responses [k_begin, k_end): packs (if `pack`) and
// launches clusters of CS workgroups, at most as many clusters per launch as are resident at once
// (a cluster's members wait for each other, so every member must be on a CU at the same time).
static int pack_and_launch1(const SweepParams& prm, int T, int CS, Tiling1 til, int k_begin, int k_end,
                            int what, void* xb_ws, size_t xb_bytes, hipStream_t st)
{
    const int L = mfma1_lag(CS);
    const int KT = til.KT, ODD = til.ODD;
    if (!((T == 1 && CS >= 1 && CS <= 3) || (T == 2 && CS >= 2 && CS <= 3))) {
        set_error("mfma sweep: no kernel for this (tiles per cluster, cluster size)");
        return LOCUS_HIP_EUNSUPPORTED;
    }
    const int NW = CS * 6;
    const int rows_pad = tiling_rows(til, CS);
    if (KT <= 0 || (ODD && KT < 2) || rows_pad < prm.n) {
        set_error("mfma sweep: geometry does not cover n rows");
        return LOCUS_HIP_EINVAL;
    }
    const int cus = (CS > 1) ? cluster_cus() : device_cus();
    const int nblk = (prm.nsteps + 15) / 16;
    const size_t nit = (size_t)nblk + L;
    const size_t xs = nit * rows_pad * 32;
    char* base = static_cast<char*>(xb_ws);
    base += (256 - ((uintptr_t)base & 255)) & 255;
    double* XS = reinterpret_cast<double*>(base);
    double* cgp = XS + xs + 2048;
    double* axp = cgp + nit * 256;
    double* sc2 = axp + 4 * (nit + 1) * 256;
    int* jidx = reinterpret_cast<int*>(sc2 + 16 * nit * 4 + 256);
    unsigned long long* xbuf = reinterpret_cast<unsigned long long*>(jidx) + 8 * nit + 256;
    int* xflag = reinterpret_cast<int*>(xbuf + (size_t)cus * (xbuf_tile_words(2, 2) / 2) + 256);
    const size_t used = (size_t)(reinterpret_cast<char*>(xflag) - static_cast<char*>(xb_ws)) +
                        sizeof(int) * ((size_t)cus * 32 + 64);
    if (used > xb_bytes) {
        set_error("mfma sweep: workspace too small");
        return LOCUS_HIP_EINVAL;
    }
    if (what & kPack) {
        const size_t lds = sizeof(double) * (size_t)L * 16 * 130 + sizeof(int) * 16 * L;
        if (L == 3) {
            hipLaunchKernelGGL(pack_blocks1_kernel<3>, dim3((unsigned)nit), dim3(256), lds, st, prm.Xp,
                               prm.steps, XS, cgp, axp, sc2, jidx, prm.ld, KT, nblk, NW, prm.nsteps, ODD,
                               mfma1_lane_groups(KT, CS, T, ODD));
        } else {
            // 83 KB of dynamic LDS: above the 64 KB default limit.  The attribute belongs to the
            // (function, device) pair, so it is set before every launch (cheap; as in gemm.hip).
            LOCUS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pack_blocks1_kernel<5>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(pack_blocks1_kernel<5>, dim3((unsigned)nit), dim3(256), lds, st, prm.Xp,
                               prm.steps, XS, cgp, axp, sc2, jidx, prm.ld, KT, nblk, NW, prm.nsteps, ODD,
                               mfma1_lane_groups(KT, CS, T, ODD));
        }
        LOCUS_HIP_CHECK(hipGetLastError());
    }
    if (what & kRefresh) {
        hipLaunchKernelGGL(refresh_lod_kernel, dim3((unsigned)((16 * nblk + 255) / 256)), dim3(256), 0, st,
                           prm.steps, sc2, 4, 16 * nblk);
        LOCUS_HIP_CHECK(hipGetLastError());
    }
    if (!(what & kLaunch)) return 0;
    Mfma1Params mp;
    mp.n = prm.n; mp.p = prm.p; mp.d = prm.d;
    mp.nblk = nblk; mp.nsteps = prm.nsteps;
    mp.err = const_cast<int*>(prm.err);
    mp.XS = XS; mp.cg = cgp; mp.ax = axp; mp.sc2 = sc2; mp.jidx = jidx;
    mp.scale = prm.scale; mp.hinv = prm.hinv; mp.cst = prm.cst; mp.s2k = prm.s2k;
    mp.cfac = prm.cfac;
    mp.R = prm.R; mp.state = prm.state; mp.st = prm.st;
    mp.xbuf = xbuf; mp.xflag = xflag;
    mp.spread = getenv("LOCUS_HIP_TEST_SPREAD_MEMBERS") != nullptr;
    mp.drop_member = -1;
    if (const char* e = getenv("LOCUS_HIP_TEST_DROP_MEMBER")) mp.drop_member = atoi(e);

    // clusters resident at once: one workgroup per CU, members of a cluster 8 blocks apart
    const int max_tiles = std::max(1, cus / (8 * CS)) * 8;
    if (CS > 1 && cus < 8 * CS) {
        set_error("mfma sweep: too few CUs for row-split clusters");
        return LOCUS_HIP_EUNSUPPORTED;
    }
    const int TW = 16 * T;  // responses per cluster
    for (int k = k_begin; k < k_end; k += TW * max_tiles) {
        const int ke = std::min(k_end, k + TW * max_tiles);
        mp.k_begin = k; mp.k_end = ke;
        mp.ntiles = (ke - k + TW - 1) / TW;
        if (CS > 1) {  // a granule / flag is valid when it carries this launch's tag: start from zero
            LOCUS_HIP_CHECK(hipMemsetAsync(xbuf, 0, sizeof(unsigned long long) * mp.ntiles *
                                                        xbuf_tile_words(CS, T), st));
            LOCUS_HIP_CHECK(hipMemsetAsync(xflag, 0, sizeof(int) * (size_t)mp.ntiles * CS * 32, st));
        }
        dim3 grid((unsigned)(((mp.ntiles + 7) / 8) * 8 * CS));
        int rc;
        if (T == 2)
            rc = (CS == 2) ? launch_mfma_sweep1_c2t2(mp, KT, ODD, grid, st)
                           : launch_mfma_sweep1_c3t2(mp, KT, ODD, grid, st);
        else
            rc = (CS == 1)   ? launch_mfma_sweep1_c1(mp, KT, ODD, grid, st)
                 : (CS == 2) ? launch_mfma_sweep1_c2(mp, KT, ODD, grid, st)
                             : launch_mfma_sweep1_c3(mp, KT, ODD, grid, st);
        if (rc) return rc;
    }
    return 0;
}

// Row-split cluster size for `tiles` tiles of 16 responses.  One workgroup per tile while its two
// matrix waves per SIMD have less work per block than the chain (a wave issues 8 KT MFMAs of 64
// cycles per block: KT <= 4 is <= 4096 cycles against a chain of ~5600; n = 200, p = 20 000,
// d = 1000: 3.31 ms with one workgroup, 3.66 with two).  Otherwise the rows are cut over 2 or 3
// workgroups when the members of every cluster fit on the chip at once: 3 when a 2-cluster's
// matrix waves would still be bound by their operand fetch (KT >= 5 at CS = 2, n > 768:
// n = 1000, d_loc = 1250: 10.0 ms against 10.5; C2 4.04 against 4.20), 2 below that (n = 500,
// d = 1000: 3.88 ms against 3.95).  LOCUS_HIP_MFMA_CS overrides (1, 2, 3).
static int choose_cs(int n, int tiles)
{
    int want = 0;
    if (const char* e = getenv("LOCUS_HIP_MFMA_CS")) want = atoi(e);
    if (want == 1) return 1;
    const int cus = cluster_cus();
    const int t8 = ((tiles + 7) / 8) * 8;
    if (want == 2 || want == 3) {
        const bool ok = mfma1_kt_for(n, want) > 0 && cus >= 8 * want;
        return ok ? want : 1;
    }
    const int kt1 = mfma1_kt_for(n, 1), kt2 = mfma1_kt_for(n, 2), kt3 = mfma1_kt_for(n, 3);
    if (kt1 > 0 && kt1 <= 4) return 1;
    if (kt3 > 0 && kt2 >= 5 && t8 * 3 <= cus) return 3;
    return (kt2 > 0 && t8 * 2 <= cus) ? 2 : 1;
}

// ---------------------------------------------------------------------------
// Launch plan.  A launch's duration does not depend on how many of its workgroups are busy --
// every cluster walks the same p covariates -- but on what bounds one block of 16 steps in its
// geometry.  Microseconds per block, measured on MI355X at n = 400, 700, 1000, p = 50 000 (round 3,
// scripts/shard_sweeps.sh, scripts/perf_sweep.py) as a function of kt, the 16-row tiles per matrix
// wave averaged over a SIMD (KT, or KT - 1/2 with the odd deal):
//   A  = mfma_sweep_kernel<2, KT>: 32 responses on ONE CU, 7 matrix waves, the chain shares a
//        SIMD with one of them; matrix-bound: 8.58 at KT = 9                       0.953 KT
//        or bound by that chain wave (32 responses, next to an MFMA stream): 4.85 at KT = 2,
//        5.7 at KT = 4 (4.0 + 0.43 KT) -- n = 200, d = 1000 ran here in 6.3 ms until the plan
//        knew; one tile per CU takes 3.3
//   one tile of 16 responses on 1, 2, 3 CUs: bound by the matrix waves (their operand fetch
//        through the CU's L1 and their MFMAs): 5.15 at kt = 10.5 alone (0.49 kt), 3.0 at kt = 5.5
//        in a cluster of two (0.55 kt);
//        or by the chain wave: 2.2 / 2.8 at kt = 2.5 / 4.5 alone (1.48 + 0.29 kt: matrix wave 0
//        follows it), 2.3 in a cluster of two, 2.44 / 2.66 at kt = 1.5 / 3.5 in one of three
//        (2.3 + 0.1 kt, with the cluster hand-over)
//   two tiles (32 responses) on 2, 3 CUs: matrix-bound, 5.68 at kt = 5.5 on two (1.04 kt), 3.78 at
//        kt = 3.5 on three (1.08 kt); or by the chain wave and the helper, which serve two tiles:
//        3.33 / 3.82 at kt = 1.5 / 2.5 on three, 3.55 at 2.5 on two (3.0 + 0.25 kt)
//        (on 4 CUs: measured no faster than on 3 and its helper spills: not built)
// Members of a cluster must all be resident, so a cluster launch takes at most
// floor(CUs / 8 CS) 8 clusters.  The plan minimises the sum of the launch durations over all ways
// of covering d responses (dynamic programme over units of 16 responses); every launch of a new
// geometry re-packs the block operands for its row tiling (~0.2 us per block).
// ---------------------------------------------------------------------------
struct LaunchType {
    int T, CS;      // CS = 0: mfma_sweep_kernel<2, KT> (one workgroup per 32 responses)
    int KT, ODD;
    int cap;        // responses one launch takes
    double cost;    // microseconds per block of 16 steps (model above)
};

static std::vector<LaunchType> launch_types(int n)
{
    std::vector<LaunchType> v;
    const int cus = device_cus(), ccus = cluster_cus();
    int only_cs = 0, only_t = 0;
    if (const char* e = getenv("LOCUS_HIP_MFMA_CS")) {  // one tile per cluster of that size only
        only_cs = atoi(e);
        only_t = 1;
    }
    if (const char* e = getenv("LOCUS_HIP_MFMA_PLAN")) {  // "T,CS": that cluster geometry only
        int t = 0, c = 0;
        if (sscanf(e, "%d,%d", &t, &c) == 2) { only_t = t; only_cs = c; }
    }
    const double pack = 0.2;
    if (const int kt = mfma_kt_for(n, 2); kt > 0 && (only_t == 0 || only_t == 1))
        v.push_back({2, 0, kt, 0, 32 * cus, std::max(0.953 * kt, 4.0 + 0.43 * kt) + pack});
    for (int T = 1; T <= 2; ++T)
        for (int CS = (T == 1 ? 1 : 2); CS <= 3; ++CS) {
            if (only_t != 0 && (T != only_t || CS != only_cs)) continue;
            const Tiling1 til = mfma1_tiling_for(n, CS, T);
            const double kt = til.KT - 0.5 * til.ODD;     // tiles per matrix wave, SIMD average
            const int c = (CS == 1) ? cus : ccus;
            if (til.KT == 0 || (CS > 1 && c < 8 * CS)) continue;
            const int clusters = (CS == 1) ? c : (c / (8 * CS)) * 8;
            const double chain = (T == 2) ? 3.0 + 0.25 * kt
                                          : (CS == 1 ? 1.48 + 0.29 * kt : (CS == 2 ? 2.3 : 2.3 + 0.1 * kt));
            const double work = (T == 2) ? (CS == 2 ? 1.04 : 1.08) * kt : (CS == 1 ? 0.49 : 0.55) * kt;
            v.push_back({T, CS, til.KT, til.ODD, 16 * T * clusters, std::max(work, chain) + pack});
        }
    return v;
}

// G of an explicit geometry: 0 = the tiling the library picks; 1 .. 99 = KT tiles for every matrix
// wave; 100 + KT = the odd deal (waves 3..5 take KT - 1)
static Tiling1 explicit_tiling(int G, int n, int CS, int T)
{
    if (G <= 0) return mfma1_tiling_for(n, CS, T);
    return G >= 100 ? Tiling1{G - 100, 1} : Tiling1{G, 0};
}

// One launch of the plan: responses [k_begin, k_end) in one geometry; launches of one geometry
// share `region`, the slice of the workspace that holds its packed operands.
struct PlannedLaunch {
    int T, CS;          // CS = 0: mfma_sweep_kernel<2, KT>
    Tiling1 til;
    int k_begin, k_end;
    int region;
};

static int make_plan(int n, int d, const SweepConfig& cfg, std::vector<PlannedLaunch>* plan)
{
    plan->clear();
    if (cfg.C < 0 || cfg.C > 2) {
        set_error("mfma sweep: bad geometry (T in 0..2)");
        return LOCUS_HIP_EINVAL;
    }
    if (cfg.C == 1) {  // one tile per cluster: mfma_sweep1_kernel, S = workgroups per cluster
        // S = workgroups per cluster (1, 2, 3); 0 = chosen from d; anything else (legacy) = 1
        const int CS = (cfg.S == 2 || cfg.S == 3) ? cfg.S : (cfg.S == 0 ? choose_cs(n, (d + 15) / 16) : 1);
        plan->push_back({1, CS, explicit_tiling(cfg.G, n, CS, 1), 0, d, 0});
        return 0;
    }
    if (cfg.C == 2 && cfg.S >= 2 && cfg.S <= 3) {
        plan->push_back({2, cfg.S, explicit_tiling(cfg.G, n, cfg.S, 2), 0, d, 0});
        return 0;
    }
    if (cfg.C != 0) {
        plan->push_back({cfg.C, 0, Tiling1{cfg.G > 0 ? cfg.G : mfma_kt_for(n, cfg.C), 0}, 0, d, 0});
        return 0;
    }
    if (mfma_kt_for(n, 1) == 0) {
        set_error("mfma sweep: n too large for the matrix-core kernel");
        return LOCUS_HIP_EUNSUPPORTED;
    }
    const std::vector<LaunchType> types = launch_types(n);
    if (types.empty()) {
        set_error("mfma sweep: no launch geometry for this shape");
        return LOCUS_HIP_EUNSUPPORTED;
    }
    // best[u] = cheapest cover of u units of 16 responses
    const int U = (d + 15) / 16;
    std::vector<double> best((size_t)U + 1, 0.0);
    std::vector<int> pick((size_t)U + 1, -1);
    for (int u = 1; u <= U; ++u) {
        best[u] = 1e300;
        for (size_t t = 0; t < types.size(); ++t) {
            const int c = std::max(1, types[t].cap / 16);
            const double v = types[t].cost + best[std::max(0, u - c)];
            if (v < best[u]) { best[u] = v; pick[u] = (int)t; }
        }
    }
    std::vector<int> region_of(types.size(), -1);
    int k = 0, nregions = 0;
    for (int u = U; u > 0;) {
        const LaunchType& lt = types[pick[u]];
        const int take = std::min(d - k, lt.cap);
        if (region_of[pick[u]] < 0) region_of[pick[u]] = nregions++;
        plan->push_back({lt.T, lt.CS, Tiling1{lt.KT, lt.ODD}, k, k + take, region_of[pick[u]]});
        k += take;
        u -= std::max(1, lt.cap / 16);
    }
    return 0;
}

// The plan as rows {T, CS, KT, ODD, k_begin, k_end, region, capacity} (introspection, tests)
int mfma_plan_rows(int n, int d, const SweepConfig& cfg, int32_t* out, int max_rows)
{
    std::vector<PlannedLaunch> plan;
    if (make_plan(n, d, cfg, &plan)) return -1;
    const int cus = device_cus(), ccus = cluster_cus();
    int r = 0;
    for (const PlannedLaunch& l : plan) {
        if (r >= max_rows) break;
        const int cap = (l.CS == 0) ? 32 * cus
                                    : 16 * l.T * ((l.CS == 1) ? cus : (ccus / (8 * l.CS)) * 8);
        const int32_t row[8] = {l.T, l.CS, l.til.KT, l.til.ODD, l.k_begin, l.k_end, l.region, cap};
        for (int i = 0; i < 8; ++i) out[8 * r + i] = row[i];
        ++r;
    }
    return (int)plan.size();
}

static size_t region_stride(int64_t n, int64_t p) { return (mfma_workspace_bytes(n, p) + 255) / 256 * 256; }

// Geometries (= workspace regions of packed operands) in the plan for (n, d): what a workspace
// must hold for locus_hip_sweep_prepack_dev, which keeps all of them side by side.
int mfma_plan_regions(int n, int d, const SweepConfig& cfg)
{
    std::vector<PlannedLaunch> plan;
    if (make_plan(n, d, cfg, &plan)) return 0;
    int r = 0;
    for (const PlannedLaunch& l : plan) r = std::max(r, l.region + 1);
    return r;
}

// cfg: C = T (tiles of 16 responses per cluster; 0 = plan by d), S = workgroups per cluster,
// G = KT (16-row tiles per matrix wave, 0 = smallest instantiated value that covers n; 100 + KT =
// waves 3..5 of every workgroup take KT - 1).
//   C = 1: mfma_sweep1_kernel<KT, S>, S = 1, 2, 3 (0 = chosen from d; anything else = 1);
//   C = 2: S = 2, 3: mfma_sweep1_kernel<KT, S, 2> (two tiles per cluster);
//          any other S: mfma_sweep_kernel<2, KT> (two tiles on one workgroup).
// mode 0: every launch packs its operands into the one region (xb_bytes >= mfma_workspace_bytes)
//         unless the launch before it used the same geometry, then runs;
// mode 1: pack only, every geometry into a region of its own (xb_bytes >= regions x that);
// mode 2: the operands are there (mode 1 with the same n, p, d, sweep order, geometry): refresh
//         the steps' log-odds, which is all that changes between iterations, and run.
int run_mfma_sweep(const SweepParams& prm, const SweepConfig& cfg, void* xb_ws, size_t xb_bytes,
                   hipStream_t st, int mode)
{
    std::vector<PlannedLaunch> plan;
    int rc = make_plan(prm.n, prm.d, cfg, &plan);
    if (rc) return rc;
    const size_t stride = region_stride(prm.n, prm.p);
    int nregions = 0;
    for (const PlannedLaunch& l : plan) nregions = std::max(nregions, l.region + 1);
    if (mode != 0 && (size_t)nregions * stride > xb_bytes) {
        set_error("mfma sweep: workspace too small for operands packed ahead (" + std::to_string(nregions) +
                  " geometries; see locus_hip_sweep_prepack_workspace_bytes)");
        return LOCUS_HIP_EINVAL;
    }
    std::vector<char> touched((size_t)nregions + 1, 0);
    int last_region = -1;
    for (const PlannedLaunch& l : plan) {
        int what;
        char* base = static_cast<char*>(xb_ws);
        size_t bytes = xb_bytes;
        if (mode == 0) {
            what = kLaunch | (l.region != last_region ? kPack : 0);
        } else {
            base += (size_t)l.region * stride;
            bytes = stride;
            const bool first = !touched[l.region];
            touched[l.region] = 1;
            if (mode == 1) {
                if (!first) continue;
                what = kPack;
            } else {
                what = kLaunch | (first ? kRefresh : 0);
            }
        }
        rc = (l.CS == 0) ? pack_and_launch(prm, l.T, l.til.KT, l.k_begin, l.k_end, what, base, bytes, st)
                         : pack_and_launch1(prm, l.T, l.CS, l.til, l.k_begin, l.k_end, what, base, bytes, st);
        if (rc) return rc;
        last_region = l.region;
    }
    return 0;
}

}  // namespace locus
