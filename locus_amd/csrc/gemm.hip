// fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64), arbitrary strides.
//
// Serves the dense contractions of the path: the two products of coreBatch /
// coreZBatch / coreProbitBatch (src/coreBatch.cpp:32-33,40; :63-64,71; :91-92,98),
// the initial mat_x_m1 = X %*% beta_vb (R/update_vb.R:114) and the X^2 products of
// the logit / probit drivers (R/update_vb.R:107-111,163-167; R/elbo.R:210-237),
// for which `sqA` squares the A operand on load instead of materialising X^2.
//
// Tiling: (32 TW) x (32 TW) block tile, K-tile BK, waves owning a quadrant as TW x TW MFMA tiles.
// LDS is double-buffered: ONE barrier per K-tile, tile t + 1 is written to the other buffer while
// the MFMAs of tile t read this one and tile t + 2 is in flight from global memory (raw values in
// registers; squaring, when asked, happens on the way to LDS one K-tile later, so nothing waits
// for a load inside the K-tile that issued it).  LDS rows are padded by 17 doubles (odd: the
// transposing stores of a k-contiguous operand spread over the banks).  KS = 2: two groups of 4
// waves split the k-steps of every K-tile and add their partial tiles through LDS at the end.
//
// The K loop has NO vector ALU instruction.  On gfx950 the fp64 MFMA executes on the SIMD's fp64
// vector lanes (matrix and vector fp64 peaks are the same 78.6 TFLOP/s), so every VALU instruction
// any wave of the SIMD issues -- address arithmetic, masks, selects -- is time its matrix pipe
// does not get.  Hence: tile elements are raw buffer loads (wave-uniform 64-bit base in the scalar
// resource, advanced with s_add; loop-invariant 32-bit lane offsets), the two LDS buffers are
// compile-time constants (the loop is unrolled by two: every LDS address is a loop-invariant
// register plus an immediate), interior K-tiles carry no k mask, rows beyond the matrix read the
// tile's first row instead of being masked (they only feed output rows that are never stored),
// and squaring is compiled in only when asked.  Removing ~28 VALU instructions per 8 MFMAs took
// the C3 init product from 46 to 58 TFLOP/s.
// Shipped: 64 x 128 tiles (TWM = 2, TWN = 4, KS = 2, BK = 16; 58 KB of LDS, two workgroups per CU)
// for products with many output tiles and an unsplit contraction, else
// TW = 2 (64 x 64), KS = 2, BK = 16 (41 KB of LDS, three workgroups per CU), BK = 32 when
// there is at most about one output tile per CU.  Measured on MI355X (scripts/perf_gemm.py):
// 58.3 TFLOP/s (74 % of peak) on the C3 mat_x_m1, 49-50 on the C2 coreBatch products, 43-46 on
// the C4 X^2 products.  TW = 4 (128 x 128, one workgroup per CU) stays behind LOCUS_HIP_GEMM_TW=4
// for experiments; LOCUS_HIP_GEMM_KS / _BK / _SPLITK / _OFF64 override the other choices.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace locus {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LDP = 17;

// One element of an operand tile.  OFF32: the address is a wave-uniform 64-bit base plus a 32-bit
// lane offset, which is exactly a raw buffer load (base in the scalar resource, `offen` lane
// offset); the resource is rebuilt from the running base on the scalar unit.
template <bool OFF32, class O>
__device__ __forceinline__ double ld_elem(const char* base, O off)
{
    if constexpr (OFF32) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(base), 0, 0x7fffffff, 0x00020000 /* 32-bit data format, gfx9 */);
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    } else {
        return *reinterpret_cast<const double*>(base + off);
    }
}

// SQ: 0 = no operand is squared on its way to LDS, 1 = A is, 2 = g.sqA / g.sqB say which.
// OFF32: the elements of a tile lie within 2^31 bytes after the tile's first element (non-negative
// strides), so the K loop has no vector address arithmetic at all: the bases advance on the scalar
// unit.  On gfx950 the fp64 MFMA runs on the SIMD's fp64 vector lanes (same 78.6 TFLOP/s peak):
// every vector instruction a wave issues is time the matrix pipe of that SIMD does not get.
template <int TWM, int TWN, int KS, int BK, int SQ, bool OFF32>
__global__ void __launch_bounds__(256 * KS) gemm_f64_kernel(const GemmDesc g)
{
    constexpr int BM = 32 * TWM, BN = 32 * TWN;   // a wave owns TWM x TWN MFMA tiles of 16 x 16
    constexpr int NT = 256 * KS;        // KS groups of 4 waves split the k-steps of every K-tile
    constexpr int NQA = BM * BK / NT;   // elements of the A tile per thread
    constexpr int NQB = BN * BK / NT;   // ... of the B tile
    constexpr int LDA = BM + LDP, LDB = BN + LDP;
    constexpr int ASZ = 2 * BK * LDA, BSZ = 2 * BK * LDB;
    __shared__ double lds[ASZ + BSZ];  // one array: the KS == 2 epilogue reuses all of it
    typedef typename std::conditional<OFF32, unsigned, long>::type off_t;  // byte offsets

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ks = tid >> 8;
    const int wm = wave >> 1, wn = wave & 1;
    // Tile order.  Default: the launch order (x fastest).  LOCUS_HIP_GEMM_XCD=1 gives every XCD
    // (workgroups go to the 8 XCDs round-robin by linear id) a contiguous eighth of the tile
    // list with the short grid dimension fastest, so that the workgroups of one XCD share few
    // strips of the long operand side.  Measured on MI355X (scripts/gemm_traffic.sh, round 3):
    // it does NOT pay -- the C3 init product reads 58.6 GB through the L2s instead of 40.8 GB and
    // takes 17.8 ms instead of 17.4; the X^2 products of C4 read 0.92 GB instead of 2.63 GB for
    // the same time.  Workgroups that share a strip are dispatched as earlier ones retire, so
    // after the first wave they sit at unrelated k and a 4 MB L2 holds no strip long enough to
    // be hit twice, whatever the order; the kernel is not bound by that traffic either (the
    // time follows SQ_VALU_MFMA_BUSY: 84 % at the 2.09 GHz the chip sustains under this load).
    int bx, by;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int lin = blockIdx.x + gx * blockIdx.y, tiles = gx * gy;
        const int xcd = lin & 7, slot = lin >> 3;
        const int q8 = tiles >> 3, rem = tiles & 7;
        const int tile = !g.xcd_order ? lin : xcd * q8 + (xcd < rem ? xcd : rem) + slot;
        if (!g.xcd_order) { bx = blockIdx.x; by = blockIdx.y; }
        else if (gx <= gy) { bx = tile % gx; by = tile / gx; }
        else { by = tile % gy; bx = tile / gy; }
    }
    const int m0 = bx * BM, n0 = by * BN;
    // split-K: this workgroup's slice of the contraction
    const int kb = g.kchunk ? (int)blockIdx.z * g.kchunk : 0;
    const int kend = g.kchunk ? (kb + g.kchunk < g.k ? kb + g.kchunk : g.k) : g.k;

    // element -> (row-in-tile, k) mapping follows the contiguous dimension of each operand
    const bool a_kfast = (g.sa_k == 1);
    const bool b_kfast = (g.sb_k == 1);
    const char* Ab = reinterpret_cast<const char*>(g.A + (long)m0 * g.sa_m + (long)kb * g.sa_k);
    const char* Bb = reinterpret_cast<const char*>(g.B + (long)n0 * g.sb_n + (long)kb * g.sb_k);
    off_t oa[NQA], ob[NQB];
    int ak[NQA], bk[NQB];
    double* al[NQA];  // where the element goes in buffer 0 of its LDS tile
    double* bl[NQB];
    // rows beyond the matrix read the tile's first row: they only feed output rows that are
    // never stored
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
        const int e = tid + q * NT;
        int am;
        if (a_kfast) { ak[q] = e % BK; am = e / BK; } else { am = e % BM; ak[q] = e / BM; }
        const long ea = (long)(m0 + am < g.m ? am : 0) * g.sa_m + (long)ak[q] * g.sa_k;
        oa[q] = (off_t)(ea * 8);
        al[q] = lds + ak[q] * LDA + am;
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        const int e = tid + q * NT;
        int bn;
        if (b_kfast) { bk[q] = e % BK; bn = e / BK; } else { bn = e % BN; bk[q] = e / BN; }
        const long eb = (long)(n0 + bn < g.n ? bn : 0) * g.sb_n + (long)bk[q] * g.sb_k;
        ob[q] = (off_t)(eb * 8);
        bl[q] = lds + ASZ + bk[q] * LDB + bn;
    }
    const long da = (long)BK * g.sa_k * 8, db = (long)BK * g.sb_k * 8;
    // this lane's operands of k-step 0 in buffer 0
    const double* ar = lds + (4 * ks + (lane >> 4)) * LDA + wm * 16 * TWM + (lane & 15);
    const double* br = lds + ASZ + (4 * ks + (lane >> 4)) * LDB + wn * 16 * TWN + (lane & 15);

    double4_t acc[TWM][TWN];
#pragma unroll
    for (int i = 0; i < TWM; ++i)
#pragma unroll
        for (int j = 0; j < TWN; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // A tile travels global memory -> registers (raw values, nothing depends on them until the
    // next iteration, so the loads stay in flight behind a whole K-tile of MFMAs) -> LDS (squared
    // there when asked).  Interior K-tiles load without any mask; only the last one checks k.
    double ra[NQA], rb[NQB];
    auto load_tile = [&](int k0) {  // tile starting at k0 -> registers; advances the bases
        if (k0 + BK <= kend) {
#pragma unroll
            for (int q = 0; q < NQA; ++q) ra[q] = ld_elem<OFF32>(Ab, oa[q]);
#pragma unroll
            for (int q = 0; q < NQB; ++q) rb[q] = ld_elem<OFF32>(Bb, ob[q]);
        } else {
#pragma unroll
            for (int q = 0; q < NQA; ++q) ra[q] = (k0 + ak[q] < kend) ? ld_elem<OFF32>(Ab, oa[q]) : 0.0;
#pragma unroll
            for (int q = 0; q < NQB; ++q) rb[q] = (k0 + bk[q] < kend) ? ld_elem<OFF32>(Bb, ob[q]) : 0.0;
        }
        Ab += da;
        Bb += db;
    };
    // One K-tile: tile t + 1 goes from registers to the other buffer, tile t + 2 is requested,
    // the MFMAs of tile t read buffer `buf` (a compile-time constant: every LDS address of the
    // loop is a loop-invariant register plus an immediate).
    auto k_tile = [&](auto bufc, int k0) {
        constexpr int buf = decltype(bufc)::value;
        if (k0 + BK < kend) {
#pragma unroll
            for (int q = 0; q < NQA; ++q) {
                double va = ra[q];
                if constexpr (SQ == 1) va = va * va;
                if constexpr (SQ == 2) va = g.sqA ? va * va : va;
                al[q][(buf ^ 1) * BK * LDA] = va;
            }
#pragma unroll
            for (int q = 0; q < NQB; ++q) {
                double vb = rb[q];
                if constexpr (SQ == 2) vb = g.sqB ? vb * vb : vb;
                bl[q][(buf ^ 1) * BK * LDB] = vb;
            }
        }
        if (k0 + 2 * BK < kend) load_tile(k0 + 2 * BK);
#pragma unroll
        for (int s = 0; s < BK / (4 * KS); ++s) {
            double a[TWM], b[TWN];
#pragma unroll
            for (int i = 0; i < TWM; ++i) a[i] = ar[buf * BK * LDA + 4 * KS * s * LDA + i * 16];
#pragma unroll
            for (int j = 0; j < TWN; ++j) b[j] = br[buf * BK * LDB + 4 * KS * s * LDB + j * 16];
#pragma unroll
            for (int i = 0; i < TWM; ++i)
#pragma unroll
                for (int j = 0; j < TWN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    };

    load_tile(kb);
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
        double va = ra[q];
        if constexpr (SQ == 1) va = va * va;
        if constexpr (SQ == 2) va = g.sqA ? va * va : va;
        al[q][0] = va;
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        double vb = rb[q];
        if constexpr (SQ == 2) vb = g.sqB ? vb * vb : vb;
        bl[q][0] = vb;
    }
    if (kb + BK < kend) load_tile(kb + BK);
    __syncthreads();
    for (int k0 = kb; k0 < kend; k0 += 2 * BK) {
        k_tile(std::integral_constant<int, 0>{}, k0);
        if (k0 + BK < kend) k_tile(std::integral_constant<int, 1>{}, k0 + BK);
    }

    if constexpr (KS == 2) {
        // the upper wave group hands its partial tile to the lower one through LDS (the tile
        // buffers are free: the loop ended with a barrier); in NH passes over the rows of MFMA
        // tiles when the partial tile is larger than the buffers (TW = 4)
        double* xch = lds;
        constexpr int NH = (BM * BN > ASZ + BSZ) ? 2 : 1;
        constexpr int TH = TWM / NH;  // rows of MFMA tiles per pass
        static_assert(ASZ + BSZ >= BM * BN / NH && TWM % NH == 0, "exchange buffer");
        const int slot = ((wave * TH * TWN) * 4) * 64 + lane;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (ks == 1) {
#pragma unroll
                for (int i = 0; i < TH; ++i)
#pragma unroll
                    for (int j = 0; j < TWN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xch[slot + ((i * TWN + j) * 4 + r) * 64] = acc[h * TH + i][j][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < TH; ++i)
#pragma unroll
                    for (int j = 0; j < TWN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[h * TH + i][j][r] += xch[slot + ((i * TWN + j) * 4 + r) * 64];
            }
            if (h + 1 < NH) __syncthreads();
        }
        if (ks == 1) return;
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < TWM; ++i)
#pragma unroll
        for (int j = 0; j < TWN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * TWM + i * 16 + (lane >> 4) + 4 * r;
                const int n = n0 + wn * 16 * TWN + j * 16 + (lane & 15);
                if (m < g.m && n < g.n) {
                    double* c = g.C + (long)blockIdx.z * g.cz + (long)m * g.sc_m + (long)n * g.sc_n;
                    double v = g.alpha * acc[i][j][r];
                    if (g.beta != 0.0) v += g.beta * (*c);
                    *c = v;
                }
            }
}

// C = alpha * sum_z part[z] + beta * C, partial products summed in the order of z (deterministic)
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(const double* part, int nz, int m,
                                                                 int n, double* C, long sc_m,
                                                                 long sc_n, double alpha,
                                                                 double beta)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long mn = (long)m * n;
    if (e >= mn) return;
    double acc = 0.0;
    for (int z = 0; z < nz; ++z) acc += part[(long)z * mn + e];
    double* c = C + (e % m) * sc_m + (e / m) * sc_n;
    double v = alpha * acc;
    if (beta != 0.0) v += beta * (*c);
    *c = v;
}

// Picks the instantiation: squaring code only when an operand is squared; 32-bit lane offsets when
// every element of a tile lies within 2^31 bytes after the tile's first one.
template <int TWM, int TWN, int KS, int BK>
static int launch_gemm(const GemmDesc& g, dim3 grid, hipStream_t st)
{
    constexpr int BM = 32 * TWM, BN = 32 * TWN;
    const long lim = (1L << 28) - 1;  // elements
    const bool off32 = !getenv("LOCUS_HIP_GEMM_OFF64") && g.sa_m >= 0 && g.sa_k >= 0 &&
                       g.sb_n >= 0 && g.sb_k >= 0 &&
                       (BM - 1) * g.sa_m + (BK - 1) * g.sa_k <= lim &&
                       (BN - 1) * g.sb_n + (BK - 1) * g.sb_k <= lim;
    auto go = [&](auto kern) {
        if (sizeof(double) * 2 * BK * (BM + BN + 2 * LDP) > 64 * 1024)
            LOCUS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 0));
        hipLaunchKernelGGL(kern, grid, dim3(256 * KS), 0, st, g);
        return 0;
    };
    const int sq = (g.sqA && !g.sqB) ? 1 : ((g.sqA || g.sqB) ? 2 : 0);
    if (sq == 1)
        return off32 ? go(gemm_f64_kernel<TWM, TWN, KS, BK, 1, true>)
                     : go(gemm_f64_kernel<TWM, TWN, KS, BK, 1, false>);
    if (sq == 2)
        return off32 ? go(gemm_f64_kernel<TWM, TWN, KS, BK, 2, true>)
                     : go(gemm_f64_kernel<TWM, TWN, KS, BK, 2, false>);
    return off32 ? go(gemm_f64_kernel<TWM, TWN, KS, BK, 0, true>)
                 : go(gemm_f64_kernel<TWM, TWN, KS, BK, 0, false>);
}

int gemm_f64(const GemmDesc& g_in, hipStream_t st)
{
    if (g_in.m <= 0 || g_in.n <= 0) return 0;
    GemmDesc g = g_in;
    g.xcd_order = getenv("LOCUS_HIP_GEMM_XCD") ? 1 : 0;
    // Few output tiles and a deep contraction (X beta of coreBatch at C2, Q = X^2 D at C4: one
    // 64 x 64 tile per CU, 20 000 k-steps each): split the contraction over nz workgroups per
    // tile, partial products to a stream-ordered scratch buffer, summed in a fixed order.
    double* part = nullptr;
    int nz = 1;
    {
        const long tiles = (long)((g.m + 63) / 64) * ((g.n + 63) / 64);
        const int cus = device_cus();
        const char* sk = getenv("LOCUS_HIP_GEMM_SPLITK");
        int want = (int)((3L * cus + tiles / 2) / (tiles > 0 ? tiles : 1));
        if (sk) want = atoi(sk);
        if (want > 8) want = 8;
        while (want > 1 && g.k / want < 1024) --want;  // keep >= 1024 k-steps per slice
        if (want > 1 && !getenv("LOCUS_HIP_GEMM_TW")) {
            nz = want;
            const int chunk = ((g.k + nz - 1) / nz + 31) / 32 * 32;
            nz = (g.k + chunk - 1) / chunk;
            LOCUS_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&part),
                                           sizeof(double) * (size_t)nz * g.m * g.n, st));
            g.kchunk = chunk;
            g.cz = (long)g.m * g.n;
            g.C = part;
            g.sc_m = 1;
            g.sc_n = g.m;
            g.alpha = 1.0;
            g.beta = 0.0;
        }
    }
    const char* tw = getenv("LOCUS_HIP_GEMM_TW");
    const char* ksv = getenv("LOCUS_HIP_GEMM_KS");
    int rc = 0;
    // 64 x 128 tiles (a wave owns 2 x 4 MFMA tiles: 6 LDS reads per 8 MFMAs instead of 4 per 4, half
    // the barriers per MFMA) when the contraction is not split, n pads to 128 as well as to 64 and
    // there are two workgroups' worth of such tiles for every CU: C3 init 58.8 -> 60.1 TFLOP/s,
    // X'R of coreBatch 49.9 -> 51.8, (X^2)'(2 psi) 42.6 -> 45.0.  LOCUS_HIP_GEMM_TW=w forces it
    // (no split-K then), =2 keeps 64 x 64.
    const bool wide = tw ? tw[0] == 'w'
                         : (nz == 1 && ((g.n + 127) / 128 * 128) * 100L <= ((g.n + 63) / 64 * 64) * 104L &&
                            (long)((g.m + 63) / 64) * ((g.n + 127) / 128) >= 2L * device_cus());
    if (wide) {
        dim3 grid((g.m + 63) / 64, (g.n + 127) / 128, nz);
        rc = launch_gemm<2, 4, 2, 16>(g, grid, st);
    } else if (tw && tw[0] == '4') {
        dim3 grid((g.m + 127) / 128, (g.n + 127) / 128);
        if (ksv && ksv[0] == '2')
            rc = launch_gemm<4, 4, 2, 16>(g, grid, st);
        else
            rc = launch_gemm<4, 4, 1, 16>(g, grid, st);
    } else {
        dim3 grid((g.m + 63) / 64, (g.n + 63) / 64, nz);
        const long tiles = (long)grid.x * grid.y * nz;
        // few output tiles: a workgroup has its CU to itself, so two wave groups split the
        // k-steps of every K-tile (two waves per SIMD overlap each other's LDS / barrier latency)
        const bool split = ksv ? (ksv[0] == '2') : true;
        const char* bkv = getenv("LOCUS_HIP_GEMM_BK");
        // at most ~one output tile per CU: the K loop is long and serial, a deeper K-tile halves
        // its barriers (82 KB of LDS: one workgroup per CU, which is all there is anyway)
        const bool deep = bkv ? (bkv[0] == '3') : (tiles <= 384);
        if (split && deep)
            rc = launch_gemm<2, 2, 2, 32>(g, grid, st);
        else if (split)
            rc = launch_gemm<2, 2, 2, 16>(g, grid, st);
        else
            rc = launch_gemm<2, 2, 1, 16>(g, grid, st);
    }
    if (rc) return rc;
    LOCUS_HIP_CHECK(hipGetLastError());
    if (part != nullptr) {
        const long mn = (long)g_in.m * g_in.n;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256),
                           0, st, part, nz, g_in.m, g_in.n, g_in.C, g_in.sc_m, g_in.sc_n,
                           g_in.alpha, g_in.beta);
        LOCUS_HIP_CHECK(hipGetLastError());
        LOCUS_HIP_CHECK(hipFreeAsync(part, st));
    }
    return 0;
}

}  // namespace locus
