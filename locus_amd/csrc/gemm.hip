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
// the MFMAs of tile t read this one and tile t + 2 is in flight from global memory (register
// prefetch through running pointers).  LDS rows are padded by 16 doubles so that the two 32-lane
// halves of a ds_read_b64 fall on disjoint banks.  KS = 2: two groups of 4 waves split the k-steps
// of every K-tile and add their partial tiles through LDS at the end, so that even a workgroup
// that has its CU to itself runs two waves per SIMD.  Shipped: TW = 2 (64 x 64), KS = 2, BK = 16
// (74 registers, 41 KB of LDS), BK = 32 when there is at most about one output tile per CU.
// Measured on MI355X (scripts/perf_gemm.py): 46 TFLOP/s on the C3 mat_x_m1 (TW = 4, 128 x 128,
// one workgroup per CU: 29 TFLOP/s, kept behind LOCUS_HIP_GEMM_TW=4 for experiments;
// LOCUS_HIP_GEMM_KS / LOCUS_HIP_GEMM_BK override the other two choices).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace locus {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LDP = 17;

template <int TW, int KS, int BK>
__global__ void __launch_bounds__(256 * KS) gemm_f64_kernel(const GemmDesc g)
{
    constexpr int BM = 32 * TW, BN = 32 * TW;
    constexpr int NT = 256 * KS;        // KS groups of 4 waves split the k-steps of every K-tile
    constexpr int NQ = BM * BK / NT;    // elements of each operand tile per thread
    constexpr int ASZ = 2 * BK * (BM + LDP), BSZ = 2 * BK * (BN + LDP);
    __shared__ double lds[ASZ + BSZ];  // one array: the KS == 2 epilogue reuses all of it
    double (*As)[BK][BM + LDP] = reinterpret_cast<double (*)[BK][BM + LDP]>(lds);
    double (*Bs)[BK][BN + LDP] = reinterpret_cast<double (*)[BK][BN + LDP]>(lds + ASZ);

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ks = tid >> 8;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    // split-K: this workgroup's slice of the contraction
    const int kb = g.kchunk ? (int)blockIdx.z * g.kchunk : 0;
    const int kend = g.kchunk ? (kb + g.kchunk < g.k ? kb + g.kchunk : g.k) : g.k;

    // element -> (row-in-tile, k) mapping follows the contiguous dimension of each operand; per
    // element: a running pointer, its k index inside a K-tile and whether its row is in range
    const bool a_kfast = (g.sa_k == 1);
    const bool b_kfast = (g.sb_k == 1);
    const double* pa[NQ];
    const double* pb[NQ];
    int ak[NQ], bk[NQ], al[NQ], bl[NQ];  // k index; LDS offset of the element
    unsigned okm = 0, okn = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * NT;
        int am, bn;
        if (a_kfast) { ak[q] = e % BK; am = e / BK; } else { am = e % BM; ak[q] = e / BM; }
        if (b_kfast) { bk[q] = e % BK; bn = e / BK; } else { bn = e % BN; bk[q] = e / BN; }
        const bool oa = m0 + am < g.m, ob = n0 + bn < g.n;
        okm |= (oa ? 1u : 0u) << q;
        okn |= (ob ? 1u : 0u) << q;
        pa[q] = g.A + (long)(oa ? m0 + am : 0) * g.sa_m + (long)(kb + ak[q]) * g.sa_k;
        pb[q] = g.B + (long)(ob ? n0 + bn : 0) * g.sb_n + (long)(kb + bk[q]) * g.sb_k;
        al[q] = ak[q] * (BM + LDP) + am;
        bl[q] = bk[q] * (BN + LDP) + bn;
    }
    const long da = (long)BK * g.sa_k, db = (long)BK * g.sb_k;

    double4_t acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double ra[NQ], rb[NQ];
    auto load_tile = [&](int k0) {  // tile starting at k0 -> registers; advances the pointers
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double v = (((okm >> q) & 1u) && k0 + ak[q] < kend) ? *pa[q] : 0.0;
            ra[q] = g.sqA ? v * v : v;
            const double u = (((okn >> q) & 1u) && k0 + bk[q] < kend) ? *pb[q] : 0.0;
            rb[q] = g.sqB ? u * u : u;
            pa[q] += da;
            pb[q] += db;
        }
    };
    auto store_tile = [&](int buf) {
        double* as = &As[buf][0][0];
        double* bs = &Bs[buf][0][0];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            as[al[q]] = ra[q];
            bs[bl[q]] = rb[q];
        }
    };

    // double-buffered LDS: ONE barrier per K-tile; tile t + 1 is written to the other buffer
    // while the MFMAs of tile t read this one, tile t + 2 is in flight from global memory
    load_tile(kb);
    store_tile(0);
    if (kb + BK < kend) load_tile(kb + BK);
    __syncthreads();
    int buf = 0;
    for (int k0 = kb; k0 < kend; k0 += BK) {
        if (k0 + BK < kend) store_tile(buf ^ 1);
        if (k0 + 2 * BK < kend) load_tile(k0 + 2 * BK);
#pragma unroll
        for (int kk = 4 * ks; kk < BK; kk += 4 * KS) {
            const int kr = kk + (lane >> 4);
            double a[TW], b[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) a[i] = As[buf][kr][wm * 16 * TW + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < TW; ++j) b[j] = Bs[buf][kr][wn * 16 * TW + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        buf ^= 1;
    }

    if constexpr (KS == 2) {
        // the upper wave group hands its partial tile to the lower one through LDS (the tile
        // buffers are free: the loop ended with a barrier); in NH passes over the rows of MFMA
        // tiles when the partial tile is larger than the buffers (TW = 4)
        double* xch = lds;
        constexpr int NH = (BM * BN > ASZ + BSZ) ? 2 : 1;
        constexpr int TH = TW / NH;  // rows of MFMA tiles per pass
        static_assert(ASZ + BSZ >= BM * BN / NH && TW % NH == 0, "exchange buffer");
        const int slot = ((wave * TH * TW) * 4) * 64 + lane;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (ks == 1) {
#pragma unroll
                for (int i = 0; i < TH; ++i)
#pragma unroll
                    for (int j = 0; j < TW; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xch[slot + ((i * TW + j) * 4 + r) * 64] = acc[h * TH + i][j][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < TH; ++i)
#pragma unroll
                    for (int j = 0; j < TW; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[h * TH + i][j][r] += xch[slot + ((i * TW + j) * 4 + r) * 64];
            }
            if (h + 1 < NH) __syncthreads();
        }
        if (ks == 1) return;
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 16 * TW + i * 16 + (lane >> 4) + 4 * r;
                const int n = n0 + wn * 16 * TW + j * 16 + (lane & 15);
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

int gemm_f64(const GemmDesc& g_in, hipStream_t st)
{
    if (g_in.m <= 0 || g_in.n <= 0) return 0;
    GemmDesc g = g_in;
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
    if (tw && tw[0] == '4') {
        dim3 grid((g.m + 127) / 128, (g.n + 127) / 128);
        if (ksv && ksv[0] == '2') {
            auto kern = gemm_f64_kernel<4, 2, 16>;
            LOCUS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 0));
            hipLaunchKernelGGL(kern, grid, dim3(512), 0, st, g);
        } else {
            hipLaunchKernelGGL((gemm_f64_kernel<4, 1, 16>), grid, dim3(256), 0, st, g);
        }
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
        if (split && deep) {
            auto kern = gemm_f64_kernel<2, 2, 32>;
            LOCUS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 0));
            hipLaunchKernelGGL(kern, grid, dim3(512), 0, st, g);
        } else if (split) {
            hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 16>), grid, dim3(512), 0, st, g);
        } else {
            hipLaunchKernelGGL((gemm_f64_kernel<2, 1, 16>), grid, dim3(256), 0, st, g);
        }
    }
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
