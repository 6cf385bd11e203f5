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

constexpr int LDP = 16;

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
        pa[q] = g.A + (long)(oa ? m0 + am : 0) * g.sa_m + (long)ak[q] * g.sa_k;
        pb[q] = g.B + (long)(ob ? n0 + bn : 0) * g.sb_n + (long)bk[q] * g.sb_k;
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
            const double v = (((okm >> q) & 1u) && k0 + ak[q] < g.k) ? *pa[q] : 0.0;
            ra[q] = g.sqA ? v * v : v;
            const double u = (((okn >> q) & 1u) && k0 + bk[q] < g.k) ? *pb[q] : 0.0;
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
    load_tile(0);
    store_tile(0);
    if (BK < g.k) load_tile(BK);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.k; k0 += BK) {
        if (k0 + BK < g.k) store_tile(buf ^ 1);
        if (k0 + 2 * BK < g.k) load_tile(k0 + 2 * BK);
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
        // the upper wave group hands its partial tile to the lower one through LDS (As is free:
        // the loop ended with a barrier)
        double* xch = lds;
        static_assert(ASZ + BSZ >= BM * BN, "exchange buffer");
        const int slot = ((wave * TW * TW) * 4) * 64 + lane;
        if (ks == 1) {
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[slot + ((i * TW + j) * 4 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (ks == 1) return;
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += xch[slot + ((i * TW + j) * 4 + r) * 64];
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
                    double* c = g.C + (long)m * g.sc_m + (long)n * g.sc_n;
                    double v = g.alpha * acc[i][j][r];
                    if (g.beta != 0.0) v += g.beta * (*c);
                    *c = v;
                }
            }
}

int gemm_f64(const GemmDesc& g, hipStream_t st)
{
    if (g.m <= 0 || g.n <= 0) return 0;
    const char* tw = getenv("LOCUS_HIP_GEMM_TW");
    const char* ksv = getenv("LOCUS_HIP_GEMM_KS");
    if (tw && tw[0] == '4') {
        dim3 grid((g.m + 127) / 128, (g.n + 127) / 128);
        hipLaunchKernelGGL((gemm_f64_kernel<4, 1, 16>), grid, dim3(256), 0, st, g);
    } else {
        dim3 grid((g.m + 63) / 64, (g.n + 63) / 64);
        const long tiles = (long)grid.x * grid.y;
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
    return 0;
}

}  // namespace locus
