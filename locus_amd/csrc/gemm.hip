// fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64), arbitrary strides.
//
// Serves the dense contractions of the path: the two products of coreBatch /
// coreZBatch / coreProbitBatch (src/coreBatch.cpp:32-33,40; :63-64,71; :91-92,98),
// the initial mat_x_m1 = X %*% beta_vb (R/update_vb.R:114) and the X^2 products of
// the logit / probit drivers (R/update_vb.R:107-111,163-167; R/elbo.R:210-237),
// for which `sqA` squares the A operand on load instead of materialising X^2.
//
// Tiling: (32 TW) x (32 TW) block tile, K-tile 16, 4 waves each owning a quadrant as TW x TW MFMA
// tiles; global loads of the next K-tile are issued before the MFMAs of the current one
// (register prefetch), LDS rows are padded by 16 doubles so that the two 32-lane halves of a
// ds_read_b64 fall on disjoint banks.  TW = 2 (64 x 64, several workgroups per CU) is the
// shipped tile: measured on MI355X it beats TW = 4 (128 x 128, one workgroup per CU, nothing to
// overlap its LDS fill with) on every product of the path (scripts/perf_gemm.py: 40.7 vs 29.7
// TFLOP/s on the C3 mat_x_m1).  LOCUS_HIP_GEMM_TW=4 selects the large tile for experiments.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace locus {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BK = 16, LDP = 16;

template <int TW>
__global__ void __launch_bounds__(256) gemm_f64_kernel(const GemmDesc g)
{
    constexpr int BM = 32 * TW, BN = 32 * TW;
    constexpr int NQ = BM * BK / 256;  // elements of each operand tile per thread
    __shared__ double As[BK][BM + LDP];
    __shared__ double Bs[BK][BN + LDP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // element -> (row-in-tile, k) mapping follows the contiguous dimension of each operand
    const bool a_kfast = (g.sa_k == 1);
    const bool b_kfast = (g.sb_k == 1);
    int am[NQ], ak[NQ], bn[NQ], bk[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * 256;
        if (a_kfast) { ak[q] = e % BK; am[q] = e / BK; } else { am[q] = e % BM; ak[q] = e / BM; }
        if (b_kfast) { bk[q] = e % BK; bn[q] = e / BK; } else { bn[q] = e % BN; bk[q] = e / BN; }
    }

    double4_t acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double ra[NQ], rb[NQ];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = m0 + am[q], k = k0 + ak[q];
            double v = (m < g.m && k < g.k) ? g.A[(long)m * g.sa_m + (long)k * g.sa_k] : 0.0;
            ra[q] = g.sqA ? v * v : v;
            const int n = n0 + bn[q], kb = k0 + bk[q];
            double u = (n < g.n && kb < g.k) ? g.B[(long)kb * g.sb_k + (long)n * g.sb_n] : 0.0;
            rb[q] = g.sqB ? u * u : u;
        }
    };

    load_tile(0);
    for (int k0 = 0; k0 < g.k; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            As[ak[q]][am[q]] = ra[q];
            Bs[bk[q]][bn[q]] = rb[q];
        }
        __syncthreads();
        if (k0 + BK < g.k) load_tile(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a[TW], b[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) a[i] = As[kr][wm * 16 * TW + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < TW; ++j) b[j] = Bs[kr][wn * 16 * TW + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
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
    if (tw && tw[0] == '4') {
        dim3 grid((g.m + 127) / 128, (g.n + 127) / 128);
        hipLaunchKernelGGL(gemm_f64_kernel<4>, grid, dim3(256), 0, st, g);
    } else {
        dim3 grid((g.m + 63) / 64, (g.n + 63) / 64);
        hipLaunchKernelGGL(gemm_f64_kernel<2>, grid, dim3(256), 0, st, g);
    }
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace locus
