// Small bandwidth-bound helper kernels around the sweep (layout changes, the
// per-sweep prologue, the gamma reductions that update_vb.R / elbo.R need).
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "fastmath.h"

namespace locus {

// ---------------------------------------------------------------------------
// X (n x p, column-major, unpadded) -> Xp [p][ld] zero padded, xtx[j] = X_j'X_j
// One workgroup per column; coalesced 8-byte accesses.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_x_kernel(const double* __restrict__ X,
                                                     double* __restrict__ Xp,
                                                     double* __restrict__ xtx, int n, int ld)
{
    const size_t j = blockIdx.x;
    const double* src = X + j * (size_t)n;
    double* dst = Xp + j * (size_t)ld;
    double acc = 0.0;
    for (int r = threadIdx.x; r < ld; r += blockDim.x) {
        const double v = (r < n) ? src[r] : 0.0;
        dst[r] = v;
        acc = fma(v, v, acc);
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) xtx[j] = red[0];
}

int pack_x(const double* X, double* Xp, double* xtx, int n, int p, int ld, hipStream_t st)
{
    hipLaunchKernelGGL(pack_x_kernel, dim3(p), dim3(256), 0, st, X, Xp, xtx, n, ld);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Sweep prologue: gather the per-covariate quantities in sweep order and check
// that shuffled_ind is a permutation prefix (every index in range, none twice).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) build_steps_kernel(const double* __restrict__ shuffled_ind,
                                                          const double* __restrict__ lo,
                                                          const double* __restrict__ l1,
                                                          const double* __restrict__ xtx,
                                                          StepRec* __restrict__ steps,
                                                          int* __restrict__ visited,
                                                          int* __restrict__ err, int p, int nsteps,
                                                          int ntotal)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntotal) return;
    StepRec s;
    s.lod = 0.0;
    s.xtx = 0.0;
    s.j = 0;
    s.pad_ = 0;
    if (i < nsteps) {
        const double v = shuffled_ind[i];
        int j = (int)v;  // truncation, as `int j = shuffled_ind[i]` (src/coreLoop.cpp:38)
        if (!(v >= 0.0) || !(v < (double)p) || j < 0 || j >= p) {
            atomicOr(err, 1);
            j = 0;
        } else if (atomicAdd(&visited[j], 1) != 0) {
            atomicOr(err, 2);
        }
        s.j = j;
        s.lod = (lo != nullptr) ? l1[j] - lo[j] : 0.0;   // packing ahead: the log-odds come later
        s.xtx = xtx[j];
    }
    steps[i] = s;
}

int build_steps(const double* shuffled_ind, const double* lo, const double* l1, const double* xtx,
                StepRec* steps, int* visited, int* err, int p, int nsteps, int ntotal,
                hipStream_t st)
{
    LOCUS_HIP_CHECK(hipMemsetAsync(visited, 0, sizeof(int) * (size_t)p, st));
    LOCUS_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), st));
    hipLaunchKernelGGL(build_steps_kernel, dim3((ntotal + 255) / 256), dim3(256), 0, st,
                       shuffled_ind, lo, l1, xtx, steps, visited, err, p, nsteps, ntotal);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Residual <-> mat_x_m1 (all n x d column-major, element-wise).
//   identity family: R = Y - M - Zmu          M = Y - R - Zmu
//   logit:           R = Y - W2 (M + Zmu)     M = (Y - R) / W2 - Zmu   (W2 = 2 psi > 0)
// Zmu / W2 may be null.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) residual_kernel(const double* __restrict__ Y,
                                                       const double* __restrict__ A,
                                                       const double* __restrict__ Zmu,
                                                       const double* __restrict__ W2,
                                                       double* __restrict__ out, size_t len,
                                                       int to_residual)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < len;
         e += (size_t)gridDim.x * blockDim.x) {
        const double y = Y[e], a = A[e];
        const double z = Zmu ? Zmu[e] : 0.0;
        double o;
        if (W2) {
            const double w = W2[e];
            o = to_residual ? (y - w * (a + z)) : ((y - a) / w - z);
        } else {
            o = y - a - z;  // same form both ways
        }
        out[e] = o;
    }
}

int residual_transform(const double* Y, const double* A, const double* Zmu, const double* W2,
                       double* out, size_t len, int to_residual, hipStream_t st)
{
    const unsigned blocks = (unsigned)std::min<size_t>((len + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(residual_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, Y, A, Zmu, W2,
                       out, len, to_residual);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// out[c * rows + r] <-> in[r * cols + c] : 32x32 tiles through LDS (padded).
// transpose(in[rows_in][cols_in]) -> out[cols_in][rows_in], row-major both.
// R's p x d column-major matrix is a row-major [d][p]; the sweep wants [p][d].
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_kernel(const double* __restrict__ in,
                                                        double* __restrict__ out, int rows,
                                                        int cols, int by0)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = (by0 + blockIdx.y) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = by + i, c = bx + tx;
        if (r < rows && c < cols) tile[i][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int r = bx + i, c = by + tx;  // output row = input col
        if (r < cols && c < rows) out[(size_t)r * rows + c] = tile[tx][i];
    }
}

int transpose(const double* in, double* out, int rows, int cols, hipStream_t st)
{
    // grid.y is limited to 65535 blocks: more than 2 097 120 input rows go in several launches
    const int tiles_y = (rows + 31) / 32;
    for (int by0 = 0; by0 < tiles_y; by0 += 65535) {
        dim3 grid((cols + 31) / 32, std::min(65535, tiles_y - by0));
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, in, out, rows, cols, by0);
        LOCUS_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// log(w) for w in (0, 1 + eps]: most inclusion probabilities are tiny, so 1 - gam + eps sits just
// below 1, where log(w) = log1p(u), u = w - 1 (exact), is a short alternating series: u P(u) with P
// of degree 9 is accurate to 2^-63 relative for |u| <= 2^-6 -- 11 instructions instead of the ~35
// of log_pos_normal, which the kernel is bound by.  Same argument w as the reference's
// log(1 - gam + eps) (R/elbo.R:52), so only the last-bit rounding of the logarithm differs.
__device__ __forceinline__ double log_upto_one(double w)
{
    const double u = w - 1.0;
    if (u > -0.015625) {
        double q = -0.1;
        q = fma(q, u, 1.0 / 9.0);
        q = fma(q, u, -0.125);
        q = fma(q, u, 1.0 / 7.0);
        q = fma(q, u, -1.0 / 6.0);
        q = fma(q, u, 0.2);
        q = fma(q, u, -0.25);
        q = fma(q, u, 1.0 / 3.0);
        q = fma(q, u, -0.5);
        q = fma(q, u, 1.0);
        return u * q;
    }
    return log_pos_normal(w);
}

// ---------------------------------------------------------------------------
// Row statistics of gam [p][d] (response-contiguous): one wave per covariate.
//   rs_gam[j]  = sum_k gam_jk                         (R/locus_core.R:95)
//   ent[j]     = sum_k gam log(gam + eps) + (1 - gam) log(1 - gam + eps)
//                (the only O(p d) term of e_beta_gamma_, R/elbo.R:52)
// Deterministic: fixed lane striding + butterfly.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gamma_rows_kernel(const double* __restrict__ gam,
                                                         double* __restrict__ rs_gam,
                                                         double* __restrict__ ent, int p, int d,
                                                         double eps)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= p) return;
    const double* row = gam + (size_t)wave * d;
    // gam in [0, 1] and eps > 0: both log arguments are positive normal numbers, which is all
    // log_pos_normal handles (the library log spends half its instructions on special cases)
    double s0 = 0.0, s1 = 0.0, h0 = 0.0, h1 = 0.0;
    int k = lane;
    for (; k + 64 < d; k += 128) {  // two independent element streams per lane
        const double g0 = row[k], g1 = row[k + 64];
        s0 += g0;
        s1 += g1;
        if (ent) {
            h0 = fma(g0, log_pos_normal(g0 + eps), fma(1.0 - g0, log_upto_one(1.0 - g0 + eps), h0));
            h1 = fma(g1, log_pos_normal(g1 + eps), fma(1.0 - g1, log_upto_one(1.0 - g1 + eps), h1));
        }
    }
    if (k < d) {
        const double g0 = row[k];
        s0 += g0;
        if (ent) h0 = fma(g0, log_pos_normal(g0 + eps), fma(1.0 - g0, log_upto_one(1.0 - g0 + eps), h0));
    }
    double s = s0 + s1, h = h0 + h1;
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        h += __shfl_xor(h, off);
    }
    if (lane == 0) {
        rs_gam[wave] = s;
        if (ent) ent[wave] = h;
    }
}

int gamma_rows(const double* gam, double* rs_gam, double* ent, int p, int d, double eps,
               hipStream_t st)
{
    const int waves_per_block = 4;
    hipLaunchKernelGGL(gamma_rows_kernel, dim3((p + waves_per_block - 1) / waves_per_block),
                       dim3(64 * waves_per_block), 0, st, gam, rs_gam, ent, p, d, eps);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Column statistics of the state [3][p][d] without a sweep (initialisation,
// R/locus_core.R:36-42): sum_j gam, sum_j gam (mu^2 + s2_k), sum_j beta^2 where
// beta = gam * mu is (re)written to plane 2.  One thread per response,
// coalesced across k, sequential in j (deterministic).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) state_cols_kernel(double* __restrict__ state,
                                                         const double* __restrict__ s2k,
                                                         const double* __restrict__ s2jk,
                                                         double* __restrict__ st, int p, int d,
                                                         int jchunk)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j0 = blockIdx.y * jchunk;
    const int j1 = min(p, j0 + jchunk);
    if (k >= d) return;
    const size_t pd = (size_t)p * d;
    double sg = 0.0, sm2 = 0.0, sb2 = 0.0;
    const double s2c = s2k ? s2k[k] : 0.0;
    for (int j = j0; j < j1; ++j) {
        const size_t off = (size_t)j * d + k;
        const double g = state[off], m = state[pd + off];
        const double b = g * m;
        state[2 * pd + off] = b;
        const double s2 = s2jk ? s2jk[off] : s2c;
        sg += g;
        sm2 = fma(g, fma(m, m, s2), sm2);
        sb2 = fma(b, b, sb2);
    }
    // partial sums per j-chunk: st layout [gridDim.y][3][d]
    double* o = st + (size_t)blockIdx.y * 3 * d;
    o[k] = sg;
    o[(size_t)d + k] = sm2;
    o[2 * (size_t)d + k] = sb2;
}

int state_cols(double* state, const double* s2k, const double* s2jk, double* st_partial, int p,
               int d, int nchunks, hipStream_t st)
{
    const int jchunk = (p + nchunks - 1) / nchunks;
    dim3 grid((d + 255) / 256, nchunks);
    hipLaunchKernelGGL(state_cols_kernel, grid, dim3(256), 0, st, state, s2k, s2jk, st_partial, p,
                       d, jchunk);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Element-wise part of coreBatch / coreZBatch / coreProbitBatch on the [p][d]
// planes (src/coreBatch.cpp:32-39, :63-70, :91-97).  On entry the mu plane holds
// T = X'(Y - M [- Zmu]); beta still holds the OLD values.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) batch_update_kernel(double* __restrict__ state,
                                                           const double* __restrict__ lo,
                                                           const double* __restrict__ l1,
                                                           const double* __restrict__ fac,
                                                           const double* __restrict__ hinv,
                                                           const double* __restrict__ cstk,
                                                           double log_sig2_inv, double nm1,
                                                           double cfac, int probit, int p, int d)
{
    const size_t pd = (size_t)p * d;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < pd;
         e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e / d), k = (int)(e - (size_t)j * d);
        const double t = state[pd + e] + nm1 * state[2 * pd + e];
        const double m = probit ? fac[k] * t : t * fac[k];
        const double q = -(m * m) * hinv[k];
        const double lod = l1[j] - lo[j];
        // cfac: annealing factor of the internal scheme "x" (R/locus_core.R:104-111: c multiplies
        // the whole argument; fac already carries it); coreBatch itself has none (cfac == 1)
        const double a = cfac * (probit ? (((q + lod) - log_sig2_inv / 2) - cstk[k])
                                        : (((q - cstk[k]) + lod) - log_sig2_inv / 2));
        const double ex = exp(-fabs(a));
        const double g = ((a >= 0.0) ? ex : 1.0) / (1.0 + ex);
        state[e] = g;
        state[pd + e] = m;
        state[2 * pd + e] = m * g;
    }
}

int batch_update(double* state, const double* lo, const double* l1, const double* fac,
                 const double* hinv, const double* cstk, double log_sig2_inv, double nm1,
                 double cfac, int probit, int p, int d, hipStream_t st)
{
    const size_t pd = (size_t)p * d;
    const unsigned blocks = (unsigned)std::min<size_t>((pd + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(batch_update_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, state, lo,
                       l1, fac, hinv, cstk, log_sig2_inv, nm1, cfac, probit, p, d);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace locus
