// Device side of the binary-response drivers (logit: R/locus_logistic_core.R, probit:
// R/locus_probit_core.R): the n x d and p x d element-wise updates and reductions that
// surround the sweep.  All kernels are HBM-bound single passes over their operands;
// the X^2 contractions they feed run on the fp64 matrix cores (gemm.hip).
#include "common.h"
#include "kernels.h"

namespace locus {

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Sum over a 256-thread block; result valid in thread 0.  Fixed order => deterministic.
__device__ __forceinline__ double block_sum_256(double v, double* red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // red may still be read from a previous call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (threadIdx.x == 0) ? ((red[0] + red[1]) + (red[2] + red[3])) : 0.0;
}

// log(1 / (1 + exp(-chi))), chi >= 0 (R/utils.R:158-162)
__device__ __forceinline__ double log_sigmoid(double chi) { return -log(1.0 + exp(-chi)); }

// log Phi(x) (R: pnorm(x, log.p = TRUE)), accurate in both tails.
__device__ __forceinline__ double log_ndtr(double x)
{
    const double rs2 = 0.70710678118654752440;
    if (x < -1.0) {
        const double y = -x * rs2;  // Phi(x) = erfc(y) / 2 = erfcx(y) exp(-y^2) / 2
        return log(0.5 * erfcx(y)) - y * y;
    }
    if (x < 0.0) return log(0.5 * erfc(-x * rs2));
    return log1p(-0.5 * erfc(x * rs2));
}

}  // namespace

// ---------------------------------------------------------------------------
// D[j][k] = m2_beta - beta_vb^2 = gam (mu^2 + s2) - (gam mu)^2 on the [p][d] planes: the
// right-hand factor of every X^2 contraction of the binary drivers
// (update_chi_vb_ R/update_vb.R:163-167, e_y_logit_ R/elbo.R:210-218, e_y_probit_ :221-237).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) m2_minus_b2_kernel(const double* __restrict__ state,
                                                          const double* __restrict__ s2jk,
                                                          double s2c, double* __restrict__ D,
                                                          size_t pd)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < pd;
         e += (size_t)gridDim.x * blockDim.x) {
        const double g = state[e], m = state[pd + e];
        const double s2 = s2jk ? s2jk[e] : s2c;
        const double b = g * m;
        D[e] = (m * m + s2) * g - b * b;
    }
}

int m2_minus_b2(const double* state, const double* s2jk, double s2c, double* D, size_t pd,
                hipStream_t st)
{
    const unsigned blocks = (unsigned)std::min<size_t>((pd + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(m2_minus_b2_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, state, s2jk,
                       s2c, D, pd);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// s2 = 1 / (xwx + sig2_inv)   (update_sig2_beta_logit_vb_, R/update_vb.R:107-111, with
// xwx = crossprod(X^2, 2 psi) from the matrix cores)
__global__ void __launch_bounds__(256) s2_from_xwx_kernel(const double* __restrict__ xwx,
                                                          double sig2_inv, double* __restrict__ s2,
                                                          size_t len)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < len;
         e += (size_t)gridDim.x * blockDim.x)
        s2[e] = 1.0 / (xwx[e] + sig2_inv);
}

int s2_from_xwx(const double* xwx, double sig2_inv, double* s2, size_t len, hipStream_t st)
{
    const unsigned blocks = (unsigned)std::min<size_t>((len + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(s2_from_xwx_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, xwx,
                       sig2_inv, s2, len);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// out = log(in) / 2: log(sig2_beta_vb) / 2 of src/coreLoop.cpp:125, tabulated once per iteration
// for the whole p x d matrix (HBM-bound) instead of inside the latency-bound sweep
__global__ void __launch_bounds__(256) half_log_kernel(const double* __restrict__ in,
                                                       double* __restrict__ out, size_t len)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < len;
         e += (size_t)gridDim.x * blockDim.x)
        out[e] = 0.5 * log(in[e]);
}

int half_log(const double* in, double* out, size_t len, hipStream_t st)
{
    const unsigned blocks = (unsigned)std::min<size_t>((len + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(half_log_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, in, out, len);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Logit, one workgroup per response column k:
//   c2      = Q + M^2 + Z^2 sig2_alpha + Zmu^2 + 2 M Zmu          (chi_vb^2, update_vb.R:165-166)
//   ey[k]   = sum_i log_sigmoid(chi_old) + Y (M + Zmu) - chi_old / 2 - psi_old (c2 - chi_old^2)
//             (e_y_logit_, R/elbo.R:213-216: its bracket is c2, because
//              Z^2 m2_alpha - Z^2 alpha^2 = Z^2 sig2_alpha)                  [skipped if ey == NULL]
//   chi_new = sqrt(c2);  W2_new = 2 psi(chi_new)                   (update_vb.R:170-174)
// Q = X^2 (m2_beta - beta^2) comes from the GEMM.  The chi of iteration t+1 and the ELBO of
// iteration t use the same c2, so ONE contraction per iteration serves both.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
logit_chi_psi_kernel(const double* __restrict__ Q, const double* __restrict__ M,
                     const double* __restrict__ Zmu, const double* __restrict__ Y,
                     const double* __restrict__ Z, const double* __restrict__ s2a,
                     const double* __restrict__ chi_old, const double* __restrict__ W2_old,
                     double* __restrict__ chi_new, double* __restrict__ W2_new,
                     double* __restrict__ ey, int n, int d, int q)
{
    __shared__ double red[4];
    const int k = blockIdx.x;
    const size_t base = (size_t)k * n;
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) {
        const size_t e = base + r;
        double zs = 0.0;
        for (int i = 0; i < q; ++i) {
            const double z = Z[(size_t)i * n + r];
            zs = fma(z * z, s2a[(size_t)i * d + k], zs);
        }
        const double m = M[e], zm = Zmu[e];
        const double c2 = Q[e] + m * m + zs + zm * zm + 2.0 * m * zm;
        if (ey) {
            const double co = chi_old[e], po = 0.5 * W2_old[e];
            acc += log_sigmoid(co) + Y[e] * (m + zm) - co / 2.0 - po * (c2 - co * co);
        }
        const double c = sqrt(c2);
        chi_new[e] = c;
        // exp(log(exp(log_sigmoid(chi)) - 1/2) - log(2 chi)) = (sigmoid(chi) - 1/2) / (2 chi)
        const double sg = 1.0 / (1.0 + exp(-c));
        W2_new[e] = 2.0 * ((sg - 0.5) / (2.0 * c));
    }
    if (ey) {
        const double tot = block_sum_256(acc, red);
        if (threadIdx.x == 0) ey[k] = tot;
    }
}

int logit_chi_psi(const double* Q, const double* M, const double* Zmu, const double* Y,
                  const double* Z, const double* s2a, const double* chi_old, const double* W2_old,
                  double* chi_new, double* W2_new, double* ey, int n, int d, int q, hipStream_t st)
{
    hipLaunchKernelGGL(logit_chi_psi_kernel, dim3(d), dim3(256), 0, st, Q, M, Zmu, Y, Z, s2a,
                       chi_old, W2_old, chi_new, W2_new, ey, n, d, q);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// sig2_alpha[i][k] = 1 / (sum_r Z_ri^2 W2_rk + zeta2_inv[i])
// (update_sig2_alpha_logit_vb_, R/update_vb.R:52-56); one workgroup per column k.
__global__ void __launch_bounds__(256)
logit_sig2_alpha_kernel(const double* __restrict__ Z, const double* __restrict__ W2,
                        const double* __restrict__ zeta2_inv, double* __restrict__ s2a, int n,
                        int d, int q)
{
    __shared__ double red[4];
    const int k = blockIdx.x;
    for (int i = 0; i < q; ++i) {
        double acc = 0.0;
        for (int r = threadIdx.x; r < n; r += 256) {
            const double z = Z[(size_t)i * n + r];
            acc = fma(z * z, W2[(size_t)k * n + r], acc);
        }
        const double tot = block_sum_256(acc, red);
        if (threadIdx.x == 0) s2a[(size_t)i * d + k] = 1.0 / (tot + zeta2_inv[i]);
    }
}

int logit_sig2_alpha(const double* Z, const double* W2, const double* zeta2_inv, double* s2a,
                     int n, int d, int q, hipStream_t st)
{
    hipLaunchKernelGGL(logit_sig2_alpha_kernel, dim3(d), dim3(256), 0, st, Z, W2, zeta2_inv, s2a, n,
                       d, q);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// One covariate of the alpha loop (R/locus_logistic_core.R:79-85, R/locus_probit_core.R:76-84),
// one workgroup per response column k:
//   zm      = Zmu - Z_i alpha_old
//   alpha_i = sig2_alpha_ik * sum_r Z_ri (Yw - W2 (zm + M))        (W2 == NULL: probit, W2 = 1,
//                                                                   Yw = W, order W - zm - M)
//   Zmu     = zm + Z_i alpha_i
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
alpha_step_kernel(const double* __restrict__ Yw, const double* __restrict__ W2,
                  const double* __restrict__ M, double* __restrict__ Zmu,
                  const double* __restrict__ Zi, const double* __restrict__ s2a_i,
                  double* __restrict__ alpha_i, int n)
{
    __shared__ double red[4];
    __shared__ double a_new_s;
    const int k = blockIdx.x;
    const size_t base = (size_t)k * n;
    const double a_old = alpha_i[k];
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) {
        const size_t e = base + r;
        const double z = Zi[r];
        const double zm = Zmu[e] - z * a_old;
        const double t = W2 ? (Yw[e] - W2[e] * (zm + M[e])) : ((Yw[e] - zm) - M[e]);
        acc = fma(t, z, acc);
    }
    const double tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        a_new_s = s2a_i[k] * tot;
        alpha_i[k] = a_new_s;
    }
    __syncthreads();
    const double a_new = a_new_s;
    for (int r = threadIdx.x; r < n; r += 256) {
        const size_t e = base + r;
        const double z = Zi[r];
        Zmu[e] = (Zmu[e] - z * a_old) + z * a_new;
    }
}

int alpha_step(const double* Yw, const double* W2, const double* M, double* Zmu, const double* Zi,
               const double* s2a_i, double* alpha_i, int n, int d, hipStream_t st)
{
    hipLaunchKernelGGL(alpha_step_kernel, dim3(d), dim3(256), 0, st, Yw, W2, M, Zmu, Zi, s2a_i,
                       alpha_i, n);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Probit, one workgroup per response column k, U = M + Zmu:
//   W     = U + inv_mills_ratio(Y, U)                  (update_W_probit_, R/update_vb.R:382-386,
//                                                       inv_mills_ratio_matrix_ R/utils.R:178-198)
//   ey[k] = sum_i Y log Phi(U) + (1 - Y) log Phi(-U) - Q / 2 - zs_i / 2
//           (e_y_probit_, R/elbo.R:221-237; Q = X^2 (m2_beta - beta^2), zs = Z^2 sig2_alpha)
// Q == NULL skips the ELBO part (initial W, R/locus_probit_core.R:30).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
probit_w_kernel(const double* __restrict__ Y, const double* __restrict__ M,
                const double* __restrict__ Zmu, const double* __restrict__ Q,
                const double* __restrict__ zs, double* __restrict__ W, double* __restrict__ ey,
                int n)
{
    __shared__ double red[4];
    const int k = blockIdx.x;
    const size_t base = (size_t)k * n;
    const double lsq2pi = 0.91893853320467274178;  // log(sqrt(2 pi))
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) {
        const size_t e = base + r;
        const double u = M[e] + Zmu[e];
        const double y = Y[e];
        const double lp = log_ndtr(u), lq = log_ndtr(-u);
        const double dl = -u * u / 2.0 - lsq2pi;  // dnorm(u, log = TRUE)
        double imr;
        if (y == 1.0) {
            imr = exp(dl - lp);
            if (imr < -u) imr = -u;
        } else {
            imr = -exp(dl - lq);
            if (imr > -u) imr = -u;
        }
        W[e] = u + imr;
        if (Q) acc += y * lp + (1.0 - y) * lq - Q[e] / 2.0 - zs[r] / 2.0;
    }
    if (Q) {
        const double tot = block_sum_256(acc, red);
        if (threadIdx.x == 0) ey[k] = tot;
    }
}

int probit_w(const double* Y, const double* M, const double* Zmu, const double* Q, const double* zs,
             double* W, double* ey, int n, int d, hipStream_t st)
{
    hipLaunchKernelGGL(probit_w_kernel, dim3(d), dim3(256), 0, st, Y, M, Zmu, Q, zs, W, ey, n);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Row statistics of the binary drivers, one wave per covariate j (fixed order):
//   rs_gam[j] = sum_k gam                    ent[j] = sum_k gam log(gam+eps) + (1-gam) log(1-gam+eps)
//   gls[j]    = sum_k gam log(s2_jk)         (s2jk == NULL: gls = log(s2c) * rs_gam)
//   m2[j]     = sum_k gam (mu^2 + s2)        -- the O(p d) sums of e_beta_gamma_bin_ (R/elbo.R:77-87),
//                                               update_nu_bin_vb_ (R/update_vb.R:210)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bin_rows_kernel(const double* __restrict__ state, const double* __restrict__ s2jk, double s2c,
                double* __restrict__ rs_gam, double* __restrict__ ent, double* __restrict__ gls,
                double* __restrict__ m2, int p, int d, double eps)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= p) return;
    const size_t pd = (size_t)p * d;
    const double* grow = state + (size_t)wave * d;
    const double* mrow = grow + pd;
    const double* srow = s2jk ? s2jk + (size_t)wave * d : nullptr;
    const double ls2c = log(s2c);
    double s = 0.0, h = 0.0, gl = 0.0, sm = 0.0;
    for (int k = lane; k < d; k += 64) {
        const double g = grow[k], m = mrow[k];
        const double s2 = srow ? srow[k] : s2c;
        s += g;
        h += g * log(g + eps) + (1.0 - g) * log(1.0 - g + eps);
        gl = fma(g, srow ? log(s2) : ls2c, gl);
        sm = fma(g, fma(m, m, s2), sm);
    }
    s = wave_sum(s);
    h = wave_sum(h);
    gl = wave_sum(gl);
    sm = wave_sum(sm);
    if (lane == 0) {
        rs_gam[wave] = s;
        ent[wave] = h;
        gls[wave] = gl;
        m2[wave] = sm;
    }
}

int bin_rows(const double* state, const double* s2jk, double s2c, double* rs_gam, double* ent,
             double* gls, double* m2, int p, int d, double eps, hipStream_t st)
{
    const int wpb = 4;
    hipLaunchKernelGGL(bin_rows_kernel, dim3((p + wpb - 1) / wpb), dim3(64 * wpb), 0, st, state,
                       s2jk, s2c, rs_gam, ent, gls, m2, p, d, eps);
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace locus
