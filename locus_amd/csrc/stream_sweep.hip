// CAVI sweep for n beyond the register-resident kernels (n > 4096): one workgroup per response, the residual column in LDS
// when it fits (else updated in place in global memory), X columns read through L2, which all the
// workgroups share because they advance through the same sweep order.
//
// Same update as sweep_kernel.h: src/coreLoop.cpp:36-52, :78-94, :115-131, :154-169,
// src/coreStructLoop.cpp:34-50, with the remove / dot / add-back triple folded as in
// src/coreBatch.cpp:32-33.  Not a tuned kernel: one barrier and ~1 us per step; it exists so that
// the library has no limit on the number of samples.
#include "common.h"
#include "fastmath.h"

namespace locus {

constexpr int kStreamThreads = 256;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// RPT > 0: n <= 256 RPT and every thread keeps its RPT values of the current X column in
// registers, loaded one step ahead (the HBM / L2 latency of a column is off the critical path
// and each value is read once); RPT == 0: any n, columns read in place, twice.
template <bool LOGIT, int RPT>
__global__ void __launch_bounds__(kStreamThreads)
stream_sweep_kernel(const SweepParams P, const int in_lds, const int n_al)
{
    extern __shared__ __align__(16) double smem[];
    if (*P.err != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = P.n, d = P.d, k = blockIdx.x;
    const size_t pd = (size_t)P.p * d;

    double* const red = smem;                   // [2][4] partial dot products of the waves
    double* res = P.R + (size_t)k * n;          // residual column of this response
    const double* wgt = LOGIT ? P.W2 + (size_t)k * n : nullptr;   // 2 psi column
    if (in_lds) {
        double* l_res = smem + 8;
        double* l_wgt = l_res + n_al;
        for (int i = tid; i < n; i += kStreamThreads) {
            l_res[i] = res[i];
            if constexpr (LOGIT) l_wgt[i] = wgt[i];
        }
        res = l_res;
        if constexpr (LOGIT) wgt = l_wgt;
    }
    // every thread touches the same rows i = tid + 256 m in the dot product and in the rank-1
    // update, so the residual needs no barrier of its own

    double c_scale = 0.0, c_hinv = 0.0, c_cst = 0.0, c_s2 = 0.0;
    if constexpr (!LOGIT) {
        c_scale = P.scale[k];
        c_hinv = P.hinv[k];
        c_cst = P.cst[k];
        c_s2 = P.s2k[k];
    }
    double acc_g = 0.0, acc_m2 = 0.0, acc_b2 = 0.0;

    StepRec cur = P.steps[0];
    constexpr int NX = RPT > 0 ? RPT : 1;
    double xc[NX];
    if constexpr (RPT > 0) {
        const double* x0 = P.Xp + (size_t)cur.j * P.ld;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int i = tid + u * kStreamThreads;
            xc[u] = (i < n) ? x0[i] : 0.0;
        }
    }
    size_t off = (size_t)cur.j * d + k;
    double bo = P.state[2 * pd + off], s2v = c_s2, xw = 0.0;
    if constexpr (LOGIT) {
        s2v = P.s2[off];
        xw = P.xwx[off];
    }
    for (int s = 0; s < P.nsteps; ++s) {
        // requests of the next step (the step table is padded with j = 0 records)
        const StepRec nxt = P.steps[s + 1];
        const size_t off_n = (size_t)nxt.j * d + k;
        const double bo_n = P.state[2 * pd + off_n];
        double s2_n = c_s2, xw_n = 0.0;
        if constexpr (LOGIT) {
            s2_n = P.s2[off_n];
            xw_n = P.xwx[off_n];
        }

        const double* xj = P.Xp + (size_t)cur.j * P.ld;
        double xn[NX];
        double part = 0.0;
        if constexpr (RPT > 0) {
            const double* x1 = P.Xp + (size_t)nxt.j * P.ld;
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const int i = tid + u * kStreamThreads;
                xn[u] = (i < n) ? x1[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const int i = tid + u * kStreamThreads;
                if (i < n) part = fma(xc[u], res[i], part);
            }
        } else {
            for (int i = tid; i < n; i += kStreamThreads) part = fma(xj[i], res[i], part);
        }
        part = wave_sum(part);
        double* rb = red + 4 * (s & 1);
        if (lane == 0) rb[wv] = part;
        __syncthreads();
        const double U = (rb[0] + rb[1]) + (rb[2] + rb[3]);

        double m, a;
        if constexpr (LOGIT) {
            const double dot = fma(xw, bo, U);
            m = s2v * dot;                                               // coreLoop.cpp:121
            a = cur.lod - P.half_lsi - m * m * (0.5 * rcp_pos_normal(s2v)) -
                0.5 * log_pos_normal(s2v);                               // :123-125
        } else {
            const double dot = fma(cur.xtx, bo, U);
            m = c_scale * dot;                                           // coreLoop.cpp:42-43
            a = P.cfac * ((cur.lod - m * m * c_hinv) + c_cst);           // :45-46
        }
        const double e = exp_nonpos(-fabs(a));
        const double gm = div_1to2((a >= 0.0) ? e : 1.0, 1.0 + e);
        const double bn = m * gm;                                        // :48
        const double delta = bo - bn;
        if (tid == 0) {
            double* o = P.state + off;
            o[0] = gm;
            o[pd] = m;
            o[2 * pd] = bn;
        }
        acc_g += gm;
        acc_m2 = fma(gm, fma(m, m, s2v), acc_m2);
        acc_b2 = fma(bn, bn, acc_b2);

        if constexpr (RPT > 0) {
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const int i = tid + u * kStreamThreads;
                if (i < n) {
                    if constexpr (LOGIT)
                        res[i] = fma(xc[u] * wgt[i], delta, res[i]);
                    else
                        res[i] = fma(xc[u], delta, res[i]);
                }
                xc[u] = xn[u];
            }
        } else {
            for (int i = tid; i < n; i += kStreamThreads) {
                if constexpr (LOGIT)
                    res[i] = fma(xj[i] * wgt[i], delta, res[i]);
                else
                    res[i] = fma(xj[i], delta, res[i]);
            }
        }
        cur = nxt;
        off = off_n;
        bo = bo_n;
        s2v = s2_n;
        xw = xw_n;
    }

    // residual back (if it lived in LDS) and the fused statistics
    double* out = P.R + (size_t)k * n;
    double ss = 0.0;
    for (int i = tid; i < n; i += kStreamThreads) {
        const double v = res[i];
        if (in_lds) out[i] = v;
        ss = fma(v, v, ss);
    }
    if (P.st != nullptr) {
        ss = wave_sum(ss);
        __syncthreads();
        if (lane == 0) red[wv] = ss;
        __syncthreads();
        if (tid == 0) {
            P.st[k] = acc_g;
            P.st[(size_t)d + k] = acc_m2;
            P.st[2 * (size_t)d + k] = acc_b2;
            P.st[3 * (size_t)d + k] = (red[0] + red[1]) + (red[2] + red[3]);
        }
    }
}

// Any n, any d, both families.  LDS residency of the residual column(s) up to 120 KB.
int run_stream_sweep(const SweepParams& prm, bool logit, hipStream_t st)
{
    const int n_al = (prm.n + 1) & ~1;
    const size_t cols = logit ? 2 : 1;
    size_t lds = sizeof(double) * (8 + cols * (size_t)n_al);
    int in_lds = 1;
    if (lds > 120 * 1024) {
        in_lds = 0;
        lds = sizeof(double) * 8;
    }
    dim3 grid((unsigned)prm.d), block(kStreamThreads);
    const int rows = (prm.n + kStreamThreads - 1) / kStreamThreads;  // per thread
#define LOCUS_STREAM(L, R)                                                                      \
    do {                                                                                        \
        if (lds > 64 * 1024)                                                                    \
            LOCUS_HIP_CHECK(hipFuncSetAttribute(                                                \
                reinterpret_cast<const void*>(stream_sweep_kernel<L, R>),                       \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
        hipLaunchKernelGGL((stream_sweep_kernel<L, R>), grid, block, lds, st, prm, in_lds, n_al); \
    } while (0)
    if (logit) {
        if (rows <= 8) LOCUS_STREAM(true, 8);
        else if (rows <= 16) LOCUS_STREAM(true, 16);
        else if (rows <= 32) LOCUS_STREAM(true, 32);
        else LOCUS_STREAM(true, 0);
    } else {
        if (rows <= 8) LOCUS_STREAM(false, 8);
        else if (rows <= 16) LOCUS_STREAM(false, 16);
        else if (rows <= 32) LOCUS_STREAM(false, 32);
        else LOCUS_STREAM(false, 0);
    }
#undef LOCUS_STREAM
    LOCUS_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace locus
