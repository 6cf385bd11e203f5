// Host side of the matrix-core sweep: per-sweep block packing of X and launch dispatch.
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "mfma_sweep_kernel.h"

namespace locus {

int launch_mfma_sweep_t1(const MfmaSweepParams&, int, dim3, dim3, size_t, hipStream_t);
int launch_mfma_sweep_t2(const MfmaSweepParams&, int, dim3, dim3, size_t, hipStream_t);

// ---------------------------------------------------------------------------
// One workgroup per block of 16 consecutive sweep steps: gathers the 16 columns of the packed
// X (128 rows at a time through LDS) and writes them in the two MFMA operand orders
//   dot operand g:        l -> X[4 g + (l >> 4)][j_{l & 15}]            (A of G = X_J' R)
//   update operand (t,kk): l -> X[16 t + (l & 15)][j_{4 kk + (l >> 4)}]  (A of R += X_J delta)
// stored as XB[b][wave][KW][64][2] (KW = k-steps of 4 rows per wave; consecutive operands paired
// per lane): the order in which wave `wave` consumes them.  Also the block's Gram matrix gram[b][t][s] = X_{j_t}' X_{j_s}.  Rows >= n are zero.
// Bandwidth-bound: reads p*ld, writes 2*p*rows_pad doubles per sweep.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_blocks_kernel(const double* __restrict__ Xp, const StepRec* __restrict__ steps,
                   double* __restrict__ XB, double* __restrict__ gram, int* __restrict__ jidx,
                   int ld, int rows_pad, int KW)
{
    __shared__ double xt[16][130];
    __shared__ int js[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 16) {
        js[tid] = steps[16 * b + tid].j;
        jidx[16 * b + tid] = js[tid];
        if (b == gridDim.x - 1) jidx[16 * (b + 1) + tid] = 0;  // readable tail (prefetch)
    }
    const size_t blk = (size_t)rows_pad * 32;  // doubles per block (both operand sets)
    const int gt = tid >> 4, gs = tid & 15;
    double gacc = 0.0;
    for (int c0 = 0; c0 < rows_pad; c0 += 128) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            const int col = e >> 7, rr = e & 127, row = c0 + rr;
            xt[col][rr] = (row < ld) ? Xp[(size_t)js[col] * ld + row] : 0.0;
        }
        __syncthreads();
        const int rows_here = min(128, rows_pad - c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = tid + 256 * i;
            const int ln = o & 63;
            const int gl = o >> 6;  // k-step inside the chunk
            if (4 * gl < rows_here) {
                // global k-step -> (wave, k-step in wave); update operands use the same index
                // (4 * tile + kk) and sit KW operands after the wave's dot operands
                const int gg = (c0 >> 2) + gl;
                const int w = gg / KW, gi = gg - w * KW;
                // operand o of the wave's stream -> pair o >> 1, element o & 1 of lane ln
                double* wb = XB + (size_t)b * blk + (size_t)w * KW * 128 + ln * 2;
                wb[(size_t)(gi >> 1) * 128 + (gi & 1)] = xt[ln & 15][4 * gl + (ln >> 4)];
                const int tl = gl >> 2, kk = gl & 3;  // 16-row tile inside the chunk, k-slice
                const int ou = KW + kk * (KW >> 2) + (gi >> 2);  // k-slice major, row tile minor
                wb[(size_t)(ou >> 1) * 128 + (ou & 1)] = xt[4 * kk + (ln >> 4)][16 * tl + (ln & 15)];
            }
        }
        for (int rr = 0; rr < 128; ++rr) gacc = fma(xt[gt][rr], xt[gs][rr], gacc);
    }
    gram[(size_t)b * 256 + tid] = gacc;
}

size_t mfma_rows_pad_max(int64_t n)
{
    size_t r = 64;
    while ((int64_t)r < n) r <<= 1;
    return r;
}

// bytes of the block-packed operands for a sweep over p covariates of n rows
size_t mfma_workspace_bytes(int64_t n, int64_t p)
{
    const size_t nblk = ((size_t)p + 15) / 16;
    const size_t rp = mfma_rows_pad_max(n);
    return sizeof(double) * (2 * nblk * rp * 16 + (nblk + 1) * 256) + sizeof(int) * 16 * (nblk + 1) +
           1024;
}

size_t mfma_lds_bytes(int T, int S)
{
    return sizeof(double) * (2 * (size_t)S * T * 256 + 2 * 16 * 64 + 2 * 256 + 2 * 32 + 16 * 64);
}

// cfg: C = T (tiles per workgroup), S = waves per workgroup, G = KW4 (16-row tiles per wave)
int run_mfma_sweep(const SweepParams& prm, const SweepConfig& cfg, void* xb_ws, size_t xb_bytes,
                   hipStream_t st)
{
    const int T = cfg.C, S = cfg.S, KW4 = cfg.G;
    if (T < 1 || T > 2 || (S != 4 && S != 8) ||
        (KW4 != 1 && KW4 != 2 && KW4 != 4 && KW4 != 8 && KW4 != 16)) {
        set_error("mfma sweep: bad geometry (T in 1..2, S in {4,8}, KW4 in {1,2,4,8,16})");
        return LOCUS_HIP_EINVAL;
    }
    const int rows_pad = 16 * KW4 * S;
    if (rows_pad < prm.n) {
        set_error("mfma sweep: geometry does not cover n rows");
        return LOCUS_HIP_EINVAL;
    }
    const int nblk = (prm.nsteps + 15) / 16;
    const size_t xb = (size_t)nblk * rows_pad * 32;  // doubles of the packed operand stream
    if (sizeof(double) * (xb + ((size_t)nblk + 1) * 256) + sizeof(int) * 16 * ((size_t)nblk + 1) +
            512 > xb_bytes) {
        set_error("mfma sweep: workspace too small");
        return LOCUS_HIP_EINVAL;
    }
    char* base = static_cast<char*>(xb_ws);
    base += (256 - ((uintptr_t)base & 255)) & 255;
    double* XB = reinterpret_cast<double*>(base);
    double* gram = XB + xb;                                  // [nblk + 1][256] (tail: prefetch)
    int* jidx = reinterpret_cast<int*>(gram + ((size_t)nblk + 1) * 256);

    hipLaunchKernelGGL(pack_blocks_kernel, dim3(nblk), dim3(256), 0, st, prm.Xp, prm.steps, XB, gram,
                       jidx, prm.ld, rows_pad, 4 * KW4);
    LOCUS_HIP_CHECK(hipGetLastError());

    MfmaSweepParams mp;
    mp.n = prm.n; mp.p = prm.p; mp.d = prm.d;
    mp.nblk = nblk; mp.nsteps = prm.nsteps; mp.rows_pad = rows_pad;
    mp.steps = prm.steps; mp.err = prm.err;
    mp.XB = XB; mp.gram = gram; mp.jidx = jidx;
    mp.scale = prm.scale; mp.hinv = prm.hinv; mp.cst = prm.cst; mp.s2k = prm.s2k;
    mp.cfac = prm.cfac;
    mp.R = prm.R; mp.state = prm.state; mp.st = prm.st;

    const long n_wg = ((long)prm.d + 16L * T - 1) / (16L * T);
    dim3 grid((unsigned)n_wg), block((unsigned)(64 * S));
    const size_t lds = mfma_lds_bytes(T, S);
    switch (T) {
        case 1: return launch_mfma_sweep_t1(mp, KW4, grid, block, lds, st);
        default: return launch_mfma_sweep_t2(mp, KW4, grid, block, lds, st);
    }
}

}  // namespace locus
