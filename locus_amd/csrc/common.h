// Shared declarations of liblocus_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/locus_hip.h"  // error codes

namespace locus {

// Last error text of the calling thread (returned by locus_hip_last_error()).
void set_error(const std::string& msg);
const char* get_error();

#define LOCUS_HIP_CHECK(expr)                                                         \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            ::locus::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));    \
            return (int)_e;                                                           \
        }                                                                             \
    } while (0)

#define LOCUS_REQUIRE(cond, msg)                                                      \
    do {                                                                              \
        if (!(cond)) {                                                                \
            ::locus::set_error(std::string("invalid argument: ") + (msg));            \
            return LOCUS_HIP_EINVAL;                                                  \
        }                                                                             \
    } while (0)


// ---------------------------------------------------------------------------
// Sweep kernel parameters (device pointers).  See sweep.hip for the layout
// contract; DESIGN.md §3 for the rationale.
// ---------------------------------------------------------------------------
// One covariate visit of a sweep, gathered in sweep order by the prologue kernel.
struct StepRec {
    double lod;  // log_1_min_om_vb[j] - log_om_vb[j]  (struct: log(1-Phi) - log Phi)
    double xtx;  // X[,j]' X[,j]
    int j;       // covariate index
    int pad_;
};

struct SweepParams {
    int n, p, d;        // rows, covariates, local responses
    int ld;             // leading dimension (doubles) of the packed X: multiple of 128, >= n
    int S, G;           // waves per column (row slices), column groups per workgroup
    int npad;           // 64 * NPL * S  (>= ld)
    int nsteps;         // covariates visited this sweep (== p for a full sweep)
    int nsteps_pad;     // nsteps rounded up to a multiple of 12 (every prefetch depth)
    const double* Xp;   // [p][ld]   packed, zero padded columns of X
    const StepRec* steps;  // [nsteps_pad + 16]  (pad entries: j = 0, lod = xtx = 0)
    const int* err;        // device error word written by the prologue (non-zero => abort)
    // identity family: per-response constants
    const double* scale;  // [d]  (c * sig2_beta_k) * tau_k      (probit: sig2_beta)
    const double* hinv;   // [d]  1 / (2 sig2_beta_k)
    const double* cst;    // [d]  -(log_tau_k + log_sig2_inv + log sig2_beta_k) / 2
    const double* s2k;    // [d]  sig2_beta_k (for the fused m2_beta column sums)
    double cfac;          // annealing factor c
    // logit
    const double* s2;     // [p][d]  sig2_beta_vb (response-contiguous)
    const double* xwx;    // [p][d]  sum_i X_ij^2 * 2 psi_ik
    const double* W2;     // [n x d] column-major: 2 * psi_vb
    const double* ls2;    // [p][d]  log(sig2_beta_vb) / 2, or nullptr: the kernel takes the log
    double half_lsi;      // log_sig2_inv_vb / 2
    // in/out
    double* R;            // [n x d] column-major residual
    double* state;        // [3][p][d]: planes gam_vb, mu_beta_vb, beta_vb, response-contiguous
    // fused per-response statistics, [4][d]: sum_j gam, sum_j gam (mu^2 + sig2_beta),
    // sum_j beta^2, sum_i R_ik^2 (after the sweep); nullptr => skipped
    double* st;
};

// Kernel geometry chosen by the host for (n, d, variant).
struct SweepConfig {
    int NPL;  // rows per lane (2,4,8,16)
    int C;    // responses per wave
    int S;    // waves per column
    int G;    // column groups per workgroup
};

// NPL == -1 selects the matrix-core kernel (mfma_sweep_kernel.h); then C = tiles of 16 responses
// per workgroup (1, 2; 0 = planned from d), S is ignored, G = 16-row tiles per matrix wave (0 =
// smallest instantiated value that covers n).
int choose_sweep_config(int n, int d, bool logit, SweepConfig* cfg);
int device_cus();  // CUs of the current device (LOCUS_HIP_MFMA_CUS overrides: tests)
int real_cus();    // CUs of the current device as the runtime reports them
size_t mfma_workspace_bytes(int64_t n, int64_t p);
bool mfma_sweep_covers(int n);
int run_mfma_sweep(const SweepParams& prm, const SweepConfig& cfg, void* xb_ws, size_t xb_bytes,
                   hipStream_t st, int mode = 0);
int mfma_plan_regions(int n, int d, const SweepConfig& cfg);
int mfma_plan_rows(int n, int d, const SweepConfig& cfg, int32_t* out, int max_rows);
int launch_sweep(const SweepParams& prm, const SweepConfig& cfg, bool logit, hipStream_t stream);
// NPL == -2 selects the streaming kernel (stream_sweep.hip): any n, both families; what the
// chooser falls back to when n exceeds the register-resident kernels.
int run_stream_sweep(const SweepParams& prm, bool logit, hipStream_t stream);
size_t sweep_lds_bytes(const SweepConfig& cfg);
// NPL == -3 selects the logit look-ahead kernel (logit_la.hip): latency regime of the logit family
// (one workgroup per CU); then C = responses per workgroup (1, 2, 4; 0 = from d), G = rows per
// lane (0 = smallest instantiated value that covers n).
int logit_la_max_cr(int n);  // largest C for n rows; 0 = n not covered (n > 3072)
int run_logit_la_sweep(const SweepParams& prm, const SweepConfig& cfg, hipStream_t stream);

}  // namespace locus
