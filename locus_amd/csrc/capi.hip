// C ABI of liblocus_hip.so (declared in include/locus_hip.h).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/locus_hip.h"
#include "common.h"
#include "kernels.h"

namespace locus {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

namespace {

// RAII device allocation for the stateless entry points.
struct DevMem {
    void* p = nullptr;
    ~DevMem() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    int alloc(size_t bytes)
    {
        LOCUS_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
        return 0;
    }
    double* d() const { return static_cast<double*>(p); }
};

int upload(DevMem& m, const double* host, size_t count)
{
    int rc = m.alloc(count * sizeof(double));
    if (rc) return rc;
    if (count) LOCUS_HIP_CHECK(hipMemcpy(m.p, host, count * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int download(double* host, const DevMem& m, size_t count)
{
    if (count) LOCUS_HIP_CHECK(hipMemcpy(host, m.p, count * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int check_dims(int64_t n, int64_t p, int64_t d)
{
    LOCUS_REQUIRE(n >= 1 && p >= 1 && d >= 1, "n, p and d must be >= 1");
    LOCUS_REQUIRE(n < (1ll << 31) && p < (1ll << 31) && d < (1ll << 31) && n * d < (1ll << 40) &&
                      p * d < (1ll << 40),
                  "problem too large");
    return 0;
}

// The reference reads shuffled_ind[i] for i < ncol(X) (src/coreLoop.cpp:36-38); R always
// passes sample(0:(p-1)).  Anything else (out of range, repeated index) is rejected.
int check_permutation(const double* ind, int64_t p)
{
    std::vector<unsigned char> seen((size_t)p, 0);
    for (int64_t i = 0; i < p; ++i) {
        const double v = ind[i];
        if (!(v >= 0.0) || !(v < (double)p)) {
            set_error("shuffled_ind: index out of range [0, p)");
            return LOCUS_HIP_EPERM;
        }
        const int64_t j = (int64_t)v;
        if (seen[(size_t)j]) {
            set_error("shuffled_ind: repeated index (must be a permutation of 0..p-1)");
            return LOCUS_HIP_EPERM;
        }
        seen[(size_t)j] = 1;
    }
    return 0;
}

// The device error word of a sweep (WsView::err): bits 0, 1 from the prologue's permutation
// check (the sweep kernels then return at entry: state untouched), bit 2 (value 4) from a
// row-split cluster whose partner did not show up (mfma_sweep1_kernel.h: state partly written).
int decode_sweep_error(int herr)
{
    if (herr & 3) {
        set_error(herr & 1 ? "shuffled_ind: index out of range [0, p)"
                           : "shuffled_ind: repeated index (must be a permutation of 0..p-1)");
        return LOCUS_HIP_EPERM;
    }
    if (herr & 4) {
        set_error("sweep: a row-split cluster timed out waiting for a partner workgroup (its "
                  "workgroups were not resident at once); the sweep stopped part-way: state "
                  "invalid, re-initialise the residual from gam_vb / mu_beta_vb");
        return LOCUS_HIP_ETIMEOUT;
    }
    if (herr != 0) {
        set_error("sweep: unknown device error word " + std::to_string(herr));
        return LOCUS_HIP_EINVAL;
    }
    return 0;
}

// step records: p rounded up to a multiple of 48 (every prefetch depth of the vector kernel,
// whole blocks of 16 for the matrix-core kernel) plus a readable tail
size_t step_slots(int64_t p) { return ((size_t)p + 47) / 48 * 48 + 32; }

size_t small_workspace_bytes(int64_t p)
{
    size_t b = sizeof(StepRec) * step_slots(p);
    b = (b + 255) / 256 * 256;
    b += sizeof(int) * (size_t)p;  // visited[]
    b = (b + 255) / 256 * 256;
    b += 256;  // error word
    return b;
}

size_t workspace_bytes(int64_t n, int64_t p)
{
    return small_workspace_bytes(p) + mfma_workspace_bytes(n, p);
}

struct WsView {
    StepRec* steps;
    int* visited;
    int* err;
    void* xb;  // block-packed X of the matrix-core kernel
};

WsView carve(void* ws, int64_t p)
{
    char* c = static_cast<char*>(ws);
    WsView v;
    v.steps = reinterpret_cast<StepRec*>(c);
    size_t b = sizeof(StepRec) * step_slots(p);
    b = (b + 255) / 256 * 256;
    v.visited = reinterpret_cast<int*>(c + b);
    b += sizeof(int) * (size_t)p;
    b = (b + 255) / 256 * 256;
    v.err = reinterpret_cast<int*>(c + b);
    v.xb = c + b + 256;
    return v;
}

// ---- scratch of the entry point that allocates nothing (locus_hip_sweep_static_dev) ----------
// One array per GPU, part of the code object: every device the library runs on holds these
// 64 MiB from the moment the library is loaded until the process ends.  In host code the NAME
// `scratch` is only a handle for hipGetSymbolAddress; the device address differs per GPU and is
// looked up once per device (static_scratch_ptr), and only that pointer is ever handed to
// kernels, memsets and copies.
static __device__ __align__(256) unsigned char scratch[LOCUS_HIP_STATIC_WORKSPACE_BYTES];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};

// Device address of `scratch` on the CURRENT device (looked up on the first call per device).
int static_scratch_ptr(void** out)
{
    int dev = -1;
    LOCUS_HIP_CHECK(hipGetDevice(&dev));
    LOCUS_REQUIRE(dev >= 0 && dev < kMaxDevices, "static workspace: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        LOCUS_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        LOCUS_REQUIRE(ptr != nullptr, "static workspace: hipGetSymbolAddress returned NULL");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

// mode 0: a whole sweep; 1: pack the operands of the sweep that will visit `shuffled_ind`
// (locus_hip_sweep_prepack_dev: only X, the order and the shape are read); 2: the sweep itself on
// operands packed by mode 1 (args->prepacked).
int sweep_dev(const locus_hip_sweep_args* a, int logit, hipStream_t st, int mode = 0)
{
    LOCUS_REQUIRE(a != nullptr, "args is NULL");
    int rc = check_dims(a->n, a->p, a->d);
    if (rc) return rc;
    LOCUS_REQUIRE(a->Xp && a->xtx && a->shuffled_ind && a->workspace, "sweep: NULL device pointer");
    if (mode != 1) {
        LOCUS_REQUIRE(a->lo && a->l1 && a->R && a->state, "sweep: NULL device pointer");
        if (logit)
            LOCUS_REQUIRE(a->s2 && a->xwx && a->W2, "sweep: logit needs s2, xwx, W2");
        else
            LOCUS_REQUIRE(a->scale && a->hinv && a->cst && a->s2k,
                          "sweep: identity family needs scale, hinv, cst, s2k");
    }
    LOCUS_REQUIRE((size_t)a->workspace_bytes >= workspace_bytes(a->n, a->p),
                  "sweep: workspace too small");
    const int64_t nsteps = (a->nsteps <= 0 || a->nsteps > a->p) ? a->p : a->nsteps;

    SweepConfig cfg;
    if (a->cfg[0] != 0) {
        cfg = SweepConfig{a->cfg[0], a->cfg[1], a->cfg[2], a->cfg[3]};
        LOCUS_REQUIRE(cfg.NPL < 0 ? (cfg.C >= 0 && cfg.G >= 0)
                                  : (cfg.C >= 1 && cfg.S >= 1 && cfg.G >= 1 &&
                                     (int64_t)cfg.NPL * cfg.S * 64 >= a->n),
                      "sweep: bad explicit geometry");
    } else {
        rc = choose_sweep_config((int)a->n, (int)a->d, logit != 0, &cfg);
        if (rc) return rc;
    }

    WsView ws = carve(a->workspace, a->p);
    const int nsteps_pad = (int)((nsteps + 11) / 12 * 12);
    if (mode != 0 && (logit || cfg.NPL != -1)) {
        set_error("sweep: operands are packed ahead for the matrix-core sweep of the identity family only");
        return LOCUS_HIP_EUNSUPPORTED;
    }
    rc = build_steps(a->shuffled_ind, mode == 1 ? nullptr : a->lo, mode == 1 ? nullptr : a->l1, a->xtx,
                     ws.steps, ws.visited, ws.err, (int)a->p, (int)nsteps, (int)step_slots(nsteps), st);
    if (rc) return rc;

    SweepParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.n = (int)a->n;
    prm.p = (int)a->p;
    prm.d = (int)a->d;
    prm.ld = (int)locus_hip_packed_ld(a->n);
    prm.nsteps = (int)nsteps;
    prm.nsteps_pad = nsteps_pad;
    prm.Xp = a->Xp;
    prm.steps = ws.steps;
    prm.err = ws.err;
    prm.scale = a->scale;
    prm.hinv = a->hinv;
    prm.cst = a->cst;
    prm.s2k = a->s2k;
    prm.cfac = a->cfac;
    prm.s2 = a->s2;
    prm.xwx = a->xwx;
    prm.W2 = a->W2;
    prm.ls2 = a->ls2;
    prm.half_lsi = a->half_lsi;
    prm.R = a->R;
    prm.state = a->state;
    prm.st = a->stats;
    if (cfg.NPL == -2) return run_stream_sweep(prm, logit != 0, st);
    if (cfg.NPL == -3) {
        LOCUS_REQUIRE(logit, "sweep: the look-ahead kernel {-3, ..} covers the logit family only");
        return run_logit_la_sweep(prm, cfg, st);
    }
    if (cfg.NPL < 0) {
        LOCUS_REQUIRE(!logit, "sweep: the matrix-core kernel covers the identity family only");
        // (modes 1, 2: the regions of all the plan's geometries follow the small part)
        const size_t xb_bytes = mode == 0 ? mfma_workspace_bytes(a->n, a->p)
                                          : (size_t)a->workspace_bytes - small_workspace_bytes(a->p) - 256;
        return run_mfma_sweep(prm, cfg, ws.xb, xb_bytes, st, mode);
    }
    return launch_sweep(prm, cfg, logit != 0, st);
}

// Common body of the five stateless *_loop entry points.
//   kind 0: identity family with per-response sig2_beta / tau (coreLoop, coreZLoop, coreStructLoop)
//   kind 1: probit (scalar sig2_beta, no tau)
//   kind 2: logit  (psi, p x d sig2_beta)
int run_loop(int kind, const double* X, const double* Y, double* gam, const double* lo,
             const double* l1, double log_sig2_inv, const double* log_tau, double* beta, double* M,
             const double* Zmu, double* mu, const double* sig2_beta, const double* tau,
             const double* psi, const double* shuffled_ind, double c, int64_t n, int64_t p,
             int64_t d)
{
    int rc = check_dims(n, p, d);
    if (rc) return rc;
    LOCUS_REQUIRE(X && Y && gam && lo && l1 && beta && M && mu && sig2_beta && shuffled_ind,
                  "NULL pointer argument");
    if (kind == 0) LOCUS_REQUIRE(log_tau && tau, "NULL pointer argument");
    if (kind == 2) LOCUS_REQUIRE(psi && Zmu, "NULL pointer argument");
    rc = check_permutation(shuffled_ind, p);
    if (rc) return rc;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available (liblocus_hip has no CPU fallback)");
        return (int)hipErrorNoDevice;
    }

    const size_t nd = (size_t)n * d, pd = (size_t)p * d, np = (size_t)n * p;
    const int64_t ld = locus_hip_packed_ld(n);
    hipStream_t st = nullptr;

    // per-response constants on the host (d values each)
    std::vector<double> h_scale(d), h_hinv(d), h_cst(d), h_s2k(d);
    if (kind != 2) {
        for (int64_t k = 0; k < d; ++k) {
            const double s2 = (kind == 1) ? sig2_beta[0] : sig2_beta[k];
            h_s2k[k] = s2;
            h_hinv[k] = 1.0 / (2.0 * s2);
            if (kind == 1) {
                h_scale[k] = s2;                                       // coreLoop.cpp:160
                h_cst[k] = -(log_sig2_inv + std::log(s2)) / 2;         // :152
            } else {
                h_scale[k] = c * s2 * tau[k];                          // :42
                h_cst[k] = -(log_tau[k] + log_sig2_inv + std::log(s2)) / 2;  // :34
            }
        }
    }

    DevMem dX, dXp, dxtx, dY, dA, dZ, dW2, dR, dstate, dtmp, dlo, dl1, dind, dscale, dhinv, dcst,
        ds2k, ds2, dls2, dxwx, dws;
    // X: raw copy -> packed; the raw copy is released before the big state buffers are
    // allocated (logit keeps it: the X^2 contraction below reads it)
    {
        if ((rc = upload(dX, X, np))) return rc;
        if ((rc = dXp.alloc(sizeof(double) * (size_t)p * ld))) return rc;
        if ((rc = dxtx.alloc(sizeof(double) * (size_t)p))) return rc;
        if ((rc = pack_x(dX.d(), dXp.d(), dxtx.d(), (int)n, (int)p, (int)ld, st))) return rc;
        if (kind != 2) {
            LOCUS_HIP_CHECK(hipStreamSynchronize(st));
            dX.release();
        }
    }
    if ((rc = upload(dY, Y, nd))) return rc;
    if ((rc = upload(dA, M, nd))) return rc;
    if (Zmu && (rc = upload(dZ, Zmu, nd))) return rc;
    if ((rc = dR.alloc(sizeof(double) * nd))) return rc;
    if (kind == 2) {
        // W2 = 2 psi: upload psi and scale in place with the residual kernel's helpers
        std::vector<double> w2(nd);
        for (size_t e = 0; e < nd; ++e) w2[e] = 2.0 * psi[e];
        if ((rc = upload(dW2, w2.data(), nd))) return rc;
    }
    if ((rc = residual_transform(dY.d(), dA.d(), Zmu ? dZ.d() : nullptr,
                                 kind == 2 ? dW2.d() : nullptr, dR.d(), nd, 1, st)))
        return rc;

    // state planes [3][p][d] from R's column-major p x d matrices
    if ((rc = dstate.alloc(sizeof(double) * 3 * pd))) return rc;
    if ((rc = dtmp.alloc(sizeof(double) * pd))) return rc;
    const double* planes_in[3] = {gam, mu, beta};
    for (int q = 0; q < 3; ++q) {
        LOCUS_HIP_CHECK(hipMemcpy(dtmp.p, planes_in[q], sizeof(double) * pd, hipMemcpyHostToDevice));
        if ((rc = transpose(dtmp.d(), dstate.d() + q * pd, (int)d, (int)p, st))) return rc;
        LOCUS_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (kind == 2) {
        if ((rc = ds2.alloc(sizeof(double) * pd))) return rc;
        if ((rc = dxwx.alloc(sizeof(double) * pd))) return rc;
        LOCUS_HIP_CHECK(hipMemcpy(dtmp.p, sig2_beta, sizeof(double) * pd, hipMemcpyHostToDevice));
        if ((rc = transpose(dtmp.d(), ds2.d(), (int)d, (int)p, st))) return rc;
        if ((rc = dls2.alloc(sizeof(double) * pd))) return rc;
        if ((rc = half_log(ds2.d(), dls2.d(), pd, st))) return rc;
        // xwx[j][k] = sum_i X_ij^2 W2_ik : (X^2)' W2 on the matrix cores, written [p][d]
        GemmDesc g;
        g.m = (int)p; g.n = (int)d; g.k = (int)n;
        g.A = dX.d(); g.sa_m = n; g.sa_k = 1; g.sqA = 1;
        g.B = dW2.d(); g.sb_k = 1; g.sb_n = n; g.sqB = 0;
        g.C = dxwx.d(); g.sc_m = d; g.sc_n = 1;
        g.alpha = 1.0; g.beta = 0.0;
        if ((rc = gemm_f64(g, st))) return rc;
    }
    if ((rc = upload(dlo, lo, (size_t)p))) return rc;
    if ((rc = upload(dl1, l1, (size_t)p))) return rc;
    if ((rc = upload(dind, shuffled_ind, (size_t)p))) return rc;
    if (kind != 2) {
        if ((rc = upload(dscale, h_scale.data(), (size_t)d))) return rc;
        if ((rc = upload(dhinv, h_hinv.data(), (size_t)d))) return rc;
        if ((rc = upload(dcst, h_cst.data(), (size_t)d))) return rc;
        if ((rc = upload(ds2k, h_s2k.data(), (size_t)d))) return rc;
    }
    if ((rc = dws.alloc(workspace_bytes(n, p)))) return rc;

    locus_hip_sweep_args a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.p = p; a.d = d;
    a.Xp = dXp.d(); a.xtx = dxtx.d(); a.shuffled_ind = dind.d(); a.lo = dlo.d(); a.l1 = dl1.d();
    a.scale = dscale.d(); a.hinv = dhinv.d(); a.cst = dcst.d(); a.s2k = ds2k.d(); a.cfac = c;
    a.s2 = ds2.d(); a.xwx = dxwx.d(); a.W2 = dW2.d(); a.half_lsi = log_sig2_inv / 2;
    a.ls2 = dls2.d();
    a.R = dR.d(); a.state = dstate.d(); a.stats = nullptr;
    a.workspace = dws.p; a.workspace_bytes = (int64_t)workspace_bytes(n, p); a.nsteps = p;
    if ((rc = sweep_dev(&a, kind == 2, st))) return rc;
    // the permutation was checked on the host; what can still be raised is a cluster time-out:
    // return before anything is copied back, the caller's matrices keep their values
    if ((rc = locus_hip_sweep_status_dev(dws.p, p, st))) return rc;

    // back to the reference's layout: mat_x_m1 from the residual, planes transposed back
    if ((rc = residual_transform(dY.d(), dR.d(), Zmu ? dZ.d() : nullptr,
                                 kind == 2 ? dW2.d() : nullptr, dA.d(), nd, 0, st)))
        return rc;
    LOCUS_HIP_CHECK(hipStreamSynchronize(st));
    if ((rc = download(M, dA, nd))) return rc;
    double* planes_out[3] = {gam, mu, beta};
    for (int q = 0; q < 3; ++q) {
        if ((rc = transpose(dstate.d() + q * pd, dtmp.d(), (int)p, (int)d, st))) return rc;
        LOCUS_HIP_CHECK(hipStreamSynchronize(st));
        if ((rc = download(planes_out[q], dtmp, pd))) return rc;
    }
    return 0;
}

// Common body of the three stateless *_batch entry points (src/coreBatch.cpp).
int run_batch(int probit, const double* X, const double* Y, double* gam, const double* lo,
              const double* l1, double log_sig2_inv, const double* log_tau, double* beta, double* M,
              const double* Zmu, double* mu, const double* sig2_beta, const double* tau, int64_t n,
              int64_t p, int64_t d)
{
    LOCUS_REQUIRE(n >= 1 && p >= 1 && d >= 1, "n, p and d must be >= 1");
    LOCUS_REQUIRE(p < (1ll << 31) && d < (1ll << 31) && n < (1ll << 31), "problem too large");
    LOCUS_REQUIRE(X && Y && gam && lo && l1 && beta && M && mu && sig2_beta, "NULL pointer argument");
    if (!probit) LOCUS_REQUIRE(log_tau && tau, "NULL pointer argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available (liblocus_hip has no CPU fallback)");
        return (int)hipErrorNoDevice;
    }
    const size_t nd = (size_t)n * d, pd = (size_t)p * d, np = (size_t)n * p;
    hipStream_t st = nullptr;
    int rc;

    std::vector<double> h_fac(d), h_hinv(d), h_cstk(d);
    for (int64_t k = 0; k < d; ++k) {
        const double s2 = probit ? sig2_beta[0] : sig2_beta[k];
        h_hinv[k] = 1.0 / (2.0 * s2);
        if (probit) {
            h_fac[k] = s2;                                   // coreBatch.cpp:91
            h_cstk[k] = std::log(s2) / 2;                    // :95
        } else {
            h_fac[k] = s2 * tau[k];                          // :33
            h_cstk[k] = log_tau[k] / 2 + std::log(s2) / 2;   // :36
        }
    }

    DevMem dX, dY, dA, dZ, dR, dstate, dtmp, dlo, dl1, dfac, dhinv, dcstk;
    if ((rc = upload(dX, X, np))) return rc;
    if ((rc = upload(dY, Y, nd))) return rc;
    if ((rc = upload(dA, M, nd))) return rc;
    if (Zmu && (rc = upload(dZ, Zmu, nd))) return rc;
    if ((rc = dR.alloc(sizeof(double) * nd))) return rc;
    if ((rc = residual_transform(dY.d(), dA.d(), Zmu ? dZ.d() : nullptr, nullptr, dR.d(), nd, 1, st)))
        return rc;
    if ((rc = dstate.alloc(sizeof(double) * 3 * pd))) return rc;
    if ((rc = dtmp.alloc(sizeof(double) * pd))) return rc;
    LOCUS_HIP_CHECK(hipMemcpy(dtmp.p, beta, sizeof(double) * pd, hipMemcpyHostToDevice));
    if ((rc = transpose(dtmp.d(), dstate.d() + 2 * pd, (int)d, (int)p, st))) return rc;
    // T = X' R  (p x d), written straight into the mu plane [p][d]
    {
        GemmDesc g;
        g.m = (int)p; g.n = (int)d; g.k = (int)n;
        g.A = dX.d(); g.sa_m = n; g.sa_k = 1; g.sqA = 0;
        g.B = dR.d(); g.sb_k = 1; g.sb_n = n; g.sqB = 0;
        g.C = dstate.d() + pd; g.sc_m = d; g.sc_n = 1;
        g.alpha = 1.0; g.beta = 0.0;
        if ((rc = gemm_f64(g, st))) return rc;
    }
    if ((rc = upload(dlo, lo, (size_t)p))) return rc;
    if ((rc = upload(dl1, l1, (size_t)p))) return rc;
    if ((rc = upload(dfac, h_fac.data(), (size_t)d))) return rc;
    if ((rc = upload(dhinv, h_hinv.data(), (size_t)d))) return rc;
    if ((rc = upload(dcstk, h_cstk.data(), (size_t)d))) return rc;
    if ((rc = batch_update(dstate.d(), dlo.d(), dl1.d(), dfac.d(), dhinv.d(), dcstk.d(),
                           log_sig2_inv, (double)(n - 1), 1.0, probit, (int)p, (int)d, st)))
        return rc;
    // mat_x_m1 = X beta  (n x d column-major) from the new beta plane
    {
        GemmDesc g;
        g.m = (int)n; g.n = (int)d; g.k = (int)p;
        g.A = dX.d(); g.sa_m = 1; g.sa_k = n; g.sqA = 0;
        g.B = dstate.d() + 2 * pd; g.sb_k = d; g.sb_n = 1; g.sqB = 0;
        g.C = dA.d(); g.sc_m = 1; g.sc_n = n;
        g.alpha = 1.0; g.beta = 0.0;
        if ((rc = gemm_f64(g, st))) return rc;
    }
    LOCUS_HIP_CHECK(hipStreamSynchronize(st));
    if ((rc = download(M, dA, nd))) return rc;
    double* planes_out[3] = {gam, mu, beta};
    for (int q = 0; q < 3; ++q) {
        if ((rc = transpose(dstate.d() + q * pd, dtmp.d(), (int)p, (int)d, st))) return rc;
        LOCUS_HIP_CHECK(hipStreamSynchronize(st));
        if ((rc = download(planes_out[q], dtmp, pd))) return rc;
    }
    return 0;
}

}  // namespace
}  // namespace locus

using namespace locus;

extern "C" {

const char* locus_hip_last_error(void) { return get_error(); }
int locus_hip_version(void) { return 100; }

int locus_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int locus_hip_core_loop(const double* X, const double* Y, double* gam_vb, const double* log_om_vb,
                        const double* log_1_min_om_vb, double log_sig2_inv_vb,
                        const double* log_tau_vb, double* beta_vb, double* mat_x_m1,
                        double* mu_beta_vb, const double* sig2_beta_vb, const double* tau_vb,
                        const double* shuffled_ind, double c, int64_t n, int64_t p, int64_t d)
{
    return run_loop(0, X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                    beta_vb, mat_x_m1, nullptr, mu_beta_vb, sig2_beta_vb, tau_vb, nullptr,
                    shuffled_ind, c, n, p, d);
}

int locus_hip_core_z_loop(const double* X, const double* Y, double* gam_vb, const double* log_om_vb,
                          const double* log_1_min_om_vb, double log_sig2_inv_vb,
                          const double* log_tau_vb, double* beta_vb, double* mat_x_m1,
                          const double* mat_z_mu, double* mu_beta_vb, const double* sig2_beta_vb,
                          const double* tau_vb, const double* shuffled_ind, double c, int64_t n,
                          int64_t p, int64_t d)
{
    LOCUS_REQUIRE(mat_z_mu != nullptr, "mat_z_mu is NULL");
    return run_loop(0, X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                    beta_vb, mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb, tau_vb, nullptr,
                    shuffled_ind, c, n, p, d);
}

int locus_hip_core_struct_loop(const double* X, const double* Y, double* gam_vb,
                               const double* log_Phi_mu_theta_vb,
                               const double* log_1_min_Phi_mu_theta_vb, double log_sig2_inv_vb,
                               const double* log_tau_vb, double* beta_vb, double* mat_x_m1,
                               double* mu_beta_vb, const double* sig2_beta_vb, const double* tau_vb,
                               const double* shuffled_ind, int64_t n, int64_t p, int64_t d)
{
    return run_loop(0, X, Y, gam_vb, log_Phi_mu_theta_vb, log_1_min_Phi_mu_theta_vb,
                    log_sig2_inv_vb, log_tau_vb, beta_vb, mat_x_m1, nullptr, mu_beta_vb,
                    sig2_beta_vb, tau_vb, nullptr, shuffled_ind, 1.0, n, p, d);
}

int locus_hip_core_probit_loop(const double* X, const double* W, double* gam_vb,
                               const double* log_om_vb, const double* log_1_min_om_vb,
                               double log_sig2_inv_vb, double* beta_vb, double* mat_x_m1,
                               const double* mat_z_mu, double* mu_beta_vb, double sig2_beta_vb,
                               const double* shuffled_ind, int64_t n, int64_t p, int64_t d)
{
    LOCUS_REQUIRE(mat_z_mu != nullptr, "mat_z_mu is NULL");
    return run_loop(1, X, W, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, nullptr, beta_vb,
                    mat_x_m1, mat_z_mu, mu_beta_vb, &sig2_beta_vb, nullptr, nullptr, shuffled_ind,
                    1.0, n, p, d);
}

int locus_hip_core_logit_loop(const double* X, const double* Y, double* gam_vb,
                              const double* log_om_vb, const double* log_1_min_om_vb,
                              double log_sig2_inv_vb, double* beta_vb, double* mat_x_m1,
                              const double* mat_z_mu, double* mu_beta_vb, const double* psi_vb,
                              const double* sig2_beta_vb, const double* shuffled_ind, int64_t n,
                              int64_t p, int64_t d)
{
    return run_loop(2, X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, nullptr, beta_vb,
                    mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb, nullptr, psi_vb, shuffled_ind, 1.0,
                    n, p, d);
}

int locus_hip_core_batch(const double* X, const double* Y, double* gam_vb, const double* log_om_vb,
                         const double* log_1_min_om_vb, double log_sig2_inv_vb,
                         const double* log_tau_vb, double* beta_vb, double* mat_x_m1,
                         double* mu_beta_vb, const double* sig2_beta_vb, const double* tau_vb,
                         int64_t n, int64_t p, int64_t d)
{
    return run_batch(0, X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                     beta_vb, mat_x_m1, nullptr, mu_beta_vb, sig2_beta_vb, tau_vb, n, p, d);
}

int locus_hip_core_z_batch(const double* X, const double* Y, double* gam_vb,
                           const double* log_om_vb, const double* log_1_min_om_vb,
                           double log_sig2_inv_vb, const double* log_tau_vb, double* beta_vb,
                           double* mat_x_m1, const double* mat_z_mu, double* mu_beta_vb,
                           const double* sig2_beta_vb, const double* tau_vb, int64_t n, int64_t p,
                           int64_t d)
{
    LOCUS_REQUIRE(mat_z_mu != nullptr, "mat_z_mu is NULL");
    return run_batch(0, X, Y, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, log_tau_vb,
                     beta_vb, mat_x_m1, mat_z_mu, mu_beta_vb, sig2_beta_vb, tau_vb, n, p, d);
}

int locus_hip_core_probit_batch(const double* X, const double* W, double* gam_vb,
                                const double* log_om_vb, const double* log_1_min_om_vb,
                                double log_sig2_inv_vb, double* beta_vb, double* mat_x_m1,
                                const double* mat_z_mu, double* mu_beta_vb, double sig2_beta_vb,
                                int64_t n, int64_t p, int64_t d)
{
    LOCUS_REQUIRE(mat_z_mu != nullptr, "mat_z_mu is NULL");
    return run_batch(1, X, W, gam_vb, log_om_vb, log_1_min_om_vb, log_sig2_inv_vb, nullptr, beta_vb,
                     mat_x_m1, mat_z_mu, mu_beta_vb, &sig2_beta_vb, nullptr, n, p, d);
}

// ---- device-resident API ---------------------------------------------------

int64_t locus_hip_packed_ld(int64_t n) { return (n + 127) / 128 * 128; }

int locus_hip_pack_x_dev(const double* X, int64_t n, int64_t p, double* Xp, double* xtx,
                         void* stream)
{
    LOCUS_REQUIRE(X && Xp && xtx, "pack_x: NULL pointer");
    LOCUS_REQUIRE(n >= 1 && p >= 1 && p < (1ll << 31) && n < (1ll << 31), "pack_x: bad shape");
    return pack_x(X, Xp, xtx, (int)n, (int)p, (int)locus_hip_packed_ld(n), (hipStream_t)stream);
}

int64_t locus_hip_sweep_workspace_bytes(int64_t n, int64_t p)
{
    return (int64_t)workspace_bytes(n < 1 ? 1 : n, p < 1 ? 1 : p);
}

int locus_hip_sweep_dev(const locus_hip_sweep_args* args, int logit, void* stream)
{
    return sweep_dev(args, logit, (hipStream_t)stream, (args != nullptr && args->prepacked) ? 2 : 0);
}

int locus_hip_sweep_prepack_dev(const locus_hip_sweep_args* args, void* stream)
{
    return sweep_dev(args, 0, (hipStream_t)stream, 1);
}

int64_t locus_hip_sweep_prepack_workspace_bytes(int64_t n, int64_t p, int64_t d)
{
    if (n < 1 || p < 1 || d < 1 || n >= (1ll << 31) || d >= (1ll << 31)) return 0;
    SweepConfig c;
    if (choose_sweep_config((int)n, (int)d, false, &c) || c.NPL != -1) return 0;
    const int regions = mfma_plan_regions((int)n, (int)d, c);
    if (regions <= 0) return 0;
    const size_t stride = (mfma_workspace_bytes(n, p) + 255) / 256 * 256;
    return (int64_t)(small_workspace_bytes(p) + 512 + (size_t)regions * stride);
}

int locus_hip_sweep_static_dev(const locus_hip_sweep_args* args, int logit, void* stream)
{
    LOCUS_REQUIRE(args != nullptr, "args is NULL");
    int rc = check_dims(args->n, args->p, args->d);
    if (rc) return rc;
    const size_t need = workspace_bytes(args->n, args->p);
    if (need > (size_t)LOCUS_HIP_STATIC_WORKSPACE_BYTES) {
        set_error("sweep_static: a sweep over p = " + std::to_string(args->p) + " covariates of n = " +
                  std::to_string(args->n) + " rows needs " + std::to_string(need) +
                  " bytes of scratch, the static workspace holds " +
                  std::to_string((size_t)LOCUS_HIP_STATIC_WORKSPACE_BYTES) +
                  " (use locus_hip_sweep_dev with a caller-owned workspace)");
        return LOCUS_HIP_EUNSUPPORTED;
    }
    void* ptr = nullptr;
    if ((rc = static_scratch_ptr(&ptr))) return rc;
    locus_hip_sweep_args a = *args;
    a.prepacked = 0;    // (operands packed ahead need a workspace of the caller's, sized for them)
    a.workspace = ptr;  // the device address, never the symbol's name
    a.workspace_bytes = (int64_t)LOCUS_HIP_STATIC_WORKSPACE_BYTES;
    // everything a sweep expects to start from zero (visited[], the error word) is cleared by
    // build_steps through `ptr` on `stream`; the rest of the scratch is written before it is read
    return sweep_dev(&a, logit, (hipStream_t)stream);
}

int locus_hip_sweep_status_dev(const void* workspace, int64_t p, void* stream)
{
    LOCUS_REQUIRE(p >= 1, "sweep_status: bad arguments");
    if (workspace == nullptr) {  // the sweep ran on the static workspace of the current device
        void* ptr = nullptr;
        int rc = static_scratch_ptr(&ptr);
        if (rc) return rc;
        workspace = ptr;
    }
    WsView ws = carve(const_cast<void*>(workspace), p);
    int herr = 0;
    LOCUS_HIP_CHECK(hipMemcpyAsync(&herr, ws.err, sizeof(int), hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
    LOCUS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return decode_sweep_error(herr);
}

int locus_hip_sweep_plan(int64_t n, int64_t d, int32_t* rows, int max_rows)
{
    LOCUS_REQUIRE(rows != nullptr && max_rows >= 1 && n >= 1 && d >= 1 && n < (1ll << 31) && d < (1ll << 31),
                  "sweep_plan: bad arguments");
    SweepConfig c;
    int rc = choose_sweep_config((int)n, (int)d, false, &c);
    if (rc) return rc;
    if (c.NPL != -1) return 0;  // not swept by the matrix-core kernels: no plan
    const int got = mfma_plan_rows((int)n, (int)d, c, rows, max_rows);
    return got < 0 ? LOCUS_HIP_EUNSUPPORTED : got;
}

int locus_hip_sweep_config(int64_t n, int64_t d, int logit, int32_t* cfg)
{
    LOCUS_REQUIRE(cfg != nullptr, "sweep_config: NULL cfg");
    SweepConfig c;
    int rc = choose_sweep_config((int)n, (int)d, logit != 0, &c);
    if (rc) return rc;
    cfg[0] = c.NPL; cfg[1] = c.C; cfg[2] = c.S; cfg[3] = c.G;
    return 0;
}

int locus_hip_residual_dev(const double* Y, const double* A, const double* Zmu, const double* W2,
                           double* out, int64_t n, int64_t d, int to_residual, void* stream)
{
    LOCUS_REQUIRE(Y && A && out && n >= 1 && d >= 1, "residual: bad arguments");
    return residual_transform(Y, A, Zmu, W2, out, (size_t)n * d, to_residual, (hipStream_t)stream);
}

int locus_hip_transpose_dev(const double* in, double* out, int64_t rows, int64_t cols, void* stream)
{
    LOCUS_REQUIRE(in && out && rows >= 1 && cols >= 1 && rows < (1ll << 31) && cols < (1ll << 31),
                  "transpose: bad arguments");
    return transpose(in, out, (int)rows, (int)cols, (hipStream_t)stream);
}

int locus_hip_identity_tail_dev(const locus_hip_tail_args* a, int phase, void* stream)
{
    LOCUS_REQUIRE(a != nullptr && phase >= 0 && phase <= 2, "identity_tail: bad arguments");
    LOCUS_REQUIRE(a->n >= 1 && a->p >= 1 && a->d_loc >= 1 && a->d_total >= a->d_loc &&
                      a->p < (1ll << 31) && a->d_loc < (1ll << 31),
                  "identity_tail: bad dimensions");
    LOCUS_REQUIRE(a->eta && a->kappa && a->stats && a->shard2 && a->scal && a->si_in,
                  "identity_tail: NULL device pointer");
    if (phase == 0)
        LOCUS_REQUIRE(a->tau && a->s2, "identity_tail: phase 0 needs tau, s2");
    else if (phase == 2)
        LOCUS_REQUIRE(a->part && a->lo == nullptr && a->l1 == nullptr &&
                          (a->tau_next == nullptr ||
                           (a->s2_next && a->scale && a->hinv && a->cst)),
                      "identity_tail: phase 2 needs part, no lo / l1, and complete output sets");
    else
        LOCUS_REQUIRE(a->a && a->b && a->rs_gam && a->part && (a->lo == nullptr) == (a->l1 == nullptr) &&
                          (a->tau_next == nullptr ||
                           (a->s2_next && a->scale && a->hinv && a->cst)),
                      "identity_tail: phase 1 needs a, b, rs_gam, part and complete output sets");
    return identity_tail(*a, phase, (hipStream_t)stream);
}

int locus_hip_gamma_rows_dev(const double* gam, double* rs_gam, double* ent, int64_t p, int64_t d,
                             double eps, void* stream)
{
    LOCUS_REQUIRE(gam && rs_gam && p >= 1 && d >= 1 && p < (1ll << 31) && d < (1ll << 31),
                  "gamma_rows: bad arguments");
    return gamma_rows(gam, rs_gam, ent, (int)p, (int)d, eps, (hipStream_t)stream);
}

int locus_hip_state_cols_dev(double* state, const double* s2k, const double* s2jk,
                             double* st_partial, int64_t p, int64_t d, int64_t nchunks, void* stream)
{
    LOCUS_REQUIRE(state && st_partial && (s2k || s2jk) && p >= 1 && d >= 1 && p < (1ll << 31) &&
                      d < (1ll << 31) && nchunks >= 1 && nchunks <= 65535,
                  "state_cols: bad arguments");
    return state_cols(state, s2k, s2jk, st_partial, (int)p, (int)d, (int)nchunks,
                      (hipStream_t)stream);
}

int locus_hip_gemm_dev(int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t sa_m,
                       int64_t sa_k, int sqA, const double* B, int64_t sb_k, int64_t sb_n, int sqB,
                       double beta, double* C, int64_t sc_m, int64_t sc_n, void* stream)
{
    LOCUS_REQUIRE(A && B && C && m >= 0 && n >= 0 && k >= 0 && m < (1ll << 31) && n < (1ll << 31) &&
                      k < (1ll << 31),
                  "gemm: bad arguments");
    LOCUS_REQUIRE((n + 63) / 64 <= 65535, "gemm: n too large (grid.y)");
    GemmDesc g;
    g.m = (int)m; g.n = (int)n; g.k = (int)k;
    g.A = A; g.sa_m = sa_m; g.sa_k = sa_k; g.sqA = sqA;
    g.B = B; g.sb_k = sb_k; g.sb_n = sb_n; g.sqB = sqB;
    g.C = C; g.sc_m = sc_m; g.sc_n = sc_n;
    g.alpha = alpha; g.beta = beta;
    return gemm_f64(g, (hipStream_t)stream);
}

int locus_hip_batch_update_dev(double* state, const double* lo, const double* l1, const double* fac,
                               const double* hinv, const double* cstk, double log_sig2_inv,
                               double nm1, double cfac, int probit, int64_t p, int64_t d,
                               void* stream)
{
    LOCUS_REQUIRE(state && lo && l1 && fac && hinv && cstk && p >= 1 && d >= 1 && p < (1ll << 31) &&
                      d < (1ll << 31),
                  "batch_update: bad arguments");
    return batch_update(state, lo, l1, fac, hinv, cstk, log_sig2_inv, nm1, cfac, probit, (int)p,
                        (int)d, (hipStream_t)stream);
}

// ---- binary-response drivers ------------------------------------------------------

int locus_hip_m2_minus_b2_dev(const double* state, const double* s2jk, double s2c, double* D,
                              int64_t p, int64_t d, void* stream)
{
    LOCUS_REQUIRE(state && D && p >= 1 && d >= 1 && p < (1ll << 31) && d < (1ll << 31),
                  "m2_minus_b2: bad arguments");
    return m2_minus_b2(state, s2jk, s2c, D, (size_t)p * d, (hipStream_t)stream);
}

int locus_hip_s2_from_xwx_dev(const double* xwx, double sig2_inv, double* s2, int64_t len,
                              void* stream)
{
    LOCUS_REQUIRE(xwx && s2 && len >= 1, "s2_from_xwx: bad arguments");
    return s2_from_xwx(xwx, sig2_inv, s2, (size_t)len, (hipStream_t)stream);
}

int locus_hip_half_log_dev(const double* in, double* out, int64_t len, void* stream)
{
    LOCUS_REQUIRE(in && out && len >= 1, "half_log: bad arguments");
    return half_log(in, out, (size_t)len, (hipStream_t)stream);
}

int locus_hip_logit_chi_psi_dev(const double* Q, const double* M, const double* Zmu,
                                const double* Y, const double* Z, const double* s2a,
                                const double* chi_old, const double* W2_old, double* chi_new,
                                double* W2_new, double* ey, int64_t n, int64_t d, int64_t q,
                                void* stream)
{
    LOCUS_REQUIRE(Q && M && Zmu && Z && s2a && chi_new && W2_new && n >= 1 && d >= 1 && q >= 1 &&
                      n < (1ll << 31) && d < (1ll << 31) && q < (1ll << 31),
                  "logit_chi_psi: bad arguments");
    if (ey) LOCUS_REQUIRE(Y && chi_old && W2_old, "logit_chi_psi: ELBO part needs Y, chi_old, W2_old");
    return logit_chi_psi(Q, M, Zmu, Y, Z, s2a, chi_old, W2_old, chi_new, W2_new, ey, (int)n, (int)d,
                         (int)q, (hipStream_t)stream);
}

int locus_hip_logit_sig2_alpha_dev(const double* Z, const double* W2, const double* zeta2_inv,
                                   double* s2a, int64_t n, int64_t d, int64_t q, void* stream)
{
    LOCUS_REQUIRE(Z && W2 && zeta2_inv && s2a && n >= 1 && d >= 1 && q >= 1 && n < (1ll << 31) &&
                      d < (1ll << 31) && q < (1ll << 31),
                  "logit_sig2_alpha: bad arguments");
    return logit_sig2_alpha(Z, W2, zeta2_inv, s2a, (int)n, (int)d, (int)q, (hipStream_t)stream);
}

int locus_hip_alpha_step_dev(const double* Yw, const double* W2, const double* M, double* Zmu,
                             const double* Zi, const double* s2a_i, double* alpha_i, int64_t n,
                             int64_t d, void* stream)
{
    LOCUS_REQUIRE(Yw && M && Zmu && Zi && s2a_i && alpha_i && n >= 1 && d >= 1 && n < (1ll << 31) &&
                      d < (1ll << 31),
                  "alpha_step: bad arguments");
    return alpha_step(Yw, W2, M, Zmu, Zi, s2a_i, alpha_i, (int)n, (int)d, (hipStream_t)stream);
}

int locus_hip_probit_w_dev(const double* Y, const double* M, const double* Zmu, const double* Q,
                           const double* zs, double* W, double* ey, int64_t n, int64_t d,
                           void* stream)
{
    LOCUS_REQUIRE(Y && M && Zmu && W && n >= 1 && d >= 1 && n < (1ll << 31) && d < (1ll << 31),
                  "probit_w: bad arguments");
    if (Q) LOCUS_REQUIRE(zs && ey, "probit_w: ELBO part needs zs and ey");
    return probit_w(Y, M, Zmu, Q, zs, W, ey, (int)n, (int)d, (hipStream_t)stream);
}

int locus_hip_bin_rows_dev(const double* state, const double* s2jk, double s2c, double* rs_gam,
                           double* ent, double* gls, double* m2, int64_t p, int64_t d, double eps,
                           void* stream)
{
    LOCUS_REQUIRE(state && rs_gam && ent && gls && m2 && p >= 1 && d >= 1 && p < (1ll << 31) &&
                      d < (1ll << 31),
                  "bin_rows: bad arguments");
    return bin_rows(state, s2jk, s2c, rs_gam, ent, gls, m2, (int)p, (int)d, eps,
                    (hipStream_t)stream);
}

}  // extern "C"
