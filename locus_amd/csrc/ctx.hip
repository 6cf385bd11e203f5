// Resident context of the identity-family sweep for hosts that cannot hold device memory
// themselves (an R session): host pointers in and out, X / Y / the p x d state stay in HBM
// between calls, only O(p + d) numbers cross PCIe per CAVI iteration.
// (include/locus_hip.h, section 3; SURVEY.md §8(b) "resident-context API".)
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/locus_hip.h"
#include "common.h"
#include "kernels.h"

struct locus_hip_ctx {
    int64_t n = 0, p = 0, d = 0, ld = 0;
    int device = 0;
    bool has_state = false, has_zmu = false;
    double *Xp = nullptr, *xtx = nullptr, *Y = nullptr, *R = nullptr, *M = nullptr, *Zmu = nullptr;
    double *state = nullptr;  // [3][p][d]
    double *tmp = nullptr;    // p x d staging (layout changes)
    double *vec = nullptr;    // lo, l1, shuffled_ind (p each), scale, hinv, cst, s2k (d each)
    double *stats = nullptr;  // [4][d]
    double *rows = nullptr;   // rs_gam (p), entropy rows (p)
    double *part = nullptr;   // [64][3][d] partial column sums of state_cols
    void* ws = nullptr;
    int64_t ws_bytes = 0;
};

namespace locus {
namespace {

int dev_alloc(double** p, size_t count)
{
    LOCUS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), sizeof(double) * (count ? count : 1)));
    return 0;
}

void release(locus_hip_ctx* c)
{
    for (double* q : {c->Xp, c->xtx, c->Y, c->R, c->M, c->Zmu, c->state, c->tmp, c->vec, c->stats,
                      c->rows, c->part})
        if (q) (void)hipFree(q);
    if (c->ws) (void)hipFree(c->ws);
    delete c;
}

int enter(const locus_hip_ctx* c)
{
    LOCUS_REQUIRE(c != nullptr, "ctx is NULL");
    LOCUS_HIP_CHECK(hipSetDevice(c->device));
    return 0;
}

}  // namespace
}  // namespace locus

using namespace locus;

extern "C" {

int locus_hip_ctx_create(const double* X, const double* Y, int64_t n, int64_t p, int64_t d,
                         locus_hip_ctx** out)
{
    LOCUS_REQUIRE(X && Y && out, "ctx_create: NULL pointer argument");
    LOCUS_REQUIRE(n >= 1 && p >= 1 && d >= 1 && p < (1ll << 31) && d < (1ll << 31) &&
                      n < (1ll << 31) && n * d < (1ll << 40) && p * d < (1ll << 40),
                  "ctx_create: bad shape");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available (liblocus_hip has no CPU fallback)");
        return (int)hipErrorNoDevice;
    }
    locus_hip_ctx* c = new locus_hip_ctx;
    c->n = n; c->p = p; c->d = d; c->ld = locus_hip_packed_ld(n);
    int rc = 0;
    auto fail = [&](int code) { release(c); return code; };
    if (hipGetDevice(&c->device) != hipSuccess) return fail((int)hipErrorInvalidDevice);
    const size_t nd = (size_t)n * d, pd = (size_t)p * d;
    c->ws_bytes = locus_hip_sweep_workspace_bytes(n, p);
    if ((rc = dev_alloc(&c->Xp, (size_t)p * c->ld)) || (rc = dev_alloc(&c->xtx, p)) ||
        (rc = dev_alloc(&c->Y, nd)) || (rc = dev_alloc(&c->R, nd)) || (rc = dev_alloc(&c->M, nd)) ||
        (rc = dev_alloc(&c->state, 3 * pd)) || (rc = dev_alloc(&c->tmp, pd)) ||
        (rc = dev_alloc(&c->vec, 3 * (size_t)p + 4 * (size_t)d)) ||
        (rc = dev_alloc(&c->stats, 4 * (size_t)d)) || (rc = dev_alloc(&c->rows, 2 * (size_t)p)) ||
        (rc = dev_alloc(&c->part, 64 * 3 * (size_t)d)))
        return fail(rc);
    if (hipMalloc(&c->ws, (size_t)c->ws_bytes) != hipSuccess) {
        set_error("ctx_create: out of device memory (sweep workspace)");
        return fail((int)hipErrorOutOfMemory);
    }
    {   // X: raw copy -> packed columns + X_j'X_j; the raw copy is released at once
        double* raw = nullptr;
        if ((rc = dev_alloc(&raw, (size_t)n * p))) return fail(rc);
        hipError_t e = hipMemcpy(raw, X, sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            rc = pack_x(raw, c->Xp, c->xtx, (int)n, (int)p, (int)c->ld, nullptr);
            if (!rc) e = hipDeviceSynchronize();
        }
        (void)hipFree(raw);
        if (e != hipSuccess) {
            set_error(std::string("ctx_create: ") + hipGetErrorString(e));
            return fail((int)e);
        }
        if (rc) return fail(rc);
    }
    if (hipMemcpy(c->Y, Y, sizeof(double) * nd, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("ctx_create: copy of Y failed");
        return fail((int)hipErrorUnknown);
    }
    *out = c;
    return 0;
}

int locus_hip_ctx_set_state(locus_hip_ctx* c, const double* gam_vb, const double* mu_beta_vb,
                            const double* mat_z_mu)
{
    int rc = enter(c);
    if (rc) return rc;
    LOCUS_REQUIRE(gam_vb && mu_beta_vb, "ctx_set_state: NULL pointer argument");
    const int64_t n = c->n, p = c->p, d = c->d;
    const size_t nd = (size_t)n * d, pd = (size_t)p * d;
    const double* planes[2] = {gam_vb, mu_beta_vb};
    for (int q = 0; q < 2; ++q) {  // R's column-major p x d == row-major [d][p] -> plane [p][d]
        LOCUS_HIP_CHECK(hipMemcpy(c->tmp, planes[q], sizeof(double) * pd, hipMemcpyHostToDevice));
        if ((rc = transpose(c->tmp, c->state + q * pd, (int)d, (int)p, nullptr))) return rc;
        LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    // beta_vb = gam_vb * mu_beta_vb (R/update_vb.R:66); the column sums it also forms are not used
    LOCUS_HIP_CHECK(hipMemsetAsync(c->stats, 0, sizeof(double) * 4 * (size_t)d, nullptr));
    const int nchunks = (int)std::min<int64_t>(64, p);
    if ((rc = state_cols(c->state, c->stats, nullptr, c->part, (int)p, (int)d, nchunks, nullptr)))
        return rc;
    // mat_x_m1 = X beta_vb (R/update_vb.R:114) on the matrix cores, then the residual
    GemmDesc g;
    g.m = (int)n; g.n = (int)d; g.k = (int)p;
    g.A = c->Xp; g.sa_m = 1; g.sa_k = c->ld; g.sqA = 0;
    g.B = c->state + 2 * pd; g.sb_k = d; g.sb_n = 1; g.sqB = 0;
    g.C = c->M; g.sc_m = 1; g.sc_n = n;
    g.alpha = 1.0; g.beta = 0.0;
    if ((rc = gemm_f64(g, nullptr))) return rc;
    c->has_zmu = mat_z_mu != nullptr;
    if (c->has_zmu) {
        if (!c->Zmu && (rc = dev_alloc(&c->Zmu, nd))) return rc;
        LOCUS_HIP_CHECK(hipMemcpy(c->Zmu, mat_z_mu, sizeof(double) * nd, hipMemcpyHostToDevice));
    }
    if ((rc = residual_transform(c->Y, c->M, c->has_zmu ? c->Zmu : nullptr, nullptr, c->R, nd, 1,
                                 nullptr)))
        return rc;
    LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
    c->has_state = true;
    return 0;
}

int locus_hip_ctx_set_response(locus_hip_ctx* c, const double* Y_or_W, const double* mat_z_mu)
{
    int rc = enter(c);
    if (rc) return rc;
    LOCUS_REQUIRE(c->has_state, "ctx_set_response: call locus_hip_ctx_set_state first");
    LOCUS_REQUIRE(Y_or_W || mat_z_mu, "ctx_set_response: nothing to replace (both pointers NULL)");
    const size_t nd = (size_t)c->n * c->d;
    // mat_x_m1 = Y_old - R - Zmu_old is what the sweeps have built; it does not change here
    if ((rc = residual_transform(c->Y, c->R, c->has_zmu ? c->Zmu : nullptr, nullptr, c->M, nd, 0,
                                 nullptr)))
        return rc;
    LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (Y_or_W) LOCUS_HIP_CHECK(hipMemcpy(c->Y, Y_or_W, sizeof(double) * nd, hipMemcpyHostToDevice));
    if (mat_z_mu) {
        if (!c->Zmu && (rc = dev_alloc(&c->Zmu, nd))) return rc;
        LOCUS_HIP_CHECK(hipMemcpy(c->Zmu, mat_z_mu, sizeof(double) * nd, hipMemcpyHostToDevice));
        c->has_zmu = true;
    }
    // R = Y_new - mat_x_m1 - Zmu_new
    if ((rc = residual_transform(c->Y, c->M, c->has_zmu ? c->Zmu : nullptr, nullptr, c->R, nd, 1,
                                 nullptr)))
        return rc;
    LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
    return 0;
}

int locus_hip_ctx_sweep(locus_hip_ctx* c, const double* log_om_vb, const double* log_1_min_om_vb,
                        double log_sig2_inv_vb, const double* log_tau_vb,
                        const double* sig2_beta_vb, const double* tau_vb,
                        const double* shuffled_ind, double cc, double ent_eps, double* rs_gam,
                        double* col_stats, double* ent_sum)
{
    int rc = enter(c);
    if (rc) return rc;
    LOCUS_REQUIRE(c->has_state, "ctx_sweep: call locus_hip_ctx_set_state first");
    LOCUS_REQUIRE(log_om_vb && log_1_min_om_vb && log_tau_vb && sig2_beta_vb && tau_vb &&
                      shuffled_ind,
                  "ctx_sweep: NULL pointer argument");
    const int64_t p = c->p, d = c->d;
    // per-response constants of the kernel (src/coreLoop.cpp:34,42), then one staged upload
    std::vector<double> h(3 * (size_t)p + 4 * (size_t)d);
    memcpy(h.data(), log_om_vb, sizeof(double) * p);
    memcpy(h.data() + p, log_1_min_om_vb, sizeof(double) * p);
    memcpy(h.data() + 2 * p, shuffled_ind, sizeof(double) * p);
    double* hs = h.data() + 3 * p;
    for (int64_t k = 0; k < d; ++k) {
        const double s2 = sig2_beta_vb[k];
        hs[k] = cc * s2 * tau_vb[k];
        hs[d + k] = 1.0 / (2.0 * s2);
        hs[2 * d + k] = -(log_tau_vb[k] + log_sig2_inv_vb + std::log(s2)) / 2;
        hs[3 * d + k] = s2;
    }
    LOCUS_HIP_CHECK(hipMemcpy(c->vec, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));

    locus_hip_sweep_args a;
    memset(&a, 0, sizeof(a));
    a.n = c->n; a.p = p; a.d = d;
    a.Xp = c->Xp; a.xtx = c->xtx;
    a.lo = c->vec; a.l1 = c->vec + p; a.shuffled_ind = c->vec + 2 * p;
    a.scale = c->vec + 3 * p; a.hinv = a.scale + d; a.cst = a.scale + 2 * d; a.s2k = a.scale + 3 * d;
    a.cfac = cc;
    a.R = c->R; a.state = c->state; a.stats = c->stats;
    a.workspace = c->ws; a.workspace_bytes = c->ws_bytes; a.nsteps = p;
    if ((rc = locus_hip_sweep_dev(&a, 0, nullptr))) return rc;
    const bool want_ent = ent_sum != nullptr;
    if ((rc = gamma_rows(c->state, c->rows, want_ent ? c->rows + p : nullptr, (int)p, (int)d, ent_eps,
                         nullptr)))
        return rc;
    if ((rc = locus_hip_sweep_status_dev(c->ws, p, nullptr))) return rc;  // synchronises
    if (rs_gam) LOCUS_HIP_CHECK(hipMemcpy(rs_gam, c->rows, sizeof(double) * p, hipMemcpyDeviceToHost));
    if (col_stats)
        LOCUS_HIP_CHECK(hipMemcpy(col_stats, c->stats, sizeof(double) * 4 * (size_t)d,
                                  hipMemcpyDeviceToHost));
    if (want_ent) {
        std::vector<double> e((size_t)p);
        LOCUS_HIP_CHECK(hipMemcpy(e.data(), c->rows + p, sizeof(double) * p, hipMemcpyDeviceToHost));
        double s = 0.0;
        for (double v : e) s += v;
        *ent_sum = s;
    }
    return 0;
}

int locus_hip_ctx_stats(locus_hip_ctx* c, const double* sig2_beta_vb, double ent_eps,
                        double* rs_gam, double* col_stats, double* ent_sum)
{
    int rc = enter(c);
    if (rc) return rc;
    LOCUS_REQUIRE(c->has_state, "ctx_stats: call locus_hip_ctx_set_state first");
    LOCUS_REQUIRE(sig2_beta_vb, "ctx_stats: NULL pointer argument");
    const int64_t n = c->n, p = c->p, d = c->d;
    double* s2k = c->vec + 3 * p + 3 * d;
    LOCUS_HIP_CHECK(hipMemcpy(s2k, sig2_beta_vb, sizeof(double) * d, hipMemcpyHostToDevice));
    const int nchunks = (int)std::min<int64_t>(64, p);
    if ((rc = state_cols(c->state, s2k, nullptr, c->part, (int)p, (int)d, nchunks, nullptr))) return rc;
    const bool want_ent = ent_sum != nullptr;
    if ((rc = gamma_rows(c->state, c->rows, want_ent ? c->rows + p : nullptr, (int)p, (int)d, ent_eps,
                         nullptr)))
        return rc;
    LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (rs_gam) LOCUS_HIP_CHECK(hipMemcpy(rs_gam, c->rows, sizeof(double) * p, hipMemcpyDeviceToHost));
    if (col_stats) {
        std::vector<double> part((size_t)nchunks * 3 * d), R((size_t)n * d);
        LOCUS_HIP_CHECK(hipMemcpy(part.data(), c->part, sizeof(double) * part.size(),
                                  hipMemcpyDeviceToHost));
        LOCUS_HIP_CHECK(hipMemcpy(R.data(), c->R, sizeof(double) * R.size(), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < 4 * d; ++k) col_stats[k] = 0.0;
        for (int ch = 0; ch < nchunks; ++ch)
            for (int64_t k = 0; k < 3 * d; ++k) col_stats[k] += part[(size_t)ch * 3 * d + k];
        for (int64_t k = 0; k < d; ++k) {  // one-off: sum of squares of the residual columns
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += R[(size_t)k * n + i] * R[(size_t)k * n + i];
            col_stats[3 * d + k] = s;
        }
    }
    if (want_ent) {
        std::vector<double> e((size_t)p);
        LOCUS_HIP_CHECK(hipMemcpy(e.data(), c->rows + p, sizeof(double) * p, hipMemcpyDeviceToHost));
        double s = 0.0;
        for (double v : e) s += v;
        *ent_sum = s;
    }
    return 0;
}

int locus_hip_ctx_get_state(locus_hip_ctx* c, double* gam_vb, double* mu_beta_vb, double* beta_vb,
                            double* mat_x_m1)
{
    int rc = enter(c);
    if (rc) return rc;
    LOCUS_REQUIRE(c->has_state, "ctx_get_state: no state yet");
    const int64_t n = c->n, p = c->p, d = c->d;
    const size_t nd = (size_t)n * d, pd = (size_t)p * d;
    double* outs[3] = {gam_vb, mu_beta_vb, beta_vb};
    for (int q = 0; q < 3; ++q) {
        if (!outs[q]) continue;
        if ((rc = transpose(c->state + q * pd, c->tmp, (int)p, (int)d, nullptr))) return rc;
        LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
        LOCUS_HIP_CHECK(hipMemcpy(outs[q], c->tmp, sizeof(double) * pd, hipMemcpyDeviceToHost));
    }
    if (mat_x_m1) {  // mat_x_m1 = Y - R (- mat_z_mu)
        if ((rc = residual_transform(c->Y, c->R, c->has_zmu ? c->Zmu : nullptr, nullptr, c->M, nd, 0,
                                     nullptr)))
            return rc;
        LOCUS_HIP_CHECK(hipStreamSynchronize(nullptr));
        LOCUS_HIP_CHECK(hipMemcpy(mat_x_m1, c->M, sizeof(double) * nd, hipMemcpyDeviceToHost));
    }
    return 0;
}

int locus_hip_ctx_destroy(locus_hip_ctx* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    release(c);
    return 0;
}

}  // extern "C"
