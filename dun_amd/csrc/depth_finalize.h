// The depth-level part of the objective, shared by the stand-alone finalisation kernels (elbo.hip) and the persistent
// small-net forward (persist.hip), whose last workgroup to arrive runs it inside the forward launch.
#pragma once
#include <math.h>

#include "common.h"

namespace dun {

// One wavefront (all 64 lanes active) turns the per-depth sums into everything the objective and its backward need (fp64
// inside).  Its dozen wave-wide reductions run on the DPP path: as ds_bpermute butterflies over doubles they were most
// of the 7 us this took for D = 11.
// `sums` may live in global memory or in LDS (generic pointer).
__device__ __forceinline__ void depth_finalize_body(const double* sums, int gaussian,
                                                    const float* __restrict__ q_logits,
                                                    const float* __restrict__ logprior,
                                                    const float* __restrict__ prior_ll,
                                                    const float* __restrict__ log_std, int D, double N, double batch,
                                                    double beta, double eps, int objective, float* __restrict__ res) {
    const int lane = threadIdx.x & 63;
    float* joint_o = res + DUN_RES_HEADER;
    float* w_o = joint_o + D;
    float* dq_o = w_o + D;
    float* post_o = dq_o + D;
    float* llsum_o = post_o + D;

    double sigma = 1.0, live = 0.0;
    if (gaussian) {
        const double raw = exp((double)log_std[0]);
        // the reference clamps in fp32: compare in fp32 so the branch agrees with it at the boundary
        live = (expf(log_std[0]) >= 1e-4f) ? 1.0 : 0.0;
        sigma = live != 0.0 ? raw : (double)1e-4f;
    }
    const double inv_var = 1.0 / (sigma * sigma);

    // joint and the exact marginal / posterior
    double mx = -INFINITY;
    for (int d = lane; d < D; d += 64) {
        const double LL = gaussian ? -batch * (LOG_SQRT_2PI_D + log(sigma)) - 0.5 * sums[d] * inv_var : sums[d];
        const double J = (N / batch) * LL + (prior_ll ? (double)prior_ll[d] : 0.0);
        llsum_o[d] = (float)LL;
        joint_o[d] = (float)J;
        mx = fmax(mx, J + (double)logprior[d]);
    }
    mx = wave_allmax(mx);
    double se = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double LL = gaussian ? -batch * (LOG_SQRT_2PI_D + log(sigma)) - 0.5 * sums[d] * inv_var : sums[d];
        const double J = (N / batch) * LL + (prior_ll ? (double)prior_ll[d] : 0.0);
        se += exp(J + (double)logprior[d] - mx);
    }
    se = wave_allsum(se);
    const double log_marg = mx + log(se);

    // q = softmax(q_logits) and log q = log_softmax(q_logits) in fp32, exactly the arithmetic the reference's
    // F.softmax / F.log_softmax do: a q_d that underflows to 0 there is 0 here (and gets the eps clamp).
    float qmx = -INFINITY, qs = 0.f;
    if (objective == DUN_OBJ_VI) {
        for (int d = lane; d < D; d += 64) qmx = fmaxf(qmx, q_logits[d]);
        qmx = wave_allmax(qmx);
        for (int d = lane; d < D; d += 64) qs += expf(q_logits[d] - qmx);
        qs = wave_allsum(qs);
    }
    const float log_qs = logf(qs);

    double E = 0.0, KL = 0.0, Asum = 0.0, Csum = 0.0, dls = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double LL = gaussian ? -batch * (LOG_SQRT_2PI_D + log(sigma)) - 0.5 * sums[d] * inv_var : sums[d];
        const double J = (N / batch) * LL + (prior_ll ? (double)prior_ll[d] : 0.0);
        const double lp = (double)logprior[d];
        const double post = exp(J + lp - log_marg);
        post_o[d] = (float)post;
        double wgt = post;
        if (objective == DUN_OBJ_VI) {
            const double lq = (double)((q_logits[d] - qmx) - log_qs);
            const double q = (double)(expf(q_logits[d] - qmx) / qs);
            const double c = fmin(fmax(q, eps), 1.0 - eps);
            const double m = (q >= eps && q <= 1.0 - eps) ? 1.0 : 0.0;
            E += q * J;
            KL += c * (lq - lp);
            Asum += m * q * (lq - lp);
            Csum += c;
            wgt = q;
        }
        w_o[d] = (float)wgt;
        if (gaussian) dls += wgt * (sums[d] * inv_var - batch);
    }
    E = wave_allsum(E);
    KL = wave_allsum(KL);
    Asum = wave_allsum(Asum);
    Csum = wave_allsum(Csum);
    dls = wave_allsum(dls);

    for (int d = lane; d < D; d += 64) {
        double dq = 0.0;
        if (objective == DUN_OBJ_VI) {
            const double LL = gaussian ? -batch * (LOG_SQRT_2PI_D + log(sigma)) - 0.5 * sums[d] * inv_var : sums[d];
            const double J = (N / batch) * LL + (prior_ll ? (double)prior_ll[d] : 0.0);
            const double lp = (double)logprior[d];
            const double lq = (double)((q_logits[d] - qmx) - log_qs);
            const double q = (double)(expf(q_logits[d] - qmx) / qs);
            const double c = fmin(fmax(q, eps), 1.0 - eps);
            const double m = (q >= eps && q <= 1.0 - eps) ? 1.0 : 0.0;
            const double dKL = m * q * (lq - lp) - q * Asum + c - q * Csum;
            dq = -(1.0 / N) * (q * (J - E) - beta * dKL);
        }
        dq_o[d] = (float)dq;
    }
    if (lane == 0) {
        res[DUN_RES_OBJECTIVE] = (float)(objective == DUN_OBJ_VI ? E - beta * KL : log_marg);
        res[DUN_RES_E_LOGLIKE] = (float)E;
        res[DUN_RES_KL] = (float)KL;
        res[DUN_RES_LOG_MARG] = (float)log_marg;
        res[DUN_RES_DLOG_STD] = (float)(gaussian ? -(1.0 / batch) * dls * live : 0.0);
        res[5] = 0.f;
        res[6] = 0.f;
        res[7] = 0.f;
    }
}

}  // namespace dun
