// One-launch forward and one-launch backward for SMALL residual-MLP DUNs (toy / UCI configurations: width <= 128,
// batch <= 2048, one output) -- src/DUN/stochastic_fc_models.py:62-70 over src/DUN/layers.py:33-53 in the reference.
//
// smallnet.hip runs such a net as one launch per block (a workgroup owns four feature columns for every row); the step
// is then ~30 dependent launches of 4-10 us, and each launch re-ingests the whole [B,W] hidden state per workgroup.
// Here the ROWS are owned instead: workgroup p keeps rows [16p, 16p+16) of the batch through ALL blocks.
//   * Linear layers are row-independent: the 16 x W hidden tile lives in LDS, the [16,W] x [W,W] product runs on
//     v_mfma_f32_16x16x4_f32 (wave w owns output columns [16w, 16w+16); its slice of W_i sits in registers, fetched
//     while the previous block's exchange is in flight), so no hidden state ever crosses workgroups.
//   * BatchNorm1d couples rows through per-column batch statistics only: each wave leaves ONE 16-byte record per
//     column {partial a, tag, partial b, tag} (forward: Welford mean / M2 of its 16 rows, cancellation-free in fp32;
//     backward: sum g, sum g*xhat) with a write-through (sc1) store, and reads the P records of its own 16 columns with
//     sc1 loads until every tag carries this phase's number -- the data is its own flag (MI355X guide, Guideline 16
//     R2), so there is no grid barrier, no fence, no counter on the per-block path, and waves exchange independently
//     of each other.  Totals are added in fp64 in a fixed order (bitwise reproducible).  Two record slots alternate by
//     phase parity; tags count up across launches from an epoch word in the sync buffer, so nothing is zeroed per call
//     (graph replays included: the epoch lives in device memory, not in a kernel argument).
//   * The shared head (one output), the Gaussian likelihood terms (act, d loss / d act, per-depth sums of squared
//     residuals), the head's weight / bias gradient partials and -- by the workgroup whose partial sums arrive last --
//     the whole depth-level finalisation (dun_depth_finalize_partials' arithmetic) ride in the forward launch.
// Every spin is bounded by the wall clock; a wave that gives up records a code in sync[1], the launch drains with
// every later wait skipped, and the outputs of that call are NaN.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "depth_finalize.h"

namespace dun {

constexpr int PN_T = 512;     // 8 waves: one 16-column MFMA tile each
constexpr int PN_R = 16;      // rows of the batch a workgroup owns
constexpr int PN_MAX_P = 128;
constexpr int PN_MAX_W = 128;
constexpr int PN_MAX_L = 255;
constexpr int PN_HDR_BYTES = 256;    // sync header: word 0 epoch, word 1 error code, word 2 finalisation ticket
constexpr int PN_STAGE_DIN = 32;     // input widths up to this are staged in LDS for the input layer
constexpr int PN_REC_STRIDE = 128;   // records per (parity, workgroup) row, one per column
constexpr long long PN_TIMEOUT_TICKS = 200000000LL;  // wall_clock64 runs at 100 MHz: 2 s

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// -DPN_PROFILE (scripts/dev/pn_phases.py): wave 0 of every workgroup stamps the 100 MHz wall clock into the unused record
// columns 104..127 of its two record rows (48 stamps per row)
#ifdef PN_PROFILE
#define PN_STAMP(k)                                                                                                 \
    do {                                                                                                            \
        if (threadIdx.x == 0 && (k) < 96)                                                                           \
            reinterpret_cast<long long*>(reinterpret_cast<char*>(p.sync) +                                          \
                                         pn_rec_offset((k) / 48, p.P, blockIdx.x, 104))[(k) % 48] = wall_clock64(); \
    } while (0)
#else
#define PN_STAMP(k) do { } while (0)
#endif

static int g_small_net_mode = -1;  // -1: take it from the environment on first use

int small_net_mode() {
    if (g_small_net_mode < 0) {
        const char* off = getenv("DUN_SMALLNET");
        const char* pers = getenv("DUN_SMALLNET_PERSIST");
        g_small_net_mode = (off && off[0] == '0') ? 0 : ((pers && pers[0] == '0') ? 1 : 2);
    }
    return g_small_net_mode;
}

static inline int pn_workgroups(int B) { return (B + PN_R - 1) / PN_R; }
size_t pn_sync_bytes(int B, int W) {
    if (W < 4 || W > PN_MAX_W || (W & 3) || B < 1 || B > PN_R * PN_MAX_P) return 0;
    return (size_t)PN_HDR_BYTES + (size_t)2 * pn_workgroups(B) * PN_REC_STRIDE * 16;
}

bool pn_eligible(const dun_fc_ctx* c) {
    if (small_net_mode() != 2) return false;
    const size_t need = pn_sync_bytes(c->B, c->W);
    if (need == 0 || c->sync == nullptr || c->sync_bytes < need) return false;
    if (c->O != 1 || c->L > PN_MAX_L) return false;
    return aligned16(c->p.w) && (reinterpret_cast<uintptr_t>(c->sync) & 255) == 0;
}

// ---- exchange of per-column partials between the row-owning workgroups ---------------------------------------------------
struct PnXch {
    __amdgpu_buffer_rsrc_t rs;  // the sync buffer
    unsigned* words;            // its header
    int* abort_flag;            // LDS: a wave of this workgroup has given up
    int P, pid;
};

__device__ __forceinline__ int pn_rec_offset(int parity, int P, int p, int col) {
    return PN_HDR_BYTES + ((parity * P + p) * PN_REC_STRIDE + col) * 16;
}

__device__ __forceinline__ void pn_publish(const PnXch& x, int parity, int col, float a, float b, unsigned tag) {
    const u32x4 v = {__float_as_uint(a), tag, __float_as_uint(b), tag};
    __builtin_amdgcn_raw_buffer_store_b128(v, x.rs, pn_rec_offset(parity, x.P, x.pid, col), 0, 16);  // aux 16 = sc1
}

// Lane (j = lane & 15, q = lane >> 4) reads the records of column `col` from workgroups q, q+4, q+8, ... and adds them
// in that order; the four lane groups are then combined ((q0 + q1) + (q2 + q3)), so every lane of a column ends with the
// same fp64 totals.  WELFORD: the records are {mean, M2} of n_p rows -> t1 = sum n_p mean_p, t2 = sum M2_p + n_p mean_p^2.
template <bool WELFORD>
__device__ __forceinline__ void pn_gather(const PnXch& x, int parity, int col, bool active, unsigned tag, int B,
                                          unsigned code, double& t1, double& t2) {
    constexpr int CH = 8;
    const int lane = threadIdx.x & 63, q = lane >> 4;
    const int M = (x.P + 3) >> 2;
    double s1 = 0.0, s2 = 0.0;
    for (int m0 = 0; m0 < M; m0 += CH) {
        // A full sweep first (all records of this chunk in flight together); if some are not tagged yet, WATCH the first
        // missing one with a single load per lane until it arrives, then sweep again: the chip's 200-odd polling waves
        // would otherwise re-read every record every round (MB per round through the fabric the records travel on).
        u32x4 g[CH];
        const long long t_begin = wall_clock64();
        for (;;) {
            // (compiler barrier: the records change under this loop; it keeps the loads inside it and, with every
            // access to them being an sc1 buffer access, nothing of them is cached in registers or L1)
            asm volatile("" ::: "memory");
            unsigned missing = 0;  // wave-uniform: chunk entries some lane still waits for
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int pi = q + 4 * (m0 + u);
                g[u] = __builtin_amdgcn_raw_buffer_load_b128(x.rs, pn_rec_offset(parity, x.P, min(pi, x.P - 1), col), 0, 16);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int pi = q + 4 * (m0 + u);
                const bool ok = !(active && pi < x.P) || (g[u][1] == tag && g[u][3] == tag);
                if (!__all(ok)) missing |= 1u << u;
            }
            if (missing == 0) break;
            const int pw = q + 4 * (m0 + __builtin_ctz(missing));
            const int woff = pn_rec_offset(parity, x.P, min(pw, x.P - 1), col);
            bool lost = false;
            for (;;) {
                asm volatile("" ::: "memory");
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(x.rs, woff, 0, 16);
                const bool ok = !(active && pw < x.P) || (r[1] == tag && r[3] == tag);
                if (__all(ok)) break;
                if (*reinterpret_cast<volatile int*>(x.abort_flag)) {  // this launch is already lost: drain
                    lost = true;
                    break;
                }
                if (wall_clock64() - t_begin > PN_TIMEOUT_TICKS) {
                    if (lane == 0) {
                        *reinterpret_cast<volatile int*>(x.abort_flag) = 1;
                        __hip_atomic_store(x.words + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    lost = true;
                    break;
                }
            }
            if (lost) break;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int pi = q + 4 * (m0 + u);
            if (active && pi < x.P) {
                // (through scalars: __builtin_bit_cast applied to a vector ELEMENT read element 0 for both, ROCm 7.2)
                const unsigned ua = g[u][0], ub = g[u][2];
                const double a = (double)__uint_as_float(ua);
                const double b = (double)__uint_as_float(ub);
                if (WELFORD) {
                    const double n = (double)min(PN_R, B - PN_R * pi);
                    s1 += n * a;
                    s2 += b + n * a * a;
                } else {
                    s1 += a;
                    s2 += b;
                }
            }
        }
    }
    // (q0 + q1) + (q2 + q3) by two butterfly steps: addition commutes, so all four lane groups end with the same bits
    s1 += __shfl_xor(s1, 16, 64);
    s2 += __shfl_xor(s2, 16, 64);
    t1 = s1 + __shfl_xor(s1, 32, 64);
    t2 = s2 + __shfl_xor(s2, 32, 64);
}

// Workgroup barrier for data handed over through LDS only: waits for this wave's LDS operations, not for its global loads
// and stores (__syncthreads() drains both, which would expose the latency of every prefetch and result store in flight)
__device__ __forceinline__ void pn_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// sum over the four lane groups of a wave (rows 4q .. 4q+3 of a column live in lane group q of the MFMA result layout)
__device__ __forceinline__ float pn_sum_groups(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float pn_row16_sum(float v) {
    v += dpp_move<DPP_QUAD_XOR1>(v);
    v += dpp_move<DPP_QUAD_XOR2>(v);
    v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move<DPP_ROW_MIRROR>(v);
    return v;
}

// [16, K] x [K, 16] on v_mfma_f32_16x16x4_f32: A rows from the LDS tile (16-byte reads along k), this wave's operand
// columns in registers.  MFMA step (s, e) contracts k = 16 s + 4 (lane >> 4) + e on both operands; two accumulators
// alternate over s so that consecutive MFMAs do not wait for each other's result.
template <int NT>
__device__ __forceinline__ f32x4 pn_tile_matmul(const float* tile, int ld, const f32x4 (&wq)[NT]) {
    const int lane = threadIdx.x & 63;
    const float* arow = tile + (lane & 15) * ld + 4 * (lane >> 4);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const f32x4 aq = *reinterpret_cast<const f32x4*>(arow + 16 * s);
        if (s & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[e], wq[s][e], acc1, 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[e], wq[s][e], acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

struct PnFwdP {
    int B, din, W, depth, residual, P;
    const float* x;
    const float *w_in, *b_in, *w_out, *b_out, *w, *b, *gamma, *beta;
    float *running_mean, *running_var;
    float *H, *Z, *bn_mean, *bn_rstd, *act;
    long long* nbt;
    float eps, momentum;
    unsigned* sync;
    int sync_bytes;
    // likelihood pass (y == NULL: plain forward)
    const float* y;
    const float* qw;  // depth weights / logits, or NULL (sums only)
    int w_is_logits;
    const float* log_std;
    float inv_batch;
    float* dact;
    double* partials;  // [depth+1][nparts]
    int nparts;
    float* dwpart;     // [P][W] or NULL
    float* dbpart;     // [P]
    // finalisation by the last workgroup (fin_result == NULL: none)
    float* fin_result;
    double* fin_sums;
    const float *fin_q_logits, *fin_logprior, *fin_prior_ll;
    double fin_N, fin_batch, fin_beta, fin_eps;
    int fin_objective;
};

template <int NT, bool TRAIN>
__global__ __launch_bounds__(PN_T) void pn_forward_kernel(PnFwdP p) {
    constexpr int WP = 16 * NT, LD = WP + 4;
    __shared__ __attribute__((aligned(16))) float hT[2][PN_R][LD];
    __shared__ __attribute__((aligned(16))) float s_gw[PN_R][WP];
    __shared__ float s_rsq[PN_MAX_L + 1][PN_R];
    __shared__ float s_coef[PN_MAX_L + 1];
    __shared__ float s_gb[PN_R];
    __shared__ double s_sums[PN_MAX_L + 1];
    __shared__ float s_x[PN_R][PN_STAGE_DIN];
    __shared__ float s_win[PN_MAX_W][PN_STAGE_DIN + 1];
    __shared__ int s_abort, s_ticket;

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pid = blockIdx.x, r0 = pid * PN_R;
    const int B = p.B, W = p.W, depth = p.depth, D = depth + 1;
    const int nvalid = min(PN_R, B - r0);
    const bool exchanging = TRAIN && depth > 0;
    PN_STAMP(90);

    PnXch xch;
    xch.rs = __builtin_amdgcn_make_buffer_rsrc(p.sync, 0, p.sync_bytes, 0x00020000);
    xch.words = p.sync;
    xch.abort_flag = &s_abort;
    xch.P = p.P;
    xch.pid = pid;
    unsigned base = 0;
    if (exchanging) base = __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) s_abort = 0;

    // this wave's slice of a block's weights: wq[s][e] = W_i[n][16 s + 4 (lane >> 4) + e], n = 16 wv + (lane & 15)
    const int col = 16 * wv + (lane & 15), q = lane >> 4;
    const bool colok = wv < NT && col < W;
    auto load_w = [&](int i, f32x4 (&dst)[NT]) {
        // (32-bit offsets from the uniform base: with 64-bit per-lane addresses the compiler kept one address pair per
        // load alive across the loop)
        const unsigned wi = (unsigned)i * (unsigned)(W * W) + (unsigned)min(col, W - 1) * (unsigned)W;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int k = 16 * s + 4 * q;
            // (times a 0/1 mask, not a select: the compiler sinks a load whose value is only selected into a branch of
            // its own, and every such branch waits for its load alone -- one memory round trip after another)
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + (wi + (unsigned)min(k, W - 4)));
            dst[s] = v * ((colok && k < W) ? 1.f : 0.f);
        }
    };
    // per-column parameters of a block (this lane's column), fetched one block ahead like the weights
    struct ColP { float bias, gm, bt, rmean, rvar; };
    auto load_colp = [&](int i, ColP& c) {
        const unsigned pi = (unsigned)i * (unsigned)W + (unsigned)min(col, W - 1);
        c.bias = p.b[pi]; c.gm = p.gamma[pi]; c.bt = p.beta[pi];
        c.rmean = 0.f; c.rvar = 1.f;
        if (!TRAIN || pid == 0) { c.rmean = p.running_mean[pi]; c.rvar = p.running_var[pi]; }
    };
    // two register sets, used alternately by even and odd blocks (a copy between them would wait for the prefetch)
    f32x4 wA[NT], wB[NT];
    ColP cA = {0.f, 1.f, 0.f, 0.f, 1.f}, cB = cA;
    if (depth > 0 && wv < NT) {  // the longest latency of the prologue goes out first
        load_w(0, wA);
        load_colp(0, cA);
    }

    // ---- input layer (no activation) ---------------------------------------------------------------------------------
    // din <= 32: x rows and w_in staged in LDS by coalesced loads that are all in flight together; thread (c, rq) then
    // takes rows 4rq .. 4rq+3 of column c.  Wider inputs read both operands from global memory inside the k loop.
    const bool staged = p.din <= PN_STAGE_DIN;
    if (staged) {
        for (int idx = t; idx < PN_R * p.din; idx += PN_T) {
            const int r = idx / p.din, k = idx - r * p.din;
            s_x[r][k] = p.x[(size_t)min(r0 + r, B - 1) * p.din + k];
        }
        for (int idx = t; idx < W * p.din; idx += PN_T) {
            const int c = idx / p.din, k = idx - c * p.din;
            s_win[c][k] = p.w_in[idx];
        }
    }

    // per-depth coefficient of the likelihood gradient: d(-objective/N)/d act = coef_d * (y - act)
    const bool has_obj = p.y != nullptr;
    if (has_obj && wv == PN_T / 64 - 1) {
        const float sigma = fmaxf(expf(p.log_std[0]), 1e-4f);
        const float scale = -p.inv_batch / (sigma * sigma);
        float mx = 0.f, ssum = 1.f;
        if (p.qw && p.w_is_logits) {
            mx = -INFINITY;
            for (int d = lane; d < D; d += 64) mx = fmaxf(mx, p.qw[d]);
            mx = wave_allmax(mx);
            ssum = 0.f;
            for (int d = lane; d < D; d += 64) ssum += expf(p.qw[d] - mx);
            ssum = wave_allsum(ssum);
        }
        for (int d = lane; d < D; d += 64) {
            float wd = 0.f;
            if (p.qw) wd = p.w_is_logits ? expf(p.qw[d] - mx) / ssum : p.qw[d];
            s_coef[d] = wd * scale;
        }
    }

    // head: thread (row = t >> 5, part = t & 31) holds four columns of its row
    const int hrow = t >> 5, hpart = t & 31;
    f32x4 wo4 = {0.f, 0.f, 0.f, 0.f};
    if (hpart * 4 < W) wo4 = *reinterpret_cast<const f32x4*>(p.w_out + hpart * 4);  // (W % 4 == 0; w_out 16-byte aligned: host)
    const float b0 = p.b_out[0];
    const float yv = has_obj ? p.y[min(r0 + hrow, B - 1)] : 0.f;
    const bool want_dw = p.dwpart != nullptr;
    f32x4 gw = {0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;

    PN_STAMP(91);
    if (staged) pn_lds_barrier();  // s_x, s_win
    {
        const int c = t & 127, rq = t >> 7;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < W) {
            if (staged) {
                for (int k = 0; k < p.din; ++k) {
                    const float wk = s_win[c][k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = fmaf(s_x[4 * rq + j][k], wk, a[j]);
                }
            } else {
                const float* wr = p.w_in + (size_t)c * p.din;
                const float* xr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[j] = p.x + (size_t)min(r0 + 4 * rq + j, B - 1) * p.din;
#pragma unroll 4
                for (int k = 0; k < p.din; ++k) {
                    const float wk = wr[k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = fmaf(xr[j][k], wk, a[j]);
                }
            }
            const float bi = p.b_in[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += bi;
        }
        if (c < WP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 4 * rq + j;
                const bool live = c < W && row < nvalid;
                hT[0][row][c] = live ? a[j] : 0.f;
                if (live) p.H[(size_t)(r0 + row) * W + c] = a[j];
            }
        }
    }

    auto head = [&](int d, const float (*hb)[LD]) {
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if (hpart * 4 < WP) hv = *reinterpret_cast<const f32x4*>(&hb[hrow][hpart * 4]);
        float dot = (hv[0] * wo4[0] + hv[1] * wo4[1]) + (hv[2] * wo4[2] + hv[3] * wo4[3]);
        dot = pn_row16_sum(dot);
        dot += __shfl_xor(dot, 16, 64);
        const float a = dot + b0;
        const bool live = hrow < nvalid;
        float g = 0.f, rr = 0.f;
        if (has_obj) {
            const float r = yv - a;
            rr = live ? r * r : 0.f;
            g = live ? s_coef[d] * r : 0.f;
        }
        if (hpart == 0) {
            if (live) {
                p.act[(size_t)d * B + r0 + hrow] = a;
                if (p.dact) p.dact[(size_t)d * B + r0 + hrow] = g;
            }
            s_rsq[d][hrow] = rr;
        }
        if (want_dw) {
            gw += g * hv;
            if (hpart == 0) gb += g;
        }
    };

    PN_STAMP(0);
    pn_lds_barrier();  // h_0 tile, s_coef, s_abort
    PN_STAMP(1);

    // Block i: matmul -> statistics -> publish -> [head of depth i on the tile just consumed: it runs while the records
    // travel] -> gather -> [operands of block i+1 requested] -> normalise into the other tile -> barrier.
    auto block = [&](int i, f32x4 (&wq)[NT], ColP& cp, f32x4 (&wnext)[NT], ColP& cpn) {
        const float (*hc)[LD] = hT[i & 1];
        float(*hn)[LD] = hT[(i + 1) & 1];
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned tag = base + 1u + (unsigned)i;
        if (wv < NT) {
            PN_STAMP(2 + 6 * i);
            const f32x4 acc = pn_tile_matmul<NT>(&hc[0][0], LD, wq);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = fmaxf(acc[j] + cp.bias, 0.f);
            if (TRAIN) {
                // Welford partial of this workgroup's rows: mean, then M2 around it (two passes over four registers)
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) sm += (4 * q + j < nvalid) ? z[j] : 0.f;
                const float mean_p = pn_sum_groups(sm) / (float)nvalid;
                float m2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dlt = z[j] - mean_p;
                    m2 += (4 * q + j < nvalid) ? dlt * dlt : 0.f;
                }
                m2 = pn_sum_groups(m2);
                PN_STAMP(3 + 6 * i);
                if (q == 0 && colok) pn_publish(xch, i & 1, col, mean_p, m2, tag);
            }
        }
        head(i, hc);
        if (wv < NT) {
            float mu, rs;
            if (TRAIN) {
                double t1, t2;
                PN_STAMP(4 + 6 * i);
                pn_gather<true>(xch, i & 1, min(col, W - 1), colok, tag, B, 0x100u + (unsigned)i, t1, t2);
                PN_STAMP(5 + 6 * i);
                if (i + 1 < depth) {  // behind the gather: its loads are not queued behind these
                    load_w(i + 1, wnext);
                    load_colp(i + 1, cpn);
                }
                const double inv_n = 1.0 / (double)B;
                const double mean = t1 * inv_n;
                double var = t2 * inv_n - mean * mean;
                var = var > 0.0 ? var : 0.0;
                mu = (float)mean;
                rs = (float)rsqrt(var + (double)p.eps);
                if (colok) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j < nvalid) p.Z[((size_t)i * B + r0 + 4 * q + j) * W + col] = z[j];
                }
                if (pid == 0 && q == 0 && colok) {
                    const size_t pi = (size_t)i * W + col;
                    const double mom = (double)p.momentum, n = (double)B;
                    p.bn_mean[pi] = mu;
                    p.bn_rstd[pi] = rs;
                    p.running_mean[pi] = (float)((1.0 - mom) * (double)cp.rmean + mom * mean);
                    p.running_var[pi] = (float)((1.0 - mom) * (double)cp.rvar + mom * (var * (n / (n - 1.0))));
                }
            } else {
                if (i + 1 < depth) {
                    load_w(i + 1, wnext);
                    load_colp(i + 1, cpn);
                }
                mu = cp.rmean;
                rs = 1.f / sqrtf(cp.rvar + p.eps);
            }
            const float sc = cp.gm * rs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 4 * q + j;
                float o = (z[j] - mu) * sc + cp.bt;
                if (p.residual) o = hc[row][col] + o;
                hn[row][col] = colok ? o : 0.f;
                if (colok && row < nvalid) p.H[((size_t)(i + 1) * B + r0 + row) * W + col] = o;
            }
            PN_STAMP(6 + 6 * i);
        }
        pn_lds_barrier();  // h_{i+1} tile complete; every wave is done with the A operand (and the head) of block i
        PN_STAMP(7 + 6 * i);
    };
    for (int i = 0; i < depth; i += 2) {
        block(i, wA, cA, wB, cB);
        if (i + 1 < depth) block(i + 1, wB, cB, wA, cA);
    }
    head(depth, hT[depth & 1]);
    PN_STAMP(92);

    // ---- what the workgroup leaves behind -------------------------------------------------------------------------------
    if (want_dw) {
        if (hpart * 4 < WP) *reinterpret_cast<f32x4*>(&s_gw[hrow][hpart * 4]) = gw;
        if (hpart == 0) s_gb[hrow] = gb;
    }
    __syncthreads();  // s_rsq, s_gw, s_gb, s_abort
    const bool lost = s_abort != 0;
    if (want_dw) {
        if (t < W) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < PN_R; ++r) a += s_gw[r][t];
            p.dwpart[(size_t)pid * W + t] = a;
        }
        if (t == 0) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < PN_R; ++r) a += s_gb[r];
            p.dbpart[pid] = a;
        }
    }
    if (pid == 0) {
        if (TRAIN && p.nbt != nullptr && t < depth) p.nbt[t] += 1;
        // tags of the next launch start behind this one's (every workgroup has read `base` by now: it published into
        // the last exchange this workgroup completed)
        if (exchanging && t == 0) __hip_atomic_store(p.sync, base + (unsigned)depth + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lost) {  // an exchange gave up: nothing computed from it may look valid
        const float nanv = __builtin_nanf("");
        for (int idx = t; idx < D * PN_R; idx += PN_T) {
            const int d = idx / PN_R, r = idx % PN_R;
            if (r < nvalid) p.act[(size_t)d * B + r0 + r] = nanv;
        }
    }
    PN_STAMP(93);
    if (has_obj && p.partials != nullptr) {
        const bool fin = p.fin_result != nullptr;
        unsigned long long* pw = reinterpret_cast<unsigned long long*>(p.partials);
        for (int d = t; d < D; d += PN_T) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < PN_R; ++r) a += (double)s_rsq[d][r];
            if (lost) a = (double)__builtin_nanf("");
            // (write-through: the finalising workgroup reads them inside this launch)
            __hip_atomic_store(pw + (size_t)d * p.nparts + pid, __builtin_bit_cast(unsigned long long, a), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (fin) {
            // every storing wave drains its stores, then ONE lane takes a ticket; the workgroup whose ticket is the last
            // reads all partials (sc1 loads) and finishes the objective
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                const unsigned tk = __hip_atomic_fetch_add(p.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk == (unsigned)p.P - 1u) __hip_atomic_store(p.sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ticket = (int)tk;
            }
            __syncthreads();
            if (s_ticket == p.P - 1) {
                for (int d = wv; d < D; d += PN_T / 64) {
                    double a = 0.0;
                    for (int i = lane; i < p.nparts; i += 64)
                        a += __builtin_bit_cast(double, __hip_atomic_load(pw + (size_t)d * p.nparts + i, __ATOMIC_RELAXED,
                                                                          __HIP_MEMORY_SCOPE_AGENT));
                    a = wave_sum(a);
                    if (lane == 0) {
                        s_sums[d] = a;
                        if (p.fin_sums) p.fin_sums[d] = a;
                    }
                }
                __syncthreads();
                if (wv == 0)
                    depth_finalize_body(s_sums, 1, p.fin_q_logits, p.fin_logprior, p.fin_prior_ll, p.log_std, D, p.fin_N,
                                        p.fin_batch, p.fin_beta, p.fin_eps, p.fin_objective, p.fin_result);
            }
        }
    }
    PN_STAMP(94);
}

// ---- backward --------------------------------------------------------------------------------------------------------
// Work of the backward launch that sits on no critical path, done by ONE extra workgroup (index P) beside the row
// owners: the head's weight / bias gradient from the partial sums the forward left ([P][W], [P]; fp64, fixed order) and
// the depth-level finalisation of the objective the forward deferred (dun_depth_finalize_partials' arithmetic and
// summation order).  Both read only what earlier launches wrote.
struct PnSide {
    const float *dwpart, *dbpart;  // NULL: the head gradients come from elsewhere
    float *dw_out, *db_out;
    int P, W;
    float* fin_result;             // NULL: no deferred finalisation
    int fin_D, fin_nparts, fin_objective;
    const double* fin_partials;
    double* fin_sums;
    const float *fin_q_logits, *fin_logprior, *fin_prior_ll, *fin_log_std;
    double fin_N, fin_batch, fin_beta, fin_eps;
};

__device__ __forceinline__ void pn_side_work(const PnSide& sd) {
    __shared__ double s_side[512];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (sd.dwpart != nullptr) {
        for (int c = wv; c <= sd.W; c += PN_T / 64) {  // column W: the bias gradient
            double a = 0.0;
            for (int q = lane; q < sd.P; q += 64) a += (double)(c < sd.W ? sd.dwpart[(size_t)q * sd.W + c] : sd.dbpart[q]);
            a = wave_sum(a);
            if (lane == 0) {
                if (c < sd.W) sd.dw_out[c] = (float)a;
                else sd.db_out[0] = (float)a;
            }
        }
    }
    if (sd.fin_result != nullptr) {
        for (int d = wv; d < sd.fin_D; d += PN_T / 64) {
            double a = 0.0;
            for (int k = lane; k < sd.fin_nparts; k += 64) a += sd.fin_partials[(size_t)d * sd.fin_nparts + k];
            a = wave_sum(a);
            if (lane == 0) {
                s_side[d] = a;
                if (sd.fin_sums) sd.fin_sums[d] = a;
            }
        }
        __syncthreads();
        if (wv == 0)
            depth_finalize_body(s_side, 1, sd.fin_q_logits, sd.fin_logprior, sd.fin_prior_ll, sd.fin_log_std, sd.fin_D,
                                sd.fin_N, sd.fin_batch, sd.fin_beta, sd.fin_eps, sd.fin_objective, sd.fin_result);
    }
}

struct PnBwdP {
    PnSide side;
    int B, W, L, residual, P;
    const float *w, *w_out, *gamma;
    const float *Z, *bn_mean, *bn_rstd;
    const float* dact;  // [L+1][B]
    float* dU;          // [L][B][W]
    float* G;           // [B][W]: d loss / d h_0 (the input layer's output), for the weight-gradient launch
    float *dgamma, *dbeta;
    unsigned* sync;
    int sync_bytes;
};

template <int NT>
__global__ __launch_bounds__(PN_T) void pn_backward_kernel(PnBwdP p) {
    constexpr int WP = 16 * NT, LD = WP + 4;
    __shared__ __attribute__((aligned(16))) float duT[2][PN_R][LD];
    __shared__ int s_abort;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pid = blockIdx.x, r0 = pid * PN_R;
    if (pid == p.P) {  // the extra workgroup (launched only when there is side work)
        pn_side_work(p.side);
        return;
    }
    const int B = p.B, W = p.W, L = p.L;
    const int nvalid = min(PN_R, B - r0);

    PnXch xch;
    xch.rs = __builtin_amdgcn_make_buffer_rsrc(p.sync, 0, p.sync_bytes, 0x00020000);
    xch.words = p.sync;
    xch.abort_flag = &s_abort;
    xch.P = p.P;
    xch.pid = pid;
    PN_STAMP(90);
    unsigned base = 0;
    if (L > 0) base = __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) s_abort = 0;
    __syncthreads();

    const int col = 16 * wv + (lane & 15), q = lane >> 4;
    const bool colok = wv < NT && col < W;
    const int colc = min(col, W - 1);
    const float wo = colok ? p.w_out[colc] : 0.f;

    auto dact_rows = [&](int d, float (&da)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 4 * q + j;
            const float v = p.dact[(unsigned)d * (unsigned)B + (unsigned)min(row, B - 1)];
            da[j] = v * (row < B ? 1.f : 0.f);
        }
    };
    // operand of g = dU W_i: the contraction runs over W_i's OUTPUT index, wq[s][e] = W_i[16 s + 4 q + e][col]
    auto load_w = [&](int i, f32x4 (&dst)[NT]) {
        const unsigned wi = (unsigned)i * (unsigned)(W * W) + (unsigned)colc;  // (32-bit offsets: see the forward)
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * s + 4 * q + e;
                const float v = p.w[wi + (unsigned)min(k, W - 1) * (unsigned)W];
                dst[s][e] = v * ((colok && k < W) ? 1.f : 0.f);  // (mask, not select: see the forward's load_w)
            }
    };
    auto load_z = [&](int i, float (&z)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            z[j] = p.Z[((unsigned)i * (unsigned)B + (unsigned)min(r0 + 4 * q + j, B - 1)) * (unsigned)W + (unsigned)colc];
    };

    // operands of a block: BatchNorm statistics and scale of this lane's column, its rows of ReLU(Linear) and of
    // d loss / d act, this wave's slice of the weights.  Two register sets, used alternately by successive blocks.
    struct Ops {
        float mu, rs, gm;
        float z[4], da[4];
        f32x4 wq[NT];
    };
    auto load_ops = [&](int i, Ops& o) {
        const unsigned pi = (unsigned)i * (unsigned)W + (unsigned)colc;
        o.mu = p.bn_mean[pi]; o.rs = p.bn_rstd[pi]; o.gm = p.gamma[pi];
        load_z(i, o.z);
        dact_rows(i, o.da);
        load_w(i, o.wq);
    };
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    Ops oA, oB;
    if (wv < NT) {
        if (L > 0) load_ops(L - 1, oA);  // the long latencies first
        float da[4];
        dact_rows(L, da);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = da[j] * wo;
    }

    // Block i: column sums of g and g*xhat -> publish -> gather -> [operands of block i-1 requested] -> dU into LDS and
    // global memory -> barrier -> g = [g +] dU W_i + dact_i w_out on the matrix cores.
    auto block = [&](int i, Ops& o, Ops& on) {
        const int phase = L - 1 - i;
        float(*dt)[LD] = duT[phase & 1];
        PN_STAMP(1 + 5 * phase);
        if (wv < NT) {
            float xh[4], sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[j] = (o.z[j] - o.mu) * o.rs;
                sg += g[j];            // (rows past the batch carry g = 0)
                sgx += g[j] * xh[j];
            }
            sg = pn_sum_groups(sg);
            sgx = pn_sum_groups(sgx);
            const unsigned tag = base + 1u + (unsigned)phase;
            PN_STAMP(2 + 5 * phase);
            if (q == 0 && colok) pn_publish(xch, phase & 1, col, sg, sgx, tag);
            double t1, t2;
            pn_gather<false>(xch, phase & 1, colc, colok, tag, B, 0x200u + (unsigned)phase, t1, t2);
            PN_STAMP(3 + 5 * phase);
            if (i > 0) load_ops(i - 1, on);  // behind the gather: its loads are not queued behind these
            const double inv_n = 1.0 / (double)B;
            const float mg = (float)(t1 * inv_n), mgx = (float)(t2 * inv_n);
            if (pid == 0 && q == 0 && colok) {
                p.dbeta[(size_t)i * W + col] = (float)t1;
                p.dgamma[(size_t)i * W + col] = (float)t2;
            }
            const float k0 = o.gm * o.rs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 4 * q + j;
                const bool live = colok && row < nvalid;
                const float du = (live && o.z[j] > 0.f) ? k0 * (g[j] - mg - xh[j] * mgx) : 0.f;
                dt[row][col] = du;
                if (live) p.dU[((size_t)i * B + r0 + row) * W + col] = du;
            }
        }
        PN_STAMP(4 + 5 * phase);
        pn_lds_barrier();  // dU tile of this block complete
        PN_STAMP(5 + 5 * phase);
        if (wv < NT) {
            const f32x4 acc = pn_tile_matmul<NT>(&dt[0][0], LD, o.wq);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = (p.residual ? g[j] : 0.f) + acc[j] + o.da[j] * wo;
        }
        // (two dU slots: a wave writes slot (phase+1)&1 only after the barrier of phase, behind which every wave has finished
        // reading slot (phase-1)&1 = (phase+1)&1 in the matmul of phase-1)
    };
    for (int i = L - 1; i >= 0; i -= 2) {
        block(i, oA, oB);
        if (i > 0) block(i - 1, oB, oA);
    }
    if (colok) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < nvalid) p.G[(size_t)(r0 + 4 * q + j) * W + col] = g[j];
    }
    if (pid == 0 && t == 0 && L > 0)
        __hip_atomic_store(p.sync, base + (unsigned)L + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PN_STAMP(94);
    __syncthreads();
    if (s_abort && colok) {  // an exchange gave up: poison what the weight-gradient launch reads
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < nvalid) p.G[(size_t)(r0 + 4 * q + j) * W + col] = __builtin_nanf("");
    }
}

// ---- weight gradients ---------------------------------------------------------------------------------------------------
// dW_i[n][k] = sum_b dU_i[b][n] H_i[b][k] and db_i[n] = sum_b dU_i[b][n] for every block, and the same for the input
// layer (dU := d loss / d h_0, H := x), in ONE launch behind the trunk's: workgroup (x, y, z) owns the 16 x 16 tile
// (n-tile x, k-tile y) of layer z; its eight waves split the batch, each contracting its rows on
// v_mfma_f32_16x16x4_f32 with both operands read straight from global memory (all loads of a batch of 16 MFMA steps in
// flight together, branch-free), and the eight partial tiles are added through LDS in a fixed order.  The bias gradient
// is the virtual column k = K of ones.  (smallnet.hip's column-owned form of this launch re-read the whole H_i per
// workgroup through LDS: 13 us for the UCI net against ~6 here.)
struct PnDwP {
    const float* dU;   // [L][B][W]
    const float* H;    // [L+1][B][W]
    const float* G;    // [B][W]
    const float* x;    // [B][din]
    float *dW, *db;    // [L][W][W], [L][W]
    float *dw_in, *db_in;
    int B, W, L, din;
};

__global__ __launch_bounds__(PN_T) void pn_dw_kernel(PnDwP p) {
    __shared__ __attribute__((aligned(16))) f32x4 s_acc[PN_T / 64][64];
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int layer = blockIdx.z, n0 = blockIdx.x * 16, k0 = blockIdx.y * 16;
    const bool input = layer == p.L;
    const int K = input ? p.din : p.W, ldh = K, B = p.B, W = p.W;
    if (k0 > K) return;  // (uniform) this layer has no such tile
    const float* A = input ? p.G : p.dU + (size_t)layer * B * W;  // [B][W]
    const float* Hm = input ? p.x : p.H + (size_t)layer * B * W;  // [B][K]
    // rows of this wave: a contiguous slice, a multiple of four rows long
    const int per = (((B + 7) / 8) + 3) & ~3;
    const int b0 = wv * per, b1 = min(B, b0 + per);
    const int m = lane & 15, kq = lane >> 4;
    const int n = n0 + m, kk = k0 + m;
    const float amask = n < W ? 1.f : 0.f;
    const float hmask = kk < K ? 1.f : 0.f, ones = kk == K ? 1.f : 0.f;
    const unsigned acol = (unsigned)min(n, W - 1), hcol = (unsigned)min(kk, K - 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int CH = 16;
    for (int s0 = b0; s0 < b1; s0 += 4 * CH) {
        float av[CH], hv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int b = s0 + 4 * u + kq;
            const unsigned bc = (unsigned)min(b, B - 1);
            const float live = b < b1 ? 1.f : 0.f;
            av[u] = A[bc * (unsigned)W + acol] * (amask * live);
            hv[u] = Hm[bc * (unsigned)ldh + hcol] * hmask + ones;
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], hv[u], acc, 0, 0, 0);
    }
    s_acc[wv][lane] = acc;
    __syncthreads();
    if (wv == 0) {
        f32x4 r = s_acc[0][lane];
#pragma unroll
        for (int w = 1; w < PN_T / 64; ++w) r += s_acc[w][lane];
        float* dW = input ? p.dw_in : p.dW + (size_t)layer * W * W;
        float* db = input ? p.db_in : p.db + (size_t)layer * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = n0 + 4 * kq + j;  // result layout: row 4 (lane >> 4) + j, column lane & 15
            if (row < W) {
                if (kk < K) dW[(size_t)row * K + kk] = r[j];
                else if (kk == K) db[row] = r[j];
            }
        }
    }
}

int pn_dw_all(const dun_fc_ctx* c, const float* dUall, hipStream_t s) {
    PnDwP p;
    memset(&p, 0, sizeof(p));
    p.dU = dUall; p.H = c->H; p.G = c->G; p.x = c->x;
    p.dW = c->g.w; p.db = c->g.b; p.dw_in = c->g.w_in; p.db_in = c->g.b_in;
    p.B = c->B; p.W = c->W; p.L = c->L; p.din = c->din;
    const int kmax = c->W > c->din ? c->W : c->din;
    DUN_REQUIRE((long long)c->B * kmax < (1LL << 31), "dun_fc_backward: input too wide for the one-launch path");
    hipLaunchKernelGGL(pn_dw_kernel, dim3((c->W + 15) / 16, kmax / 16 + 1, c->L + 1), dim3(PN_T), 0, s, p);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------
static inline int pn_tiles(int W) { return W <= 32 ? 2 : (W <= 64 ? 4 : (W <= 112 ? 7 : 8)); }

int pn_forward(const dun_fc_ctx* c, hipStream_t s) {
    if (c->train && c->depth > 0)
        DUN_REQUIRE(c->B > 1, "dun_fc_forward: BatchNorm in train mode needs more than 1 row (got %d)", c->B);
    DUN_REQUIRE(aligned16(c->p.w_out), "dun_fc_forward: w_out must be 16-byte aligned");
    PnFwdP p;
    memset(&p, 0, sizeof(p));
    p.B = c->B; p.din = c->din; p.W = c->W; p.depth = c->depth; p.residual = c->residual;
    p.P = pn_workgroups(c->B);
    p.x = c->x;
    p.w_in = c->p.w_in; p.b_in = c->p.b_in; p.w_out = c->p.w_out; p.b_out = c->p.b_out;
    p.w = c->p.w; p.b = c->p.b; p.gamma = c->p.gamma; p.beta = c->p.beta;
    p.running_mean = c->running_mean; p.running_var = c->running_var;
    p.H = c->H; p.Z = c->Z; p.bn_mean = c->bn_mean; p.bn_rstd = c->bn_rstd; p.act = c->act;
    p.nbt = reinterpret_cast<long long*>(c->num_batches_tracked);
    p.eps = c->bn_eps; p.momentum = c->bn_momentum;
    p.sync = c->sync; p.sync_bytes = (int)pn_sync_bytes(c->B, c->W);
    if (c->obj_y) {
        p.y = c->obj_y; p.qw = c->obj_w; p.w_is_logits = c->obj_w_is_logits; p.log_std = c->obj_log_std;
        p.inv_batch = c->obj_inv_batch;
        p.dact = c->obj_w ? c->obj_dact : nullptr;
        p.partials = c->obj_partials;
        p.nparts = p.P;
        if (p.dact && c->obj_dwpart && c->obj_dbpart && c->train) { p.dwpart = c->obj_dwpart; p.dbpart = c->obj_dbpart; }
        if (c->obj_fin_result && !c->obj_fin_defer) {
            DUN_REQUIRE(c->obj_fin_logprior != nullptr, "dun_fc_forward: obj_fin_result without obj_fin_logprior");
            DUN_REQUIRE(c->obj_fin_objective == DUN_OBJ_MLL || c->obj_fin_q_logits != nullptr,
                        "dun_fc_forward: the variational objective needs obj_fin_q_logits");
            p.fin_result = c->obj_fin_result; p.fin_sums = c->obj_fin_sums; p.fin_q_logits = c->obj_fin_q_logits;
            p.fin_logprior = c->obj_fin_logprior; p.fin_prior_ll = c->obj_fin_prior_ll;
            p.fin_N = c->obj_fin_N_train; p.fin_batch = c->obj_fin_batch; p.fin_beta = c->obj_fin_beta;
            p.fin_eps = c->obj_fin_eps; p.fin_objective = c->obj_fin_objective;
        }
    }
    const dim3 grid(p.P), block(PN_T);
#define PN_FWD(N)                                                                                   \
    case N:                                                                                         \
        if (c->train) hipLaunchKernelGGL((pn_forward_kernel<N, true>), grid, block, 0, s, p);       \
        else hipLaunchKernelGGL((pn_forward_kernel<N, false>), grid, block, 0, s, p);               \
        break;
    switch (pn_tiles(c->W)) { PN_FWD(2) PN_FWD(4) PN_FWD(7) PN_FWD(8) }
#undef PN_FWD
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

int pn_backward_trunk(const dun_fc_ctx* c, float* dUall, bool head_from_forward, hipStream_t s) {
    PnBwdP p;
    memset(&p, 0, sizeof(p));
    bool side = false;
    if (head_from_forward) {
        p.side.dwpart = c->obj_dwpart; p.side.dbpart = c->obj_dbpart; p.side.dw_out = c->g.w_out; p.side.db_out = c->g.b_out;
        p.side.P = pn_workgroups(c->B); p.side.W = c->W;
        side = true;
    }
    if (c->obj_fin_result && c->obj_fin_defer && c->obj_y && c->obj_partials) {
        // the forward left the partial sums and deferred the depth-level finalisation to this launch
        DUN_REQUIRE(c->obj_fin_logprior != nullptr && (c->obj_fin_objective == DUN_OBJ_MLL || c->obj_fin_q_logits != nullptr),
                    "dun_fc_backward: deferred finalisation without logprior / q_logits");
        DUN_REQUIRE(c->depth + 1 <= 512, "dun_fc_backward: deferred finalisation of more than 512 depths");
        p.side.fin_result = c->obj_fin_result; p.side.fin_D = c->depth + 1; p.side.fin_nparts = pn_workgroups(c->B);
        p.side.fin_objective = c->obj_fin_objective; p.side.fin_partials = c->obj_partials; p.side.fin_sums = c->obj_fin_sums;
        p.side.fin_q_logits = c->obj_fin_q_logits; p.side.fin_logprior = c->obj_fin_logprior;
        p.side.fin_prior_ll = c->obj_fin_prior_ll; p.side.fin_log_std = c->obj_log_std;
        p.side.fin_N = c->obj_fin_N_train; p.side.fin_batch = c->obj_fin_batch; p.side.fin_beta = c->obj_fin_beta;
        p.side.fin_eps = c->obj_fin_eps;
        side = true;
    }
    p.B = c->B; p.W = c->W; p.L = c->L; p.residual = c->residual; p.P = pn_workgroups(c->B);
    p.w = c->p.w; p.w_out = c->p.w_out; p.gamma = c->p.gamma;
    p.Z = c->Z; p.bn_mean = c->bn_mean; p.bn_rstd = c->bn_rstd;
    p.dact = c->dact; p.dU = dUall; p.G = c->G;
    p.dgamma = c->g.gamma; p.dbeta = c->g.beta;
    p.sync = c->sync; p.sync_bytes = (int)pn_sync_bytes(c->B, c->W);
    const dim3 grid(p.P + (side ? 1 : 0)), block(PN_T);
#define PN_BWD(N) case N: hipLaunchKernelGGL((pn_backward_kernel<N>), grid, block, 0, s, p); break;
    switch (pn_tiles(c->W)) { PN_BWD(2) PN_BWD(4) PN_BWD(7) PN_BWD(8) }
#undef PN_BWD
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

}  // namespace dun

using namespace dun;

extern "C" int dun_fc_small_net_mode(int mode) {
    const int prev = small_net_mode();
    if (mode >= 0 && mode <= 2) g_small_net_mode = mode;
    return prev;
}

extern "C" size_t dun_fc_sync_bytes(int B, int W) { return pn_sync_bytes(B, W); }

extern "C" int dun_fc_sync_status(const dun_fc_ctx* c, dun_stream_t stream) {
    if (c == nullptr || c->sync == nullptr) return 0;
    unsigned code = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&code, c->sync + 1, sizeof(code), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    return (int)code;
}
