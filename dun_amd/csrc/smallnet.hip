// Whole-block kernels for SMALL residual-MLP DUNs (the toy / UCI configurations: width <= 128, batch <= 2048).
//
// At these sizes a block is ~10 Mflop: the tiled-GEMM path (trunk.hip) spends its time in launches -- GEMM, column
// reduction, statistics, normalise forward; two reductions, two elementwise passes and two GEMMs backward -- because
// BatchNorm1d (src/DUN/layers.py:38) needs the whole batch of a column between the Linear and the residual add.
// Here a workgroup OWNS four feature columns for ALL rows of the batch instead, so everything that couples rows is
// local to it and a block is ONE launch forward and ONE launch backward:
//   forward : z[:,c] = ReLU(h_i W_i[c,:]^T + b_i[c]); batch statistics of its columns (fp64); running-stat update;
//             h_{i+1}[:,c] = [h_i[:,c] +] gamma (z - mean) rstd + beta.
//   backward: g[:,c] = [g[:,c] +] dU_{i+1} W_{i+1}[:,c] + dact_{i+1} w_out[c]   (the dX of the DEEPER block, whose dU
//             the previous launch left in global memory: the only data that crosses workgroups);
//             BatchNorm+ReLU backward on its columns -> dU_i[:,c], dgamma, dbeta, db_i;  dW_i[c,:] = dU_i[:,c]^T h_i.
//   The last backward launch does the same gradient update for the input layer and dW_in = g^T x, db_in.
// The contractions run on the vector ALUs (fmaf chains): every operand row is staged once through LDS with coalesced
// 16-byte loads and read back row-per-thread, the four weight rows/columns of the workgroup are wave-uniform loads.
// MFMA tiles would leave W/16 = 7 workgroups on a 256-CU chip; this regime is latency bound, not a roofline claim.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace dun {

constexpr int SN_T = 1024;        // threads per workgroup: 16 waves, four per SIMD (latency hiding: 25 workgroups on 256 CUs)
constexpr int SN_R = 256;         // rows per staged chunk (one per thread of a K part)
constexpr int SN_KP = SN_T / SN_R;  // K parts: waves 4p..4p+3 contract the p-th quarter of K for the chunk's 256 rows
constexpr int SN_NC = 4;          // columns owned by a workgroup
constexpr int SN_MAX_RPT = 8;     // row chunks per workgroup -> B <= 2048 (g and z of every chunk stay in registers)
constexpr int SN_MAX_W = 128;
constexpr size_t SN_LDS_LIMIT = 160 * 1024;
constexpr int SN_STAGE_U = 8;     // 16-byte loads in flight per thread while staging
constexpr int SN_DW_U = 8;        // rows in flight per thread in the dW phase

// row stride (floats) of the staged tile: S/4 odd, so that 16 consecutive lanes reading 16 bytes of their own row
// hit 64 different banks
static inline int sn_stride(int K) { return ((K >> 2) & 1) ? K : K + 4; }
static inline size_t sn_tile_floats(int W) {
    const size_t a = (size_t)SN_R * sn_stride(W);
    return a > 8192 ? a : 8192;  // the dW phase reuses the tile as its [16][4][128] reduction buffer
}
// dynamic LDS carve (every offset a multiple of 16 bytes):
//   [red: 4 x 8 doubles = 256 B][part: SN_KP x 256 rows x 4 floats = 16 KiB][dul: B x 4 floats (backward)][tile]
constexpr size_t SN_OFF_PART = 256;
constexpr size_t SN_OFF_DUL = SN_OFF_PART + (size_t)SN_KP * SN_R * 16;
static inline size_t sn_fwd_lds(int W) { return SN_OFF_DUL + sn_tile_floats(W) * 4; }
static inline size_t sn_bwd_lds(int B, int W) { return SN_OFF_DUL + (size_t)B * 16 + sn_tile_floats(W) * 4; }

static const bool g_sn_off = getenv("DUN_SMALLNET") != nullptr && getenv("DUN_SMALLNET")[0] == '0';

bool sn_shape_ok(int B, int W) {
    if (g_sn_off) return false;
    if (W < 4 || W > SN_MAX_W || (W & 3) || B < 1 || B > SN_R * SN_MAX_RPT) return false;
    return sn_bwd_lds(B, W) <= SN_LDS_LIMIT;
}
size_t sn_extra_ws_floats(int B, int W) { return sn_shape_ok(B, W) ? (size_t)B * W : 0; }

// sum of the values held by threads 0..255 (the threads that own rows); every thread of the workgroup gets the
// totals.  The values go through LDS transposed ([NV][256] doubles in `xbuf`) so that only NV waves shuffle, one
// double each: sixteen waves butterflying NV doubles apiece kept the LDS crossbar busy for 6 us.
template <int NV>
__device__ __forceinline__ void sn_block_sum(double (&v)[NV], double* red, double* xbuf) {
    const int t = threadIdx.x;
    if (t < SN_R) {
#pragma unroll
        for (int i = 0; i < NV; ++i) xbuf[i * SN_R + t] = v[i];
    }
    __syncthreads();
    if (t < NV * 64) {
        const int i = t >> 6, l = t & 63;
        const double* x = xbuf + i * SN_R + l;
        double a = (x[0] + x[64]) + (x[128] + x[192]);
        a = wave_sum(a);
        if (l == 0) red[i] = a;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i];
    __syncthreads();
}

// `rows` consecutive rows of a dense row-major [., W] matrix -> LDS tile with row stride S (coalesced 16-byte loads,
// up to SN_STAGE_U of them in flight per thread).  step_row / step_col = SN_T / (W/4), SN_T % (W/4).
__device__ __forceinline__ void sn_stage(float* tile, const float* src, int rows, int W, int S, int step_row,
                                         int step_col) {
    const int nq = W >> 2;
    const int nv = rows * nq;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    int row = threadIdx.x / nq, cq = threadIdx.x - row * nq;
    for (int base = threadIdx.x; base < nv; base += SN_T * SN_STAGE_U) {
        f32x4 v[SN_STAGE_U];
#pragma unroll
        for (int u = 0; u < SN_STAGE_U; ++u) {
            const int i = base + u * SN_T;
            if (i < nv) v[u] = s4[i];
        }
#pragma unroll
        for (int u = 0; u < SN_STAGE_U; ++u) {
            const int i = base + u * SN_T;
            if (i < nv) *reinterpret_cast<f32x4*>(tile + row * S + cq * 4) = v[u];
            row += step_row;
            cq += step_col;
            if (cq >= nq) { cq -= nq; ++row; }
        }
    }
}

// this wave's share of the W/4 sixteen-byte K steps (wave-uniform)
__device__ __forceinline__ void sn_k_range(int nq, int* k_lo, int* k_hi) {
    const int kp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int per = (nq + SN_KP - 1) / SN_KP;
    *k_lo = min(nq, kp * per);
    *k_hi = min(nq, kp * per + per);
}

// combine the SN_KP partial sums of every row: valid in threads 0..255 (row = thread) after the call.
// The call contains one barrier, which also ends every read of the staged tile.
__device__ __forceinline__ f32x4 sn_combine_parts(f32x4 a, float* part) {
    *reinterpret_cast<f32x4*>(part + 4 * threadIdx.x) = a;  // [kp][row][4]
    __syncthreads();
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < SN_R) {
#pragma unroll
        for (int q = 0; q < SN_KP; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + 4 * (q * SN_R + threadIdx.x));
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] += v[c];
        }
    }
    return r;
}

struct SnFwdP {
    const float* Hin;   // [B,W]
    const float* Wt;    // [W,W] (out,in)
    const float* bias;  // [W]
    const float* gamma;
    const float* beta;
    float* Z;           // [B,W] ReLU(Linear) kept for backward, or NULL (eval)
    float* Hout;        // [B,W]
    float* mean;        // [W] train: batch mean out
    float* rstd;        // [W] train: 1/sqrt(var+eps) out
    float* running_mean;
    float* running_var;
    int B, W, S, train, residual, step_row, step_col;
    float eps, momentum;
    double count;
#ifdef SN_PROFILE
    long long* dbg;  // scripts/dev/sn_phase_probe.hip: [workgroup][32] clock stamps (10 ns units)
#endif
};

#ifdef SN_PROFILE
#define SN_STAMP(i) do { if (threadIdx.x == 0) p.dbg[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define SN_STAMP(i) do { } while (0)
#endif

template <int RPT>
__global__ __launch_bounds__(SN_T) void sn_block_fwd_kernel(SnFwdP p) {
    extern __shared__ __attribute__((aligned(16))) char sn_smem[];
    double* red = reinterpret_cast<double*>(sn_smem);
    float* part = reinterpret_cast<float*>(sn_smem + SN_OFF_PART);
    float* tile = reinterpret_cast<float*>(sn_smem + SN_OFF_DUL);
    const int t = threadIdx.x, c0 = blockIdx.x * SN_NC;
    const int W = p.W, S = p.S, B = p.B, nq = W >> 2;
    const int trow = t & (SN_R - 1);
    int k_lo, k_hi;
    sn_k_range(nq, &k_lo, &k_hi);
    const f32x4* w0 = reinterpret_cast<const f32x4*>(p.Wt + (size_t)(c0 + 0) * W);
    const f32x4* w1 = reinterpret_cast<const f32x4*>(p.Wt + (size_t)(c0 + 1) * W);
    const f32x4* w2 = reinterpret_cast<const f32x4*>(p.Wt + (size_t)(c0 + 2) * W);
    const f32x4* w3 = reinterpret_cast<const f32x4*>(p.Wt + (size_t)(c0 + 3) * W);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c0);
    f32x4 z[RPT];
    SN_STAMP(0);
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r0 = j * SN_R;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (r0 < B) {  // uniform
            sn_stage(tile, p.Hin + (size_t)r0 * W, min(SN_R, B - r0), W, S, p.step_row, p.step_col);
            SN_STAMP(1 + 4 * j);
            __syncthreads();
            SN_STAMP(2 + 4 * j);
            // every thread runs the loop (rows past the batch read stale LDS; their results are never used)
            const float* hrow = tile + trow * S;
#pragma unroll 2
            for (int k4 = k_lo; k4 < k_hi; ++k4) {
                const f32x4 h = *reinterpret_cast<const f32x4*>(hrow + 4 * k4);
                const f32x4 x0 = w0[k4], x1 = w1[k4], x2 = w2[k4], x3 = w3[k4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[0] = fmaf(h[e], x0[e], a[0]);
                    a[1] = fmaf(h[e], x1[e], a[1]);
                    a[2] = fmaf(h[e], x2[e], a[2]);
                    a[3] = fmaf(h[e], x3[e], a[3]);
                }
            }
            SN_STAMP(3 + 4 * j);
            a = sn_combine_parts(a, part);
            SN_STAMP(4 + 4 * j);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) z[j][c] = fmaxf(a[c] + bv[c], 0.f);
    }
    const bool owner = t < SN_R;  // threads 0..255 own row j*256 + t of every chunk
    float mu[4], sc[4];
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c0);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c0);
    // running statistics are read BEFORE anything of this launch is stored: no load waits behind a store later
    const f32x4 rm = *reinterpret_cast<const f32x4*>(p.running_mean + c0);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.running_var + c0);
    if (p.train) {
        double s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.0;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            if (owner && j * SN_R + t < B) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double zd = (double)z[j][c];
                    s[c] += zd;
                    s[4 + c] += zd * zd;
                }
            }
        }
        sn_block_sum<8>(s, red, reinterpret_cast<double*>(tile));
        f32x4 o_mean, o_rstd, o_rm, o_rv;
        const double mom = (double)p.momentum;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double m = s[c] / p.count;
            double var = s[4 + c] / p.count - m * m;
            var = var > 0.0 ? var : 0.0;
            const double rs = 1.0 / sqrt(var + (double)p.eps);
            const double unbiased = var * (p.count / (p.count - 1.0));
            mu[c] = (float)m;
            sc[c] = gm[c] * (float)rs;
            o_mean[c] = (float)m;
            o_rstd[c] = (float)rs;
            o_rm[c] = (float)((1.0 - mom) * (double)rm[c] + mom * m);
            o_rv[c] = (float)((1.0 - mom) * (double)rv[c] + mom * unbiased);
        }
        if (t == 0) {
            *reinterpret_cast<f32x4*>(p.mean + c0) = o_mean;
            *reinterpret_cast<f32x4*>(p.rstd + c0) = o_rstd;
            *reinterpret_cast<f32x4*>(p.running_mean + c0) = o_rm;
            *reinterpret_cast<f32x4*>(p.running_var + c0) = o_rv;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mu[c] = rm[c];
            sc[c] = gm[c] * (1.f / sqrtf(rv[c] + p.eps));
        }
    }
    SN_STAMP(4 * RPT + 1);
    if (!owner) return;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = j * SN_R + t;
        if (r < B) {
            const size_t off = (size_t)r * W + c0;
            if (p.Z) *reinterpret_cast<f32x4*>(p.Z + off) = z[j];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (z[j][c] - mu[c]) * sc[c] + bt[c];
            if (p.residual) {
                const f32x4 hi = *reinterpret_cast<const f32x4*>(p.Hin + off);
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = hi[c] + o[c];
            }
            *reinterpret_cast<f32x4*>(p.Hout + off) = o;
        }
    }
    SN_STAMP(4 * RPT + 2);
}

struct SnBwdP {
    int B, W, S, O, step_row, step_col;
    int keep;             // 1: g += ...  (residual net below the deepest exit), 0: g = ...
    float* G;             // [B,W] running dL/dh, this workgroup's columns updated in place
    const float* dUprev;  // [B,W] dL/d(Linear output) of the deeper block, or NULL (deepest exit)
    const float* Wprev;   // [W,W] (out,in) weights of the deeper block
    const float* dact;    // [B,O] d loss / d act of the exit that reads this hidden state
    const float* w_out;   // [O,W]
    // BatchNorm + ReLU backward of this block (HAS_BN)
    const float* Z;
    const float* mean;
    const float* rstd;
    const float* gamma;
    float* dgamma;
    float* dbeta;
    float* dbias;         // [W]: db_i (HAS_BN) or db_in
    float* dU;            // [B,W] out (HAS_BN)
    double count;
    // weight gradient: dW[c,k] = sum_r dU[r,c] A[r,k]
    const float* A;       // h_i [B,W] or x [B,din]
    int K, lda;
    float* dW;            // [W,K]
};

template <int RPT, bool HAS_BN, int VEC>
__global__ __launch_bounds__(SN_T) void sn_block_bwd_kernel(SnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char sn_smem[];
    double* red = reinterpret_cast<double*>(sn_smem);
    float* part = reinterpret_cast<float*>(sn_smem + SN_OFF_PART);
    float* dul = reinterpret_cast<float*>(sn_smem + SN_OFF_DUL);
    float* tile = dul + (size_t)p.B * 4;
    const int t = threadIdx.x, c0 = blockIdx.x * SN_NC;
    const int W = p.W, S = p.S, B = p.B, nq = W >> 2;
    const int trow = t & (SN_R - 1);
    const bool owner = t < SN_R;
    int k_lo, k_hi;
    sn_k_range(nq, &k_lo, &k_hi);

    // ---- 1. gradient w.r.t. this block's output, own columns -------------------------------------------------
    f32x4 g[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r0 = j * SN_R;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (r0 < B && p.dUprev) {  // uniform
            sn_stage(tile, p.dUprev + (size_t)r0 * W, min(SN_R, B - r0), W, S, p.step_row, p.step_col);
            __syncthreads();
            const float* urow = tile + trow * S;
            const float* wcol = p.Wprev + c0;
#pragma unroll 2
            for (int k4 = k_lo; k4 < k_hi; ++k4) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(urow + 4 * k4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wcol + (size_t)(4 * k4 + e) * W);
                    a[0] = fmaf(u[e], w[0], a[0]);
                    a[1] = fmaf(u[e], w[1], a[1]);
                    a[2] = fmaf(u[e], w[2], a[2]);
                    a[3] = fmaf(u[e], w[3], a[3]);
                }
            }
            a = sn_combine_parts(a, part);
        }
        const int r = r0 + t;
        if (owner && r < B) {
            const size_t off = (size_t)r * W + c0;
            for (int o = 0; o < p.O; ++o) {
                const float da = p.dact[(size_t)r * p.O + o];
                const f32x4 wo = *reinterpret_cast<const f32x4*>(p.w_out + (size_t)o * W + c0);
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fmaf(da, wo[c], a[c]);
            }
            if (p.keep) {
                const f32x4 old = *reinterpret_cast<const f32x4*>(p.G + off);
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] += old[c];
            }
            *reinterpret_cast<f32x4*>(p.G + off) = a;
        }
        g[j] = a;
    }

    // ---- 2. BatchNorm + ReLU backward on own columns -> dU (global + LDS), dgamma, dbeta, db -------------------
    if (HAS_BN) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c0);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(p.rstd + c0);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c0);
        double s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.0;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = j * SN_R + t;
            if (owner && r < B) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(p.Z + (size_t)r * W + c0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xh = (z[c] - mu[c]) * rs[c];
                    s[c] += (double)g[j][c];
                    s[4 + c] += (double)(g[j][c] * xh);
                }
            }
        }
        sn_block_sum<8>(s, red, reinterpret_cast<double*>(tile));
        float mg[4], mgx[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mg[c] = (float)(s[c] / p.count);
            mgx[c] = (float)(s[4 + c] / p.count);
            if (t == 0) {
                p.dbeta[c0 + c] = (float)s[c];
                p.dgamma[c0 + c] = (float)s[4 + c];
            }
        }
        double sb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = j * SN_R + t;
            if (owner && r < B) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(p.Z + (size_t)r * W + c0);  // again: cache hit
                f32x4 du;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xh = (z[c] - mu[c]) * rs[c];
                    du[c] = z[c] > 0.f ? (gm[c] * rs[c]) * (g[j][c] - mg[c] - xh * mgx[c]) : 0.f;
                    sb[c] += (double)du[c];
                }
                *reinterpret_cast<f32x4*>(p.dU + (size_t)r * W + c0) = du;
                *reinterpret_cast<f32x4*>(dul + 4 * r) = du;
            }
        }
        sn_block_sum<4>(sb, red, reinterpret_cast<double*>(tile));  // its barriers also publish dul
        if (t == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) p.dbias[c0 + c] = (float)sb[c];
        }
    } else {
        double sb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = j * SN_R + t;
            if (owner && r < B) {
#pragma unroll
                for (int c = 0; c < 4; ++c) sb[c] += (double)g[j][c];
                *reinterpret_cast<f32x4*>(dul + 4 * r) = g[j];
            }
        }
        sn_block_sum<4>(sb, red, reinterpret_cast<double*>(tile));
        if (t == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) p.dbias[c0 + c] = (float)sb[c];
        }
    }

    // ---- 3. dW[c0+c, k] = sum_r dU[r,c] A[r,k]: lanes across k, 32 row groups, shuffle + LDS combine ----------
    constexpr int KW = 32 * VEC;  // k values per pass
    const int kq = t & 31, rg = t >> 5, wv = t >> 6;
    float* rbuf = tile;  // [16 waves][4][KW]
    for (int kc = 0; kc < p.K; kc += KW) {
        const int k = kc + kq * VEC;
        float acc[4][VEC];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[c][e] = 0.f;
        if (k < p.K) {
            const float* ap = p.A + k;
            for (int rb = rg; rb < B; rb += (SN_T / 32) * SN_DW_U) {
                float av[SN_DW_U][VEC];
#pragma unroll
                for (int u = 0; u < SN_DW_U; ++u) {
                    const int r = rb + (SN_T / 32) * u;
                    if (r < B) {
                        if (VEC == 4) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(ap + (size_t)r * p.lda);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) av[u][e] = v[e];
                        } else {
                            av[u][0] = ap[(size_t)r * p.lda];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) av[u][e] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < SN_DW_U; ++u) {
                    const int r = rb + (SN_T / 32) * u;
                    if (r < B) {
                        const f32x4 d = *reinterpret_cast<const f32x4*>(dul + 4 * r);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) acc[c][e] = fmaf(d[c], av[u][e], acc[c][e]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                acc[c][e] += __shfl_xor(acc[c][e], 32, 64);  // the wave's two row groups
                if ((t & 32) == 0) rbuf[(wv * 4 + c) * KW + kq * VEC + e] = acc[c][e];
            }
        __syncthreads();
        for (int idx = t; idx < 4 * KW; idx += SN_T) {
            const int c = idx / KW, kk = idx - c * KW;
            if (kc + kk < p.K) {
                float sum = 0.f;
#pragma unroll
                for (int gI = 0; gI < SN_T / 64; ++gI) sum += rbuf[(gI * 4 + c) * KW + kk];
                p.dW[(size_t)(c0 + c) * p.K + kc + kk] = sum;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------
static int sn_rpt(int B) {
    int r = 1;
    while (r * SN_R < B) r <<= 1;
    return r;
}

// more than 64 KiB of dynamic LDS has to be granted per kernel (and device) once
template <typename K>
static int sn_allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return DUN_OK;
    static size_t granted[64] = {0};  // one table per kernel instantiation
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (bytes <= granted[dev]) return DUN_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)bytes);
    if (e != hipSuccess) {
        set_error("smallnet: cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
        return DUN_ERR_LAUNCH;
    }
    granted[dev] = bytes;
    return DUN_OK;
}

template <int RPT>
static int sn_launch_fwd(const SnFwdP& p, hipStream_t s) {
    const size_t lds = sn_fwd_lds(p.W);
    int rc = sn_allow_lds(sn_block_fwd_kernel<RPT>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(sn_block_fwd_kernel<RPT>, dim3(p.W / SN_NC), dim3(SN_T), lds, s, p);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

template <int RPT, bool HAS_BN, int VEC>
static int sn_launch_bwd(const SnBwdP& p, hipStream_t s) {
    const size_t lds = sn_bwd_lds(p.B, p.W);
    int rc = sn_allow_lds(sn_block_bwd_kernel<RPT, HAS_BN, VEC>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((sn_block_bwd_kernel<RPT, HAS_BN, VEC>), dim3(p.W / SN_NC), dim3(SN_T), lds, s, p);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

template <bool HAS_BN, int VEC>
static int sn_dispatch_bwd(const SnBwdP& p, hipStream_t s) {
    switch (sn_rpt(p.B)) {
        case 1: return sn_launch_bwd<1, HAS_BN, VEC>(p, s);
        case 2: return sn_launch_bwd<2, HAS_BN, VEC>(p, s);
        case 4: return sn_launch_bwd<4, HAS_BN, VEC>(p, s);
        default: return sn_launch_bwd<8, HAS_BN, VEC>(p, s);
    }
}

static inline bool sn_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// may this context run on the column-owned kernels?  (shape rule + 16-byte alignment of everything touched in vectors)
bool sn_eligible(const dun_fc_ctx* c, bool backward) {
    if (!sn_shape_ok(c->B, c->W)) return false;
    if (c->depth == 0 && !backward) return false;
    bool ok = sn_al16(c->H) && sn_al16(c->p.w) && sn_al16(c->p.b) && sn_al16(c->p.gamma) && sn_al16(c->p.beta) &&
              sn_al16(c->running_mean) && sn_al16(c->running_var) && (!c->train || sn_al16(c->Z));
    if (c->train) ok = ok && sn_al16(c->bn_mean) && sn_al16(c->bn_rstd);
    if (backward)
        ok = ok && sn_al16(c->G) && sn_al16(c->dU) && sn_al16(c->ws) && sn_al16(c->p.w_out) && sn_al16(c->Z) &&
             sn_al16(c->bn_mean) && sn_al16(c->bn_rstd);
    return ok;
}

static void sn_steps(int W, int* step_row, int* step_col) {
    const int nq = W >> 2;
    *step_row = SN_T / nq;
    *step_col = SN_T % nq;
}

int sn_forward_block(const dun_fc_ctx* c, int i, hipStream_t s) {
    const int W = c->W, B = c->B;
    if (c->train) DUN_REQUIRE(B > 1, "dun_fc_forward: BatchNorm in train mode needs more than 1 row (got %d)", B);
    SnFwdP p;
    memset(&p, 0, sizeof(p));
    p.Hin = c->H + (size_t)i * B * W;
    p.Hout = c->H + (size_t)(i + 1) * B * W;
    p.Wt = c->p.w + (size_t)i * W * W;
    p.bias = c->p.b + (size_t)i * W;
    p.gamma = c->p.gamma + (size_t)i * W;
    p.beta = c->p.beta + (size_t)i * W;
    p.Z = c->train ? c->Z + (size_t)i * B * W : nullptr;
    p.mean = c->train ? c->bn_mean + (size_t)i * W : nullptr;
    p.rstd = c->train ? c->bn_rstd + (size_t)i * W : nullptr;
    p.running_mean = c->running_mean + (size_t)i * W;
    p.running_var = c->running_var + (size_t)i * W;
    p.B = B; p.W = W; p.S = sn_stride(W); p.train = c->train; p.residual = c->residual;
    sn_steps(W, &p.step_row, &p.step_col);
    p.eps = c->bn_eps; p.momentum = c->bn_momentum; p.count = (double)B;
    switch (sn_rpt(B)) {
        case 1: return sn_launch_fwd<1>(p, s);
        case 2: return sn_launch_fwd<2>(p, s);
        case 4: return sn_launch_fwd<4>(p, s);
        default: return sn_launch_fwd<8>(p, s);
    }
}

// all blocks (deepest first) and the input layer; `dU2` is a second [B,W] buffer (the launches ping-pong: a launch
// reads the whole dU of the deeper block while it writes its own columns of this block's)
int sn_backward_blocks(const dun_fc_ctx* c, float* dU2, hipStream_t s) {
    const int W = c->W, B = c->B, L = c->L;
    SnBwdP p;
    const float* prev = nullptr;
    for (int i = L - 1; i >= -1; --i) {
        memset(&p, 0, sizeof(p));
        p.B = B; p.W = W; p.S = sn_stride(W); p.O = c->O;
        sn_steps(W, &p.step_row, &p.step_col);
        p.G = c->G;
        p.dUprev = prev;
        p.Wprev = prev ? c->p.w + (size_t)(i + 1) * W * W : nullptr;
        p.keep = (c->residual && prev) ? 1 : 0;
        p.dact = c->dact + (size_t)(i + 1) * B * c->O;
        p.w_out = c->p.w_out;
        p.count = (double)B;
        int rc;
        if (i >= 0) {
            float* cur = ((L - 1 - i) & 1) ? dU2 : c->dU;
            p.Z = c->Z + (size_t)i * B * W;
            p.mean = c->bn_mean + (size_t)i * W;
            p.rstd = c->bn_rstd + (size_t)i * W;
            p.gamma = c->p.gamma + (size_t)i * W;
            p.dgamma = c->g.gamma + (size_t)i * W;
            p.dbeta = c->g.beta + (size_t)i * W;
            p.dbias = c->g.b + (size_t)i * W;
            p.dU = cur;
            p.A = c->H + (size_t)i * B * W; p.K = W; p.lda = W;
            p.dW = c->g.w + (size_t)i * W * W;
            rc = sn_dispatch_bwd<true, 4>(p, s);
            prev = cur;
        } else {
            p.dbias = c->g.b_in;
            p.A = c->x; p.K = c->din; p.lda = c->din;
            p.dW = c->g.w_in;
            if ((c->din & 3) == 0 && sn_al16(c->x) && sn_al16(c->g.w_in))
                rc = sn_dispatch_bwd<false, 4>(p, s);
            else
                rc = sn_dispatch_bwd<false, 1>(p, s);
        }
        if (rc) return rc;
    }
    return DUN_OK;
}

}  // namespace dun
