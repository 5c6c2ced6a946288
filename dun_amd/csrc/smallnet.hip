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
// The [rows, W] x [W, 4] contraction runs on the vector ALUs with K split over the 16 lanes of a DPP row: a lane
// keeps the weights of its (up to two) 16-byte K steps in registers for the whole launch, loads the matching 16 bytes
// of four rows per wave instruction straight from global memory (a row's lanes read consecutive addresses), and the
// four column sums are combined across the 16 lanes with DPP adds: no LDS staging, no barrier, no uniform-operand
// loads inside the loop.  MFMA tiles would leave W/16 = 7 workgroups on a 256-CU chip, and f32 MFMA runs at the
// vector rate anyway; this regime is latency bound, not a roofline claim.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace dun {

// tunables (-D overrides are for A/B builds; measured on the UCI shape, forward / backward of 10 blocks in us:
// 512 threads, 16 groups in flight 101 / 174; 512, 8: 97.5 / 171 (default); 1024, 8: 105 / 163; 256, 16: 116 / 210;
// prefetching the dW operand rows above the BatchNorm phase: +5 us backward)
#ifndef SN_T_VALUE
#define SN_T_VALUE 512
#endif
#ifndef SN_U_VALUE
#define SN_U_VALUE 8
#endif
#ifndef SN_DW_U_VALUE
#define SN_DW_U_VALUE 16
#endif
#ifndef SN_PREFETCH
#define SN_PREFETCH 0
#endif
constexpr int SN_T = SN_T_VALUE;  // threads per workgroup: 8 waves, two per SIMD
constexpr int SN_WAVES = SN_T / 64;
constexpr int SN_NC = 4;          // columns owned by a workgroup
constexpr int SN_MAX_B = 2048;
constexpr int SN_MAX_W = 128;     // 16 lanes x 2 K steps x 4 floats
constexpr int SN_U = SN_U_VALUE;    // row groups (4 rows per wave each) in flight per batch of loads
constexpr int SN_DW_U = SN_DW_U_VALUE;  // rows in flight per thread in the dW phase

// dynamic LDS carve (every offset a multiple of 16 bytes):
//   [red: 8 doubles][xbuf: 8 x 512 doubles, block sums][rbuf: 8 waves x 4 x 128 floats, dW combine][obuf: B x 4 floats]
constexpr size_t SN_OFF_XBUF = 256;
constexpr size_t SN_OFF_RBUF = SN_OFF_XBUF + (size_t)8 * SN_T * 8;
constexpr size_t SN_OFF_OBUF = SN_OFF_RBUF + (size_t)SN_WAVES * 4 * 128 * 4;
static inline size_t sn_lds(int B) { return SN_OFF_OBUF + (size_t)B * 16; }

int small_net_mode();  // persist.hip: 0 tiled path, 1 these kernels, 2 the one-launch kernels (dun_fc_small_net_mode)

// shape rule only (it sizes the workspace, which must not depend on the mode a later call runs in)
bool sn_shape_ok(int B, int W) {
    return W >= 4 && W <= SN_MAX_W && (W & 3) == 0 && B >= 1 && B <= SN_MAX_B;
}
size_t sn_extra_ws_floats(int B, int W, int L) { return sn_shape_ok(B, W) ? (size_t)L * B * W : 0; }

// ---- sums of four values over the 16 lanes of a DPP row (every lane gets the totals; fixed order) -------------
// v_add_f32 with a DPP-permuted first operand, written as one asm block: the compiler's builtins give a
// v_mov_b32_dpp + v_add_f32 pair per step (twice the instructions).  Stage-major order keeps three instructions
// between a value's consecutive steps, which covers the two wait states a DPP read needs after a VALU write; the
// leading s_nop covers the FMA that produced the inputs (the hazard recogniser does not look inside asm).
__device__ __forceinline__ void sn_row16_sum4(f32x4& a) {
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3;
}

// sum of one value set per thread over the workgroup; every thread gets the totals.  The values go through LDS
// transposed ([NV][512] doubles) so that only NV waves shuffle, one double each (waves butterflying NV doubles
// apiece keep the LDS crossbar busy for microseconds).
template <int NV, bool TRAILING_BARRIER = false>  // the trailing barrier is only needed if red / xbuf are written again
__device__ __forceinline__ void sn_block_sum(double (&v)[NV], double* red, double* xbuf) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) xbuf[i * SN_T + t] = v[i];
    __syncthreads();
    if (t < NV * 64) {
        const int i = t >> 6, l = t & 63;
        const double* x = xbuf + i * SN_T + l;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < SN_T / 64; q += 2) a += x[q * 64] + x[q * 64 + 64];
        a = wave_sum(a);
        if (l == 0) red[i] = a;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i];
    if (TRAILING_BARRIER) __syncthreads();
}

// obuf[r][c] = sum_k A[r,k] wk[k][c] for all rows r < B, c < 4: lane (r4 = lane>>4, kp = lane&15) of wave w covers
// K steps kp and kp+16 of rows (g*8 + w)*4 + r4.  wr[s][e][c]: weight of k = 4*(kp+16 s)+e, column c (zero beyond W).
__device__ __forceinline__ void sn_rowdot(const float* __restrict__ A, int B, int W, const float (&wr)[2][4][4],
                                          float* obuf) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r4 = lane >> 4, kp = lane & 15;
    const int nq = W >> 2;
    const bool has0 = kp < nq, has1 = kp + 16 < nq;
    const int ngroups = (B + 4 * SN_WAVES - 1) / (4 * SN_WAVES);
    const float* a0 = A + (has0 ? 4 * kp : 0);
    const int off1 = has1 ? 64 : 0;
    for (int g0 = 0; g0 < ngroups; g0 += SN_U) {
        // branch-free loads (a branch per load serialises them behind vmcnt(0) waits): addresses are clamped into
        // the matrix, K steps past W are zeroed afterwards (their weights are zero, but 0 * garbage may be NaN);
        // rows past the batch compute garbage that is never stored
        f32x4 h0[SN_U], h1[SN_U];
#pragma unroll
        for (int u = 0; u < SN_U; ++u) {
            const int row = min(((g0 + u) * SN_WAVES + wv) * 4 + r4, B - 1);
            const float* ar = a0 + (size_t)row * W;
            h0[u] = *reinterpret_cast<const f32x4*>(ar);
            h1[u] = *reinterpret_cast<const f32x4*>(ar + off1);
        }
        if (!has1) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < SN_U; ++u) h1[u] = zero;
        }
        if (!has0) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < SN_U; ++u) h0[u] = zero;
        }
#pragma unroll
        for (int u = 0; u < SN_U; ++u) {
            const int row = ((g0 + u) * SN_WAVES + wv) * 4 + r4;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fmaf(h0[u][e], wr[0][e][c], a[c]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fmaf(h1[u][e], wr[1][e][c], a[c]);
            sn_row16_sum4(a);
            if (kp == 0 && row < B) *reinterpret_cast<f32x4*>(obuf + 4 * row) = a;
        }
    }
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
    long long* nbt;     // this block's nn.BatchNorm1d.num_batches_tracked, or NULL
    int B, W, train, residual;
    float eps, momentum;
    double count, inv_count, unbias;  // rows, 1/rows, rows/(rows-1)
#ifdef SN_PROFILE
    long long* dbg;  // scripts/dev/sn_phase_probe.hip: [workgroup][32] clock stamps (10 ns units)
#endif
};

#ifdef SN_PROFILE
#define SN_STAMP(i) do { if (threadIdx.x == 0) p.dbg[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define SN_STAMP(i) do { } while (0)
#endif

template <int RPT>  // rows per thread in the elementwise phases: B <= 512 * RPT
__global__ __launch_bounds__(SN_T) void sn_block_fwd_kernel(SnFwdP p) {
    extern __shared__ __attribute__((aligned(16))) char sn_smem[];
    double* red = reinterpret_cast<double*>(sn_smem);
    double* xbuf = reinterpret_cast<double*>(sn_smem + SN_OFF_XBUF);
    float* obuf = reinterpret_cast<float*>(sn_smem + SN_OFF_OBUF);
    const int t = threadIdx.x, c0 = blockIdx.x * SN_NC;
    const int W = p.W, B = p.B, nq = W >> 2;
    SN_STAMP(0);
    {
        const int kp = t & 15;
        float wr[2][4][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k4 = kp + 16 * s;
            const int k4c = min(k4, nq - 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.Wt + (size_t)(c0 + c) * W + 4 * k4c);
#pragma unroll
                for (int e = 0; e < 4; ++e) wr[s][e][c] = k4 < nq ? v[e] : 0.f;
            }
        }
        sn_rowdot(p.Hin, B, W, wr, obuf);
    }
    SN_STAMP(1);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c0);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c0);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c0);
    // running statistics are read BEFORE anything of this launch is stored: no load waits behind a store later
    const f32x4 rm = *reinterpret_cast<const f32x4*>(p.running_mean + c0);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.running_var + c0);
    f32x4 hin[RPT];
    if (p.residual) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = j * SN_T + t;
            if (r < B) hin[j] = *reinterpret_cast<const f32x4*>(p.Hin + (size_t)r * W + c0);
        }
    }
    __syncthreads();
    SN_STAMP(2);
    f32x4 z[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = j * SN_T + t;
        if (r < B) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(obuf + 4 * r);
#pragma unroll
            for (int c = 0; c < 4; ++c) z[j][c] = fmaxf(a[c] + bv[c], 0.f);
        }
    }
    float mu[4], sc[4];
    if (p.train) {
        double s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.0;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            if (j * SN_T + t < B) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double zd = (double)z[j][c];
                    s[c] += zd;
                    s[4 + c] += zd * zd;
                }
            }
        }
        sn_block_sum<8>(s, red, xbuf);
        f32x4 o_mean, o_rstd, o_rm, o_rv;
        const double mom = (double)p.momentum;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double m = s[c] * p.inv_count;
            double var = s[4 + c] * p.inv_count - m * m;
            var = var > 0.0 ? var : 0.0;
            const double rs = 1.0 / sqrt(var + (double)p.eps);
            const double unbiased = var * p.unbias;
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
            if (p.nbt && blockIdx.x == 0) p.nbt[0] += 1;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mu[c] = rm[c];
            sc[c] = gm[c] * (1.f / sqrtf(rv[c] + p.eps));
        }
    }
    SN_STAMP(3);
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = j * SN_T + t;
        if (r < B) {
            const size_t off = (size_t)r * W + c0;
            if (p.Z) *reinterpret_cast<f32x4*>(p.Z + off) = z[j];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (z[j][c] - mu[c]) * sc[c] + bt[c];
            if (p.residual) {
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = hin[j][c] + o[c];
            }
            *reinterpret_cast<f32x4*>(p.Hout + off) = o;
        }
    }
    SN_STAMP(4);
}

// dW[c0+c, k] = sum_r dU[r,c] A[r,k] for the workgroup's four columns c and all k < K, plus db[c0+c] = sum_r dU[r,c]
// as a virtual column k = K of ones.  dU columns come from LDS (obuf [B][4]); lanes run across k, the 16 row groups
// of the workgroup are combined by one shuffle and a fixed-order LDS sum.
template <int VEC>
__device__ __forceinline__ void sn_dw_phase(const float* obuf, float* rbuf, const float* __restrict__ A, int lda, int K,
                                            int B, float* __restrict__ dW, float* __restrict__ dbias, int c0) {
    constexpr int KW = 32 * VEC;  // k values per pass
    constexpr int NRG = SN_T / 32;
    const int t = threadIdx.x;
    const int kq = t & 31, rg = t >> 5, wv = t >> 6;
    float av[SN_DW_U][VEC];
    for (int kc = 0; kc <= K; kc += KW) {
        const int k = kc + kq * VEC;
        float acc[4][VEC];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[c][e] = 0.f;
        for (int rb = rg; rb < B; rb += NRG * SN_DW_U) {
            if (k < K) {  // branch-free inside: rows past the batch re-read the last row (their dU weight is 0)
                const float* ap = A + k;
#pragma unroll
                for (int u = 0; u < SN_DW_U; ++u) {
                    const int r = min(rb + NRG * u, B - 1);
                    if (VEC == 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(ap + (size_t)r * lda);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) av[u][e] = v[e];
                    } else {
                        av[u][0] = ap[(size_t)r * lda];
                    }
                }
            } else {  // k == K: the column of ones that yields the bias gradient (lanes beyond it are never stored)
#pragma unroll
                for (int u = 0; u < SN_DW_U; ++u)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) av[u][e] = e == 0 ? 1.f : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SN_DW_U; ++u) {
                const int r = rb + NRG * u;
                f32x4 d = *reinterpret_cast<const f32x4*>(obuf + 4 * min(r, B - 1));
                if (r >= B) d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[c][e] = fmaf(d[c], av[u][e], acc[c][e]);
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
            if (kc + kk <= K) {
                float sum = 0.f;
#pragma unroll
                for (int gI = 0; gI < SN_WAVES; ++gI) sum += rbuf[(gI * 4 + c) * KW + kk];
                if (kc + kk < K)
                    dW[(size_t)(c0 + c) * K + kc + kk] = sum;
                else
                    dbias[c0 + c] = sum;
            }
        }
        __syncthreads();
    }
}

struct SnBwdP {
    int B, W, O;
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
    double count, inv_count;
    // weight gradient: dW[c,k] = sum_r dU[r,c] A[r,k]
    const float* A;       // h_i [B,W] or x [B,din]
    int K, lda;
    float* dW;            // [W,K]
#ifdef SN_PROFILE
    long long* dbg;
#endif
};

template <int RPT, bool HAS_BN, int VEC, bool DO_DW>
__global__ __launch_bounds__(SN_T) void sn_block_bwd_kernel(SnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char sn_smem[];
    double* red = reinterpret_cast<double*>(sn_smem);
    double* xbuf = reinterpret_cast<double*>(sn_smem + SN_OFF_XBUF);
    float* rbuf = reinterpret_cast<float*>(sn_smem + SN_OFF_RBUF);
    float* obuf = reinterpret_cast<float*>(sn_smem + SN_OFF_OBUF);  // dU W columns, then this block's dU columns
    const int t = threadIdx.x, c0 = blockIdx.x * SN_NC;
    const int W = p.W, B = p.B, nq = W >> 2;

    // ---- 1. gradient w.r.t. this block's output, own columns -------------------------------------------------
    SN_STAMP(0);
    if (p.dUprev) {  // uniform
        const int kp = t & 15;
        float wr[2][4][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k4 = kp + 16 * s;
            const int k4c = min(k4, nq - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.Wprev + (size_t)(4 * k4c + e) * W + c0);
#pragma unroll
                for (int c = 0; c < 4; ++c) wr[s][e][c] = k4 < nq ? v[c] : 0.f;
            }
        }
        sn_rowdot(p.dUprev, B, W, wr, obuf);
    }
    SN_STAMP(1);
    // operands of the elementwise phases: issued before the barrier
    f32x4 g[RPT], z[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = j * SN_T + t;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (r < B) {
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
            if (HAS_BN) z[j] = *reinterpret_cast<const f32x4*>(p.Z + off);
        }
        g[j] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = j * SN_T + t;
        if (r < B) {
            if (p.dUprev) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(obuf + 4 * r);
#pragma unroll
                for (int c = 0; c < 4; ++c) g[j][c] += d[c];
            }
            *reinterpret_cast<f32x4*>(p.G + (size_t)r * W + c0) = g[j];
        }
    }

    // ---- 2. BatchNorm + ReLU backward on own columns -> dU (global + LDS), dgamma, dbeta ----------------------
    if (HAS_BN) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c0);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(p.rstd + c0);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c0);
        double s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.0;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            if (j * SN_T + t < B) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xh = (z[j][c] - mu[c]) * rs[c];
                    s[c] += (double)g[j][c];
                    s[4 + c] += (double)(g[j][c] * xh);
                }
            }
        }
        sn_block_sum<8>(s, red, xbuf);
        SN_STAMP(3);
        float mg[4], mgx[4];
        f32x4 o_db, o_dg;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mg[c] = (float)(s[c] * p.inv_count);
            mgx[c] = (float)(s[4 + c] * p.inv_count);
            o_db[c] = (float)s[c];
            o_dg[c] = (float)s[4 + c];
        }
        if (t == 0) {
            *reinterpret_cast<f32x4*>(p.dbeta + c0) = o_db;
            *reinterpret_cast<f32x4*>(p.dgamma + c0) = o_dg;
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = j * SN_T + t;
            if (r < B) {
                f32x4 du;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xh = (z[j][c] - mu[c]) * rs[c];
                    du[c] = z[j][c] > 0.f ? (gm[c] * rs[c]) * (g[j][c] - mg[c] - xh * mgx[c]) : 0.f;
                }
                *reinterpret_cast<f32x4*>(p.dU + (size_t)r * W + c0) = du;
                *reinterpret_cast<f32x4*>(obuf + 4 * r) = du;  // same thread read this row's entry above
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = j * SN_T + t;
            if (r < B) *reinterpret_cast<f32x4*>(obuf + 4 * r) = g[j];
        }
    }
    __syncthreads();  // obuf = this block's dU columns (or g for the input layer)
    SN_STAMP(4);

    // ---- 3. weight and bias gradients of the own columns (DO_DW), or left to sn_dw_batched_kernel -------------
    if (DO_DW) sn_dw_phase<VEC>(obuf, rbuf, p.A, p.lda, p.K, B, p.dW, p.dbias, c0);
    SN_STAMP(6);
}

// Weight and bias gradients of ALL blocks in one launch, after the serial chain of block launches: grid (W/4, L),
// workgroup (x, i) computes dW_i[4x..4x+3, :] = dU_i[:, 4x..4x+3]^T h_i and db_i.  Taking this phase out of the
// per-block launches shortens the serial chain by ~3.5 us per block and runs it L-fold parallel instead.
struct SnDwP {
    const float* dU;  // [L,B,W]
    const float* H;   // [L(+1),B,W]
    float* dW;        // [L,W,W]
    float* db;        // [L,W]
    int B, W;
};
__global__ __launch_bounds__(SN_T) void sn_dw_batched_kernel(SnDwP p) {
    extern __shared__ __attribute__((aligned(16))) char sn_smem[];
    float* rbuf = reinterpret_cast<float*>(sn_smem + SN_OFF_RBUF);
    float* obuf = reinterpret_cast<float*>(sn_smem + SN_OFF_OBUF);
    const int c0 = blockIdx.x * SN_NC, i = blockIdx.y;
    const size_t plane = (size_t)p.B * p.W;
    const float* dU = p.dU + (size_t)i * plane + c0;
    for (int r = threadIdx.x; r < p.B; r += SN_T)
        *reinterpret_cast<f32x4*>(obuf + 4 * r) = *reinterpret_cast<const f32x4*>(dU + (size_t)r * p.W);
    __syncthreads();
    sn_dw_phase<4>(obuf, rbuf, p.H + (size_t)i * plane, p.W, p.W, p.B, p.dW + (size_t)i * p.W * p.W, p.db + (size_t)i * p.W, c0);
}

// ---------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------
static int sn_rpt(int B) {
    int r = 1;
    while (r * SN_T < B) r <<= 1;
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
    const size_t lds = sn_lds(p.B);
    int rc = sn_allow_lds(sn_block_fwd_kernel<RPT>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(sn_block_fwd_kernel<RPT>, dim3(p.W / SN_NC), dim3(SN_T), lds, s, p);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

template <int RPT, bool HAS_BN, int VEC, bool DO_DW>
static int sn_launch_bwd(const SnBwdP& p, hipStream_t s) {
    const size_t lds = sn_lds(p.B);
    int rc = sn_allow_lds(sn_block_bwd_kernel<RPT, HAS_BN, VEC, DO_DW>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((sn_block_bwd_kernel<RPT, HAS_BN, VEC, DO_DW>), dim3(p.W / SN_NC), dim3(SN_T), lds, s, p);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

template <bool HAS_BN, int VEC, bool DO_DW>
static int sn_dispatch_bwd(const SnBwdP& p, hipStream_t s) {
    switch (sn_rpt(p.B)) {
        case 1: return sn_launch_bwd<1, HAS_BN, VEC, DO_DW>(p, s);
        case 2: return sn_launch_bwd<2, HAS_BN, VEC, DO_DW>(p, s);
        default: return sn_launch_bwd<4, HAS_BN, VEC, DO_DW>(p, s);
    }
}

static inline bool sn_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// may this context run on the column-owned kernels?  (shape rule + 16-byte alignment of everything touched in vectors)
bool sn_eligible(const dun_fc_ctx* c, bool backward) {
    if (small_net_mode() == 0 || !sn_shape_ok(c->B, c->W)) return false;
    if (c->depth == 0 && !backward) return false;
    bool ok = sn_al16(c->H) && sn_al16(c->p.w) && sn_al16(c->p.b) && sn_al16(c->p.gamma) && sn_al16(c->p.beta) &&
              sn_al16(c->running_mean) && sn_al16(c->running_var) && (!c->train || sn_al16(c->Z));
    if (c->train) ok = ok && sn_al16(c->bn_mean) && sn_al16(c->bn_rstd);
    if (backward)
        ok = ok && sn_al16(c->G) && sn_al16(c->dU) && sn_al16(c->ws) && sn_al16(c->p.w_out) && sn_al16(c->Z) &&
             sn_al16(c->bn_mean) && sn_al16(c->bn_rstd);
    return ok;
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
    p.nbt = (c->train && c->num_batches_tracked) ? reinterpret_cast<long long*>(c->num_batches_tracked) + i : nullptr;
    p.B = B; p.W = W; p.train = c->train; p.residual = c->residual;
    p.eps = c->bn_eps; p.momentum = c->bn_momentum; p.count = (double)B;
    p.inv_count = 1.0 / (double)B; p.unbias = B > 1 ? (double)B / ((double)B - 1.0) : 1.0;
    switch (sn_rpt(B)) {
        case 1: return sn_launch_fwd<1>(p, s);
        case 2: return sn_launch_fwd<2>(p, s);
        default: return sn_launch_fwd<4>(p, s);
    }
}

// all blocks (deepest first), the input layer, then the weight gradients of all blocks in one launch.  `dUall` holds
// the dU of every block ([L,B,W]): a launch reads the whole dU of the deeper block while it writes its own columns of
// this block's, and the batched weight-gradient launch at the end reads them all.
int sn_backward_blocks(const dun_fc_ctx* c, float* dUall, hipStream_t s) {
    const int W = c->W, B = c->B, L = c->L;
    const size_t plane = (size_t)B * W;
    SnBwdP p;
    const float* prev = nullptr;
    for (int i = L - 1; i >= -1; --i) {
        memset(&p, 0, sizeof(p));
        p.B = B; p.W = W; p.O = c->O;
        p.G = c->G;
        p.dUprev = prev;
        p.Wprev = prev ? c->p.w + (size_t)(i + 1) * W * W : nullptr;
        p.keep = (c->residual && prev) ? 1 : 0;
        p.dact = c->dact + (size_t)(i + 1) * B * c->O;
        p.w_out = c->p.w_out;
        p.count = (double)B; p.inv_count = 1.0 / (double)B;
        int rc;
        if (i >= 0) {
            float* cur = dUall + (size_t)i * plane;
            p.Z = c->Z + (size_t)i * plane;
            p.mean = c->bn_mean + (size_t)i * W;
            p.rstd = c->bn_rstd + (size_t)i * W;
            p.gamma = c->p.gamma + (size_t)i * W;
            p.dgamma = c->g.gamma + (size_t)i * W;
            p.dbeta = c->g.beta + (size_t)i * W;
            p.dU = cur;
            rc = sn_dispatch_bwd<true, 4, false>(p, s);
            prev = cur;
        } else {
            p.dbias = c->g.b_in;
            p.A = c->x; p.K = c->din; p.lda = c->din;
            p.dW = c->g.w_in;
            if ((c->din & 3) == 0 && sn_al16(c->x) && sn_al16(c->g.w_in))
                rc = sn_dispatch_bwd<false, 4, true>(p, s);
            else
                rc = sn_dispatch_bwd<false, 1, true>(p, s);
        }
        if (rc) return rc;
    }
    if (L > 0) {
        SnDwP q;
        q.dU = dUall; q.H = c->H; q.dW = c->g.w; q.db = c->g.b; q.B = B; q.W = W;
        const size_t lds = sn_lds(B);
        int rc = sn_allow_lds(sn_dw_batched_kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(sn_dw_batched_kernel, dim3(W / SN_NC, L), dim3(SN_T), lds, s, q);
        DUN_LAUNCH_CHECK();
    }
    return DUN_OK;
}

}  // namespace dun
