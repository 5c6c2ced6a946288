// Residual-MLP trunk of a DUN on device (gfx950): "forward_get_acts" and its hand-derived backward.
//
// Per block i (src/DUN/layers.py:38-42 in the reference):  u = h_i W_i^T + b_i ; z = ReLU(u) ;
// h_{i+1} = [h_i +] gamma_i * (z - mean_i) * rstd_i + beta_i, BatchNorm1d statistics over the batch.
//   forward : MFMA GEMM with bias+ReLU epilogue that also emits per-128-row column sums of z and z^2
//             -> fp64 reduction of those partials -> (all-reduce point for a sharded batch)
//             -> statistics/running-stat update -> one fused normalise+affine+residual pass.
//   backward: column reductions sum_b g, sum_b g*xhat -> (all-reduce point) -> fused BN+ReLU backward
//             producing dU and the bias-gradient partials -> dW = dU^T H_i and g <- [g +] dU W_i (+ the shared
//             head's rank-O term dact_i w_out folded into that epilogue, so no dH tensor is ever written).
// Hidden states of all depths live in one [D,B,W] buffer: the head reads it as one batched contraction.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace dun {

int gemm_f32(const dun_gemm_desc* d, hipStream_t s);
int gemm_colstat_tiles(int M, int N);
int head_forward(const float* H, const float* w, const float* b, int rows, int K, int O, float* act, hipStream_t s);
int head_backward(const float* H, const float* w, const float* dact, int rows, int K, int O, float* dw, float* db,
                  float* dH, float* ws, size_t ws_floats, hipStream_t s);
size_t head_backward_ws(int rows, int K, int O);
// smallnet.hip: one launch per block for widths <= 128 (toy / UCI configurations)
bool sn_shape_ok(int B, int W);
size_t sn_extra_ws_floats(int B, int W, int L);
bool sn_eligible(const dun_fc_ctx* c, bool backward);
int sn_forward_block(const dun_fc_ctx* c, int i, hipStream_t s);
int sn_backward_blocks(const dun_fc_ctx* c, float* dUall, hipStream_t s);
int pn_dw_all(const dun_fc_ctx* c, const float* dUall, hipStream_t s);
// persist.hip: the whole forward / the whole backward of a small net in ONE launch each (row-owning workgroups)
bool pn_eligible(const dun_fc_ctx* c);
int pn_forward(const dun_fc_ctx* c, hipStream_t s);
int pn_backward_trunk(const dun_fc_ctx* c, float* dUall, bool head_from_forward, hipStream_t s);

// rows per column-reduction partial: 128, or 32 for small batches (four times the workgroups, a quarter of the serial
// row loop: the UCI-sized nets are latency bound)
static inline int slab_rows_for(int B) { return B >= 8192 ? 128 : 32; }
constexpr int CX = 64;     // threads across columns in the elementwise / column kernels
constexpr int RY = 4;      // row lanes per workgroup

template <int VW>
struct Vec;
template <>
struct Vec<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, T v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ T load_nt(const float* p) { return ld4_stream<true>(p); }
    static __device__ __forceinline__ void store_nt(float* p, T v) { st4_stream<true>(p, v); }
    static __device__ __forceinline__ float get(const T& v, int e) { return v[e]; }
    static __device__ __forceinline__ void set(T& v, int e, float x) { v[e] = x; }
};
template <>
struct Vec<1> {
    typedef float T;
    static __device__ __forceinline__ T load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
    static __device__ __forceinline__ T load_nt(const float* p) { return *p; }
    static __device__ __forceinline__ void store_nt(float* p, T v) { *p = v; }
    static __device__ __forceinline__ float get(const T& v, int) { return v; }
    static __device__ __forceinline__ void set(T& v, int, float x) { v = x; }
};

// out_d[a][c] = sum_p partials[a][p][c] (fp64, fixed order); optional float copies per array.
// workgroup = 16 columns x 16 partial-lanes: W/16 workgroups (the partials are L2/Infinity-Cache resident, what
// matters is enough workgroups and a short serial loop: P/16 steps)
constexpr int RCX = 16, RCY = 16;
template <typename PT>
__global__ __launch_bounds__(256) void reduce_cols_kernel(const PT* __restrict__ partials, int P, int W,
                                                          double* __restrict__ out_d, float* __restrict__ out_f0,
                                                          float* __restrict__ out_f1) {
    __shared__ double red[RCY][RCX];
    const int tx = threadIdx.x & (RCX - 1), ty = threadIdx.x / RCX;
    const int c = blockIdx.x * RCX + tx, a = blockIdx.y;
    double s = 0.0;
    if (c < W) {
        const PT* src = partials + (size_t)a * P * W + c;
        // (unrolled: the loads of eight steps go out together; the additions keep their order -- as a rolled loop the
        // P/16 steps were a chain of cache round trips, 16 us for the 512 row tiles of the wide trunk)
#pragma unroll 8
        for (int p = ty; p < P; p += RCY) s += (double)src[(size_t)p * W];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < W) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < RCY; ++i) t += red[i][tx];
        if (out_d) out_d[(size_t)a * W + c] = t;
        float* of = a == 0 ? out_f0 : out_f1;
        if (of) of[c] = (float)t;
    }
}

// batch statistics from the (possibly all-reduced) fp64 sums; running stats as nn.BatchNorm1d (unbiased var)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums, int W, double count,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, long long* nbt) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;  // nn.BatchNorm1d.num_batches_tracked of this block
    if (c >= W) return;
    const double m = sums[c] / count;
    double var = sums[W + c] / count - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = var * (count / (count - 1.0));
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
}

// reduce_cols_kernel<double> over the two statistic arrays + bn_finalize_kernel in one launch (single-device forward:
// no all-reduce point between them).  Same summation order as the two-kernel path, so both give the same bits.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const double* __restrict__ partials, int P, int W,
                                                                 double count, float eps, float momentum,
                                                                 double* __restrict__ sums, float* __restrict__ mean,
                                                                 float* __restrict__ rstd,
                                                                 float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, long long* nbt) {
    __shared__ double red[2][RCY][RCX];
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
    const int tx = threadIdx.x & (RCX - 1), ty = threadIdx.x / RCX;
    const int c = blockIdx.x * RCX + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < W) {
        const double* a1 = partials + c;
        const double* a2 = partials + (size_t)P * W + c;
        // (in-step at the wide configuration, same box: rolled 17.1 us, unrolled by 4: 8.1 us, by 8: 22.7 us)
#pragma unroll 4
        for (int p = ty; p < P; p += RCY) {
            s1 += a1[(size_t)p * W];
            s2 += a2[(size_t)p * W];
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < W) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int i = 0; i < RCY; ++i) {
            t1 += red[0][i][tx];
            t2 += red[1][i][tx];
        }
        sums[c] = t1;
        sums[W + c] = t2;
        const double m = t1 / count;
        double var = t2 / count - m * m;
        var = var > 0.0 ? var : 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = var * (count / (count - 1.0));
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// h_out = [h_in +] gamma * (z - mean) * rstd + beta.  stat_is_var: `rstd` holds a variance (eval mode).
template <int VW, bool NT = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ Z, const float* __restrict__ Hin,
                                                       float* __restrict__ Hout, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int stat_is_var, float eps,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int B, int W, int residual,
                                                       int rows_per_block) {
    typedef Vec<VW> V;
    const int tx = threadIdx.x & (CX - 1), ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * CX + tx) * VW;
    if (c >= W) return;
    // (z - mean) first, then scale: folding the mean into a single fma offset (z*sc + (beta - mean*sc)) loses
    // the low bits of every column whose spread is small next to its mean
    float sc[VW], mu[VW], sh[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        const float rs = stat_is_var ? 1.f / sqrtf(rstd[c + e] + eps) : rstd[c + e];
        sc[e] = gamma[c + e] * rs;
        mu[e] = mean[c + e];
        sh[e] = beta[c + e];
    }
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(B, r0 + rows_per_block);
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += RY) {
        const size_t off = (size_t)r * W + c;
        typename V::T z = NT ? V::load_nt(Z + off) : V::load(Z + off), o;
        if (residual) {
            typename V::T hi = NT ? V::load_nt(Hin + off) : V::load(Hin + off);
#pragma unroll
            for (int e = 0; e < VW; ++e) V::set(o, e, V::get(hi, e) + ((V::get(z, e) - mu[e]) * sc[e] + sh[e]));
        } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) V::set(o, e, (V::get(z, e) - mu[e]) * sc[e] + sh[e]);
        }
        if (NT) V::store_nt(Hout + off, o); else V::store(Hout + off, o);
    }
}

// ---- row-owned forms: BatchNorm apply + residual + shared head (+ Gaussian likelihood terms) in one pass -------------
// The column-owned bn_apply_kernel above cannot produce the head value act[row] = h_out[row,:] . w_out + b: a row is
// spread over W/256 workgroups there.  Here a WAVEFRONT owns a row: lane l holds columns (j*64 + l)*4 .. +3 of every
// 256-column chunk j, keeps the per-column BatchNorm scale / mean / shift and its slice of w_out in registers for all
// the rows it walks, and the row-dot is one DPP reduction per row.  With the head value in a register the likelihood
// pass of dun_ll_gauss is free: residual y - act, dact = coef_d * residual, and the per-depth sum of squared residuals
// (fp64 from the wavefront on, one partial per workgroup, fixed order).  So act_vec, dact and the objective's sums
// come out of the pass that writes h_out, and the 27 GB re-read of H by a separate head kernel disappears.
struct HeadObj {
    const float* y;        // [B] or nullptr
    const float* w;        // [D] depth weights / logits or nullptr
    int w_is_logits, D, d; // d: depth of the rows of this launch (grid.y adds to it)
    const float* log_std;
    float inv_batch;
    float* dact;           // [D,B] or nullptr
    double* partials;      // [D,nparts] or nullptr
    int nparts;
    float* dwpart;         // [D,nparts,K] or nullptr: per-workgroup partial sums of dact[row] * h[row,:] (the head's dw)
    float* dbpart;         // [D,nparts]: per-workgroup partial sums of dact[row]
};

__device__ __forceinline__ float head_obj_coef(const HeadObj& o, int d) {
    if (o.y == nullptr) return 0.f;
    const float wd = depth_weight(o.w, o.w_is_logits, o.D, d);
    const float sigma = fmaxf(expf(o.log_std[0]), 1e-4f);
    return -(wd * o.inv_batch) / (sigma * sigma);
}

// one row's head value -> act, dact, running sum of squared residuals (all lanes hold `a`; lane 0 stores)
// -> this row's d loss / d act (0 without an objective)
__device__ __forceinline__ float head_obj_row(const HeadObj& o, int d, int B, int row, float a, float coef, int lane,
                                              float* __restrict__ act_d, double& acc) {
    if (lane == 0) act_d[row] = a;
    float g = 0.f;
    if (o.y) {
        const float r = o.y[row] - a;
        g = coef * r;
        if (lane == 0 && o.dact) o.dact[(size_t)d * B + row] = g;
        acc += (double)(r * r);
    }
    return g;
}

// The head's weight gradient from the forward pass: for the ELBO d loss / d act of a row is known the moment its head
// value is (q is known a priori), and the row of h is in registers -- so dw_out = sum_rows dact[row] * h[row,:] is
// accumulated per lane over the rows a wavefront walks (fp32, <= 32 rows), the four wavefronts of a workgroup are added
// in a fixed order through LDS, and one [K] partial per workgroup and depth is left for a fp64 reduction at the start
// of backward.  This is what removes head_bwd's re-read of all hidden states (27 GB at the wide configuration).
template <int NCH>
__device__ __forceinline__ void head_dw_finish(const HeadObj& o, int d, const f32x4 (&gw)[NCH], float gb, int K,
                                               float* gwred /* [4][NCH*256] */, float* gbred /* [4] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NCH; ++j) *reinterpret_cast<f32x4*>(gwred + wv * (NCH * 256) + (j * 64 + lane) * 4) = gw[j];
    if (lane == 0) gbred[wv] = gb;
    __syncthreads();
    float* dst = o.dwpart + ((size_t)d * o.nparts + blockIdx.x) * K;
    for (int c = threadIdx.x * 4; c < K; c += 1024) {
        f32x4 t = *reinterpret_cast<const f32x4*>(gwred + c);
#pragma unroll
        for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(gwred + w * (NCH * 256) + c);
        *reinterpret_cast<f32x4*>(dst + c) = t;
    }
    if (threadIdx.x == 0) o.dbpart[(size_t)d * o.nparts + blockIdx.x] = (gbred[0] + gbred[1]) + (gbred[2] + gbred[3]);
}

__device__ __forceinline__ void head_obj_finish(const HeadObj& o, int d, double acc, double* red) {
    if (o.partials == nullptr) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = acc;  // (every lane of a wavefront holds the same sum)
    __syncthreads();
    if (threadIdx.x == 0) o.partials[(size_t)d * o.nparts + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int NCH, bool NT, bool DW>
__global__ __launch_bounds__(256) void bn_apply_head_kernel(const float* __restrict__ Z, const float* __restrict__ Hin,
                                                            float* __restrict__ Hout, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int stat_is_var, float eps,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int B, int residual,
                                                            const float* __restrict__ w_out,
                                                            const float* __restrict__ b_out, float* __restrict__ act_d,
                                                            HeadObj o) {
    __shared__ double red[4];
    __shared__ __attribute__((aligned(16))) float gwred[DW ? 4 * NCH * 256 : 4];
    __shared__ float gbred[4];
    constexpr int W = NCH * 256;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    f32x4 sc[NCH], mu[NCH], sh[NCH], wo[NCH], gw[NCH];
    float gb = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) gw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (j * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), st = *reinterpret_cast<const f32x4*>(rstd + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[j][e] = g[e] * (stat_is_var ? 1.f / sqrtf(st[e] + eps) : st[e]);
        mu[j] = *reinterpret_cast<const f32x4*>(mean + c);
        sh[j] = *reinterpret_cast<const f32x4*>(beta + c);
        wo[j] = *reinterpret_cast<const f32x4*>(w_out + c);
    }
    const float coef = head_obj_coef(o, o.d);
    const float b0 = b_out[0];
    double acc = 0.0;
    for (int row = wave; row < B; row += nwaves) {
        const size_t off = (size_t)row * W + lane * 4;
        f32x4 z[NCH], h[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) z[j] = ld4_stream<NT>(Z + off + j * 256);
        if (residual) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) h[j] = ld4_stream<NT>(Hin + off + j * 256);
        }
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 v;
            // (z - mean) first, then scale: see bn_apply_kernel
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (z[j][e] - mu[j][e]) * sc[j][e] + sh[j][e];
            if (residual) v += h[j];
            st4_stream<NT>(Hout + off + j * 256, v);
            dot += (v[0] * wo[j][0] + v[1] * wo[j][1]) + (v[2] * wo[j][2] + v[3] * wo[j][3]);
            if (DW) z[j] = v;  // the row of h_out is needed once more, for the head's weight gradient
        }
        const float g = head_obj_row(o, o.d, B, row, wave_allsum(dot) + b0, coef, lane, act_d, acc);
        if (DW) {
            gb += g;
#pragma unroll
            for (int j = 0; j < NCH; ++j) gw[j] += g * z[j];
        }
    }
    head_obj_finish(o, o.d, acc, red);
    if (DW) head_dw_finish<NCH>(o, o.d, gw, gb, W, gwred, gbred);
}

// The shared head alone (O == 1) with the same likelihood epilogue: depth blockIdx.y of H [.,B,K]; any K.
template <bool VEC, bool NT, bool DW = false>
__global__ __launch_bounds__(256) void head_obj_kernel(const float* __restrict__ H, const float* __restrict__ w_out,
                                                       const float* __restrict__ b_out, int B, int K,
                                                       float* __restrict__ act, HeadObj o) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int dl = blockIdx.y, d = o.d + dl;  // dl: depth inside this launch's H / act, d: global depth (weights, sums)
    const float* Hd = H + (size_t)dl * B * K;
    float* act_d = act + (size_t)dl * B;
    const float coef = head_obj_coef(o, d);
    const float b0 = b_out[0];
    double acc = 0.0;
    // DW (host: VEC, K % 256 == 0, K <= 2048): the row stays in registers for the head's weight gradient
    __shared__ __attribute__((aligned(16))) float gwred[DW ? 4 * 8 * 256 : 4];
    __shared__ float gbred[4];
    f32x4 gw[8], hv8[8];
    float gb = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) gw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = wave; row < B; row += nwaves) {
        const float* h = Hd + (size_t)row * K;
        float dot = 0.f;
        if (DW) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = (j * 64 + lane) * 4;
                hv8[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < K) {
                    hv8[j] = ld4_stream<NT>(h + k);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w_out + k);
                    dot += (hv8[j][0] * wv[0] + hv8[j][1] * wv[1]) + (hv8[j][2] * wv[2] + hv8[j][3] * wv[3]);
                }
            }
        } else if (VEC) {
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4 hv = ld4_stream<NT>(h + k), wv = *reinterpret_cast<const f32x4*>(w_out + k);
                dot += (hv[0] * wv[0] + hv[1] * wv[1]) + (hv[2] * wv[2] + hv[3] * wv[3]);
            }
        } else {
            for (int k = lane; k < K; k += 64) dot += h[k] * w_out[k];
        }
        // (dact / partials are indexed by the global depth; act_d is already offset)
        const float g = head_obj_row(o, d, B, row, wave_allsum(dot) + b0, coef, lane, act_d, acc);
        if (DW) {
            gb += g;
#pragma unroll
            for (int j = 0; j < 8; ++j) gw[j] += g * hv8[j];
        }
    }
    head_obj_finish(o, d, acc, red);
    if (DW) {
        // (the 8-chunk LDS image is K-strided per wavefront as in head_dw_finish<8>; columns >= K hold zeros)
        head_dw_finish<8>(o, d, gw, gb, K, gwred, gbred);
    }
}

// stage 1 of dw_out = sum over all (depth, workgroup) partials: grid (K/256, row splits); a thread owns four columns,
// the four wavefronts split the rows of the split, fp64 from the first add on; out double [splits][K]
__global__ __launch_bounds__(256) void head_dw_reduce_kernel(const float* __restrict__ part, int P, int K, int rows_per_split,
                                                             double* __restrict__ out) {
    __shared__ double red[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;  // this thread's four columns (K % 256 == 0: always inside)
    const int r0 = blockIdx.y * rows_per_split, r1 = min(P, r0 + rows_per_split);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int r = r0 + wv; r < r1; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)r * K + c);
        a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
    red[wv][lane * 4 + 0] = a0; red[wv][lane * 4 + 1] = a1; red[wv][lane * 4 + 2] = a2; red[wv][lane * 4 + 3] = a3;
    __syncthreads();
    const int t = threadIdx.x;  // 256 threads <-> the 256 columns of this block
    out[(size_t)blockIdx.y * K + blockIdx.x * 256 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// db_out = sum of n floats in fp64, one workgroup, fixed order
__global__ __launch_bounds__(256) void sum_to_scalar_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)v[i];
    const double t = block_sum<256, double>(a, red);
    if (threadIdx.x == 0) out[0] = (float)t;
}

// per-slab column sums of g and g*xhat (xhat = (z-mean)*rstd): partials [2][P][W]
template <int VW, bool NT = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ G, const float* __restrict__ Z,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int B, int W, int P,
                                                            int SLAB, float* __restrict__ partials) {
    typedef Vec<VW> V;
    __shared__ float red[2][RY][CX * VW];
    const int tx = threadIdx.x & (CX - 1), ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * CX + tx) * VW;
    const int slab = blockIdx.y;
    float s0[VW], s1[VW], mu[VW], rs[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = 0.f; rs[e] = 0.f; }
    if (c < W) {
#pragma unroll
        for (int e = 0; e < VW; ++e) { mu[e] = mean[c + e]; rs[e] = rstd[c + e]; }
        const int r1 = min(B, (slab + 1) * SLAB);
#pragma unroll 4
        for (int r = slab * SLAB + ty; r < r1; r += RY) {
            const size_t off = (size_t)r * W + c;
            const typename V::T g = NT ? V::load_nt(G + off) : V::load(G + off), z = NT ? V::load_nt(Z + off) : V::load(Z + off);
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float ge = V::get(g, e);
                s0[e] += ge;
                s1[e] += ge * ((V::get(z, e) - mu[e]) * rs[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        red[0][ty][tx * VW + e] = s0[e];
        red[1][ty][tx * VW + e] = s1[e];
    }
    __syncthreads();
    if (ty == 0 && c < W) {
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const int i = tx * VW + e;
            partials[(size_t)slab * W + c + e] = (red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i]);
            partials[(size_t)(P + slab) * W + c + e] = (red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i]);
        }
    }
}

// dU = (z > 0) * gamma * rstd * (g - sum_g/count - xhat * sum_gx/count); per-slab column sums of dU -> partials[P][W]
template <int VW, bool NT = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ Z,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, double count, int B,
                                                           int W, int SLAB, float* __restrict__ dU,
                                                           float* __restrict__ partials) {
    typedef Vec<VW> V;
    __shared__ float red[RY][CX * VW];
    const int tx = threadIdx.x & (CX - 1), ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * CX + tx) * VW;
    const int slab = blockIdx.y;
    float acc[VW], mu[VW], rs[VW], k0[VW], mg[VW], mgx[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { acc[e] = 0.f; mu[e] = rs[e] = k0[e] = mg[e] = mgx[e] = 0.f; }
    if (c < W) {
        // one fp64 division per thread, not 2*VW: a software division is ~100 instructions and eight of them were as
        // long as the thread's whole 32-row loop (exact for power-of-two batches, one fp64 rounding otherwise)
        const double inv_count = 1.0 / count;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            mu[e] = mean[c + e];
            rs[e] = rstd[c + e];
            k0[e] = gamma[c + e] * rs[e];
            mg[e] = (float)(sums[c + e] * inv_count);
            mgx[e] = (float)(sums[W + c + e] * inv_count);
        }
        const int r1 = min(B, (slab + 1) * SLAB);
#pragma unroll 4
        for (int r = slab * SLAB + ty; r < r1; r += RY) {
            const size_t off = (size_t)r * W + c;
            const typename V::T g = NT ? V::load_nt(G + off) : V::load(G + off), z = NT ? V::load_nt(Z + off) : V::load(Z + off);
            typename V::T o;
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float ze = V::get(z, e);
                const float xh = (ze - mu[e]) * rs[e];
                const float du = ze > 0.f ? k0[e] * (V::get(g, e) - mg[e] - xh * mgx[e]) : 0.f;
                V::set(o, e, du);
                acc[e] += du;
            }
            if (NT) V::store_nt(dU + off, o); else V::store(dU + off, o);
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) red[ty][tx * VW + e] = acc[e];
    __syncthreads();
    if (ty == 0 && c < W) {
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const int i = tx * VW + e;
            partials[(size_t)slab * W + c + e] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        }
    }
}

// Row-owned form of bn_bwd_apply for the wide trunk (width = NCH*256 <= 2048, planes beyond the Infinity Cache): a
// wavefront owns a row, like bn_apply_head_kernel, and streams 8 KB-contiguous rows instead of the 1 KB pieces at an
// 8 KB stride that a column-owned workgroup reads.  Per-column constants in registers (k0 = gamma*rstd, mean, mean_g,
// rstd*mean_gx: four, so that with the db accumulators and 16 loads in flight the lane stays under 256 VGPRs -- the
// column-owned kernel keeps rstd and mean_gx apart, one rounding of difference); the bias-gradient column sums are
// accumulated per lane over the wavefront's rows and combined over the four wavefronts through LDS in a fixed order:
// one [W] partial per workgroup, the layout reduce_cols_kernel expects.
template <int NCH, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const float* __restrict__ G, const float* __restrict__ Z,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma,
                                                                const double* __restrict__ sums, double count, int B,
                                                                float* __restrict__ dU, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) float red[4 * NCH * 256];
    constexpr int W = NCH * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    f32x4 k0[NCH], mu[NCH], mg[NCH], cx[NCH], acc[NCH];
    const double inv_count = 1.0 / count;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (j * 64 + lane) * 4;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
        mu[j] = *reinterpret_cast<const f32x4*>(mean + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            k0[j][e] = g4[e] * rs[e];
            mg[j][e] = (float)(sums[c + e] * inv_count);
            cx[j][e] = rs[e] * (float)(sums[W + c + e] * inv_count);
            acc[j][e] = 0.f;
        }
    }
    for (int row = wave; row < B; row += nwaves) {
        const size_t off = (size_t)row * W + lane * 4;
        f32x4 g[NCH], z[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) g[j] = ld4_stream<NT>(G + off + j * 256);
#pragma unroll
        for (int j = 0; j < NCH; ++j) z[j] = ld4_stream<NT>(Z + off + j * 256);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 du;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                du[e] = z[j][e] > 0.f ? k0[j][e] * ((g[j][e] - mg[j][e]) - (z[j][e] - mu[j][e]) * cx[j][e]) : 0.f;
            st4_stream<NT>(dU + off + j * 256, du);
            acc[j] += du;
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) *reinterpret_cast<f32x4*>(red + wv * W + (j * 64 + lane) * 4) = acc[j];
    __syncthreads();
    float* dst = partials + (size_t)blockIdx.x * W;
    for (int c = threadIdx.x * 4; c < W; c += 1024) {
        f32x4 t = *reinterpret_cast<const f32x4*>(red + c);
#pragma unroll
        for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + w * W + c);
        *reinterpret_cast<f32x4*>(dst + c) = t;
    }
}

// per-slab column sums of X [B,W] -> partials [P][W]
template <int VW, bool NT = false>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int B, int W, int SLAB,
                                                     float* __restrict__ partials) {
    typedef Vec<VW> V;
    __shared__ float red[RY][CX * VW];
    const int tx = threadIdx.x & (CX - 1), ty = threadIdx.x >> 6;
    const int c = (blockIdx.x * CX + tx) * VW;
    const int slab = blockIdx.y;
    float acc[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) acc[e] = 0.f;
    if (c < W) {
        const int r1 = min(B, (slab + 1) * SLAB);
#pragma unroll 4
        for (int r = slab * SLAB + ty; r < r1; r += RY) {
            const typename V::T x = NT ? V::load_nt(X + (size_t)r * W + c) : V::load(X + (size_t)r * W + c);
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[e] += V::get(x, e);
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) red[ty][tx * VW + e] = acc[e];
    __syncthreads();
    if (ty == 0 && c < W) {
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const int i = tx * VW + e;
            partials[(size_t)slab * W + c + e] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        }
    }
}

// G[m,n] = sum_o dact[m,o] * w_out[o,n]: the deepest exit's contribution starts the running gradient
__global__ __launch_bounds__(256) void rank_init_kernel(float* __restrict__ G, const float* __restrict__ dact,
                                                        const float* __restrict__ w_out, int B, int W, int O) {
    const size_t n = (size_t)B * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / W), c = (int)(i % W);
        float v = 0.f;
        for (int o = 0; o < O; ++o) v += dact[(size_t)m * O + o] * w_out[(size_t)o * W + c];
        G[i] = v;
    }
}

// out[l][c][r] = in[l][r][c] for L square W x W matrices (W % 64 == 0): 64x64 tiles through LDS (+1 padding)
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int W) {
    __shared__ float tile[64][65];
    const size_t mat = (size_t)blockIdx.z * W * W;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = ty * 16 + j;
        tile[r][tx] = in[mat + (size_t)(r0 + r) * W + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = ty * 16 + j;
        out[mat + (size_t)(c0 + c) * W + r0 + tx] = tile[tx][c];
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, size_t n, float lr, float momentum,
                                                  float wd) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 bv = reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] + wd * pv[e];
            bv[e] = momentum * bv[e] + ge;
            pv[e] -= lr * bv[e];
        }
        reinterpret_cast<f32x4*>(buf)[i] = bv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
    }
    const size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float ge = g[i] + wd * p[i];
        const float b = momentum * buf[i] + ge;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

// the same update, element by element: tensors that are not 16-byte aligned (the q_logits / log_std gradients are
// views into the objective's result block)
__global__ __launch_bounds__(256) void sgd_scalar_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ buf, size_t n, float lr, float momentum,
                                                         float wd) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float ge = g[i] + wd * p[i];
        const float b = momentum * buf[i] + ge;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------------
static inline int num_slabs(int B) { return ceil_div(B, slab_rows_for(B)); }
static inline bool vec4_ok(int W, const void* a, const void* b = nullptr, const void* c = nullptr) {
    return W % 4 == 0 && aligned16(a) && (!b || aligned16(b)) && (!c || aligned16(c));
}

static int check_ctx(const dun_fc_ctx* c, const char* who) {
    DUN_REQUIRE(c != nullptr, "%s: null context", who);
    DUN_REQUIRE(c->B > 0 && c->din > 0 && c->W > 0 && c->O > 0 && c->L >= 0, "%s: bad sizes B=%d din=%d W=%d O=%d L=%d",
                who, c->B, c->din, c->W, c->O, c->L);
    DUN_REQUIRE(c->depth >= 0 && c->depth <= c->L, "%s: depth=%d outside [0,%d]", who, c->depth, c->L);
    DUN_REQUIRE(c->p.w_in && c->p.b_in && c->p.w_out && c->p.b_out, "%s: null input/output layer parameters", who);
    DUN_REQUIRE(c->L == 0 || (c->p.w && c->p.b && c->p.gamma && c->p.beta && c->running_mean && c->running_var),
                "%s: null block parameters", who);
    DUN_REQUIRE(c->x && c->H && c->act, "%s: null x/H/act", who);
    DUN_REQUIRE(c->depth == 0 || c->Z, "%s: null Z", who);
    DUN_REQUIRE(c->ws && c->ws_floats >= dun_fc_workspace_floats(c->B, c->din, c->W, c->O, c->L),
                "%s: workspace too small", who);
    DUN_REQUIRE(aligned16(c->ws), "%s: workspace must be 16-byte aligned", who);
    DUN_REQUIRE(!c->train || c->depth == 0 || (c->bn_mean && c->bn_rstd && c->bn_sums), "%s: train mode needs bn_mean/bn_rstd/bn_sums", who);
    if (c->obj_y) {
        DUN_REQUIRE(c->O == 1, "%s: the likelihood pass inside the head needs one output dimension (O=%d)", who, c->O);
        DUN_REQUIRE(c->obj_log_std && c->obj_partials, "%s: obj_y without obj_log_std / obj_partials", who);
        DUN_REQUIRE(!c->obj_w || c->obj_dact, "%s: depth weights without obj_dact", who);
    }
    return DUN_OK;
}

static inline int obj_nparts(int B) {
    // up to 2048 rows: one partial per 16 rows -- the row-owning workgroups of the one-launch forward (persist.hip) write
    // the same [depth][nparts] layout; beyond: four wavefronts per workgroup, >= 8 rows per wavefront up to 512 workgroups
    if (B <= 2048) return B < 1 ? 1 : ceil_div(B, 16);
    const int n = ceil_div(B, 32);
    return n > 512 ? 512 : n;
}

static bool head_fused(const dun_fc_ctx* c);

static HeadObj make_obj(const dun_fc_ctx* c, int d) {
    HeadObj o;
    memset(&o, 0, sizeof(o));
    o.D = c->depth + 1;
    o.d = d;
    o.nparts = obj_nparts(c->B);
    if (c->obj_y && c->O == 1) {
        o.y = c->obj_y;
        o.w = c->obj_w;
        o.w_is_logits = c->obj_w_is_logits;
        o.log_std = c->obj_log_std;
        o.inv_batch = c->obj_inv_batch;
        o.dact = c->obj_w ? c->obj_dact : nullptr;
        o.partials = c->obj_partials;
        if (o.dact && c->obj_dwpart && c->obj_dbpart && head_fused(c)) {
            o.dwpart = c->obj_dwpart;
            o.dbpart = c->obj_dbpart;
        }
    }
    return o;
}

// block_norm(i) also yields act[i+1] (and the likelihood terms): row-owned pass, see bn_apply_head_kernel
static bool head_fused(const dun_fc_ctx* c) {
    if (c->O != 1 || c->W % 256 != 0 || c->W > 2048 || c->depth == 0) return false;
    const bool stats_ok = c->train ? (aligned16(c->bn_mean) && aligned16(c->bn_rstd))
                                   : (aligned16(c->running_mean) && aligned16(c->running_var));
    return stats_ok && aligned16(c->H) && aligned16(c->Z) && aligned16(c->p.w_out) && aligned16(c->p.gamma) &&
           aligned16(c->p.beta);
}

// shared head (O == 1) over `ndepth` depths starting at depth d0, with the likelihood epilogue when the context asks
static int head_obj(const dun_fc_ctx* c, int d0, int ndepth, hipStream_t s) {
    const float* H = c->H + (size_t)d0 * c->B * c->W;
    float* act = c->act + (size_t)d0 * c->B;
    const HeadObj o = make_obj(c, d0);
    const dim3 grid(obj_nparts(c->B), ndepth);
    const bool vec = c->W % 4 == 0 && aligned16(H) && aligned16(c->p.w_out);
    const bool nt = stream_nt((size_t)ndepth * c->B * c->W * 4);
    if (o.dwpart != nullptr && vec && nt)  // (dwpart is only set where head_fused: K % 256 == 0, K <= 2048, aligned)
        hipLaunchKernelGGL((head_obj_kernel<true, true, true>), grid, dim3(256), 0, s, H, c->p.w_out, c->p.b_out, c->B, c->W, act, o);
    else if (o.dwpart != nullptr && vec)
        hipLaunchKernelGGL((head_obj_kernel<true, false, true>), grid, dim3(256), 0, s, H, c->p.w_out, c->p.b_out, c->B, c->W, act, o);
    else if (vec && nt)
        hipLaunchKernelGGL((head_obj_kernel<true, true>), grid, dim3(256), 0, s, H, c->p.w_out, c->p.b_out, c->B, c->W, act, o);
    else if (vec)
        hipLaunchKernelGGL((head_obj_kernel<true, false>), grid, dim3(256), 0, s, H, c->p.w_out, c->p.b_out, c->B, c->W, act, o);
    else
        hipLaunchKernelGGL((head_obj_kernel<false, false>), grid, dim3(256), 0, s, H, c->p.w_out, c->p.b_out, c->B, c->W, act, o);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

static int check_block(const dun_fc_ctx* c, int i, const char* who) {
    int rc = check_ctx(c, who);
    if (rc) return rc;
    DUN_REQUIRE(i >= 0 && i < c->depth, "%s: block %d outside [0,%d)", who, i, c->depth);
    return DUN_OK;
}

static inline float* Hd(const dun_fc_ctx* c, int d) { return c->H + (size_t)d * c->B * c->W; }
static inline float* Zd(const dun_fc_ctx* c, int i) { return c->train ? c->Z + (size_t)i * c->B * c->W : c->Z; }

static int input_layer(const dun_fc_ctx* c, hipStream_t s) {
    dun_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = c->x; g.lda = c->din; g.a_kcontig = 1;
    g.B = c->p.w_in; g.ldb = c->din; g.b_kcontig = 1;
    g.C = Hd(c, 0); g.ldc = c->W;
    g.M = c->B; g.N = c->W; g.K = c->din;
    g.bias = c->p.b_in;
    return gemm_f32(&g, s);
}

static int block_linear(const dun_fc_ctx* c, int i, hipStream_t s, bool reduce_stats = true) {
    const int W = c->W, P = gemm_colstat_tiles(c->B, W);  // the GEMM's own row tiling (128, or 64 on small problems)
    dun_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = Hd(c, i); g.lda = W; g.a_kcontig = 1;
    g.B = c->p.w + (size_t)i * W * W; g.ldb = W; g.b_kcontig = 1;
    g.C = Zd(c, i); g.ldc = W;
    g.M = c->B; g.N = W; g.K = W;
    g.bias = c->p.b + (size_t)i * W;
    g.relu = 1;
    double* stats = reinterpret_cast<double*>(c->ws);  // ws is 16-byte aligned (checked in check_ctx)
    g.colstats = c->train ? stats : nullptr;
    int rc = gemm_f32(&g, s);
    if (rc) return rc;
    if (c->train && reduce_stats) {
        hipLaunchKernelGGL(reduce_cols_kernel<double>, dim3(ceil_div(W, RCX), 2), dim3(256), 0, s, stats, P, W,
                           c->bn_sums, (float*)nullptr, (float*)nullptr);
        DUN_LAUNCH_CHECK();
    }
    return DUN_OK;
}

// from_partials: the GEMM's per-tile statistics are still unreduced in the workspace (single-device forward)
// apply_only: the statistics of block i are already in bn_mean / bn_rstd (dun_fc_block_apply)
static int block_norm(const dun_fc_ctx* c, int i, double global_batch, hipStream_t s, bool from_partials = false,
                      bool apply_only = false) {
    const int W = c->W, B = c->B;
    const float* mean;
    const float* stat;
    int is_var;
    if (c->train && apply_only) {
        mean = c->bn_mean + (size_t)i * W; stat = c->bn_rstd + (size_t)i * W; is_var = 0;
    } else if (c->train) {
        DUN_REQUIRE(global_batch > 1.0, "dun_fc_block_norm: BatchNorm in train mode needs more than 1 row (got %g)",
                    global_batch);
        float* m = c->bn_mean + (size_t)i * W;
        float* r = c->bn_rstd + (size_t)i * W;
        long long* nbt = c->num_batches_tracked ? reinterpret_cast<long long*>(c->num_batches_tracked) + i : nullptr;
        if (from_partials)
            hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(ceil_div(W, RCX)), dim3(256), 0, s,
                               reinterpret_cast<const double*>(c->ws), gemm_colstat_tiles(B, W), W, global_batch,
                               c->bn_eps, c->bn_momentum, c->bn_sums, m, r, c->running_mean + (size_t)i * W,
                               c->running_var + (size_t)i * W, nbt);
        else
            hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(W, 256)), dim3(256), 0, s, c->bn_sums, W, global_batch,
                               c->bn_eps, c->bn_momentum, m, r, c->running_mean + (size_t)i * W,
                               c->running_var + (size_t)i * W, nbt);
        DUN_LAUNCH_CHECK();
        mean = m; stat = r; is_var = 0;
    } else {
        mean = c->running_mean + (size_t)i * W;
        stat = c->running_var + (size_t)i * W;
        is_var = 1;
    }
    const float* Z = Zd(c, i);
    const float* hin = Hd(c, i);
    float* hout = Hd(c, i + 1);
    const float* gamma = c->p.gamma + (size_t)i * W;
    const float* beta = c->p.beta + (size_t)i * W;
    const int rpb = 32;
    const int nt = stream_nt((size_t)B * W * 4) ? 1 : 0;  // [B,W] planes beyond the Infinity Cache are streamed
    if (head_fused(c)) {
        const HeadObj o = make_obj(c, i + 1);
        float* act_d = c->act + (size_t)(i + 1) * B;
        const dim3 grid(obj_nparts(B));
#define BAH1(N, NTF, DWF)                                                                                              \
    hipLaunchKernelGGL((bn_apply_head_kernel<N, NTF, DWF>), grid, dim3(256), 0, s, Z, hin, hout, mean, stat, is_var,     \
                       c->bn_eps, gamma, beta, B, c->residual, c->p.w_out, c->p.b_out, act_d, o)
#define BAH(N)                                                                                                         \
    case N:                                                                                                            \
        if (o.dwpart != nullptr) { if (nt) BAH1(N, true, true); else BAH1(N, false, true); }                          \
        else { if (nt) BAH1(N, true, false); else BAH1(N, false, false); }                                            \
        break;
        switch (W / 256) {
            BAH(1) BAH(2) BAH(3) BAH(4) BAH(5) BAH(6) BAH(7) BAH(8)
        }
#undef BAH
#undef BAH1
        DUN_LAUNCH_CHECK();
        return DUN_OK;
    }
    if (vec4_ok(W, Z, hin, hout)) {
        if (nt)
            hipLaunchKernelGGL((bn_apply_kernel<4, true>), dim3(ceil_div(W, CX * 4), ceil_div(B, rpb)), dim3(256), 0, s, Z,
                               hin, hout, mean, stat, is_var, c->bn_eps, gamma, beta, B, W, c->residual, rpb);
        else
            hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(ceil_div(W, CX * 4), ceil_div(B, rpb)), dim3(256), 0, s, Z, hin,
                               hout, mean, stat, is_var, c->bn_eps, gamma, beta, B, W, c->residual, rpb);
    } else {
        hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(ceil_div(W, CX), ceil_div(B, rpb)), dim3(256), 0, s, Z, hin, hout,
                           mean, stat, is_var, c->bn_eps, gamma, beta, B, W, c->residual, rpb);
    }
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// ---- inference trunk: BatchNorm(eval) + residual + head partials inside the block GEMM's epilogue ---------------------
// scale[i][n] = gamma / sqrt(running_var + eps) for every block at once (the expression of bn_apply_kernel)
__global__ __launch_bounds__(256) void bn_eval_scale_kernel(const float* __restrict__ var, const float* __restrict__ gamma,
                                                            float eps, size_t n, float* __restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) scale[i] = gamma[i] * (1.f / sqrtf(var[i] + eps));
}

// act[r] = b_out + sum over the column tiles of the epilogue's row-dots, in tile order
__global__ __launch_bounds__(256) void head_part_sum_kernel(const float* __restrict__ part, int T, int B,
                                                            const float* __restrict__ b_out, float* __restrict__ act) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += part[(size_t)t * B + r];
    act[r] = a + b_out[0];
}

static inline size_t eval_fused_ws(const dun_fc_ctx* c) { return (size_t)c->depth * c->W + (size_t)(c->W / 128) * c->B; }

// whole 128 x 128 tiles, enough of them for the LDS-DMA GEMM, one head output, no likelihood pass riding on the head
static bool eval_fused(const dun_fc_ctx* c) {
    if (c->train || c->depth == 0 || c->O != 1 || c->obj_y) return false;
    if (c->W % 128 != 0 || c->B % 128 != 0 || (size_t)(c->B / 128) * (c->W / 128) < 128) return false;
    if (eval_fused_ws(c) > c->ws_floats) return false;
    return aligned16(c->H) && aligned16(c->p.w) && aligned16(c->ws);
}

static int eval_forward_blocks(const dun_fc_ctx* c, hipStream_t s) {
    const int W = c->W, B = c->B, T = W / 128;
    float* scale = c->ws;
    float* part = c->ws + (size_t)c->depth * W;
    const size_t n = (size_t)c->depth * W;
    hipLaunchKernelGGL(bn_eval_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->running_var, c->p.gamma,
                       c->bn_eps, n, scale);
    DUN_LAUNCH_CHECK();
    for (int i = 0; i < c->depth; ++i) {
        dun_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.A = Hd(c, i); g.lda = W; g.a_kcontig = 1;
        g.B = c->p.w + (size_t)i * W * W; g.ldb = W; g.b_kcontig = 1;
        g.C = Hd(c, i + 1); g.ldc = W;
        g.M = B; g.N = W; g.K = W;
        g.bias = c->p.b + (size_t)i * W;
        g.relu = 1;
        if (c->residual) { g.R = Hd(c, i); g.ldr = W; }
        g.ev_mean = c->running_mean + (size_t)i * W;
        g.ev_scale = scale + (size_t)i * W;
        g.ev_shift = c->p.beta + (size_t)i * W;
        g.ev_head_w = c->p.w_out;
        g.ev_head_part = part;
        int rc = gemm_f32(&g, s);
        if (rc) return rc;
        hipLaunchKernelGGL(head_part_sum_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, part, T, B, c->p.b_out,
                           c->act + (size_t)(i + 1) * B);
        DUN_LAUNCH_CHECK();
    }
    return DUN_OK;
}

// the shared head over every depth at once; with the likelihood epilogue when the context asks for it (O == 1)
static int head(const dun_fc_ctx* c, hipStream_t s) {
    if (c->obj_y && c->O == 1) return head_obj(c, 0, c->depth + 1, s);
    return head_forward(c->H, c->p.w_out, c->p.b_out, (c->depth + 1) * c->B, c->W, c->O, c->act, s);
}

static int check_bwd(const dun_fc_ctx* c, const char* who) {
    int rc = check_ctx(c, who);
    if (rc) return rc;
    DUN_REQUIRE(c->train, "%s: backward needs the buffers of a train-mode forward", who);
    DUN_REQUIRE(c->depth == c->L, "%s: backward through a truncated forward (depth=%d < L=%d) is not supported", who,
                c->depth, c->L);
    DUN_REQUIRE(c->dact && c->G, "%s: null dact/G", who);
    DUN_REQUIRE(c->L == 0 || c->dU, "%s: null dU", who);
    DUN_REQUIRE(c->g.w_in && c->g.b_in && c->g.w_out && c->g.b_out, "%s: null gradient outputs", who);
    DUN_REQUIRE(c->L == 0 || (c->g.w && c->g.b && c->g.gamma && c->g.beta), "%s: null block gradient outputs", who);
    return DUN_OK;
}

static int transpose_block_weights(const dun_fc_ctx* c, hipStream_t s);

static int bwd_begin(const dun_fc_ctx* c, hipStream_t s) {
    const int rows = (c->depth + 1) * c->B;
    int rc = transpose_block_weights(c, s);
    if (rc) return rc;
    if (c->head_grads_from_forward) {
        // dw_out / db_out partials were accumulated by the forward's head kernels (dact = the forward's own obj_dact):
        // add them up in fp64, two stages, fixed order -- no pass over the hidden states
        DUN_REQUIRE(c->obj_dwpart && c->obj_dbpart && c->O == 1 && head_fused(c),
                    "dun_fc_bwd_begin: head_grads_from_forward without the forward's partial sums");
        const int P = (c->depth + 1) * obj_nparts(c->B), K = c->W;
        const int rps = ceil_div(P, 256), splits = ceil_div(P, rps);
        DUN_REQUIRE((size_t)splits * K * 2 <= c->ws_floats, "dun_fc_bwd_begin: workspace too small");
        double* stage = reinterpret_cast<double*>(c->ws);
        hipLaunchKernelGGL(head_dw_reduce_kernel, dim3(K / 256, splits), dim3(256), 0, s, c->obj_dwpart, P, K, rps, stage);
        DUN_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_cols_kernel<double>, dim3(ceil_div(K, RCX), 1), dim3(256), 0, s, stage, splits, K,
                           (double*)nullptr, c->g.w_out, (float*)nullptr);
        DUN_LAUNCH_CHECK();
        hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(256), 0, s, c->obj_dbpart, P, c->g.b_out);
        DUN_LAUNCH_CHECK();
    } else {
        rc = head_backward(c->H, c->p.w_out, c->dact, rows, c->W, c->O, c->g.w_out, c->g.b_out, nullptr, c->ws,
                           c->ws_floats, s);
        if (rc) return rc;
    }
    const size_t n = (size_t)c->B * c->W;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(rank_init_kernel, dim3(grid), dim3(256), 0, s, c->G, c->dact + (size_t)c->depth * c->B * c->O,
                       c->p.w_out, c->B, c->W, c->O);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// The column sums sum_b g and sum_b g*xhat of block i are taken in the epilogue of the dX GEMM of block i+1, while the
// tile of g it has just computed is in registers (one read of Z_i instead of a pass over g and Z_i): every block but the
// deepest, whose incoming gradient does not come out of a GEMM.
// (one head output only: the statistics epilogue is compiled without the general rank-O update, whose per-row address
// registers made it spill its Z tile one HBM round trip at a time)
static inline bool stats_from_gemm(const dun_fc_ctx* c, int i) { return c->O == 1 && i < c->L - 1; }

static int bwd_block_reduce(const dun_fc_ctx* c, int i, hipStream_t s) {
    const int W = c->W, B = c->B;
    if (stats_from_gemm(c, i)) {
        // per-row-tile partials [2][T][W] left in the workspace by bwd_block_dx_gemm(i+1)
        hipLaunchKernelGGL(reduce_cols_kernel<float>, dim3(ceil_div(W, RCX), 2), dim3(256), 0, s, c->ws,
                           gemm_colstat_tiles(B, W), W, c->bn_sums, c->g.beta + (size_t)i * W, c->g.gamma + (size_t)i * W);
        DUN_LAUNCH_CHECK();
        return DUN_OK;
    }
    const int P = num_slabs(B);
    const int nt = stream_nt((size_t)B * W * 4) ? 1 : 0;
    const float* Z = Zd(c, i);
    const float* mean = c->bn_mean + (size_t)i * W;
    const float* rstd = c->bn_rstd + (size_t)i * W;
    if (vec4_ok(W, c->G, Z)) {
        if (nt)
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), dim3(ceil_div(W, CX * 4), P), dim3(256), 0, s, c->G, Z, mean,
                               rstd, B, W, P, slab_rows_for(B), c->ws);
        else
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(ceil_div(W, CX * 4), P), dim3(256), 0, s, c->G, Z, mean, rstd,
                               B, W, P, slab_rows_for(B), c->ws);
    } else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(ceil_div(W, CX), P), dim3(256), 0, s, c->G, Z, mean, rstd, B, W,
                           P, slab_rows_for(B), c->ws);
    DUN_LAUNCH_CHECK();
    // local sums are this rank's dbeta / dgamma; the fp64 copy is what a sharded job all-reduces
    hipLaunchKernelGGL(reduce_cols_kernel<float>, dim3(ceil_div(W, RCX), 2), dim3(256), 0, s, c->ws, P, W, c->bn_sums,
                       c->g.beta + (size_t)i * W, c->g.gamma + (size_t)i * W);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// BatchNorm + ReLU backward of block i: dU, db_i (the exchanged column sums are in bn_sums)
static int bwd_block_du(const dun_fc_ctx* c, int i, double global_batch, hipStream_t s) {
    const int W = c->W, B = c->B, P = num_slabs(B);
    const int nt = stream_nt((size_t)B * W * 4) ? 1 : 0;
    const float* Z = Zd(c, i);
    const float* mean = c->bn_mean + (size_t)i * W;
    const float* rstd = c->bn_rstd + (size_t)i * W;
    const float* gamma = c->p.gamma + (size_t)i * W;
    if (nt && W % 256 == 0 && W <= 2048 && vec4_ok(W, c->G, Z, c->dU) && aligned16(c->ws)) {
        // P workgroups of four wavefronts, slab_rows/4 rows per wavefront; partials [P][W] as below
#define BBR(N) case N: hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<N, true>), dim3(P), dim3(256), 0, s, c->G, Z, mean, rstd, \
                                          gamma, c->bn_sums, global_batch, B, c->dU, c->ws); break;
        switch (W / 256) { BBR(1) BBR(2) BBR(3) BBR(4) BBR(5) BBR(6) BBR(7) BBR(8) }
#undef BBR
    } else if (vec4_ok(W, c->G, Z, c->dU)) {
        if (nt)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), dim3(ceil_div(W, CX * 4), P), dim3(256), 0, s, c->G, Z, mean,
                               rstd, gamma, c->bn_sums, global_batch, B, W, slab_rows_for(B), c->dU, c->ws);
        else
            hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ceil_div(W, CX * 4), P), dim3(256), 0, s, c->G, Z, mean, rstd,
                               gamma, c->bn_sums, global_batch, B, W, slab_rows_for(B), c->dU, c->ws);
    } else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ceil_div(W, CX), P), dim3(256), 0, s, c->G, Z, mean, rstd, gamma,
                           c->bn_sums, global_batch, B, W, slab_rows_for(B), c->dU, c->ws);
    DUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_cols_kernel<float>, dim3(ceil_div(W, RCX), 1), dim3(256), 0, s, c->ws, P, W,
                       (double*)nullptr, c->g.b + (size_t)i * W, (float*)nullptr);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// dW_i[n_out, k_in] = sum_b dU[b, n_out] * H_i[b, k_in]   (dU of block i must still be in place)
static int bwd_block_dw(const dun_fc_ctx* c, int i, hipStream_t s) {
    const int W = c->W;
    dun_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = c->dU; g.lda = W; g.a_kcontig = 0;
    g.B = Hd(c, i); g.ldb = W; g.b_kcontig = 0;
    g.C = c->g.w + (size_t)i * W * W; g.ldc = W;
    g.M = W; g.N = W; g.K = c->B;
    g.splitk_ws = c->ws; g.splitk_ws_floats = c->ws_floats;  // whatever else lives in ws was consumed by then (stream order)
    return gemm_f32(&g, s);
}

// Transposed copies of the block weights for the dX GEMMs (behind the tiled path's workspace).  g = dU W_i contracts over
// the OUTPUT index of W_i[n_out][k_in], so W_i as stored is the N-contiguous operand; with both operands K-contiguous
// the same kernel runs at the forward GEMM's rate (measured box to box 25 ... 150 us per launch faster at the wide
// configuration's shape), for one 2 x L x W^2 x 4-byte transposition per backward (0.3 ms at 50 x 2048^2) and one more
// copy of the block weights in the workspace.
static inline bool dx_transposed(int B, int W, int L) {
    return L > 0 && W % 128 == 0 && B % 128 == 0 && (long long)ceil_div(B, 128) * (W / 128) >= 128;
}
static size_t base_workspace_floats(int B, int W, int O, int L);
static inline float* wT_base(const dun_fc_ctx* c) { return c->ws + base_workspace_floats(c->B, c->W, c->O, c->L); }

static int transpose_block_weights(const dun_fc_ctx* c, hipStream_t s) {
    if (!dx_transposed(c->B, c->W, c->L)) return DUN_OK;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(c->W / 64, c->W / 64, c->L), dim3(256), 0, s, c->p.w, wT_base(c), c->W);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// g <- [g +] dU W_i + dact_i w_out   (in place: every element is read and written by its own thread)
static int bwd_block_dx_gemm(const dun_fc_ctx* c, int i, hipStream_t s) {
    const int W = c->W;
    dun_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = c->dU; g.lda = W; g.a_kcontig = 1;
    g.B = c->p.w + (size_t)i * W * W; g.ldb = W; g.b_kcontig = 0;
    if (dx_transposed(c->B, W, c->L)) { g.B = wT_base(c) + (size_t)i * W * W; g.b_kcontig = 1; }
    g.C = c->G; g.ldc = W;
    g.M = c->B; g.N = W; g.K = W;
    if (c->residual) { g.R = c->G; g.ldr = W; }
    g.U = c->dact + (size_t)i * c->B * c->O; g.V = c->p.w_out; g.O = c->O; g.ldv = W;
    if (i >= 1 && stats_from_gemm(c, i - 1)) {  // the BatchNorm backward of block i-1 starts from this g
        g.bnz = Zd(c, i - 1); g.ldz = W;
        g.bn_mean = c->bn_mean + (size_t)(i - 1) * W;
        g.bn_rstd = c->bn_rstd + (size_t)(i - 1) * W;
        g.bnstats = c->ws;  // float [2][T][W]: fits the forward statistics' fp64 [2][T][W] region
    }
    return gemm_f32(&g, s);
}

static int bwd_block_apply(const dun_fc_ctx* c, int i, double global_batch, hipStream_t s) {
    int rc = bwd_block_du(c, i, global_batch, s);
    if (rc) return rc;
    if ((rc = bwd_block_dw(c, i, s))) return rc;
    return bwd_block_dx_gemm(c, i, s);
}

static int bwd_end(const dun_fc_ctx* c, hipStream_t s) {
    const int W = c->W, B = c->B, P = num_slabs(B);
    const int nt = stream_nt((size_t)B * W * 4) ? 1 : 0;
    // dW_in[n_out, k] = sum_b G[b, n_out] * x[b, k]
    dun_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.A = c->G; g.lda = W; g.a_kcontig = 0;
    g.B = c->x; g.ldb = c->din; g.b_kcontig = 0;
    g.C = c->g.w_in; g.ldc = c->din;
    g.M = W; g.N = c->din; g.K = B;
    g.splitk_ws = c->ws; g.splitk_ws_floats = c->ws_floats;  // ws is free here; colsum below reuses it afterwards
    int rc = gemm_f32(&g, s);
    if (rc) return rc;
    if (vec4_ok(W, c->G)) {
        if (nt)
            hipLaunchKernelGGL((colsum_kernel<4, true>), dim3(ceil_div(W, CX * 4), P), dim3(256), 0, s, c->G, B, W,
                               slab_rows_for(B), c->ws);
        else
            hipLaunchKernelGGL(colsum_kernel<4>, dim3(ceil_div(W, CX * 4), P), dim3(256), 0, s, c->G, B, W, slab_rows_for(B),
                               c->ws);
    } else
        hipLaunchKernelGGL(colsum_kernel<1>, dim3(ceil_div(W, CX), P), dim3(256), 0, s, c->G, B, W, slab_rows_for(B), c->ws);
    DUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_cols_kernel<float>, dim3(ceil_div(W, RCX), 1), dim3(256), 0, s, c->ws, P, W,
                       (double*)nullptr, c->g.b_in, (float*)nullptr);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

}  // namespace dun

using namespace dun;

// workspace of the tiled path, rounded to 256 bytes; behind it the small-net path keeps its second dU buffer, or the
// tiled path the transposed block weights of its dX GEMMs
namespace dun {
static size_t base_workspace_floats(int B, int W, int O, int L) {
    const size_t P = (size_t)ceil_div(B, 64);  // upper bound of the GEMM's row tiles (and of num_slabs(B))
    const size_t cols = 4 * P * (size_t)W;  // fp64 [2][P][W] forward statistics partials
    const size_t hb = head_backward_ws((L + 1) * B, W, O);
    const size_t sk = L > 0 ? 4 * (size_t)W * W : 0;  // split-K partials of dW = dU^T H (2 ranges at C3, 4 on small nets)
    size_t m = cols > hb ? cols : hb;
    if (sk > m) m = sk;
    return (m + 64 + 63) / 64 * 64;
}
}  // namespace dun

extern "C" int dun_fc_obj_num_partials(int B, int O) { return (B > 0 && O == 1) ? obj_nparts(B) : 0; }

extern "C" int dun_fc_head_is_fused(const dun_fc_ctx* c) { return (c != nullptr && head_fused(c)) ? 1 : 0; }

extern "C" int dun_fc_head_grads_in_forward(const dun_fc_ctx* c) {
    return (c != nullptr && c->train && c->O == 1 && (head_fused(c) || pn_eligible(c))) ? 1 : 0;
}

extern "C" int dun_fc_objective_in_forward(const dun_fc_ctx* c) {
    return (c != nullptr && c->obj_y && c->obj_fin_result && c->O == 1 && pn_eligible(c)) ? 1 : 0;
}

extern "C" size_t dun_fc_workspace_floats(int B, int din, int W, int O, int L) {
    (void)din;
    const size_t wT = dx_transposed(B, W, L) ? (size_t)L * W * W : 0;
    return base_workspace_floats(B, W, O, L) + sn_extra_ws_floats(B, W, L) + wT;
}

#define STREAM static_cast<hipStream_t>(stream)

extern "C" int dun_fc_input_layer(const dun_fc_ctx* c, dun_stream_t stream) {
    int rc = check_ctx(c, "dun_fc_input_layer");
    return rc ? rc : input_layer(c, STREAM);
}
extern "C" int dun_fc_block_linear(const dun_fc_ctx* c, int i, dun_stream_t stream) {
    int rc = check_block(c, i, "dun_fc_block_linear");
    return rc ? rc : block_linear(c, i, STREAM);
}
extern "C" int dun_fc_block_norm(const dun_fc_ctx* c, int i, double global_batch, dun_stream_t stream) {
    int rc = check_block(c, i, "dun_fc_block_norm");
    return rc ? rc : block_norm(c, i, global_batch, STREAM);
}
extern "C" int dun_fc_block_apply(const dun_fc_ctx* c, int i, dun_stream_t stream) {
    int rc = check_block(c, i, "dun_fc_block_apply");
    return rc ? rc : block_norm(c, i, (double)c->B, STREAM, false, /*apply_only=*/true);
}
extern "C" int dun_fc_head(const dun_fc_ctx* c, dun_stream_t stream) {
    int rc = check_ctx(c, "dun_fc_head");
    return rc ? rc : head(c, STREAM);
}
extern "C" int dun_fc_head_depth(const dun_fc_ctx* c, int d, dun_stream_t stream) {
    int rc = check_ctx(c, "dun_fc_head_depth");
    if (rc) return rc;
    DUN_REQUIRE(d >= 0 && d <= c->depth, "dun_fc_head_depth: depth %d outside [0,%d]", d, c->depth);
    if (c->O == 1 && (c->obj_y || head_fused(c))) return head_obj(c, d, 1, STREAM);
    return head_forward(c->H + (size_t)d * c->B * c->W, c->p.w_out, c->p.b_out, c->B, c->W, c->O,
                        c->act + (size_t)d * c->B * c->O, STREAM);
}
extern "C" int dun_fc_forward(const dun_fc_ctx* c, dun_stream_t stream) {
    int rc = check_ctx(c, "dun_fc_forward");
    if (rc) return rc;
    if (pn_eligible(c)) return pn_forward(c, STREAM);  // small net, one output: the whole forward in one launch
    if ((rc = input_layer(c, STREAM))) return rc;
    if (sn_eligible(c, false)) {  // small net: one column-owned launch per block
        for (int i = 0; i < c->depth; ++i)
            if ((rc = sn_forward_block(c, i, STREAM))) return rc;
        return head(c, STREAM);
    }
    if (eval_fused(c)) {  // inference on whole tiles: each block is ONE GEMM whose epilogue normalises, adds and dots
        if ((rc = head_obj(c, 0, 1, STREAM))) return rc;
        return eval_forward_blocks(c, STREAM);
    }
    const bool fused = head_fused(c);  // block_norm(i) writes act[i+1] itself: only depth 0 needs the head kernel
    if (fused && (rc = head_obj(c, 0, 1, STREAM))) return rc;
    for (int i = 0; i < c->depth; ++i) {
        if ((rc = block_linear(c, i, STREAM, /*reduce_stats=*/false))) return rc;
        if ((rc = block_norm(c, i, (double)c->B, STREAM, /*from_partials=*/c->train != 0))) return rc;
    }
    return fused ? DUN_OK : head(c, STREAM);
}

extern "C" int dun_fc_bwd_begin(const dun_fc_ctx* c, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_bwd_begin");
    return rc ? rc : bwd_begin(c, STREAM);
}
extern "C" int dun_fc_bwd_block_reduce(const dun_fc_ctx* c, int i, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_bwd_block_reduce");
    if (rc) return rc;
    DUN_REQUIRE(i >= 0 && i < c->L, "dun_fc_bwd_block_reduce: block %d outside [0,%d)", i, c->L);
    return bwd_block_reduce(c, i, STREAM);
}
extern "C" int dun_fc_bwd_block_apply(const dun_fc_ctx* c, int i, double global_batch, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_bwd_block_apply");
    if (rc) return rc;
    DUN_REQUIRE(i >= 0 && i < c->L, "dun_fc_bwd_block_apply: block %d outside [0,%d)", i, c->L);
    return bwd_block_apply(c, i, global_batch, STREAM);
}
extern "C" int dun_fc_bwd_block_dx(const dun_fc_ctx* c, int i, double global_batch, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_bwd_block_dx");
    if (rc) return rc;
    DUN_REQUIRE(i >= 0 && i < c->L, "dun_fc_bwd_block_dx: block %d outside [0,%d)", i, c->L);
    if ((rc = bwd_block_du(c, i, global_batch, STREAM))) return rc;
    return bwd_block_dx_gemm(c, i, STREAM);
}
extern "C" int dun_fc_bwd_block_dw(const dun_fc_ctx* c, int i, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_bwd_block_dw");
    if (rc) return rc;
    DUN_REQUIRE(i >= 0 && i < c->L, "dun_fc_bwd_block_dw: block %d outside [0,%d)", i, c->L);
    return bwd_block_dw(c, i, STREAM);
}
extern "C" int dun_fc_bwd_end(const dun_fc_ctx* c, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_bwd_end");
    return rc ? rc : bwd_end(c, STREAM);
}
extern "C" int dun_fc_backward(const dun_fc_ctx* c, dun_stream_t stream) {
    int rc = check_bwd(c, "dun_fc_backward");
    if (rc) return rc;
    if (pn_eligible(c)) {  // small net, one output: one launch for the trunk, one for every weight gradient
        float* dUall = c->ws + base_workspace_floats(c->B, c->W, c->O, c->L);
        const bool from_fwd = c->head_grads_from_forward != 0;
        if (from_fwd)
            DUN_REQUIRE(c->obj_dwpart && c->obj_dbpart, "dun_fc_backward: head_grads_from_forward without the forward's partial sums");
        else if ((rc = head_backward(c->H, c->p.w_out, c->dact, (c->depth + 1) * c->B, c->W, c->O, c->g.w_out, c->g.b_out,
                                     nullptr, c->ws, c->ws_floats, STREAM)))
            return rc;
        if ((rc = pn_backward_trunk(c, dUall, from_fwd, STREAM))) return rc;
        return pn_dw_all(c, dUall, STREAM);
    }
    if (sn_eligible(c, true)) {  // small net: head gradients, then one column-owned launch per block + input layer
        const int rows = (c->depth + 1) * c->B;
        if ((rc = head_backward(c->H, c->p.w_out, c->dact, rows, c->W, c->O, c->g.w_out, c->g.b_out, nullptr, c->ws,
                                c->ws_floats, STREAM)))
            return rc;
        return sn_backward_blocks(c, c->ws + base_workspace_floats(c->B, c->W, c->O, c->L), STREAM);
    }
    if ((rc = bwd_begin(c, STREAM))) return rc;
    for (int i = c->L - 1; i >= 0; --i) {
        if ((rc = bwd_block_reduce(c, i, STREAM))) return rc;
        if ((rc = bwd_block_apply(c, i, (double)c->B, STREAM))) return rc;
    }
    return bwd_end(c, STREAM);
}

extern "C" int dun_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                            float weight_decay, dun_stream_t stream) {
    DUN_REQUIRE(p && g && buf, "dun_sgd_step: null pointer");
    if (n == 0) return DUN_OK;
    if (!(aligned16(p) && aligned16(g) && aligned16(buf))) {
        size_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(sgd_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM, p, g, buf, n, lr, momentum,
                           weight_decay);
        DUN_LAUNCH_CHECK();
        return DUN_OK;
    }
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM, p, g, buf, n, lr, momentum, weight_decay);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// ---- the same update over several tensors in one launch ----------------------------------------------------------------
namespace dun {
struct SgdSegs {
    dun_sgd_segment seg[DUN_SGD_MAX_SEGMENTS];
    int nseg;
    const double* gate;  // nullptr, or a device count: the launch changes nothing when it is positive
};
// grid (workgroups per segment, segments); element by element (any alignment), the arithmetic of sgd_kernel
__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdSegs a, float lr, float momentum, float wd) {
    if (a.gate != nullptr && *a.gate > 0.0) return;  // (uniform: every workgroup reads the same word)
    const dun_sgd_segment sg = a.seg[blockIdx.y];
    float* __restrict__ p = sg.p;
    const float* __restrict__ g = sg.g;
    float* __restrict__ buf = sg.buf;
    const size_t n = sg.n;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf)) & 15) == 0) {
        const size_t n4 = n / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
            const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
            f32x4 bv = reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gv[e] + wd * pv[e];
                bv[e] = momentum * bv[e] + ge;
                pv[e] -= lr * bv[e];
            }
            reinterpret_cast<f32x4*>(buf)[i] = bv;
            reinterpret_cast<f32x4*>(p)[i] = pv;
        }
        const size_t i = n4 * 4 + threadIdx.x;
        if (blockIdx.x == 0 && i < n) {
            const float ge = g[i] + wd * p[i];
            const float b = momentum * buf[i] + ge;
            buf[i] = b;
            p[i] -= lr * b;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float ge = g[i] + wd * p[i];
        const float b = momentum * buf[i] + ge;
        buf[i] = b;
        p[i] -= lr * b;
    }
}
}  // namespace dun

// number of labels outside [0, C) other than the ignored -100: one workgroup (label vectors are a batch long)
namespace dun {
__global__ __launch_bounds__(256) void label_count_bad_kernel(const long long* __restrict__ y, size_t n, int C,
                                                              double* __restrict__ count) {
    __shared__ double red[4];
    double bad = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        const long long v = y[i];
        bad += ((v < 0 || v >= (long long)C) && v != -100) ? 1.0 : 0.0;
    }
    const double t = block_sum<256, double>(bad, red);
    if (threadIdx.x == 0) count[0] = t;
}
}  // namespace dun

extern "C" int dun_label_count_bad(const int64_t* y, size_t n, int C, double* count, dun_stream_t stream) {
    DUN_REQUIRE(count != nullptr && (n == 0 || y != nullptr) && C > 0, "dun_label_count_bad: null pointer or C=%d", C);
    hipLaunchKernelGGL(label_count_bad_kernel, dim3(1), dim3(256), 0, STREAM, reinterpret_cast<const long long*>(y), n, C,
                       count);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

static int sgd_multi_impl(const dun_sgd_segment* segments, int nseg, float lr, float momentum, float weight_decay,
                          const double* gate, dun_stream_t stream);

extern "C" int dun_sgd_step_multi(const dun_sgd_segment* segments, int nseg, float lr, float momentum, float weight_decay,
                                  dun_stream_t stream) {
    return sgd_multi_impl(segments, nseg, lr, momentum, weight_decay, nullptr, stream);
}

extern "C" int dun_sgd_step_multi_gated(const dun_sgd_segment* segments, int nseg, float lr, float momentum,
                                        float weight_decay, const double* skip_if_positive, dun_stream_t stream) {
    DUN_REQUIRE(skip_if_positive != nullptr, "dun_sgd_step_multi_gated: null gate");
    return sgd_multi_impl(segments, nseg, lr, momentum, weight_decay, skip_if_positive, stream);
}

static int sgd_multi_impl(const dun_sgd_segment* segments, int nseg, float lr, float momentum, float weight_decay,
                          const double* gate, dun_stream_t stream) {
    DUN_REQUIRE(nseg >= 0 && nseg <= DUN_SGD_MAX_SEGMENTS, "dun_sgd_step_multi: %d segments (at most %d)", nseg,
                DUN_SGD_MAX_SEGMENTS);
    DUN_REQUIRE(nseg == 0 || segments != nullptr, "dun_sgd_step_multi: null segments");
    SgdSegs a;
    memset(&a, 0, sizeof(a));
    a.gate = gate;
    size_t nmax = 0;
    for (int k = 0; k < nseg; ++k) {
        DUN_REQUIRE(segments[k].n == 0 || (segments[k].p && segments[k].g && segments[k].buf),
                    "dun_sgd_step_multi: null pointer in segment %d", k);
        if (segments[k].n == 0) continue;
        a.seg[a.nseg++] = segments[k];
        if (segments[k].n > nmax) nmax = segments[k].n;
    }
    if (a.nseg == 0) return DUN_OK;
    size_t blocks = (nmax / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)blocks, (unsigned)a.nseg), dim3(256), 0, STREAM, a, lr, momentum,
                       weight_decay);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// ---- two device-to-device copies in one launch ---------------------------------------------------------------------
namespace dun {
__global__ __launch_bounds__(256) void copy_pair_kernel(char* d0, const char* s0, size_t n0, char* d1, const char* s1,
                                                        size_t n1) {
    char* d = blockIdx.y == 0 ? d0 : d1;
    const char* s = blockIdx.y == 0 ? s0 : s1;
    const size_t n = blockIdx.y == 0 ? n0 : n1;
    const size_t stride = (size_t)gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        const size_t n16 = n / 16;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
            reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
        for (size_t i = n16 * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
}  // namespace dun

namespace dun {
__global__ __launch_bounds__(256) void copy_pair_indirect_kernel(char* d0, size_t n0, char* d1, size_t n1,
                                                                 const char* const* table) {
    char* d = blockIdx.y == 0 ? d0 : d1;
    const char* s = table[blockIdx.y];
    const size_t n = blockIdx.y == 0 ? n0 : n1;
    const size_t stride = (size_t)gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        const size_t n16 = n / 16;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
            reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
        for (size_t i = n16 * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
}  // namespace dun

extern "C" int dun_copy_pair_indirect(void* dst0, size_t bytes0, void* dst1, size_t bytes1, const void* const* src_table,
                                      dun_stream_t stream) {
    DUN_REQUIRE(src_table != nullptr && (bytes0 == 0 || dst0) && (bytes1 == 0 || dst1), "dun_copy_pair_indirect: null pointer");
    const size_t nmax = bytes0 > bytes1 ? bytes0 : bytes1;
    if (nmax == 0) return DUN_OK;
    size_t blocks = (nmax / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_pair_indirect_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, STREAM, static_cast<char*>(dst0),
                       bytes0, static_cast<char*>(dst1), bytes1, reinterpret_cast<const char* const*>(src_table));
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

extern "C" int dun_copy_pair(void* dst0, const void* src0, size_t bytes0, void* dst1, const void* src1, size_t bytes1,
                             dun_stream_t stream) {
    DUN_REQUIRE((bytes0 == 0 || (dst0 && src0)) && (bytes1 == 0 || (dst1 && src1)), "dun_copy_pair: null pointer");
    const size_t nmax = bytes0 > bytes1 ? bytes0 : bytes1;
    if (nmax == 0) return DUN_OK;
    size_t blocks = (nmax / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_pair_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, STREAM, static_cast<char*>(dst0),
                       static_cast<const char*>(src0), bytes0, static_cast<char*>(dst1), static_cast<const char*>(src1),
                       bytes1);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}
