// Depth-categorical probability module on device (gfx950): fused per-depth likelihood passes with the
// matching backward, deterministic depth reductions, ELBO / exact-marginal finalisation, and the
// depth-marginal predictor with fused train/eval metrics.
//
// All of these are HBM-streaming kernels: every element of act / dact is touched exactly once with
// coalesced (16-byte where alignment allows) accesses, per-depth sums go wavefront-shuffle -> LDS -> one
// fp64 partial per workgroup, and a second tiny kernel adds the partials in a fixed order, so results are
// bitwise reproducible run to run (no float atomics).
#include <math.h>

#include <mutex>

#include "common.h"
#include "depth_finalize.h"

namespace dun {

constexpr int GAUSS_PER_BLOCK = 4096;  // 256 threads x float4 x 4
constexpr int ROWS_PER_BLOCK = 256;    // softmax rows per workgroup

// Labels as nn.CrossEntropyLoss / nn.NLLLoss(reduction='none') treat them (src/DUN/training_wrappers.py:29-30):
// ignore_index = -100 contributes 0 loss and 0 gradient; any other label outside [0, C) is an error there (IndexError
// on CPU, device assert on CUDA).  Kernels cannot raise: such a row contributes NaN to the per-depth sum (so nothing
// computed from it looks valid) and no gradient, and the metric kernels count it (metrics[2]), which the wrappers turn
// into the IndexError with the one host copy a step makes anyway.
constexpr long long IGNORE_INDEX = -100;
__device__ __forceinline__ bool label_ok(long long y, int C) { return y >= 0 && y < (long long)C; }
__device__ __forceinline__ float label_fallback_ll(long long y) { return y == IGNORE_INDEX ? 0.f : __builtin_nanf(""); }

template <bool VEC, bool NT = false>
__global__ __launch_bounds__(256) void ll_gauss_kernel(const float* __restrict__ act, const float* __restrict__ y,
                                                       const float* __restrict__ log_std,
                                                       const float* __restrict__ w, int w_is_logits, int D, int B,
                                                       float inv_batch, float* __restrict__ dact,
                                                       float* __restrict__ ll, double* __restrict__ partials,
                                                       int nparts) {
    __shared__ double red[4];
    const int d = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const float* a_row = act + (size_t)d * B;
    float* g_row = dact ? dact + (size_t)d * B : nullptr;
    float* l_row = ll ? ll + (size_t)d * B : nullptr;
    const int base = part * GAUSS_PER_BLOCK;
    float acc = 0.f;
    if (VEC) {
        // the streaming loads go out first (branch-free, clamped: B % 4 == 0 and B >= 4 on this path); the depth weight
        // and sigma (a handful of uniform loads, two DPP reductions, three transcendentals) are worked out under them
        f32x4 a[4], t[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = min(base + (it * 256 + tid) * 4, B - 4);
            a[it] = ld4_stream<NT>(a_row + idx);
            t[it] = *reinterpret_cast<const f32x4*>(y + idx);
        }
        const float wd = depth_weight(w, w_is_logits, D, d);
        const float sigma = fmaxf(expf(log_std[0]), 1e-4f);
        const float inv_var = 1.f / (sigma * sigma);
        const float c0 = LOG_SQRT_2PI_F + logf(sigma);
        const float coef = -(wd * inv_batch) * inv_var;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = base + (it * 256 + tid) * 4;
            if (idx < B) {
                const f32x4 r = t[it] - a[it];
                acc += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                if (g_row) st4_stream<NT>(g_row + idx, coef * r);
                if (l_row) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = -(c0 + 0.5f * r[e] * r[e] * inv_var);
                    *reinterpret_cast<f32x4*>(l_row + idx) = o;
                }
            }
        }
    } else {
        const float wd = depth_weight(w, w_is_logits, D, d);
        const float sigma = fmaxf(expf(log_std[0]), 1e-4f);
        const float inv_var = 1.f / (sigma * sigma);
        const float c0 = LOG_SQRT_2PI_F + logf(sigma);
        const float coef = -(wd * inv_batch) * inv_var;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = base + (it * 256 + tid) * 4 + e;
                if (i < B) {
                    const float r = y[i] - a_row[i];
                    acc += r * r;
                    if (g_row) g_row[i] = coef * r;
                    if (l_row) l_row[i] = -(c0 + 0.5f * r * r * inv_var);
                }
            }
        }
    }
    // per-thread fp32 (<= 16 squares) -> fp64 from the wavefront on; DPP reduction, one barrier, fixed order
    const double ws = wave_allsum((double)acc);
    if ((tid & 63) == 0) red[tid >> 6] = ws;
    __syncthreads();
    if (tid == 0) partials[(size_t)d * nparts + part] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Large-batch form of the Gaussian pass (16-byte aligned rows): a workgroup owns 4096 batch columns and walks a
// RANGE of depths [blockIdx.y * dgroup, ...) over them.  The targets are loaded once and stay in registers for every
// depth (the per-depth form re-reads y D times), the depth softmax is set up once per workgroup instead of once per
// (depth, chunk), and the per-depth sums leave through one wavefront butterfly + one LDS slot per (depth, wave) with a
// single barrier at the end instead of two per depth.  Same partial layout, same summation tree per depth.
constexpr int GAUSS_DGROUP_MAX = 64;

template <bool NT>
__global__ __launch_bounds__(256) void ll_gauss_depths_kernel(const float* __restrict__ act, const float* __restrict__ y,
                                                              const float* __restrict__ log_std,
                                                              const float* __restrict__ w, int w_is_logits, int D,
                                                              int B, float inv_batch, float* __restrict__ dact,
                                                              float* __restrict__ ll, double* __restrict__ partials,
                                                              int nparts, int dgroup) {
    __shared__ double red[GAUSS_DGROUP_MAX][4];
    __shared__ float sh_sum;
    const int part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d0 = blockIdx.y * dgroup, d1 = min(D, d0 + dgroup);
    // depth softmax normaliser (fp32, as F.softmax), once per workgroup
    float mx = 0.f, inv_sum = 1.f;
    if (w != nullptr && w_is_logits) {
        mx = -INFINITY;
        for (int i = 0; i < D; ++i) mx = fmaxf(mx, w[i]);
        if (tid < 64) {
            float t = 0.f;
            for (int i = tid; i < D; i += 64) t += expf(w[i] - mx);
            t = wave_sum(t);
            if (tid == 0) sh_sum = t;
        }
        __syncthreads();
        inv_sum = sh_sum;
    }
    const float sigma = fmaxf(expf(log_std[0]), 1e-4f);
    const float inv_var = 1.f / (sigma * sigma);
    const float c0 = LOG_SQRT_2PI_F + logf(sigma);
    const int base = part * GAUSS_PER_BLOCK + tid * 4;
    f32x4 t[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = base + it * 1024;
        t[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < B) t[it] = *reinterpret_cast<const f32x4*>(y + idx);
    }
    for (int d = d0; d < d1; ++d) {
        float wd = 0.f;
        if (w != nullptr) wd = w_is_logits ? expf(w[d] - mx) / inv_sum : w[d];
        const float coef = -(wd * inv_batch) * inv_var;
        const float* a_row = act + (size_t)d * B;
        f32x4 a[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = base + it * 1024;
            a[it] = t[it];  // zero residual outside the batch
            if (idx < B) a[it] = ld4_stream<NT>(a_row + idx);
        }
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = base + it * 1024;
            const f32x4 r = t[it] - a[it];
            acc += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
            if (idx < B) {
                if (dact) st4_stream<NT>(dact + (size_t)d * B + idx, coef * r);
                if (ll) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = -(c0 + 0.5f * r[e] * r[e] * inv_var);
                    *reinterpret_cast<f32x4*>(ll + (size_t)d * B + idx) = o;
                }
            }
        }
        const double ws = wave_sum((double)acc);
        if (lane == 0) red[d - d0][wave] = ws;
    }
    __syncthreads();
    if (tid < d1 - d0) {
        // same order as block_sum: waves 0..3 left to right
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) r += red[tid][i];
        partials[(size_t)(d0 + tid) * nparts + part] = r;
    }
}

// G lanes cooperate on one row of C logits (G = pow2 >= min(C,64)); each lane keeps KMAX logits in registers.
template <int G, int KMAX>
__global__ __launch_bounds__(256) void ll_softmax_kernel(const float* __restrict__ act,
                                                         const int64_t* __restrict__ y,
                                                         const float* __restrict__ w, int w_is_logits, int D, int B,
                                                         int C, float inv_batch, float* __restrict__ dact,
                                                         float* __restrict__ ll, double* __restrict__ partials,
                                                         int nparts) {
    __shared__ double red[4];
    const int d = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const int gl = tid % G, grp = tid / G;
    constexpr int NGROUPS = 256 / G;
    const float wd = depth_weight(w, w_is_logits, D, d);
    const float coef = wd * inv_batch;
    const int row_end = min(B, (part + 1) * ROWS_PER_BLOCK);
    float acc = 0.f;
    for (int row = part * ROWS_PER_BLOCK + grp; row < row_end; row += NGROUPS) {
        const size_t off = ((size_t)d * B + row) * C;
        const float* a = act + off;
        float v[KMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int j = gl + k * G;
            v[k] = j < C ? a[j] : -INFINITY;
            mx = fmaxf(mx, v[k]);
        }
        mx = group_max<G>(mx);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s += (gl + k * G < C) ? expf(v[k] - mx) : 0.f;
        s = group_sum<G>(s);
        const float lse = mx + logf(s);
        const long long yl = y[row];
        const bool ok = label_ok(yl, C);
        const int yb = ok ? (int)yl : -1;
        float ay = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) ay += (gl + k * G == yb) ? v[k] : 0.f;
        ay = group_sum<G>(ay);
        const float row_ll = ok ? ay - lse : label_fallback_ll(yl);
        if (gl == 0) {
            acc += row_ll;
            if (ll) ll[(size_t)d * B + row] = row_ll;
        }
        if (dact) {
            float* g = dact + off;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int j = gl + k * G;
                if (j < C) {
                    const float pr = expf(v[k] - lse);
                    g[j] = ok ? -coef * ((j == yb ? 1.f : 0.f) - pr) : 0.f;
                }
            }
        }
    }
    const double tot = block_sum<256, double>((double)acc, red);
    if (tid == 0) partials[(size_t)d * nparts + part] = tot;
}

// Few classes (C <= CMAX <= 32, the 10-class image DUNs): the lane-group reductions above cost a dozen dependent
// cross-lane shuffles per four rows and leave most lanes idle (1.9 TB/s at C = 10); with the rows staged in LDS a THREAD can own a row
// instead: C LDS reads, max / exp / sum / log in registers, C LDS writes, no shuffle, all 64 lanes busy.
// One pass: the workgroup's 256 rows (<= 32 KiB) in, gradient out, both with 16-byte lane accesses (4.5 TB/s at C = 10).
// Needs B*C % 4 == 0 and 16-byte aligned act/dact; anything else, and C > 32, takes the lane-group kernel.
template <int CMAX, bool NT>
__global__ __launch_bounds__(256) void ll_softmax_rowthread_kernel(const float* __restrict__ act,
                                                                   const int64_t* __restrict__ y,
                                                                   const float* __restrict__ w, int w_is_logits, int D,
                                                                   int B, int C, float inv_batch,
                                                                   float* __restrict__ dact, float* __restrict__ ll,
                                                                   double* __restrict__ partials, int nparts) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ double red[4];
    const int d = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const int r0 = part * ROWS_PER_BLOCK;
    const int nrows = min(ROWS_PER_BLOCK, B - r0);
    const int nfl = nrows * C;  // a multiple of 4: r0 % 4 == 0 and B*C % 4 == 0
    const size_t off = ((size_t)d * B + r0) * C;
    const float* src = act + off;
    // every global load of the workgroup is issued before anything waits: the label, then the thread's share of the
    // tile into registers (a run-time trip count made this a load / wait / LDS-store round trip per 4 KiB; branch-free,
    // clamped addresses), and the depth weight (a few loads and two DPP reductions per wavefront) is worked out underneath
    const int64_t yv = y[r0 + min(tid, nrows - 1)];
    constexpr int NV = (ROWS_PER_BLOCK * CMAX / 4 + 255) / 256;
    f32x4 stage[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) stage[j] = ld4_stream<NT>(src + min((j * 256 + tid) * 4, nfl - 4));
    const float wd = depth_weight(w, w_is_logits, D, d);
    const float coef = wd * inv_batch;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 256 + tid) * 4;
        if (i < nfl) *reinterpret_cast<f32x4*>(tile + i) = stage[j];
    }
    __syncthreads();
    float acc = 0.f;
    if (tid < nrows) {
        float* a = tile + tid * C;
        float v[CMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < CMAX; ++k) {
            v[k] = k < C ? a[k] : -INFINITY;
            mx = fmaxf(mx, v[k]);
        }
        const bool ok = label_ok(yv, C);  // -100 (ignored) and out-of-range labels: no gradient, see label_ok
        const int yb = ok ? (int)yv : 0;
        const float ay = a[yb];  // the target's logit straight from the staged row (no select per class)
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CMAX; ++k) {
            v[k] = expf(v[k] - mx);  // exp(-inf) = 0 for the padding
            s += v[k];
        }
        const float lse = mx + logf(s);
        const float row_ll = ok ? ay - lse : label_fallback_ll(yv);
        acc = row_ll;
        if (ll) ll[(size_t)d * B + r0 + tid] = row_ll;
        if (dact) {
            // -coef (onehot - softmax) = (coef / s) exp(.) everywhere, minus coef at the target class
            const float sc = ok ? coef / s : 0.f;
            const float gy = sc * expf(ay - mx) - (ok ? coef : 0.f);
#pragma unroll
            for (int k = 0; k < CMAX; ++k)
                if (k < C) a[k] = sc * v[k];
            a[yb] = gy;
        }
    }
    if (dact) {
        __syncthreads();
        float* dst = dact + off;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (j * 256 + tid) * 4;
            if (i < nfl) st4_stream<NT>(dst + i, *reinterpret_cast<const f32x4*>(tile + i));
        }
    }
    const double tot = block_sum<256, double>((double)acc, red);
    if (tid == 0) partials[(size_t)d * nparts + part] = tot;
}

// 32 < C <= 1024 (CIFAR-100 / ImageNet-sized heads): the same LDS staging with G lanes per row (C/G <= 32 logits per
// lane in registers, log2 G shuffle levels per reduction instead of six), `tile_rows` rows per pass (<= 32 KiB).
template <int G, int KMAX, bool NT>
__global__ __launch_bounds__(256) void ll_softmax_rowgroup_kernel(const float* __restrict__ act,
                                                                  const int64_t* __restrict__ y,
                                                                  const float* __restrict__ w, int w_is_logits, int D,
                                                                  int B, int C, float inv_batch,
                                                                  float* __restrict__ dact, float* __restrict__ ll,
                                                                  double* __restrict__ partials, int nparts,
                                                                  int tile_rows) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ double red[4];
    __shared__ int ytile[ROWS_PER_BLOCK];
    const int d = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const int gl = tid % G, grp = tid / G;
    constexpr int NGROUPS = 256 / G;
    constexpr int NV = 8;  // float4 per thread and pass: a pass is at most 32 KiB
    const int row_begin = part * ROWS_PER_BLOCK;
    const int row_end = min(B, row_begin + ROWS_PER_BLOCK);
    // Loads first, waits later: the labels of all rows of the workgroup (one per thread, kept in LDS: a load per row
    // inside the row loop was a dependent global round trip per row) and the first pass of logits into registers
    // (branch-free, clamped addresses; pass sizes are multiples of 4 floats: r0 % 4 == 0 and B*C % 4 == 0); the depth
    // weight is worked out underneath them.  The logits of pass p+1 are fetched into the same registers while pass p
    // is reduced out of LDS.
    // label code kept in LDS: the class, -100 (ignored) or -1 (any other label outside [0, C): see label_ok)
    const long long yl = y[row_begin + min(tid, row_end - row_begin - 1)];
    const int yv = label_ok(yl, C) ? (int)yl : (yl == IGNORE_INDEX ? (int)IGNORE_INDEX : -1);
    f32x4 stage[NV];
    {
        const int nfl0 = min(tile_rows, row_end - row_begin) * C;
        const float* src0 = act + ((size_t)d * B + row_begin) * C;
#pragma unroll
        for (int j = 0; j < NV; ++j) stage[j] = ld4_stream<NT>(src0 + min((j * 256 + tid) * 4, nfl0 - 4));
    }
    ytile[tid] = yv;
    const float wd = depth_weight(w, w_is_logits, D, d);
    const float coef = wd * inv_batch;
    float acc = 0.f;
    for (int r0 = row_begin; r0 < row_end; r0 += tile_rows) {
        const int nrows = min(tile_rows, row_end - r0);
        const int nfl = nrows * C;
        const size_t off = ((size_t)d * B + r0) * C;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (j * 256 + tid) * 4;
            if (i < nfl) *reinterpret_cast<f32x4*>(tile + i) = stage[j];
        }
        __syncthreads();
        if (r0 + tile_rows < row_end) {  // (uniform) next pass on its way while this one is reduced
            const int nfl1 = min(tile_rows, row_end - r0 - tile_rows) * C;
            const float* src1 = act + off + (size_t)tile_rows * C;
#pragma unroll
            for (int j = 0; j < NV; ++j) stage[j] = ld4_stream<NT>(src1 + min((j * 256 + tid) * 4, nfl1 - 4));
        }
        for (int row = grp; row < nrows; row += NGROUPS) {  // (uniform per group: the shuffles below stay converged)
            float* a = tile + row * C;
            float v[KMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int j = gl + k * G;
                v[k] = j < C ? a[j] : -INFINITY;
                mx = fmaxf(mx, v[k]);
            }
            mx = group_max<G>(mx);
            const int yb = ytile[r0 - row_begin + row];
            const bool ok = yb >= 0;
            float s = 0.f, ay = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                ay = (gl + k * G == yb) ? v[k] : ay;
                v[k] = expf(v[k] - mx);  // exp(-inf) = 0 for the padding
                s += v[k];
            }
            s = group_sum<G>(s);
            ay = group_sum<G>(ay);
            const float row_ll = ok ? ay - (mx + logf(s)) : label_fallback_ll(yb);
            if (gl == 0) {
                acc += row_ll;
                if (ll) ll[(size_t)d * B + r0 + row] = row_ll;
            }
            if (dact) {
                const float inv_s = 1.f / s;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int j = gl + k * G;
                    if (j < C) a[j] = ok ? -coef * ((j == yb ? 1.f : 0.f) - v[k] * inv_s) : 0.f;
                }
            }
        }
        __syncthreads();
        if (dact) {
            float* dst = dact + off;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = (j * 256 + tid) * 4;
                if (i < nfl) st4_stream<NT>(dst + i, *reinterpret_cast<const f32x4*>(tile + i));
            }
            __syncthreads();  // the next pass overwrites the tile
        }
    }
    const double tot = block_sum<256, double>((double)acc, red);
    if (tid == 0) partials[(size_t)d * nparts + part] = tot;
}

// out[j] = sum_i in[j*n + i], fixed order; one wavefront per output
__global__ __launch_bounds__(64) void reduce_rows_f64_kernel(const double* __restrict__ in, int n,
                                                             double* __restrict__ out) {
    const int j = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += in[(size_t)j * n + i];
    s = wave_sum(s);
    if (lane == 0) out[j] = s;
}

// depth_finalize_body: depth_finalize.h

__global__ __launch_bounds__(64) void depth_finalize_kernel(const double* __restrict__ sums, int gaussian,
                                                            const float* __restrict__ q_logits,
                                                            const float* __restrict__ logprior,
                                                            const float* __restrict__ prior_ll,
                                                            const float* __restrict__ log_std, int D, double N,
                                                            double batch, double beta, double eps, int objective,
                                                            float* __restrict__ res) {
    depth_finalize_body(sums, gaussian, q_logits, logprior, prior_ll, log_std, D, N, batch, beta, eps, objective, res);
}

// reduce_rows_f64_kernel + depth_finalize_kernel in ONE launch (single-device jobs: nothing to exchange between the
// two): sixteen wavefronts add the per-workgroup partials of the depths (the same lanes-strided + butterfly order as
// reduce_rows_f64_kernel, so both routes give the same bits), the sums stay in LDS, the first wavefront finalises.
constexpr int FINALIZE_WAVES = 16;
__global__ __launch_bounds__(64 * FINALIZE_WAVES) void depth_reduce_finalize_kernel(
    const double* __restrict__ partials, int nparts, double* __restrict__ sums_out, int gaussian,
    const float* __restrict__ q_logits, const float* __restrict__ logprior, const float* __restrict__ prior_ll,
    const float* __restrict__ log_std, int D, double N, double batch, double beta, double eps, int objective,
    float* __restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) double sh_sums[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = wave; d < D; d += FINALIZE_WAVES) {
        double t = 0.0;
        for (int i = lane; i < nparts; i += 64) t += partials[(size_t)d * nparts + i];
        t = wave_sum(t);
        if (lane == 0) {
            sh_sums[d] = t;
            if (sums_out) sums_out[d] = t;
        }
    }
    __syncthreads();
    if (wave == 0)
        depth_finalize_body(sh_sums, gaussian, q_logits, logprior, prior_ll, log_std, D, N, batch, beta, eps, objective,
                            res);
}

// ---- depth-marginal prediction ----
// value of an fp32 expression, rounded here: the empty asm makes it opaque, so it cannot become one half of an FMA
__device__ __forceinline__ float rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <bool VEC, bool NT = false>
__global__ __launch_bounds__(256) void predict_gauss_kernel(const float* __restrict__ act, const float* __restrict__ q,
                                                            int D, int n, float* __restrict__ mean,
                                                            float* __restrict__ std, const float* __restrict__ y,
                                                            const float* __restrict__ log_std,
                                                            double* __restrict__ partials, int nparts,
                                                            int q_is_logits, float* __restrict__ q_out,
                                                            const float* __restrict__ obj_src,
                                                            double* __restrict__ obj_dst, double* done,
                                                            const float* __restrict__ noise_log_std) {
    __shared__ double red[4];
    __shared__ float s_q[256];
    const int tid = threadIdx.x;
    const int idx = (blockIdx.x * 256 + tid) * 4;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool any = idx < n;
    // what depends on nothing goes out first: at the sizes of a training step's metric pass (one workgroup, a few KB) the
    // kernel is a chain of memory round trips, and every load issued late is one more of them
    float yv[4] = {0.f, 0.f, 0.f, 0.f};
    float ls = 0.f, nls = 0.f;
    if (noise_log_std) nls = noise_log_std[0];
    if (partials) {
        ls = log_std[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) yv[e] = y[min(idx + e, n - 1)];
    }
    constexpr int DCH = 8;  // depths whose loads are in flight together
    f32x4 first[DCH];
    if (VEC && any) {
#pragma unroll
        for (int u = 0; u < DCH; ++u) first[u] = ld4_stream<NT>(act + (size_t)min(u, D - 1) * n + idx);
    }
    // q given as logits (the post-step q(d) of DUN_VI.fit): softmax in fp32 as F.softmax, every wavefront for itself
    float qmx = 0.f, qinv = 1.f;
    if (q_is_logits) {
        const int lane = tid & 63;
        qmx = -INFINITY;
        for (int i = lane; i < D; i += 64) qmx = fmaxf(qmx, q[i]);
        qmx = wave_allmax(qmx);
        float s = 0.f;
        for (int i = lane; i < D; i += 64) s += expf(q[i] - qmx);
        qinv = wave_allsum(s);
    }
    const bool q_in_lds = D <= 256;
    if (q_in_lds) {
        if (tid < D) s_q[tid] = q_is_logits ? expf(q[tid] - qmx) / qinv : q[tid];
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (q_out)
            for (int i = tid; i < D; i += 256) q_out[i] = q_in_lds ? s_q[i] : (q_is_logits ? expf(q[i] - qmx) / qinv : q[i]);
        if (obj_dst && tid == 0) obj_dst[0] = (double)obj_src[0];
    }
    if (any) {
        for (int d0 = 0; d0 < D; d0 += DCH) {
            f32x4 t[DCH];
#pragma unroll
            for (int u = 0; u < DCH; ++u) {
                const float* a = act + (size_t)min(d0 + u, D - 1) * n + idx;
                if (VEC) {
                    t[u] = d0 == 0 ? first[u] : ld4_stream<NT>(a);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[u][e] = a[min(e, n - 1 - idx)] * ((idx + e < n) ? 1.f : 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < DCH; ++u) {
                const int d = d0 + u;
                if (d < D) {
                    const float qd = q_in_lds ? s_q[d] : (q_is_logits ? expf(q[d] - qmx) / qinv : q[d]);
                    // every product rounded on its own, as (q * preds).sum(0) and (q * preds**2).sum(0) do in the
                    // reference: an FMA here would make var = sum(q a^2) - mu^2 come out slightly negative (-> NaN std)
                    // where the reference gets exactly 0 (one-hot posteriors, single depth)
                    // (hipcc contracts a*b+c into an FMA by default, through __fmul_rn too: rounded() hides each product)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        mu[e] += rounded(qd * t[u][e]);
                        m2[e] += rounded(qd * rounded(t[u][e] * t[u][e]));
                    }
                }
            }
        }
    }
    float sd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sd[e] = sqrtf(m2[e] - rounded(mu[e] * mu[e]));  // NaN if negative, as pow(0.5)
    double nll = 0.0, se = 0.0;
    if (any) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (idx + e < n) {
                mean[idx + e] = mu[e];
                if (std) {
                    float so = sd[e];
                    if (noise_log_std) {  // (model_std ** 2 + log_std.exp() ** 2).pow(0.5), training_wrappers.py:153
                        const float sg = expf(nls);
                        so = sqrtf(rounded(so * so) + rounded(sg * sg));
                    }
                    std[idx + e] = so;
                }
            }
        }
        if (partials) {
            const float sig = fmaxf(expf(ls), 1e-4f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (idx + e < n) {
                    const float st = sqrtf(sig * sig + sd[e] * sd[e]);
                    const float r = yv[e] - mu[e];
                    nll += (double)(LOG_SQRT_2PI_F + logf(st) + r * r / (2.f * st * st));
                    se += (double)(r * r);
                }
            }
        }
    }
    if (partials) {
        const double t0 = block_sum<256, double>(nll, red);
        const double t1 = block_sum<256, double>(se, red);
        if (tid == 0) {
            partials[blockIdx.x] = t0;
            partials[nparts + blockIdx.x] = t1;
        }
    }
    if (done != nullptr && tid == 0 && blockIdx.x == 0) {  // (one-workgroup launches only: this thread wrote every scalar)
        __threadfence_system();
        __hip_atomic_store(done, 1.0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int G, int KMAX>
__global__ __launch_bounds__(256) void predict_softmax_kernel(const float* __restrict__ act,
                                                              const float* __restrict__ q, int D, int B, int C,
                                                              float* __restrict__ probs,
                                                              const int64_t* __restrict__ y,
                                                              double* __restrict__ partials, int nparts) {
    __shared__ double red[4];
    const int part = blockIdx.x, tid = threadIdx.x;
    const int gl = tid % G, grp = tid / G;
    constexpr int NGROUPS = 256 / G;
    const int row_end = min(B, (part + 1) * ROWS_PER_BLOCK);
    double nll = 0.0, nerr = 0.0, nbad = 0.0;
    for (int row = part * ROWS_PER_BLOCK + grp; row < row_end; row += NGROUPS) {
        float pr[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) pr[k] = 0.f;
        for (int d = 0; d < D; ++d) {
            const float* a = act + ((size_t)d * B + row) * C;
            float v[KMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int j = gl + k * G;
                v[k] = j < C ? a[j] : -INFINITY;
                mx = fmaxf(mx, v[k]);
            }
            mx = group_max<G>(mx);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                v[k] = (gl + k * G < C) ? expf(v[k] - mx) : 0.f;
                s += v[k];
            }
            s = group_sum<G>(s);
            const float sc = q[d] / s;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) pr[k] += sc * v[k];
        }
        float* o = probs + (size_t)row * C;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (gl + k * G < C) o[gl + k * G] = pr[k];
        if (partials) {
            const long long yl = y[row];
            const bool ok = label_ok(yl, C);
            const int yb = ok ? (int)yl : -1;
            float py = 0.f, best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int j = gl + k * G;
                if (j < C) {
                    if (j == yb) py = pr[k];
                    if (pr[k] > best) { best = pr[k]; bi = j; }
                }
            }
            py = group_sum<G>(py);
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (gl == 0) {
                // NLLLoss(reduction='none'): 0 for the ignored label; pred.ne(y) counts it as an error all the same
                nll += ok ? (double)(-logf(py)) : (double)(-label_fallback_ll(yl));
                nerr += (bi != yb) ? 1.0 : 0.0;
                nbad += (!ok && yl != IGNORE_INDEX) ? 1.0 : 0.0;
            }
        }
    }
    if (partials) {
        const double t0 = block_sum<256, double>(nll, red);
        const double t1 = block_sum<256, double>(nerr, red);
        const double t2 = block_sum<256, double>(nbad, red);
        if (tid == 0) {
            partials[part] = t0;
            partials[nparts + part] = t1;
            partials[2 * nparts + part] = t2;
        }
    }
}

// LDS-staged depth-marginal class probabilities (16-byte aligned problems).  The kernel above walks D depths x 16
// passes of lane-group shuffles per workgroup as one serial chain (100 us for 4096 rows x 12 exits x 10 classes).
// Here the rows of one depth are one contiguous run: it is prefetched into registers with 16-byte lane accesses while
// the previous depth is reduced out of LDS, G lanes own a row (G = 1 for C <= 32: a thread per row, no shuffles) and
// keep its running sum_d q_d softmax(a_d) in registers; probabilities leave through the same LDS tile.
template <int G, int KMAX, bool NT>
__global__ __launch_bounds__(256) void predict_softmax_lds_kernel(const float* __restrict__ act,
                                                                  const float* __restrict__ q, int D, int B, int C,
                                                                  float* __restrict__ probs,
                                                                  const int64_t* __restrict__ y,
                                                                  double* __restrict__ partials, int nparts,
                                                                  int rows_wg) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ double red[4];
    constexpr int RPP = 256 / G;  // rows per pass
    constexpr int NF = (KMAX + 3) / 4;  // float4 per thread and tile: RPP * C <= 256 * KMAX floats
    const int part = blockIdx.x, tid = threadIdx.x;
    const int gl = tid % G, grp = tid / G;
    const int row_end = min(B, (part + 1) * rows_wg);  // rows_wg: a multiple of RPP
    double nll = 0.0, nerr = 0.0, nbad = 0.0;
    for (int r0 = part * rows_wg; r0 < row_end; r0 += RPP) {
        const int nrows = min(RPP, row_end - r0);
        const int nfl = nrows * C;
        const bool mine = grp < nrows;
        float pr[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) pr[k] = 0.f;
        f32x4 pre[NF];
        // branch-free, clamped addresses (tile sizes are multiples of 4 floats): a branch per load serialises them
        auto fetch = [&](int d) {
            const float* src = act + ((size_t)d * B + r0) * C;  // multiple of 4 floats (r0 % 4 == 0, B*C % 4 == 0)
#pragma unroll
            for (int f = 0; f < NF; ++f) pre[f] = ld4_stream<NT>(src + min((f * 256 + tid) * 4, nfl - 4));
        };
        const long long yl = partials ? y[r0 + min(grp, nrows - 1)] : 0;  // on its way under the whole depth loop
        const bool ok = label_ok(yl, C);
        const int yb = ok ? (int)yl : -1;
        fetch(0);
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int i = (f * 256 + tid) * 4;
                if (i < nfl) *reinterpret_cast<f32x4*>(tile + i) = pre[f];  // (tile is padded to a multiple of 4 floats)
            }
            __syncthreads();
            if (d + 1 < D) fetch(d + 1);
            if (mine) {
                const float* a = tile + grp * C;
                float v[KMAX];
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int j = gl + k * G;
                    v[k] = j < C ? a[j] : -INFINITY;
                    mx = fmaxf(mx, v[k]);
                }
                mx = group_max<G>(mx);
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    v[k] = expf(v[k] - mx);  // exp(-inf) = 0 for the padding
                    s += v[k];
                }
                s = group_sum<G>(s);
                const float sc = q[d] / s;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) pr[k] += sc * v[k];
            }
            __syncthreads();
        }
        if (mine) {
            float* o = tile + grp * C;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (gl + k * G < C) o[gl + k * G] = pr[k];
            if (partials) {
                float py = 0.f, best = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int j = gl + k * G;
                    if (j < C) {
                        if (j == yb) py = pr[k];
                        if (pr[k] > best) { best = pr[k]; bi = j; }
                    }
                }
                py = group_sum<G>(py);
#pragma unroll
                for (int off = G / 2; off > 0; off >>= 1) {
                    const float ob = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (gl == 0) {
                    nll += ok ? (double)(-logf(py)) : (double)(-label_fallback_ll(yl));
                    nerr += (bi != yb) ? 1.0 : 0.0;
                    nbad += (!ok && yl != IGNORE_INDEX) ? 1.0 : 0.0;
                }
            }
        }
        __syncthreads();
        float* dst = probs + (size_t)r0 * C;  // r0 * C % 4 == 0
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int i = (f * 256 + tid) * 4;
            if (i < nfl) *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(tile + i);
        }
        __syncthreads();  // the next pass overwrites the tile
    }
    if (partials) {
        const double t0 = block_sum<256, double>(nll, red);
        const double t1 = block_sum<256, double>(nerr, red);
        const double t2 = block_sum<256, double>(nbad, red);
        if (tid == 0) {
            partials[part] = t0;
            partials[nparts + part] = t1;
            partials[2 * nparts + part] = t2;
        }
    }
}

// ---- host side ----
static int softmax_group(int C, int* kmax) {
    int G = 2;
    while (G < C && G < 64) G <<= 1;
    int k = 1;
    while (G * k < C) k <<= 1;
    *kmax = k;
    return G;
}

// Rows of the depth-marginal softmax predictor per workgroup.  The LDS kernel reduces 256/g rows per pass (g lanes per
// row); 256 rows per workgroup keep the partial count small on large batches, but a small batch of wide rows
// (4096 x 100 classes: 16 workgroups walking 4 passes x D depths each) leaves the chip empty: one pass per workgroup
// until there are at least four workgroups per CU.
static int predict_rows_per_wg(int B, int C, bool lds_path) {
    if (!lds_path) return ROWS_PER_BLOCK;
    int g = 1;
    while (g * 32 < C) g <<= 1;
    const int rpp = 256 / g;
    return ceil_div(B, ROWS_PER_BLOCK) >= 1024 ? ROWS_PER_BLOCK : rpp;
}

#define DUN_SOFTMAX_DISPATCH(KERNEL, G, KM, grid, s, ...) DUN_SOFTMAX_DISPATCH_LDS(KERNEL, G, KM, grid, 0, s, __VA_ARGS__)
#define DUN_SOFTMAX_DISPATCH_LDS(KERNEL, G, KM, grid, lds, s, ...)                                             \
    do {                                                                                              \
        if (KM == 1) {                                                                                \
            switch (G) {                                                                              \
                case 2: hipLaunchKernelGGL((KERNEL<2, 1>), grid, dim3(256), lds, s, __VA_ARGS__); break;   \
                case 4: hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(256), lds, s, __VA_ARGS__); break;   \
                case 8: hipLaunchKernelGGL((KERNEL<8, 1>), grid, dim3(256), lds, s, __VA_ARGS__); break;   \
                case 16: hipLaunchKernelGGL((KERNEL<16, 1>), grid, dim3(256), lds, s, __VA_ARGS__); break; \
                case 32: hipLaunchKernelGGL((KERNEL<32, 1>), grid, dim3(256), lds, s, __VA_ARGS__); break; \
                default: hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(256), lds, s, __VA_ARGS__); break; \
            }                                                                                         \
        } else if (KM == 2) {                                                                         \
            hipLaunchKernelGGL((KERNEL<64, 2>), grid, dim3(256), lds, s, __VA_ARGS__);                  \
        } else if (KM == 4) {                                                                         \
            hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(256), lds, s, __VA_ARGS__);                  \
        } else if (KM == 8) {                                                                         \
            hipLaunchKernelGGL((KERNEL<64, 8>), grid, dim3(256), lds, s, __VA_ARGS__);                  \
        } else {                                                                                      \
            hipLaunchKernelGGL((KERNEL<64, 16>), grid, dim3(256), lds, s, __VA_ARGS__);                 \
        }                                                                                             \
    } while (0)

}  // namespace dun

using namespace dun;

extern "C" int dun_ll_num_partials(int B, int C) {
    if (B <= 0) return 0;
    return C <= 1 ? ceil_div(B, GAUSS_PER_BLOCK) : ceil_div(B, ROWS_PER_BLOCK);
}

extern "C" int dun_predict_num_partials(int n, int C) {
    if (n <= 0) return 0;
    if (C <= 1) return ceil_div(n, 1024);
    return ceil_div(n, predict_rows_per_wg(n, C, true));  // an upper bound of what dun_predict_softmax launches
}

extern "C" int dun_ll_gauss(const float* act, const float* y, const float* log_std, const float* w, int w_is_logits,
                            int D, int B, float inv_batch, float* dact, float* ll, double* partials,
                            dun_stream_t stream) {
    DUN_REQUIRE(act && y && log_std && partials, "dun_ll_gauss: null pointer");
    DUN_REQUIRE(D > 0 && B > 0, "dun_ll_gauss: empty input D=%d B=%d", D, B);
    DUN_REQUIRE(D <= 65535, "dun_ll_gauss: D=%d exceeds grid.y", D);
    DUN_REQUIRE(dact == nullptr || w != nullptr, "dun_ll_gauss: dact requested without depth weights");
    const int nparts = dun_ll_num_partials(B, 1);
    const bool vec = (B % 4 == 0) && aligned16(act) && aligned16(y) && (!dact || aligned16(dact)) &&
                     (!ll || aligned16(ll));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool nt = (size_t)D * B * 4 * (dact ? 2 : 1) >= NT_MIN_BYTES;
    if (vec && nparts >= 512) {
        // B >= 2^21: enough batch chunks to fill the chip without spreading single depths over workgroups, and act/dact
        // no longer fit the 256 MiB Infinity Cache: walk the depths inside the workgroup (measured: 4.9 -> 5.35 TB/s at
        // B = 2^22, D = 51; below 2^21 the per-depth grid is the faster one)
        int ny = ceil_div(2048, nparts);
        if (ny < ceil_div(D, GAUSS_DGROUP_MAX)) ny = ceil_div(D, GAUSS_DGROUP_MAX);
        if (ny > D) ny = D;
        const int dgroup = ceil_div(D, ny);
        if (nt)
            hipLaunchKernelGGL(ll_gauss_depths_kernel<true>, dim3(nparts, ceil_div(D, dgroup)), dim3(256), 0, s, act, y,
                               log_std, w, w_is_logits, D, B, inv_batch, dact, ll, partials, nparts, dgroup);
        else
            hipLaunchKernelGGL(ll_gauss_depths_kernel<false>, dim3(nparts, ceil_div(D, dgroup)), dim3(256), 0, s, act, y,
                               log_std, w, w_is_logits, D, B, inv_batch, dact, ll, partials, nparts, dgroup);
    } else if (vec && nt)
        hipLaunchKernelGGL((ll_gauss_kernel<true, true>), dim3(nparts, D), dim3(256), 0, s, act, y, log_std, w, w_is_logits,
                           D, B, inv_batch, dact, ll, partials, nparts);
    else if (vec)
        hipLaunchKernelGGL(ll_gauss_kernel<true>, dim3(nparts, D), dim3(256), 0, s, act, y, log_std, w, w_is_logits, D,
                           B, inv_batch, dact, ll, partials, nparts);
    else
        hipLaunchKernelGGL(ll_gauss_kernel<false>, dim3(nparts, D), dim3(256), 0, s, act, y, log_std, w, w_is_logits,
                           D, B, inv_batch, dact, ll, partials, nparts);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

extern "C" int dun_ll_softmax(const float* act, const int64_t* y, const float* w, int w_is_logits, int D, int B, int C,
                              float inv_batch, float* dact, float* ll, double* partials, dun_stream_t stream) {
    DUN_REQUIRE(act && y && partials, "dun_ll_softmax: null pointer");
    DUN_REQUIRE(D > 0 && B > 0 && C > 0, "dun_ll_softmax: empty input D=%d B=%d C=%d", D, B, C);
    DUN_REQUIRE(D <= 65535, "dun_ll_softmax: D=%d exceeds grid.y", D);
    DUN_REQUIRE(dact == nullptr || w != nullptr, "dun_ll_softmax: dact requested without depth weights");
    if (C > 1024) {
        set_error("dun_ll_softmax: C=%d classes > 1024 not implemented", C);
        return DUN_ERR_UNSUPPORTED;
    }
    int km;
    const int G = softmax_group(C, &km);
    const int nparts = dun_ll_num_partials(B, C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lds_ok = ((long long)B * C) % 4 == 0 && aligned16(act) && (!dact || aligned16(dact));
    const int nt = (size_t)D * B * C * 4 * (dact ? 2 : 1) >= NT_MIN_BYTES ? 1 : 0;
    if (lds_ok && C <= 32) {
        const size_t lds = (size_t)ROWS_PER_BLOCK * C * sizeof(float);
#define DUN_RT(CM)                                                                                                   \
    do {                                                                                                             \
        if (nt)                                                                                                      \
            hipLaunchKernelGGL((ll_softmax_rowthread_kernel<CM, true>), dim3(nparts, D), dim3(256), lds, s, act, y, w, \
                               w_is_logits, D, B, C, inv_batch, dact, ll, partials, nparts);                         \
        else                                                                                                         \
            hipLaunchKernelGGL((ll_softmax_rowthread_kernel<CM, false>), dim3(nparts, D), dim3(256), lds, s, act, y, \
                               w, w_is_logits, D, B, C, inv_batch, dact, ll, partials, nparts);                      \
    } while (0)
        if (C <= 10)
            DUN_RT(10);
        else if (C <= 16)
            DUN_RT(16);
        else
            DUN_RT(32);
#undef DUN_RT
    } else if (lds_ok) {
        // G lanes per row with at most 32 logits per lane; rows per pass: <= 32 KiB, a multiple of 4 (16-byte aligned
        // passes) and of the rows one sweep of the lane groups covers
        int g = 2;
        while (g * 32 < C) g <<= 1;
        const int ngroups = 256 / g;
        const int quantum = ngroups > 4 ? ngroups : 4;
        int tile_rows = 8192 / C / quantum * quantum;
        if (tile_rows < quantum) tile_rows = quantum;
        if (tile_rows > ROWS_PER_BLOCK) tile_rows = ROWS_PER_BLOCK;
        const size_t lds = (size_t)tile_rows * C * sizeof(float);
#define DUN_RG(GG)                                                                                                  \
    do {                                                                                                            \
        if (nt)                                                                                                     \
            hipLaunchKernelGGL((ll_softmax_rowgroup_kernel<GG, 32, true>), dim3(nparts, D), dim3(256), lds, s, act, y, w, \
                               w_is_logits, D, B, C, inv_batch, dact, ll, partials, nparts, tile_rows);             \
        else                                                                                                        \
            hipLaunchKernelGGL((ll_softmax_rowgroup_kernel<GG, 32, false>), dim3(nparts, D), dim3(256), lds, s, act, y, \
                               w, w_is_logits, D, B, C, inv_batch, dact, ll, partials, nparts, tile_rows);          \
    } while (0)
        switch (g) {
            case 2: DUN_RG(2); break;
            case 4: DUN_RG(4); break;
            case 8: DUN_RG(8); break;
            case 16: DUN_RG(16); break;
            default: DUN_RG(32); break;
        }
#undef DUN_RG
    } else {
        DUN_SOFTMAX_DISPATCH(ll_softmax_kernel, G, km, dim3(nparts, D), s, act, y, w, w_is_logits, D, B, C, inv_batch,
                             dact, ll, partials, nparts);
    }
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

extern "C" int dun_ll_reduce(const double* partials, int D, int nparts, double* sums, dun_stream_t stream) {
    DUN_REQUIRE(partials && sums, "dun_ll_reduce: null pointer");
    DUN_REQUIRE(D > 0 && nparts > 0, "dun_ll_reduce: empty input");
    hipLaunchKernelGGL(reduce_rows_f64_kernel, dim3(D), dim3(64), 0, static_cast<hipStream_t>(stream), partials,
                       nparts, sums);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

extern "C" int dun_depth_finalize(const double* sums, int gaussian, const float* q_logits, const float* logprior,
                                  const float* prior_ll, const float* log_std, int D, double N_train, double batch,
                                  double beta, double eps, int objective, float* result, dun_stream_t stream) {
    DUN_REQUIRE(sums && logprior && result, "dun_depth_finalize: null pointer");
    DUN_REQUIRE(D > 0 && batch > 0 && N_train > 0, "dun_depth_finalize: bad sizes");
    DUN_REQUIRE(objective == DUN_OBJ_VI || objective == DUN_OBJ_MLL, "dun_depth_finalize: bad objective %d", objective);
    DUN_REQUIRE(objective != DUN_OBJ_VI || q_logits != nullptr, "dun_depth_finalize: VI needs q_logits");
    DUN_REQUIRE(!gaussian || log_std != nullptr, "dun_depth_finalize: gaussian needs log_std");
    hipLaunchKernelGGL(depth_finalize_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums, gaussian,
                       q_logits, logprior, prior_ll, log_std, D, N_train, batch, beta, eps, objective, result);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

constexpr int FINALIZE_FUSED_MAX_D = 4096;  // LDS sums of the one-launch form: 32 KiB

static int finalize_args_ok(const void* in, const float* q_logits, const float* logprior, const float* log_std,
                            int gaussian, int D, double N_train, double batch, int objective, const float* result,
                            const char* who) {
    DUN_REQUIRE(in && logprior && result, "%s: null pointer", who);
    DUN_REQUIRE(D > 0 && batch > 0 && N_train > 0, "%s: bad sizes", who);
    DUN_REQUIRE(objective == DUN_OBJ_VI || objective == DUN_OBJ_MLL, "%s: bad objective %d", who, objective);
    DUN_REQUIRE(objective != DUN_OBJ_VI || q_logits != nullptr, "%s: VI needs q_logits", who);
    DUN_REQUIRE(!gaussian || log_std != nullptr, "%s: gaussian needs log_std", who);
    return DUN_OK;
}

extern "C" int dun_depth_finalize_partials(const double* partials, int nparts, double* sums, int gaussian,
                                           const float* q_logits, const float* logprior, const float* prior_ll,
                                           const float* log_std, int D, double N_train, double batch, double beta,
                                           double eps, int objective, float* result, dun_stream_t stream) {
    int rc = finalize_args_ok(partials, q_logits, logprior, log_std, gaussian, D, N_train, batch, objective, result,
                              "dun_depth_finalize_partials");
    if (rc) return rc;
    DUN_REQUIRE(nparts > 0, "dun_depth_finalize_partials: nparts=%d", nparts);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D > FINALIZE_FUSED_MAX_D) {  // (no DUN has thousands of exits; kept correct rather than fast)
        DUN_REQUIRE(sums != nullptr, "dun_depth_finalize_partials: D=%d > %d needs the sums buffer", D, FINALIZE_FUSED_MAX_D);
        hipLaunchKernelGGL(reduce_rows_f64_kernel, dim3(D), dim3(64), 0, s, partials, nparts, sums);
        DUN_LAUNCH_CHECK();
        hipLaunchKernelGGL(depth_finalize_kernel, dim3(1), dim3(64), 0, s, sums, gaussian, q_logits, logprior, prior_ll,
                           log_std, D, N_train, batch, beta, eps, objective, result);
    } else {
        hipLaunchKernelGGL(depth_reduce_finalize_kernel, dim3(1), dim3(64 * FINALIZE_WAVES), (size_t)D * sizeof(double), s,
                           partials, nparts, sums, gaussian, q_logits, logprior, prior_ll, log_std, D, N_train, batch,
                           beta, eps, objective, result);
    }
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// ---- the whole VI objective of a Gaussian DUN in two launches ------------------------------------------------------
static int elbo_gauss_impl(const float* act, const float* y, const float* log_std, const float* q_logits,
                           const float* logprior, const float* prior_ll, int D, int B, double N_train, double beta,
                           double eps, float* dact, float* result, double* ws, dun_stream_t stream, const char* who) {
    DUN_REQUIRE(q_logits && logprior && result && ws, "%s: null pointer", who);
    int rc = dun_ll_gauss(act, y, log_std, q_logits, 1, D, B, (float)(1.0 / (double)B), dact, nullptr, ws, stream);
    if (rc) return rc;
    return dun_depth_finalize_partials(ws, dun_ll_num_partials(B, 1), nullptr, 1, q_logits, logprior, prior_ll, log_std,
                                       D, N_train, (double)B, beta, eps, DUN_OBJ_VI, result, stream);
}

extern "C" size_t dun_elbo_gauss_workspace_doubles(int D, int B) {
    if (D <= 0 || B <= 0) return 0;
    return (size_t)D * (size_t)dun_ll_num_partials(B, 1);
}

extern "C" int dun_elbo_gauss(const float* act, const float* y, const float* log_std, const float* q_logits,
                              const float* logprior, const float* prior_ll, int D, int B, double N_train, double beta,
                              double eps, float* dact, float* result, double* ws, size_t ws_doubles,
                              dun_stream_t stream) {
    DUN_REQUIRE(D > 0 && B > 0, "dun_elbo_gauss: empty input D=%d B=%d", D, B);
    DUN_REQUIRE(D <= FINALIZE_FUSED_MAX_D, "dun_elbo_gauss: D=%d > %d", D, FINALIZE_FUSED_MAX_D);
    DUN_REQUIRE(ws_doubles >= dun_elbo_gauss_workspace_doubles(D, B), "dun_elbo_gauss: workspace too small");
    return elbo_gauss_impl(act, y, log_std, q_logits, logprior, prior_ll, D, B, N_train, beta, eps, dact, result, ws,
                           stream, "dun_elbo_gauss");
}

// ---- the same objective for callers that cannot allocate (no counterpart in the reference) --------------------------
// Code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc cannot bring a workspace.
// dun_elbo_gauss_static is dun_elbo_gauss with the per-workgroup partial sums in a file-scope device array instead of
// the `ws` argument.  Every GPU the library runs on holds its own copy of the array (512 KiB, until the process
// ends) at its own address: the address is looked up once per device with hipGetSymbolAddress -- in host code the
// array's NAME is only a stand-in, never a device address -- and only that pointer is handed to kernels.
// Nothing in the array has to start at zero: the pass writes every [depth][workgroup] slot the finalisation reads.
// Calls on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race.
constexpr size_t STATIC_SCRATCH_DOUBLES = 65536;
static __device__ __align__(256) double scratch[STATIC_SCRATCH_DOUBLES];
constexpr int STATIC_MAX_DEVICES = 64;
static double* g_scratch_ptr[STATIC_MAX_DEVICES];  // device address of `scratch` on each device, 0 = not looked up yet
static std::mutex g_scratch_mutex;

// the current device's copy of the array; looked up (and checked) on the first call per device
static int static_scratch(double** out, const char* who) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("%s: hipGetDevice failed: %s", who, hipGetErrorString(e));
        return DUN_ERR_LAUNCH;
    }
    DUN_REQUIRE(dev >= 0 && dev < STATIC_MAX_DEVICES, "%s: device ordinal %d outside [0,%d)", who, dev,
                STATIC_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            set_error("%s: hipGetSymbolAddress(scratch) failed on device %d: %s", who, dev, hipGetErrorString(e));
            return DUN_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_scratch_ptr[dev];
    return DUN_OK;
}

extern "C" int dun_static_init(void) {
    double* p = nullptr;
    return static_scratch(&p, "dun_static_init");
}

extern "C" size_t dun_elbo_gauss_static_max_doubles(void) { return STATIC_SCRATCH_DOUBLES; }

extern "C" int dun_elbo_gauss_static(const float* act, const float* y, const float* log_std, const float* q_logits,
                                     const float* logprior, const float* prior_ll, int D, int B, double N_train,
                                     double beta, double eps, float* dact, float* result, dun_stream_t stream) {
    DUN_REQUIRE(D > 0 && B > 0, "dun_elbo_gauss_static: empty input D=%d B=%d", D, B);
    DUN_REQUIRE(D <= FINALIZE_FUSED_MAX_D, "dun_elbo_gauss_static: D=%d > %d", D, FINALIZE_FUSED_MAX_D);
    const size_t need = dun_elbo_gauss_workspace_doubles(D, B);
    DUN_REQUIRE(need <= STATIC_SCRATCH_DOUBLES,
                "dun_elbo_gauss_static: D=%d x B=%d needs %zu partial sums, the built-in scratch holds %zu "
                "(use dun_elbo_gauss with a workspace)", D, B, need, (size_t)STATIC_SCRATCH_DOUBLES);
    double* ws = nullptr;
    int rc = static_scratch(&ws, "dun_elbo_gauss_static");
    if (rc) return rc;
    return elbo_gauss_impl(act, y, log_std, q_logits, logprior, prior_ll, D, B, N_train, beta, eps, dact, result, ws,
                           stream, "dun_elbo_gauss_static");
}

static int predict_gauss_impl(const float* act, const float* q, int q_is_logits, int D, int n, float* mean, float* std,
                              const float* y, const float* log_std, double* partials, double* metrics, float* q_out,
                              const float* obj_src, double* obj_dst, double* done, dun_stream_t stream, const char* who) {
    DUN_REQUIRE(act && q && mean, "%s: null pointer", who);
    DUN_REQUIRE(D > 0 && n > 0, "%s: empty input", who);
    DUN_REQUIRE(metrics == nullptr || (y && log_std && partials), "%s: metrics need y, log_std, partials", who);
    DUN_REQUIRE((obj_src == nullptr) == (obj_dst == nullptr), "%s: objective_src and objective_dst go together", who);
    const int nparts = dun_predict_num_partials(n, 1);
    if (done != nullptr && nparts != 1) {
        set_error("%s: the completion word needs a one-workgroup predictor (n = %d)", who, n);
        return DUN_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one workgroup: its [2][1] partials ARE the metric sums, no reduction launch
    double* part = metrics ? (nparts == 1 ? metrics : partials) : nullptr;
    const bool vec = (n % 4 == 0) && aligned16(act);
    // log_std without targets: no metrics, `std` is the TOTAL predictive std (model spread + observation noise)
    const float* noise = (y == nullptr && std != nullptr) ? log_std : nullptr;
    if (vec && stream_nt((size_t)D * n * 4))
        hipLaunchKernelGGL((predict_gauss_kernel<true, true>), dim3(nparts), dim3(256), 0, s, act, q, D, n, mean, std, y,
                           log_std, part, nparts, q_is_logits, q_out, obj_src, obj_dst, done, noise);
    else if (vec)
        hipLaunchKernelGGL(predict_gauss_kernel<true>, dim3(nparts), dim3(256), 0, s, act, q, D, n, mean, std, y,
                           log_std, part, nparts, q_is_logits, q_out, obj_src, obj_dst, done, noise);
    else
        hipLaunchKernelGGL(predict_gauss_kernel<false>, dim3(nparts), dim3(256), 0, s, act, q, D, n, mean, std, y,
                           log_std, part, nparts, q_is_logits, q_out, obj_src, obj_dst, done, noise);
    DUN_LAUNCH_CHECK();
    if (metrics && nparts > 1) {
        hipLaunchKernelGGL(reduce_rows_f64_kernel, dim3(2), dim3(64), 0, s, partials, nparts, metrics);
        DUN_LAUNCH_CHECK();
    }
    return DUN_OK;
}

extern "C" int dun_predict_gauss(const float* act, const float* q, int D, int n, float* mean, float* std,
                                 const float* y, const float* log_std, double* partials, double* metrics,
                                 dun_stream_t stream) {
    return predict_gauss_impl(act, q, 0, D, n, mean, std, y, log_std, partials, metrics, nullptr, nullptr, nullptr, nullptr,
                              stream, "dun_predict_gauss");
}

extern "C" int dun_predict_gauss_step(const float* act, const float* q, int q_is_logits, int D, int n, float* mean,
                                      float* std, const float* y, const float* log_std, double* partials, double* metrics,
                                      float* q_out, const float* objective_src, double* objective_dst,
                                      dun_stream_t stream) {
    return predict_gauss_impl(act, q, q_is_logits, D, n, mean, std, y, log_std, partials, metrics, q_out, objective_src,
                              objective_dst, nullptr, stream, "dun_predict_gauss_step");
}

extern "C" int dun_predict_gauss_step_signal(const float* act, const float* q, int q_is_logits, int D, int n, float* mean,
                                             float* std, const float* y, const float* log_std, double* metrics,
                                             float* q_out, const float* objective_src, double* objective_dst, double* done,
                                             dun_stream_t stream) {
    DUN_REQUIRE(metrics != nullptr && done != nullptr, "dun_predict_gauss_step_signal: null metrics / done");
    return predict_gauss_impl(act, q, q_is_logits, D, n, mean, std, y, log_std, metrics, metrics, q_out, objective_src,
                              objective_dst, done, stream, "dun_predict_gauss_step_signal");
}

extern "C" int dun_predict_softmax(const float* act, const float* q, int D, int B, int C, float* probs,
                                   const int64_t* y, double* partials, double* metrics, dun_stream_t stream) {
    DUN_REQUIRE(act && q && probs, "dun_predict_softmax: null pointer");
    DUN_REQUIRE(D > 0 && B > 0 && C > 0, "dun_predict_softmax: empty input");
    DUN_REQUIRE(metrics == nullptr || (y && partials), "dun_predict_softmax: metrics need y and partials");
    if (C > 1024) {
        set_error("dun_predict_softmax: C=%d classes > 1024 not implemented", C);
        return DUN_ERR_UNSUPPORTED;
    }
    int km;
    const int G = softmax_group(C, &km);
    const bool lds_ok = ((long long)B * C) % 4 == 0 && aligned16(act) && aligned16(probs);
    const int rows_wg = predict_rows_per_wg(B, C, lds_ok);
    const int nparts = ceil_div(B, rows_wg);  // <= dun_predict_num_partials(B, C), which sizes the caller's buffer
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* part = metrics ? (nparts == 1 ? metrics : partials) : nullptr;
    if (lds_ok) {
        int g = 1;
        while (g * 32 < C) g <<= 1;
        const size_t lds = ((size_t)(256 / g) * C + 4) * sizeof(float);
        const bool nt = stream_nt((size_t)D * B * C * 4);
#define DUN_PS(GG, KK)                                                                                            \
    do {                                                                                                          \
        if (nt)                                                                                                   \
            hipLaunchKernelGGL((predict_softmax_lds_kernel<GG, KK, true>), dim3(nparts), dim3(256), lds, s, act, q, D, B, \
                               C, probs, y, part, nparts, rows_wg);                                               \
        else                                                                                                      \
            hipLaunchKernelGGL((predict_softmax_lds_kernel<GG, KK, false>), dim3(nparts), dim3(256), lds, s, act, q, D, \
                               B, C, probs, y, part, nparts, rows_wg);                                            \
    } while (0)
        if (C <= 10) DUN_PS(1, 10);
        else if (C <= 16) DUN_PS(1, 16);
        else if (g == 1) DUN_PS(1, 32);
        else if (g == 2) DUN_PS(2, 32);
        else if (g == 4) DUN_PS(4, 32);
        else if (g == 8) DUN_PS(8, 32);
        else if (g == 16) DUN_PS(16, 32);
        else DUN_PS(32, 32);
#undef DUN_PS
    } else {
        DUN_SOFTMAX_DISPATCH(predict_softmax_kernel, G, km, dim3(nparts), s, act, q, D, B, C, probs, y, part, nparts);
    }
    DUN_LAUNCH_CHECK();
    if (metrics && nparts > 1) {
        hipLaunchKernelGGL(reduce_rows_f64_kernel, dim3(3), dim3(64), 0, s, partials, nparts, metrics);
        DUN_LAUNCH_CHECK();
    }
    return DUN_OK;
}
