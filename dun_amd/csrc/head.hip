// Shared output head over all depths as one batched contraction (gfx950).
//
// The reference applies ONE Linear(width, O) to the hidden state of every depth (FC) or to the pooled
// features of every exit (image DUNs).  Here the D*B rows are one HBM stream: a wavefront takes a row,
// lanes take 16-byte column chunks, O accumulators per lane, one butterfly reduction per output.  With O
// small (1 for regression, 10 for MNIST/CIFAR) this is bandwidth bound: 4*K bytes per row read once.
// Backward streams the same rows once more: thread-owned column chunks accumulate dw over slabs of up to 1024 rows
// (fixed-order LDS combine, then an fp64 reduction over slabs: reproducible), db rides along, and the
// optional dH = dact w is written in the same pass for callers that need it (the image trunk's autograd).
#include "common.h"

namespace dun {

constexpr int HSLAB_MAX = 1024;  // rows per dw partial (upper bound; small inputs take shorter slabs)
constexpr int HCX = 64, HRY = 4;

template <int OC, bool VEC, bool NT = false>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ H, const float* __restrict__ w,
                                                       const float* __restrict__ b, int rows, int K, int O,
                                                       float* __restrict__ act) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int row = wave; row < rows; row += nwaves) {
        const float* h = H + (size_t)row * K;
        for (int o0 = 0; o0 < O; o0 += OC) {
            float acc[OC];
#pragma unroll
            for (int j = 0; j < OC; ++j) acc[j] = 0.f;
            if (VEC) {
                for (int k = lane * 4; k < K; k += 256) {
                    const f32x4 hv = ld4_stream<NT>(h + k);
#pragma unroll
                    for (int j = 0; j < OC; ++j) {
                        if (o0 + j < O) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(o0 + j) * K + k);
                            acc[j] += (hv[0] * wv[0] + hv[1] * wv[1]) + (hv[2] * wv[2] + hv[3] * wv[3]);
                        }
                    }
                }
            } else {
                for (int k = lane; k < K; k += 64) {
                    const float hv = h[k];
#pragma unroll
                    for (int j = 0; j < OC; ++j)
                        if (o0 + j < O) acc[j] += hv * w[(size_t)(o0 + j) * K + k];
                }
            }
#pragma unroll
            for (int j = 0; j < OC; ++j) {
                const float t = wave_sum(acc[j]);
                if (lane == 0 && o0 + j < O) act[(size_t)row * O + o0 + j] = t + b[o0 + j];
            }
        }
    }
}

// O > 4 (10-class image DUNs): the kernel above issues O weight loads per 16 bytes of H and is bound by load issue,
// not by HBM.  Here a wavefront takes R consecutive rows, so a weight chunk is loaded once per R rows, and the R*O
// lane-partials are reduced together by a halving butterfly: at the xor-32/16/8/4 steps a lane keeps one value of
// each pair and ships the other, so R*O values cost ~R*O shuffles instead of 6*R*O.  After four halvings lane l holds
// value 16*i + 8*b2 + 4*b3 + 2*b4 + b5 (bk = bit k of l) in slot i; value index = r*O + j is the output offset.
template <int O, int R>
__global__ __launch_bounds__(256) void head_fwd_rows_kernel(const float* __restrict__ H, const float* __restrict__ w,
                                                            const float* __restrict__ b, int rows, int K,
                                                            float* __restrict__ act) {
    static_assert((R * O) % 16 == 0, "four halving steps need R*O divisible by 16");
    constexpr int NV = R * O;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int row0 = wave * R; row0 < rows; row0 += nwaves * R) {
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            f32x4 hv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                hv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row0 + r < rows) hv[r] = *reinterpret_cast<const f32x4*>(H + (size_t)(row0 + r) * K + k);
            }
#pragma unroll
            for (int j = 0; j < O; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)j * K + k);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    v[r * O + j] += (hv[r][0] * wv[0] + hv[r][1] * wv[1]) + (hv[r][2] * wv[2] + hv[r][3] * wv[3]);
            }
        }
#define DUN_HALVE(N, OFF)                                            \
    _Pragma("unroll") for (int i = 0; i < (N) / 2; ++i) {            \
        const bool up = (lane & (OFF)) != 0;                         \
        const float keep = up ? v[2 * i + 1] : v[2 * i];             \
        const float send = up ? v[2 * i] : v[2 * i + 1];             \
        v[i] = keep + __shfl_xor(send, (OFF), 64);                   \
    }
        DUN_HALVE(NV, 32)
        DUN_HALVE(NV / 2, 16)
        DUN_HALVE(NV / 4, 8)
        DUN_HALVE(NV / 8, 4)
#undef DUN_HALVE
        const int sub = ((lane >> 2) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
#pragma unroll
        for (int i = 0; i < NV / 16; ++i) {
            float t = v[i];
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 1, 64);
            const int idx = 16 * i + sub;  // = r * O + j
            if ((lane & 3) == 0 && row0 + idx / O < rows) act[(size_t)row0 * O + idx] = t + b[idx % O];
        }
    }
}

// One workgroup: 64 column-chunks x 4 row lanes over a slab of slab_rows rows.  o0: first output of this launch.
template <int OC, int VW, bool NT = false>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ H, const float* __restrict__ w,
                                                       const float* __restrict__ dact, int rows, int K, int O, int o0,
                                                       float* __restrict__ part_dw, float* __restrict__ part_db,
                                                       float* __restrict__ dH, int slab_rows) {
    __shared__ float red[HRY][HCX * VW];
    // ty is the wavefront index; with several outputs, telling the compiler so turns the O per-row dact loads into
    // scalar loads (one output: the single vector load is cheaper than the scalar-load latency)
    const int tx = threadIdx.x & (HCX - 1);
    const int ty = OC > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    const int c = (blockIdx.x * HCX + tx) * VW;
    const int slab = blockIdx.y;
    const bool live = c < K;
    float acc[OC][VW], wv[OC][VW], sdb[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) {
        sdb[j] = 0.f;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            acc[j][e] = 0.f;
            wv[j][e] = (dH && live && o0 + j < O) ? w[(size_t)(o0 + j) * K + c + e] : 0.f;
        }
    }
    const int r1 = min(rows, (slab + 1) * slab_rows);
    if (live) {
        for (int r = slab * slab_rows + ty; r < r1; r += HRY) {
            float hv[VW];
            if (VW == 4) {
                const f32x4 t = ld4_stream<NT>(H + (size_t)r * K + c);
                hv[0] = t[0]; hv[1] = t[1]; hv[2] = t[2]; hv[3] = t[3];
            } else {
                hv[0] = H[(size_t)r * K + c];
            }
            float dh[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) dh[e] = 0.f;
#pragma unroll
            for (int j = 0; j < OC; ++j) {
                if (o0 + j < O) {
                    const float d = dact[(size_t)r * O + o0 + j];
                    sdb[j] += d;
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        acc[j][e] += d * hv[e];
                        dh[e] += d * wv[j][e];
                    }
                }
            }
            if (dH) {
                float* o = dH + (size_t)r * K + c;
                if (VW == 4) {
                    f32x4 t = {dh[0], dh[1], dh[2], dh[3]};
                    if (o0 != 0) t += *reinterpret_cast<const f32x4*>(o);
                    *reinterpret_cast<f32x4*>(o) = t;
                } else {
                    o[0] = (o0 == 0 ? 0.f : o[0]) + dh[0];
                }
            }
        }
    }
    // fixed-order combine of the 4 row lanes, one output at a time
#pragma unroll
    for (int j = 0; j < OC; ++j) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VW; ++e) red[ty][tx * VW + e] = acc[j][e];
        __syncthreads();
        if (ty == 0 && live && o0 + j < O) {
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const int i = tx * VW + e;
                part_dw[((size_t)slab * O + o0 + j) * K + c + e] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
            }
        }
    }
    // db: every column thread saw the same dact values; the tx == 0 threads of the first column block own it
    if (blockIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < OC; ++j) {
            __syncthreads();
            if (tx == 0) red[ty][0] = sdb[j];
            __syncthreads();
            if (threadIdx.x == 0 && o0 + j < O)
                part_db[(size_t)slab * O + o0 + j] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        }
    }
}

// out[c] = sum_p partials[p][c] in fp64, fixed order (64 columns x 4 partial lanes per workgroup);
// the weight- and bias-gradient partials in ONE launch: workgroups [0, nb_w) reduce dw, the rest db (same arithmetic)
__global__ __launch_bounds__(256) void head_reduce2_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                           int P, size_t n_w, size_t n_b, int nb_w,
                                                           float* __restrict__ out_w, float* __restrict__ out_b) {
    __shared__ double red[HRY][HCX];
    const int tx = threadIdx.x & (HCX - 1), ty = threadIdx.x >> 6;
    const bool is_w = (int)blockIdx.x < nb_w;
    const float* partials = is_w ? part_w : part_b;
    const size_t n = is_w ? n_w : n_b;
    float* out = is_w ? out_w : out_b;
    const size_t c = (size_t)(is_w ? blockIdx.x : blockIdx.x - nb_w) * HCX + tx;
    double s = 0.0;
    if (c < n)
        for (int p = ty; p < P; p += HRY) s += (double)partials[(size_t)p * n + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < n) out[c] = (float)((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
}

static int pick_oc(int O) {
    if (O <= 1) return 1;
    if (O <= 2) return 2;
    if (O <= 4) return 4;
    if (O <= 8) return 8;
    if (O <= 10) return 10;  // the 10-class heads of the image DUNs
    return 16;
}

int head_forward(const float* H, const float* w, const float* b, int rows, int K, int O, float* act, hipStream_t s) {
    DUN_REQUIRE(H && w && b && act, "dun_head_forward: null pointer");
    DUN_REQUIRE(rows > 0 && K > 0 && O > 0, "dun_head_forward: empty input rows=%d K=%d O=%d", rows, K, O);
    const bool vec = K % 4 == 0 && aligned16(H) && aligned16(w);
    if (vec && O == 10) {  // MNIST / CIFAR-10 heads of the image DUNs
        int g = ceil_div(rows, 4 * 8);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL((head_fwd_rows_kernel<10, 8>), dim3(g), dim3(256), 0, s, H, w, b, rows, K, act);
        DUN_LAUNCH_CHECK();
        return DUN_OK;
    }
    int grid = ceil_div(rows, 4);
    if (grid > 4096) grid = 4096;
    const int nt = stream_nt((size_t)rows * K * 4) ? 1 : 0;  // H is read once: beyond the Infinity Cache, stream it
#define HF(OC)                                                                                              \
    do {                                                                                                    \
        if (vec && nt) hipLaunchKernelGGL((head_fwd_kernel<OC, true, true>), dim3(grid), dim3(256), 0, s, H, w, b, rows, K, O, act); \
        else if (vec) hipLaunchKernelGGL((head_fwd_kernel<OC, true>), dim3(grid), dim3(256), 0, s, H, w, b, rows, K, O, act); \
        else hipLaunchKernelGGL((head_fwd_kernel<OC, false>), dim3(grid), dim3(256), 0, s, H, w, b, rows, K, O, act); \
    } while (0)
    switch (pick_oc(O)) {
        case 1: HF(1); break;
        case 2: HF(2); break;
        case 4: HF(4); break;
        case 8: HF(8); break;
        case 10: HF(10); break;
        default: HF(16); break;
    }
#undef HF
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// rows per slab: about 1024 workgroups over the (column block, slab) grid, 64..HSLAB_MAX rows each
static int head_slab_rows(int rows, int K) {
    const int colblocks = ceil_div(K, HCX * 4);
    const int want = 1024 / colblocks > 0 ? 1024 / colblocks : 1;
    int sr = ceil_div(rows, want);
    sr = (sr + 7) & ~7;
    if (sr < 64) sr = 64;
    if (sr > HSLAB_MAX) sr = HSLAB_MAX;
    return sr;
}

size_t head_backward_ws(int rows, int K, int O) {
    const size_t nslab = (size_t)ceil_div(rows, head_slab_rows(rows, K));
    return nslab * (size_t)O * (size_t)K + nslab * (size_t)O + 64;
}

int head_backward(const float* H, const float* w, const float* dact, int rows, int K, int O, float* dw, float* db,
                  float* dH, float* ws, size_t ws_floats, hipStream_t s) {
    DUN_REQUIRE(H && w && dact && dw && db && ws, "dun_head_backward: null pointer");
    DUN_REQUIRE(rows > 0 && K > 0 && O > 0, "dun_head_backward: empty input rows=%d K=%d O=%d", rows, K, O);
    DUN_REQUIRE(ws_floats >= head_backward_ws(rows, K, O), "dun_head_backward: workspace too small");
    const int slab_rows = head_slab_rows(rows, K);
    const int nslab = ceil_div(rows, slab_rows);
    float* part_dw = ws;
    float* part_db = ws + (size_t)nslab * O * K;
    const bool vec = K % 4 == 0 && aligned16(H) && aligned16(w) && (!dH || aligned16(dH));
    const int oc = pick_oc(O);
    // H streamed once per launch; with several launches (O > 16) the default policy lets later ones hit the cache
    const int nt = (stream_nt((size_t)rows * K * 4) && O <= oc) ? 1 : 0;
    for (int o0 = 0; o0 < O; o0 += oc) {
#define HB(OC)                                                                                                   \
    do {                                                                                                         \
        if (vec && nt) hipLaunchKernelGGL((head_bwd_kernel<OC, 4, true>), dim3(ceil_div(K, HCX * 4), nslab), dim3(256), 0, s, \
                                          H, w, dact, rows, K, O, o0, part_dw, part_db, dH, slab_rows);          \
        else if (vec) hipLaunchKernelGGL((head_bwd_kernel<OC, 4>), dim3(ceil_div(K, HCX * 4), nslab), dim3(256), 0, s, H, w, \
                                         dact, rows, K, O, o0, part_dw, part_db, dH, slab_rows);                 \
        else hipLaunchKernelGGL((head_bwd_kernel<OC, 1>), dim3(ceil_div(K, HCX), nslab), dim3(256), 0, s, H, w, dact,   \
                                rows, K, O, o0, part_dw, part_db, dH, slab_rows);                                \
    } while (0)
        switch (oc) {
            case 1: HB(1); break;
            case 2: HB(2); break;
            case 4: HB(4); break;
            case 8: HB(8); break;
            case 10: HB(10); break;
            default: HB(16); break;
        }
#undef HB
        DUN_LAUNCH_CHECK();
    }
    const size_t n = (size_t)O * K;
    const int nb_w = (int)((n + HCX - 1) / HCX), nb_b = (O + HCX - 1) / HCX;
    hipLaunchKernelGGL(head_reduce2_kernel, dim3((unsigned)(nb_w + nb_b)), dim3(256), 0, s, part_dw, part_db, nslab, n,
                       (size_t)O, nb_w, dw, db);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

}  // namespace dun

extern "C" int dun_head_forward(const float* H, const float* w, const float* b, int rows, int K, int O, float* act,
                                dun_stream_t stream) {
    return dun::head_forward(H, w, b, rows, K, O, act, static_cast<hipStream_t>(stream));
}

extern "C" size_t dun_head_backward_workspace_floats(int rows, int K, int O) {
    return dun::head_backward_ws(rows, K, O);
}

extern "C" int dun_head_backward(const float* H, const float* w, const float* dact, int rows, int K, int O, float* dw,
                                 float* db, float* dH, float* ws, size_t ws_floats, dun_stream_t stream) {
    return dun::head_backward(H, w, dact, rows, K, O, dw, db, dH, ws, ws_floats, static_cast<hipStream_t>(stream));
}
