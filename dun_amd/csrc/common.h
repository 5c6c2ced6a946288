// Shared device/host helpers for the dun_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dun_hip.h"

namespace dun {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// ---- error plumbing: every extern "C" entry returns DUN_OK or an error code and records a message ----
void set_error(const char* fmt, ...);

#define DUN_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            dun::set_error(__VA_ARGS__);      \
            return DUN_ERR_INVALID;           \
        }                                     \
    } while (0)

#define DUN_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            dun::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                           hipGetErrorString(e__));                               \
            return DUN_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- streaming accesses -------------------------------------------------------------------------------------------
// Nontemporal 16-byte loads / stores for tensors that are read or written once and are larger than the 256 MiB Infinity
// Cache.  Measured on MI355X: Gaussian likelihood pass +8.5 % at 1.7 GB (5.35 -> 5.80 TB/s) but -16 % at 0.1 GB (the
// default policy keeps such a stream cache resident); shared head over 27 GB 5.8 -> 6.45 TB/s; BatchNorm passes of
// the wide trunk -0.6 % step time.  Hosts pick by size (stream_nt).
template <bool NT>
__device__ __forceinline__ f32x4 ld4_stream(const float* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st4_stream(float* p, f32x4 v) {
    if (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else
        *reinterpret_cast<f32x4*>(p) = v;
}
// the same choice as a (wave-uniform) run-time flag, for kernels with many instantiations already
__device__ __forceinline__ f32x4 ld4_stream(const float* p, int nt) { return nt ? ld4_stream<true>(p) : ld4_stream<false>(p); }
__device__ __forceinline__ void st4_stream(float* p, f32x4 v, int nt) {
    if (nt) st4_stream<true>(p, v); else st4_stream<false>(p, v);
}
constexpr size_t NT_MIN_BYTES = (size_t)320 << 20;
static inline bool stream_nt(size_t bytes) { return bytes >= NT_MIN_BYTES; }

// ---- wavefront (64 lanes) and workgroup reductions ----
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// reduce inside a power-of-two group of G consecutive lanes (G <= 64)
template <int G, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int G, typename T>
__device__ __forceinline__ T group_max(T v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---- all-lanes wavefront reductions on the DPP path (no LDS crossbar) ----
// __shfl_xor is a ds_bpermute: six dependent LDS-crossbar round trips per 64-lane reduction.  Here four DPP steps
// (xor 1, xor 2 inside quads, mirror inside half rows, mirror inside rows) leave every lane of a 16-lane row with the
// row total, and the four row totals are combined through v_readlane (scalar broadcast).  Every lane must be active.
// Fixed order: ((r0 + r1) + (r2 + r3)) over rows, butterfly inside a row.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

template <typename T>
__device__ __forceinline__ T wave_allsum(T v) {
    v += dpp_move<DPP_QUAD_XOR1>(v);
    v += dpp_move<DPP_QUAD_XOR2>(v);
    v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move<DPP_ROW_MIRROR>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_allmax(float v) {
    v = fmaxf(v, dpp_move<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_move<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_move<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_move<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

__device__ __forceinline__ double wave_allmax(double v) {
    v = fmax(v, dpp_move<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_move<DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_move<DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_move<DPP_ROW_MIRROR>(v));
    return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}

constexpr float LOG_SQRT_2PI_F = 0.91893853320467274178f;
constexpr double LOG_SQRT_2PI_D = 0.91893853320467274178;

// weight of depth d: w[d], or softmax(w)[d] when the caller hands over logits (q = softmax(q_logits)).
// Every wavefront works it out for itself -- lanes stride over the D logits, max and sum by DPP reductions -- so there
// is no LDS slot, no barrier and no serial scalar loop in front of the streaming part of a kernel (the first version
// walked the D logits in a scalar loop, summed in wave 0 and published through LDS behind a barrier: of the 9.6 us the
// Gaussian pass took at D = 51, B = 65536 that chain was the longest dependent piece).  fp32 throughout, as F.softmax.
// Call with all 64 lanes active.
__device__ __forceinline__ float depth_weight(const float* __restrict__ w, int w_is_logits, int D, int d) {
    if (w == nullptr) return 0.f;
    if (!w_is_logits) return w[d];
    const int lane = threadIdx.x & 63;
    float mx = -INFINITY;
    for (int i = lane; i < D; i += 64) mx = fmaxf(mx, w[i]);
    mx = wave_allmax(mx);
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += expf(w[i] - mx);
    s = wave_allsum(s);
    return expf(w[d] - mx) / s;
}

// sum over a workgroup of NT threads (NT multiple of 64, <= 1024); result valid in thread 0.
// `red` is caller-provided LDS of at least NT/64 elements; the call contains two barriers.
template <int NT, typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
    v = wave_allsum(v);  // (DPP path: every thread of the workgroup calls this, so all lanes are active)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += red[i];
    }
    return r;
}

}  // namespace dun
