// fp32 MFMA GEMM for the DUN residual-MLP trunk (gfx950).
//
// C[M,N] = epi(sum_k A(m,k) B(k,n)) on v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), so the
// trunk keeps the reference's fp32 numerics (north-star: 1e-5 relative) while running on the matrix cores.
// Workgroup = 256 threads = 4 waves (2x2), tile 128x128x32, each wave a 64x64 sub-tile = 2x2 MFMA 32x32
// accumulators.  Operand tiles are staged global -> registers -> LDS, double buffered, one barrier per K
// tile; the register prefetch of tile t+1 is in flight while tile t is multiplied.
//
// Operands keep their global orientation in LDS, so no transposes are ever materialised:
//   K-contiguous operand (x / h / dU rows, Linear weight rows): LDS [128][32+4] floats; a lane reads its
//     four k values with one ds_read_b128 (144-B row stride: conflict-free for the 16-lane b128 groups);
//   M/N-contiguous operand (W in dX = dU W, dU^T and H in dW = dU^T H): LDS [32][128]; a lane reads one
//     float per MFMA with ds_read_b32, 32 consecutive lanes on 32 consecutive banks.
// Both forms feed the MFMA with k(step s=(t,e), lane half h) = 8t + 4h + e, so any pairing is consistent.
//
// Epilogues (all optional, fused): + bias[n], ReLU, + R[m,n] (residual / running gradient), + U[m,:] V[:,n]
// (the head's rank-O gradient contribution), and per-128-row-tile column sums of C and C^2 (BatchNorm
// batch statistics, reduced later in fp64).
#include <stdlib.h>

#include "common.h"

namespace dun {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;

struct GemmP {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias;
    int relu;
    const float* R;
    int ldr;
    const float* U;
    const float* V;
    int O, ldv;
    double* colstats;
    int gridM, gridN;
};

// ---- global -> register staging of one operand tile: 4 x float4 per thread ----
// TB = tile edge (128: four float4 per thread, 64: two)
template <bool KC, bool VEC, int TB>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int r0, int rows, int k0, int K,
                                          int tid, f32x4 (&reg)[TB / 32]) {
    constexpr int NR = TB / 32;
    if (KC) {
        const int c4 = (tid & 7) * 4, rr = tid >> 3;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = r0 + rr + 32 * r, k = k0 + c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) {
                const float* src = P + (size_t)row * ld + k;
                if (VEC) {
                    if (k < K) v = *reinterpret_cast<const f32x4*>(src);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < K) v[e] = src[e];
                }
            }
            reg[r] = v;
        }
    } else {
        constexpr int TPR = TB / 4, KSTEP = 256 / TPR;  // threads per k-row, k-rows per pass
        const int c4 = (tid & (TPR - 1)) * 4, kk = tid / TPR;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int k = k0 + kk + KSTEP * r, row = r0 + c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                const float* src = P + (size_t)k * ld + row;
                if (VEC) {
                    if (row < rows) v = *reinterpret_cast<const f32x4*>(src);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row + e < rows) v[e] = src[e];
                }
            }
            reg[r] = v;
        }
    }
}

template <bool KC, int TB>
__device__ __forceinline__ void store_tile(float* lds, int tid, const f32x4 (&reg)[TB / 32]) {
    constexpr int NR = TB / 32;
    if (KC) {
        const int c4 = (tid & 7) * 4, rr = tid >> 3;
#pragma unroll
        for (int r = 0; r < NR; ++r) *reinterpret_cast<f32x4*>(lds + (rr + 32 * r) * LDK + c4) = reg[r];
    } else {
        constexpr int TPR = TB / 4, KSTEP = 256 / TPR;
        const int c4 = (tid & (TPR - 1)) * 4, kk = tid / TPR;
#pragma unroll
        for (int r = 0; r < NR; ++r) *reinterpret_cast<f32x4*>(lds + (kk + KSTEP * r) * TB + c4) = reg[r];
    }
}

// fragment for MFMA steps (t, e=0..3): element e of lane (i, h) is operand[row i][k = 8t + 4h + e]
template <bool KC, int TB>
__device__ __forceinline__ void read_frag(const float* lds, int row, int t, int h, float (&f)[4]) {
    if (KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + row * LDK + 8 * t + 4 * h);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = lds[(8 * t + 4 * h + e) * TB + row];
    }
}

// ---- epilogue shared by both kernels.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// FULL: the tile lies completely inside C (no bounds checks).
template <bool FULL, int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[MI][MI], float* smem, int m0, int n0, int bm,
                                              int tid, int wave, int wr, int wc, int li, int h) {
    constexpr int WT = 32 * MI;  // wave sub-tile edge
    // BatchNorm statistics are accumulated in fp64 from the first element on: sum z^2 - (sum z)^2/n in fp32
    // loses the variance of a nearly constant column (narrow inputs make many of those), and the reference's
    // batch_norm uses a cancellation-free update.  64 dfma per lane per tile: free next to the K loop.
    double s1[MI], s2[MI];
#pragma unroll
    for (int ni = 0; ni < MI; ++ni) s1[ni] = s2[ni] = 0.0;
#pragma unroll
    for (int ni = 0; ni < MI; ++ni) {
        const int col = n0 + wc * WT + ni * 32 + li;
        const bool col_ok = FULL || col < p.N;
        const float bv = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * WT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (FULL || (row < p.M && col_ok)) {
                    float v = acc[mi][ni][r] + bv;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    if (p.R != nullptr) v += p.R[(size_t)row * p.ldr + col];
                    if (p.U != nullptr) {
                        for (int o = 0; o < p.O; ++o) v += p.U[(size_t)row * p.O + o] * p.V[(size_t)o * p.ldv + col];
                    }
                    p.C[(size_t)row * p.ldc + col] = v;
                    if (p.colstats != nullptr) {
                        const double dv = (double)v;
                        s1[ni] += dv;
                        s2[ni] += dv * dv;
                    }
                }
            }
        }
    }
    if (p.colstats != nullptr) {
        double* red = reinterpret_cast<double*>(smem);  // all waves are past the last barrier: LDS is free
#pragma unroll
        for (int ni = 0; ni < MI; ++ni) {
            s1[ni] += __shfl_xor(s1[ni], 32, 64);
            s2[ni] += __shfl_xor(s2[ni], 32, 64);
            if (h == 0) {
                red[((wave * MI + ni) * 32 + li) * 2 + 0] = s1[ni];
                red[((wave * MI + ni) * 32 + li) * 2 + 1] = s2[ni];
            }
        }
        __syncthreads();
        if (tid < 2 * WT) {
            const int wc2 = tid / WT, ni2 = (tid / 32) % MI, l2 = tid & 31, col = n0 + tid;
            if (FULL || col < p.N) {
                double t1 = 0.0, t2 = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const int wv = w2 * 2 + wc2;
                    t1 += red[((wv * MI + ni2) * 32 + l2) * 2 + 0];
                    t2 += red[((wv * MI + ni2) * 32 + l2) * 2 + 1];
                }
                p.colstats[(size_t)bm * p.N + col] = t1;
                p.colstats[(size_t)(p.gridM + bm) * p.N + col] = t2;
            }
        }
    }
}

// MI = MFMA tiles per wave and dimension: 2 -> 128x128 workgroup tile, 1 -> 64x64 (problems too small to fill the
// chip with 128-tiles: four times the workgroups, a quarter of the serial MFMA chain per workgroup)
template <bool A_KC, bool B_KC, bool VEC, int MI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmP p) {
    constexpr int TB = 64 * MI, WT = 32 * MI, NR = TB / 32;
    constexpr int TILE = TB * LDK;  // >= BK * TB
    // MI = 2: 72 KiB static LDS, [buffer][A tile | B tile]; two workgroups per CU fit in the 160 KiB
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE];

    // XCD-aware remap (bijective): the 8 XCDs take contiguous ranges of tiles so that the tiles sharing an
    // A row-panel (all bn of one bm) hit the same L2.
    const int nwg = p.gridM * p.gridN;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.gridN, bn = bid % p.gridN;
    const int m0 = bm * TB, n0 = bn * TB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, h = lane >> 5;

    f32x16 acc[MI][MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < MI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    f32x4 ra[NR], rb[NR];
    const int nk = (p.K + BK - 1) / BK;
    load_tile<A_KC, VEC, TB>(p.A, p.lda, m0, p.M, 0, p.K, tid, ra);
    load_tile<B_KC, VEC, TB>(p.B, p.ldb, n0, p.N, 0, p.K, tid, rb);
    store_tile<A_KC, TB>(smem, tid, ra);
    store_tile<B_KC, TB>(smem + TILE, tid, rb);
    __syncthreads();

    // Fragment registers are double buffered over the four 8-deep k groups of a tile: the ds_reads of group t+1
    // (and, in group 0, the global loads of the next tile; in group 2, its ds_writes) are issued in front of the
    // MFMAs of group t, so they execute under the matrix pipe instead of in front of it.
    float fa[2][MI][4], fb[2][MI][4];
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        const float* a_s = smem + cur * (2 * TILE);
        const float* b_s = a_s + TILE;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) read_frag<A_KC, TB>(a_s, wr * WT + mi * 32 + li, 0, h, fa[0][mi]);
#pragma unroll
        for (int ni = 0; ni < MI; ++ni) read_frag<B_KC, TB>(b_s, wc * WT + ni * 32 + li, 0, h, fb[0][ni]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < 3) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    read_frag<A_KC, TB>(a_s, wr * WT + mi * 32 + li, t + 1, h, fa[(t + 1) & 1][mi]);
#pragma unroll
                for (int ni = 0; ni < MI; ++ni)
                    read_frag<B_KC, TB>(b_s, wc * WT + ni * 32 + li, t + 1, h, fb[(t + 1) & 1][ni]);
            }
            if (t == 0 && more) {
                load_tile<A_KC, VEC, TB>(p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid, ra);
                load_tile<B_KC, VEC, TB>(p.B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, tid, rb);
            }
            if (t == 2 && more) {
                // Safe: every wave has passed the barrier that ended iteration kt-1, nobody reads buffer cur^1.
                float* nxt = smem + (cur ^ 1) * (2 * TILE);
                store_tile<A_KC, TB>(nxt, tid, ra);
                store_tile<B_KC, TB>(nxt + TILE, tid, rb);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < MI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][mi][e], fb[t & 1][ni][e],
                                                                           acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    gemm_epilogue<false, MI>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
}

// ---------------------------------------------------------------------------------------------------------
// Fast path for tile-aligned problems (M%128 == N%128 == K%32 == 0, 16-byte aligned rows): operand tiles go
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), no VGPR round trip and no ds_write.  A wave-instruction
// writes 1 KiB of LDS linearly (wave-uniform base + lane*16), so the images are unpadded:
//   K-contiguous tile  [128 rows][32 k] = 128-B rows.  Bank-conflict-free ds_read_b128 comes from an XOR
//     swizzle of the 16-B chunk index with (row>>1)&7, applied to the per-lane SOURCE address on the way in
//     and to the read address on the way out (the same involution on both sides);
//   M/N-contiguous tile [32 k][128] = 512-B rows, read with ds_read_b32 (32 lanes on 32 consecutive banks).
// Two 32 KiB buffers; the DMA of tile t+1 is issued before the 64 MFMAs of tile t and retired by the
// vmcnt(0) that __syncthreads() carries at the end of the tile.  64 KiB LDS -> two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------
constexpr int DMA_TILE = BM * BK;  // 4096 floats = 16 KiB per operand

template <bool KC>
__device__ __forceinline__ void dma_tile(const float* __restrict__ P, int ld, int r0, int k0, float* lds, int wave,
                                         int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;  // 1 KiB piece of the 16 KiB image
        const float* src;
        if (KC) {
            const int row = piece * 8 + (lane >> 3), pc = lane & 7;
            const int c = pc ^ ((row >> 1) & 7);
            src = P + (size_t)(r0 + row) * ld + k0 + c * 4;
        } else {
            const int k = piece * 2 + (lane >> 5), col = (lane & 31) * 4;
            src = P + (size_t)(k0 + k) * ld + r0 + col;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + piece * 256), 16, 0, 0);
    }
}

template <bool KC>
__device__ __forceinline__ void read_frag_dma(const float* lds, int row, int t, int h, float (&f)[4]) {
    if (KC) {
        const int pc = (2 * t + h) ^ ((row >> 1) & 7);
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + row * BK + pc * 4);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = lds[(8 * t + 4 * h + e) * BM + row];
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * DMA_TILE];  // [buffer][A image | B image]

    const int nwg = p.gridM * p.gridN;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.gridN, bn = bid % p.gridN;
    const int m0 = bm * BM, n0 = bn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases live in M0
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = p.K / BK;
    dma_tile<A_KC>(p.A, p.lda, m0, 0, smem, wave, lane);
    dma_tile<B_KC>(p.B, p.ldb, n0, 0, smem + DMA_TILE, wave, lane);
    __syncthreads();  // carries the vmcnt(0) that retires the DMA

    float fa[2][2][4], fb[2][2][4];
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* a_s = smem + cur * (2 * DMA_TILE);
        const float* b_s = a_s + DMA_TILE;
        if (kt + 1 < nk) {
            // buffer cur^1 was last read in iteration kt-1, which every wave left through the barrier below
            float* nxt = smem + (cur ^ 1) * (2 * DMA_TILE);
            dma_tile<A_KC>(p.A, p.lda, m0, (kt + 1) * BK, nxt, wave, lane);
            dma_tile<B_KC>(p.B, p.ldb, n0, (kt + 1) * BK, nxt + DMA_TILE, wave, lane);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) read_frag_dma<A_KC>(a_s, wr * 64 + mi * 32 + li, 0, h, fa[0][mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) read_frag_dma<B_KC>(b_s, wc * 64 + ni * 32 + li, 0, h, fb[0][ni]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < 3) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    read_frag_dma<A_KC>(a_s, wr * 64 + mi * 32 + li, t + 1, h, fa[(t + 1) & 1][mi]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    read_frag_dma<B_KC>(b_s, wc * 64 + ni * 32 + li, t + 1, h, fb[(t + 1) & 1][ni]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][mi][e], fb[t & 1][ni][e],
                                                                           acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_epilogue<true, 2>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
}

template <bool A_KC, bool B_KC>
static int launch_gemm(const GemmP& p, bool vec, bool dma, bool small, hipStream_t s) {
    const int grid = p.gridM * p.gridN;
    if (dma)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<A_KC, B_KC>), dim3(grid), dim3(256), 0, s, p);
    else if (small && vec)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, 1>), dim3(grid), dim3(256), 0, s, p);
    else if (small)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, 1>), dim3(grid), dim3(256), 0, s, p);
    else if (vec)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, 2>), dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, 2>), dim3(grid), dim3(256), 0, s, p);
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// test hook: DUN_GEMM_GENERIC=1 in the environment keeps aligned problems on the register-staged kernel
static const bool g_force_generic = getenv("DUN_GEMM_GENERIC") != nullptr && getenv("DUN_GEMM_GENERIC")[0] == '1';
// test hook: DUN_GEMM_TILE128=1 keeps small problems on 128x128 tiles
static const bool g_force_tile128 = getenv("DUN_GEMM_TILE128") != nullptr && getenv("DUN_GEMM_TILE128")[0] == '1';

// fewer 128-tiles than half the CUs: take 64x64 tiles (UCI-sized nets are one to four 128-tiles per GEMM)
static bool small_problem(int M, int N) {
    return (long long)ceil_div(M, BM) * ceil_div(N, BN) < 128 && !g_force_tile128;
}

// row tiles of an M x N problem = rows of the colstats partial matrix
int gemm_colstat_tiles(int M, int N) { return ceil_div(M, small_problem(M, N) ? 64 : BM); }

int gemm_f32(const dun_gemm_desc* d, hipStream_t s) {
    DUN_REQUIRE(d != nullptr, "dun_gemm_f32: null descriptor");
    DUN_REQUIRE(d->A && d->B && d->C, "dun_gemm_f32: null operand");
    DUN_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "dun_gemm_f32: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    DUN_REQUIRE(d->lda >= (d->a_kcontig ? d->K : d->M), "dun_gemm_f32: lda=%d too small", d->lda);
    DUN_REQUIRE(d->ldb >= (d->b_kcontig ? d->K : d->N), "dun_gemm_f32: ldb=%d too small", d->ldb);
    DUN_REQUIRE(d->ldc >= d->N, "dun_gemm_f32: ldc=%d too small", d->ldc);
    DUN_REQUIRE(d->R == nullptr || d->ldr >= d->N, "dun_gemm_f32: ldr=%d too small", d->ldr);
    DUN_REQUIRE(d->U == nullptr || (d->V != nullptr && d->O > 0 && d->ldv >= d->N),
                "dun_gemm_f32: rank update needs V, O>0 and ldv>=N");
    GemmP p;
    p.A = d->A; p.B = d->B; p.C = d->C;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.bias = d->bias; p.relu = d->relu;
    p.R = d->R; p.ldr = d->ldr;
    p.U = d->U; p.V = d->V; p.O = d->O; p.ldv = d->ldv;
    p.colstats = d->colstats;
    const bool small = small_problem(d->M, d->N);
    const int tile = small ? 64 : BM;
    p.gridM = ceil_div(d->M, tile);
    p.gridN = ceil_div(d->N, tile);
    DUN_REQUIRE((long long)p.gridM * p.gridN < (1ll << 31), "dun_gemm_f32: grid too large");
    // float4 staging needs 16-byte aligned rows on both operands; anything else takes the scalar loads
    const bool va = aligned16(d->A) && d->lda % 4 == 0 && (d->a_kcontig ? d->K % 4 == 0 : d->M % 4 == 0);
    const bool vb = aligned16(d->B) && d->ldb % 4 == 0 && (d->b_kcontig ? d->K % 4 == 0 : d->N % 4 == 0);
    const bool vec = va && vb;
    // whole tiles only: LDS-DMA cannot zero-fill, ragged problems take the register-staged kernel
    const bool dma = !small && vec && d->M % BM == 0 && d->N % BN == 0 && d->K % BK == 0 && !g_force_generic;
    if (d->a_kcontig) {
        if (d->b_kcontig) return launch_gemm<true, true>(p, vec, dma, small, s);
        return launch_gemm<true, false>(p, vec, dma, small, s);
    }
    if (d->b_kcontig) return launch_gemm<false, true>(p, vec, dma, small, s);
    return launch_gemm<false, false>(p, vec, dma, small, s);
}

}  // namespace dun

extern "C" int dun_gemm_colstat_tiles(int M, int N) { return dun::gemm_colstat_tiles(M, N); }

extern "C" int dun_gemm_f32(const dun_gemm_desc* desc, dun_stream_t stream) {
    return dun::gemm_f32(desc, static_cast<hipStream_t>(stream));
}
