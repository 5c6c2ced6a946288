// fp32 MFMA GEMM for the DUN residual-MLP trunk (gfx950).
//
// C[M,N] = epi(sum_k A(m,k) B(k,n)) on v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), so the
// trunk keeps the reference's fp32 numerics (north-star: 1e-5 relative) while running on the matrix cores.
// Three kernels share one epilogue:
//   gemm_f32_wide_kernel  whole 128x256 tiles, >= 512 of them, no split-K (the wide config's forward and dX GEMMs): the
//                         LDS-DMA pipeline below on three stages with 2x4 accumulators per wave;
//   gemm_f32_dma4_kernel  whole 128x128 tiles, 16-byte aligned operands, >= 128 tiles (every trunk GEMM of the wide
//                         config): LDS-DMA staging through four 16-deep stages, mid-tile barrier, counted waits,
//                         optional split-K (see the comment above the kernel);
//   gemm_f32_kernel       everything else (ragged, unaligned, or too small to fill the chip): operand tiles staged
//                         global -> registers -> LDS, double buffered, one barrier per K tile; 128x128 tiles or, under
//                         128 tiles, 64x64 tiles; optional split-K.
// Workgroup = 256 threads = 4 waves (2x2); a wave owns a 64x64 (or 32x32) sub-tile = 2x2 (1x1) MFMA accumulators.
//
// Operands keep their global orientation in LDS, so no transposes are ever materialised:
//   K-contiguous operand (x / h / dU rows, Linear weight rows): a lane reads its four k values with one ds_read_b128
//     (register-staged tiles: [T][32+4] floats, 144-B row stride; DMA tiles: unpadded with an XOR chunk swizzle);
//   M/N-contiguous operand (W in dX = dU W, dU^T and H in dW = dU^T H): LDS [k][T]; a lane reads one float per MFMA
//     with ds_read_b32, 32 consecutive lanes on 32 consecutive banks.
// Both forms feed the MFMA with k(step s=(t,e), lane half h) = 8t + 4h + e, so any pairing is consistent.
//
// Epilogues (all optional, fused): + bias[n], ReLU, + R[m,n] (residual / running gradient, may alias C), + U[m,:] V[:,n]
// (the head's rank-O gradient contribution), and per-row-tile column sums of C and C^2 in fp64 (BatchNorm batch
// statistics, reduced later in a fixed order).
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
    const float* bnz;      // BatchNorm backward statistics of the block BELOW fused into this epilogue (dX GEMM): its
    const float* bn_mean;  // ReLU output Z [M, ldz], batch mean / rstd [N]; bnstats float [2][gridM][N] receives the
    const float* bn_rstd;  // per-row-tile column sums of C and C * (Z - mean) * rstd
    float* bnstats;
    int ldz;
    int gridM, gridN;
    int ksplit;   // > 1: blockIdx.y takes K range [y*K/ksplit, (y+1)*K/ksplit) and stores its raw sum to part[y]
    float* part;  // [ksplit][M][N]
    int rinit;    // LDS-DMA kernel: accumulators start from R (loaded under the pipeline prologue), not added at the end
    int wide;     // 128 x 256 tiles (gemm_f32_wide_kernel): gridN counts 256-column tiles
    // inference epilogue (dun_gemm_desc::ev_*): eval-mode BatchNorm + residual + the head's partial row-dot
    const float* ev_mean;
    const float* ev_scale;
    const float* ev_shift;
    const float* ev_hw;
    float* ev_hpart;
    int x3;       // exploratory bf16 x 3 kernel (gemm_precision() == 1)
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

// ---- element access of the 128x128 (64x64) output tile in the MFMA C/D layout -----------------------------------------
// Element (mi, ni, r) of a lane: row = m0 + wr*WT + mi*32 + (r&3) + 8*(r>>2) + 4*h, col = n0 + wc*WT + ni*32 + li.
// FULL tiles (the LDS-DMA kernels) go through buffer instructions: ONE per-lane byte offset per matrix
// ((4h*ld + wc*WT + li)*4, + ni*128 as an immediate) and the row part ((m0 + wr*WT + mi*32 + (r&3) + 8(r>>2))*ld + n0)*4
// as a SCALAR offset -- with flat addresses the 64 loads + 64 stores of an epilogue held a 64-bit address each and
// the epilogue spilled.  Ragged tiles (register-staged kernels) use bounds-checked flat accesses.
template <bool FULL, int MI, int NI = MI>
struct TileIO {
    static constexpr int WT = 32 * MI, WTN = 32 * NI;  // rows / columns of a wave's sub-tile
    __amdgpu_buffer_rsrc_t rs;
    const float* base;
    int ld, voff, M, N, m0w, n0, colw, h;

    __device__ __forceinline__ TileIO(const float* ptr, int ld_, const GemmP& p, int m0, int n0_, int wr, int wc, int li,
                                      int h_) {
        base = ptr; ld = ld_; M = p.M; N = p.N; h = h_;
        m0w = m0 + wr * WT;        // wave-uniform on the FULL path (wr comes from a readfirstlane)
        n0 = n0_;
        colw = wc * WTN + li;
        if (FULL) {
            rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, (int)((((long long)p.M - 1) * ld + p.N) * 4),
                                                   0x00020000);
            voff = (4 * h * ld + colw) * 4;
        }
    }
    __device__ __forceinline__ int rowu(int mi, int r) const { return m0w + mi * 32 + (r & 3) + 8 * (r >> 2); }
    __device__ __forceinline__ bool inside(int mi, int ni, int r) const {
        return FULL || (rowu(mi, r) + 4 * h < M && n0 + colw + ni * 32 < N);
    }
    __device__ __forceinline__ float load(int mi, int ni, int r) const {
        if (FULL)
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + ni * 128,
                                                                                  (rowu(mi, r) * ld + n0) * 4, 0));
        return inside(mi, ni, r) ? base[(size_t)(rowu(mi, r) + 4 * h) * ld + n0 + colw + ni * 32] : 0.f;
    }
    __device__ __forceinline__ void store(int mi, int ni, int r, float v) const {
        if (FULL)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff + ni * 128,
                                                  (rowu(mi, r) * ld + n0) * 4, 0);
        else if (inside(mi, ni, r))
            const_cast<float*>(base)[(size_t)(rowu(mi, r) + 4 * h) * ld + n0 + colw + ni * 32] = v;
    }
};

// C = R + A B with the accumulators STARTING from the R tile: its loads go out in front of the operand pipeline's
// prologue and land under it, instead of a dependent HBM round trip in the epilogue behind an empty matrix pipe.
template <bool FULL, int MI, int NI = MI>
__device__ __forceinline__ void seed_acc(const GemmP& p, f32x16 (&acc)[MI][NI], int m0, int n0, int wr, int wc, int li,
                                         int h) {
    if (p.rinit) {
        const TileIO<FULL, MI, NI> R(p.R, p.ldr, p, m0, n0, wr, wc, li, h);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = R.load(mi, ni, r);
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
}

// ---- epilogue shared by all kernels.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// A wave owns MI x NI accumulators (32*MI rows, 32*NI columns); the four waves are 2 x 2.
// FULL: the tile lies completely inside C (no bounds checks).
// BNS: the launch takes the BatchNorm backward statistics (GemmP::bnz) -- then there is no bias / ReLU / residual /
// forward statistics in the epilogue (host-checked), and the two forms are compiled apart so that neither carries the
// other's registers.
// LEAN (the wide kernel, 128 accumulator registers): no residual gather (the host only sends residuals that seed the
// accumulators) and no rank update here -- the kernel adds the rank-1 term u v^T as one more MFMA step per accumulator
// (holding the 32 row values of u next to 128 accumulators and a Z tile made the register allocator spill).
template <bool FULL, int MI, int NI, bool BNS, bool LEAN = false>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmP& p, f32x16 (&acc)[MI][NI], float* smem, int m0, int n0,
                                                   int bm, int tid, int wave, int wr, int wc, int li, int h) {
    constexpr int WTN = 32 * NI;  // columns of a wave's sub-tile
    // Columns are worked through in chunks of NC accumulator columns: the addend tile of a chunk (residual, or the Z
    // tile of the BatchNorm statistics) is gathered into MI*NC*16 registers before the chunk's stores.  Square wave
    // tiles are one chunk; the 2 x 4 tile of the wide kernel takes two (all four at once would be 128 registers next to
    // 128 accumulators).
    constexpr int NC = NI > 2 ? 2 : NI;
    // BatchNorm statistics are accumulated in fp64 from the first element on: sum z^2 - (sum z)^2/n in fp32
    // loses the variance of a nearly constant column (narrow inputs make many of those), and the reference's
    // batch_norm uses a cancellation-free update.  64 dfma per lane per tile: free next to the K loop.
    double s1[NI], s2[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) s1[ni] = s2[ni] = 0.0;
    // Addends first, stores later.  R may BE C (the running gradient is updated in place), so the compiler cannot
    // move a residual load above an earlier store of this thread and would pay one HBM round trip per element; every
    // element is read and written by its own thread only, so gathering all of them up front is safe.
    // The residual tile and the Z tile of the BatchNorm statistics share one register array: a launch has one or the
    // other in its epilogue (the trunk's dX GEMM seeds its accumulators with the residual instead, `rinit`).
    constexpr bool has_bn = BNS;
    const bool has_r = !BNS && !LEAN && p.R != nullptr && !p.rinit;
    const bool has_u1 = !LEAN && p.U != nullptr && p.O == 1;
    const TileIO<FULL, MI, NI> Cio(p.C, p.ldc, p, m0, n0, wr, wc, li, h);
    const TileIO<FULL, MI, NI> Aio(has_bn ? p.bnz : p.R, has_bn ? p.ldz : p.ldr, p, m0, n0, wr, wc, li, h);
    float uu[LEAN ? 1 : MI][16], sg[NI], sgx[NI];  // LEAN: the row values of one mi at a time (see below)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) sg[ni] = sgx[ni] = 0.f;
    if (has_u1 && !LEAN) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = Cio.rowu(mi, r) + 4 * h;
                uu[mi][r] = (FULL || row < p.M) ? p.U[row] : 0.f;
            }
    }
#pragma unroll
    for (int n0c = 0; n0c < NI; n0c += NC) {
        float aux[MI][NC][16];
        if (has_r || has_bn) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                    for (int r = 0; r < 16; ++r) aux[mi][nc][r] = Aio.load(mi, n0c + nc, r);
        }
        float bv[NC], v1[NC], bmu[NC], brs[NC];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) {
            const int col = n0 + wc * WTN + (n0c + nc) * 32 + li;
            const bool col_ok = FULL || col < p.N;
            bv[nc] = (!BNS && p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
            v1[nc] = (has_u1 && col_ok) ? p.V[col] : 0.f;
            bmu[nc] = (has_bn && col_ok) ? p.bn_mean[col] : 0.f;
            brs[nc] = (has_bn && col_ok) ? p.bn_rstd[col] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (LEAN && has_u1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) uu[0][r] = p.U[Cio.rowu(mi, r) + 4 * h];
            }
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) {
                const int ni = n0c + nc;
                const int col = n0 + wc * WTN + ni * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (Cio.inside(mi, ni, r)) {
                        float v = acc[mi][ni][r] + bv[nc];
                        if (!BNS && p.relu) v = v > 0.f ? v : 0.f;
                        if (has_r) v += aux[mi][nc][r];
                        if (has_u1) {
                            v += uu[LEAN ? 0 : mi][r] * v1[nc];
                        } else if (!BNS && !LEAN && p.U != nullptr) {  // (rank-O with the statistics epilogue: refused on the host)
                            const int row = Cio.rowu(mi, r) + 4 * h;
                            for (int o = 0; o < p.O; ++o) v += p.U[(size_t)row * p.O + o] * p.V[(size_t)o * p.ldv + col];
                        }
                        Cio.store(mi, ni, r, v);
                        if (has_bn) {
                            sg[ni] += v;
                            sgx[ni] += v * ((aux[mi][nc][r] - bmu[nc]) * brs[nc]);
                        }
                        if (!BNS && p.colstats != nullptr) {
                            const double dv = (double)v;
                            s1[ni] += dv;
                            s2[ni] += dv * dv;
                        }
                    }
                }
                // (next to 128 accumulators the scheduler must not interleave the 16-element groups: it would hold
                // every group's temporaries at once and spill)
                if (LEAN) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (has_bn) {
        // 32*MI rows per lane pair (fp32, as the column-owned reduction kernel sums 32 rows per thread), the two lane
        // halves, then the two wave rows through LDS: one partial per row tile, fixed order
        float* redf = smem;  // all waves are past the last barrier: LDS is free
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            sg[ni] += __shfl_xor(sg[ni], 32, 64);
            sgx[ni] += __shfl_xor(sgx[ni], 32, 64);
            if (h == 0) {
                redf[((wave * NI + ni) * 32 + li) * 2 + 0] = sg[ni];
                redf[((wave * NI + ni) * 32 + li) * 2 + 1] = sgx[ni];
            }
        }
        __syncthreads();
        if (tid < 2 * WTN) {
            const int wc2 = tid / WTN, ni2 = (tid / 32) % NI, l2 = tid & 31, col = n0 + tid;
            if (FULL || col < p.N) {
                const int w0 = wc2, w1 = 2 + wc2;  // wave rows 0 and 1 of this wave column
                p.bnstats[(size_t)bm * p.N + col] =
                    redf[((w0 * NI + ni2) * 32 + l2) * 2 + 0] + redf[((w1 * NI + ni2) * 32 + l2) * 2 + 0];
                p.bnstats[(size_t)(p.gridM + bm) * p.N + col] =
                    redf[((w0 * NI + ni2) * 32 + l2) * 2 + 1] + redf[((w1 * NI + ni2) * 32 + l2) * 2 + 1];
            }
        }
    }
    if (!BNS && p.colstats != nullptr) {
        double* red = reinterpret_cast<double*>(smem);  // all waves are past the last barrier: LDS is free
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            s1[ni] += __shfl_xor(s1[ni], 32, 64);
            s2[ni] += __shfl_xor(s2[ni], 32, 64);
            if (h == 0) {
                red[((wave * NI + ni) * 32 + li) * 2 + 0] = s1[ni];
                red[((wave * NI + ni) * 32 + li) * 2 + 1] = s2[ni];
            }
        }
        __syncthreads();
        if (tid < 2 * WTN) {
            const int wc2 = tid / WTN, ni2 = (tid / 32) % NI, l2 = tid & 31, col = n0 + tid;
            if (FULL || col < p.N) {
                double t1 = 0.0, t2 = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const int wv = w2 * 2 + wc2;
                    t1 += red[((wv * NI + ni2) * 32 + l2) * 2 + 0];
                    t2 += red[((wv * NI + ni2) * 32 + l2) * 2 + 1];
                }
                p.colstats[(size_t)bm * p.N + col] = t1;
                p.colstats[(size_t)(p.gridM + bm) * p.N + col] = t2;
            }
        }
    }
}

// Inference epilogue of a trunk block (whole 128 x 128 tiles): with BatchNorm in eval mode the block is a per-column
// affine map of ReLU(Linear), so  C = [R +] (max(acc + bias, 0) - mean) * scale + shift  leaves the GEMM as the next
// hidden state -- no ReLU(Linear) tensor is written and no normalisation pass re-reads it -- and the shared head's
// row-dot over this tile's 128 columns, hpart[bn][row] = sum_n C[row][n] hw[n], is taken while the tile is in
// registers ((z - mean) first, then scale: the arithmetic of bn_apply_kernel, see there).  The 32 row sums of a lane are
// reduced over the 32 lanes of its half on the DPP path, the two wave columns are added through LDS in a fixed order.
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue_eval(const GemmP& p, f32x16 (&acc)[MI][NI], float* smem, int m0, int n0, int bn,
                                                   int tid, int wr, int wc, int li, int h) {
    constexpr int WTN = 32 * NI;           // columns of a wave's sub-tile: 64 (two wave columns per 128-column partial) or 128
    constexpr int NC = NI > 2 ? 2 : NI;    // accumulator columns whose residual values are gathered at a time
    const TileIO<true, MI, NI> Cio(p.C, p.ldc, p, m0, n0, wr, wc, li, h);
    const TileIO<true, MI, NI> Rio(p.R != nullptr ? p.R : p.C, p.R != nullptr ? p.ldr : p.ldc, p, m0, n0, wr, wc, li, h);
    float bv[NI], mu[NI], sc[NI], sh[NI], hw[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = n0 + wc * WTN + ni * 32 + li;
        bv[ni] = p.bias[col]; mu[ni] = p.ev_mean[col]; sc[ni] = p.ev_scale[col]; sh[ni] = p.ev_shift[col];
        hw[ni] = p.ev_hw != nullptr ? p.ev_hw[col] : 0.f;
    }
    float* hp_s = smem;  // [2 wave columns][128 rows]; all waves are past the last barrier: LDS is free
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        float hp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hp[r] = 0.f;
#pragma unroll
        for (int n0c = 0; n0c < NI; n0c += NC) {
            float aux[NC][16];
            if (p.R != nullptr) {
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                    for (int r = 0; r < 16; ++r) aux[nc][r] = Rio.load(mi, n0c + nc, r);
            }
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) {
                const int ni = n0c + nc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = fmaxf(acc[mi][ni][r] + bv[ni], 0.f);
                    v = (v - mu[ni]) * sc[ni] + sh[ni];
                    if (p.R != nullptr) v = aux[nc][r] + v;
                    Cio.store(mi, ni, r, v);
                    hp[r] += v * hw[ni];
                }
            }
            if (NI > 2) __builtin_amdgcn_sched_barrier(0);  // (128 accumulators: keep one chunk's temporaries alive at a time)
        }
        if (p.ev_hpart != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = hp[r];
                x += dpp_move<DPP_QUAD_XOR1>(x);
                x += dpp_move<DPP_QUAD_XOR2>(x);
                x += dpp_move<DPP_ROW_HALF_MIRROR>(x);
                x += dpp_move<DPP_ROW_MIRROR>(x);
                x += __shfl_xor(x, 16, 64);  // the other 16-lane row of this half
                if (li == 0) hp_s[wc * 128 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = x;
            }
        }
    }
    if (p.ev_hpart != nullptr) {
        __syncthreads();
        if (NI == 2) {  // 128-column tile: the two wave columns are one partial
            if (tid < 128) p.ev_hpart[(size_t)bn * p.M + m0 + tid] = hp_s[tid] + hp_s[128 + tid];
        } else {        // 256-column tile: a partial per wave column, the [N/128][M] layout all the same
            p.ev_hpart[(size_t)(bn * 2 + (tid >> 7)) * p.M + m0 + (tid & 127)] = hp_s[tid];
        }
    }
}

template <bool FULL, int MI, int NI = MI, bool LEAN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[MI][NI], float* smem, int m0, int n0, int bm,
                                              int tid, int wave, int wr, int wc, int li, int h) {
    if (p.bnz != nullptr)
        gemm_epilogue_impl<FULL, MI, NI, true, LEAN>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    else
        gemm_epilogue_impl<FULL, MI, NI, false, LEAN>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
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
    seed_acc<false, MI>(p, acc, m0, n0, wr, wc, li, h);  // (rinit is never set together with split-K)

    f32x4 ra[NR], rb[NR];
    // split-K: blockIdx.y owns [kb, ke), a whole number of K tiles (the last range may be short)
    int kb = 0, ke = p.K;
    if (p.ksplit > 1) {
        const int chunk = ((p.K + p.ksplit - 1) / p.ksplit + BK - 1) / BK * BK;
        kb = (int)blockIdx.y * chunk;
        ke = kb + chunk < p.K ? kb + chunk : p.K;
    }
    const int nk = ke > kb ? (ke - kb + BK - 1) / BK : 0;
    load_tile<A_KC, VEC, TB>(p.A, p.lda, m0, p.M, kb, ke, tid, ra);
    load_tile<B_KC, VEC, TB>(p.B, p.ldb, n0, p.N, kb, ke, tid, rb);
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
                load_tile<A_KC, VEC, TB>(p.A, p.lda, m0, p.M, kb + (kt + 1) * BK, ke, tid, ra);
                load_tile<B_KC, VEC, TB>(p.B, p.ldb, n0, p.N, kb + (kt + 1) * BK, ke, tid, rb);
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

    if (p.ksplit > 1) {
        GemmP q = p;  // raw partial sum of this K range; splitk_sum_kernel adds the ranges in order
        q.C = p.part + (size_t)blockIdx.y * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.relu = 0; q.R = nullptr; q.U = nullptr; q.colstats = nullptr; q.bnz = nullptr;
        gemm_epilogue<false, MI>(q, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    } else {
        gemm_epilogue<false, MI>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fast path for tile-aligned problems (M%128 == N%128 == K%32 == 0, 16-byte aligned rows): operand tiles go HBM/L2 -> LDS by
// LDS-DMA (buffer_load ... lds, 16 B per lane), no VGPR round trip and no ds_write.  A wave-instruction writes 1 KiB of LDS
// linearly (wave-uniform base + lane*16), so the images are unpadded and the bank-conflict swizzle of the K-contiguous image
// is applied to the per-lane SOURCE address on the way in and undone on the read (the same involution on both sides).
// The K loop advances in 16-deep half tiles through a ring of four 16 KiB stages (64 KiB, two workgroups per CU) and the
// barrier sits in the MIDDLE of a half tile, behind 16 issued MFMAs (a first version with two 32-deep buffers ended every
// tile with a barrier that every wave reached with an EMPTY matrix pipe: 82-87 % of peak against 91-92 %):
//     half j:   [read frags (j, group 1)]  16 MFMAs (j, group 0)
//               wait for the DMA of half j+1 (issued two half tiles ago; counted vmcnt, half j+2 stays in flight)
//               s_barrier          -> half j+1 is visible to all waves; all waves are done reading half j-1
//               DMA half j+3 -> stage (j+3)&3 (the stage of half j-1)
//               [read frags (j+1, group 0)]  16 MFMAs (j, group 1)
// so no wave ever crosses a tile boundary with nothing to issue.  Images: K-contiguous [128][16] (64-B rows, 16-B chunk
// index XOR (row>>2)&3 on the source address and on the read: 16 consecutive rows of one chunk column then cover all
// 16 chunk slots of the 256-byte bank line -- rows r, r+4, r+8, r+12 share (row&3) and differ in (row>>2)&3; with
// (row>>1)&3 rows r and r+8 collided, 2.9 % of the forward GEMM's wave cycles), M/N-contiguous [16][128].
// ---------------------------------------------------------------------------------------------------------
#ifndef DUN_KC_SWZ_SHIFT
#define DUN_KC_SWZ_SHIFT 2  // A/B builds: -DDUN_KC_SWZ_SHIFT=1 is the first version's swizzle
#endif
constexpr int HK = 16;                 // K depth of a half tile
constexpr int HALF_TILE = BM * HK;     // 2048 floats = 8 KiB per operand
constexpr int STAGE = 2 * HALF_TILE;   // A image | B image

// The DMA goes through buffer_load ... lds (MUBUF), not global_load_lds: the compiler treats a pending FLAT-encoded
// LDS-DMA as "may be an LDS access" and then flushes lgkmcnt to 0 at every LDS wait, which would serialise the
// fragment prefetch behind the MFMAs it is supposed to hide under; with MUBUF it emits counted lgkmcnt waits.
// A raw buffer descriptor addresses 4 GiB from the operand's base with a 32-bit byte offset (checked on the host);
// its num_records bound makes an out-of-range lane read zeros instead of faulting.
struct DmaOp {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[2];  // per-lane byte offsets of the wave's two 1 KiB pieces at k0 = 0
    int kstep;    // bytes per unit of k0
};

template <bool KC>
__device__ __forceinline__ DmaOp dma_setup(const float* P, int ld, int r0, int rows, int K, int wave, int lane) {
    DmaOp d;
    const long long extent = KC ? ((long long)(rows - 1) * ld + K) * 4 : ((long long)(K - 1) * ld + rows) * 4;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (int)extent, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j;
        if (KC) {
            const int row = piece * 16 + (lane >> 2), pc = lane & 3;
            const int c = pc ^ ((row >> DUN_KC_SWZ_SHIFT) & 3);
            d.voff[j] = ((r0 + row) * ld + c * 4) * 4;
        } else {
            const int k = piece * 2 + (lane >> 5), col = (lane & 31) * 4;
            d.voff[j] = (k * ld + r0 + col) * 4;
        }
    }
    d.kstep = KC ? 4 : ld * 4;
    return d;
}

__device__ __forceinline__ void dma_half(const DmaOp& d, int k0, float* lds, int wave) {
    const int soff = k0 * d.kstep;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * 2 + j) * 256),
                                                 16, d.voff[j], soff, 0, 0);
}

// fragment of k group t (0/1) of a half tile: element e of lane (row, h) is operand[row][8t + 4h + e]
template <bool KC>
__device__ __forceinline__ void read_frag_half(const float* lds, int row, int t, int h, float (&f)[4]) {
    if (KC) {
        const int pc = (2 * t + h) ^ ((row >> DUN_KC_SWZ_SHIFT) & 3);
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + row * HK + pc * 4);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = lds[(8 * t + 4 * h + e) * BM + row];
    }
}

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt = bits [3:0] and [15:14]; expcnt / lgkmcnt fields left at "no wait")
#define DUN_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
// s_waitcnt lgkmcnt(0) alone.  Placed behind each group of 16 MFMAs, where the fragment reads issued in front of that
// group landed long ago, so it costs nothing -- but it tells the compiler that nothing is pending when the next
// prefetch goes out.  Without it the M/N-contiguous operand images (eight ds_read2_b32 per group) have 16 LDS reads
// outstanding at the first MFMA of a group, one more than the 4-bit counter can express, and the compiler falls back
// to lgkmcnt(0) THERE: the prefetch it just issued is waited for instead of hidden (dW: K loop at 0.908 of peak
// against 0.942 for the K-contiguous images, whose four ds_read_b128 per group stay countable).
#ifdef DUN_NO_LGKM_WAIT  // A/B build
#define DUN_WAIT_LGKMCNT0() do { } while (0)
#else
#define DUN_WAIT_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)
#endif

#ifdef DUN_GEMM_TIMING
// Diagnostic build only (scripts/gemm_phases.py builds its own library with -DDUN_GEMM_TIMING): every workgroup of the
// four-stage kernel records the shader clock at its start, at its first MFMA, at the end of the K loop and at its end,
// plus the hardware slot it ran on.  Not part of the shipped library.
__device__ unsigned long long* g_gemm_timing = nullptr;
#define DUN_TSTAMP(i) do { if (tid == 0 && g_gemm_timing) tstamp[i] = clock64(); } while (0)
#else
#define DUN_TSTAMP(i) do { } while (0)
#endif

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma4_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * STAGE];
#ifdef DUN_GEMM_TIMING
    unsigned long long tstamp[5];
    if (threadIdx.x == 0 && g_gemm_timing) tstamp[0] = clock64();
#endif

    const int nwg = p.gridM * p.gridN;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // (tiles of one row panel are consecutive; grouping 4 / 8 / 16 row panels changed none of the three trunk GEMMs)
    const int bm = bid / p.gridN, bn = bid % p.gridN;
    const int m0 = bm * BM, n0 = bn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
    seed_acc<true, 2>(p, acc, m0, n0, wr, wc, li, h);

    const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * (p.K / p.ksplit) : 0;
    const int nh = (p.ksplit > 1 ? p.K / p.ksplit : p.K) / HK;  // >= 2 (K range % 32 == 0)
    const DmaOp dA = dma_setup<A_KC>(p.A, p.lda, m0, p.M, p.K, wave, lane);
    const DmaOp dB = dma_setup<B_KC>(p.B, p.ldb, n0, p.N, p.K, wave, lane);
    // prologue: halves 0..2 in flight (4 DMA instructions per half and wave), half 0 landed and visible
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nh) {
            dma_half(dA, kbeg + j * HK, smem + j * STAGE, wave);
            dma_half(dB, kbeg + j * HK, smem + j * STAGE + HALF_TILE, wave);
        }
    }
    if (nh >= 3) DUN_WAIT_VMCNT(8); else DUN_WAIT_VMCNT(4);
    __builtin_amdgcn_s_barrier();
    DUN_TSTAMP(1);

    // Every kernel argument the epilogue needs is pulled into SGPRs HERE: a scalar (kernel-argument) load still
    // pending at the loop entry shares lgkmcnt with the LDS reads, the counter is then no longer in-order for the
    // compiler, and it would wait for lgkmcnt(0) -- i.e. for the prefetch just issued -- at every MFMA group.
    asm volatile("" ::"s"(p.C), "s"(p.bias), "s"(p.R), "s"(p.U), "s"(p.V), "s"(p.colstats), "s"(p.M), "s"(p.N),
                 "s"(p.ldc), "s"(p.ldr), "s"(p.O), "s"(p.ldv), "s"(p.relu), "s"(p.gridM), "s"(p.rinit), "s"(p.bnz),
                 "s"(p.bn_mean), "s"(p.bn_rstd), "s"(p.bnstats), "s"(p.ldz));

    float fa[2][2][4], fb[2][2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) read_frag_half<A_KC>(smem, wr * 64 + mi * 32 + li, 0, h, fa[0][mi]);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) read_frag_half<B_KC>(smem + HALF_TILE, wc * 64 + ni * 32 + li, 0, h, fb[0][ni]);
    DUN_WAIT_LGKMCNT0();  // (the loop is entered with nothing pending on either path: see DUN_WAIT_LGKMCNT0)

    // One half tile.  NEXT: half j+1 exists; INFLIGHT: half j+2 was issued (so it may stay in flight); ISSUE: half j+3
    // exists.  Compile-time flags: a run-time branch around the fragment prefetch would make the compiler wait for
    // ALL outstanding LDS reads at the join, i.e. for the prefetch itself.
    // (fragment reads and DMA issues spread over the gaps between the MFMAs: see DUN_WIDE_SPREAD; a group is 16 MFMAs here.
    //  K-contiguous images are read with one ds_read_b128 per fragment, M/N-contiguous ones with two ds_read2_b32)
    constexpr int DUN_HALF_NREAD = (A_KC ? 2 : 4) + (B_KC ? 2 : 4);
#ifndef DUN_NO_INTERLEAVE
#define DUN_HALF_SPREAD(NMEM, PER)                                                                                     \
    do {                                                                                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < (NMEM); ++g_) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);                                                         \
        }                                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - (NMEM) * (PER), 0);                                           \
    } while (0)
#define DUN_HALF_SEP() do { } while (0)
#else
#define DUN_HALF_SPREAD(NMEM, PER) do { } while (0)
#define DUN_HALF_SEP() __builtin_amdgcn_sched_barrier(0)
#endif
#define DUN_HALF_STEP(J, NEXT, INFLIGHT, ISSUE)                                                                        \
    do {                                                                                                               \
        const float* a_s = smem + ((J) & 3) * STAGE;                                                                   \
        const float* b_s = a_s + HALF_TILE;                                                                            \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                               \
            read_frag_half<A_KC>(a_s, wr * 64 + mi * 32 + li, 1, h, fa[1][mi]);                                        \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                               \
            read_frag_half<B_KC>(b_s, wc * 64 + ni * 32 + li, 1, h, fb[1][ni]);                                        \
        DUN_HALF_SEP(); /* (burst form: keep the prefetch IN FRONT of the MFMAs it hides under) */                     \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                       \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][mi][e], fb[0][ni][e], acc[mi][ni], 0, 0, 0); \
        DUN_HALF_SPREAD(DUN_HALF_NREAD, 16 / DUN_HALF_NREAD > 3 ? 3 : 16 / DUN_HALF_NREAD);                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        DUN_WAIT_LGKMCNT0();                                                                                           \
        if (NEXT) {                                                                                                    \
            if (INFLIGHT) DUN_WAIT_VMCNT(4); else DUN_WAIT_VMCNT(0);                                                   \
            __builtin_amdgcn_s_barrier();                                                                              \
            if (ISSUE) {                                                                                               \
                float* st = smem + (((J) + 3) & 3) * STAGE; /* stage of half J-1: every wave is past its last read */  \
                dma_half(dA, kbeg + ((J) + 3) * HK, st, wave);                                                         \
                dma_half(dB, kbeg + ((J) + 3) * HK, st + HALF_TILE, wave);                                             \
            }                                                                                                          \
            const float* a_n = smem + (((J) + 1) & 3) * STAGE;                                                         \
            const float* b_n = a_n + HALF_TILE;                                                                        \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                read_frag_half<A_KC>(a_n, wr * 64 + mi * 32 + li, 0, h, fa[0][mi]);                                    \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
                read_frag_half<B_KC>(b_n, wc * 64 + ni * 32 + li, 0, h, fb[0][ni]);                                    \
        }                                                                                                              \
        DUN_HALF_SEP();                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                       \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][mi][e], fb[1][ni][e], acc[mi][ni], 0, 0, 0); \
        if (NEXT) DUN_HALF_SPREAD((ISSUE ? 4 : 0) + DUN_HALF_NREAD, 1);                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        DUN_WAIT_LGKMCNT0();                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

    int j = 0;
    for (; j + 3 < nh; ++j) DUN_HALF_STEP(j, true, true, true);   // steady state
    if (j + 2 < nh) { DUN_HALF_STEP(j, true, true, false); ++j; }  // half nh-3: nothing left to issue
    if (j + 1 < nh) { DUN_HALF_STEP(j, true, false, false); ++j; } // half nh-2: the last DMA must land
    DUN_HALF_STEP(j, false, false, false);                         // half nh-1
#undef DUN_HALF_STEP
#undef DUN_HALF_SPREAD
#undef DUN_HALF_SEP
    DUN_TSTAMP(2);
    __syncthreads();  // the epilogue reuses the LDS for the column statistics
    if (p.ksplit > 1) {
        GemmP q = p;  // raw partial sum of this K range; splitk_sum_kernel adds the ranges in order
        q.C = p.part + (size_t)blockIdx.y * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.relu = 0; q.R = nullptr; q.U = nullptr; q.colstats = nullptr; q.bnz = nullptr;
        gemm_epilogue<true, 2>(q, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    } else if (p.ev_scale != nullptr) {
        gemm_epilogue_eval<2, 2>(p, acc, smem, m0, n0, bn, tid, wr, wc, li, h);
    } else {
        gemm_epilogue<true, 2>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    }
#ifdef DUN_GEMM_TIMING
    if (tid == 0 && g_gemm_timing) {
        tstamp[4] = clock64();          // the epilogue's last instruction is issued
        __builtin_amdgcn_s_waitcnt(0);  // ... and its stores have been acknowledged
        tstamp[3] = clock64();
        unsigned long long* o = g_gemm_timing + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        for (int i = 0; i < 5; ++i) __builtin_nontemporal_store(tstamp[i], o + i);
        __builtin_nontemporal_store((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4), o + 5);    // HW_ID
        __builtin_nontemporal_store((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20), o + 6);   // XCC_ID
    }
#endif
}

#ifdef DUN_GEMM_TIMING
extern "C" int dun_debug_gemm_timing(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(dun::g_gemm_timing), &buf, sizeof(buf));
}
#endif

// ---------------------------------------------------------------------------------------------------------
// EXPLORATORY, off by default (dun_gemm_precision(1) / DUN_GEMM_BF16X3=1), never the measured headline: the same
// 128 x 128 LDS-DMA pipeline with each fp32 operand element split EXACTLY into three bf16 digits in registers
// (x = hi + mid + lo: 8 + 8 + 8 mantissa bits, by truncation, so nothing is lost in the split) and the product taken
// as six bf16 MFMAs with fp32 accumulation -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the dropped terms
// (mid*lo, lo*mid, lo*lo) are below 2^-23 of |a*b| per element product.  bf16 MFMA runs at 16x the fp32 MFMA rate,
// so six products are 0.375 of the matrix-pipe time; the split costs ~5.5 vector-ALU operations per operand element
// per wave, which the other wave of the SIMD runs under the MFMAs.  HBM and LDS traffic are those of the fp32 kernel
// (operands travel as fp32).  Any assignment of k to (lane half, element) is valid for the MFMA as long as A and B use
// the same one: a lane's eight k values are the fp32 kernel's fragments of k groups 0 and 1.
// ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// planes of a fragment: [0] = hi, [1] = mid, [2] = lo; 8 k values per lane as packed bf16 pairs
__device__ __forceinline__ void x3_split(const float (&f0)[4], const float (&f1)[4], u32x4 (&o)[3]) {
    float x[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
    float r[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float hi = __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
        r[i] = x[i] - hi;                                                      // exact: the low 16 mantissa bits
        const float mid = __uint_as_float(__float_as_uint(r[i]) & 0xffff0000u);
        q[i] = r[i] - mid;                                                     // exact: <= 8 significant bits left
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // (element 2i in the low half, 2i+1 in the high half: top 16 bits of each)
        o[0][i] = __builtin_amdgcn_perm(__float_as_uint(x[2 * i + 1]), __float_as_uint(x[2 * i]), 0x07060302u);
        o[1][i] = __builtin_amdgcn_perm(__float_as_uint(r[2 * i + 1]), __float_as_uint(r[2 * i]), 0x07060302u);
        o[2][i] = __builtin_amdgcn_perm(__float_as_uint(q[2 * i + 1]), __float_as_uint(q[2 * i]), 0x07060302u);
    }
}

#define DUN_X3_MFMA(a, b, c) \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
// the four accumulators of a wave with plane PA of A against plane PB of B
#define DUN_X3_PRODUCT(XA, XB, PA, PB)                                                                                 \
    do {                                                                                                               \
        DUN_X3_MFMA(XA[0][PA], XB[0][PB], acc[0][0]); DUN_X3_MFMA(XA[0][PA], XB[1][PB], acc[0][1]);                    \
        DUN_X3_MFMA(XA[1][PA], XB[0][PB], acc[1][0]); DUN_X3_MFMA(XA[1][PA], XB[1][PB], acc[1][1]);                    \
    } while (0)

__device__ __forceinline__ void x3_wait_groups(int n) {  // at most n DMA groups (4 loads each) of this wave in flight
    if (n >= 3) DUN_WAIT_VMCNT(12);
    else if (n == 2) DUN_WAIT_VMCNT(8);
    else if (n == 1) DUN_WAIT_VMCNT(4);
    else DUN_WAIT_VMCNT(0);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * STAGE];
    const int nwg = p.gridM * p.gridN;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.gridN, bn = bid % p.gridN;
    const int m0 = bm * BM, n0 = bn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
    seed_acc<true, 2>(p, acc, m0, n0, wr, wc, li, h);

    const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * (p.K / p.ksplit) : 0;
    const int nh = (p.ksplit > 1 ? p.K / p.ksplit : p.K) / HK;  // >= 2
    const DmaOp dA = dma_setup<A_KC>(p.A, p.lda, m0, p.M, p.K, wave, lane);
    const DmaOp dB = dma_setup<B_KC>(p.B, p.ldb, n0, p.N, p.K, wave, lane);
    // Software pipeline over half tiles (16 k): during step J the wave issues the MFMAs of half J from bf16 registers,
    // splits the fp32 fragments of half J+1 (read from LDS during step J-1) under them, and reads the fragments of
    // half J+2; the DMA runs three halves ahead of the reads through a ring of four stages.
    const int npre = nh < 4 ? nh : 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nh) {
            dma_half(dA, kbeg + j * HK, smem + j * STAGE, wave);
            dma_half(dB, kbeg + j * HK, smem + j * STAGE + HALF_TILE, wave);
        }
    }
    asm volatile("" ::"s"(p.C), "s"(p.bias), "s"(p.R), "s"(p.U), "s"(p.V), "s"(p.colstats), "s"(p.M), "s"(p.N),
                 "s"(p.ldc), "s"(p.ldr), "s"(p.O), "s"(p.ldv), "s"(p.relu), "s"(p.gridM), "s"(p.rinit), "s"(p.bnz),
                 "s"(p.bn_mean), "s"(p.bn_rstd), "s"(p.bnstats), "s"(p.ldz));

    float fa[2][2][4], fb[2][2][4];  // fp32 fragments [k group][tile][e] of the half being split next
#define DUN_X3_READ(ST)                                                                                                \
    do {                                                                                                               \
        const float* a_s = smem + (ST) * STAGE;                                                                        \
        const float* b_s = a_s + HALF_TILE;                                                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                read_frag_half<A_KC>(a_s, wr * 64 + mi * 32 + li, t, h, fa[t][mi]);                                    \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
                read_frag_half<B_KC>(b_s, wc * 64 + ni * 32 + li, t, h, fb[t][ni]);                                    \
        }                                                                                                              \
    } while (0)
#define DUN_X3_SPLIT(XA, XB)                                                                                           \
    do {                                                                                                               \
        x3_split(fa[0][0], fa[1][0], XA[0]); x3_split(fa[0][1], fa[1][1], XA[1]);                                      \
        x3_split(fb[0][0], fb[1][0], XB[0]); x3_split(fb[0][1], fb[1][1], XB[1]);                                      \
    } while (0)

    u32x4 x0a[2][3], x0b[2][3], x1a[2][3], x1b[2][3];  // two sets of bf16 operands: [tile][plane]
    x3_wait_groups(npre - 1);  // half 0 landed ...
    __builtin_amdgcn_s_barrier();  // ... everyone's pieces
    DUN_X3_READ(0);
    DUN_WAIT_LGKMCNT0();
    DUN_X3_SPLIT(x0a, x0b);
    x3_wait_groups(npre - 2);  // half 1 (nh >= 2)
    __builtin_amdgcn_s_barrier();  // (and everyone is past reading half 0: its stage takes half 4, as "step -1" would)
    if (nh > 4) {
        dma_half(dA, kbeg + 4 * HK, smem, wave);
        dma_half(dB, kbeg + 4 * HK, smem + HALF_TILE, wave);
    }
    DUN_X3_READ(1);

    // (timing experiments, results wrong: -DDUN_X3_NOSPLIT / _NODMA / _NOREAD take a piece out of the steady state)
#ifdef DUN_X3_NOSPLIT
#define DUN_X3E_SPLIT false
#else
#define DUN_X3E_SPLIT true
#endif
#ifdef DUN_X3_NODMA
#define DUN_X3E_DMA false
#else
#define DUN_X3E_DMA true
#endif
#ifdef DUN_X3_NOPATTERN
#define DUN_X3E_PATTERN false
#else
#define DUN_X3E_PATTERN true
#endif
#ifdef DUN_X3_NOREAD
#define DUN_X3E_READ false
#else
#define DUN_X3E_READ true
#endif
    // FAST: steady state (halves J+1 .. J+5 exist), compile-time shape and a forced 1 MFMA : 15 vector-ALU issue pattern
#define DUN_X3_STEP(J, CA, CB, NA, NB, FAST)                                                                           \
    do {                                                                                                               \
        DUN_WAIT_LGKMCNT0(); /* the fp32 fragments of half J+1 are in registers */                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((FAST && DUN_X3E_SPLIT) || (!FAST && (J) + 1 < nh)) DUN_X3_SPLIT(NA, NB);                                  \
        DUN_X3_PRODUCT(CA, CB, 0, 2); DUN_X3_PRODUCT(CA, CB, 2, 0); DUN_X3_PRODUCT(CA, CB, 1, 1); /* small terms */    \
        if (FAST && DUN_X3E_PATTERN) {                                                                                 \
            _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                                        \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
                __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);                                                    \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (FAST || (J) + 2 < nh) {                                                                                    \
            if (FAST) DUN_WAIT_VMCNT(8); else x3_wait_groups((nh - 1 < (J) + 4 ? nh - 1 : (J) + 4) - ((J) + 2));       \
            __builtin_amdgcn_s_barrier(); /* half J+2 landed for everyone; everyone is past reading half J+1 */       \
            if ((FAST && DUN_X3E_DMA) || (!FAST && (J) + 5 < nh)) {                                                    \
                float* st = smem + (((J) + 1) & 3) * STAGE;                                                            \
                dma_half(dA, kbeg + ((J) + 5) * HK, st, wave);                                                         \
                dma_half(dB, kbeg + ((J) + 5) * HK, st + HALF_TILE, wave);                                             \
            }                                                                                                          \
            if (!FAST || DUN_X3E_READ) DUN_X3_READ(((J) + 2) & 3);                                                     \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        DUN_X3_PRODUCT(CA, CB, 0, 1); DUN_X3_PRODUCT(CA, CB, 1, 0); DUN_X3_PRODUCT(CA, CB, 0, 0); /* leading last */   \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

    int j = 0;
    for (; j + 6 < nh; j += 2) {
        DUN_X3_STEP(j, x0a, x0b, x1a, x1b, true);
        DUN_X3_STEP(j + 1, x1a, x1b, x0a, x0b, true);
    }
    while (j < nh) {  // the last (up to six) halves: run-time shape
        DUN_X3_STEP(j, x0a, x0b, x1a, x1b, false);
        if (++j >= nh) break;
        DUN_X3_STEP(j, x1a, x1b, x0a, x0b, false);
        ++j;
    }
#undef DUN_X3_STEP
#undef DUN_X3_SPLIT
#undef DUN_X3_READ
    __syncthreads();
    if (p.ksplit > 1) {
        GemmP q = p;
        q.C = p.part + (size_t)blockIdx.y * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.relu = 0; q.R = nullptr; q.U = nullptr; q.colstats = nullptr; q.bnz = nullptr;
        gemm_epilogue<true, 2>(q, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    } else if (p.ev_scale != nullptr) {
        gemm_epilogue_eval<2, 2>(p, acc, smem, m0, n0, bn, tid, wr, wc, li, h);
    } else {
        gemm_epilogue<true, 2>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Wide variant: 128 x 256 tiles (a wave owns 64 x 128 = 2 x 4 MFMA accumulators), three 16-deep stages of 24 KiB
// (72 KiB: still two workgroups per CU).  Per barrier a wave now issues 64 MFMAs instead of 32 and reads 12 fragment
// vectors instead of 8, and a tile's prologue / store bubble is paid once per 2x the work.  Three stages carry the same
// prefetch distance as the four of the kernel above: the last fragment of half j is read at the TOP of step j and
// waited for (lgkmcnt) in front of the midpoint barrier, so behind that barrier the stage of half j itself is free and
// takes half j+3, while half j+2 stays in flight.  (First version: half j+2 into the stage of half j-1, waited for one
// step later with vmcnt(0) -- a tenth of the wave cycles waiting on memory, SQ_WAIT_ANY 0.10 against 0.04.)
// ---------------------------------------------------------------------------------------------------------
constexpr int WBN = 256;
constexpr int WA_HALF = BM * HK, WB_HALF = WBN * HK;  // floats per operand image
constexpr int WSTAGE = WA_HALF + WB_HALF;

template <bool KC, int ROWS>
struct DmaOpW {
    static constexpr int NP = ROWS / 64;  // 1 KiB pieces per wave and half tile
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[NP];
    int kstep;
    __device__ __forceinline__ void setup(const float* P, int ld, int r0, int rows, int K, int wave, int lane) {
        const long long extent = KC ? ((long long)(rows - 1) * ld + K) * 4 : ((long long)(K - 1) * ld + rows) * 4;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (int)extent, 0x00020000);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int piece = wave * NP + j;
            if (KC) {  // image [ROWS][16], 16-byte chunk index XOR (row >> 2) & 3
                const int row = piece * 16 + (lane >> 2), pc = lane & 3;
                const int c = pc ^ ((row >> DUN_KC_SWZ_SHIFT) & 3);
                voff[j] = ((r0 + row) * ld + c * 4) * 4;
            } else if (ROWS == 128) {  // image [16][128]: a piece is two k rows
                const int k = piece * 2 + (lane >> 5), col = (lane & 31) * 4;
                voff[j] = (k * ld + r0 + col) * 4;
            } else {  // image [16][256]: a piece is one k row
                voff[j] = (piece * ld + r0 + lane * 4) * 4;
            }
        }
        kstep = KC ? 4 : ld * 4;
    }
    __device__ __forceinline__ void issue(int k0, float* lds, int wave) const {
        const int soff = k0 * kstep;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * NP + j) * 256),
                                                     16, voff[j], soff, 0, 0);
    }
};

template <bool KC, int ROWS>
__device__ __forceinline__ void read_frag_w(const float* lds, int row, int t, int h, float (&f)[4]) {
    if (KC) {
        const int pc = (2 * t + h) ^ ((row >> DUN_KC_SWZ_SHIFT) & 3);
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + row * HK + pc * 4);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = lds[(8 * t + 4 * h + e) * ROWS + row];
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_wide_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float smem[3 * WSTAGE];
    const int nwg = p.gridM * p.gridN;  // gridN counts 256-column tiles here
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.gridN, bn = bid % p.gridN;
    const int m0 = bm * BM, n0 = bn * WBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, h = lane >> 5;

    f32x16 acc[2][4];
    seed_acc<true, 2, 4>(p, acc, m0, n0, wr, wc, li, h);

    // (K ranges: dW = dU^T H is 128 such tiles, four ranges fill the chip; splitk_sum_kernel adds them in order)
    const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * (p.K / p.ksplit) : 0;
    const int nh = (p.ksplit > 1 ? p.K / p.ksplit : p.K) / HK;  // >= 4 (host)
    DmaOpW<A_KC, BM> dA;
    DmaOpW<B_KC, WBN> dB;
    dA.setup(p.A, p.lda, m0, p.M, p.K, wave, lane);
    dB.setup(p.B, p.ldb, n0, p.N, p.K, wave, lane);
    // prologue: halves 0..2 in flight (2 + 4 DMA instructions per half and wave), half 0 landed and visible (nh >= 4)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dA.issue(kbeg + j * HK, smem + j * WSTAGE, wave);
        dB.issue(kbeg + j * HK, smem + j * WSTAGE + WA_HALF, wave);
    }
    DUN_WAIT_VMCNT(12);
    __builtin_amdgcn_s_barrier();
    // (kernel arguments of the epilogue into SGPRs before the loop: see gemm_f32_dma4_kernel)
    asm volatile("" ::"s"(p.C), "s"(p.bias), "s"(p.R), "s"(p.U), "s"(p.V), "s"(p.colstats), "s"(p.M), "s"(p.N),
                 "s"(p.ldc), "s"(p.ldr), "s"(p.O), "s"(p.ldv), "s"(p.relu), "s"(p.gridM), "s"(p.rinit), "s"(p.bnz),
                 "s"(p.bn_mean), "s"(p.bn_rstd), "s"(p.bnstats), "s"(p.ldz));

    float fa[2][2][4], fb[2][4][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) read_frag_w<A_KC, BM>(smem, wr * 64 + mi * 32 + li, 0, h, fa[0][mi]);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) read_frag_w<B_KC, WBN>(smem + WA_HALF, wc * 128 + ni * 32 + li, 0, h, fb[0][ni]);
    DUN_WAIT_LGKMCNT0();

    int cur = 0, nxt = WSTAGE, iss = 2 * WSTAGE;  // float offsets of the stages of half j, j+1, j+2 (a ring of three)
    // LDS read instructions per fragment group: one ds_read_b128 per K-contiguous fragment, two ds_read2_b32 otherwise
    constexpr int NREADW = (A_KC ? 2 : 4) + (B_KC ? 4 : 8);
// timing experiments only (results are wrong): what the mid-step barrier / the operand DMA / the fragment reads cost
#ifdef DUN_EXP_NODMA
#define DUN_EXP_DMA false
#else
#define DUN_EXP_DMA true
#endif
#ifdef DUN_EXP_NOLDS
#define DUN_EXP_LDS false
#else
#define DUN_EXP_LDS true
#endif
#ifdef DUN_EXP_NOBAR
#define DUN_WIDE_BARRIER() do { } while (0)
#else
#define DUN_WIDE_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifndef DUN_NO_INTERLEAVE  // (A/B build: -DDUN_NO_INTERLEAVE restores the burst in front of the group)
// Fragment reads (and, behind the barrier, the DMA issues) are SPREAD over the gaps between the group's MFMAs -- one LDS read
// or DMA issue per PER MFMAs -- instead of going out as one burst in front of them: the two waves of a SIMD share its issue
// port, and a burst of 12 reads + 6 DMA issues holds it while the partner wave's MFMAs wait for their slot.  Forward 3.795 ->
// 3.693 ms, dX 3.845 -> 3.724 ms on one box (0.946 / 0.938 of peak with their epilogues).
#define DUN_WIDE_SPREAD(NMEM, PER)                                                                                     \
    do {                                                                                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < (NMEM); ++g_) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);                                                         \
        }                                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 32 - (NMEM) * (PER), 0);                                           \
    } while (0)
#define DUN_WIDE_SEP() do { } while (0)
#else
#define DUN_WIDE_SPREAD(NMEM, PER) do { } while (0)
#define DUN_WIDE_SEP() __builtin_amdgcn_sched_barrier(0)
#endif
#define DUN_WIDE_STEP(J, NEXT, INFLIGHT, ISSUE)                                                                                 \
    do {                                                                                                               \
        const float* a_s = smem + cur;                                                                                 \
        const float* b_s = a_s + WA_HALF;                                                                              \
        if (DUN_EXP_LDS) {                                                                                             \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                               \
            read_frag_w<A_KC, BM>(a_s, wr * 64 + mi * 32 + li, 1, h, fa[1][mi]);                                       \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                               \
            read_frag_w<B_KC, WBN>(b_s, wc * 128 + ni * 32 + li, 1, h, fb[1][ni]);                                     \
        }                                                                                                              \
        DUN_WIDE_SEP();                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                       \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][mi][e], fb[0][ni][e], acc[mi][ni], 0, 0, 0); \
        DUN_WIDE_SPREAD(NREADW, 32 / NREADW > 4 ? 4 : 32 / NREADW);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        DUN_WAIT_LGKMCNT0();                                                                                           \
        if (NEXT) {                                                                                                    \
            if (INFLIGHT) DUN_WAIT_VMCNT(6); else DUN_WAIT_VMCNT(0); /* half J+1 landed, J+2 may stay in flight */     \
            DUN_WIDE_BARRIER();                                                                                        \
            if (ISSUE && DUN_EXP_DMA) { /* every wave has read the last fragment of half J: its stage takes half J+3 */ \
                dA.issue(kbeg + ((J) + 3) * HK, smem + cur, wave);                                                     \
                dB.issue(kbeg + ((J) + 3) * HK, smem + cur + WA_HALF, wave);                                           \
            }                                                                                                          \
            const float* a_n = smem + nxt;                                                                             \
            const float* b_n = a_n + WA_HALF;                                                                          \
            if (DUN_EXP_LDS) {                                                                                         \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                read_frag_w<A_KC, BM>(a_n, wr * 64 + mi * 32 + li, 0, h, fa[0][mi]);                                   \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                           \
                read_frag_w<B_KC, WBN>(b_n, wc * 128 + ni * 32 + li, 0, h, fb[0][ni]);                                 \
            }                                                                                                          \
        }                                                                                                              \
        DUN_WIDE_SEP();                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                       \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][mi][e], fb[1][ni][e], acc[mi][ni], 0, 0, 0); \
        if (NEXT) DUN_WIDE_SPREAD((ISSUE ? 6 : 0) + NREADW, 32 / (6 + NREADW));                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        DUN_WAIT_LGKMCNT0();                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        { const int t_ = cur; cur = nxt; nxt = iss; iss = t_; }                                                        \
    } while (0)

    int j = 0;
    for (; j + 3 < nh; ++j) DUN_WIDE_STEP(j, true, true, true);    // steady state
    { DUN_WIDE_STEP(j, true, true, false); ++j; }                  // half nh-3: nothing left to issue
    { DUN_WIDE_STEP(j, true, false, false); ++j; }                 // half nh-2: the last DMA must land
    DUN_WIDE_STEP(j, false, false, false);                         // half nh-1
#undef DUN_WIDE_STEP

    // rank-1 term of the head (C += u v^T, one output; host: never together with bias / ReLU): one more k step with
    // A column u and B row v in the k = 0 half of the lanes, zeros in the other -- acc + u*v per element, like the
    // epilogue of the other kernels computes it
    if (p.U != nullptr) {
        float ua[2], vb[4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) ua[mi] = h == 0 ? p.U[m0 + wr * 64 + mi * 32 + li] : 0.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) vb[ni] = h == 0 ? p.V[n0 + wc * 128 + ni * 32 + li] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[mi], vb[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();  // the epilogue reuses the LDS for the column statistics
    if (p.ksplit > 1) {
        GemmP q = p;  // raw partial sum of this K range
        q.C = p.part + (size_t)blockIdx.y * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.relu = 0; q.R = nullptr; q.U = nullptr; q.colstats = nullptr; q.bnz = nullptr;
        gemm_epilogue<true, 2, 4, true>(q, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    } else if (p.ev_scale != nullptr) {
        gemm_epilogue_eval<2, 4>(p, acc, smem, m0, n0, bn, tid, wr, wc, li, h);
    } else {
        gemm_epilogue<true, 2, 4, true>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    }
}

// (explicit instantiations: the host pass emitted the launch stub of <true, true> only)
template __global__ void gemm_f32_wide_kernel<true, true>(const GemmP);
template __global__ void gemm_f32_wide_kernel<true, false>(const GemmP);
template __global__ void gemm_f32_wide_kernel<false, true>(const GemmP);
template __global__ void gemm_f32_wide_kernel<false, false>(const GemmP);

// The bf16 x 3 kernel on 128 x 256 tiles (wave tile 64 x 128): the split of the A fragments serves four accumulator
// columns instead of two and each fragment read twice the MFMAs -- per MFMA a quarter less vector-ALU work, 5/8 fewer LDS
// reads, a quarter fewer DMA pieces, which is what this issue-bound kernel is short of.  128 accumulator registers
// leave room for ONE set of bf16 operands, B two columns at a time: split A and B[0..1] -> 24 MFMAs -> split B[2..3] ->
// barrier, DMA, fragment reads of the next half -> 24 MFMAs.  Three 24 KiB stages like gemm_f32_wide_kernel.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_wide_kernel(const GemmP p) {
    __shared__ __attribute__((aligned(16))) float smem[3 * WSTAGE];
    const int nwg = p.gridM * p.gridN;  // gridN counts 256-column tiles
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / p.gridN, bn = bid % p.gridN;
    const int m0 = bm * BM, n0 = bn * WBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, h = lane >> 5;

    f32x16 acc[2][4];
    seed_acc<true, 2, 4>(p, acc, m0, n0, wr, wc, li, h);

    // (split-K here, unlike the exact wide kernel: dW = dU^T H is 128 such tiles, four K ranges fill the chip)
    const int kbeg = p.ksplit > 1 ? (int)blockIdx.y * (p.K / p.ksplit) : 0;
    const int nh = (p.ksplit > 1 ? p.K / p.ksplit : p.K) / HK;  // >= 4 (host)
    DmaOpW<A_KC, BM> dA;
    DmaOpW<B_KC, WBN> dB;
    dA.setup(p.A, p.lda, m0, p.M, p.K, wave, lane);
    dB.setup(p.B, p.ldb, n0, p.N, p.K, wave, lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dA.issue(kbeg + j * HK, smem + j * WSTAGE, wave);
        dB.issue(kbeg + j * HK, smem + j * WSTAGE + WA_HALF, wave);
    }
    DUN_WAIT_VMCNT(12);  // half 0 landed (6 loads per half and wave)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::"s"(p.C), "s"(p.bias), "s"(p.R), "s"(p.U), "s"(p.V), "s"(p.colstats), "s"(p.M), "s"(p.N),
                 "s"(p.ldc), "s"(p.ldr), "s"(p.O), "s"(p.ldv), "s"(p.relu), "s"(p.gridM), "s"(p.rinit), "s"(p.bnz),
                 "s"(p.bn_mean), "s"(p.bn_rstd), "s"(p.bnstats), "s"(p.ldz));

    float fa[2][2][4], fb[2][4][4];  // fp32 fragments [k group][tile][e]
#define DUN_X3W_READ(OFF)                                                                                              \
    do {                                                                                                               \
        const float* a_s = smem + (OFF);                                                                               \
        const float* b_s = a_s + WA_HALF;                                                                              \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                           \
                read_frag_w<A_KC, BM>(a_s, wr * 64 + mi * 32 + li, t, h, fa[t][mi]);                                   \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                           \
                read_frag_w<B_KC, WBN>(b_s, wc * 128 + ni * 32 + li, t, h, fb[t][ni]);                                 \
        }                                                                                                              \
    } while (0)
// the two accumulators of column NI (operand set XBI of B) against all six plane pairs, small terms first
#define DUN_X3W_COL(NI, XBI)                                                                                           \
    do {                                                                                                               \
        DUN_X3_MFMA(xa[0][0], XBI[2], acc[0][NI]); DUN_X3_MFMA(xa[1][0], XBI[2], acc[1][NI]);                          \
        DUN_X3_MFMA(xa[0][2], XBI[0], acc[0][NI]); DUN_X3_MFMA(xa[1][2], XBI[0], acc[1][NI]);                          \
        DUN_X3_MFMA(xa[0][1], XBI[1], acc[0][NI]); DUN_X3_MFMA(xa[1][1], XBI[1], acc[1][NI]);                          \
        DUN_X3_MFMA(xa[0][0], XBI[1], acc[0][NI]); DUN_X3_MFMA(xa[1][0], XBI[1], acc[1][NI]);                          \
        DUN_X3_MFMA(xa[0][1], XBI[0], acc[0][NI]); DUN_X3_MFMA(xa[1][1], XBI[0], acc[1][NI]);                          \
        DUN_X3_MFMA(xa[0][0], XBI[0], acc[0][NI]); DUN_X3_MFMA(xa[1][0], XBI[0], acc[1][NI]);                          \
    } while (0)
// the DMA issues and fragment reads behind the barrier spread over column 3's MFMAs (two per MFMA) instead of a burst
#ifndef DUN_X3_NOMEMSPREAD  // (2.63 -> 2.61 ms forward shape, 2.62 -> 2.59 dW shape)
#define DUN_X3W_MEMSPREAD()                                                                                            \
    do {                                                                                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x120, 2, 0);                                                         \
        }                                                                                                              \
    } while (0)
#define DUN_X3W_SEP() do { } while (0)
#else
#define DUN_X3W_MEMSPREAD() do { } while (0)
#define DUN_X3W_SEP() __builtin_amdgcn_sched_barrier(0)
#endif
// issue pattern of a column's 12 MFMAs with one split (44 vector-ALU operations) underneath
#define DUN_X3W_PATTERN()                                                                                              \
    do {                                                                                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                         \
        }                                                                                                              \
    } while (0)
    DUN_X3W_READ(0);

    u32x4 xa[2][3], xb[2][3];
    int cur = 0, nxt = WSTAGE, iss = 2 * WSTAGE;  // stages of half j, j+1, j+2
#define DUN_X3W_STEP(J, NEXT, INFLIGHT, ISSUE)                                                                         \
    do {                                                                                                               \
        DUN_WAIT_LGKMCNT0(); /* fp32 fragments of half J in registers */                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        /* (one split at a time: interleaved, their temporaries spill next to 128 accumulators) */                    \
        x3_split(fa[0][0], fa[1][0], xa[0]); __builtin_amdgcn_sched_barrier(0);                                        \
        x3_split(fa[0][1], fa[1][1], xa[1]); __builtin_amdgcn_sched_barrier(0);                                        \
        x3_split(fb[0][0], fb[1][0], xb[0]); __builtin_amdgcn_sched_barrier(0);                                        \
        /* column 0 with the split of column 1 under it, column 1 with that of column 2 (into the operand registers  \
           column 0 has just been issued from), column 2 with that of column 3.  (Measured and not kept: the fragment  \
           reads moved up in front of column 2 and the next half's column 0 split under column 3 -- spills in the     \
           loop, 2.44 -> 2.49 ms.) */                                                                                 \
        x3_split(fb[0][1], fb[1][1], xb[1]); DUN_X3W_COL(0, xb[0]); DUN_X3W_PATTERN();                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        x3_split(fb[0][2], fb[1][2], xb[0]); DUN_X3W_COL(1, xb[1]); DUN_X3W_PATTERN();                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        x3_split(fb[0][3], fb[1][3], xb[1]); DUN_X3W_COL(2, xb[0]); DUN_X3W_PATTERN();                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (NEXT) {                                                                                                    \
            if (INFLIGHT) DUN_WAIT_VMCNT(6); else DUN_WAIT_VMCNT(0); /* half J+1 landed, J+2 may stay in flight */     \
            __builtin_amdgcn_s_barrier();                            /* everyone has read half J: its stage is free */ \
            if (ISSUE) {                                                                                               \
                dA.issue(kbeg + ((J) + 3) * HK, smem + cur, wave);                                                     \
                dB.issue(kbeg + ((J) + 3) * HK, smem + cur + WA_HALF, wave);                                           \
            }                                                                                                          \
            DUN_X3W_READ(nxt);                                                                                         \
        }                                                                                                              \
        DUN_X3W_SEP();                                                                                                 \
        DUN_X3W_COL(3, xb[1]);                                                                                         \
        if (NEXT) DUN_X3W_MEMSPREAD();                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        { const int t_ = cur; cur = nxt; nxt = iss; iss = t_; }                                                        \
    } while (0)

    int j = 0;
    for (; j + 3 < nh; ++j) DUN_X3W_STEP(j, true, true, true);
    { DUN_X3W_STEP(j, true, true, false); ++j; }
    { DUN_X3W_STEP(j, true, false, false); ++j; }
    DUN_X3W_STEP(j, false, false, false);
#undef DUN_X3W_STEP
#undef DUN_X3W_MEMSPREAD
#undef DUN_X3W_SEP
#undef DUN_X3W_PATTERN
#undef DUN_X3W_COL
#undef DUN_X3W_READ

    if (p.U != nullptr) {  // rank-1 term of the head as one more exact-fp32 MFMA step (see gemm_f32_wide_kernel)
        float ua[2], vb[4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) ua[mi] = h == 0 ? p.U[m0 + wr * 64 + mi * 32 + li] : 0.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) vb[ni] = h == 0 ? p.V[n0 + wc * 128 + ni * 32 + li] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[mi], vb[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
    if (p.ksplit > 1) {
        GemmP q = p;  // raw partial sum of this K range; splitk_sum_kernel adds the ranges in order
        q.C = p.part + (size_t)blockIdx.y * p.M * p.N;
        q.ldc = p.N;
        q.bias = nullptr; q.relu = 0; q.R = nullptr; q.U = nullptr; q.colstats = nullptr; q.bnz = nullptr;
        gemm_epilogue<true, 2, 4, true>(q, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    } else if (p.ev_scale != nullptr) {
        gemm_epilogue_eval<2, 4>(p, acc, smem, m0, n0, bn, tid, wr, wc, li, h);
    } else {
        gemm_epilogue<true, 2, 4, true>(p, acc, smem, m0, n0, bm, tid, wave, wr, wc, li, h);
    }
}

template __global__ void gemm_bf16x3_wide_kernel<true, true>(const GemmP);
template __global__ void gemm_bf16x3_wide_kernel<true, false>(const GemmP);
template __global__ void gemm_bf16x3_wide_kernel<false, true>(const GemmP);
template __global__ void gemm_bf16x3_wide_kernel<false, false>(const GemmP);

// C[m][n] = part[0][m][n] + part[1][m][n] + ... (fixed order: reproducible)
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ part, int ksplit, int M, int N,
                                                         float* __restrict__ C, int ldc) {
    const size_t n4 = (size_t)M * N / 4;  // N % 4 == 0
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
        for (int k = 1; k < ksplit; ++k) v += reinterpret_cast<const f32x4*>(part + (size_t)k * M * N)[i];
        const size_t e = i * 4;
        const int m = (int)(e / N), n = (int)(e % N);
        *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = v;
    }
}

// 0 = exact fp32 MFMA (the product), 1 = the exploratory bf16 x 3 split on LDS-DMA-eligible problems; -1: environment
static int g_gemm_precision = -1;
int gemm_precision() {
    if (g_gemm_precision < 0) {
        const char* e = getenv("DUN_GEMM_BF16X3");
        g_gemm_precision = (e && e[0] == '1') ? 1 : 0;
    }
    return g_gemm_precision;
}

template <bool A_KC, bool B_KC>
static int launch_gemm(const GemmP& p, bool vec, bool dma, bool small, hipStream_t s) {
    const int grid = p.gridM * p.gridN;
    if (dma && p.x3 && p.wide) {
        hipLaunchKernelGGL((gemm_bf16x3_wide_kernel<A_KC, B_KC>), dim3(grid, p.ksplit), dim3(256), 0, s, p);
        if (p.ksplit > 1) {
            const size_t n4 = (size_t)p.M * p.N / 4;
            size_t blocks = (n4 + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.part, p.ksplit, p.M, p.N, p.C, p.ldc);
        }
        DUN_LAUNCH_CHECK();
        return DUN_OK;
    }
    if (dma && p.x3) {
        hipLaunchKernelGGL((gemm_bf16x3_kernel<A_KC, B_KC>), dim3(grid, p.ksplit), dim3(256), 0, s, p);
        if (p.ksplit > 1) {
            const size_t n4 = (size_t)p.M * p.N / 4;
            size_t blocks = (n4 + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.part, p.ksplit, p.M, p.N, p.C, p.ldc);
        }
        DUN_LAUNCH_CHECK();
        return DUN_OK;
    }
    if (p.wide) {
        hipLaunchKernelGGL((gemm_f32_wide_kernel<A_KC, B_KC>), dim3(grid, p.ksplit), dim3(256), 0, s, p);
        if (p.ksplit > 1) {
            DUN_LAUNCH_CHECK();
            const size_t n4 = (size_t)p.M * p.N / 4;
            const int g = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_sum_kernel, dim3(g), dim3(256), 0, s, p.part, p.ksplit, p.M, p.N, p.C, p.ldc);
        }
    } else if (dma && p.ksplit > 1) {
        hipLaunchKernelGGL((gemm_f32_dma4_kernel<A_KC, B_KC>), dim3(grid, p.ksplit), dim3(256), 0, s, p);
        DUN_LAUNCH_CHECK();
        const size_t n4 = (size_t)p.M * p.N / 4;
        const int g = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_sum_kernel, dim3(g), dim3(256), 0, s, p.part, p.ksplit, p.M, p.N, p.C, p.ldc);
    } else if (dma)
        hipLaunchKernelGGL((gemm_f32_dma4_kernel<A_KC, B_KC>), dim3(grid), dim3(256), 0, s, p);
    else {
        const dim3 g2(grid, p.ksplit > 1 ? p.ksplit : 1);
        if (small && vec)
            hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, 1>), g2, dim3(256), 0, s, p);
        else if (small)
            hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, 1>), g2, dim3(256), 0, s, p);
        else if (vec)
            hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, 2>), g2, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, 2>), g2, dim3(256), 0, s, p);
        if (p.ksplit > 1) {
            DUN_LAUNCH_CHECK();
            const size_t n4 = (size_t)p.M * p.N / 4;
            const int g = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_sum_kernel, dim3(g), dim3(256), 0, s, p.part, p.ksplit, p.M, p.N, p.C, p.ldc);
        }
    }
    DUN_LAUNCH_CHECK();
    return DUN_OK;
}

// fewer 128-tiles than half the CUs: take 64x64 tiles (UCI-sized nets are one to four 128-tiles per GEMM)
static bool small_problem(int M, int N) {
    return (long long)ceil_div(M, BM) * ceil_div(N, BN) < 128;
}

// row tiles of an M x N problem = rows of the colstats partial matrix
int gemm_colstat_tiles(int M, int N) { return ceil_div(M, small_problem(M, N) ? 64 : BM); }

int gemm_precision();
// measurement window of dun_gemm_profile_begin / _end (below)
struct GemmProfSlot {
    hipEvent_t a, b;
    dun_gemm_profile_rec rec;
};
static GemmProfSlot* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0, g_prof_lim = 0;  // slots allocated, used, allowed in this window
static bool g_prof_on = false;

static GemmProfSlot* gemm_profile_open(const dun_gemm_desc* d, int kernel, int ksplit, hipStream_t s) {
    if (!g_prof_on || g_prof_n >= g_prof_lim) return nullptr;
    GemmProfSlot* slot = &g_prof[g_prof_n++];
    dun_gemm_profile_rec& r = slot->rec;
    r.M = d->M; r.N = d->N; r.K = d->K;
    r.a_kcontig = d->a_kcontig; r.b_kcontig = d->b_kcontig;
    r.epilogue = (d->bias ? 1 : 0) | (d->relu ? 2 : 0) | (d->R ? 4 : 0) | (d->U ? 8 : 0) | (d->colstats ? 16 : 0) | (d->bnz ? 32 : 0);
    r.kernel = kernel; r.ksplit = ksplit; r.ms = 0.f;
    (void)hipEventRecord(slot->a, s);
    return slot;
}

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
    p.bnz = d->bnz; p.bn_mean = d->bn_mean; p.bn_rstd = d->bn_rstd; p.bnstats = d->bnstats; p.ldz = d->ldz;
    DUN_REQUIRE(d->bnz == nullptr || (d->bn_mean && d->bn_rstd && d->bnstats && d->ldz >= d->N),
                "dun_gemm_f32: bnz needs bn_mean, bn_rstd, bnstats and ldz >= N");
    p.rinit = 0;
    p.wide = 0;
    const bool small = small_problem(d->M, d->N);
    const int tile = small ? 64 : BM;
    p.gridM = ceil_div(d->M, tile);
    p.gridN = ceil_div(d->N, tile);
    p.ev_mean = d->ev_mean; p.ev_scale = d->ev_scale; p.ev_shift = d->ev_shift; p.ev_hw = d->ev_head_w; p.ev_hpart = d->ev_head_part;
    DUN_REQUIRE((long long)p.gridM * p.gridN < (1ll << 31), "dun_gemm_f32: grid too large");
    // float4 staging needs 16-byte aligned rows on both operands; anything else takes the scalar loads
    const bool va = aligned16(d->A) && d->lda % 4 == 0 && (d->a_kcontig ? d->K % 4 == 0 : d->M % 4 == 0);
    const bool vb = aligned16(d->B) && d->ldb % 4 == 0 && (d->b_kcontig ? d->K % 4 == 0 : d->N % 4 == 0);
    const bool vec = va && vb;
    // whole tiles only: LDS-DMA cannot zero-fill, ragged problems take the register-staged kernel
    // ... and the whole-tile kernels address C / R / Z tiles through buffer instructions with 32-bit byte offsets from the
    // matrix base: a matrix of 2 GiB or more (batch >= 262144 at width 2048) takes the register-staged kernel's flat path
    const auto below_2g = [&](int ld) { return ((long long)(d->M - 1) * ld + d->N) * 4 < (1ll << 31); };
    const bool out32 = below_2g(d->ldc) && (!d->R || below_2g(d->ldr)) && (!d->bnz || below_2g(d->ldz));
    // ... and their operands by LDS-DMA with 32-bit byte offsets as well: operands of 2 GiB or more go the same way
    const bool fits32 = (long long)(d->a_kcontig ? d->M : d->K) * d->lda * 4 < (1ll << 31) &&
                        (long long)(d->b_kcontig ? d->N : d->K) * d->ldb * 4 < (1ll << 31);
    const bool dma = !small && vec && out32 && fits32 && d->M % BM == 0 && d->N % BN == 0 && d->K % BK == 0;
    // One workgroup per CU cannot hide its own barriers: an epilogue-free problem with fewer tiles than 2 x 256 CUs and
    // a long K (dW = dU^T H: 256 tiles, K = batch) runs two K ranges per tile and adds them in a fixed order.
    p.ksplit = 1;
    p.part = nullptr;
    const long long tiles = (long long)p.gridM * p.gridN;
    const bool plain = !d->bias && !d->relu && !d->R && !d->U && !d->colstats && !d->bnz;
    if (dma && plain && tiles < 512 && d->K % 64 == 0 && d->K >= 2048 &&
        d->splitk_ws != nullptr && d->splitk_ws_floats >= 2ull * d->M * d->N && aligned16(d->splitk_ws) &&
        aligned16(d->C) && d->ldc % 4 == 0) {
        p.ksplit = 2;
        p.part = d->splitk_ws;
    } else if (!dma && plain && d->splitk_ws != nullptr && aligned16(d->splitk_ws) && aligned16(d->C) &&
               d->ldc % 4 == 0 && d->N % 4 == 0) {
        // register-staged kernels (ragged or small problems, e.g. dW_in = G^T x: 32 tiles, K = batch; the 100-wide
        // UCI nets: 4 tiles, K = 512): one K range per 128 rows of K, as many as fill ~512 workgroups and fit the
        // workspace, at most 16
        long long ks = d->K / 128;
        if (ks > 512 / tiles) ks = 512 / tiles;
        if (ks > 16) ks = 16;
        const long long fit = (long long)(d->splitk_ws_floats / ((size_t)d->M * d->N));
        if (ks > fit) ks = fit;
        if (ks >= 2) {
            p.ksplit = (int)ks;
            p.part = d->splitk_ws;
        }
    }
    {
        // a residual seeds the accumulators (its loads go out in front of the pipeline prologue: dX 3.94 -> 3.90 ms at the
        // wide configuration) wherever nothing runs between the contraction and the add
        const bool can_seed = d->R && !d->bias && !d->relu && p.ksplit == 1;
        if (d->bnz && (d->bias || d->relu || d->colstats || (d->R && !can_seed) || (d->U && d->O != 1))) {
            set_error("dun_gemm_f32: bnz together with bias / ReLU / colstats / a rank-O update with O > 1 is not implemented");
            return DUN_ERR_UNSUPPORTED;
        }
        p.rinit = can_seed ? 1 : 0;
    }
    if (d->ev_scale != nullptr) {
        // the inference epilogue exists in the two LDS-DMA kernels (whole tiles, no split-K)
        if (!(dma && p.ksplit == 1 && d->bias && d->relu && d->ev_mean && d->ev_shift && !d->colstats && !d->bnz && !d->U &&
              (d->ev_head_w == nullptr) == (d->ev_head_part == nullptr))) {
            set_error("dun_gemm_f32: the inference epilogue needs whole 128x128 tiles (>= 128 of them), 16-byte aligned operands, "
                      "bias + ReLU, ev_mean / ev_scale / ev_shift, and no statistics / rank update");
            return DUN_ERR_UNSUPPORTED;
        }
        p.rinit = 0;
    }
    // 128 x 256 tiles where they fill the chip without split-K (the trunk's forward and dX GEMMs): the per-tile
    // prologue / store bubble is paid once per twice the work.  (A residual must seed the accumulators there, or be added
    // by the inference epilogue, which gathers it chunk by chunk.)
    p.x3 = (dma && gemm_precision() == 1) ? 1 : 0;
#ifdef DUN_NO_WIDE_SPLITK  // A/B build
    const bool wide_splitk = p.x3 != 0;
#else
    const bool wide_splitk = true;
#endif
    if (wide_splitk && p.ksplit > 1 && d->N % WBN == 0 && (d->a_kcontig || !d->b_kcontig)) {
        // K ranges on 128 x 256 tiles, as many as fill 512 workgroups and fit the workspace (dW = dU^T H: 128 tiles x 4
        // ranges; 3.79 -> 3.73 ms against two ranges on 128 x 128 tiles: fewer fragment reads and DMA issues per MFMA)
        const long long wtiles = (long long)p.gridM * (d->N / WBN);
        int ks = (int)(512 / wtiles);
        if (ks > 8) ks = 8;
        while (ks > 1 && (d->K % (ks * 64) != 0 || d->splitk_ws_floats < (size_t)ks * d->M * d->N)) --ks;
        if (ks >= 2 && d->K / ks >= 64) {
            p.ksplit = ks;
            p.wide = 1;
            p.gridN = d->N / WBN;
        }
    }
    if (dma && p.ksplit == 1 && d->N % WBN == 0 &&
        (long long)p.gridM * (d->N / WBN) >= 512 && d->K >= 64 && (!d->R || p.rinit || d->ev_scale) &&
        (!d->U || (d->O == 1 && !d->bias && !d->relu)) &&
        (d->a_kcontig || !d->b_kcontig)) {  // (the <M-contiguous, K-contiguous> instantiation spills: none of the trunk's GEMMs)
        p.wide = 1;
        p.gridN = d->N / WBN;
    }
    GemmProfSlot* slot = gemm_profile_open(d, p.x3 ? 3 : (p.wide ? 2 : (dma ? 1 : 0)), p.ksplit, s);
    int rc;
    if (d->a_kcontig)
        rc = d->b_kcontig ? launch_gemm<true, true>(p, vec, dma, small, s) : launch_gemm<true, false>(p, vec, dma, small, s);
    else
        rc = d->b_kcontig ? launch_gemm<false, true>(p, vec, dma, small, s) : launch_gemm<false, false>(p, vec, dma, small, s);
    if (slot != nullptr) (void)hipEventRecord(slot->b, s);
    return rc;
}

}  // namespace dun

extern "C" int dun_gemm_colstat_tiles(int M, int N) { return dun::gemm_colstat_tiles(M, N); }

extern "C" int dun_gemm_precision(int mode) {
    const int prev = dun::gemm_precision();
    if (mode == 0 || mode == 1) dun::g_gemm_precision = mode;
    return prev;
}

extern "C" int dun_gemm_f32(const dun_gemm_desc* desc, dun_stream_t stream) {
    return dun::gemm_f32(desc, static_cast<hipStream_t>(stream));
}

// ---- measurement hook: duration of every GEMM launch INSIDE a real step (bench.py's roofline) ----------------------------
// Between dun_gemm_profile_begin and dun_gemm_profile_end every dun_gemm_f32 call -- the trunk's own included -- is
// bracketed by two HIP events on its stream; _end waits for them and hands back shape, flags and milliseconds per launch.
// Off (no events, no overhead) outside that window.  Host-side state: one window at a time, not thread safe.

extern "C" int dun_gemm_profile_begin(int max_launches) {
    using namespace dun;
    DUN_REQUIRE(!g_prof_on, "dun_gemm_profile_begin: a window is already open");
    DUN_REQUIRE(max_launches > 0 && max_launches <= (1 << 16), "dun_gemm_profile_begin: 1 .. 65536 launches");
    if (g_prof_cap < max_launches) {
        GemmProfSlot* n = static_cast<GemmProfSlot*>(realloc(g_prof, sizeof(GemmProfSlot) * (size_t)max_launches));
        DUN_REQUIRE(n != nullptr, "dun_gemm_profile_begin: out of host memory");
        g_prof = n;
        for (int i = g_prof_cap; i < max_launches; ++i) {
            if (hipEventCreate(&g_prof[i].a) != hipSuccess || hipEventCreate(&g_prof[i].b) != hipSuccess) {
                set_error("dun_gemm_profile_begin: hipEventCreate failed");
                g_prof_cap = i;
                return DUN_ERR_LAUNCH;
            }
        }
        g_prof_cap = max_launches;
    }
    g_prof_n = 0;
    g_prof_lim = max_launches;
    g_prof_on = true;
    return DUN_OK;
}

extern "C" int dun_gemm_profile_end(dun_gemm_profile_rec* out, int cap) {
    using namespace dun;
    if (!g_prof_on) return 0;
    g_prof_on = false;
    int n = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof[i].b) != hipSuccess || hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) != hipSuccess)
            continue;
        if (out != nullptr && n < cap) {
            out[n] = g_prof[i].rec;
            out[n].ms = ms;
            ++n;
        }
    }
    return n;
}
