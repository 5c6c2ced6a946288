// Where do workgroups land (XCC_ID per blockIdx), and what does a record hand-off between two workgroups cost when they
// share an XCD's L2 (agent-scope store, then TCP invalidate + workgroup-scope load) against the agent-scope loads the
// small-net exchange uses?  Bounded spins everywhere.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/dev/xcd_probe scripts/dev/xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void where_kernel(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}

template <int MODE>
__device__ __forceinline__ u32x4 poll_load(__amdgpu_buffer_rsrc_t rs, int off) {
    if (MODE == 0) return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);  // sc1: agent scope
    asm volatile("buffer_inv sc0" ::: "memory");                                   // drop this CU's L1 lines
    if (MODE == 1) return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 1);   // sc0
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
}

// blocks `a` and `b` bounce a tagged record N times; everyone else leaves at once
template <int MODE>
__global__ void pingpong_kernel(unsigned* buf, int a, int b, int n, long long* out) {
    const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    const int lane = threadIdx.x;
    const int mine = me * 1024 + lane * 16, theirs = (1 - me) * 1024 + lane * 16;
    const long long t0 = wall_clock64();
    bool lost = false;
    for (int i = 1; i <= n && !lost; ++i) {
        const unsigned tag = (unsigned)i;
        if (me == 0) {
            const u32x4 v = {tag * 3u, tag, tag * 5u, tag};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine, 0, 16);
        }
        const long long tb = wall_clock64();
        for (;;) {
            asm volatile("" ::: "memory");
            const u32x4 r = poll_load<MODE>(rs, theirs);
            if (__all(r[1] == tag && r[3] == tag)) break;
            if (wall_clock64() - tb > 20000000LL) { lost = true; break; }  // 0.2 s
        }
        if (me == 1 && !lost) {
            const u32x4 v = {tag * 3u, tag, tag * 5u, tag};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine, 0, 16);
        }
    }
    if (lane == 0) { out[2 * me] = wall_clock64() - t0; out[2 * me + 1] = lost ? 1 : 0; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
int run(unsigned* buf, long long* out, int a, int b, int n, const char* what) {
    CK(hipMemset(buf, 0, 4096));
    CK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL(pingpong_kernel<MODE>, dim3(64), dim3(64), 0, 0, buf, a, b, n, out);
    CK(hipDeviceSynchronize());
    long long h[4];
    CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    printf("%-44s blocks %2d,%2d  %s  one-way %.3f us\n", what, a, b, (h[1] || h[3]) ? "LOST (timeout)" : "ok",
           h[0] / 100.0 / (2.0 * n));
    fflush(stdout);
    return 0;
}

int main() {
    unsigned *where, *buf;
    long long* out;
    CK(hipMalloc(&where, 64 * 4)); CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(where_kernel, dim3(64), dim3(512), 0, 0, where);
    CK(hipDeviceSynchronize());
    unsigned h[64];
    CK(hipMemcpy(h, where, sizeof h, hipMemcpyDeviceToHost));
    printf("XCC_ID by blockIdx:");
    for (int i = 0; i < 64; ++i) printf(" %u", h[i] & 15);
    printf("\n");
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>(buf, out, 0, 8, n, "sc1 loads (as shipped), same XCD")) return 1;
        if (run<0>(buf, out, 0, 1, n, "sc1 loads (as shipped), different XCD")) return 1;
        if (run<1>(buf, out, 0, 8, n, "buffer_inv sc0 + sc0 loads, same XCD")) return 1;
        if (run<2>(buf, out, 0, 8, n, "buffer_inv sc0 + plain loads, same XCD")) return 1;
    }
    if (run<1>(buf, out, 0, 1, 20, "buffer_inv sc0 + sc0 loads, different XCD")) return 1;
    return 0;
}
