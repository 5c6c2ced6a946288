// Development probe (not part of the library): per-phase clock stamps of the small-net forward kernel on the GPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSN_PROFILE -I dun_amd/csrc scripts/dev/sn_phase_probe.hip -o gpurun_out/sn_probe
#include <vector>
#include "../../dun_amd/csrc/smallnet.hip"
namespace dun { void set_error(const char*, ...) {} }
using namespace dun;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, W = argc > 2 ? atoi(argv[2]) : 100;
    float *H, *Wt, *vec, *Z, *Hout, *st;
    long long* dbg;
    CK(hipMalloc(&H, (size_t)B * W * 4)); CK(hipMalloc(&Hout, (size_t)B * W * 4)); CK(hipMalloc(&Z, (size_t)B * W * 4));
    CK(hipMalloc(&Wt, (size_t)W * W * 4)); CK(hipMalloc(&vec, 8 * W * 4)); CK(hipMalloc(&st, 8 * W * 4));
    CK(hipMalloc(&dbg, 64 * 8 * 32));
    std::vector<float> h((size_t)B * W, 0.5f), w((size_t)W * W, 0.01f), v(8 * W, 1.f);
    CK(hipMemcpy(H, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wt, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(st, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    SnFwdP p; memset(&p, 0, sizeof(p));
    p.Hin = H; p.Wt = Wt; p.bias = vec; p.gamma = vec + W; p.beta = vec + 2 * W; p.Z = Z; p.Hout = Hout;
    p.mean = st; p.rstd = st + W; p.running_mean = st + 2 * W; p.running_var = st + 3 * W;
    p.B = B; p.W = W; p.train = 1; p.residual = 1;
    p.eps = 1e-5f; p.momentum = 0.1f; p.count = B; p.inv_count = 1.0 / B; p.unbias = (double)B / (B - 1.0); p.dbg = dbg;
    for (int it = 0; it < 3; ++it) {
        CK(hipMemset(dbg, 0, 64 * 8 * 32));
        int rc = sn_rpt(B) == 1 ? sn_launch_fwd<1>(p, 0) : sn_rpt(B) == 2 ? sn_launch_fwd<2>(p, 0) : sn_launch_fwd<4>(p, 0);
        if (rc) { printf("launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        long long hb[64 * 32];
        CK(hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost));
        for (int blk = 0; blk < 25 && blk < W / 4; blk += 12) {
            printf("it %d block %2d:", it, blk);
            for (int i = 1; i < 32 && hb[blk * 32 + i]; ++i) printf(" %lld", (hb[blk * 32 + i] - hb[blk * 32 + i - 1]) * 10);
            printf("  ns (total %lld)\n", (hb[blk * 32 + 31] ? 0 : 0) + 0LL);
        }
    }
    // backward kernel of one block (deeper block's dU and weights present)
    float *G, *dUp, *dU, *dact, *gr, *dW;
    CK(hipMalloc(&G, (size_t)B * W * 4)); CK(hipMalloc(&dUp, (size_t)B * W * 4)); CK(hipMalloc(&dU, (size_t)B * W * 4));
    CK(hipMalloc(&dact, (size_t)B * 4)); CK(hipMalloc(&gr, 8 * W * 4)); CK(hipMalloc(&dW, (size_t)W * W * 4));
    CK(hipMemcpy(G, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dUp, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dact, h.data(), (size_t)B * 4, hipMemcpyHostToDevice));
    SnBwdP q; memset(&q, 0, sizeof(q));
    q.B = B; q.W = W; q.O = 1; q.keep = 1; q.G = G; q.dUprev = dUp; q.Wprev = Wt; q.dact = dact; q.w_out = vec;
    q.Z = Z; q.mean = st; q.rstd = st + W; q.gamma = vec + W; q.dgamma = gr; q.dbeta = gr + W; q.dbias = gr + 2 * W;
    q.dU = dU; q.count = B; q.inv_count = 1.0 / B; q.A = H; q.K = W; q.lda = W; q.dW = dW; q.dbg = dbg;
    for (int it = 0; it < 3; ++it) {
        CK(hipMemset(dbg, 0, 64 * 8 * 32));
        if (sn_dispatch_bwd<true, 4, true>(q, 0)) { printf("launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        long long hb[64 * 32];
        CK(hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost));
        for (int blk = 0; blk < 25 && blk < W / 4; blk += 12) {
            printf("bwd it %d block %2d:", it, blk);
            for (int i = 1; i < 32 && hb[blk * 32 + i]; ++i) printf(" %lld", (hb[blk * 32 + i] - hb[blk * 32 + i - 1]) * 10);
            printf("  ns\n");
        }
    }
    return 0;
}
