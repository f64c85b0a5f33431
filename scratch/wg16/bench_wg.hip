// Stand-alone timing harness of the library's fp16-split weight-gradient kernel (wgrad_bf16_kernel<..., SPL = 2>) on the
// 3x3 stride-1 layer shapes of the B = 32 train step.  Built in several variants (-DMC_EXP_NO_MFMA / NO_STORE / NO_FETCH:
// results WRONG by construction) to see which phase the time belongs to.
#include "../../monocon-pytorch_amd/csrc/wgrad_bf16.hip"
#include "../../monocon-pytorch_amd/csrc/wgrad_pipe.hip"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace mc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef WG_KERNEL
#define WG_KERNEL wgrad_bf16_kernel
#endif

struct Case { int B, H, W, Cin, Cout, ks; };

static float run_pipe(WgradArgs a, int ks, int iters) {
    const int tile = wgrad_pipe_tile(a, ks, 1);
    if (!tile) return -1.f;
    wgrad_pipe_plan(a, tile);
    CK(hipMalloc(&a.partial, (size_t)a.ksplit * ks * ks * a.Cout * a.Cin * 4));
    for (int i = 0; i < 2; ++i) CK(launch_wgrad_pipe(a, ks, 0));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) CK(launch_wgrad_pipe(a, ks, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(a.partial));
    return ms / iters * 1e3f;
}

template <int KS, int WN, int WC>
static float run(WgradArgs a, int iters) {
    using Cfg = WgB16Cfg<KS, 1, WN, WC, 2>;
    a.n_tiles = (a.Cout + Cfg::NB - 1) / Cfg::NB;
    a.c_tiles = (a.Cin + Cfg::CB - 1) / Cfg::CB;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.pb = 1;
    a.groups_per_img = a.ppi;
    const long long G = (long long)a.B * a.groups_per_img;
    int blocks = 512;
    if (const char *e = getenv("WG_BLOCKS")) blocks = atoi(e);
    int ks_ = blocks * 4 / (WN * WC) / (a.n_tiles * a.c_tiles);
    if (ks_ < 1) ks_ = 1;
    if (ks_ > G) ks_ = (int)G;
    a.ksplit = ks_;
    CK(hipMalloc(&a.partial, (size_t)a.ksplit * KS * KS * a.Cout * a.Cin * 4));
    auto kern = WG_KERNEL<KS, 1, WN, WC, 2, false, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES));
    const int nb = a.ksplit * a.n_tiles * a.c_tiles;
#ifdef MC_EXP_PHASE_DELAY
    {
        int dly = 0;
        if (const char *e = getenv("WG_DELAY")) dly = atoi(e);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_delay), &dly, sizeof dly));
        int mode = 0;
        if (const char *e = getenv("WG_MODE")) mode = atoi(e);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_mode), &mode, sizeof mode));
        static unsigned *dbg = nullptr;
        if (!dbg) {
            CK(hipMalloc(&dbg, 4096 * 4));
            CK(hipMemset(dbg, 0xff, 4096 * 4));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lds_dbg), &dbg, sizeof dbg));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
            CK(hipDeviceSynchronize());
            std::vector<unsigned> h(nb);
            CK(hipMemcpy(h.data(), dbg, nb * 4, hipMemcpyDeviceToHost));
            int nz = 0, nz_hi = 0;
            for (int i = 0; i < nb; ++i) { nz += (h[i] & 0xfff) != 0; nz_hi += (h[i] & 0xfff) != 0 && i >= nb / 2; }
            printf("LDS_ALLOC: %d of %d blocks with a non-zero base (%d of them in the upper half of the grid); block 0 %08x, block %d %08x\n", nz, nb, nz_hi, h[0], nb - 1, h[nb - 1]);
            unsigned *null_ = nullptr;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lds_dbg), &null_, sizeof null_));
        }
    }
#endif
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipFree(a.partial));
    return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
    const Case cases[] = {{32, 96, 320, 64, 64, 3}, {32, 48, 160, 128, 128, 3}, {32, 24, 80, 256, 256, 3}, {32, 12, 40, 512, 512, 3},
                          {32, 48, 160, 256, 128, 3}, {32, 96, 320, 128, 64, 3}, {32, 96, 320, 64, 576, 3}, {32, 48, 160, 448, 128, 1}};
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (const Case &c : cases) {
        const size_t nx = (size_t)c.B * c.H * c.W * c.Cin, nd_ = (size_t)c.B * c.H * c.W * c.Cout;
        std::vector<float> hx(nx), hd(nd_);
        float mx = 0.f, md = 0.f;
        // (a cheap generator is enough: ReLU-like activations, zero-mean gradients)
        const size_t T = (size_t)1 << 22;
        for (size_t i = 0; i < nx; ++i) {
            if (i < T) { const float v = nd(rng); hx[i] = v > 0.f ? v : 0.f; } else hx[i] = hx[i & (T - 1)];
            mx = fmaxf(mx, hx[i]);
        }
        for (size_t i = 0; i < nd_; ++i) {
            hd[i] = i < T ? nd(rng) * 1e-3f : hd[i & (T - 1)];
            md = fmaxf(md, fabsf(hd[i]));
        }
        float *dx, *dd;
        unsigned *am;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dd, nd_ * 4)); CK(hipMalloc(&am, 8));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dd, hd.data(), nd_ * 4, hipMemcpyHostToDevice));
        unsigned hb[2];
        memcpy(&hb[0], &mx, 4); memcpy(&hb[1], &md, 4);
        CK(hipMemcpy(am, hb, 8, hipMemcpyHostToDevice));
        WgradArgs a{};
        a.src[0].p = dx; a.src[0].C = c.Cin; a.nsrc = 1;
        a.B = c.B; a.Hin = a.Hout = c.H; a.Win = a.Wout = c.W; a.Cin = c.Cin; a.Cout = c.Cout;
        a.dy = dd; a.dy_ld = c.Cout;
        a.prec = 3;
        a.amax_x[0] = am; a.amax_dy = am + 1;
        float us = -1.f;
        if (getenv("WG_PIPE")) us = run_pipe(a, c.ks, iters);
        if (us < 0.f) us = c.ks == 3 ? run<3, 2, 2>(a, iters) : run<1, 2, 2>(a, iters);
        const double gf = 2.0 * c.B * c.H * c.W * (double)c.Cin * c.Cout * c.ks * c.ks * 3;
        printf("wgrad k%d %4d -> %4d @ %3dx%-3d  %8.1f us  %7.1f TF (3 products)\n", c.ks, c.Cin, c.Cout, c.H, c.W, us, gf / us * 1e-6);
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dd)); CK(hipFree(am));
    }
    return 0;
}
