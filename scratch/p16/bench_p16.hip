// Stand-alone harness of the P16 convolution prototype: correctness against a double-precision host reference on a
// small case, then timing on the layer shapes of the B=32 train step.
#include "conv_p16.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace p16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static int f16_scale_exp(float amax) {
    unsigned b; memcpy(&b, &amax, 4);
    const int E = (b >> 23) & 0xff;
    if (!E) return 0;
    const int e = 141 - E;
    return e > 126 ? 126 : e;
}
static unsigned short f16bits(_Float16 h) { unsigned short u; memcpy(&u, &h, 2); return u; }

struct Case { int B, H, W, Cin, Cout, nsrc; };

static unsigned long long *g_prof = nullptr;
static double g_cyc[3];
template <int KS, int WM, int WN, int WTM, int WTN>
static float run(const Args &a0, int iters, hipStream_t st) {
    using C = Cfg<KS, WM, WN, WTM, WTN>;
    Args a = a0;
    a.ppr = (a.W + 7) / 8;
    a.ppi = a.ppr * ((a.H + 3) / 4);
    a.chunks = (a.ppi + C::PB - 1) / C::PB;
    if (a.CoutP % C::BNT) { printf("bad CoutP\n"); exit(1); }
    auto kern = conv_p16_kernel<KS, WM, WN, WTM, WTN>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    const unsigned grid = (unsigned)(a.B * a.chunks * (a.CoutP / C::BNT));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, st, a);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    if (iters <= 0) return 0.f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, st, a);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    {   // one more launch with the cycle counters on
        if (!g_prof) CK(hipMalloc(&g_prof, (size_t)1 << 24));
        a.prof = g_prof;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, st, a);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hp((size_t)grid * 3);
        CK(hipMemcpy(hp.data(), g_prof, hp.size() * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; ++k) { double s = 0; for (unsigned i = 0; i < grid; ++i) s += (double)hp[i * 3 + k]; g_cyc[k] = s / grid; }
        g_cyc[1] /= (double)(a.Cin / 16) * 9;     // per K-step
    }
    return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
    const char *which = argc > 1 ? argv[1] : "all";
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    const Case cases[] = {
        {2, 20, 40, 64, 128, 1},          // correctness (odd sizes: partial patches), checked in full
        {2, 12, 40, 96, 128, 2},          // two sources (64 + 32)
        {32, 48, 160, 128, 128, 1},       // level3 3x3
        {32, 96, 320, 64, 64, 1},         // level2 3x3
        {32, 24, 80, 256, 256, 1},        // level4 3x3
        {32, 12, 40, 512, 512, 1},        // level5 3x3
        {32, 48, 160, 256, 128, 2},       // neck node: cat 128 + 128
    };
    for (size_t ci = 0; ci < sizeof(cases) / sizeof(cases[0]); ++ci) {
        const Case c = cases[ci];
        const bool check = ci < 2;
        if (!check && !strcmp(which, "check")) continue;
        if (check && !strcmp(which, "time")) continue;
        const int CoutP = (c.Cout + 127) / 128 * 128 >= 128 && c.Cout >= 128 ? (c.Cout + 127) / 128 * 128 : 64;
        const size_t npx = (size_t)c.B * c.H * c.W;
        // sources: split Cin over nsrc tensors (first gets the larger share when uneven: 64 + 32)
        int Cs[4] = {c.Cin, 0, 0, 0};
        if (c.nsrc == 2) { Cs[0] = c.Cin == 96 ? 64 : c.Cin / 2; Cs[1] = c.Cin - Cs[0]; }
        std::vector<float> x(npx * c.Cin), w((size_t)c.Cout * c.Cin * 9);
        const bool zero = getenv("P16_ZERO") != nullptr;
        for (auto &v : x) v = zero ? 0.f : U(rng);
        for (auto &v : w) v = zero ? 0.f : U(rng) * 0.05f;
        if (zero) { x[0] = 1.f; w[0] = 0.05f; }
        float ax = 0.f, aw = 0.f;
        for (float v : x) ax = fmaxf(ax, fabsf(v));
        for (float v : w) aw = fmaxf(aw, fabsf(v));
        const int ea = f16_scale_exp(ax), ew = f16_scale_exp(aw);
        // P16 activations per source
        void *dsrc[4] = {nullptr};
        int coff = 0;
        for (int s = 0; s < c.nsrc; ++s) {
            std::vector<unsigned short> p(npx * Cs[s] * 2);
            for (size_t px = 0; px < npx; ++px)
                for (int ch = 0; ch < Cs[s]; ++ch) {
                    const float v = ldexpf(x[px * c.Cin + coff + ch], ea);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t base = (px * Cs[s] + (ch & ~7)) * 2;
                    p[base + (ch & 7)] = f16bits(hi);
                    p[base + 8 + (ch & 7)] = f16bits(lo);
                }
            CK(hipMalloc(&dsrc[s], p.size() * 2));
            CK(hipMemcpy(dsrc[s], p.data(), p.size() * 2, hipMemcpyHostToDevice));
            coff += Cs[s];
        }
        // weight panel [2][tap][Cin/8][CoutP][8]
        const size_t plane = (size_t)9 * c.Cin * CoutP;
        std::vector<unsigned short> wp(2 * plane, 0);
        for (int n = 0; n < c.Cout; ++n)
            for (int ch = 0; ch < c.Cin; ++ch)
                for (int tap = 0; tap < 9; ++tap) {
                    const float v = ldexpf(w[((size_t)n * c.Cin + ch) * 9 + tap], ew);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t o = (((size_t)tap * (c.Cin >> 3) + (ch >> 3)) * CoutP + n) * 8 + (ch & 7);
                    wp[o] = f16bits(hi);
                    wp[plane + o] = f16bits(lo);
                }
        void *dw;
        CK(hipMalloc(&dw, wp.size() * 2));
        CK(hipMemcpy(dw, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
        float *dout, *dstats;
        CK(hipMalloc(&dout, npx * c.Cout * 4));
        CK(hipMemset(dout, 0xff, npx * c.Cout * 4));
        const int ppi = ((c.W + 7) / 8) * ((c.H + 3) / 4);
        CK(hipMalloc(&dstats, (size_t)c.B * ppi * CoutP * 2 * 4));
        Args a{};
        for (int s = 0; s < c.nsrc; ++s) { a.src[s].p = dsrc[s]; a.src[s].C = Cs[s]; }
        a.nsrc = c.nsrc; a.B = c.B; a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.Cout = c.Cout; a.CoutP = CoutP;
        a.wpk16 = dw; a.out = dout; a.stats = dstats; a.omul = ldexpf(1.f, -ea) * ldexpf(1.f, -ew);
        const int iters = check ? 0 : 20;
        const double gflop = 2.0 * npx * c.Cout * c.Cin * 9 / 1e9;
        struct V { const char *name; float us; double c0, c1, c2; };
        std::vector<V> res;
        auto verify = [&](const char *name) {
            if (!check) return;
            std::vector<float> o(npx * c.Cout);
            CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0;
            for (int b = 0; b < c.B; ++b)
                for (int y = 0; y < c.H; ++y)
                    for (int xx = 0; xx < c.W; ++xx)
                        for (int n = 0; n < c.Cout; ++n) {
                            double s = 0;
                            for (int r = 0; r < 3; ++r)
                                for (int q = 0; q < 3; ++q) {
                                    const int iy = y + r - 1, ix = xx + q - 1;
                                    if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) continue;
                                    const float *xp = &x[(((size_t)b * c.H + iy) * c.W + ix) * c.Cin];
                                    const float *wq = &w[(size_t)n * c.Cin * 9 + r * 3 + q];
                                    for (int ch = 0; ch < c.Cin; ++ch) s += (double)xp[ch] * wq[(size_t)ch * 9];
                                }
                            const double d = fabs(s - o[(((size_t)b * c.H + y) * c.W + xx) * c.Cout + n]);
                            maxerr = fmax(maxerr, d); maxref = fmax(maxref, fabs(s));
                        }
            printf("  check %-18s max|err| %.3e  max|ref| %.3e  rel %.3e %s\n", name, maxerr, maxref, maxerr / maxref,
                   maxerr / maxref < 2e-6 ? "OK" : "FAIL");
            CK(hipMemset(dout, 0xff, npx * c.Cout * 4));
        };
        printf("case B=%d %dx%d Cin=%d(%d src) Cout=%d  %.1f GFLOP\n", c.B, c.H, c.W, c.Cin, c.nsrc, c.Cout, gflop);
        if (CoutP % 128 == 0) {
            { float us_ = run<3, 2, 2, 2, 2>(a, iters, st); res.push_back({"2x2 waves 2x2", us_, g_cyc[0], g_cyc[1], g_cyc[2]}); } verify("2x2w 2x2t");
            { float us_ = run<3, 1, 4, 2, 1>(a, iters, st); res.push_back({"1x4 waves 2x1", us_, g_cyc[0], g_cyc[1], g_cyc[2]}); } verify("1x4w 2x1t");
        }
        { float us_ = run<3, 2, 2, 2, 1>(a, iters, st); res.push_back({"2x2 waves 2x1", us_, g_cyc[0], g_cyc[1], g_cyc[2]}); } verify("2x2w 2x1t");
        { float us_ = run<3, 4, 1, 2, 2>(a, iters, st); res.push_back({"4x1 waves 2x2", us_, g_cyc[0], g_cyc[1], g_cyc[2]}); } verify("4x1w 2x2t");
        { float us_ = run<3, 2, 1, 2, 2>(a, iters, st); res.push_back({"2x1 waves 2x2", us_, g_cyc[0], g_cyc[1], g_cyc[2]}); } verify("2x1w 2x2t");
        { float us_ = run<3, 4, 1, 1, 2>(a, iters, st); res.push_back({"4x1 waves 1x2", us_, g_cyc[0], g_cyc[1], g_cyc[2]}); } verify("4x1w 1x2t");
        if (!check)
            for (auto &r : res) printf("  %-16s %8.1f us  %7.1f TF fp32-equiv  %7.1f TF executed   cycles: prologue %.0f  per K-step %.0f  epilogue %.0f\n", r.name, r.us, gflop / r.us * 1e3, 3 * gflop / r.us * 1e3, r.c0, r.c1, r.c2);
        for (int s = 0; s < c.nsrc; ++s) CK(hipFree(dsrc[s]));
        CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(dstats));
    }
    return 0;
}
