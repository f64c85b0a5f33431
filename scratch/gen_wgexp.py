#!/usr/bin/env python
"""Generate scratch/wgexp.hip: the library's wgrad_mfma_kernel with wall-clock phase timers per workgroup."""
import os
root = os.path.join(os.path.dirname(__file__), "..")
src = open(os.path.join(root, "monocon-pytorch_amd/csrc/wgrad_mfma.hip")).read()
i = src.index("template <int KS, int S, int WN, int WC>\nstruct WgCfg")
j = src.index("// dW (O,I,kh,kw) = sum_ks partial")
k = src[i:j]
k = k.replace("void wgrad_mfma_kernel(const WgradArgs a) {", "void wg_timed(const WgradArgs a) {\n    const long long t_start = wall_clock64();\n    long long sum_stage = 0, sum_mfma = 0;")
assert "TIMER_STAGE_BEGIN" in k and "TIMER_STAGE_END" in k and "TIMER_MFMA_END" in k, "markers missing in wgrad kernel"
k = k.replace("// TIMER_STAGE_BEGIN", "const long long ts0 = wall_clock64();")
k = k.replace("// TIMER_STAGE_END", "const long long ts1 = wall_clock64(); sum_stage += ts1 - ts0;")
k = k.replace("// TIMER_MFMA_END", "sum_mfma += wall_clock64() - ts1;")
k = k.replace("// TIMER_EPILOGUE_BEGIN", "const long long t_loop = wall_clock64();")
k = k.replace("// TIMER_KERNEL_END", "asm volatile(\"s_waitcnt vmcnt(0)\");\n    if (threadIdx.x == 0) { long long *o = g_times + (size_t)blockIdx.x * 8; o[0] = t_start; o[1] = sum_stage; o[2] = sum_mfma; o[3] = t_loop; o[4] = wall_clock64(); }")
host = r'''
template <int KS, int S, int WN, int WC>
static void run(WgradArgs a, double gf, const char *name) {
    using Cfg = WgCfg<KS, S, WN, WC>;
    a.n_tiles = (a.Cout + Cfg::NB - 1) / Cfg::NB;
    a.c_tiles = (a.Cin + Cfg::CB - 1) / Cfg::CB;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.groups_per_img = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    const long long G = (long long)a.B * a.groups_per_img;
    int ks_ = WG_TARGET_BLOCKS / (a.n_tiles * a.c_tiles);
    if (ks_ < 1) ks_ = 1;
    if (ks_ > G) ks_ = (int)G;
    a.ksplit = ks_;
    (void)hipMalloc(&a.partial, (size_t)a.ksplit * KS * KS * a.Cout * a.Cin * 4);
    auto kern = wg_timed<KS, S, WN, WC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    const int nb = a.ksplit * a.n_tiles * a.c_tiles;
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::NT, Cfg::LDS_BYTES);
    long long *dt;
    (void)hipMalloc(&dt, (size_t)nb * 64);
    (void)hipMemset(dt, 0, (size_t)nb * 64);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_times), &dt, sizeof(dt));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    (void)hipEventRecord(e0);
    const int it = 5;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    std::vector<long long> t((size_t)nb * 8);
    (void)hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = t[0], t1 = 0;
    double tot = 0, st = 0, mf = 0, ep = 0;
    for (int b = 0; b < nb; ++b) {
        if (t[b * 8] < t0) t0 = t[b * 8];
        if (t[b * 8 + 4] > t1) t1 = t[b * 8 + 4];
        tot += t[b * 8 + 4] - t[b * 8]; st += t[b * 8 + 1]; mf += t[b * 8 + 2]; ep += t[b * 8 + 4] - t[b * 8 + 3];
    }
    const double rounds = (double)G / a.ksplit;
    const double ideal = rounds * Cfg::PB * 16 * KS * KS * 64 / 2.4e3 * WG_TILES_PER_WAVE;
    printf("%-10s %7.3f ms %6.1f TF | %5d blocks (ksplit %d), %d/CU resident, LDS %zu KB, span %6.1f us | per block us: total %6.1f  stage %6.1f  mfma %6.1f (ideal alone %5.1f)  epilogue %5.1f\n",
           name, ms, gf / ms, nb, a.ksplit, occ, Cfg::LDS_BYTES / 1024, (t1 - t0) * 0.01, tot / nb * 0.01, st / nb * 0.01, mf / nb * 0.01, ideal, ep / nb * 0.01);
    (void)hipFree(dt);
    (void)hipFree(a.partial);
}

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 48, W = argc > 3 ? atoi(argv[3]) : 160;
    const int B = argc > 4 ? atoi(argv[4]) : 32;
    const int CO = argc > 5 ? atoi(argv[5]) : C;
    WgradArgs a{};
    size_t nin = (size_t)B * H * W * C, nout = (size_t)B * H * W * CO;
    std::vector<float> hin(nin), hd(nout);
    unsigned s = 1;
    for (auto &v : hin) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (auto &v : hd) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float *din, *ddy;
    (void)hipMalloc(&din, nin * 4); (void)hipMalloc(&ddy, nout * 4);
    (void)hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(ddy, hd.data(), nout * 4, hipMemcpyHostToDevice);
    a.src[0] = {din, C}; a.nsrc = 1; a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = C; a.Cout = CO;
    a.dy = ddy; a.dy_ld = CO;
    const double gf = 2.0 * B * H * W * (double)C * CO * 9 / 1e9;
    printf("wgrad 3x3 s1 C=%d->%d H=%d W=%d B=%d  %.1f GFLOP\n", C, CO, H, W, B, gf);
    WG_RUNS
    return 0;
}
'''
import re
cfgs = re.findall(r"launch_wg<KS_, S_, (\d+), (\d+)>", src)
runs = "".join('    if (%s) run<3, 1, %s, %s>(a, gf, "n%dxc%d");\n' % ("CO >= %d" % (32 * int(n)) if int(n) > 1 else "true", n, c, 32 * int(n), 32 * int(c)) for n, c in dict.fromkeys(cfgs))
tiles = "1"
tb = re.search(r"int ks_ = (\d+) / \(a.n_tiles \* a.c_tiles\);", src).group(1)
host = host.replace("WG_RUNS", runs).replace("WG_TARGET_BLOCKS", tb).replace("WG_TILES_PER_WAVE", tiles)
out = ('// GENERATED by scratch/gen_wgexp.py from csrc/wgrad_mfma.hip -- phase timers around the library kernel.\n'
       '#include "../monocon-pytorch_amd/csrc/conv_mfma.h"\n#include "../monocon-pytorch_amd/csrc/train.h"\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n'
       'using namespace mc;\n__device__ long long *g_times;\n\n' + k + host)
open(os.path.join(root, "scratch/wgexp.hip"), "w").write(out)
