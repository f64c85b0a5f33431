#!/usr/bin/env python
"""Generate scratch/convexp.hip: the library's conv_mfma_kernel with wall-clock phase timers per workgroup
(stage / MFMA / epilogue), to see where a workgroup's time goes on the GPU box."""
import os, re
root = os.path.join(os.path.dirname(__file__), "..")
src = open(os.path.join(root, "monocon-pytorch_amd/csrc/conv_mfma.h")).read()
i = src.index("template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>\n__global__ __launch_bounds__(64 * WM * WN, 3) void conv_mfma_kernel")
j = src.index("// ---- wave-specialised variant")
k = src[i:j]
k = k.replace("void conv_mfma_kernel(const ConvArgs a) {", "void timed_kernel(const ConvArgs a) {\n    const long long t_start = wall_clock64();\n    long long sum_stage = 0, sum_mfma = 0;")
k = k.replace("            if (kbase + c0 > 0) __syncthreads();   // previous chunk's fragment reads done",
              "            if (kbase + c0 > 0) __syncthreads();   // previous chunk's fragment reads done\n            const long long ts0 = wall_clock64();")
k = k.replace("            __syncthreads();\n            // ---- MFMA over taps", "            __syncthreads();\n            const long long ts1 = wall_clock64();\n            sum_stage += ts1 - ts0;\n            // ---- MFMA over taps")
k = k.replace("                for (int tn = 0; tn < WTN; ++tn) bcur[tn] = bnext[tn];\n            }\n        }\n        kbase += Cs;",
              "                for (int tn = 0; tn < WTN; ++tn) bcur[tn] = bnext[tn];\n            }\n            sum_mfma += wall_clock64() - ts1;\n        }\n        kbase += Cs;")
k = k.replace("    conv_epilogue<WM, WN, WTM, WTN, BNT>(a, acc, pinfo, sred, img, n0, wm, wn, g, li);\n    if (a.stats) {",
              "    const long long t_loop = wall_clock64();\n    conv_epilogue<WM, WN, WTM, WTN, BNT>(a, acc, pinfo, sred, img, n0, wm, wn, g, li);\n    asm volatile(\"s_waitcnt vmcnt(0)\");\n    if (tid == 0) {\n        long long *o = g_times + (size_t)blockIdx.x * 8;\n        o[0] = t_start; o[1] = sum_stage; o[2] = sum_mfma; o[3] = t_loop; o[4] = wall_clock64();\n    }\n    if (a.stats) {")
assert k.count("wall_clock64") == 6, k.count("wall_clock64")
host = r'''
template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
static void run(ConvArgs a, double gf, const char *name) {
    using Cfg = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    if (a.CoutP % Cfg::BNT) return;
    auto kern = timed_kernel<KS, S, CK, WM, WN, WTM, WTN>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    const int nb = a.B * a.chunks * (a.CoutP / Cfg::BNT);
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::NT, Cfg::LDS_BYTES);
    long long *dt;
    (void)hipMalloc(&dt, (size_t)nb * 64);
    (void)hipMemset(dt, 0, (size_t)nb * 64);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_times), &dt, sizeof(dt));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    (void)hipEventRecord(e0);
    const int it = 10;
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
    const int nch = a.Cin / CK;
    const double ideal = (double)nch * KS * KS * (CK / 8) * 4 * WTM * WTN * 64 / 2.4e3;   // us of MFMA issue per wave at 2.4 GHz
    printf("%-8s %7.3f ms %6.1f TF | %5d blocks, %d/CU resident, span %6.1f us | per block us: total %6.1f  stage %6.1f  mfma %6.1f (ideal alone %5.1f)  epilogue %5.1f\n",
           name, ms, gf / ms, nb, occ, (t1 - t0) * 0.01, tot / nb * 0.01, st / nb * 0.01, mf / nb * 0.01, ideal, ep / nb * 0.01);
    // start/end histograms (10 bins over the span)
    int hs[10] = {0}, he[10] = {0};
    for (int b = 0; b < nb; ++b) {
        hs[(int)((t[b * 8] - t0) * 10 / (t1 - t0 + 1))]++;
        he[(int)((t[b * 8 + 4] - t0) * 10 / (t1 - t0 + 1))]++;
    }
    printf("         starts:"); for (int i = 0; i < 10; ++i) printf(" %5d", hs[i]);
    printf("\n         ends:  "); for (int i = 0; i < 10; ++i) printf(" %5d", he[i]); printf("\n");
    (void)hipFree(dt);
}

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 512, H = argc > 2 ? atoi(argv[2]) : 12, W = argc > 3 ? atoi(argv[3]) : 40;
    const int B = argc > 4 ? atoi(argv[4]) : 32;
    const int CO = argc > 5 ? atoi(argv[5]) : C;
    ConvArgs a{};
    size_t nin = (size_t)B * H * W * C, nw = (size_t)9 * C * CO, nout = (size_t)B * H * W * CO;
    std::vector<float> hin(nin), hw(nw);
    unsigned s = 1;
    for (auto &v : hin) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (auto &v : hw) { s = s * 1664525u + 1013904223u; v = 0.05f * (((s >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    float *din, *dw, *dout, *dsc;
    (void)hipMalloc(&din, nin * 4); (void)hipMalloc(&dw, nw * 4); (void)hipMalloc(&dout, nout * 4); (void)hipMalloc(&dsc, CO * 4);
    (void)hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsc, hw.data(), CO * 4, hipMemcpyHostToDevice);
    a.src[0] = {din, C}; a.nsrc = 1; a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = C; a.Cout = a.CoutP = CO;
    a.wpk = dw; a.scale = dsc; a.bias = dsc; a.out = dout; a.out_ld = CO; a.relu = 1;
    const double gf = 2.0 * B * H * W * (double)C * CO * 9 / 1e9;
    printf("3x3 s1 C=%d->%d H=%d W=%d B=%d  %.1f GFLOP\n", C, CO, H, W, B, gf);
    run<3, 1, 32, 2, 2, 2, 2>(a, gf, "128x128");
    run<3, 1, 32, 1, 4, 2, 1>(a, gf, "64x128");
    run<3, 1, 32, 4, 1, 1, 2>(a, gf, "128x64m");
    run<3, 1, 32, 2, 2, 1, 1>(a, gf, "64x64");
    return 0;
}
'''
out = ('// GENERATED by scratch/gen_convexp.py from csrc/conv_mfma.h -- phase timers around the library kernel.\n'
       '#include "../monocon-pytorch_amd/csrc/conv_mfma.h"\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n'
       'using namespace mc;\n__device__ long long *g_times;\n\n' + k + host)
open(os.path.join(root, "scratch/convexp.hip"), "w").write(out)
