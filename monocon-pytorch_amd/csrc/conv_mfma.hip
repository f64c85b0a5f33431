// Instantiations + dispatch of the fused MFMA convolution (see conv_mfma.h).
#include "conv_mfma.h"

namespace mc {

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
static hipError_t launch_one(ConvArgs a, hipStream_t st, ConvArgs *resolved) {
    using Cfg = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    if (resolved) *resolved = a;
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<KS, S, CK, WM, WN, WTM, WTN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntiles = a.CoutP / Cfg::BNT;
    dim3 grid((unsigned)(a.B * a.chunks * ntiles));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    return hipGetLastError();
}

// tile families:  T32 = 256 px x 32 ch, T64 = 256 px x 64 ch, T128 = 128 px x 128 ch
template <int KS, int S, int CK>
static hipError_t launch_tile(const ConvArgs &a, hipStream_t st, ConvArgs *resolved) {
    const int nt = a.ntile ? a.ntile : conv_ntile(a.Cout);
    if (nt == 128) return launch_one<KS, S, CK, 2, 2, 2, 2>(a, st, resolved);
    if constexpr (S == 2) {
        // stride-2 halos are 9x17 pixels per patch: keep 4 patches per workgroup so that several
        // workgroups fit a CU's LDS and load / MFMA / store phases of different groups overlap
        if (nt == 64) return launch_one<KS, S, CK, 4, 1, 1, 2>(a, st, resolved);
        return launch_one<KS, S, CK, 4, 1, 1, 1>(a, st, resolved);
    } else {
        if (nt == 64) return launch_one<KS, S, CK, 4, 1, 2, 2>(a, st, resolved);
        return launch_one<KS, S, CK, 4, 1, 2, 1>(a, st, resolved);
    }
}

hipError_t launch_conv(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved) {
    int sc[4];
    for (int i = 0; i < a.nsrc; ++i) sc[i] = a.src[i].C;
    const int ck = conv_ck(ks, stride, sc, a.nsrc);
    for (int i = 0; i < a.nsrc; ++i)
        if (sc[i] % ck) return hipErrorInvalidValue;
    if (a.CoutP % (a.ntile ? a.ntile : conv_ntile(a.Cout))) return hipErrorInvalidValue;
    if (ks == 3 && stride == 1) {
        return ck == 32 ? launch_tile<3, 1, 32>(a, st, resolved) : launch_tile<3, 1, 16>(a, st, resolved);
    } else if (ks == 3 && stride == 2) {
        return launch_tile<3, 2, 16>(a, st, resolved);
    } else if (ks == 1 && stride == 1) {
        return ck == 32 ? launch_tile<1, 1, 32>(a, st, resolved) : launch_tile<1, 1, 16>(a, st, resolved);
    }
    return hipErrorInvalidValue;
}

}  // namespace mc
