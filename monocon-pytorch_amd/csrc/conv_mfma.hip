// Instantiations + dispatch of the fused MFMA convolution (see conv_mfma.h).
#include <cstdlib>
#include "conv_mfma.h"

namespace mc {

thread_local ProfLast prof_last = {0, 0.0, 0.0};
static ProfLast conv_cost(const ConvArgs &a, int ks) {
    const double px_in = (double)a.B * a.Hin * a.Win, px_out = (double)a.B * a.Hout * a.Wout;
    const double taps = win_h(ks) * win_w(ks);
    // algorithmic bytes: the input, the output (+ the residual / the gradient accumulated into), the weights, and for a
    // backward-statistics launch the forward's y (and z where the ReLU mask is the stored activation's)
    const int extra = (a.res ? 1 : 0) + (a.bm_y ? 1 : 0) + ((a.bm_y && a.bm_relu == 1 && !a.bm_zbits) ? 1 : 0);      // (a bit-packed mask is 1/32 of a map: not counted)
    return {1, 2.0 * px_out * a.Cout * a.Cin * taps,
            4.0 * (px_in * a.Cin + px_out * a.Cout * (1 + extra) + taps * a.Cin * a.Cout)};
}

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN, bool BM = false>
static hipError_t launch_one(ConvArgs a, hipStream_t st, ConvArgs *resolved) {
    using Cfg = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    if constexpr (!BM && S == 1 && (KS == 3 || KS == 1)) {
        // backward-statistics epilogue (ConvArgs::bm_y): its own instantiation, so that every other launch keeps the
        // lean epilogue (the y / z loads cost ~25 VGPRs)
        if (a.bm_y) return launch_one<KS, S, CK, WM, WN, WTM, WTN, true>(a, st, resolved);
    } else if constexpr (!BM) {
        if (a.bm_y) return hipErrorInvalidValue;
    }
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    if (a.CoutP % Cfg::BNT) return hipErrorInvalidValue;
    if (resolved) *resolved = a;
    static DynLdsOnce attr_set;
    auto kern = conv_mfma_kernel<KS, S, CK, WM, WN, WTM, WTN, BM>;
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(kern), (int)(Cfg::LDS_BYTES));
        if (e != hipSuccess) return e;
    }
    const int ntiles = a.CoutP / Cfg::BNT;
    dim3 grid((unsigned)(a.B * a.chunks * ntiles));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    return hipGetLastError();
}

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
static hipError_t launch_one_ws(ConvArgs a, hipStream_t st, ConvArgs *resolved) {
    using Cfg = ConvCfgWS<KS, S, CK, WM, WN, WTM, WTN>;
    if (Cfg::LDS_BYTES > 160 * 1024) return hipErrorInvalidValue;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    if (a.CoutP % Cfg::BNT) return hipErrorInvalidValue;
    if (resolved) *resolved = a;
    static DynLdsOnce attr_set;
    auto kern = conv_mfma_ws_kernel<KS, S, CK, WM, WN, WTM, WTN>;
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(kern), (int)(Cfg::LDS_BYTES));
        if (e != hipSuccess) return e;
    }
    const int ntiles = a.CoutP / Cfg::BNT;
    dim3 grid((unsigned)(a.B * a.chunks * ntiles));
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    return hipGetLastError();
}

template <int KS, int S, int CK>
static hipError_t launch_shape(const ConvArgs &a, hipStream_t st, ConvArgs *resolved) {
    if constexpr (KS == 3 || KS == 1) if ((a.cfg & CFG_WS) && !a.bm_y) {   // (the WS variant has no bm epilogue)
        switch (a.cfg & ~CFG_WS) {
            case CFG_128x128: return launch_one_ws<KS, S, CK, 2, 2, 2, 2>(a, st, resolved);
            case CFG_128x64: return launch_one_ws<KS, S, CK, 2, 2, 2, 1>(a, st, resolved);
            case CFG_128x64m: return launch_one_ws<KS, S, CK, 4, 1, 1, 2>(a, st, resolved);
            case CFG_128x32: return launch_one_ws<KS, S, CK, 4, 1, 1, 1>(a, st, resolved);
            case CFG_64x128: return launch_one_ws<KS, S, CK, 1, 4, 2, 1>(a, st, resolved);
            case CFG_64x64: return launch_one_ws<KS, S, CK, 2, 2, 1, 1>(a, st, resolved);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.cfg & ~CFG_WS) {
        case CFG_128x128: return launch_one<KS, S, CK, 2, 2, 2, 2>(a, st, resolved);
        case CFG_128x64: return launch_one<KS, S, CK, 2, 2, 2, 1>(a, st, resolved);
        case CFG_128x64m: return launch_one<KS, S, CK, 4, 1, 1, 2>(a, st, resolved);
        case CFG_128x32: return launch_one<KS, S, CK, 4, 1, 1, 1>(a, st, resolved);
        case CFG_64x128: return launch_one<KS, S, CK, 1, 4, 2, 1>(a, st, resolved);
        case CFG_64x64: return launch_one<KS, S, CK, 2, 2, 1, 1>(a, st, resolved);
        default: return hipErrorInvalidValue;
    }
}

int conv_pick_cfg(int Cout, int CoutP, int ks, int stride, int B, int Hout, int Wout) {
    // Measured on MI355X at B=32 (scratch/tune_conv.py, profiles/r1_conv_shapes.txt): the shapes
    // are within ~10 % of each other; small 64-pixel groups win where the K loop is short (1x1)
    // because more, shorter workgroups balance better across the 256 CUs.
    (void)B; (void)Hout; (void)Wout; (void)CoutP;
    const int nt = conv_ntile(Cout);
    if (nt == 128) return ks == 1 ? CFG_64x128 : CFG_128x128;
    if (nt == 64) return (ks == 1 || stride == 2) ? CFG_64x64 : CFG_128x64m;
    return CFG_128x32;
}

bool conv_lazy_capable(const ConvArgs &a, int ks, int stride) {
    if (a.prec != 3 || a.bm_y) return false;
    if (conv_thin_ok(a, ks, stride) || conv_small_lazy_ok(a, ks, stride)) return true;
    return conv_bf16_ok(a, ks, stride) && (ks == 3 || ks == 1);
}

hipError_t launch_conv(const ConvArgs &a_in, int ks, int stride, hipStream_t st, ConvArgs *resolved) {
    ConvArgs a = a_in;
    bool lazy = false;
    for (int i = 0; i < a.nsrc; ++i) lazy |= a.src[i].la != nullptr;
    const bool dense_out = a.o_px == 0;
    if (dense_out) { a.o_px = a.out_ld; a.o_row = a.Wout * a.out_ld; a.o_img = a.Hout * a.Wout * a.out_ld; }
    if (a.r_px == 0) { a.r_px = a.res_ld; a.r_row = a.Wout * a.res_ld; a.r_img = a.Hout * a.Wout * a.res_ld; }
    if (a.cfg == CFG_SMALL && !dense_out) return hipErrorInvalidValue;
    int sc[4];
    for (int i = 0; i < a.nsrc; ++i) sc[i] = a.src[i].C;
    const int ck = conv_ck(ks, stride, sc, a.nsrc);
    for (int i = 0; i < a.nsrc; ++i)
        if (sc[i] % ck) return hipErrorInvalidValue;
    if (a.cfg == CFG_SMALL) {
        if (!conv_small_ok(a, ks, stride)) return hipErrorInvalidValue;
        if (lazy && !conv_thin_ok(a, ks, stride) && !conv_small_lazy_ok(a, ks, stride)) return hipErrorInvalidValue;
        a.ppr = (a.Wout + 7) / 8;
        a.ppi = a.ppr * ((a.Hout + 3) / 4);
        a.chunks = a.Hout;
        if (resolved) *resolved = a;
        prof_last = conv_cost(a, ks);
        return launch_conv_small(a, stride, st);
    }
    if (a.cfg == CFG_AUTO) a.cfg = conv_pick_cfg(a.Cout, a.CoutP, ks, stride, a.B, a.Hout, a.Wout);
    prof_last = conv_cost(a, ks);
    if (a.cfg & CFG_WRES) {
        // MONOCON_HIP_WRES=0: A/B switch, every launch takes the tiling its shape bits name
        static const bool wres_on = [] { const char *e = std::getenv("MONOCON_HIP_WRES"); return !e || std::atoi(e) != 0; }();
        if (wres_on && conv_wres_ok(a, ks, stride)) return launch_conv_wres(a, ks, stride, st, resolved);
        a.cfg &= ~CFG_WRES;
    }
    if (a.prec >= 1 && conv_bf16_ok(a, ks, stride)) return launch_conv_bf16(a, ks, stride, st, resolved);
    if (lazy) return hipErrorInvalidValue;      // (nor do the fp32 MFMA kernels)
    if (ks == 3 && stride == 1) {
        return ck == 32 ? launch_shape<3, 1, 32>(a, st, resolved) : launch_shape<3, 1, 16>(a, st, resolved);
    } else if (ks == 3 && stride == 2) {
        return launch_shape<3, 2, 16>(a, st, resolved);
    } else if (ks == 1 && stride == 1) {
        return ck == 32 ? launch_shape<1, 1, 32>(a, st, resolved) : launch_shape<1, 1, 16>(a, st, resolved);
    } else if (stride == 1 && (ks == 12 || ks == 21 || ks == 22)) {   // stride-2 data-gradient parity classes
        if (a.cfg & CFG_WS) a.cfg &= ~CFG_WS;
        if (ks == 12) return ck == 32 ? launch_shape<12, 1, 32>(a, st, resolved) : launch_shape<12, 1, 16>(a, st, resolved);
        if (ks == 21) return ck == 32 ? launch_shape<21, 1, 32>(a, st, resolved) : launch_shape<21, 1, 16>(a, st, resolved);
        return ck == 32 ? launch_shape<22, 1, 32>(a, st, resolved) : launch_shape<22, 1, 16>(a, st, resolved);
    }
    return hipErrorInvalidValue;
}

}  // namespace mc
