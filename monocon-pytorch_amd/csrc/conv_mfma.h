// Fused implicit-GEMM convolution for gfx950 on the fp32 MFMA pipe.
//
//   out[b,y,x,n] = act( scale[n] * sum_{src,c,r,s} in_src[b, y*S-P+r, x*S-P+s, c] * W[n, c, r, s]
//                       + bias[n] + residual[b,y,x,n] )
//
// Replaces nn.Conv2d(+torch.cat)+BatchNorm2d(eval)+ReLU(+residual) of the reference's BasicBlock /
// Root / Conv2dBlock / head 3x3 (model/backbone/dla.py:34-51,124-132; dla_neck.py:34-38;
// monocon_heads.py:114-120).
//
// Mapping (M = output pixels, N = output channels, K = taps x input channels):
//   * M is cut into 4x8-pixel patches (one 32-row MFMA tile each); a workgroup owns PB = WM*WTM
//     consecutive patches of ONE image and BNT = WN*WTN*32 output channels.
//   * v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TF peak == the fp32 VALU peak but issued
//     from one wave per SIMD with the VALU left free for address math and the epilogue.
//   * A operand: the patch's input halo (IHxIW pixels x CK channels) is staged once per K-chunk in
//     LDS as [patch][pixel][CK+4] (NHWC => channels contiguous); every lane pulls 4 consecutive
//     channels of "its" pixel with one ds_read_b128, re-used across the 9 taps by address offset.
//   * B operand: weights are pre-packed as [tap][Cin/4][CoutP][4] so that the same lane->k mapping
//     (k-pair {c+j, c+4+j} for MFMA j of a group of 4) is one 16-byte global load per 32-column
//     tile straight from L2 -- the weights never occupy LDS.
//   * virtual concat: up to 4 source tensors are walked chunk by chunk; torch.cat is never
//     materialised.
//   * epilogue: folded-BN scale/shift or conv bias, residual add, ReLU, optional per-(image,
//     channel) sum / sum-of-squares partials (for AttnBN instance statistics / train-mode BN).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvSrc {
    const float *p;
    int C;
};

struct ConvArgs {
    ConvSrc src[4];
    int nsrc;
    int B, Hin, Win, Hout, Wout;
    int Cin;                 // total input channels of the virtual concat
    int Cout, CoutP;         // CoutP: packed/padded column count (multiple of the N tile)
    const float *wpk;        // [k*k][Cin/4][CoutP][4]
    const float *scale;      // [Cout] or null (1)
    const float *bias;       // [Cout] or null (0)
    const float *res;        // NHWC [B,Hout,Wout,res_ld] or null
    int res_ld;
    float *out;              // NHWC, channel stride out_ld, channel offset out_coff
    int out_ld, out_coff;
    int relu;
    float *stats;            // optional [B][chunks][CoutP][2] partial (sum, sumsq) of (v - shift)
    const float *stat_shift; // [Cout] or null
    int ppr, ppi, chunks;    // patches per row / per image, workgroup chunks per image
    int cfg;                 // ConvCfgId workgroup shape (CFG_AUTO = conv_pick_cfg)
};

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
struct ConvCfg {
    static constexpr int PB = WM * WTM;
    static constexpr int BNT = WN * WTN * 32;
    static constexpr int NT = 64 * WM * WN;
    static constexpr int PAD = KS / 2;
    static constexpr int IH = 3 * S + KS;
    static constexpr int IW = 7 * S + KS;
    static constexpr int NPIX = IH * IW;
    static constexpr int CKP = CK + 4;
    static constexpr int LDS_FLOATS = PB * NPIX * CKP + PB * 4 + 2 * WM * BNT;
    static constexpr size_t LDS_BYTES = sizeof(float) * LDS_FLOATS;
};

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(const ConvArgs a) {
    using Cfg = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    constexpr int PB = Cfg::PB, BNT = Cfg::BNT, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CKP = Cfg::CKP;
    constexpr int C4 = CK / 4;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    int *pinfo = reinterpret_cast<int *>(lds + PB * NPIX * CKP);   // [PB][4] = b, oy0, ox0, valid
    float *sred = lds + PB * NPIX * CKP + PB * 4;                    // [WM][BNT][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 5, li = lane & 31;

    const int ntiles = a.CoutP / BNT;
    const int nt = blockIdx.x % ntiles;
    const int mchunk = blockIdx.x / ntiles;
    const int img = mchunk / a.chunks, chunk = mchunk % a.chunks;
    const int n0 = nt * BNT;

    if (tid < PB) {
        const int pp = chunk * PB + tid;
        const int valid = pp < a.ppi;
        const int py = pp / a.ppr, px = pp % a.ppr;
        pinfo[tid * 4 + 0] = img;
        pinfo[tid * 4 + 1] = py * 4;
        pinfo[tid * 4 + 2] = px * 8;
        pinfo[tid * 4 + 3] = valid;
    }

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // A-fragment base offsets (floats) inside the LDS image for this lane
    int a_off[WTM];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
        a_off[tm] = ((wm * WTM + tm) * NPIX + ((li >> 3) * S) * IW + (li & 7) * S) * CKP + 4 * g;

    const int Cin4 = a.Cin >> 2;
    const size_t colP = (size_t)a.CoutP;
    const float *wlane = a.wpk + ((size_t)g * colP + n0 + wn * WTN * 32 + li) * 4;

    // B fragment of step s (= tap * CK/8 + k8) of the K-chunk starting at concat channel kc
    constexpr int K8 = CK / 8, NS = KS * KS * K8;
    auto load_b = [&](f32x4(&dst)[WTN], int kc, int s) {
        const int tap = s / K8, k8 = s % K8;
        const float *wp = wlane + ((size_t)(tap * Cin4 + ((kc + k8 * 8) >> 2)) * colP) * 4;
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn) dst[tn] = *reinterpret_cast<const f32x4 *>(wp + tn * 32 * 4);
    };
    auto load_a = [&](f32x4(&dst)[WTM], int s) {
        const int tap = s / K8, k8 = s % K8;
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
            dst[tm] = *reinterpret_cast<const f32x4 *>(
                &lds[a_off[tm] + ((tap / KS) * IW + (tap % KS)) * CKP + k8 * 8]);
    };
    f32x4 bcur[WTN];
    load_b(bcur, 0, 0);   // weights do not depend on the staged tile: in flight across the barriers

    int kbase = 0;   // channel offset of the current source inside the virtual concat
    for (int si = 0; si < a.nsrc; ++si) {
        const float *sp = a.src[si].p;
        const int Cs = a.src[si].C;
        for (int c0 = 0; c0 < Cs; c0 += CK) {
            __syncthreads();   // previous chunk's fragment reads done (also publishes pinfo)
            // ---- stage [PB][NPIX][CK] input halo, zero-filled outside the image
            constexpr int TOTAL = PB * NPIX * C4;
#pragma unroll 4
            for (int e = tid; e < TOTAL; e += NT) {
                const int c4 = e % C4;
                const int t = e / C4;
                const int pix = t % NPIX;
                const int p = t / NPIX;
                const int iy = pix / IW, ix = pix % IW;
                const int pb = pinfo[p * 4 + 0];
                const int y = pinfo[p * 4 + 1] * S - PAD + iy;
                const int x = pinfo[p * 4 + 2] * S - PAD + ix;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (pinfo[p * 4 + 3] && y >= 0 && y < a.Hin && x >= 0 && x < a.Win)
                    v = *reinterpret_cast<const f32x4 *>(
                        sp + (((size_t)pb * a.Hin + y) * a.Win + x) * Cs + c0 + c4 * 4);
                *reinterpret_cast<f32x4 *>(&lds[(p * NPIX + pix) * CKP + c4 * 4]) = v;
            }
            __syncthreads();
            // ---- MFMA over taps x channel groups of 8; both operands are fetched one step ahead
            //      (explicit register double-buffering: hipcc otherwise issues each weight load
            //      right in front of the MFMA that consumes it and exposes the full L2 latency)
            const int kc = kbase + c0;
            const int kc_next = (kc + CK < a.Cin) ? kc + CK : kc;   // last chunk: harmless re-load
            f32x4 acur[WTM];
            load_a(acur, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 anext[WTM], bnext[WTN];
                if (s + 1 < NS) {
                    load_a(anext, s + 1);
                    load_b(bnext, kc, s + 1);
                } else {
                    load_b(bnext, kc_next, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < WTN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                acur[tm][j], bcur[tn][j], acc[tm][tn], 0, 0, 0);
                if (s + 1 < NS) {
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm) acur[tm] = anext[tm];
                }
#pragma unroll
                for (int tn = 0; tn < WTN; ++tn) bcur[tn] = bnext[tn];
            }
        }
        kbase += Cs;
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float ssum[WTN], ssq[WTN];
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) ssum[tn] = ssq[tn] = 0.f;
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) {
        const int n = n0 + (wn * WTN + tn) * 32 + li;
        const bool nok = n < a.Cout;
        const float sc = (a.scale && nok) ? a.scale[n] : 1.f;
        const float bi = (a.bias && nok) ? a.bias[n] : 0.f;
        const float sh = (a.stat_shift && nok) ? a.stat_shift[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm) {
            const int p = wm * WTM + tm;
            const int pb = pinfo[p * 4 + 0], oy0 = pinfo[p * 4 + 1], ox0 = pinfo[p * 4 + 2];
            const bool pv = pinfo[p * 4 + 3] != 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * g;
                const int y = oy0 + (m >> 3), x = ox0 + (m & 7);
                if (pv && nok && y < a.Hout && x < a.Wout) {
                    const size_t pixel = ((size_t)pb * a.Hout + y) * a.Wout + x;
                    float v = acc[tm][tn][r] * sc + bi;
                    if (a.res) v += a.res[pixel * a.res_ld + n];
                    const float d = v - sh;
                    ssum[tn] += d;
                    ssq[tn] += d * d;
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.out[pixel * a.out_ld + a.out_coff + n] = v;
                }
            }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn) {
            ssum[tn] += __shfl_xor(ssum[tn], 32);
            ssq[tn] += __shfl_xor(ssq[tn], 32);
            if (g == 0) {
                const int nl = (wn * WTN + tn) * 32 + li;
                sred[(wm * BNT + nl) * 2 + 0] = ssum[tn];
                sred[(wm * BNT + nl) * 2 + 1] = ssq[tn];
            }
        }
        __syncthreads();
        for (int nl = tid; nl < BNT; nl += NT) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s1 += sred[(w * BNT + nl) * 2 + 0];
                s2 += sred[(w * BNT + nl) * 2 + 1];
            }
            float *dst = a.stats + (((size_t)img * a.chunks + chunk) * a.CoutP + n0 + nl) * 2;
            dst[0] = s1;
            dst[1] = s2;
        }
    }
}

// ---- host-side tile selection ---------------------------------------------------------
// Workgroup shapes (WM x WN waves, each wave WTM x WTN 32x32 MFMA tiles).  A shape owns
// PB = WM*WTM patches (32 output pixels each) and BNT = WN*WTN*32 output channels.
struct ConvShape {
    int WM, WN, WTM, WTN;
    int PB() const { return WM * WTM; }
    int BNT() const { return WN * WTN * 32; }
};
enum ConvCfgId {
    CFG_AUTO = 0,
    CFG_128x128 = 1,    // 2x2 waves, 2x2 tiles : 128 px x 128 ch
    CFG_256x64 = 2,     // 4x1 waves, 2x2 tiles : 256 px x  64 ch
    CFG_256x32 = 3,     // 4x1 waves, 2x1 tiles : 256 px x  32 ch
    CFG_128x64 = 4,     // 2x2 waves, 2x1 tiles : 128 px x  64 ch
    CFG_128x64m = 5,    // 4x1 waves, 1x2 tiles : 128 px x  64 ch (waves split M only)
    CFG_128x32 = 6,     // 4x1 waves, 1x1 tiles : 128 px x  32 ch
    CFG_64x128 = 7,     // 1x4 waves, 2x1 tiles :  64 px x 128 ch
    CFG_64x64 = 8,      // 2x2 waves, 1x1 tiles :  64 px x  64 ch
    CFG_COUNT = 9
};
inline ConvShape conv_shape(int cfg) {
    switch (cfg) {
        case CFG_128x128: return {2, 2, 2, 2};
        case CFG_256x64: return {4, 1, 2, 2};
        case CFG_256x32: return {4, 1, 2, 1};
        case CFG_128x64: return {2, 2, 2, 1};
        case CFG_128x64m: return {4, 1, 1, 2};
        case CFG_128x32: return {4, 1, 1, 1};
        case CFG_64x128: return {1, 4, 2, 1};
        case CFG_64x64: return {2, 2, 1, 1};
        default: return {0, 0, 0, 0};
    }
}

// Column padding of the packed weights for a layer with Cout columns (every shape's BNT that
// may be chosen for the layer divides it).
inline int conv_ntile(int Cout) { return Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32); }
inline int conv_coutp(int Cout) {
    const int t = conv_ntile(Cout);
    return (Cout + t - 1) / t * t;
}
// K-chunk: 32 when every source is a multiple of 32 channels and the 3x3 is stride 1 (or 1x1).
inline int conv_ck(int ks, int stride, const int *src_c, int nsrc) {
    bool all32 = true;
    for (int i = 0; i < nsrc; ++i) all32 = all32 && (src_c[i] % 32 == 0);
    if (ks == 3 && stride == 2) return 16;
    return all32 ? 32 : 16;
}
// Default shape for a layer (tuned on MI355X): depends on the column count, the
// stride (stride-2 halos are 9x17 pixels per patch, so fewer patches per workgroup keep several
// workgroups resident per CU) and on how many workgroups the launch would have.
int conv_pick_cfg(int Cout, int CoutP, int ks, int stride, int B, int Hout, int Wout);
// patches per workgroup of the shape launch_conv() will use for these arguments
inline int conv_patches_per_block(int cfg) { return conv_shape(cfg).PB(); }

hipError_t launch_conv(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved = nullptr);

}  // namespace mc
