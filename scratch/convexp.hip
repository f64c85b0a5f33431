// Ablation playground for the fused conv kernel (not part of the library): the 2x2-wave 2x2-tile
// 3x3 stride-1 CK=32 kernel with single parts switched off, to see what the MFMA pipe waits for.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/convexp.hip -o scratch/convexp
//   scratch/convexp [C=512] [H=12] [W=40] [B=32]
#include "../monocon-pytorch_amd/csrc/conv_mfma.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mc;

enum { NO_B = 1, NO_A = 2, NO_STAGE = 4, NO_EPI = 8, NO_BAR = 16, TIMING = 32 };
__device__ long long *g_times;
#define NOW() ((FL & TIMING) ? (long long)wall_clock64() : 0ll)

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN, int UBX>
__global__ __launch_bounds__(64 * (WM * WN + 1), 4) void ws_exp_kernel(const ConvArgs a) {
    long long tw_bar = 0, tw_mfma = 0, tp_stage = 0, tp_bar = 0; const long long t_start = wall_clock64();
    using Cfg = ConvCfgWS<KS, S, CK, WM, WN, WTM, WTN>;
    constexpr int PB = Cfg::PB, BNT = Cfg::BNT, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CKP = Cfg::CKP, TILE = Cfg::TILE;
    constexpr int C4 = CK / 4;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    int *pinfo = reinterpret_cast<int *>(lds + 2 * TILE);   // [PB][4] = b, oy0, ox0, valid
    float *sred = lds + 2 * TILE + PB * 4;                    // [WM][BNT][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == WM * WN;
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
    __syncthreads();

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    if (producer) {
        // ---- producer wave: element e = lane + 64*i of the [PB][NPIX][C4] tile; the input pixel of
        //      each element does not depend on the chunk, so it is resolved once.
        constexpr int TOTAL = PB * NPIX * C4;
        constexpr int NIT = (TOTAL + 63) / 64;
        int pidx[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = lane + 64 * i;
            const int t = e / C4;
            const int pix = t % NPIX;
            const int p = (t / NPIX) % PB;
            const int iy = pix / IW, ix = pix % IW;
            const int y = pinfo[p * 4 + 1] * S - PAD + iy;
            const int x = pinfo[p * 4 + 2] * S - PAD + ix;
            const bool ok = e < TOTAL && pinfo[p * 4 + 3] && y >= 0 && y < a.Hin && x >= 0 && x < a.Win;
            pidx[i] = ok ? (pinfo[p * 4 + 0] * a.Hin + y) * a.Win + x : -1;
        }
        constexpr int UBV = UBX & 255;
        constexpr int UB = NIT > UBV ? UBV : NIT;
        if (UBX & 256) __builtin_amdgcn_s_setprio(3);   // loads kept in flight per batch
        int ci = 0;
        for (int si = 0; si < a.nsrc; ++si) {
            const float *sp = a.src[si].p;
            const int Cs = a.src[si].C;
            for (int c0 = 0; c0 < Cs; c0 += CK, ++ci) {
                const long long tp0 = wall_clock64();
                float *dst = lds + (ci & 1) * TILE;
                const float *spc = sp + c0 + (lane % C4) * 4;
#pragma unroll
                for (int i0 = 0; i0 < NIT; i0 += UB) {
                    f32x4 v[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int i = i0 + u;
                        if (i < NIT) {   // unconditional load (clamped pixel) + select, see conv_mfma_kernel
                            v[u] = *reinterpret_cast<const f32x4 *>(spc + (size_t)(pidx[i] < 0 ? 0 : pidx[i]) * Cs);
                            if (pidx[i] < 0) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int i = i0 + u;
                        if (i < NIT) {
                            const int e = lane + 64 * i;
                            if (64 * i + 63 < TOTAL || e < TOTAL)
                                *reinterpret_cast<f32x4 *>(&dst[(e / C4) * CKP + (e % C4) * 4]) = v[u];
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
                const long long tp1 = wall_clock64();
                __syncthreads();   // chunk ci is published; consumers are done with chunk ci-1
                tp_stage += tp1 - tp0; tp_bar += wall_clock64() - tp1;
            }
        }
    } else {
        int a_off[WTM];
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
            a_off[tm] = ((wm * WTM + tm) * NPIX + ((li >> 3) * S) * IW + (li & 7) * S) * CKP + 4 * g;

        const int Cin4 = a.Cin >> 2;
        const size_t colP = (size_t)a.CoutP;
        const float *wlane = a.wpk + ((size_t)g * colP + n0 + wn * WTN * 32 + li) * 4;

        constexpr int K8 = CK / 8, NS = KS * KS * K8;
        auto load_b = [&](f32x4(&dst)[WTN], int kc, int s) {
            const int tap = s / K8, k8 = s % K8;
            const float *wp = wlane + ((size_t)(tap * Cin4 + ((kc + k8 * 8) >> 2)) * colP) * 4;
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn) dst[tn] = *reinterpret_cast<const f32x4 *>(wp + tn * 32 * 4);
        };
        f32x4 bcur[WTN];
        load_b(bcur, 0, 0);
        const int nch = a.Cin / CK;
        __syncthreads();   // chunk 0 staged
        for (int ci = 0; ci < nch; ++ci) {
            const long long tc0 = wall_clock64();
            const float *tile = lds + (ci & 1) * TILE;
            auto load_a = [&](f32x4(&dst)[WTM], int s) {
                const int tap = s / K8, k8 = s % K8;
#pragma unroll
                for (int tm = 0; tm < WTM; ++tm)
                    dst[tm] = *reinterpret_cast<const f32x4 *>(
                        &tile[a_off[tm] + ((tap / KS) * IW + (tap % KS)) * CKP + k8 * 8]);
            };
            const int kc = ci * CK;
            const int kc_next = (ci + 1 < nch) ? kc + CK : kc;
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
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < WTN; ++tn)
                            if (!(UBX & 512)) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                acur[tm][j], bcur[tn][j], acc[tm][tn], 0, 0, 0);
                if (s + 1 < NS) {
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm) acur[tm] = anext[tm];
                }
#pragma unroll
                for (int tn = 0; tn < WTN; ++tn) bcur[tn] = bnext[tn];
            }
            const long long tc1 = wall_clock64();
            if (ci + 1 < nch) __syncthreads();   // chunk ci+1 staged, chunk ci released
            tw_mfma += tc1 - tc0; tw_bar += wall_clock64() - tc1;
        }

        // ---- epilogue (identical to conv_mfma_kernel)
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
        }
    }
    if (a.stats) {
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
    asm volatile("s_waitcnt vmcnt(0)");
    if (lane == 0 && (wave == 0 || producer)) {
        long long *o = g_times + (size_t)blockIdx.x * 8 + (producer ? 4 : 0);
        if (producer) { o[0] = tp_stage; o[1] = tp_bar; }
        else { o[0] = t_start; o[1] = wall_clock64(); o[2] = tw_mfma; o[3] = tw_bar; }
    }
}

template <int FL>
__global__ __launch_bounds__(256) void exp_kernel(const ConvArgs a) {
    constexpr int KS = 3, S = 1, CK = 32, WM = 2, WN = 2, WTM = 2, WTN = 2;
    using Cfg = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    constexpr int PB = Cfg::PB, BNT = Cfg::BNT, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CKP = Cfg::CKP;
    constexpr int C4 = CK / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int *pinfo = reinterpret_cast<int *>(lds + PB * NPIX * CKP);
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
    for (int tm = 0; tm < WTM; ++tm)
        for (int tn = 0; tn < WTN; ++tn)
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    int a_off[WTM];
    for (int tm = 0; tm < WTM; ++tm)
        a_off[tm] = ((wm * WTM + tm) * NPIX + ((li >> 3) * S) * IW + (li & 7) * S) * CKP + 4 * g;
    const int Cin4 = a.Cin >> 2;
    const size_t colP = (size_t)a.CoutP;
    const float *wlane = a.wpk + ((size_t)g * colP + n0 + wn * WTN * 32 + li) * 4;
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
    load_b(bcur, 0, 0);
    const long long t_start = NOW();
    long long sum_stage = 0, sum_mfma = 0, t_first = 0;
    int kbase = 0;
    for (int si = 0; si < a.nsrc; ++si) {
        const float *sp = a.src[si].p;
        const int Cs = a.src[si].C;
        for (int c0 = 0; c0 < Cs; c0 += CK) {
            if (!(FL & NO_BAR) || c0 == 0) __syncthreads();
            const long long ts0 = NOW();
            if (!(FL & NO_STAGE) || c0 == 0) {
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
                    const bool ok = pinfo[p * 4 + 3] && y >= 0 && y < a.Hin && x >= 0 && x < a.Win;
                    const size_t pixel = ok ? ((size_t)pb * a.Hin + y) * a.Win + x : 0;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(sp + pixel * Cs + c0 + c4 * 4);
                    if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4 *>(&lds[(p * NPIX + pix) * CKP + c4 * 4]) = v;
                }
            }
            if (!(FL & NO_BAR) || c0 == 0) __syncthreads();
            const long long ts1 = NOW();
            sum_stage += ts1 - ts0;
            if (t_first == 0) t_first = ts1;
            const int kc = kbase + c0;
            const int kc_next = (kc + CK < a.Cin) ? kc + CK : kc;
            f32x4 acur[WTM];
            load_a(acur, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 anext[WTM], bnext[WTN];
                if (s + 1 < NS) {
                    if (!(FL & NO_A)) load_a(anext, s + 1);
                    if (!(FL & NO_B)) load_b(bnext, kc, s + 1);
                } else {
                    if (!(FL & NO_B)) load_b(bnext, kc_next, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < WTN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                acur[tm][j], bcur[tn][j], acc[tm][tn], 0, 0, 0);
                if (s + 1 < NS && !(FL & NO_A)) {
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm) acur[tm] = anext[tm];
                }
                if (!(FL & NO_B)) {
#pragma unroll
                    for (int tn = 0; tn < WTN; ++tn) bcur[tn] = bnext[tn];
                }
            }
            sum_mfma += NOW() - ts1;
        }
        kbase += Cs;
    }
    const long long t_loop = NOW();
    if (FL & NO_EPI) {
        float s = 0.f;
        for (int tm = 0; tm < WTM; ++tm)
            for (int tn = 0; tn < WTN; ++tn)
                for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
        if (s == 123.456f) a.out[tid] = s;
        return;
    }
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) {
        const int n = n0 + (wn * WTN + tn) * 32 + li;
        const bool nok = n < a.Cout;
        const float sc = (a.scale && nok) ? a.scale[n] : 1.f;
        const float bi = (a.bias && nok) ? a.bias[n] : 0.f;
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
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.out[pixel * a.out_ld + a.out_coff + n] = v;
                }
            }
        }
    }
    if (FL & TIMING) {
        asm volatile("s_waitcnt vmcnt(0)");
        const long long t_end = NOW();
        if (tid == 0) {
            long long *o = g_times + (size_t)blockIdx.x * 8;
            o[0] = t_start; o[1] = t_first; o[2] = sum_stage; o[3] = sum_mfma; o[4] = t_loop; o[5] = t_end;
            unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); o[6] = hw;
        }
    }
}

template <int FL>
static float run(ConvArgs a, int iters) {
    using Cfg = ConvCfg<3, 1, 32, 2, 2, 2, 2>;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    auto kern = exp_kernel<FL>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    dim3 grid(a.B * a.chunks * (a.CoutP / Cfg::BNT));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / iters;
}


template <int UBX>
static void run_ws(ConvArgs a, double gf) {
    using Cfg = ConvCfgWS<3, 1, 32, 2, 2, 2, 2>;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    auto kern = ws_exp_kernel<3, 1, 32, 2, 2, 2, 2, UBX>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    const int nb = a.B * a.chunks * (a.CoutP / Cfg::BNT);
    { int occ = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::NT, Cfg::LDS_BYTES); printf("WS occupancy (blocks/CU) = %d, LDS %zu B\n", occ, (size_t)Cfg::LDS_BYTES); }
    long long *dt;
    hipMalloc(&dt, (size_t)nb * 64);
    hipMemset(dt, 0, (size_t)nb * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_times), &dt, sizeof(dt));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(nb), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    std::vector<long long> t((size_t)nb * 8);
    hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = t[0], t1 = 0;
    double tot = 0, mf = 0, bw = 0, ps = 0, pb = 0;
    for (int b = 0; b < nb; ++b) {
        if (t[b * 8] < t0) t0 = t[b * 8];
        if (t[b * 8 + 1] > t1) t1 = t[b * 8 + 1];
        tot += t[b * 8 + 1] - t[b * 8]; mf += t[b * 8 + 2]; bw += t[b * 8 + 3]; ps += t[b * 8 + 4]; pb += t[b * 8 + 5];
    }
    printf("WS UB=%d(+256 prio, +512 no mfma): %.3f ms %.1f TF, %d blocks; span %.1f us; per block avg (us): total %.1f  consumer mfma %.1f  consumer barrier-wait %.1f  producer stage %.1f  producer barrier-wait %.1f\n",
           UBX, ms, gf / ms, nb, (t1 - t0) * 0.01, tot / nb * 0.01, mf / nb * 0.01, bw / nb * 0.01, ps / nb * 0.01, pb / nb * 0.01);
    hipFree(dt);
}

int main(int argc, char **argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 512, H = argc > 2 ? atoi(argv[2]) : 12, W = argc > 3 ? atoi(argv[3]) : 40;
    const int B = argc > 4 ? atoi(argv[4]) : 32;
    ConvArgs a{};
    size_t nin = (size_t)B * H * W * C, nw = (size_t)9 * C * C;
    std::vector<float> hin(nin), hw(nw);
    unsigned s = 1;
    for (auto &v : hin) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (auto &v : hw) { s = s * 1664525u + 1013904223u; v = 0.05f * (((s >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    float *din, *dw, *dout, *dsc;
    hipMalloc(&din, nin * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dout, nin * 4); hipMalloc(&dsc, C * 4);
    hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsc, hw.data(), C * 4, hipMemcpyHostToDevice);
    a.src[0] = {din, C}; a.nsrc = 1; a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = C; a.Cout = a.CoutP = C;
    a.wpk = dw; a.scale = dsc; a.bias = dsc; a.out = dout; a.out_ld = C; a.relu = 1;
    const double gf = 2.0 * B * H * W * (double)C * C * 9 / 1e9;
    const int it = 20;
    struct { const char *n; float ms; } r[] = {
        {"full", run<0>(a, it)},
        {"no B loads", run<NO_B>(a, it)},
        {"no A loads", run<NO_A>(a, it)},
        {"no staging", run<NO_STAGE>(a, it)},
        {"no staging, no barriers", run<NO_STAGE | NO_BAR>(a, it)},
        {"no epilogue", run<NO_EPI>(a, it)},
        {"no A, no B", run<NO_A | NO_B>(a, it)},
        {"MFMA only", run<NO_A | NO_B | NO_STAGE | NO_BAR | NO_EPI>(a, it)},
    };
    {
        using Cfg = ConvCfg<3, 1, 32, 2, 2, 2, 2>;
        const int ppr = (W + 7) / 8, ppi = ppr * ((H + 3) / 4), chunks = (ppi + Cfg::PB - 1) / Cfg::PB;
        const int nb = B * chunks * (C / Cfg::BNT);
        long long *dt;
        hipMalloc(&dt, (size_t)nb * 8 * 8);
        hipMemset(dt, 0, (size_t)nb * 8 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_times), &dt, sizeof(dt));
        const float ms = run<TIMING>(a, 1);
        std::vector<long long> t((size_t)nb * 8);
        hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = t[0], t1 = 0;
        for (int b = 0; b < nb; ++b) { if (t[b * 8] < t0) t0 = t[b * 8]; if (t[b * 8 + 5] > t1) t1 = t[b * 8 + 5]; }
        double a_first = 0, a_stage = 0, a_mfma = 0, a_epi = 0, a_tot = 0;
        for (int b = 0; b < nb; ++b) {
            a_first += t[b * 8 + 1] - t[b * 8]; a_stage += t[b * 8 + 2]; a_mfma += t[b * 8 + 3];
            a_epi += t[b * 8 + 5] - t[b * 8 + 4]; a_tot += t[b * 8 + 5] - t[b * 8];
        }
        printf("timing kernel %.3f ms, %d blocks; span %.1f us; per block avg (us): total %.1f  first-stage %.1f  all-stages %.1f  mfma %.1f  epilogue %.1f\n",
               ms, nb, (t1 - t0) * 0.01, a_tot / nb * 0.01, a_first / nb * 0.01, a_stage / nb * 0.01, a_mfma / nb * 0.01, a_epi / nb * 0.01);
        // start-time histogram in 20 us bins and end-time histogram
        const int NBIN = 48; int hs[NBIN] = {0}, he[NBIN] = {0};
        for (int b = 0; b < nb; ++b) {
            int i0 = (int)((t[b * 8] - t0) / 2000), i1 = (int)((t[b * 8 + 5] - t0) / 2000);
            if (i0 < NBIN) hs[i0]++; if (i1 < NBIN) he[i1]++;
        }
        printf("starts/20us:"); for (int i = 0; i < NBIN; ++i) printf(" %d", hs[i]); printf("\n");
        printf("ends/20us:  "); for (int i = 0; i < NBIN; ++i) printf(" %d", he[i]); printf("\n");
    }
    run_ws<8>(a, gf); run_ws<16>(a, gf); run_ws<16 + 256>(a, gf); run_ws<8 + 512>(a, gf);
    printf("C=%d H=%d W=%d B=%d  %.1f GFLOP\n", C, H, W, B, gf);
    for (auto &x : r) printf("%-28s %8.3f ms  %6.1f TF\n", x.n, x.ms, gf / x.ms);
    return 0;
}
