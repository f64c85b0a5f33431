// Convolution weight gradient on the fp32 MFMA pipe.
//
//   dW[n, c, r, s] = sum_{b, y, x} dY[b, y, x, n] * X[b, y*S - P + r, x*S - P + s, c]
//
// A GEMM whose reduction dimension is the pixel index: for one pair of pixels (the K=2 of
// v_mfma_f32_32x32x2_f32) the A operand is 32 output channels of dY and the B operand 32 input
// channels of the tap-shifted X -- both contiguous 128-byte rows in NHWC, read from LDS with one
// conflict-free ds_read_b32 each; the dY fragment is reused by all k*k taps.  Each wave keeps the
// k*k 32x32 accumulators of "its" (n-tile, c-tile) for a whole slice of the pixel range
// (split-K across workgroups); slices are written to a partial buffer and summed by a second,
// deterministic pass that also converts to the OIHW layout of the master gradient.
// Replaces autograd's conv2d weight backward for every Conv2d of the reference model.
#include <algorithm>
#include "conv_mfma.h"
#include "train.h"

namespace mc {

template <int KS, int S, int WN, int WC>
struct WgCfg {
    static constexpr int PB = 2;                       // 32-pixel patches staged per barrier
    static constexpr int NB = 32 * WN, CB = 32 * WC;   // channel tiles
    static constexpr int NT = 64 * WN * WC;
    static constexpr int PAD = KS / 2;
    static constexpr int IH = 3 * S + KS, IW = 7 * S + KS, NPIX = IH * IW;
    static constexpr int CBP = CB + 4, NBP = NB + 4;
    static constexpr int LDS_FLOATS = PB * NPIX * CBP + PB * 32 * NBP + PB * 4;
    static constexpr size_t LDS_BYTES = sizeof(float) * LDS_FLOATS;
};

template <int KS, int S, int WN, int WC>
__global__ __launch_bounds__(64 * WN * WC) void wgrad_mfma_kernel(const WgradArgs a) {
    using Cfg = WgCfg<KS, S, WN, WC>;
    constexpr int PB = Cfg::PB, NB = Cfg::NB, CB = Cfg::CB, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CBP = Cfg::CBP, NBP = Cfg::NBP;
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xt = lds;                                   // [PB][NPIX][CBP]
    float *dyt = lds + PB * NPIX * CBP;                // [PB*32][NBP]
    int *pinfo = reinterpret_cast<int *>(dyt + PB * 32 * NBP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WC, wc = wave % WC;
    const int g = lane >> 5, li = lane & 31;
    const int ct = blockIdx.x % a.c_tiles;
    const int nt = (blockIdx.x / a.c_tiles) % a.n_tiles;
    const int ks = blockIdx.x / (a.c_tiles * a.n_tiles);
    const int n0 = nt * NB, c0 = ct * CB;
    const long long G = (long long)a.B * a.groups_per_img;
    const int g_begin = (int)(G * ks / a.ksplit), g_end = (int)(G * (ks + 1) / a.ksplit);

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int soff[5];
    soff[0] = 0;
    for (int i = 0; i < 4; ++i) soff[i + 1] = soff[i] + (i < a.nsrc ? a.src[i].C : 0);

    for (int gi = g_begin; gi < g_end; ++gi) {
        __syncthreads();
        if (tid < PB) {
            const int img = gi / a.groups_per_img;
            const int pp = (gi % a.groups_per_img) * PB + tid;
            pinfo[tid * 4 + 0] = img;
            pinfo[tid * 4 + 1] = (pp / a.ppr) * 4;
            pinfo[tid * 4 + 2] = (pp % a.ppr) * 8;
            pinfo[tid * 4 + 3] = pp < a.ppi;
        }
        __syncthreads();
        // TIMER_STAGE_BEGIN
        // ---- stage X halo tile (virtual concat, zero outside image / beyond Cin)
        constexpr int XC4 = CB / 4, XTOT = PB * NPIX * XC4;
        for (int e = tid; e < XTOT; e += NT) {
            const int c4 = e % XC4, t = e / XC4, pix = t % NPIX, p = t / NPIX;
            const int iy = pix / IW, ix = pix % IW;
            const int y = pinfo[p * 4 + 1] * S - PAD + iy, x = pinfo[p * 4 + 2] * S - PAD + ix;
            const int cg = c0 + c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pinfo[p * 4 + 3] && y >= 0 && y < a.Hin && x >= 0 && x < a.Win && cg < a.Cin) {
                int si = 0;
                while (si < 3 && cg >= soff[si + 1]) ++si;
                v = *reinterpret_cast<const f32x4 *>(a.src[si].p + (((size_t)pinfo[p * 4] * a.Hin + y) * a.Win + x) * a.src[si].C +
                                                     (cg - soff[si]));
            }
            *reinterpret_cast<f32x4 *>(&xt[(p * NPIX + pix) * CBP + c4 * 4]) = v;
        }
        // ---- stage dY tile
        constexpr int NC4 = NB / 4, DTOT = PB * 32 * NC4;
        for (int e = tid; e < DTOT; e += NT) {
            const int n4 = e % NC4, t = e / NC4, m = t % 32, p = t / 32;
            const int y = pinfo[p * 4 + 1] + (m >> 3), x = pinfo[p * 4 + 2] + (m & 7);
            const int n = n0 + n4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pinfo[p * 4 + 3] && y < a.Hout && x < a.Wout && n + 3 < a.dy_ld)
                v = *reinterpret_cast<const f32x4 *>(a.dy + (((size_t)pinfo[p * 4] * a.Hout + y) * a.Wout + x) * a.dy_ld + n);
            *reinterpret_cast<f32x4 *>(&dyt[(p * 32 + m) * NBP + n4 * 4]) = v;
        }
        __syncthreads();
        // TIMER_STAGE_END
#pragma unroll
        for (int p = 0; p < PB; ++p) {
#pragma unroll 4
            for (int kk = 0; kk < 16; ++kk) {
                const int m = 2 * kk + g;
                const float av = dyt[(p * 32 + m) * NBP + wn * 32 + li];
                const float *xb = &xt[(p * NPIX + ((m >> 3) * S) * IW + (m & 7) * S) * CBP + wc * 32 + li];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float bv = xb[((t / KS) * IW + (t % KS)) * CBP];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
        // TIMER_MFMA_END
    }
    // TIMER_EPILOGUE_BEGIN
    // ---- epilogue: partial[ks][tap][n][c];  D row = n, D col (lane) = c
    const int c = c0 + wc * 32 + li;
    if (c < a.Cin) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (n < a.Cout) a.partial[(((size_t)ks * T + t) * a.Cout + n) * a.Cin + c] = acc[t][r];
            }
    }
    // TIMER_KERNEL_END
}

// dW (O,I,kh,kw) = sum_ks partial[ks][tap][n][c]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int ksplit, int T, int Cout,
                                                           int Cin, float *__restrict__ dw) {
    const size_t total = (size_t)T * Cout * Cin;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < ksplit; ++k) s += partial[(size_t)k * total + e];
        const int c = e % Cin, n = (e / Cin) % Cout, t = e / ((size_t)Cin * Cout);
        dw[((size_t)n * Cin + c) * T + t] = s;
    }
}


// ---- 16-input-channel layers (DLA level0 / level1: 16->16 s1, 16->32 s2 at full resolution) ------
// With C = 16 a pixel is one 64-byte NHWC row, so four consecutive pixels are exactly the 64-lane
// operand of v_mfma_f32_16x16x4_f32: A[n][k] = dY[pixel k][n] and B[k][c] = X[pixel k + tap][c] are
// single coalesced dword loads per lane, straight from L1/L2 -- no LDS, no padding of the 16 columns
// up to a 32-wide tile (the generic kernel wastes 3/4 of its MFMA work and all its staging on these
// layers).  A wave walks whole output rows; one buffer descriptor per (input row, tap row) makes the
// top / bottom halo a zero-length buffer and the left / right halo an out-of-range lane offset (only
// in the two peeled edge groups of a row), so the main loop has no address arithmetic beyond SGPR
// adds.  The four waves of a workgroup are summed through LDS into one split-K partial.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int S, int NTN>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const WgradArgs a) {
    constexpr int CIN = 16, COUT = 16 * NTN;
    __shared__ float red[9 * NTN * 4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = lane >> 4, j = lane & 15;
    const int R = a.B * a.Hout;
    const int r_begin = (int)((long long)R * blockIdx.x / gridDim.x);
    const int r_end = (int)((long long)R * (blockIdx.x + 1) / gridDim.x);

    f32x4v acc[9][NTN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) acc[t][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int va = (k * COUT + j) * 4;        // dY lane offset inside a group of 4 output pixels
    const int vx = (k * S * CIN + j) * 4;     // X lane offset relative to the input pixel of output pixel 0
    const float *xsrc = a.src[0].p;

    for (int row = r_begin + wave; row < r_end; row += 4) {
        const int img = row / a.Hout, oy = row - img * a.Hout;
        const __amdgpu_buffer_rsrc_t r_dy = make_rsrc(a.dy + (size_t)row * a.Wout * COUT, (unsigned)(a.Wout * COUT) * 4u);
        __amdgpu_buffer_rsrc_t r_x[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * S + r - 1;
            const bool ok = iy >= 0 && iy < a.Hin;
            r_x[r] = make_rsrc(xsrc + ((size_t)img * a.Hin + (ok ? iy : 0)) * a.Win * CIN, ok ? (unsigned)(a.Win * CIN) * 4u : 0u);
        }
        auto group = [&](int x0, bool edge) {
            float av[NTN], bv[9];
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) av[nt] = buf_load1(r_dy, va + nt * 64, x0 * COUT * 4);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (!edge) {
                        bv[r * 3 + s] = buf_load1(r_x[r], vx + s * CIN * 4, (x0 * S - 1) * CIN * 4);
                    } else {
                        const int px = (x0 + k) * S + s - 1;
                        bv[r * 3 + s] = buf_load1(r_x[r], (px >= 0 && px < a.Win) ? (px * CIN + j) * 4 : BUF_OOB, 0);
                    }
                }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nt], bv[t], acc[t][nt], 0, 0, 0);
        };
        group(0, true);
#pragma unroll 2
        for (int x0 = 4; x0 < a.Wout - 4; x0 += 4) group(x0, false);
        if (a.Wout > 4) group(a.Wout - 4, true);
    }

    // ---- workgroup reduction (wave after wave through one LDS image: fixed order, deterministic).
    //      D layout of 16x16x4: row (n) = 4*(lane>>4) + q, column (c) = lane & 15
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float *dst = &red[(t * NTN + nt) * 4 + q][lane];
                        *dst = w == 0 ? acc[t][nt][q] : *dst + acc[t][nt][q];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < 9 * NTN * 4 * 64; e += 256) {
        const int l = e & 63, idx = e >> 6;
        const int q = idx & 3, nt = (idx >> 2) % NTN, t = (idx >> 2) / NTN;
        const int n = nt * 16 + 4 * (l >> 4) + q, c = l & 15;
        a.partial[(((size_t)blockIdx.x * 9 + t) * COUT + n) * CIN + c] = red[idx][l];
    }
}

static bool wgrad_is_small(const WgradArgs &a, int ks, int stride) {
    return ks == 3 && a.nsrc == 1 && a.Cin == 16 && (a.Cout == 16 || a.Cout == 32) && a.dy_ld == a.Cout &&
           (stride == 1 || stride == 2) && a.Wout % 4 == 0 && a.Wout >= 8 &&
           a.Hout == (a.Hin + 2 - 3) / stride + 1 && a.Wout == (a.Win + 2 - 3) / stride + 1;
}

template <int KS, int S, int WN, int WC>
static hipError_t launch_wg(WgradArgs a, hipStream_t st) {
    using Cfg = WgCfg<KS, S, WN, WC>;
    auto kern = wgrad_mfma_kernel<KS, S, WN, WC>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.ksplit * a.n_tiles * a.c_tiles), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    return hipGetLastError();
}

// shape choice: n-tile 128 x c-tile 32 for wide layers, 64 x 64 for 64-column layers, single-wave tiles for <= 32
static void wgrad_shape(int Cout, int Cin, int *WN, int *WC) {
    if (Cout > 64) { *WN = 4; *WC = 1; }
    else if (Cout > 32 && Cin > 32) { *WN = 2; *WC = 2; }
    else if (Cout > 32) { *WN = 2; *WC = 1; }
    else { *WN = 1; *WC = 1; }
}

void wgrad_plan(WgradArgs &a, int ks, int stride) {
    a.small = 0;
    if (wgrad_is_small(a, ks, stride)) {
        a.small = 1;
        const int rows = a.B * a.Hout;
        a.ksplit = std::min(rows / 4 > 0 ? rows / 4 : 1, 1024);   // >= 4 rows (one per wave) per workgroup
        a.n_tiles = a.c_tiles = 1;
        a.ppr = a.ppi = a.groups_per_img = 0;
        return;
    }
    int WN, WC;
    wgrad_shape(a.Cout, a.Cin, &WN, &WC);
    a.n_tiles = (a.Cout + 32 * WN - 1) / (32 * WN);
    a.c_tiles = (a.Cin + 32 * WC - 1) / (32 * WC);
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.groups_per_img = (a.ppi + 1) / 2;
    const long long G = (long long)a.B * a.groups_per_img;
    int ks_ = 1024 / (a.n_tiles * a.c_tiles);
    if (ks_ < 1) ks_ = 1;
    if (ks_ > G) ks_ = (int)G;
    a.ksplit = ks_;
}
size_t wgrad_partial_floats(const WgradArgs &a, int ks) { return (size_t)a.ksplit * ks * ks * a.Cout * a.Cin; }

hipError_t launch_wgrad(const WgradArgs &a, int ks, int stride, float *dw_oihw, hipStream_t st) {
    prof_last = {2, 2.0 * a.B * a.Hout * a.Wout * (double)a.Cout * a.Cin * ks * ks};
    hipError_t e = hipErrorInvalidValue;
    if (a.small) {
        if (stride == 1 && a.Cout == 16) hipLaunchKernelGGL((wgrad_small_kernel<1, 1>), dim3(a.ksplit), dim3(256), 0, st, a);
        else if (stride == 1) hipLaunchKernelGGL((wgrad_small_kernel<1, 2>), dim3(a.ksplit), dim3(256), 0, st, a);
        else if (a.Cout == 16) hipLaunchKernelGGL((wgrad_small_kernel<2, 1>), dim3(a.ksplit), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wgrad_small_kernel<2, 2>), dim3(a.ksplit), dim3(256), 0, st, a);
        e = hipGetLastError();
    } else {
        int WN, WC;
        wgrad_shape(a.Cout, a.Cin, &WN, &WC);
#define WG_DISPATCH(KS_, S_)                                                     \
    if (WN == 4) e = launch_wg<KS_, S_, 4, 1>(a, st);                            \
    else if (WN == 2 && WC == 2) e = launch_wg<KS_, S_, 2, 2>(a, st);            \
    else if (WN == 2) e = launch_wg<KS_, S_, 2, 1>(a, st);                       \
    else e = launch_wg<KS_, S_, 1, 1>(a, st);
        if (ks == 3 && stride == 1) { WG_DISPATCH(3, 1) }
        else if (ks == 3 && stride == 2) { WG_DISPATCH(3, 2) }
        else if (ks == 1 && stride == 1) { WG_DISPATCH(1, 1) }
#undef WG_DISPATCH
    }
    if (e != hipSuccess) return e;
    const size_t total = (size_t)ks * ks * a.Cout * a.Cin;
    size_t gsz = (total + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gsz), dim3(256), 0, st, a.partial, a.ksplit, ks * ks, a.Cout, a.Cin,
                       dw_oihw);
    return hipGetLastError();
}

}  // namespace mc
