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
#include <cstdlib>
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
    static constexpr int LDS_FLOATS = PB * NPIX * CBP + PB * 32 * NBP;
    static constexpr size_t LDS_BYTES = sizeof(float) * LDS_FLOATS;
};

template <int KS, int S, int WN, int WC>
__global__ __launch_bounds__(64 * WN * WC, 2) void wgrad_mfma_kernel(const WgradArgs a) {
    using Cfg = WgCfg<KS, S, WN, WC>;
    constexpr int PB = Cfg::PB, NB = Cfg::NB, CB = Cfg::CB, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CBP = Cfg::CBP, NBP = Cfg::NBP;
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xt = lds;                                   // [PB][NPIX][CBP]
    float *dyt = lds + PB * NPIX * CBP;                // [PB*32][NBP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WC, wc = wave % WC;
    const int g = lane >> 5, li = lane & 31;
    const int bid = xcd_order(blockIdx.x, gridDim.x);      // the tiles of one split-K slice read the same pixels
    const int ct = bid % a.c_tiles;
    const int nt = (bid / a.c_tiles) % a.n_tiles;
    const int ks = bid / (a.c_tiles * a.n_tiles);
    const int n0 = nt * NB, c0 = ct * CB;
    const long long G = (long long)a.B * a.groups_per_img;
    const int g_begin = (int)(G * ks / a.ksplit), g_end = (int)(G * (ks + 1) / a.ksplit);

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- the c-tile lies inside ONE source of the virtual concat (wgrad_plan picks CB so)
    int si = 0, cbase = 0;
    while (si + 1 < a.nsrc && c0 >= cbase + a.src[si].C) { cbase += a.src[si].C; ++si; }
    const int Cs = a.src[si].C;
    const float *xsrc = a.src[si].p;
    const int cs0 = c0 - cbase;                       // first channel of the tile inside the source

    // ---- staging plan (see conv_mfma.h: buffer loads, lane offset = patch base (SGPR) + static part,
    //      no per-element branches).  Per patch: X element e = tid + NT*i of [NPIX][CB/4], dY element
    //      of [32][NB/4]; a thread keeps its channel group across i and across the patches.
    //      Rows above / below the image fall outside the per-image buffer by themselves; columns left
    //      / right of it would alias the neighbouring row and are masked per element (x_ix / d_mx;
    //      statically dead elements carry a column that is never valid).
    constexpr int XC4 = CB / 4, XP = NPIX * XC4, NIX = (XP + NT - 1) / NT;
    constexpr int NC4 = NB / 4, DP = 32 * NC4, NID = (DP + NT - 1) / NT;
    static_assert(NT % XC4 == 0 && NT % NC4 == 0, "static channel group per thread");
    constexpr int DEAD = -(1 << 24);
    const int xc4 = tid % XC4, dn4 = tid % NC4;
    const bool xc_ok = cs0 + xc4 * 4 < Cs && c0 + xc4 * 4 < a.Cin;
    const bool dn_ok = n0 + dn4 * 4 + 3 < a.dy_ld;
    int x_stat[NIX], x_ix[NIX];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int e = tid + NT * i, pix = (e / XC4) % NPIX;
        const int iy = pix / IW, ix = pix % IW;
        x_ix[i] = (xc_ok && e < XP) ? ix - PAD : DEAD;
        x_stat[i] = (((iy - PAD) * a.Win + ix - PAD) * Cs + cs0 + xc4 * 4) * 4;
    }
    int d_stat[NID], d_mx[NID];
#pragma unroll
    for (int i = 0; i < NID; ++i) {
        const int e = tid + NT * i, m = (e / NC4) % 32;
        d_mx[i] = (dn_ok && e < DP) ? (m & 7) : -DEAD;
        d_stat[i] = (((m >> 3) * a.Wout + (m & 7)) * a.dy_ld + n0 + dn4 * 4) * 4;
    }
    float *x_dst = xt + (tid / XC4) * CBP + xc4 * 4;
    float *d_dst = dyt + (tid / NC4) * NBP + dn4 * 4;

    f32x4 xv[PB][NIX], dv[PB][NID];   // without PF only [0] is live
    // loads of patch p of pixel group gi into registers (with PF: in flight across the MFMA phase of
    // group gi-1; the register-hungry shapes fetch and store patch by patch instead)
    constexpr bool PF = S == 1 && NT >= 128;
    auto fetch = [&](int gi, int p) {
        const int img = gi / a.groups_per_img;
        const int pp = (gi - img * a.groups_per_img) * PB + p;
        const __amdgpu_buffer_rsrc_t r_x =
            make_rsrc(xsrc + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
        const __amdgpu_buffer_rsrc_t r_d =
            make_rsrc(a.dy + (size_t)img * a.Hout * a.Wout * a.dy_ld, (unsigned)(a.Hout * a.Wout * a.dy_ld) * 4u);
        const bool valid = pp < a.ppi;
        const int oy = (pp / a.ppr) * 4, ox = valid ? (pp % a.ppr) * 8 : DEAD;   // SGPRs
        const int xb = ((oy * S) * a.Win + ox * S) * Cs * 4;
        const int db = (oy * a.Wout + ox) * a.dy_ld * 4;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int xx = ox * S + x_ix[i];
            xv[PF ? p : 0][i] = buf_load4(r_x, (xx >= 0 && xx < a.Win) ? xb + x_stat[i] : BUF_OOB, 0);
        }
#pragma unroll
        for (int i = 0; i < NID; ++i) {
            const int xx = ox + d_mx[i];
            dv[PF ? p : 0][i] = buf_load4(r_d, (xx >= 0 && xx < a.Wout) ? db + d_stat[i] : BUF_OOB, 0);
        }
    };
    auto store = [&](int p) {
#pragma unroll
        for (int i = 0; i < NIX; ++i)
            if (NT * (i + 1) <= XP || tid + NT * i < XP)
                *reinterpret_cast<f32x4 *>(x_dst + (p * NPIX + i * (NT / XC4)) * CBP) = xv[PF ? p : 0][i];
#pragma unroll
        for (int i = 0; i < NID; ++i)
            if (NT * (i + 1) <= DP || tid + NT * i < DP)
                *reinterpret_cast<f32x4 *>(d_dst + (p * 32 + i * (NT / NC4)) * NBP) = dv[PF ? p : 0][i];
    };

    // per-lane fragment bases: pixel pair kk of a patch = pixels m = 2*kk + g
    const float *a_base = dyt + g * NBP + wn * 32 + li;
    const float *b_base = xt + (g * S) * CBP + wc * 32 + li;

    if (PF && g_begin < g_end) {
#pragma unroll
        for (int p = 0; p < PB; ++p) fetch(g_begin, p);
    }
    for (int gi = g_begin; gi < g_end; ++gi) {
        __syncthreads();   // fragment reads of the previous group are done
        // TIMER_STAGE_BEGIN
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            if (!PF) fetch(gi, p);
            store(p);
        }
        __syncthreads();
        if (PF && gi + 1 < g_end) {
#pragma unroll
            for (int p = 0; p < PB; ++p) fetch(gi + 1, p);
        }
        // TIMER_STAGE_END
#pragma unroll
        for (int p = 0; p < PB; ++p) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float av = a_base[(p * 32 + 2 * kk) * NBP];
                const float *xb = b_base + (p * NPIX + ((kk >> 2) * S) * IW + (2 * (kk & 3)) * S) * CBP;
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
    if (c < a.Cin && c - cbase < Cs) {
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
// The layers with few weights have the deepest split (64x64x9 weights: 512 slices), so the slices
// are spread over KL lanes of the workgroup (fixed order: lane kl sums slices kl, kl+KL, ...; lane 0
// then adds the KL lane sums in order) instead of one long dependent chain per element.
template <int KL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int ksplit, int T, int Cout,
                                                           int Cin, float *__restrict__ dw) {
    constexpr int EL = 256 / KL;                      // float4 columns per workgroup
    __shared__ f32x4 red[KL > 1 ? KL : 1][EL];
    const size_t total = (size_t)T * Cout * Cin;      // multiple of 4 (Cin is)
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    const size_t e = ((size_t)blockIdx.x * EL + el) * 4;
    const bool live = e < total;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float *p = partial + e + (size_t)kl * total;
#pragma unroll 4
        for (int k = kl; k < ksplit; k += KL, p += (size_t)KL * total) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    if (KL > 1) {
        red[kl][el] = s;
        __syncthreads();
        if (kl != 0) return;
        for (int j = 1; j < KL; ++j) {
            const f32x4 v = red[j][el];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    if (!live) return;
    const int c = e % Cin, n = (e / Cin) % Cout, t = e / ((size_t)Cin * Cout);
    float *d = dw + ((size_t)n * Cin + c) * T + t;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[(size_t)q * T] = s[q];
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

#ifndef WGS_PD
#define WGS_PD 3
#endif
// LZ: X is a lazy tensor (ConvSrc::la in conv_mfma.h: raw conv output + BatchNorm coefficients); a lane always holds channel
// lane & 15, so its two coefficients stay in registers and max(fma(y, la, lb), 0) is formed in front of the MFMAs (fp32
// operands here: no operand scale).  Padding stays 0: `cap`.
template <int S, int NTN, bool LZ = false>
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
    [[maybe_unused]] float lzA = 0.f, lzB = 0.f;
    if constexpr (LZ) { lzA = a.src[0].la[j]; lzB = a.src[0].lb[j]; }

    for (int row = r_begin + wave; row < r_end; row += 4) {
        const int img = row / a.Hout, oy = row - img * a.Hout;
        const __amdgpu_buffer_rsrc_t r_dy = make_rsrc(a.dy + (size_t)row * a.Wout * COUT, (unsigned)(a.Wout * COUT) * 4u);
        __amdgpu_buffer_rsrc_t r_x[3];
        [[maybe_unused]] float rowcap[3];      // LZ: +inf for an input row inside the image, 0 for a padding row
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * S + r - 1;
            const bool ok = iy >= 0 && iy < a.Hin;
            r_x[r] = make_rsrc(xsrc + ((size_t)img * a.Hin + (ok ? iy : 0)) * a.Win * CIN, ok ? (unsigned)(a.Win * CIN) * 4u : 0u);
            rowcap[r] = ok ? __builtin_inff() : 0.f;
        }
        [[maybe_unused]] float ecap[3] = {0.f, 0.f, 0.f};      // LZ, edge groups: 0 where this lane's column is padding
        auto fetch = [&](int x0, bool edge, float (&v)[NTN + 9]) {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) v[nt] = buf_load1(r_dy, va + nt * 64, x0 * COUT * 4);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (!edge) {
                        v[NTN + r * 3 + s] = buf_load1(r_x[r], vx + s * CIN * 4, (x0 * S - 1) * CIN * 4);
                    } else {
                        const int px = (x0 + k) * S + s - 1;
                        v[NTN + r * 3 + s] = buf_load1(r_x[r], (px >= 0 && px < a.Win) ? (px * CIN + j) * 4 : BUF_OOB, 0);
                        if constexpr (LZ) ecap[s] = (px >= 0 && px < a.Win) ? __builtin_inff() : 0.f;
                    }
                }
        };
        auto mma = [&](const float (&v)[NTN + 9], bool edge = false) {
            float x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if constexpr (LZ) x[t] = lazy_act(v[NTN + t], lzA, lzB, edge ? fminf(rowcap[t / 3], ecap[t % 3]) : rowcap[t / 3]);
                else x[t] = v[NTN + t];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[nt], x[t], acc[t][nt], 0, 0, 0);
        };
        // the groups of a row run as a software pipeline PD groups deep: the 9 + NTN dword loads of group g + PD are in
        // flight while the MFMAs of group g issue (without it every group paid one full memory latency: the kernel sat at
        // 21 % of the fp32 MFMA rate with the matrix pipe idle most of the time)
        constexpr int PD = WGS_PD;
        float ring[PD][NTN + 9], ev[NTN + 9];
        const int n = (a.Wout - 8) / 4;               // interior groups: output pixels 4 .. Wout - 5
        fetch(0, true, ev);
        mma(ev, true);
        int g = 0;
        if (n >= 2 * PD) {
            // (no conditional code between the fetches and the MFMAs of the steady state: the compiler's wait counts are
            // path-insensitive, one branch around a fetch and every MFMA waits for ALL loads in flight)
#pragma unroll
            for (int d = 0; d < PD; ++d) fetch(4 + 4 * d, false, ring[d]);
            for (; g + 2 * PD <= n; g += PD) {
#pragma unroll
                for (int d = 0; d < PD; ++d) {
                    mma(ring[d]);
                    fetch(4 + 4 * (g + PD + d), false, ring[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < PD; ++d) mma(ring[d]);
            g += PD;
        }
        for (; g < n; ++g) {
            fetch(4 + 4 * g, false, ev);
            mma(ev);
        }
        if (a.Wout > 4) {
            fetch(a.Wout - 4, true, ev);
            mma(ev, true);
        }
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
    static DynLdsOnce attr_set;
    // experiment knob (only with -DMC_DEBUG_HOOKS): MONOCON_HIP_WGRAD_LDS_KB pads the dynamic LDS request, i.e. caps the
    // workgroups per CU
    static const size_t lds_req = [] {
#ifdef MC_DEBUG_HOOKS
        const char *e = std::getenv("MONOCON_HIP_WGRAD_LDS_KB");
        const size_t pad = e ? (size_t)std::atoi(e) * 1024 : 0;
        return pad > Cfg::LDS_BYTES ? pad : (size_t)Cfg::LDS_BYTES;
#else
        return (size_t)Cfg::LDS_BYTES;
#endif
    }();
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(kern), (int)(lds_req));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.ksplit * a.n_tiles * a.c_tiles), dim3(Cfg::NT), lds_req, st, a);
    return hipGetLastError();
}

// shape choice: 64n x 64c (2x2 waves) wherever both dimensions allow, 128 x 32 for thin inputs, single-wave tiles for <= 32
static void wgrad_shape(const WgradArgs &a, int *WN, int *WC) {
    bool src64 = true;   // a c-tile must lie inside one source of the virtual concat
    for (int i = 0; i < a.nsrc; ++i) src64 = src64 && (a.nsrc == 1 || a.src[i].C % 64 == 0);
    // measured on MI355X (scratch/wgexp): the 64n x 64c workgroup beats 128n x 32c by 3-8 %
    if (a.Cout > 32 && a.Cin > 32 && src64) { *WN = 2; *WC = 2; }
    else if (a.Cout > 64) { *WN = 4; *WC = 1; }
    else if (a.Cout > 32) { *WN = 2; *WC = 1; }
    else { *WN = 1; *WC = 1; }
}
static bool wgrad_sources_ok(const WgradArgs &a) {
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].C % 4 || (a.nsrc > 1 && a.src[i].C % 32)) return false;
    return a.dy_ld % 4 == 0;
}

void wgrad_plan(WgradArgs &a, int ks, int stride) {
    a.small = 0;
    a.pipe = 0;
    if (wgrad_is_small(a, ks, stride)) {
        a.small = 1;
        const int rows = a.B * a.Hout;
        a.ksplit = std::min(rows / 4 > 0 ? rows / 4 : 1, 1024);   // >= 4 rows (one per wave) per workgroup
        a.n_tiles = a.c_tiles = 1;
        a.ppr = a.ppi = a.groups_per_img = 0;
        return;
    }
    if (const int tile = wgrad_pipe_tile(a, ks, stride)) { wgrad_pipe_plan(a, tile); return; }
    int WN, WC;
    wgrad_shape(a, &WN, &WC);
    a.n_tiles = (a.Cout + 32 * WN - 1) / (32 * WN);
    a.c_tiles = (a.Cin + 32 * WC - 1) / (32 * WC);
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.pb = (a.prec >= 1 && wgrad_bf16_ok(a, ks, stride)) ? wgrad_bf16_patches(a.prec) : 2;
    a.groups_per_img = (a.ppi + a.pb - 1) / a.pb;
    const long long G = (long long)a.B * a.groups_per_img;
    // two resident 4-wave workgroups per CU (MONOCON_HIP_WGRAD_BLOCKS: experiment knob, e.g. 256 = one per CU, which
    // leaves half of every SIMD's registers to whatever the main stream runs beside it)
    const char *eb = std::getenv("MONOCON_HIP_WGRAD_BLOCKS");
    const int target = eb ? std::atoi(eb) : 512;
    int ks_ = target * 4 / (WN * WC) / (a.n_tiles * a.c_tiles);
    if (ks_ < 1) ks_ = 1;
    if (ks_ > G) ks_ = (int)G;
    a.ksplit = ks_;
}
size_t wgrad_partial_floats(const WgradArgs &a, int ks) { return (size_t)a.ksplit * ks * ks * a.Cout * a.Cin; }

bool wgrad_lazy_capable(const WgradArgs &a, int ks, int stride) {
    if (a.prec != 3) return false;
    if (a.pipe) return a.pipe == 4;
    if (a.small) return wgrad_thin_ok(a, ks, stride) || (ks == 3 && stride == 2 && a.Cout == 32 && a.nsrc == 1);      // (wgrad_small_kernel<2, 2, LZ>)
    return wgrad_sources_ok(a) && wgrad_bf16_ok(a, ks, stride);
}

hipError_t launch_wgrad(const WgradArgs &a, int ks, int stride, float *dw_oihw, hipStream_t st) {
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].la && !wgrad_lazy_capable(a, ks, stride)) return hipErrorInvalidValue;
    prof_last = {2, 2.0 * a.B * a.Hout * a.Wout * (double)a.Cout * a.Cin * ks * ks,
                 4.0 * ((double)a.B * a.Hin * a.Win * a.Cin + (double)a.B * a.Hout * a.Wout * a.Cout + (double)ks * ks * a.Cin * a.Cout)};
    hipError_t e = hipErrorInvalidValue;
    if (a.pipe) {
        e = launch_wgrad_pipe(a, ks, st);
    } else if (a.small && wgrad_thin_ok(a, ks, stride)) {
        e = launch_wgrad_thin(a, st);
    } else if (a.small && a.src[0].la) {      // lazy X: the stride-2 16 -> 32 layer (DLA level1) is the one that needs it
        if (stride == 2 && a.Cout == 32 && a.src[0].lb) hipLaunchKernelGGL((wgrad_small_kernel<2, 2, true>), dim3(a.ksplit), dim3(256), 0, st, a);
        else return hipErrorInvalidValue;
        e = hipGetLastError();
    } else if (a.small) {
        if (stride == 1 && a.Cout == 16) hipLaunchKernelGGL((wgrad_small_kernel<1, 1>), dim3(a.ksplit), dim3(256), 0, st, a);
        else if (stride == 1) hipLaunchKernelGGL((wgrad_small_kernel<1, 2>), dim3(a.ksplit), dim3(256), 0, st, a);
        else if (a.Cout == 16) hipLaunchKernelGGL((wgrad_small_kernel<2, 1>), dim3(a.ksplit), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wgrad_small_kernel<2, 2>), dim3(a.ksplit), dim3(256), 0, st, a);
        e = hipGetLastError();
    } else {
        int WN, WC;
        wgrad_shape(a, &WN, &WC);
        if (!wgrad_sources_ok(a)) return hipErrorInvalidValue;
        if (a.prec >= 1 && wgrad_bf16_ok(a, ks, stride)) {
            e = launch_wgrad_bf16(a, ks, stride, WN, WC, st);
        } else {
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
    }
    if (e != hipSuccess) return e;
    return launch_splitk_reduce(a.partial, a.ksplit, ks * ks, a.Cout, a.Cin, dw_oihw, st);
}

// dw (Cout, Cin, T) = sum over `ksplit` slices of partial[slice][T][Cout][Cin]   (Cin % 4 == 0)
hipError_t launch_splitk_reduce(const float *partial, int ksplit, int T, int Cout, int Cin, float *dw, hipStream_t st) {
    if (Cin % 4) return hipErrorInvalidValue;
    const size_t total = (size_t)T * Cout * Cin;
#define WR_LAUNCH(KL_)                                                                                              \
    hipLaunchKernelGGL((wgrad_reduce_kernel<KL_>), dim3((unsigned)((total / 4 + 256 / KL_ - 1) / (256 / KL_))), \
                       dim3(256), 0, st, partial, ksplit, T, Cout, Cin, dw)
    if (ksplit >= 2048 && total <= 65536) WR_LAUNCH(64);
    else if (ksplit >= 32) WR_LAUNCH(16);
    else if (ksplit >= 4) WR_LAUNCH(4);
    else WR_LAUNCH(1);
#undef WR_LAUNCH
    return hipGetLastError();
}

}  // namespace mc
