// Mixed-precision weight gradient (BASELINE config 3; see conv_bf16.hip for the mode):
//
//   dW[n, c, r, s] = sum_{b, y, x} dY[b, y, x, n] * X[b, y - 1 + r, x - 1 + s, c]        (3x3, stride 1;  1x1: r = s = 0)
//
// on v_mfma_f32_32x32x16_bf16.  The reduction dimension K of the MFMA is the PIXEL index, and a lane
// supplies 8 consecutive k from one 16-byte register group -- so both operands must sit in LDS
// channel-major, 8 pixels of one row contiguous: the staging transposes.
//   * A[n][k]: dYt[n][patch row][8 px] bf16.  One ds_read_b128 = row 2q+g of channel n = 8 of the 16 k.
//   * B[k][c]: Xt[c][halo row][16 px] bf16 (10 used).  For tap column s the 8 pixels x+s..x+s+7 start at a
//     2-byte offset, which LDS cannot serve in one aligned read; instead a lane reads pixels 0..7 (b128) and
//     8..9 (b32) once per halo row and builds the three operands in registers: s = 0 as read, s = 2 is the
//     same registers shifted by one dword (free), s = 1 is four v_alignbit.
//   * staging: a thread loads the float4 (4 channels) of TWO horizontally adjacent pixels, rounds to bf16
//     (RNE) and writes one packed dword per channel -- the transpose costs ds_write_b32, not ds_write_b16.
//     Offsets, zero fill and the register prefetch across the MFMA phase are as in wgrad_mfma_kernel.
//   * SPL = 3 (mode 2, fp32 emulation, see conv_bf16.hip): dY and X are each staged as three bf16 pieces and every tap
//     accumulates the six partial products of weight >= 2^-16; one patch per staged group so two workgroups fit a CU.
//   * SPL = 2 (mode 3, fp32 emulation by a 2-way fp16 split, see conv_bf16.hip): dY and X are scaled by the power of two
//     their tensor's max |x| dictates (WgradArgs::amax_dy / amax_x), staged as two fp16 pieces each, three partial
//     products per tap; the partial sums leave the kernel multiplied by the exact inverse of both scales.
// Accumulation, split-K partials and the deterministic reduce stay fp32 (wgrad_reduce_kernel).
// Stride-2 3x3 layers (round 3): template parameter S = 2, see WgB16Cfg.  The 16-channel layers keep their fp32 kernel.
#include <algorithm>
#include <cstdlib>
#include "conv_mfma.h"
#include "train.h"

namespace mc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SPL> struct WPiece { typedef __bf16 T; typedef bf16x8 V8; typedef bf16x2 V2; };
template <> struct WPiece<2> { typedef _Float16 T; typedef f16x8 V8; typedef f16x2 V2; };
__device__ __forceinline__ f32x16 wmfma_k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 wmfma_k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int KS, int S, int WN, int WC, int SPL>
struct WgB16Cfg {
    static_assert(S == 1 || (S == 2 && KS == 3), "stride 2: 3x3 only");
    static constexpr int PB = SPL == 1 ? 2 : 1;                 // patches per staged group (WgradArgs::pb)
    static constexpr int NB = 32 * WN, CB = 32 * WC, NT = 64 * WN * WC;
    static constexpr int PAD = KS / 2;
    // stride 1: halo tile 6 x 10 (3x3) or 4 x 8 (1x1).  stride 2: the 4 x 8 output patch reads input rows 2y-1 .. 2y+7 and
    // columns 2x-1 .. 2x+15; a staged row holds the ODD columns (2x-1+2j, j = 0..8, 16 slots) followed by the EVEN ones
    // (2x+2j, j = 0..7), so that the 8 pixels a tap column needs are again 8 consecutive slots: tap 0 = odd slots 0..7,
    // tap 1 = even slots 0..7, tap 2 = odd slots 1..8
    static constexpr int IH = S == 2 ? 9 : 3 + KS, IW = 7 + KS;
    static constexpr int XROW = S == 2 ? 48 : (KS == 3 ? 32 : 16);   // bytes per staged halo row
    // Bytes per channel row.  The staging writes of a wave go to 16 channel groups x 4 pixel pairs: with 16-byte aligned
    // rows the 4-channel stride is a multiple of 16 banks whatever the padding, i.e. 4 distinct banks for 16 lanes (a
    // 4-way conflict on every ds_write_b32; PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.75).  Every group of 16
    // channels is therefore skewed by another 16 bytes (lds_skew): the 16 channel groups land on 16 distinct multiples
    // of 4 banks and the pixel pairs fill the gaps -- conflict-free writes; the fragment reads (8 lanes = 8 consecutive
    // channels, same skew) keep their disjoint banks.  Rows grow by the largest skew (48 bytes).
    // (stride 2: 9 * 48 + 64 = 496 = 31 * 16 -- an odd number of 16-byte windows per channel keeps the fragment reads of 16
    //  consecutive channels on 16 distinct windows, as 15 does for stride 1)
    static constexpr int XCH = IH * XROW + (S == 2 ? 64 : 48);   // 240 (3x3) / 112 (1x1) / 496 (3x3 stride 2)
    static constexpr int DCH = 4 * 16 + 48;                     // bytes per dY channel: 112
    static constexpr int X_PLANE = PB * CB * XCH, D_PLANE = PB * NB * DCH;   // one bf16 piece of each tile
    static constexpr int X_BYTES = SPL * X_PLANE, D_BYTES = SPL * D_PLANE;
    static constexpr size_t LDS_BYTES = X_BYTES + D_BYTES;
};

__device__ __forceinline__ int lds_skew(int channel) { return ((channel >> 4) & 3) * 16; }
// Which channel of its wave's 32 a lane reads (MFMA row / column li <-> channel lane_chan(li), an involution).  The LDS
// serves a ds_read_b128 in the lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+32): with the identity mapping
// a group straddles two 16-channel blocks, whose rows lds_skew() shifts against each other by 16 bytes -- half the 16-byte
// windows of a group then collide (PMC, rounds 2 / 3: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.67).  Swapping lanes
// 4-11 with 20-27 gives every group the 16 channels of ONE block: window = (7 or 15) * channel + const mod 16, a bijection.
__device__ __forceinline__ int lane_chan(int li) { return ((li & 15) >= 4 && (li & 15) < 12) ? (li ^ 16) : li; }

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

#ifdef MC_EXP_PHASE_DELAY
__device__ int g_phase_delay = 0, g_phase_mode = 0;
__device__ unsigned *g_lds_dbg = nullptr;
#endif
// LZ (SPL == 2): some source of X is a lazy tensor (ConvSrc::la in conv_mfma.h) -- see wgrad_pipe_kernel
template <int KS, int S, int WN, int WC, int SPL, bool LZ = false>
// stride 2 stages 2.7x the halo of stride 1 (6 instead of 2 float4 pairs per thread in flight across the MFMA phase): at
// two workgroups per CU (256 registers) the kernel spills 17-23 registers; one workgroup per CU without spills is faster
// (one-session A/B, weight-gradient bucket per step: fp32 kernel 14.6 ms, two per CU 14.0, one per CU 13.6)
#ifndef MC_WG16_S2_OCC
#define MC_WG16_S2_OCC 1
#endif
__global__ __launch_bounds__(64 * WN * WC, S == 2 ? MC_WG16_S2_OCC : 2) void wgrad_bf16_kernel(const WgradArgs a) {
    using Cfg = WgB16Cfg<KS, S, WN, WC, SPL>;
    typedef typename WPiece<SPL>::T pc_t;
    typedef typename WPiece<SPL>::V8 pc8;
    typedef typename WPiece<SPL>::V2 pc2;
    constexpr int XPL = Cfg::X_PLANE, DPL = Cfg::D_PLANE;
    constexpr int PB = Cfg::PB, NB = Cfg::NB, CB = Cfg::CB, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IH = Cfg::IH, IW = Cfg::IW, XROW = Cfg::XROW, XCH = Cfg::XCH, DCH = Cfg::DCH;
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *xt = lds_raw;                       // [PB][CB] x XCH
    unsigned char *dyt = lds_raw + Cfg::X_BYTES;       // [PB][NB] x DCH

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WC, wc = wave % WC;
    const int g = lane >> 5, li = lane & 31;
    const int bid = xcd_order(blockIdx.x, gridDim.x);      // the tiles of one split-K slice read the same pixels
    // (integer division by a run-time value is a vector-ALU sequence: its results are wave-uniform by construction, but
    //  the compiler only knows that once told -- else the tile origin, the source pointer and with them every buffer
    //  descriptor count as divergent and each load is wrapped in a readfirstlane "waterfall" loop)
    const int ct = __builtin_amdgcn_readfirstlane(bid % a.c_tiles);
    const int nt = __builtin_amdgcn_readfirstlane((bid / a.c_tiles) % a.n_tiles);
    const int ks = __builtin_amdgcn_readfirstlane(bid / (a.c_tiles * a.n_tiles));
    const int n0 = nt * NB, c0 = ct * CB;
    const long long G = (long long)a.B * a.groups_per_img;
    const int g_begin = __builtin_amdgcn_readfirstlane((int)(G * ks / a.ksplit));
    const int g_end = __builtin_amdgcn_readfirstlane((int)(G * (ks + 1) / a.ksplit));

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int si = 0, cbase = 0;
    while (si + 1 < a.nsrc && c0 >= cbase + a.src[si].C) { cbase += a.src[si].C; ++si; }
    const int Cs = a.src[si].C;
    const float *xsrc = a.src[si].p;
    const int cs0 = c0 - cbase;
    float x_scale = 1.f, d_scale = 1.f, omul = 1.f;      // SPL == 2: operand scales and the exact inverse of their product
    if constexpr (SPL == 2) {
        const int ex = f16_scale_exp(amax_read(a.amax_x[si]));
        const int ed = f16_scale_exp(amax_read(a.amax_dy));
        x_scale = exp2i(ex);
        d_scale = exp2i(ed);
        omul = exp2i(-ex) * exp2i(-ed);
    }

    // ---- staging plan.  X item = (halo row, pixel pair, channel group) of a patch, channel group fastest;
    //      dY item = (patch row, pixel pair, channel group).
    constexpr int XC4 = CB / 4, XPAIRS = S == 2 ? 9 : IW / 2, XP = IH * XPAIRS * XC4, NIX = (XP + NT - 1) / NT;
    constexpr int NC4 = NB / 4, DP = 4 * 4 * NC4, NID = (DP + NT - 1) / NT;
    static_assert(NT % XC4 == 0 && NT % NC4 == 0 && IW % 2 == 0, "static channel group per thread");
    constexpr int DEAD = -(1 << 24);
    constexpr int XSTEP = S;                 // the two pixels of a staged pair are S columns apart in the image
    const int xc4 = tid % XC4, dn4 = tid % NC4;
    const bool xc_ok = cs0 + xc4 * 4 < Cs && c0 + xc4 * 4 < a.Cin;
    const bool dn_ok = n0 + dn4 * 4 + 3 < a.dy_ld;
    int x_stat[NIX], x_ix[NIX], x_dst[NIX];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int e = tid + NT * i, item = (e / XC4) % (IH * XPAIRS);
        const int iy = item / XPAIRS, pr = item % XPAIRS;
        // column of the pair's first pixel relative to the tile origin (S * ox), and its byte offset inside the staged row
        int ix, roff;
        if (S == 2) {
            if (pr < 5) { ix = 4 * pr - 1; roff = 4 * pr; }                 // odd plane, slots 2pr, 2pr + 1
            else { ix = 4 * (pr - 5); roff = 32 + 4 * (pr - 5); }           // even plane
        } else {
            ix = pr * 2 - PAD; roff = pr * 4;
        }
        x_ix[i] = (xc_ok && e < XP) ? ix : DEAD;
        x_stat[i] = (((iy - PAD) * a.Win + ix) * Cs + cs0 + xc4 * 4) * 4;
        x_dst[i] = (xc4 * 4) * XCH + lds_skew(xc4 * 4) + iy * XROW + roff;
    }
    int d_stat[NID], d_mx[NID], d_dst[NID];
#pragma unroll
    for (int i = 0; i < NID; ++i) {
        const int e = tid + NT * i, item = (e / NC4) % 16;
        const int my = item / 4, mx = (item % 4) * 2;
        d_mx[i] = (dn_ok && e < DP) ? mx : DEAD;
        d_stat[i] = ((my * a.Wout + mx) * a.dy_ld + n0 + dn4 * 4) * 4;
        d_dst[i] = (dn4 * 4) * DCH + lds_skew(dn4 * 4) + my * 16 + mx * 2;
    }

    // raw fp32 data of the pixel groups in flight; PD = how many groups ahead the loads run.  Two ahead (24 more registers)
    // was tried in round 3 on a Little's-law argument (2 workgroups per CU x 23 KB in flight): A/B in one session 59.5 ms
    // (PD 1) vs 60.1 ms (PD 2) per step -- slower; it stays at one.
#ifndef MC_WG16_PD
#define MC_WG16_PD 1
#endif
    constexpr int PD = MC_WG16_PD;
    f32x4 xv[PD][PB][NIX][2], dv[PD][PB][NID][2];
    // lazy X: the workgroup's c-tile lies in ONE source; the thread's channel quad keeps its coefficients (operand scale
    // folded in); okm bit 2i + k = pixel k of item i lies inside the image (padding stays 0, not relu(lb))
    static_assert(!LZ || SPL == 2, "lazy sources: the fp16-split mode");
    const bool lz = LZ && a.src[si].la != nullptr;
    f32x4 lzA = {0.f, 0.f, 0.f, 0.f}, lzB = lzA;
    if (LZ && lz && xc_ok) {
        lzA = *reinterpret_cast<const f32x4 *>(a.src[si].la + cs0 + xc4 * 4);
        lzB = *reinterpret_cast<const f32x4 *>(a.src[si].lb + cs0 + xc4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { lzA[j] *= x_scale; lzB[j] *= x_scale; }
    }
    [[maybe_unused]] const unsigned x_bytes = (unsigned)(a.Hin * a.Win * Cs) * 4u;
    [[maybe_unused]] unsigned okm[PD][PB];
    auto fetch = [&](int gi, int p, int slot) {
        // (wave-uniform by construction; the integer division runs on the vector ALU, so say so -- otherwise every
        //  buffer load below is wrapped in a readfirstlane "waterfall" loop over its descriptor)
        const int img = __builtin_amdgcn_readfirstlane(gi / a.groups_per_img);
        const int pp = (gi - img * a.groups_per_img) * PB + p;
        const __amdgpu_buffer_rsrc_t r_x =
            make_rsrc(xsrc + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
        const __amdgpu_buffer_rsrc_t r_d =
            make_rsrc(a.dy + (size_t)img * a.Hout * a.Wout * a.dy_ld, (unsigned)(a.Hout * a.Wout * a.dy_ld) * 4u);
        const bool valid = pp < a.ppi;
        const int prow = __builtin_amdgcn_readfirstlane(pp / a.ppr);
        const int oy = prow * 4, ox = valid ? (pp - prow * a.ppr) * 8 : DEAD;
        const int xb = (oy * S * a.Win + ox * S) * Cs * 4;
        const int db = (oy * a.Wout + ox) * a.dy_ld * 4;
        // (one UNSIGNED compare per pixel: with `xx >= 0 && xx < W` hipcc branches on the shared half of the two conditions
        //  and, since both arms load into the same registers, puts s_waitcnt vmcnt(0) between them -- a full memory latency
        //  in front of the MFMAs of every group.  Dead items carry a column far outside either way.)
        if constexpr (LZ) okm[slot][p] = 0u;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int xx = ox * S + x_ix[i];
            const bool in0 = (unsigned)xx < (unsigned)a.Win, in1 = (unsigned)(xx + XSTEP) < (unsigned)a.Win;
            const int vo0 = in0 ? xb + x_stat[i] : BUF_OOB, vo1 = in1 ? xb + x_stat[i] + XSTEP * Cs * 4 : BUF_OOB;
            xv[slot][p][i][0] = buf_load4(r_x, vo0, 0);
            xv[slot][p][i][1] = buf_load4(r_x, vo1, 0);
            // (the column is inside: the offset is inside the image's bytes <=> the row is)
            if constexpr (LZ) okm[slot][p] |= (((unsigned)vo0 < x_bytes ? 1u : 0u) | ((unsigned)vo1 < x_bytes ? 2u : 0u)) << (2 * i);
        }
#pragma unroll
        for (int i = 0; i < NID; ++i) {
            const int xx = ox + d_mx[i];
            const bool in0 = (unsigned)xx < (unsigned)a.Wout, in1 = (unsigned)(xx + 1) < (unsigned)a.Wout;
            dv[slot][p][i][0] = buf_load4(r_d, in0 ? db + d_stat[i] : BUF_OOB, 0);
            dv[slot][p][i][1] = buf_load4(r_d, in1 ? db + d_stat[i] + a.dy_ld * 4 : BUF_OOB, 0);
        }
    };
    // piece q of (a, b): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (mode 2), packed pixel pair per channel
    auto put = [&](unsigned char *dst, int plane, f32x4 v0, f32x4 v1, int chan_stride, float scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r0 = SPL == 2 ? v0[j] * scale : v0[j], r1 = SPL == 2 ? v1[j] * scale : v1[j];
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                pc2 pr;
                pr[0] = (pc_t)r0;
                pr[1] = (pc_t)r1;
                *reinterpret_cast<unsigned *>(dst + q * plane + j * chan_stride) = __builtin_bit_cast(unsigned, pr);
                r0 -= (float)pr[0];
                r1 -= (float)pr[1];
            }
        }
    };
    auto store = [&](int p, int slot) {
#pragma unroll
        for (int i = 0; i < NIX; ++i)
            if (NT * (i + 1) <= XP || tid + NT * i < XP) {
                if constexpr (LZ) {
                    if (lz) {
                        const float cap0 = ((okm[slot][p] >> (2 * i)) & 1u) ? __builtin_inff() : 0.f;
                        const float cap1 = ((okm[slot][p] >> (2 * i + 1)) & 1u) ? __builtin_inff() : 0.f;
                        f32x4 t0, t1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            t0[j] = lazy_act(xv[slot][p][i][0][j], lzA[j], lzB[j], cap0);
                            t1[j] = lazy_act(xv[slot][p][i][1][j], lzA[j], lzB[j], cap1);
                        }
                        put(xt + p * CB * XCH + x_dst[i], XPL, t0, t1, XCH, 1.f);
                        continue;
                    }
                }
                put(xt + p * CB * XCH + x_dst[i], XPL, xv[slot][p][i][0], xv[slot][p][i][1], XCH, x_scale);
            }
#pragma unroll
        for (int i = 0; i < NID; ++i)
            if (NT * (i + 1) <= DP || tid + NT * i < DP) {
                put(dyt + p * NB * DCH + d_dst[i], DPL, dv[slot][p][i][0], dv[slot][p][i][1], DCH, d_scale);
            }
    };

    const int lc = lane_chan(li);
    const unsigned char *a_base = dyt + (wn * 32 + lc) * DCH + lds_skew(wn * 32 + lc) + g * 16;
    const unsigned char *b_base = xt + (wc * 32 + lc) * XCH + lds_skew(wc * 32 + lc) + g * S * XROW;

#ifdef MC_EXP_PHASE_DELAY
    // experiment: the second workgroup of a CU (LDS base != 0) starts late, so that its staging phases fall into the other
    // workgroup's MFMA phases instead of coinciding with its staging phases
    {
        const unsigned la = __builtin_amdgcn_s_getreg((31 << 11) | 6);   // HW_REG_LDS_ALLOC, all 32 bits
        if (g_lds_dbg && threadIdx.x == 0) g_lds_dbg[blockIdx.x] = la;
        if (g_phase_mode == 0 ? (la & 0xfff) != 0 : blockIdx.x >= gridDim.x / 2)
            for (int i = 0; i < g_phase_delay; ++i) __builtin_amdgcn_s_sleep(8);      // 512 cycles each
    }
#endif
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (g_begin + d < g_end) {
#pragma unroll
            for (int p = 0; p < PB; ++p) fetch(g_begin + d, p, d);
        }
    for (int g0 = g_begin; g0 < g_end; g0 += PD)
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        const int gi = g0 + d;
        if (gi >= g_end) break;       // (uniform over the workgroup)
        __syncthreads();   // fragment reads of the previous group are done
#ifndef MC_EXP_NO_STORE
#pragma unroll
        for (int p = 0; p < PB; ++p) store(p, d);
#else
#pragma unroll
        for (int p = 0; p < PB; ++p) {      // the loads stay (and are waited for), the split + LDS writes go
#pragma unroll
            for (int i = 0; i < NIX; ++i) asm volatile("" ::"v"(xv[d][p][i][0]), "v"(xv[d][p][i][1]));
#pragma unroll
            for (int i = 0; i < NID; ++i) asm volatile("" ::"v"(dv[d][p][i][0]), "v"(dv[d][p][i][1]));
        }
#endif
        __syncthreads();
#ifndef MC_EXP_NO_FETCH
        if (gi + PD < g_end) {
#pragma unroll
            for (int p = 0; p < PB; ++p) fetch(gi + PD, p, d);
        }
#endif
        // partial products (piece of dY, piece of X), smallest first; mode 1: the single (0, 0)
        constexpr int NP = SPL == 1 ? 1 : (SPL == 2 ? 3 : 6);
        constexpr int PA[6] = {SPL == 2 ? 1 : 0, SPL == 2 ? 0 : 2, SPL == 2 ? 0 : 1, 0, 1, 0};
        constexpr int PX[6] = {SPL == 2 ? 0 : 2, SPL == 2 ? 1 : 0, SPL == 2 ? 0 : 1, 1, 0, 0};
#pragma unroll
        for (int p = 0; p < PB; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // 16 pixels per MFMA: patch rows 2q (g = 0) and 2q + 1 (g = 1)
                pc8 av[SPL];
#pragma unroll
                for (int z = 0; z < SPL; ++z)
                    av[z] = *reinterpret_cast<const pc8 *>(a_base + z * DPL + p * NB * DCH + (2 * q) * 16);
#pragma unroll
                for (int r = 0; r < KS; ++r) {
                    const unsigned char *row = b_base + p * CB * XCH + (2 * q * S + r) * XROW;
                    u32x4 b0[SPL], b1[SPL], b2[SPL];     // the three tap columns of every piece of X
#pragma unroll
                    for (int z = 0; z < SPL; ++z) {
                        const u32x4 lo = *reinterpret_cast<const u32x4 *>(row + z * XPL);
                        b0[z] = lo;
                        if (KS == 3) {
                            // pixels 8, 9 of the row: a second 16-byte read (conflict-free, 4 LDS cycles) rather than a
                            // ds_read_b32, whose 32-lane groups meet 4-way on channel rows that are multiples of 16 bytes
                            // (the empty asm keeps the compiler from narrowing it back to the one dword that is used)
                            u32x4 hi4 = *reinterpret_cast<const u32x4 *>(row + z * XPL + 16);
                            asm volatile("" : "+v"(hi4));
                            const unsigned hi = hi4[0];
                            u32x4 sh;                        // slots 1..8 of the plane `lo` starts
                            sh[0] = __builtin_amdgcn_alignbit(lo[1], lo[0], 16);
                            sh[1] = __builtin_amdgcn_alignbit(lo[2], lo[1], 16);
                            sh[2] = __builtin_amdgcn_alignbit(lo[3], lo[2], 16);
                            sh[3] = __builtin_amdgcn_alignbit(hi, lo[3], 16);
                            if (S == 2) {                    // odd columns: taps 0 and 2; even columns: tap 1
                                b1[z] = *reinterpret_cast<const u32x4 *>(row + z * XPL + 32);
                                b2[z] = sh;
                            } else {
                                b1[z] = sh;
                                b2[z][0] = lo[1]; b2[z][1] = lo[2]; b2[z][2] = lo[3]; b2[z][3] = hi;
                            }
                        }
                    }
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        const int za = SPL == 1 ? 0 : PA[pp], zx = SPL == 1 ? 0 : PX[pp];
#ifdef MC_EXP_NO_MFMA
                        asm volatile("" ::"v"(av[za]), "v"(b0[zx]), "v"(b1[zx]), "v"(b2[zx]));
                        if (true) {
                        } else
#endif
                        if (KS == 1) {
                            acc[0] = wmfma_k16(av[za], __builtin_bit_cast(pc8, b0[zx]), acc[0]);
                        } else {
                            acc[r * 3 + 0] = wmfma_k16(av[za], __builtin_bit_cast(pc8, b0[zx]), acc[r * 3 + 0]);
                            acc[r * 3 + 1] = wmfma_k16(av[za], __builtin_bit_cast(pc8, b1[zx]), acc[r * 3 + 1]);
                            acc[r * 3 + 2] = wmfma_k16(av[za], __builtin_bit_cast(pc8, b2[zx]), acc[r * 3 + 2]);
                        }
                    }
                }
            }
    }
    // ---- epilogue: partial[ks][tap][n][c];  D row = n, D col (lane) = c
    const int c = c0 + wc * 32 + lc;
    if (c < a.Cin && c - cbase < Cs) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 32 + lane_chan((r & 3) + 8 * (r >> 2) + 4 * g);
                if (n < a.Cout) a.partial[(((size_t)ks * T + t) * a.Cout + n) * a.Cin + c] = SPL == 2 ? acc[t][r] * omul : acc[t][r];
            }
    }
}

template <int KS, int S, int WN, int WC, int SPL, bool LZ = false>
static hipError_t launch_wg16(const WgradArgs &a, hipStream_t st) {
    using Cfg = WgB16Cfg<KS, S, WN, WC, SPL>;
    if (a.pb != Cfg::PB) return hipErrorInvalidValue;
    if constexpr (!LZ) {
        for (int i = 0; i < a.nsrc; ++i)
            if (a.src[i].la) {
                if constexpr (SPL == 2) return launch_wg16<KS, S, WN, WC, SPL, true>(a, st);
                else return hipErrorInvalidValue;
            }
    }
    auto kern = wgrad_bf16_kernel<KS, S, WN, WC, SPL, LZ>;
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

bool wgrad_bf16_ok(const WgradArgs &a, int ks, int stride) {
    if (a.small || (ks != 3 && ks != 1)) return false;
#ifdef MC_WG16_NO_S2
    if (stride == 2) return false;
#endif
    if (stride == 2) {        // the five stride-2 3x3 layers of DLA-34 (the 16-channel one keeps its own kernel: a.small)
        if (ks != 3 || a.Hin != 2 * a.Hout || a.Win != 2 * a.Wout) return false;
    } else if (stride != 1 || a.Wout != a.Win || a.Hout != a.Hin) {
        return false;
    }
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].C % 4) return false;
    if (a.prec == 3) {        // the fp16 split needs the maxima of both operand tensors
        if (!a.amax_dy) return false;
        for (int i = 0; i < a.nsrc; ++i) {
            if (!a.amax_x[i]) return false;
        }
    }
    return a.dy_ld % 4 == 0;
}

// patches per staged pixel group of the kernel launch_wgrad_bf16 will run (wgrad_plan sizes the groups with it)
int wgrad_bf16_patches(int prec) { return prec >= 2 ? 1 : 2; }

// the main kernel of launch_wgrad in the bf16-pipe modes (the split-K reduce is shared); WN / WC as planned
template <int SPL>
static hipError_t launch_wgrad_b16_spl(const WgradArgs &a, int ks, int stride, int WN, int WC, hipStream_t st) {
#define WG16(KS_, S_)                                                        \
    if (WN == 2 && WC == 2) return launch_wg16<KS_, S_, 2, 2, SPL>(a, st);   \
    if (WN == 4) return launch_wg16<KS_, S_, 4, 1, SPL>(a, st);              \
    if (WN == 2) return launch_wg16<KS_, S_, 2, 1, SPL>(a, st);              \
    return launch_wg16<KS_, S_, 1, 1, SPL>(a, st);
    if (ks == 3 && stride == 2) { WG16(3, 2) }
    if (ks == 3) { WG16(3, 1) }
    WG16(1, 1)
#undef WG16
}
hipError_t launch_wgrad_bf16(const WgradArgs &a, int ks, int stride, int WN, int WC, hipStream_t st) {
    if (a.prec == 3) return launch_wgrad_b16_spl<2>(a, ks, stride, WN, WC, st);
    return a.prec == 2 ? launch_wgrad_b16_spl<3>(a, ks, stride, WN, WC, st) : launch_wgrad_b16_spl<1>(a, ks, stride, WN, WC, st);
}

}  // namespace mc
