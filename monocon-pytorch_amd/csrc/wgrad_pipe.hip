// Round 4: the stride-1 weight gradient of precision mode 3 (two fp16 pieces per operand, see wgrad_bf16.hip) as a software
// pipeline: ONE workgroup of 8 waves per CU, two LDS tile buffers, one barrier per pixel group.
//
// Why (scratch/wg16, B = 32, 3x3 layers, one launch): wgrad_bf16_kernel<3, 1, 2, 2, 2> takes 230-250 us; with the global
// loads and the staging compiled out it takes 166-172 us (the MFMAs with their fragment reads and barriers), with only the
// loads + barriers left 84-131 us: the three phases ADD, although two workgroups share every CU -- when a workgroup waits
// for its loads or converts them, the other one is usually doing the same.  Here the phases of ONE workgroup overlap
// instead: while the waves run the MFMAs of group g out of tile buffer g & 1 they convert and write group g + 1 into the
// other buffer (from registers loaded an iteration earlier) and have the loads of group g + 2 in flight.
//   iteration g:  issue loads(g + 2) -> register set g & 1
//                 MFMAs(g), fragment reads from buffer g & 1; convert + ds_write of register set (g + 1) & 1 -> buffer (g + 1) & 1
//                 s_waitcnt lgkmcnt(0); s_barrier          (raw: hipcc's __syncthreads would also drain the loads in flight)
// Same operand preparation, LDS image (channel-major, 8 pixels contiguous, skewed rows: conflict-free writes and reads),
// tap construction (one aligned read per halo row, the three column taps built in registers) and accumulation order per
// split-K slice as wgrad_bf16_kernel -- the same partial sums; what differs is the split-K slicing (WgradArgs::ksplit) and,
// for the 64 x 64 tile, that the two 16-pixel halves of a group go to different waves and slices (WK = 2).
// Tiles (cout x cin): 128 x 64 (WN 4, WC 2), 64 x 128 (WN 2, WC 4), 64 x 64 (WN 2, WC 2, WK 2); 8 waves each.
// Reference: what autograd computes for nn.Conv2d.weight.grad (model/backbone/dla.py:21-29, dla_neck.py:56-57).
#include <algorithm>
#include <cstdlib>
#include "conv_mfma.h"
#include "train.h"

namespace mc {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w32x4 __attribute__((ext_vector_type(4)));
template <int V> struct WpInt { static constexpr int value = V; };

template <int KS, int WN, int WC, int WK>
struct WgPipeCfg {
    static constexpr int NB = 32 * WN, CB = 32 * WC, NW = WN * WC * WK, NT = 64 * NW;
    static constexpr int PAD = KS / 2, IH = 3 + KS, IW = 7 + KS, T = KS * KS;
    static constexpr int XROW = KS == 3 ? 32 : 16;               // bytes per staged halo row (10 / 8 pixels)
    static constexpr int XCH = IH * XROW + 48, DCH = 4 * 16 + 48;   // bytes per channel row (see WgB16Cfg: the skew needs 48)
    static constexpr int X_PLANE = CB * XCH, D_PLANE = NB * DCH;  // one fp16 piece of each tile
    static constexpr int BUF = 2 * (X_PLANE + D_PLANE);
    static constexpr size_t LDS_BYTES = 2 * BUF;
    static_assert(LDS_BYTES <= 160 * 1024, "two tile buffers per CU");
};

__device__ __forceinline__ int wp_skew(int channel) { return ((channel >> 4) & 3) * 16; }
// lane <-> channel permutation inside a wave's 32 channels (an involution; see lane_chan() of wgrad_bf16.hip)
__device__ __forceinline__ int wp_lane_chan(int li) { return ((li & 15) >= 4 && (li & 15) < 12) ? (li ^ 16) : li; }

// (x0, x1) * s -> packed fp16 pair of the hi pieces and of the lo pieces, four instructions: v_fma_mix{lo,hi}_f16 computes
// fma(a, b, c) in fp32 (sources fp32 or one half of a register) and rounds once to fp16 -- hi = f16(x * s), lo = f16(x * s - hi)
__device__ __forceinline__ void split_pair(float x0, float x1, float s, unsigned &hi, unsigned &lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    hi = h;
    lo = l;
}

__device__ __forceinline__ void wp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LZ: some source of X is a lazy tensor (ConvSrc::la in conv_mfma.h: raw conv output + BatchNorm coefficients, its value
// max(fma(y, la, lb), 0) formed while X is staged); its own instantiation, so that every other launch keeps its registers
template <int KS, int WN, int WC, int WK, bool LZ = false>
// (the 4-wave build is meant to SHARE a CU with whatever the main stream runs: at most half of a SIMD's registers)
__global__ __launch_bounds__(64 * WN * WC * WK, WN * WC * WK == 4 ? 2 : 1) void wgrad_pipe_kernel(const WgradArgs a) {
    using Cfg = WgPipeCfg<KS, WN, WC, WK>;
    constexpr int NB = Cfg::NB, CB = Cfg::CB, NT = Cfg::NT, PAD = Cfg::PAD, IH = Cfg::IH, IW = Cfg::IW, T = Cfg::T;
    constexpr int XROW = Cfg::XROW, XCH = Cfg::XCH, DCH = Cfg::DCH, XPL = Cfg::X_PLANE, DPL = Cfg::D_PLANE, BUF = Cfg::BUF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WN * WC), wn = (wave % (WN * WC)) / WC, wc = wave % WC;
    const int g = lane >> 5, li = lane & 31;
    const int bid = xcd_order(blockIdx.x, gridDim.x);
    const int ct = __builtin_amdgcn_readfirstlane(bid % a.c_tiles);
    const int nt = __builtin_amdgcn_readfirstlane((bid / a.c_tiles) % a.n_tiles);
    const int ksb = __builtin_amdgcn_readfirstlane(bid / (a.c_tiles * a.n_tiles));
    const int kblocks = a.ksplit / WK;
    const int n0 = nt * NB, c0 = ct * CB;
    const long long G = (long long)a.B * a.groups_per_img;
    const int g_begin = __builtin_amdgcn_readfirstlane((int)(G * ksb / kblocks));
    const int g_end = __builtin_amdgcn_readfirstlane((int)(G * (ksb + 1) / kblocks));

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
    const int ex = f16_scale_exp(amax_read(a.amax_x[si])), ed = f16_scale_exp(amax_read(a.amax_dy));
    const float x_scale = exp2i(ex), d_scale = exp2i(ed), omul = exp2i(-ex) * exp2i(-ed);

    // ---- staging plan (as wgrad_bf16_kernel): X item = (halo row, pixel pair, channel quad), dY item = (patch row, pixel
    //      pair, channel quad), channel quad fastest
    constexpr int XC4 = CB / 4, XPAIRS = IW / 2, XP = IH * XPAIRS * XC4, NIX = (XP + NT - 1) / NT;
    constexpr int NC4 = NB / 4, DP = 4 * 4 * NC4, NID = (DP + NT - 1) / NT;
    static_assert(NT % XC4 == 0 && NT % NC4 == 0 && IW % 2 == 0, "static channel quad per thread");
    constexpr int DEAD = -(1 << 24);
    const int xc4 = tid % XC4, dn4 = tid % NC4;
    const bool xc_ok = cs0 + xc4 * 4 < Cs && c0 + xc4 * 4 < a.Cin;
    const bool dn_ok = n0 + dn4 * 4 + 3 < a.dy_ld;
    // Threads beyond the last item DUPLICATE an earlier item of their own channel quad (same loads, same values to the same
    // LDS addresses) instead of sitting the staging out: a per-thread `if (e < XP)` around the conversion is a branch, the
    // wait for the loads then sits on one arm only, and hipcc's wait-count pass -- which must assume the other arm --
    // drains all loads (vmcnt(0)) at the next re-use of those registers: the top of every other iteration.
    constexpr int KX = (NT * NIX - XP + XC4 - 1) / XC4 * XC4, KD = (NT * NID - DP + NC4 - 1) / NC4 * NC4;
    static_assert(KX <= XP && KD <= DP, "duplicate items exist");
    int x_stat[NIX], x_ix[NIX], x_dst[NIX];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int e0 = tid + NT * i, e = e0 < XP ? e0 : e0 - KX, item = (e / XC4) % (IH * XPAIRS);
        const int iy = item / XPAIRS, pr = item % XPAIRS;
        const int ix = pr * 2 - PAD;
        x_ix[i] = xc_ok ? ix : DEAD;
        x_stat[i] = (((iy - PAD) * a.Win + ix) * Cs + cs0 + xc4 * 4) * 4;
        x_dst[i] = (xc4 * 4) * XCH + wp_skew(xc4 * 4) + iy * XROW + pr * 4;
    }
    int d_stat[NID], d_mx[NID], d_dst[NID];
#pragma unroll
    for (int i = 0; i < NID; ++i) {
        const int e0 = tid + NT * i, e = e0 < DP ? e0 : e0 - KD, item = (e / NC4) % 16;
        const int my = item / 4, mx = (item % 4) * 2;
        d_mx[i] = dn_ok ? mx : DEAD;
        d_stat[i] = ((my * a.Wout + mx) * a.dy_ld + n0 + dn4 * 4) * 4;
        d_dst[i] = 2 * XPL + (dn4 * 4) * DCH + wp_skew(dn4 * 4) + my * 16 + mx * 2;
    }

    f32x4 xv[2][NIX][2], dv[2][NID][2];       // two register sets of raw fp32 data: groups g + 1 and g + 2 while g runs
    // lazy X: the workgroup's c-tile lies in ONE source.  The coefficients of the tile's CB channels (operand scale folded in)
    // wait in LDS behind the two tile buffers and are read back per staged item: kept in registers (8 of them) the kernel
    // spilled, and a scratch reload is a vector-memory load -- it retires on the same in-order counter as the prefetched
    // groups, i.e. it waits for them (measured, round 6: 0.24 -> 0.37 ms per launch).  okm[slot] bit 2i + k = pixel k of item
    // i lies inside the image (padding stays 0, not relu(lb)).
    const bool lz = LZ && a.src[si].la != nullptr;
    [[maybe_unused]] f32x4 *lz_tab = reinterpret_cast<f32x4 *>(lds_raw + 2 * BUF);      // [2][XC4]: A, B per channel quad
    if constexpr (LZ) {
        if (lz && tid < XC4) {
            f32x4 cA = {0.f, 0.f, 0.f, 0.f}, cB = cA;
            if (cs0 + tid * 4 < Cs && c0 + tid * 4 < a.Cin) {
                cA = *reinterpret_cast<const f32x4 *>(a.src[si].la + cs0 + tid * 4);
                cB = *reinterpret_cast<const f32x4 *>(a.src[si].lb + cs0 + tid * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { cA[j] *= x_scale; cB[j] *= x_scale; }
            lz_tab[tid] = cA;
            lz_tab[XC4 + tid] = cB;
        }
        __syncthreads();
    }
    [[maybe_unused]] const unsigned x_bytes = (unsigned)(a.Hin * a.Win * Cs) * 4u;
    [[maybe_unused]] unsigned okm[2] = {0u, 0u};
    auto fetch = [&](int gi, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int img = __builtin_amdgcn_readfirstlane(gi / a.groups_per_img);
        const int pp = gi - img * a.groups_per_img;
        const __amdgpu_buffer_rsrc_t r_x =
            make_rsrc(xsrc + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
        const __amdgpu_buffer_rsrc_t r_d =
            make_rsrc(a.dy + (size_t)img * a.Hout * a.Wout * a.dy_ld, (unsigned)(a.Hout * a.Wout * a.dy_ld) * 4u);
        const int prow = __builtin_amdgcn_readfirstlane(pp / a.ppr);
        const int oy = prow * 4, ox = (pp - prow * a.ppr) * 8;
        const int xb = (oy * a.Win + ox) * Cs * 4;
        const int db = (oy * a.Wout + ox) * a.dy_ld * 4;
        // (one UNSIGNED compare per pixel: with `xx >= 0 && xx < W` hipcc branches on the shared half of the two conditions
        //  and, since both arms load into the same registers, puts s_waitcnt vmcnt(0) between them -- a full memory latency
        //  in the middle of the MFMA stream, every group)
        if constexpr (LZ) okm[slot] = 0u;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int xx = ox + x_ix[i];
            const bool in0 = (unsigned)xx < (unsigned)a.Win, in1 = (unsigned)(xx + 1) < (unsigned)a.Win;
            const int vo0 = in0 ? xb + x_stat[i] : BUF_OOB, vo1 = in1 ? xb + x_stat[i] + Cs * 4 : BUF_OOB;
            xv[slot][i][0] = buf_load4(r_x, vo0, 0);
            xv[slot][i][1] = buf_load4(r_x, vo1, 0);
            // (the column is inside: the offset is inside the image's bytes <=> the row is)
            if constexpr (LZ) okm[slot] |= (((unsigned)vo0 < x_bytes ? 1u : 0u) | ((unsigned)vo1 < x_bytes ? 2u : 0u)) << (2 * i);
        }
#pragma unroll
        for (int i = 0; i < NID; ++i) {
            const int xx = ox + d_mx[i];
            const bool in0 = (unsigned)xx < (unsigned)a.Wout, in1 = (unsigned)(xx + 1) < (unsigned)a.Wout;
            dv[slot][i][0] = buf_load4(r_d, in0 ? db + d_stat[i] : BUF_OOB, 0);
            dv[slot][i][1] = buf_load4(r_d, in1 ? db + d_stat[i] + a.dy_ld * 4 : BUF_OOB, 0);
        }
    };
    // two pixels x four channels -> per channel one packed pixel pair per piece plane
    auto put = [&](unsigned char *dst, int plane, const f32x4 v0, const f32x4 v1, int chan_stride, float scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned hi, lo;
            split_pair(v0[j], v1[j], scale, hi, lo);
            *reinterpret_cast<unsigned *>(dst + j * chan_stride) = hi;
            *reinterpret_cast<unsigned *>(dst + plane + j * chan_stride) = lo;
        }
    };
    // part `part` of `parts` of the conversion of register set `slot` into tile buffer `buf`
    auto store_x = [&](unsigned char *buf, auto slot_c, int i) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (LZ) {
            if (lz) {
                const float cap0 = ((okm[slot] >> (2 * i)) & 1u) ? __builtin_inff() : 0.f;
                const float cap1 = ((okm[slot] >> (2 * i + 1)) & 1u) ? __builtin_inff() : 0.f;
                const f32x4 lzA = lz_tab[xc4], lzB = lz_tab[XC4 + xc4];
                f32x4 t0, t1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t0[j] = lazy_act(xv[slot][i][0][j], lzA[j], lzB[j], cap0);
                    t1[j] = lazy_act(xv[slot][i][1][j], lzA[j], lzB[j], cap1);
                }
                put(buf + x_dst[i], XPL, t0, t1, XCH, 1.f);
                return;
            }
        }
        put(buf + x_dst[i], XPL, xv[slot][i][0], xv[slot][i][1], XCH, x_scale);
    };
    auto store_d = [&](unsigned char *buf, auto slot_c, int i) {
        constexpr int slot = decltype(slot_c)::value;
        put(buf + d_dst[i], DPL, dv[slot][i][0], dv[slot][i][1], DCH, d_scale);
    };

    const int lc = wp_lane_chan(li);
    const int a_off = 2 * XPL + (wn * 32 + lc) * DCH + wp_skew(wn * 32 + lc) + g * 16;
    const int b_off = (wc * 32 + lc) * XCH + wp_skew(wc * 32 + lc) + g * XROW;

    // the MFMAs of one 16-pixel half (patch rows 2q, 2q + 1) of the group in buffer `buf`; `mid` runs after the first tap row
    // has been issued (the staging work of the next group is slotted in there, behind MFMAs that are already queued)
    auto half = [&](const unsigned char *buf, int q, auto &&mid) {
        h16x8 av[2];
#pragma unroll
        for (int z = 0; z < 2; ++z) av[z] = *reinterpret_cast<const h16x8 *>(buf + a_off + z * DPL + (2 * q) * 16);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const unsigned char *row = buf + b_off + (2 * q + r) * XROW;
            w32x4 b0[2], b1[2], b2[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const w32x4 lo = *reinterpret_cast<const w32x4 *>(row + z * XPL);
                b0[z] = lo;
                if (KS == 3) {
                    w32x4 hi4 = *reinterpret_cast<const w32x4 *>(row + z * XPL + 16);
                    asm volatile("" : "+v"(hi4));
                    const unsigned hi = hi4[0];
                    b1[z][0] = __builtin_amdgcn_alignbit(lo[1], lo[0], 16);
                    b1[z][1] = __builtin_amdgcn_alignbit(lo[2], lo[1], 16);
                    b1[z][2] = __builtin_amdgcn_alignbit(lo[3], lo[2], 16);
                    b1[z][3] = __builtin_amdgcn_alignbit(hi, lo[3], 16);
                    b2[z][0] = lo[1]; b2[z][1] = lo[2]; b2[z][2] = lo[3]; b2[z][3] = hi;
                }
            }
            // partial products (piece of dY, piece of X), smallest first: (lo, hi), (hi, lo), (hi, hi)
            constexpr int PA[3] = {1, 0, 0}, PX[3] = {0, 1, 0};
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                const int za = PA[pp], zx = PX[pp];
                if (KS == 1) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[za], __builtin_bit_cast(h16x8, b0[zx]), acc[0], 0, 0, 0);
                } else {
                    acc[r * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[za], __builtin_bit_cast(h16x8, b0[zx]), acc[r * 3 + 0], 0, 0, 0);
                    acc[r * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[za], __builtin_bit_cast(h16x8, b1[zx]), acc[r * 3 + 1], 0, 0, 0);
                    acc[r * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[za], __builtin_bit_cast(h16x8, b2[zx]), acc[r * 3 + 2], 0, 0, 0);
                }
            }
            if (r == 0) mid();
        }
    };

    // one iteration: group gi out of buffer `cur`; group gi + 1 (register set SN) into the other buffer; loads of gi + 2
    // into register set SC (the set group gi came from)
    auto iteration = [&](int gi, auto sc, auto sn) {
        unsigned char *cur = lds_raw + decltype(sc)::value * BUF, *nxt = lds_raw + decltype(sn)::value * BUF;
        // (both pieces of staging work are slotted in behind MFMAs that are already queued: the address arithmetic of the
        //  loads and the conversion would otherwise run while the matrix pipe of BOTH waves of a SIMD idles -- the barrier
        //  keeps them in step.  Both are UNCONDITIONAL -- past the end the last group is fetched / staged again and never
        //  used: with `if (more) fetch` the compiler's wait-count pass has to assume the path without the new loads, on
        //  which the register set staged below holds the YOUNGEST loads, and puts vmcnt(0) in front of the conversion.)
        auto fetch_next = [&] { fetch(min(gi + 2, g_end - 1), sc); };
        auto stage_next = [&] {
#pragma unroll
            for (int i = 0; i < NIX; ++i) store_x(nxt, sn, i);
#pragma unroll
            for (int i = 0; i < NID; ++i) store_d(nxt, sn, i);
        };
        if constexpr (WK == 2) {
            half(cur, wk, [&] { fetch_next(); stage_next(); });
        } else {
            half(cur, 0, fetch_next);
            half(cur, 1, stage_next);
        }
        wp_barrier();
    };

    // ---- prologue: groups g_begin (-> buffer 0) and g_begin + 1 (-> register set 1)
    if (g_begin < g_end) {
        fetch(g_begin, WpInt<0>{});
        fetch(min(g_begin + 1, g_end - 1), WpInt<1>{});      // (the compiler's own vmcnt before the conversion leaves these in flight)
#pragma unroll
        for (int i = 0; i < NIX; ++i) store_x(lds_raw, WpInt<0>{}, i);
#pragma unroll
        for (int i = 0; i < NID; ++i) store_d(lds_raw, WpInt<0>{}, i);
        wp_barrier();
    }
    for (int gi = g_begin; gi < g_end; gi += 2) {
        iteration(gi, WpInt<0>{}, WpInt<1>{});
        if (gi + 1 < g_end) iteration(gi + 1, WpInt<1>{}, WpInt<0>{});
    }

    // ---- epilogue: partial[slice][tap][n][c];  D row = n, D col (lane) = c
    const int slice = ksb * WK + wk;
    const int c = c0 + wc * 32 + lc;
    if (c < a.Cin && c - cbase < Cs) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 32 + wp_lane_chan((r & 3) + 8 * (r >> 2) + 4 * g);
                if (n < a.Cout) a.partial[(((size_t)slice * T + t) * a.Cout + n) * a.Cin + c] = acc[t][r] * omul;
            }
    }
}

template <int KS, int WN, int WC, int WK, bool LZ = false>
static hipError_t launch_wp(const WgradArgs &a, hipStream_t st) {
    using Cfg = WgPipeCfg<KS, WN, WC, WK>;
    if constexpr (!LZ) {
        for (int i = 0; i < a.nsrc; ++i)
            if (a.src[i].la) {
                if constexpr (WN == 2 && WC == 2 && WK == 1) return launch_wp<KS, WN, WC, WK, true>(a, st);      // (the default tile)
                else return hipErrorInvalidValue;
            }
    }
    auto kern = wgrad_pipe_kernel<KS, WN, WC, WK, LZ>;
    constexpr size_t lds_bytes = Cfg::LDS_BYTES + (LZ ? 2 * (Cfg::CB / 4) * 16 : 0);      // (+ the lazy coefficient table)
    static DynLdsOnce attr_set;
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(kern), (int)(lds_bytes));
        if (e != hipSuccess) return e;
    }
    if (a.ksplit % WK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((a.ksplit / WK) * a.n_tiles * a.c_tiles), dim3(Cfg::NT), lds_bytes, st, a);
    return hipGetLastError();
}

// tile of the pipelined kernel for this layer (0: not eligible): 1 = 128n x 64c, 2 = 64n x 128c, 3 = 64n x 64c (two K halves)
// -- the 8-wave builds, which own a CU while they run -- and 4 = 64n x 64c on FOUR waves, which leaves half of the registers
// and 72 KB of LDS of every CU to whatever the main stream runs beside it.
// MONOCON_HIP_WGRAD_PIPE: 0 = off (the two-barrier kernel everywhere), 1 (default) = tile 4 for every eligible layer,
// 2 = tiles 1 / 2, 3 = tiles 1 / 2 / 3.  MEASURED (B = 32 train step, one session): per launch the 8-wave tiles are the
// fastest (weight-gradient bucket alone 12.45 ms, tile 4: 13.7), but the weight gradients run on their own stream BESIDE
// the data-gradient chain, and an 8-wave workgroup (8 x 244 registers, 116-148 KB LDS) shares its CU with nothing: the two
// streams then merely take turns (sum of the three buckets 55.8 ms, step 55.3).  With tile 4 the element-wise passes of the
// main stream (HBM-bound, few registers) and one conv workgroup per CU run under the weight gradients: step 54.8 -> 53.9 ms.
int wgrad_pipe_tile(const WgradArgs &a, int ks, int stride) {
    const char *en = std::getenv("MONOCON_HIP_WGRAD_PIPE");
    const int enabled = en ? std::atoi(en) : 1;
    if (!enabled || a.prec != 3 || stride != 1 || ks != 3 || a.small) return 0;
    if (a.Wout != a.Win || a.Hout != a.Hin || a.dy_ld % 4 || !a.amax_dy) return 0;
    for (int i = 0; i < a.nsrc; ++i)
        if (!a.amax_x[i] || a.src[i].C % 4) return 0;
    auto src_mult = [&](int m) {       // a c-tile must lie inside one source of the virtual concat
        for (int i = 0; i < a.nsrc; ++i)
            if (a.nsrc > 1 && a.src[i].C % m) return false;
        return true;
    };
    if (enabled == 1) return (a.Cout % 64 == 0 && a.Cin % 64 == 0 && src_mult(64)) ? 4 : 0;
    // (a last, half-filled 128-row tile -- the fused 64 -> 576 head conv -- still beats nine 64-row tiles)
    if (a.Cout >= 128 && a.Cout % 64 == 0 && a.Cin % 64 == 0 && src_mult(64)) return 1;
    if (a.Cout == 64 && a.Cin % 128 == 0 && src_mult(128)) return 2;
    // the 64 x 64 tile (two K halves) only ties with the two-barrier kernel (222-244 vs 235-246 us)
    if (enabled >= 3 && a.Cout == 64 && a.Cin % 64 == 0 && src_mult(64)) return 3;
    return 0;
}
void wgrad_pipe_plan(WgradArgs &a, int tile) {
    const int NB = tile == 1 ? 128 : 64, CB = tile == 2 ? 128 : 64, WK = tile == 3 ? 2 : 1;       // (tile 4: 64 x 64, 4 waves)
    a.pipe = tile;
    a.small = 0;
    a.pb = 1;
    a.n_tiles = (a.Cout + NB - 1) / NB;
    a.c_tiles = (a.Cin + CB - 1) / CB;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.groups_per_img = a.ppi;
    const long long G = (long long)a.B * a.groups_per_img;
    // (read per plan, not cached: the tests shrink it to run long pixel loops on small inputs)
    const char *eb = std::getenv("MONOCON_HIP_WGRAD_PIPE_BLOCKS");
    const int blocks = eb ? std::atoi(eb) : 256;
    int kb = blocks / (a.n_tiles * a.c_tiles);        // one resident 8-wave workgroup per CU
    if (kb < 1) kb = 1;
    if (kb > G) kb = (int)G;
    a.ksplit = kb * WK;
}
hipError_t launch_wgrad_pipe(const WgradArgs &a, int ks, hipStream_t st) {
    if (ks != 3) return hipErrorInvalidValue;
    switch (a.pipe) {
    case 1: return launch_wp<3, 4, 2, 1>(a, st);
    case 2: return launch_wp<3, 2, 4, 1>(a, st);
    case 3: return launch_wp<3, 2, 2, 2>(a, st);
    case 4: return launch_wp<3, 2, 2, 1>(a, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mc
