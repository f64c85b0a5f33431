// Round 4: the fused convolution of precision mode 4 ("f16x2 arithmetic on P16 activations") -- both MFMA operands reach
// LDS by DMA (buffer_load ... lds, 16 bytes per lane), in a software pipeline with ONE barrier per K-step.
//
// Same contract, tiling (4x8-pixel patches, PB patches x BNT output channels per workgroup), arithmetic (three fp16
// partial products l*h, h*l, h*h per multiply, the two minor ones in their own accumulator) and epilogue (conv_epilogue of
// conv_mfma.h: folded scale / bias, residual, ReLU, statistics partials, backward-statistics mode, max-|out|) as
// conv_bf16_kernel<..., SPL = 2>.  What changed is how the operands travel:
//   * activations are stored as P16 (p16.h): the fp16 pieces of x * 2^e, made once by the tensor's producer.  The halo
//     tile of a 16-channel chunk is copied global -> LDS by the DMA engine, one 1 KB piece per wave instruction, into the
//     conflict-free image of conv_bf16.hip ([octet][piece][patch pair][halo row][24 slots of 16 bytes]: a piece's flat slot
//     index IS its LDS position, the lane's SOURCE address carries the gather; out-of-image lanes fetch zeros through the
//     buffer descriptor's bounds check -- verified on gfx950, scratch/p16/t_glds.hip);
//   * the weight fragments (already packed as piece planes [piece][tap][Cin/8][CoutP][8]) take the same road, one
//     (tap, 16 channels) slice per K-step into a ring of four: a workgroup fetches every weight byte ONCE instead of once
//     per wave, and from LDS (256 B/clk/CU) instead of the vector-memory path (64 B/clk/CU);
//   * no register ever holds staging data: the registers go to a second fragment set (the fragments of step s + 1 are
//     read while the MFMAs of step s run) and to 2x2-tile waves with both accumulator sets.
// Pipeline, K-step s = one tap of one 16-channel chunk (3 * WTM * WTN MFMAs per wave):
//   top of s    : DMA of the weight slice of step s + 4; this step's share of the halo tile LAC chunks ahead
//                 ds_read of the fragments of step s + 1; the MFMAs of step s
//   bottom of s : s_waitcnt vmcnt(N) with N = the DMA instructions issued at the top of THIS step (for 3x3 windows also the
//                 halo pieces of the previous step, which come after the weights in issue order) -- i.e. everything step
//                 s + 2 will read has landed -- lgkmcnt(0), s_barrier.
// WHEN a staged buffer may be read (found the hard way -- rare wrong tiles under load, first with a ring of three): the
// wait that retires a DMA, one s_barrier, and a ds_read right behind it is NOT enough on gfx950 -- the guide's rule "read a
// staged buffer one phase AFTER the wait that retires it, never in the same phase" holds: here a weight slice is retired at
// the bottom of step s - 2 and first read at the top of step s (two barriers and one MFMA phase apart), halo tiles likewise.
// A DMA has three K-steps to land and is never waited for with vmcnt(0) inside the loop (hipcc's __syncthreads would:
// raw s_barrier + inline-asm waits, one __shared__ array; MI355X guide, "Pipelining across barriers").
// Measured on the prototype (scratch/p16, B = 32, random operands): SQ_VALU_MFMA_BUSY 0.60-0.70 (conv_bf16_kernel:
// 0.43-0.50) at 1.55-1.7 GHz (1.96): the matrix pipe is kept ~50 % busier and the chip answers with a lower clock -- a pure
// v_mfma_f32_32x32x16_f16 loop on random register operands sustains only ~1500 TFLOP/s on this part (power), against which
// these kernels reach 1050-1100.
#include "conv_mfma.h"
#include "p16.h"

#include <type_traits>

namespace mc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KS, int WM, int WN, int WTM, int WTN>
struct CfgP16 {
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int PB = WM * WTM, NPAIR = (PB + 1) / 2, BNT = WN * WTN * 32;
    static constexpr int KH = win_h(KS), KW = win_w(KS), PAD = win_pad(KS), NTAP = KH * KW;
    static constexpr int IH = 3 + KH, IW = 7 + KW, RS = 24;
    static_assert(2 * IW <= RS, "two patches per 24-slot row");
    static constexpr int PLANE_SLOTS = NPAIR * IH * RS;                 // one (octet, piece) plane of the halo tile
    static constexpr int A_SLOTS = 4 * PLANE_SLOTS;                      // 2 octets x 2 pieces
    static constexpr int NAI = (A_SLOTS + 63) / 64;                      // DMA instructions per tile ...
    static constexpr int NA_W = (NAI + NW - 1) / NW;                     // ... per wave (the tile buffer is padded to whole rounds)
    static constexpr int A_BYTES = NA_W * NW * 1024;
    static constexpr int B_SLOTS = 4 * BNT;
    static_assert(B_SLOTS % (64 * NW) == 0, "weight slice: whole DMA instructions per wave");
    static constexpr int NB_W = B_SLOTS / (64 * NW);
    static constexpr int B_BYTES = B_SLOTS * 16;
    // the tile of chunk c + LAC is issued during the first NA_STEPS steps of chunk c, AQ pieces per step and wave, and
    // A piece issued at step j is retired at the bottom of step j + 1 + A_SLACK (the waits leave the current step's -- and
    // with A_SLACK the previous step's -- pieces in flight) and the tile is first read at the top of the LAST step before
    // its chunk, two barriers after its retirement: j <= LAC * NTAP - 4 - A_SLACK.
    static constexpr int LAC = NTAP >= 4 ? 1 : (NTAP == 2 ? 2 : 4);
    static constexpr int A_SLACK = NTAP >= 7 ? 1 : 0;                   // pieces of step t - 1 may still be in flight at the bottom of t
    static constexpr int AVAIL = LAC * NTAP - 3 - A_SLACK;
    static_assert(AVAIL >= 1 && AVAIL <= NTAP, "tile look-ahead");
    static constexpr int AQ = (NA_W + AVAIL - 1) / AVAIL;
    static constexpr int NA_STEPS = (NA_W + AQ - 1) / AQ;
    static constexpr int NABUF = LAC + 1;
    static constexpr int NBRING = 4;                                     // weight-slice ring: slice s + 4 is issued at the top of step s
    static constexpr int LDS_BYTES = NABUF * A_BYTES + NBRING * B_BYTES + PB * 16;
    static constexpr int na_at(int t) { return (t >= 0 && t < NA_STEPS) ? (((t + 1) * AQ <= NA_W) ? AQ : (NA_W - t * AQ)) : 0; }
    static constexpr int wait_at(int t) { return (A_SLACK ? na_at(t - 1) : 0) + NB_W + na_at(t); }
};

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)lds_dst, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm_lgkm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
template <int T> struct IntC { static constexpr int value = T; };

template <int KS, int WM, int WN, int WTM, int WTN, bool BM = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_p16_kernel(const ConvArgs a) {
    using C = CfgP16<KS, WM, WN, WTM, WTN>;
    constexpr int NW = C::NW, PB = C::PB, BNT = C::BNT, NTAP = C::NTAP, IW = C::IW, IH = C::IH, RS = C::RS, PAD = C::PAD;
    constexpr int PLANE_B = C::PLANE_SLOTS * 16, A_BYTES = C::A_BYTES, B_BYTES = C::B_BYTES, NA_W = C::NA_W, NB_W = C::NB_W;
    constexpr int AQ = C::AQ, NA_STEPS = C::NA_STEPS, NABUF = C::NABUF, LAC = C::LAC;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *const abuf = lds_raw;
    unsigned char *const bbuf = lds_raw + NABUF * A_BYTES;
    int *pinfo = reinterpret_cast<int *>(lds_raw + NABUF * A_BYTES + C::NBRING * B_BYTES);   // [PB][4] = img, oy0, ox0, valid

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 5, li = lane & 31;

    const int ntiles = a.CoutP / BNT;
    const int bid = xcd_order(blockIdx.x, gridDim.x);
    const int nt = bid % ntiles;
    const int mchunk = bid / ntiles;
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

    // ---- DMA plan.  Halo tile: piece j = wave + NW * i covers flat slots [64 j, 64 j + 64) of
    //      [octet][piece][patch pair][halo row][24 slots]; lane l of it fetches pixel (iy, ix) of patch 2 * pair + (slot >= IW).
    //      Kept per piece: the pixel's index inside the image (-1: zero fill); the byte offset follows from the source's
    //      channel count at every source switch (branch-free: this runs between chunks with ~200 registers live).
    int a_pix[NA_W], a_voff[NA_W];
#pragma unroll
    for (int i = 0; i < NA_W; ++i) {
        const int f = 64 * (wave + NW * i) + lane;
        const int plane = f / C::PLANE_SLOTS, r = f % C::PLANE_SLOTS;
        const int pair = r / (IH * RS), r2 = r % (IH * RS);
        const int iy = r2 / RS, sl = r2 % RS;
        const int j = sl >= IW ? 1 : 0, ix = sl - IW * j;
        const int p = 2 * pair + j;
        bool ok = plane < 4 && sl < 2 * IW && p < PB;
        int pix = -1;
        if (ok) {
            const int y = pinfo[p * 4 + 1] - PAD + iy, x = pinfo[p * 4 + 2] - PAD + ix;
            ok = pinfo[p * 4 + 3] && y >= 0 && y < a.Hin && x >= 0 && x < a.Win;
            pix = y * a.Win + x;
        }
        a_pix[i] = ok ? pix : -1;
    }
    auto lane_offsets = [&](int cs) {
#pragma unroll
        for (int i = 0; i < NA_W; ++i) {
            const int plane = (64 * (wave + NW * i) + lane) / C::PLANE_SLOTS;
            const int off = a_pix[i] * cs * 4 + (plane >> 1) * 32 + (plane & 1) * 16;
            a_voff[i] = a_pix[i] >= 0 ? off : BUF_OOB;
        }
    };
    // weight slice of one (tap, chunk): flat slot f -> [octet][piece][BNT columns] x 16 bytes
    const int w_plane = NTAP * a.Cin * a.CoutP * 2;            // bytes of one piece plane of the panel
    const int Cin8 = a.Cin >> 3;
    int b_voff[NB_W];
#pragma unroll
    for (int i = 0; i < NB_W; ++i) {
        const int f = 64 * (wave + NW * i) + lane;
        const int o = f / (2 * BNT), pc = (f / BNT) & 1, n = f % BNT;
        b_voff[i] = pc * w_plane + (o * a.CoutP + n0 + n) * 16;
    }
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpk16, (unsigned)(2 * w_plane));

    // ---- fragment addresses (bytes, tile buffer 0 / ring slot 0, tap (0, 0))
    int a_off[WTM], b_off[WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm) {
        const int p = wm * WTM + tm;
        a_off[tm] = (2 * g) * PLANE_B + ((p >> 1) * IH * RS + (li >> 3) * RS + (li & 7) + IW * (p & 1)) * 16;
    }
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) b_off[tn] = NABUF * A_BYTES + ((2 * g) * BNT + (wn * WTN + tn) * 32 + li) * 16;

    f32x16 acc[WTM][WTN], accm[WTM][WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accm[tm][tn][r] = 0.f; }

    // the epilogue's per-column coefficients: fetched before the K loop (their latency hides behind it) where the registers
    // allow it; the 2x2-tile waves -- 248 of 256 registers without them -- fetch them after the loop instead
    constexpr bool COEF_EARLY = WTM * WTN < 4;
    EpiCoef<WTN> coef;
    if (COEF_EARLY) coef = conv_epi_coef<WN, WTN, BM>(a, n0, wn, li);

    // ---- walk over the virtual concat, 16 channels per chunk.  Two cursors: (si_n, c0_n) runs LAC chunks ahead (the tile
    //      DMA), (si_c, c0_c) is the chunk whose MFMAs run.
    const int nch = a.Cin >> 4;
    int si_n = 0, c0_n = 0, Cs_n = a.src[0].C;
    __amdgpu_buffer_rsrc_t r_in = make_rsrc(a.src[0].p + (size_t)img * a.Hin * a.Win * Cs_n, (unsigned)(a.Hin * a.Win * Cs_n) * 4u);
    lane_offsets(Cs_n);
    auto advance_next = [&]() {        // stays on the last chunk at the end (a harmless re-load into a free buffer)
        int s2 = si_n, c2 = c0_n + 16;
        if (c2 >= Cs_n) { ++s2; c2 = 0; }
        if (s2 >= a.nsrc) return;
        if (s2 != si_n) {
            si_n = s2;
            Cs_n = a.src[s2].C;
            r_in = make_rsrc(a.src[s2].p + (size_t)img * a.Hin * a.Win * Cs_n, (unsigned)(a.Hin * a.Win * Cs_n) * 4u);
            lane_offsets(Cs_n);
        }
        c0_n = c2;
    };
    auto dma_a = [&](int i, int buf) {
        dma16(r_in, abuf + buf * A_BYTES + (wave + NW * i) * 1024, a_voff[i], c0_n * 4);
    };
    auto dma_b = [&](int slot, int tap, int ch) {
        const int soff = (tap * Cin8 + 2 * ch) * a.CoutP * 16;
#pragma unroll
        for (int i = 0; i < NB_W; ++i) dma16(r_w, bbuf + slot * B_BYTES + (wave + NW * i) * 1024, b_voff[i], soff);
    };
    auto load_frags = [&](f16x8(&fa)[2][WTM], f16x8(&fb)[2][WTN], int abuf_i, int tap, int slot) {
        const int toff = ((tap / C::KW) * RS + (tap % C::KW)) * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int tm = 0; tm < WTM; ++tm)
                fa[q][tm] = *reinterpret_cast<const f16x8 *>(lds_raw + abuf_i * A_BYTES + a_off[tm] + q * PLANE_B + toff);
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn)
                fb[q][tn] = *reinterpret_cast<const f16x8 *>(lds_raw + slot * B_BYTES + b_off[tn] + q * BNT * 16);
        }
    };
    // operand exponents: source i is stored as x * 2^e_i; the accumulators are kept in the unit of the CURRENT source
    // (rescaled by the exact power of two at a source switch), the epilogue undoes the last source's and the weights'
    int e_cur = *a.pexp[0];
    int si_c = 0, c0_c = 0;
    auto rescale_on_switch = [&]() {       // called at the top of every chunk but the first
        c0_c += 16;
        if (c0_c < a.src[si_c].C) return;
        c0_c = 0;
        ++si_c;
        const int e_new = *a.pexp[si_c];
        int d = e_new - e_cur;
        d = d > 100 ? 100 : (d < -100 ? -100 : d);
        e_cur = e_new;
        if (d == 0) return;
        const float m = exp2i(d);
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[tm][tn][r] *= m; accm[tm][tn][r] *= m; }
    };

    // ---- prologue: the tiles of chunks 0 .. LAC - 1, the weight slices of steps 0 .. 2
#pragma unroll
    for (int c = 0; c < LAC; ++c) {
#pragma unroll
        for (int i = 0; i < NA_W; ++i) dma_a(i, c);
        advance_next();
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { const int cs = s / NTAP; dma_b(s, s % NTAP, cs < nch ? cs : nch - 1); }
    wait_vm_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();             // (a staged buffer is read one barrier AFTER the one that follows its wait)
    f16x8 fa[2][2][WTM], fb[2][2][WTN];      // [register set][piece][tile]
    load_frags(fa[0], fb[0], 0, 0, 0);
    // hipcc books an LDS-DMA on its lgkmcnt model, the hardware counts it on vmcnt only: a counted lgkmcnt the compiler
    // places for these reads AFTER the DMAs of step 0 were issued would be too loose by the number of those DMAs (the MFMAs
    // of step 0 could read a fragment register before its ds_read has landed -- seen as rare wrong tiles under load).
    // Every later fragment read is retired by the explicit lgkmcnt(0) at the bottom of the step before its use.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    int ab = 0, bs = 0;                       // tile buffer of the current chunk, ring slot of its first step
    auto chunk_steps = [&](auto PARC, int ch) {
        constexpr int PAR = decltype(PARC)::value;          // parity of the chunk's first step: selects the register sets
        int ab_next = ab + 1; if (ab_next == NABUF) ab_next = 0;
        int ab_dma = ab + LAC; if (ab_dma >= NABUF) ab_dma -= NABUF;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int cur = (PAR + t) & 1;
            const int slot_t = (bs + t) & 3, slot_n = (slot_t + 1) & 3;
            {   // weights of step s + 4 into the slot step s has just left
                const int t4 = (t + 4) % NTAP, ch4 = ch + (t + 4) / NTAP;
                dma_b(slot_t, t4, ch4 < nch ? ch4 : nch - 1);
            }
            if (t < NA_STEPS) {
#pragma unroll
                for (int q = 0; q < AQ; ++q)
                    if (t * AQ + q < NA_W) dma_a(t * AQ + q, ab_dma);
            }
            if (t + 1 < NTAP) load_frags(fa[cur ^ 1], fb[cur ^ 1], ab, t + 1, slot_n);
            else load_frags(fa[cur ^ 1], fb[cur ^ 1], ab_next, 0, slot_n);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < WTN; ++tn) {
                        const int qa = pp == 0 ? 1 : 0, qb = pp == 1 ? 1 : 0;      // l*h, h*l, then h*h
                        if (pp < 2)
                            accm[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][qa][tm], fb[cur][qb][tn], accm[tm][tn], 0, 0, 0);
                        else
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][0][tm], fb[cur][0][tn], acc[tm][tn], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
            switch (t) {      // t is a constant after unrolling: the switch folds to one wait
#define MC_P16_CASE(T) case T: wait_vm_lgkm<C::wait_at(T)>(); break;
                MC_P16_CASE(0) MC_P16_CASE(1) MC_P16_CASE(2) MC_P16_CASE(3) MC_P16_CASE(4) MC_P16_CASE(5) MC_P16_CASE(6) MC_P16_CASE(7) MC_P16_CASE(8)
#undef MC_P16_CASE
                default: wait_vm_lgkm<0>(); break;
            }
            __builtin_amdgcn_s_barrier();
        }
        ab = ab_next;
        bs = (bs + NTAP) & 3;
        advance_next();     // (at the chunk boundary: control flow inside the unrolled steps costs the register allocator dearly)
    };
    constexpr int PAR1 = NTAP & 1;            // odd tap counts: the second chunk of a pair starts on the other register set
    for (int ch = 0; ch < nch; ch += 2) {
        if (ch) rescale_on_switch();
        chunk_steps(IntC<0>{}, ch);
        if (ch + 1 < nch) {
            rescale_on_switch();
            chunk_steps(IntC<PAR1>{}, ch + 1);
        } else if (PAR1) {
            break;                            // (odd chunk count, odd taps: nothing follows, the register sets need no re-alignment)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped look-ahead DMAs of the last steps

#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] += accm[tm][tn][r];
    if (!COEF_EARLY) coef = conv_epi_coef<WN, WTN, BM>(a, n0, wn, li);
    const float omul = exp2i(-e_cur) * exp2i(-f16_scale_exp(*a.amax_w));
    conv_epilogue<WM, WN, WTM, WTN, BNT, BM>(a, acc, pinfo, chunk * PB, img, n0, wm, wn, g, li, coef, omul);
}

// ---- dispatch
template <int KS, int WM, int WN, int WTM, int WTN, bool BM = false>
static hipError_t launch_p16_one(ConvArgs a, hipStream_t st, ConvArgs *resolved) {
    using C = CfgP16<KS, WM, WN, WTM, WTN>;
    if constexpr (!BM && (KS == 3 || KS == 1)) {
        if (a.bm_y) return launch_p16_one<KS, WM, WN, WTM, WTN, true>(a, st, resolved);
    } else if constexpr (!BM) {
        if (a.bm_y) return hipErrorInvalidValue;
    }
    if (C::LDS_BYTES > 160 * 1024) return hipErrorInvalidValue;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + C::PB - 1) / C::PB;
    if (a.CoutP % C::BNT) return hipErrorInvalidValue;
    if (resolved) *resolved = a;
    static bool attr_set = false;
    auto kern = conv_p16_kernel<KS, WM, WN, WTM, WTN, BM>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntiles = a.CoutP / C::BNT;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.chunks * ntiles)), dim3(C::NT), C::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <int KS>
static hipError_t launch_p16_shape(const ConvArgs &a, hipStream_t st, ConvArgs *resolved) {
    switch (a.cfg & 15) {
        case CFG_128x32: return launch_p16_one<KS, 2, 1, 2, 1>(a, st, resolved);      // two waves: 128 px x 32 ch (32-column layers)
        case CFG_128x64: return launch_p16_one<KS, 2, 2, 2, 1>(a, st, resolved);
        case CFG_128x64m: return launch_p16_one<KS, 4, 1, 1, 2>(a, st, resolved);
        case CFG_64x128: return launch_p16_one<KS, 1, 4, 2, 1>(a, st, resolved);
        case CFG_64x64: return launch_p16_one<KS, 2, 2, 1, 1>(a, st, resolved);
        default: return hipErrorInvalidValue;
    }
}

// eligible: the fp16-split arithmetic (prec 3), stride 1, every source stored as P16 (a multiple of 16 channels each), piece-plane weights
bool conv_p16_ok(const ConvArgs &a, int ks, int stride) {
    if (a.prec != 3 || !a.wpk16 || !a.amax_w || stride != 1) return false;
    if (!(ks == 3 || ks == 1 || ks == 12 || ks == 21 || ks == 22)) return false;
    if (a.Hin != a.Hout || a.Win != a.Wout) return false;
    for (int i = 0; i < a.nsrc; ++i)
        if (!a.pexp[i] || a.src[i].C % 16) return false;
    return true;
}
bool conv_p16_cfg_ok(int cfg, int CoutP, int ks) {
    (void)ks;      // (the 2x2-tile shape CFG_128x128 is not built: with the full epilogue it spills at 256 registers)
    switch (cfg & 15) {
        case CFG_128x32: case CFG_128x64: case CFG_128x64m: case CFG_64x128: case CFG_64x64: break;
        default: return false;
    }
    return !(cfg & ~15) && CoutP % conv_shape(cfg).BNT() == 0;
}

hipError_t launch_conv_p16(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved) {
    if (!conv_p16_ok(a, ks, stride)) return hipErrorInvalidValue;
    if (ks == 3) return launch_p16_shape<3>(a, st, resolved);
    if (ks == 1) return launch_p16_shape<1>(a, st, resolved);
    if (ks == 12) return launch_p16_shape<12>(a, st, resolved);
    if (ks == 21) return launch_p16_shape<21>(a, st, resolved);
    return launch_p16_shape<22>(a, st, resolved);
}

// ---- fp32 <-> P16 (op-level entry points, tests, debugging)
__global__ void p16_encode_kernel(const f32x4 *__restrict__ x, size_t nquads, int C4, const unsigned *__restrict__ amax,
                                  unsigned char *__restrict__ dst, int *__restrict__ e_out) {
    const int e = f16_scale_exp(amax_read(amax));
    const float s = exp2i(e);
    if (blockIdx.x == 0 && threadIdx.x == 0) *e_out = e;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nquads; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C4;
        const int q = (int)(i % C4);
        p16_store4(dst + pix * (size_t)C4 * 16, q, x[i], s);
    }
}
__global__ void p16_decode_kernel(const unsigned char *__restrict__ src, size_t nquads, int C4, const int *__restrict__ e,
                                  f32x4 *__restrict__ dst) {
    const float inv = exp2i(-*e);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nquads; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C4;
        const int q = (int)(i % C4);
        dst[i] = p16_load4(src + pix * (size_t)C4 * 16, q, inv);
    }
}
hipError_t launch_p16_encode(const float *x, size_t pixels, int C, const unsigned *amax_slot, void *dst, int *e_out, hipStream_t st) {
    if (C % 8 || !amax_slot || !e_out) return hipErrorInvalidValue;
    const size_t nq = pixels * (size_t)(C / 4);
    size_t gsz = (nq + 255) / 256;
    if (gsz > 8192) gsz = 8192;
    if (gsz < 1) gsz = 1;
    hipLaunchKernelGGL(p16_encode_kernel, dim3((unsigned)gsz), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(x), nq, C / 4,
                       amax_slot, static_cast<unsigned char *>(dst), e_out);
    return hipGetLastError();
}
hipError_t launch_p16_decode(const void *src, size_t pixels, int C, const int *e, float *dst, hipStream_t st) {
    if (C % 8 || !e) return hipErrorInvalidValue;
    const size_t nq = pixels * (size_t)(C / 4);
    size_t gsz = (nq + 255) / 256;
    if (gsz > 8192) gsz = 8192;
    if (gsz < 1) gsz = 1;
    hipLaunchKernelGGL(p16_decode_kernel, dim3((unsigned)gsz), dim3(256), 0, st, static_cast<const unsigned char *>(src), nq, C / 4,
                       e, reinterpret_cast<f32x4 *>(dst));
    return hipGetLastError();
}

}  // namespace mc
