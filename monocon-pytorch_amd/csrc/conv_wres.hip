// Round 5: the 64-channel 3x3 convolutions of precision mode 3 (f16x2) as a WEIGHT-RESIDENT, persistent kernel.
//
// Which launches (reference shapes): the 64 -> 64 BasicBlock convs of level2 at 1/4 resolution (model/backbone/dla.py:12-51),
// every 64-column slice of the fused 64 -> 576 head conv (model/dense_heads/monocon_heads.py:114-131) and the per-source
// 64 -> 64 data gradients of those layers and of the ida_2 nodes (model/backbone/dla_neck.py:94-106) -- K = 9 x 64 = 576,
// N = 64 per workgroup.  In conv_bf16_kernel these were the laggards of the conv bucket (profiles/r4e_pmc.txt: 0.22-0.26
// of the fp16 MFMA peak, `SQ_VALU_MFMA_BUSY` 0.43): a 128-pixel workgroup lives for ~3.5 us of matrix work and pays for
// it a prologue (patch table, max-|x| words, coefficients), 295 KB of weight fragments re-read from L2 (every wave its
// own 32-column half of the 147 KB panel, once per workgroup), two barriers per 32-channel chunk and an epilogue, none of
// which overlaps its own MFMAs.
//
// Here: 256 workgroups (one per CU, four waves with the whole register file: __launch_bounds__(256, 1) = 512 registers per
// lane) each walk a contiguous range of 4 x 16-pixel tiles.
//   * weights: a wave owns ONE 32-column tile for ALL of K: its 36 K-steps x 2 pieces = 72 B fragments (288 registers)
//     are loaded once per workgroup and stay in registers -- the hi pieces in vector registers, the lo pieces in
//     ACCUMULATION registers, from where the one MFMA per K-step that uses them reads them directly (inline asm: left to
//     itself hipcc copies every such fragment to vector registers first, 225 instructions per tile);
//   * waves = 2 column tiles x 2 patches: the two waves of a patch share its A fragments (LDS), the two patches of a tile
//     sit side by side in the conflict-free image of conv_bf16.hip ([piece][8-channel plane][halo row][24 slots of 16 B]);
//     fragments are requested two K-steps ahead;
//   * software pipeline over tiles, ONE raw barrier per tile.  A wave that is alone on its SIMD issues in order, so only
//     what stands BETWEEN two MFMAs in program order overlaps them (~5 instructions per 32-cycle MFMA); everything a tile
//     needs besides its 108 MFMAs is therefore dealt out into those gaps, K-step by K-step:
//       K-steps  0 .. 19   the EPILOGUE OF THE PREVIOUS TILE, one output row per step (its sums wait in 16 registers; the
//                          residual / BatchNorm-backward operands of a row are requested three steps before it is due)
//       K-steps 20 .. 27   conversion of the raw fp32 data of tile t + 1 (in registers since the end of tile t - 1) into
//                          its fp16 pieces (v_fma_mix{lo,hi}_f16: two instructions per element) and into image buffer
//                          (t + 1) & 1
//       K-steps 28 .. 35   addresses of tile t + 2; its loads go out after the loop (the prefetch has to be the YOUNGEST
//                          vector-memory work when it is consumed: gfx950 retires loads and stores on one in-order counter)
//     The first tile of a workgroup has no predecessor: the loop body exists twice (with / without the epilogue slots),
//     which also keeps every s_waitcnt the compiler derives exact (no merged paths with different queue contents).
// Measured on the way (64 -> 64 at 96 x 320, B = 32; conv_bf16_kernel 283-295 us): phase removal on the first version
// (epilogue as a block after the MFMAs): MFMAs alone 128-138 us, fragment waits +42, staging +37, epilogue +66-70 --
// serial, in-order; per-wave cycle counters: 4370 cycles per tile in the MFMA loop (3456 of them matrix work), 1800 in the
// epilogue, at 1.45 GHz (the chip is at its power limit).  An eight-wave variant (two K halves handing their accumulators
// through LDS, 256 registers per wave: scratch/exp/conv_wres_8wave_pipeline.hip.txt) was 30 % SLOWER: one barrier per tile
// couples all eight waves, and the wave that finishes a tile carries the epilogue and half of the MFMAs alone.
// Arithmetic, K order (32-channel chunk, tap, 16-channel step; l*h and h*l into the minor accumulator, h*h into the main
// one, folded once) and the epilogue's operations and their order are those of conv_bf16_kernel<3, 1, ..., SPL = 2> with
// conv_epilogue (conv_mfma.h): results are bit-identical to every other tiling (tests/test_hip_wres.py), so the autotuner
// may choose by time.
#include <cstdlib>
#include "conv_mfma.h"

namespace mc {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct WresCfg {
    static constexpr int NT = 256;
    static constexpr int IH = 6, IW = 10, NPIX = IH * IW, RS = 24;
    static constexpr int PPB = IH * RS * 16;            // one 8-channel plane of the patch pair: 6 rows x 24 slots x 16 B
    static constexpr int CPL = PPB + 32;                 // (+32: de-phases the planes' staging stores, see conv_bf16.hip)
    static constexpr int NPL = 8;                        // planes per piece: 64 channels
    static constexpr int PIECE = NPL * CPL;              // 18688
    static constexpr int TILE = 2 * PIECE;               // both fp16 pieces of one tile's halo image: 37376
    static constexpr int LDS_BYTES = 2 * TILE;
    static constexpr int C4 = 16;                        // channel quads per pixel
    static constexpr int TOTAL = 2 * NPIX * C4;          // staging items per tile (float4 each): 1920
    static constexpr int NIT = (TOTAL + NT - 1) / NT;    // 8 per thread, the last half-filled ...
    static constexpr int KDUP = NT * NIT - TOTAL;        // ... threads beyond duplicate an earlier item of their quad (128)
    static_assert(KDUP % C4 == 0 && KDUP <= TOTAL, "duplicate items keep the thread's channel quad");
};

// (x0, x1) * s -> packed fp16 pair of the hi pieces and of the lo pieces: hi = f16(x * s), lo = f16(x * s - hi), each ONE
// rounding of an exact fp32 value -- the same bits as the cvt / sub / cvt sequence of conv_bf16_kernel
__device__ __forceinline__ void wres_split_pair(float x0, float x1, float s, unsigned &hi, unsigned &lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
    hi = h;
    lo = l;
}

// Template flags select ONE straight-line epilogue: RES residual (or the gradient accumulated so far), STATS per-patch
// (sum, sum of squares) partials of a train-mode forward, BM backward-statistics mode (ConvArgs::bm_y; always with
// partials), ZMASK its ReLU mask from the stored activation (bm_relu == 1).
// EXP (measurement builds only, -DMC_WRES_EXP + MONOCON_WRES_EXP=bits; results WRONG by construction): 1 no MFMAs, 2 no
// conversion / LDS writes, 4 no global loads, 8 no epilogue, 16 no fragment reads
// LZ: the source is a lazy tensor (ConvSrc::la in conv_mfma.h: raw conv output + BatchNorm coefficients) -- forward launches only
template <bool BM, bool RES, bool STATS, bool ZMASK, int EXP = 0, bool LZ = false>
__global__ __launch_bounds__(256, 1) void conv_wres_kernel(const ConvArgs a, const int tiles_per_row, const int tiles_per_img,
                                                           const int total_tiles, const int tiles_per_wg, const int ngroups) {
#pragma clang fp contract(off)
    using C = WresCfg;
    constexpr int CPL = C::CPL, PIECE = C::PIECE, TILE = C::TILE, RS = C::RS, IW = C::IW, NIT = C::NIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wp = wave >> 1;            // column tile, patch of the pair
    const int g = lane >> 5, li = lane & 31;

    const int bid = xcd_order(blockIdx.x, gridDim.x);
    const int grp = __builtin_amdgcn_readfirstlane(bid % ngroups);
    const int rng = __builtin_amdgcn_readfirstlane(bid / ngroups);
    const int t_begin = rng * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, total_tiles);
    if (t_begin >= t_end) return;
    const int n0 = grp * 64;

    // the max-|x| words of the input: requested first, reduced behind the weight loads (conv_bf16_kernel)
    const unsigned am_raw = a.amax_in[0][lane < AMAX_SUB ? lane * AMAX_STRIDE : 0];

    // ---- the wave's weight fragments: [K-step][piece], K-step s = (chunk s / 18, tap (s % 18) / 2, half s % 2)
    const int Cin8 = a.Cin >> 3;
    const int w_plane = 9 * a.Cin * a.CoutP * 2;          // bytes of one piece plane of the panel
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpk16, (unsigned)(2 * w_plane));
    const int w_lane = (g * a.CoutP + n0 + wn * 32 + li) * 16;
    h16x8 breg[36][2];
#pragma unroll
    for (int s = 0; s < 36; ++s) {
        const int kc = (s / 18) * 32, tap = (s % 18) / 2, m = s % 2;
        const int soff = (tap * Cin8 + ((kc + m * 16) >> 3)) * a.CoutP * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            breg[s][q] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(r_w, w_lane, soff + q * w_plane, 0));
    }

    // ---- the epilogue's per-lane constants (conv_epilogue: column n = lane & 31 of the wave's tile, rows = pixels)
    const int ncol = n0 + wn * 32 + li;
    const bool nok = ncol < a.Cout;
    const EpiCoef<1> coef = conv_epi_coef<2, 1, BM>(a, n0, wn, li);
    const int v_out = nok ? (4 * g * a.o_px + a.out_coff + ncol) * 4 : BUF_OOB;
    const int v_res = nok ? (4 * g * a.r_px + ncol) * 4 : BUF_OOB;
    const int v_bm = nok ? (4 * g * a.o_px + ncol) * 4 : BUF_OOB;      // y / z share the dense layout of the output
    const float floor_v = a.relu ? 0.f : -__builtin_inff();
    const int bm_relu = a.bm_relu;

    // ---- staging plan: item e = tid + 256 i of [patch][halo pixel][channel quad]; its place in the tile never changes
    const int c4 = tid % C::C4;
    int rel[NIT], relb[NIT], sdst[NIT];   // rel: (halo row - 1) and (column - 1 relative to the tile origin), packed; relb: its byte offset
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int e0 = tid + C::NT * i, e = e0 < C::TOTAL ? e0 : e0 - C::KDUP;
        const int t = e / C::C4;
        const int pix = t % C::NPIX, p = t / C::NPIX;
        const int iy = pix / IW, ix = pix % IW;
        rel[i] = ((iy - 1) & 0xffff) | ((8 * p + ix - 1) << 16);
        relb[i] = ((iy - 1) * a.Win + 8 * p + ix - 1) * 256 + c4 * 16;
        sdst[i] = (c4 >> 1) * CPL + (iy * RS + ix + IW * p) * 16 + (c4 & 1) * 8;
    }

    // ---- tile cursors (wave-uniform): the fetch cursor runs two tiles ahead of the compute cursor
    struct Cursor { int img, ty, tx; };
    auto cursor_at = [&](int t) {
        Cursor c;
        c.img = __builtin_amdgcn_readfirstlane(t / tiles_per_img);
        const int r = t - c.img * tiles_per_img;
        c.ty = __builtin_amdgcn_readfirstlane(r / tiles_per_row);
        c.tx = r - c.ty * tiles_per_row;
        return c;
    };
    auto advance = [&](Cursor &c) {
        if (++c.tx == tiles_per_row) {
            c.tx = 0;
            if ((++c.ty) * tiles_per_row == tiles_per_img) { c.ty = 0; ++c.img; }
        }
    };
    f32x4 pv[NIT];
    // loads of one tile into pv, in three parts: fetch_setup (scalars), fetch_addr and fetch_load per staging item
    int f_oy = 0, f_ox = 0, f_base = 0;
    __amdgpu_buffer_rsrc_t f_rsrc = make_rsrc(a.src[0].p, 0u);
    auto fetch_setup = [&](const Cursor &c) {
        f_oy = c.ty * 4; f_ox = c.tx * 16;
        f_rsrc = make_rsrc(a.src[0].p + (size_t)c.img * a.Hin * a.Win * 64, (unsigned)(a.Hin * a.Win * 64) * 4u);
        f_base = (f_oy * a.Win + f_ox) * 256;
    };
    int f_voff[NIT];
    auto fetch_addr = [&](int i) {
        const int y = f_oy + (int)(short)(rel[i] & 0xffff), x = f_ox + (rel[i] >> 16);
        // (bitwise, unsigned: one compare per coordinate, no branch -- see conv_bf16_kernel)
        const bool ok = ((unsigned)y < (unsigned)a.Hin) & ((unsigned)x < (unsigned)a.Win);
        int off = f_base + relb[i];
        asm volatile("" : "+v"(off));
        f_voff[i] = ok ? off : BUF_OOB;
    };
    auto fetch_load = [&](int i) {
        if constexpr (EXP & 4) asm volatile("" : "+v"(pv[i]) : "v"(f_voff[i]));
        else pv[i] = buf_load4(f_rsrc, f_voff[i], 0);
    };
    float a_scale = 1.f, omul = 1.f;
    // LZ: max(fma(y, la, lb), 0) of the thread's channel quad, the operand scale folded into the coefficients (set with
    // a_scale below); padding stays 0: f_voff[i] == BUF_OOB.  The split then runs with scale 1 -- the same bits as the split
    // of the stored activation (conv_mfma.h, ConvSrc).
    static_assert(!LZ || !BM, "lazy sources: forward launches");
    f32x4 lzA = {0.f, 0.f, 0.f, 0.f}, lzB = lzA;
    if constexpr (LZ) {
        lzA = reinterpret_cast<const f32x4 *>(a.src[0].la)[c4];
        lzB = reinterpret_cast<const f32x4 *>(a.src[0].lb)[c4];
    }
    float lz_cap = 0.f;
    // conversion of item i (in pv) into image buffer `buf`, in two halves (each slotted behind one MFMA)
    unsigned st_h01 = 0, st_l01 = 0;
    auto stage_a = [&](int i) {
        if constexpr (EXP & 2) asm volatile("" ::"v"(pv[i]));
        else if constexpr (LZ) {
            lz_cap = f_voff[i] != BUF_OOB ? __builtin_inff() : 0.f;
            wres_split_pair(lazy_act(pv[i][0], lzA[0], lzB[0], lz_cap), lazy_act(pv[i][1], lzA[1], lzB[1], lz_cap), 1.f, st_h01, st_l01);
        } else wres_split_pair(pv[i][0], pv[i][1], a_scale, st_h01, st_l01);
    };
    auto stage_b = [&](int i, int buf) {
        if constexpr (EXP & 2) return;
        unsigned h23, l23;
        if constexpr (LZ)
            wres_split_pair(lazy_act(pv[i][2], lzA[2], lzB[2], lz_cap), lazy_act(pv[i][3], lzA[3], lzB[3], lz_cap), 1.f, h23, l23);
        else
        wres_split_pair(pv[i][2], pv[i][3], a_scale, h23, l23);
        unsigned char *dst = lds_raw + buf * TILE + sdst[i];
        u32x2_t hv, lv;
        hv[0] = st_h01; hv[1] = h23; lv[0] = st_l01; lv[1] = l23;
        *reinterpret_cast<u32x2_t *>(dst) = hv;
        *reinterpret_cast<u32x2_t *>(dst + PIECE) = lv;
    };

    // ---- prologue: tile t_begin -> buffer 0, tile t_begin + 1 -> registers
    Cursor cf = cursor_at(t_begin), cc = cf;
    fetch_setup(cf);
#pragma unroll
    for (int i = 0; i < NIT; ++i) { fetch_addr(i); fetch_load(i); }
    {   // operand scales (behind the loads): 2^e_a for the staging, the exact inverse of both scales for the epilogue
        unsigned v = lane < AMAX_SUB ? am_raw : 0u;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o); v = t > v ? t : v; }
        const int ea = f16_scale_exp((unsigned)__builtin_amdgcn_readfirstlane((int)v));
        const int ew = f16_scale_exp(*a.amax_w);
        a_scale = exp2i(ea);
        omul = exp2i(-ea) * exp2i(-ew);
        if constexpr (LZ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { lzA[j] *= a_scale; lzB[j] *= a_scale; }
        }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) { stage_a(i); stage_b(i, t_begin & 1); }
    int tf = t_begin;                  // tile index the fetch cursor stands on
    if (tf + 1 < t_end) { ++tf; advance(cf); }
    fetch_setup(cf);
#pragma unroll
    for (int i = 0; i < NIT; ++i) { fetch_addr(i); fetch_load(i); }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const float sc = coef.sc[0] * omul, bi = coef.bi[0], sh = coef.sh[0], ma = coef.ma[0], mb = coef.mb[0];
    const int a_off = g * CPL + ((li >> 3) * RS + (li & 7) + IW * wp) * 16;
    float vmax = 0.f;                  // max |stored value| of this lane over all its tiles (ConvArgs::amax_out)

    // ---- the streamed epilogue: sums of the finished tile in Y, its place in e_*; one row (= r of the MFMA layout: pixel
    //      (oy0 + (r >> 2), ox0 + (r & 3) + 4 g)) per call, operations and their order as in conv_epilogue
    float Y[16];
    __amdgpu_buffer_rsrc_t e_out = make_rsrc(a.out, 0u), e_res = e_out, e_y = e_out, e_z = e_out;
    int e_sout = 0, e_sres = 0, e_patch = 0, e_img = 0, e_szb = 0;
    bool e_live = true;
    float tmax = 0.f;                  // max |v| of the tile whose epilogue runs
    const int v_zb = nok ? (4 * g * (a.Cout >> 5) + (ncol >> 5)) * 4 : BUF_OOB;     // bit-packed mask: this lane's word of pixel 4g
    float ssum = 0.f, ssq = 0.f;
    // operands of the rows in flight: requested LEAD K-steps before their row is finished.  Three steps (~400 cycles) cover an
    // L2 hit; the residual of the eval forward's two BasicBlock launches comes from HBM (1-2 us under load): with a lead of
    // three they took 314 us against 188 for the same layer without residual.  The backward-statistics variants keep three
    // (their rings would need 2-3 x 11 registers the kernel does not have).
    constexpr int LEAD = (RES && !BM) ? 10 : 3, RING = LEAD + 1;
    float rvr[RING], yvr[RING], zvr[RING];
    auto epi_setup = [&](const Cursor &c) {
        const int oy0 = c.ty * 4, ox0 = c.tx * 16 + 8 * wp;
        e_img = c.img;
        e_patch = c.ty * a.ppr + 2 * c.tx + wp;
        // the second patch of a row's last tile can lie outside the image (widths of an odd number of patches): its stores
        // meet an empty buffer, its statistics and its max |v| are dropped (wave-uniform: a wave owns a patch)
        e_live = ox0 < a.Wout;
        e_out = make_rsrc(a.out + (size_t)c.img * a.o_img, e_live ? (unsigned)a.o_img * 4u : 0u);
        if constexpr (RES) e_res = make_rsrc(a.res + (size_t)c.img * a.r_img, (unsigned)a.r_img * 4u);
        if constexpr (BM) e_y = make_rsrc(a.bm_y + (size_t)c.img * a.o_img, (unsigned)a.o_img * 4u);
        if constexpr (BM && ZMASK) {
            if (a.bm_zbits)      // bit-packed mask (ConvArgs::bm_zbits): [pixel][Cout / 32] words
                e_z = make_rsrc(a.bm_zbits + (size_t)c.img * a.Hout * a.Wout * (a.Cout >> 5), (unsigned)(a.Hout * a.Wout * (a.Cout >> 5)) * 4u);
            else
                e_z = make_rsrc(a.bm_z + (size_t)c.img * a.o_img, (unsigned)a.o_img * 4u);
            e_szb = ((oy0 * a.Wout + ox0) * (a.Cout >> 5)) * 4;
        }
        e_sout = (oy0 * a.o_row + ox0 * a.o_px) * 4;
        e_sres = (oy0 * a.r_row + ox0 * a.r_px) * 4;
        ssum = 0.f; ssq = 0.f; tmax = 0.f;
    };
    auto epi_load = [&](int r) {
        if constexpr (RES) rvr[r % RING] = buf_load1(e_res, v_res, e_sres + ((r >> 2) * a.r_row + (r & 3) * a.r_px) * 4);
        if constexpr (BM) yvr[r % RING] = buf_load1(e_y, v_bm, e_sout + ((r >> 2) * a.o_row + (r & 3) * a.o_px) * 4);
        if constexpr (BM && ZMASK) {
            if (a.bm_zbits) {
                const unsigned w = __builtin_bit_cast(unsigned, buf_load1(e_z, v_zb, e_szb + ((r >> 2) * a.Wout + (r & 3)) * (a.Cout >> 5) * 4));
                zvr[r % RING] = ((w >> (ncol & 31)) & 1u) ? 1.f : 0.f;
            } else {
                zvr[r % RING] = buf_load1(e_z, v_bm, e_sout + ((r >> 2) * a.o_row + (r & 3) * a.o_px) * 4);
            }
        }
    };
    // a row in three parts, one per MFMA gap of its K-step: (1) value + mask, (2) statistics, ReLU, max |v|, (3) the store
    float e_v = 0.f;
    auto epi_row_a1 = [&](int r) {
        float v = __builtin_fmaf(Y[r], sc, bi);
        if constexpr (RES) v += rvr[r % RING];
        if constexpr (BM) {
            const bool on = ZMASK ? zvr[r % RING] > 0.f : (bm_relu == 0 || __builtin_fmaf(yvr[r % RING], ma, mb) > 0.f);
            v = on ? v : 0.f;
        }
        e_v = v;
    };
    auto epi_row_a2 = [&](int r) {
        float v = e_v;
        if constexpr (BM) {
            ssum += v;
            ssq = __builtin_fmaf(v, yvr[r % RING], ssq);
        } else if constexpr (STATS) {
            const float d = v - sh;
            ssum += d;
            ssq += d * d;      // (two roundings: contraction is off, as in conv_epilogue)
        }
        v = fmaxf(v, floor_v);
        tmax = fmaxf(tmax, fabsf(v));
        e_v = v;
    };
    auto epi_row_b = [&](int r) { buf_store1(e_v, e_out, v_out, e_sout + ((r >> 2) * a.o_row + (r & 3) * a.o_px) * 4); };
    auto epi_finish = [&] {
        vmax = fmaxf(vmax, e_live ? tmax : 0.f);
        if constexpr (BM || STATS) {
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (g == 0 && nok && e_live) {
                float *dst = a.stats + (((size_t)e_img * a.ppi + e_patch) * a.CoutP + ncol) * 2;
                dst[0] = ssum;
                dst[1] = ssq;
            }
        }
    };

    // ---- one tile: the 36 K-steps out of image buffer t & 1; in the gaps (EPI) the epilogue of the previous tile, the
    //      staging of tile t + 1 and the addresses of tile t + 2
    f32x16 acc, accm;
    auto tile_body = [&](auto epi_c, int t, const Cursor &cn) {
        constexpr bool EPI = decltype(epi_c)::value && !(EXP & 8);
        const int cur = t & 1, nxt = cur ^ 1;
        const unsigned char *abase = lds_raw + cur * TILE + a_off;
        auto load_a = [&](h16x8(&dst)[2], int s) {
            const int c = s / 18, tap = (s % 18) / 2, m = s % 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr (EXP & 16) asm volatile("" : "+v"(dst[q]));
                else dst[q] = *reinterpret_cast<const h16x8 *>(abase + q * PIECE + (4 * c + 2 * m) * CPL + ((tap / 3) * RS + tap % 3) * 16);
            }
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accm[r] = 0.f; }
        // fragments are requested TWO K-steps ahead (ring of three): one step -- 96 cycles of matrix work -- does not cover
        // an LDS round trip when the wave is alone on its SIMD
        h16x8 afr[3][2];
        load_a(afr[0], 0);
        load_a(afr[1], 1);
        constexpr int S0 = 20;         // first staging step
        static_assert(S0 + 2 * NIT <= 36, "staging and address slots inside the K loop");
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 2 < 36) load_a(afr[(s + 2) % 3], s + 2);
            const h16x8 ah = afr[s % 3][0], al = afr[s % 3][1];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (EXP & 1) asm volatile("" : "+v"(accm) : "v"(al), "v"(breg[s][0]));
            else accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, breg[s][0], accm, 0, 0, 0);      // l * h
            __builtin_amdgcn_sched_barrier(0);
            if (EPI && s >= LEAD && s < LEAD + 16) epi_row_a1(s - LEAD);
            if (s >= S0 && s < S0 + NIT) stage_a(s - S0);
            if (s == S0 + NIT) fetch_setup(cn);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (EXP & 1) asm volatile("" : "+v"(acc) : "v"(ah), "v"(breg[s][0]));
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, breg[s][0], acc, 0, 0, 0);        // h * h
            __builtin_amdgcn_sched_barrier(0);
            if (EPI && s >= LEAD && s < LEAD + 16) epi_row_a2(s - LEAD);
            if (s >= S0 && s < S0 + NIT) stage_b(s - S0, nxt);
            if (s >= S0 + NIT && s < S0 + 2 * NIT) fetch_addr(s - S0 - NIT);
            __builtin_amdgcn_sched_barrier(0);
            // h * l: the lo pieces of the weights feed the MFMA from the accumulation registers they live in
            if constexpr (EXP & 1) asm volatile("" : "+a"(accm) : "v"(ah), "a"(breg[s][1]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(accm) : "v"(ah), "a"(breg[s][1]));
            __builtin_amdgcn_sched_barrier(0);
            if (EPI && s >= LEAD && s < LEAD + 16) epi_row_b(s - LEAD);
            if (EPI && s < 16) epi_load(s);          // operands of row s, due in LEAD K-steps
            if (EPI && s == LEAD + 16) epi_finish();
        }
        // (the last MFMA is inline asm: hipcc does not pad its result hazard -- 8 passes: 12 wait states before a VALU read)
        asm volatile("s_nop 15" : "+a"(accm));
    };

    Cursor cp = cc;                    // the tile whose sums wait in Y
#ifdef MC_PHASE_TIMERS
    unsigned long long tp_tail = 0, tp_bar = 0, tp_mfma = 0, tp_head = 0;
    const unsigned long long tp_start = __builtin_readcyclecounter();
#define WTP_NOW() __builtin_readcyclecounter()
#else
#define WTP_NOW() 0ull
#endif
    for (int t = t_begin; t < t_end; ++t) {
        [[maybe_unused]] const unsigned long long tp0 = WTP_NOW();
        // the tile after next (clamped: past the end the last tile is fetched again and never used)
        const bool more = tf + 1 < t_end;
        Cursor cn = cf;
        if (more) advance(cn);
        const int tn = more ? tf + 1 : tf;
        [[maybe_unused]] unsigned long long tp1 = tp0;
        if (t == t_begin) {
            tile_body(std::false_type{}, t, cn);
        } else {
            if constexpr (!(EXP & 8)) epi_setup(cp);
            tp1 = WTP_NOW();
            tile_body(std::true_type{}, t, cn);
        }
        [[maybe_unused]] const unsigned long long tp2 = WTP_NOW();
        cf = cn; tf = tn;
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[r] = acc[r] + accm[r];
        cp = cc;
        advance(cc);
#pragma unroll
        for (int i = 0; i < NIT; ++i) fetch_load(i);
        [[maybe_unused]] const unsigned long long tp3 = WTP_NOW();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef MC_PHASE_TIMERS
        { const unsigned long long tp4 = WTP_NOW(); tp_head += tp1 - tp0; tp_mfma += tp2 - tp1; tp_tail += tp3 - tp2; tp_bar += tp4 - tp3; }
#endif
    }
#ifdef MC_PHASE_TIMERS
    if (a.phase_prof && lane == 0) {       // (reported by mc_bench_conv as: stage = tile tail, barriers, mfma-phase, epilogue = tile head)
        unsigned long long *o = a.phase_prof + ((size_t)blockIdx.x * 4 + wave) * 5;
        o[0] = tp_tail; o[1] = tp_bar; o[2] = tp_mfma; o[3] = tp_head; o[4] = WTP_NOW() - tp_start;
    }
#endif
    // the last tile's epilogue, on its own
    if constexpr (!(EXP & 8)) {
        epi_setup(cp);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r) epi_load(r);
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r) { epi_row_a1(r); epi_row_a2(r); epi_row_b(r); }
        }
        epi_finish();
    } else {
        asm volatile("" ::"v"(Y[0]), "v"(Y[15]));
    }
    if (a.amax_out) amax_update_wave(a.amax_out, vmax);
}

// eligible: f16x2 arithmetic on fp32 sources, 3x3 stride 1, ONE 64-channel source, dense NHWC output of whole 4x8 patches (a row
// whose width is an odd number of patches -- KITTI's 312-wide quarter-resolution maps -- ends in a tile whose second patch lies
// outside the image: it is computed like any other and nothing of it is kept)
bool conv_wres_ok(const ConvArgs &a, int ks, int stride) {
    if (a.prec != 3 || !a.wpk16 || !a.amax_w || ks != 3 || stride != 1) return false;
    if (a.nsrc != 1 || a.src[0].C != 64 || a.Cin != 64 || !a.amax_in[0]) return false;
    if (a.CoutP % 64 || a.Hin != a.Hout || a.Win != a.Wout || a.Wout % 8 || a.Hout % 4) return false;     // whole 4x8 patches
    if ((size_t)a.Hin * a.Win * 64 * 4 >= ((size_t)1 << 31)) return false;       // 32-bit buffer offsets per image
    if (a.bm_y && !a.stats) return false;
    return true;
}

template <bool BM, bool RES, bool STATS, bool ZMASK, int EXP = 0, bool LZ = false>
static hipError_t launch_wres_one(const ConvArgs &a, int tpr, int tpi, int total, int per_wg, int ngroups, int nwg, hipStream_t st) {
    auto kern = conv_wres_kernel<BM, RES, STATS, ZMASK, EXP, LZ>;
    static DynLdsOnce attr_set;
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(kern), (int)(WresCfg::LDS_BYTES));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nwg * ngroups)), dim3(256), WresCfg::LDS_BYTES, st, a, tpr, tpi, total, per_wg, ngroups);
    return hipGetLastError();
}

hipError_t launch_conv_wres(const ConvArgs &a_in, int ks, int stride, hipStream_t st, ConvArgs *resolved) {
    if (!conv_wres_ok(a_in, ks, stride)) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = a.ppi;
    if (resolved) *resolved = a;
    const int tpr = (a.Wout + 15) / 16, tpi = tpr * (a.Hout / 4), total = a.B * tpi, ngroups = a.CoutP / 64;
    // one workgroup per CU and column group where the work allows (>= 8 tiles each: the weight prologue is ~2 tiles of time)
    // (the CU count of the CURRENT device, remembered per device: handles on different GPUs share this process -- ADVICE r5)
    static std::atomic<int> ncu_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int ncu = ncu_of[dev].load(std::memory_order_relaxed);
    if (ncu <= 0) {
        int n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        ncu = n > 0 ? n : 256;
        ncu_of[dev].store(ncu, std::memory_order_relaxed);
    }
    int nwg = ncu;
    if (total < nwg * 8) nwg = (total + 7) / 8;
    if (nwg < 1) nwg = 1;
    const int per_wg = (total + nwg - 1) / nwg;
    nwg = (total + per_wg - 1) / per_wg;
#define WL(BM_, RES_, STATS_, ZM_, EXP_) launch_wres_one<BM_, RES_, STATS_, ZM_, EXP_>(a, tpr, tpi, total, per_wg, ngroups, nwg, st)
#ifdef MC_WRES_EXP
    static const int exp_bits = [] { const char *e = std::getenv("MONOCON_WRES_EXP"); return e ? std::atoi(e) : 0; }();
    switch (exp_bits) {
#define WX(N) case N: return WL(false, false, false, false, N);
        WX(1) WX(2) WX(4) WX(8) WX(16) WX(6) WX(14) WX(30) WX(31)
#undef WX
        default: break;
    }
#endif
    const bool res = a.res != nullptr, stats = a.stats != nullptr;
    if (a.src[0].la) {       // lazy source: forward launches (no backward-statistics twin reads an activation)
        if (a.bm_y || !a.src[0].lb) return hipErrorInvalidValue;
#define WLZ(RES_, STATS_) launch_wres_one<false, RES_, STATS_, false, 0, true>(a, tpr, tpi, total, per_wg, ngroups, nwg, st)
        return res ? (stats ? WLZ(true, true) : WLZ(true, false)) : (stats ? WLZ(false, true) : WLZ(false, false));
#undef WLZ
    }
    if (a.bm_y) {
        const bool zm = a.bm_relu == 1;
        return res ? (zm ? WL(true, true, true, true, 0) : WL(true, true, true, false, 0))
                   : (zm ? WL(true, false, true, true, 0) : WL(true, false, true, false, 0));
    }
    return res ? (stats ? WL(false, true, true, false, 0) : WL(false, true, false, false, 0))
               : (stats ? WL(false, false, true, false, 0) : WL(false, false, false, false, 0));
#undef WL
}

}  // namespace mc
