// Mixed-precision variant of the fused convolution (BASELINE config 3: "bf16"): activations and master
// weights stay fp32 in HBM, both MFMA operands are rounded to bf16 (RNE) on their way into the matrix
// pipe, accumulation / scale / bias / residual / ReLU / statistics stay fp32.
//
//   v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate, so the kernel turns from MFMA-bound into an
//   L2 / LDS-bound one; everything that already made the fp32 kernel lean on address arithmetic carries
//   over (buffer descriptors, static lane offsets, SGPR chunk offsets, wave-uniform epilogue).
//
// Same contract, tiling and epilogue as conv_mfma_kernel (conv_mfma.h); differences:
//   * the halo tile lives in LDS as bf16; a thread converts the float4 it loaded with two v_cvt_pk_bf16_f32 and
//     writes 8 bytes; one ds_read_b128 is the whole A operand of one MFMA (pixel li, channels 8g..8g+7 of a K=16 step).
//     Stride-1 layout (round 2): chunk-planar, [16-byte channel chunk c][patch pair][halo row iy][24 slots of 16 bytes],
//     pixel (iy, ix) of patch 2*pp + j in slot ix + IW*j.  Why: the LDS serves a ds_read_b128 in four fixed groups of 16
//     lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same +32 (MI355X_MICROARCH.md, LDS) -- i.e. pixels
//     (0, 0..3), (1, 4..7), (2, 4..7), (3, 0..3) of the 4x8 patch in one group.  With row-major [pixel][CK + 8] rows
//     (round 1) those 16 reads fell on only 6..8 of the 16 bank windows: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
//     0.64, the LDS array ~3x busier than the data needs.  With a row stride of 24 slots (= 8 mod 16) the window of
//     a read is (8*y + x + const) mod 16: a bijection on both lane groups for every tap -- conflict-free.  The chunk
//     planes are skewed by 32 bytes, so the 8-byte staging stores of a pixel's four chunks do not share banks either.
//     Stride-2 layers (five in DLA-34) keep the row-major [pixel][CK + 8] layout.
//   * weights are pre-packed as bf16 panels [tap][Cin/8][CoutP][8] (mc_pack_params), one 16-byte load
//     per 32 output channels and MFMA.
// SPL = 3 (mc_set_precision(h, 2)): fp32 EMULATION.  Each fp32 operand is split into three bf16 pieces
// (x = h + m + l exactly to 24 bits: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)) -- activations while they
// are staged into three LDS planes, weights when the panels are packed -- and every product a*b is the sum of the six
// partial products of weight >= 2^-24 (h*l, l*h, m*m, h*m, m*h, h*h; each exact in the fp32 accumulator, added small
// to large).  Six bf16 MFMAs cost 192 cycles per K=16 against 512 for eight fp32 MFMAs, and the result is as close
// to the fp64 reference as the fp32 FMA chain (measured: worst prediction map 5.0e-5 vs 5.3e-5, scratch/emu_split.py).
//
// SPL = 2 (mc_set_precision(h, 3)): fp32 emulation with HALF the matrix work of SPL = 3.  fp16 carries 11 significant bits
// against bf16's 8, so TWO pieces already hold 22 bits: x*s = h + l, h = fp16(x*s), l = fp16(x*s - h), and a product
// needs three partial products (l*h, h*l, h*h -- each exact in the fp32 accumulator) instead of six, two LDS planes
// instead of three.  fp16's narrow exponent is what the power-of-two scale s is for: every operand TENSOR is scaled so
// that its max |x| sits in [2^14, 2^15) (ConvArgs::amax_in / amax_w hold the maxima, written by the producers of the
// tensors; weights are scaled when their panels are packed) and the accumulator is multiplied by the exact inverse
// 2^-(e_a + e_w) in the epilogue.  Elements below 2^-18 of their tensor's maximum lose relative (not absolute)
// precision: their absolute error is 2^-40 of the maximum.  Measured through the oracle (scratch/emu_f16x2.py): worst
// prediction map 5.5e-5 from the fp64 reference (native fp32 5.3e-5, bf16x3 5.0e-5).
//
// SPL = 1 is not the parity path: results differ from the fp32 reference by bf16 operand rounding (~1e-3
// norm-wise per layer); tests/test_hip_bf16.py states the tolerance.  Selected per handle with
// mc_set_precision(h, 1); layers whose sources are not multiples of 32 channels stay on the fp32 kernels.
#include "conv_mfma.h"

namespace mc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// element type / vectors / MFMA of a split mode: SPL 1, 3 = bf16 pieces, SPL 2 = fp16 pieces
template <int SPL> struct Piece { typedef __bf16 T; typedef bf16x8 V8; typedef bf16x4 V4; };
template <> struct Piece<2> { typedef _Float16 T; typedef f16x8 V8; typedef f16x4 V4; };
__device__ __forceinline__ f32x16 mfma_k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int KS, int S, int WM, int WN, int WTM, int WTN, int SPL>
struct ConvCfgB16 {
    static constexpr int CK = 32;
    static constexpr int PB = WM * WTM, BNT = WN * WTN * 32, NT = 64 * WM * WN;
    static constexpr int KH = win_h(KS), KW = win_w(KS), PAD = win_pad(KS);
    static constexpr int IH = 3 * S + KH, IW = 7 * S + KW, NPIX = IH * IW;
    static constexpr int ROWB = (CK + 8) * 2;                      // S == 2: bytes per staged pixel (row-major layout)
    // S == 1: chunk-planar layout (see the header)
    static constexpr bool PLANAR = S == 1;
    static constexpr int RS = 24;                                  // slots per halo row: 2 patches x IW <= 24, 24 = 8 mod 16
    static constexpr int PPB = IH * RS * 16;                       // bytes of one patch pair in one chunk plane
    static constexpr int CPL = ((PB + 1) / 2) * PPB + 32;          // chunk plane (+32: de-phases the four chunks' stores)
    static_assert(!PLANAR || (2 * IW <= RS && CK == 32), "two patches per 24-slot row, four 16-byte chunks per pixel");
    static constexpr int PLANE_BYTES = PLANAR ? 4 * CPL : PB * NPIX * ROWB;   // one bf16 piece of the halo tile
    static constexpr int TILE_BYTES = SPL * PLANE_BYTES;
    static constexpr size_t LDS_BYTES = TILE_BYTES + PB * 16;
};

// LZ: the launch has lazy sources (ConvSrc::la) -- its own instantiation, so that every other launch keeps its registers
// (the coefficients cost ~10 of them, which pushed the 128 x 64 tilings into scratch) and its instruction stream
template <int KS, int S, int WM, int WN, int WTM, int WTN, int SPL, bool BM = false, bool LZ = false>
__global__ __launch_bounds__(64 * WM * WN, SPL == 1 ? 3 : 2) void conv_bf16_kernel(const ConvArgs a) {
    static_assert(!LZ || (SPL == 2 && !BM), "lazy sources: forward launches of the fp16-split mode");
    using Cfg = ConvCfgB16<KS, S, WM, WN, WTM, WTN, SPL>;
    typedef typename Piece<SPL>::T pc_t;
    typedef typename Piece<SPL>::V8 pc8;
    typedef typename Piece<SPL>::V4 pc4;
    constexpr int PLANE = Cfg::PLANE_BYTES;
    constexpr int CK = Cfg::CK, PB = Cfg::PB, BNT = Cfg::BNT, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, ROWB = Cfg::ROWB;
    constexpr bool PLANAR = Cfg::PLANAR;
    constexpr int RS = Cfg::RS, PPB = Cfg::PPB, CPL = Cfg::CPL;
    constexpr int C4 = CK / 4;
    static_assert(NT % C4 == 0, "a thread keeps one channel group across its staging elements");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int *pinfo = reinterpret_cast<int *>(lds_raw + Cfg::TILE_BYTES);   // [PB][4] = b, oy0, ox0, valid

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

    // SPL == 2: the max-|x| words of the input tensor(s) are REQUESTED first thing and reduced only when the first staged
    // tile is about to be converted (below, behind the weight / coefficient / staging loads): read where the scale is
    // declared, every workgroup started with one exposed memory round trip per source of the virtual concat -- a short-K
    // layer (64 channels: 3.6 us of matrix work per workgroup) has nothing to hide it behind.
    unsigned am_raw[4] = {0u, 0u, 0u, 0u};
    if constexpr (SPL == 2) {
        const int al = lane < AMAX_SUB ? lane * AMAX_STRIDE : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) am_raw[i] = a.amax_in[i < a.nsrc ? i : 0][al];      // (extra copies of source 0: harmless)
    }

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

    // SPL == 3: the five minor partial products (weight <= 2^-8) accumulate apart from the h*h products, so that the
    // main accumulator takes exactly as many additions as the fp32 MFMA path (its round-off, not the exactness of the
    // products, is what limits the emulation) and the minor sum's round-off is 2^-8 smaller; folded in once at the end.
    // SPL == 2: operand scale 2^e_a (activations; one scale for all sources of a virtual concat: the largest maximum
    // decides) and the exact inverse of both scales for the epilogue -- set by operand_scales() below
    float a_scale = 1.f, omul = 1.f;
    auto operand_scales = [&] {
        if constexpr (SPL == 2) {
            unsigned v = am_raw[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) v = am_raw[i] > v ? am_raw[i] : v;
            v = lane < AMAX_SUB ? v : 0u;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o); v = t > v ? t : v; }
            const int ea = f16_scale_exp((unsigned)__builtin_amdgcn_readfirstlane((int)v));
            const int ew = f16_scale_exp(*a.amax_w);
            a_scale = exp2i(ea);
            omul = exp2i(-ea) * exp2i(-ew);
        }
    };
    constexpr int NACC = SPL >= 2 ? 2 : 1;
    f32x16 acc[WTM][WTN], accm[NACC == 2 ? WTM : 1][NACC == 2 ? WTN : 1];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[tm][tn][r] = 0.f;
                if (NACC == 2) accm[tm][tn][r] = 0.f;
            }

    constexpr int TOTAL = PB * NPIX * C4;
    constexpr int NIT = (TOTAL + NT - 1) / NT;
    const int c4 = tid % C4;
    unsigned char *stage_dst = lds_raw + (tid / C4) * ROWB + c4 * 8;     // row-major layout (S == 2)

    int a_off[WTM];   // byte offsets of this lane's fragment rows (tap (0, 0), K step 0)
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm) {
        const int p = wm * WTM + tm;
        a_off[tm] = PLANAR ? g * CPL + (p >> 1) * PPB + ((li >> 3) * RS + (li & 7) + IW * (p & 1)) * 16
                           : (p * NPIX + ((li >> 3) * S) * IW + (li & 7) * S) * ROWB + g * 16;
    }

    const int Cin8 = a.Cin >> 3;
    const int w_plane = Cfg::KH * Cfg::KW * a.Cin * a.CoutP * 2;   // bytes of one bf16 piece of the panel
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpk16, (unsigned)(SPL * w_plane));
    const int w_lane = (g * a.CoutP + n0 + wn * WTN * 32 + li) * 16;   // bytes

    constexpr int KSTEPS = CK / 16, NS = Cfg::KH * Cfg::KW * KSTEPS;
    auto load_b = [&](pc8(&dst)[SPL][WTN], int kc, int s) {
        const int tap = s / KSTEPS, m = s % KSTEPS;
        const int soff = (tap * Cin8 + ((kc + m * 16) >> 3)) * a.CoutP * 16;
#pragma unroll
        for (int q = 0; q < SPL; ++q)
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn)
                dst[q][tn] = __builtin_bit_cast(
                    pc8, __builtin_amdgcn_raw_buffer_load_b128(r_w, w_lane + tn * 32 * 16, soff + q * w_plane, 0));
    };
    auto load_a = [&](pc8(&dst)[SPL][WTM], int s) {
        const int tap = s / KSTEPS, m = s % KSTEPS;
#pragma unroll
        for (int q = 0; q < SPL; ++q)
#pragma unroll
            for (int tm = 0; tm < WTM; ++tm)
                dst[q][tm] = *reinterpret_cast<const pc8 *>(
                    lds_raw + q * PLANE + a_off[tm] +
                    (PLANAR ? 2 * m * CPL + ((tap / Cfg::KW) * RS + (tap % Cfg::KW)) * 16
                            : ((tap / Cfg::KW) * IW + (tap % Cfg::KW)) * ROWB + m * 32));
    };
    // B (weight) fragments come straight from L2 and are fetched DB - 1 K-steps ahead into a register ring: slot s % DB
    // holds step s.  One step ahead (round 2) covered 192 cycles of MFMA work for the one-tile-wide shapes -- less than an
    // L2 round trip under load, so every step stalled on its weights (PMC: waves parked 40 % of the time); and the loads
    // return in order, so a weight fetch issued behind the staging prefetch of the next chunk waits for HBM.
    // Depth: as many slots as 48 registers hold (a slot is SPL * WTN fragments of 4 registers), among the divisors of NS.
#ifndef MC_CONV_DB_REGS
#define MC_CONV_DB_REGS 48
#endif
    constexpr int DB_FIT = MC_CONV_DB_REGS / (SPL * WTN * 4);
    constexpr int DB = (WTM * WTN > 2) ? 2 : ((DB_FIT >= 6 && NS % 6 == 0) ? 6 : ((DB_FIT >= 3 && NS % 3 == 0) ? 3 : 2));
    static_assert(NS % DB == 0, "ring slots must line up across K-chunks");
    pc8 bq[DB][SPL][WTN];
#pragma unroll
    for (int j = 0; j < DB - 1; ++j) load_b(bq[j], 0, j);
    const EpiCoef<WTN> coef = conv_epi_coef<WN, WTN, BM>(a, n0, wn, li);      // (see conv_mfma.h: before the K loop, not after)

    // ---- K loop: one iteration per 32-channel chunk of the virtual concat.  Where the registers allow it (PF), the raw
    //      fp32 data of chunk i+1 is fetched into registers BEFORE the MFMA phase of chunk i and converted / written to
    //      LDS after it: the HBM / L2 latency of the staging loads is hidden behind this workgroup's own matrix work
    //      instead of relying on the co-resident workgroup being in its MFMA phase at the right moment.
#ifndef MC_CONV_PF
#define MC_CONV_PF 1
#endif
    constexpr bool PF = MC_CONV_PF && NIT <= 8 && WTM * WTN <= 2;
    constexpr int UB = NIT > 8 ? 8 : NIT;
    f32x4 pv[PF ? NIT : 1];
    int voff[NIT], sdst[NIT];
    int si = 0, c0 = 0, kc = 0;
    int Cs = a.src[0].C;
    // Lazy sources (LZ): the chunk being staged holds the producer's raw conv output y; its value max(fma(y, la, lb), 0) is
    // formed in stage_one, with the operand scale folded into the coefficients.  A thread stages ONE channel quad (c4) of
    // every pixel, so a chunk's coefficients are two float4 per thread: requested (lazy_fetch) right after the previous
    // chunk has been staged -- ahead of that chunk's MFMA phase -- and scaled (lazy_advance) when their chunk is staged.
    // Padding must stay 0 (not relu(lb)): voff[i] == BUF_OOB.
    f32x4 lzA = {0.f, 0.f, 0.f, 0.f}, lzB = lzA;
    bool lz_cur = false, lz_next = false;
    auto lazy_fetch = [&](int s_idx, int cc0) {       // coefficients of chunk (source s_idx, first channel cc0)
        if constexpr (LZ) {
            lz_next = a.src[s_idx].la != nullptr;
            if (lz_next) {
                lzA = reinterpret_cast<const f32x4 *>(a.src[s_idx].la + cc0)[c4];
                lzB = reinterpret_cast<const f32x4 *>(a.src[s_idx].lb + cc0)[c4];
            }
        }
    };
    auto lazy_advance = [&] {
        if constexpr (LZ) {
            lz_cur = lz_next;
            if (lz_cur) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { lzA[j] *= a_scale; lzB[j] *= a_scale; }
            }
        }
    };
    __amdgpu_buffer_rsrc_t r_in = make_rsrc(a.src[0].p + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
    auto lane_offsets = [&](int cs) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + NT * i;
            const int t = e / C4;
            const int pix = t % NPIX;
            const int p = (t / NPIX) % PB;
            const int iy = pix / IW, ix = pix % IW;
            const int y = pinfo[p * 4 + 1] * S - PAD + iy;
            const int x = pinfo[p * 4 + 2] * S - PAD + ix;
            // (bitwise, unsigned: `a && b && ...` becomes a chain of branches, each with its own LDS round trip for pinfo)
            const bool ok = (e < TOTAL) & (pinfo[p * 4 + 3] != 0) & ((unsigned)y < (unsigned)a.Hin) & ((unsigned)x < (unsigned)a.Win);
            int off = ((y * a.Win + x) * cs + c4 * 4) * 4;
            asm volatile("" : "+v"(off));      // computed for every lane, then selected: no branch around the multiplies
            voff[i] = ok ? off : BUF_OOB;
            sdst[i] = PLANAR ? (c4 >> 1) * CPL + (p >> 1) * PPB + (iy * RS + ix + IW * (p & 1)) * 16 + (c4 & 1) * 8
                             : (int)(stage_dst - lds_raw) + i * (NT / C4) * ROWB;
        }
    };
    auto stage_load = [&](__amdgpu_buffer_rsrc_t r, int vo, int so) -> f32x4 { return buf_load4(r, vo, so); };
    // split one float4 into its pieces and write them to the LDS planes.  LZ (a type: the caller branches ONCE per chunk on the
    // wave-uniform lz_cur, so a launch with mixed sources keeps the plain path for its materialised ones): the chunk is lazy
    auto stage_one = [&](const f32x4 &v, int i, auto lz_c) {
        constexpr bool LZC = decltype(lz_c)::value;
        if (i < NIT && (NT * (i + 1) <= TOTAL || tid + NT * i < TOTAL)) {
            pc4 q[SPL];
            [[maybe_unused]] const float cap = voff[i] != BUF_OOB ? __builtin_inff() : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = SPL == 2 ? (LZC ? lazy_act(v[j], lzA[j], lzB[j], cap) : v[j] * a_scale) : v[j];
#pragma unroll
                for (int pz = 0; pz < SPL; ++pz) {   // h = rnd(x), m = rnd(x - h), l = rnd(x - h - m)
                    q[pz][j] = (pc_t)r;
                    r -= (float)q[pz][j];
                }
            }
#pragma unroll
            for (int pz = 0; pz < SPL; ++pz) *reinterpret_cast<pc4 *>(lds_raw + sdst[i] + pz * PLANE) = q[pz];
        }
    };
#ifdef MC_PHASE_TIMERS
    unsigned long long tp_stage = 0, tp_bar = 0, tp_mfma = 0, tp_epi = 0;
    const unsigned long long tp_t0 = __builtin_readcyclecounter();
#define TP_NOW() __builtin_readcyclecounter()
#else
#define TP_NOW() 0ull
#endif
    lane_offsets(Cs);
    lazy_fetch(0, 0);
    if (PF) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) pv[i] = stage_load(r_in, voff[i], 0);
    }
    __builtin_amdgcn_sched_barrier(0);      // (the reduction must not be scheduled back up in front of the staging loads)
    operand_scales();
    for (bool first = true;; first = false) {
        [[maybe_unused]] unsigned long long tp_a = TP_NOW();
        if (!first) __syncthreads();
        [[maybe_unused]] unsigned long long tp_b = TP_NOW();
        lazy_advance();
        auto stage_chunk = [&](auto lz_c) {
            if (PF) {
#pragma unroll
                for (int i = 0; i < NIT; ++i) stage_one(pv[i], i, lz_c);
            } else {
#pragma unroll
                for (int i0 = 0; i0 < NIT; i0 += UB) {
                    f32x4 v[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        if (i0 + u < NIT) v[u] = stage_load(r_in, voff[i0 + u], c0 * 4);
#pragma unroll
                    for (int u = 0; u < UB; ++u) stage_one(v[u], i0 + u, lz_c);
                }
            }
        };
        if constexpr (LZ) {
            if (lz_cur) stage_chunk(std::true_type{});
            else stage_chunk(std::false_type{});
        } else {
            stage_chunk(std::false_type{});
        }
        [[maybe_unused]] unsigned long long tp_c = TP_NOW();
        __syncthreads();
        [[maybe_unused]] unsigned long long tp_d = TP_NOW();
        // the chunk after this one (possibly the first of the next source): descriptors / lane offsets now, and with PF
        // its loads go out before the MFMA phase
        int nsi = si, nc0 = c0 + CK;
        if (nc0 >= Cs) { ++nsi; nc0 = 0; }
        const bool more = nsi < a.nsrc;
        if (more && nsi != si) {
            Cs = a.src[nsi].C;
            r_in = make_rsrc(a.src[nsi].p + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
            lane_offsets(Cs);
        }
        if (more) lazy_fetch(nsi, nc0);
        if (PF && more) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) pv[i] = stage_load(r_in, voff[i], nc0 * 4);
        }
        {
            const int kc_next = (kc + CK < a.Cin) ? kc + CK : kc;
            pc8 acur[SPL][WTM];
            load_a(acur, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                pc8 anext[SPL][WTM];
                if (s + 1 < NS) load_a(anext, s + 1);
                {   // weights of step s + DB - 1 (of this chunk or the next) into the slot step s - 1 just released
                    const int t = s + DB - 1;
                    if (t < NS) load_b(bq[t % DB], kc, t);
                    else load_b(bq[t % DB], kc_next, t - NS);
                }
                __builtin_amdgcn_sched_barrier(0);
                // partial products, smallest first: (piece of A, piece of B) with weight 2^-8*(i+j) >= 2^-16 relative
                // (SPL == 2: l*h, h*l, h*h)
                constexpr int NP = SPL == 1 ? 1 : (SPL == 2 ? 3 : 6);
                constexpr int PA[6] = {SPL == 2 ? 1 : 0, SPL == 2 ? 0 : 2, SPL == 2 ? 0 : 1, 0, 1, 0};
                constexpr int PBv[6] = {SPL == 2 ? 0 : 2, SPL == 2 ? 1 : 0, SPL == 2 ? 0 : 1, 1, 0, 0};
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < WTN; ++tn) {
                            if (NACC == 2 && pp < NP - 1)
                                accm[tm][tn] = mfma_k16(acur[PA[pp]][tm], bq[s % DB][PBv[pp]][tn], accm[tm][tn]);
                            else
                                acc[tm][tn] = mfma_k16(acur[SPL == 1 ? 0 : PA[pp]][tm], bq[s % DB][SPL == 1 ? 0 : PBv[pp]][tn], acc[tm][tn]);
                        }
                if (s + 1 < NS) {
#pragma unroll
                    for (int q = 0; q < SPL; ++q)
#pragma unroll
                        for (int tm = 0; tm < WTM; ++tm) acur[q][tm] = anext[q][tm];
                }
            }
        }
#ifdef MC_PHASE_TIMERS
        { const unsigned long long tp_e = TP_NOW(); tp_bar += (tp_b - tp_a) + (tp_d - tp_c); tp_stage += tp_c - tp_b; tp_mfma += tp_e - tp_d; }
#endif
        if (!more) break;
        si = nsi; c0 = nc0; kc += CK;
    }
    [[maybe_unused]] const unsigned long long tp_f = TP_NOW();

    if (NACC == 2) {
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] += accm[tm][tn][r];
    }
    conv_epilogue<WM, WN, WTM, WTN, BNT, BM>(a, acc, pinfo, chunk * PB, img, n0, wm, wn, g, li, coef, omul);
#ifdef MC_PHASE_TIMERS
    if (a.phase_prof && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tp_g = TP_NOW();
        tp_epi = tp_g - tp_f;
        unsigned long long *o = a.phase_prof + ((size_t)blockIdx.x * (NT / 64) + wave) * 5;
        o[0] = tp_stage; o[1] = tp_bar; o[2] = tp_mfma; o[3] = tp_epi; o[4] = tp_g - tp_t0;
    }
#endif
}

// ---- weight packing: OIHW fp32 -> [tap][Cin/8][CoutP][8] bf16 (forward) and the dgrad variants
// (transposed + flipped per source, or one output-parity class of a stride-2 data gradient; same tap
// conventions as pack_conv_w_kernel / pack_conv_w_dgrad_kernel)
// nsplit == 2: fp16 pieces of w * 2^e_w (e_w from the weight tensor's max |w|, `amax`), else bf16 pieces of w
__device__ __forceinline__ void store_pieces(float r, unsigned short *dst, size_t plane, int nsplit) {
    if (nsplit == 2) {
        const _Float16 hi = (_Float16)r;
        const _Float16 lo = (_Float16)(r - (float)hi);
        dst[0] = __builtin_bit_cast(unsigned short, hi);
        dst[plane] = __builtin_bit_cast(unsigned short, lo);
    } else {
        for (int q = 0; q < nsplit; ++q) {
            const __bf16 piece = (__bf16)r;
            dst[q * plane] = __builtin_bit_cast(unsigned short, piece);
            r -= (float)piece;
        }
    }
}
__global__ void pack_conv_w_bf16_kernel(const float *__restrict__ w, int Cout, int Cin, int k, unsigned short *__restrict__ dst,
                                        int CinPanel, int CoutP, int n_off, int c_off, int nsplit, const unsigned *amax) {
    const size_t plane = (size_t)k * k * CinPanel * CoutP;
    const float wscale = nsplit == 2 ? exp2i(f16_scale_exp(*amax)) : 1.f;
    const int kk = k * k;
    const size_t total = (size_t)Cout * Cin * kk;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tap = e % kk;
        const int c = (e / kk) % Cin;
        const int n = e / ((size_t)kk * Cin);
        const int cc = c + c_off, nn = n + n_off;
        store_pieces(w[e] * wscale, dst + (((size_t)tap * (CinPanel >> 3) + (cc >> 3)) * CoutP + nn) * 8 + (cc & 7), plane, nsplit);
    }
}
hipError_t launch_pack_conv_w_bf16(const float *w, int Cout, int Cin, int k, void *dst, int CinPanel, int CoutP, int n_off,
                                   int c_off, int nsplit, hipStream_t st, const unsigned *amax) {
    if (nsplit == 2 && !amax) return hipErrorInvalidValue;
    const size_t total = (size_t)Cout * Cin * k * k;
    size_t gsz = (total + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(pack_conv_w_bf16_kernel, dim3((unsigned)gsz), dim3(256), 0, st, w, Cout, Cin, k,
                       static_cast<unsigned short *>(dst), CinPanel, CoutP, n_off, c_off, nsplit, amax);
    return hipGetLastError();
}
__global__ void pack_conv_w_dgrad_bf16_kernel(const float *__restrict__ w, int Cout, int CinTotal, int k, int c_off, int Cs,
                                              int CsP, int CoutPad, int cls, int nsplit, unsigned short *__restrict__ dst,
                                              const unsigned *amax) {
    const float wscale = nsplit == 2 ? exp2i(f16_scale_exp(*amax)) : 1.f;
    const size_t plane = (size_t)(cls < 0 ? k * k : (1 + (cls >> 1)) * (1 + (cls & 1))) * CoutPad * CsP;
    const int kk = k * k;
    const size_t total = (size_t)Cout * Cs * kk;
    const int py = cls >> 1, px = cls & 1;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tap = e % kk;
        const int cl = (e / kk) % Cs;
        const int n = e / ((size_t)kk * Cs);
        const int r = tap / k, s = tap % k;
        int tapd;
        if (cls < 0) {
            tapd = (k - 1 - r) * k + (k - 1 - s);
        } else {
            if ((py == 0) != (r == 1) || (px == 0) != (s == 1)) continue;
            const int dr = py ? (2 - r) / 2 : 0, ds = px ? (2 - s) / 2 : 0;
            tapd = dr * (1 + px) + ds;
        }
        const float rem = w[(((size_t)n * CinTotal + c_off + cl) * k + r) * k + s] * wscale;
        store_pieces(rem, dst + (((size_t)tapd * (CoutPad >> 3) + (n >> 3)) * CsP + cl) * 8 + (n & 7), plane, nsplit);
    }
}
hipError_t launch_pack_conv_w_dgrad_bf16(const float *w, int Cout, int CinTotal, int k, int c_off, int Cs, int CsP, int CoutPad,
                                         int cls, int nsplit, void *dst, hipStream_t st, const unsigned *amax) {
    if (nsplit == 2 && !amax) return hipErrorInvalidValue;
    const size_t total = (size_t)Cout * Cs * k * k;
    size_t gsz = (total + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(pack_conv_w_dgrad_bf16_kernel, dim3((unsigned)gsz), dim3(256), 0, st, w, Cout, CinTotal, k, c_off, Cs,
                       CsP, CoutPad, cls, nsplit, static_cast<unsigned short *>(dst), amax);
    return hipGetLastError();
}

// ---- dispatch
template <int KS, int S, int WM, int WN, int WTM, int WTN, int SPL, bool BM = false, bool LZ = false>
static hipError_t launch_b16_one(ConvArgs a, hipStream_t st, ConvArgs *resolved) {
    using Cfg = ConvCfgB16<KS, S, WM, WN, WTM, WTN, SPL>;
    if constexpr (!LZ) {
        bool lazy = false;
        for (int i = 0; i < a.nsrc; ++i) lazy |= a.src[i].la != nullptr;
        if (lazy) {      // lazy sources: forward launches of the fp16-split mode (3x3 stride 1 / 2 and 1x1), never a data gradient
            if constexpr (SPL == 2 && !BM && (KS == 3 || KS == 1)) {
                if (a.bm_y) return hipErrorInvalidValue;
                return launch_b16_one<KS, S, WM, WN, WTM, WTN, SPL, false, true>(a, st, resolved);
            } else {
                return hipErrorInvalidValue;
            }
        }
    }
    if constexpr (!BM && S == 1 && (KS == 3 || KS == 1)) {      // backward-statistics epilogue: own instantiation
        if (a.bm_y) return launch_b16_one<KS, S, WM, WN, WTM, WTN, SPL, true>(a, st, resolved);
    } else if constexpr (!BM) {
        if (a.bm_y) return hipErrorInvalidValue;
    }
    if (Cfg::LDS_BYTES > 160 * 1024) return hipErrorInvalidValue;
    a.ppr = (a.Wout + 7) / 8;
    a.ppi = a.ppr * ((a.Hout + 3) / 4);
    a.chunks = (a.ppi + Cfg::PB - 1) / Cfg::PB;
    if (a.CoutP % Cfg::BNT) return hipErrorInvalidValue;
    if (resolved) *resolved = a;
    static DynLdsOnce attr_set;
    auto kern = conv_bf16_kernel<KS, S, WM, WN, WTM, WTN, SPL, BM, LZ>;
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(kern), (int)(Cfg::LDS_BYTES));
        if (e != hipSuccess) return e;
    }
    const int ntiles = a.CoutP / Cfg::BNT;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.chunks * ntiles)), dim3(Cfg::NT), Cfg::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <int KS, int S, int SPL>
static hipError_t launch_b16_shape(const ConvArgs &a, hipStream_t st, ConvArgs *resolved) {
    switch (a.cfg & 15) {
        case CFG_128x128: return launch_b16_one<KS, S, 2, 2, 2, 2, SPL>(a, st, resolved);
        case CFG_128x64: return launch_b16_one<KS, S, 2, 2, 2, 1, SPL>(a, st, resolved);
        case CFG_128x64m: return launch_b16_one<KS, S, 4, 1, 1, 2, SPL>(a, st, resolved);
        case CFG_128x32: return launch_b16_one<KS, S, 4, 1, 1, 1, SPL>(a, st, resolved);
        case CFG_64x128: return launch_b16_one<KS, S, 1, 4, 2, 1, SPL>(a, st, resolved);
        case CFG_64x64: return launch_b16_one<KS, S, 2, 2, 1, 1, SPL>(a, st, resolved);
        default: return hipErrorInvalidValue;
    }
}

bool conv_bf16_ok(const ConvArgs &a, int ks, int stride) {
    if (!a.wpk16) return false;
    if (a.prec == 3) {       // the fp16 split needs the maxima of every operand tensor
        if (!a.amax_w) return false;
        for (int i = 0; i < a.nsrc; ++i)
            if (!a.amax_in[i]) return false;
    }
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].C % 32) return false;
    if (ks == 3) return stride == 1 || stride == 2;
    return stride == 1 && (ks == 1 || ks == 12 || ks == 21 || ks == 22);
}

template <int SPL>
static hipError_t launch_conv_b16_spl(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved) {
    if (ks == 3 && stride == 1) return launch_b16_shape<3, 1, SPL>(a, st, resolved);
    if (ks == 3 && stride == 2) return launch_b16_shape<3, 2, SPL>(a, st, resolved);
    if (ks == 1) return launch_b16_shape<1, 1, SPL>(a, st, resolved);
    if (ks == 12) return launch_b16_shape<12, 1, SPL>(a, st, resolved);
    if (ks == 21) return launch_b16_shape<21, 1, SPL>(a, st, resolved);
    return launch_b16_shape<22, 1, SPL>(a, st, resolved);
}

hipError_t launch_conv_bf16(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved) {
    if (!conv_bf16_ok(a, ks, stride)) return hipErrorInvalidValue;
    if (a.prec == 3) return launch_conv_b16_spl<2>(a, ks, stride, st, resolved);
    return a.prec == 2 ? launch_conv_b16_spl<3>(a, ks, stride, st, resolved) : launch_conv_b16_spl<1>(a, ks, stride, st, resolved);
}

}  // namespace mc
