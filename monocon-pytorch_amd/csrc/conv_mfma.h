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
//     channel) sum / sum-of-squares partials (for AttnBN instance statistics / train-mode BN);
//     output (and residual) may be a strided scatter (o_px / o_row), used by the stride-2 data gradient.
//   * every global access is a buffer instruction: descriptor in SGPRs, lane offset resolved once,
//     wave-uniform SGPR offset per K-chunk / tap -- no vector address arithmetic in the K loop.
// Family: conv_mfma_kernel (this tiling), conv_mfma_ws_kernel (producer wave + double-buffered LDS),
// conv_small_kernel (conv_small.hip: 16/32-channel layers on 16x16x4, no LDS), conv_bf16_kernel
// (conv_bf16.hip: bf16 operands, or fp32 emulated by a 3-way bf16 split).  launch_conv() dispatches on
// ConvArgs::cfg / ::prec; results of the fp32 variants are bit-identical across workgroup shapes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

namespace mc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (function, DEVICE): a process-wide "already set" flag would
// launch a second handle's kernels on another GPU without it.  One bit per device in a per-instantiation mask;
// racing threads at worst set the attribute twice.
struct DynLdsOnce {
    std::atomic<unsigned long long> done{0};
    hipError_t ensure(const void *fn, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done.fetch_or(1ull << dev, std::memory_order_release);
        return e;
    }
};

struct ConvSrc {
    const float *p;
    int C;
    // "Lazy" activation (train plans of precision mode 3, round 6): p holds the RAW conv output y of the producing layer and
    // the tensor's value is max(fma(y, la[c], lb[c]), 0) -- train-mode BatchNorm apply + ReLU, formed by the CONSUMER
    // while it stages the operand (one fma + one v_med3 on top of the scale-and-split it pays anyway), so the activation is
    // never written to HBM.  Zero padding stays zero (not relu(lb)): lazy_act()'s `cap`.  null: p holds the values.
    // Scaling by a power of two commutes with the fma's rounding, so the staged pieces are bit-identical to those of the
    // stored activation under the same operand scale (the scale itself comes from a sound BOUND of max |z| in this case,
    // written by bn_finalize_kernel; the stored activation's slot holds its exact maximum).
    const float *la, *lb;
};
// one lazy element: a = la[c] * 2^e, b = lb[c] * 2^e (the operand scale folded into the coefficients), cap = +inf inside the
// image, 0 on padding.  v_med3_f32(t, 0, cap) = min(max(t, 0), cap) is the ReLU and the padding mask in one instruction.
__device__ __forceinline__ float lazy_act(float y, float a, float b, float cap) {
    return __builtin_amdgcn_fmed3f(__builtin_fmaf(y, a, b), 0.f, cap);
}

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
    float *stats;            // optional [B][patches per image][CoutP][2] partial (sum, sumsq) of (v - shift), per 4x8 patch
    const float *stat_shift; // [Cout] or null
    int ppr, ppi, chunks;    // patches per row / per image, workgroup chunks per image
    int cfg;                 // ConvCfgId workgroup shape (CFG_AUTO = conv_pick_cfg)
    // output / residual pixel mapping in floats (0 = dense NHWC: img = Hout*Wout*ld, row = Wout*ld, px = ld);
    // the stride-2 data-gradient classes scatter to every second pixel of a larger map
    int o_img, o_row, o_px, r_img, r_row, r_px;
    // conv_bf16.hip: prec 1 = bf16 MFMA operands from the bf16 panel wpk16 (fp32 everything else); prec 2 = fp32
    // emulated by a 3-way bf16 split of both operands (six bf16 MFMAs per product, three panels in wpk16)
    const void *wpk16;       // [pieces][k*k][Cin/8][CoutP][8] bf16 or null
    int prec;
    // backward-statistics mode, for a data gradient whose output completes the gradient of a BatchNorm'd map (dense
    // NHWC, Cout channels): the epilogue applies that map's ReLU mask to the total (accumulated) gradient d, stores
    // the masked d, and writes the (sum d, sum d*y) partials of the BatchNorm backward to `stats` (per 4x8 patch) --
    // the separate three-tensor reduction pass over the activation disappears
    const float *bm_y;       // pre-BN conv output y of the forward, or null (mode off)
    const float *bm_z;       // post-BN map z (mask z > 0), bm_relu == 1
    // round 6: the same mask BIT-PACKED, [pixel][Cout / 32] words (bit c % 32 of word c / 32 = z[pixel][c] > 0), written by the
    // forward's affine_act pass of a residual layer (the only layers whose mask cannot be recomputed from y alone): the
    // epilogue then reads one broadcast word per row instead of one float per lane -- 1/32 of the bytes (252 MB -> 8 MB for a
    // 64-channel 96x320 map at B = 32).  Optional; with it bm_z is not read.  Cout % 32 == 0.
    const unsigned *bm_zbits;
    const float *bm_a, *bm_b;// forward BN coefficients (mask fma(y, a, b) > 0, bit-identical to z > 0), bm_relu == 2
    int bm_relu;             // 0: no ReLU (no mask), 1, 2
    // prec 3 (conv_bf16.hip, fp32 emulated by a 2-way fp16 split): both operands are scaled by a power of two derived
    // from their tensor's max |x| so that they sit in fp16's range; the slots hold the BIT PATTERN of that maximum
    // (a non-negative float orders like its bits), written by whoever produced the tensor (amax_update_* below; a
    // tensor's slot is AMAX_SUB sub-slots, a weight's slot one word)
    const unsigned *amax_in[4];   // per source of the virtual concat
    const unsigned *amax_w;       // of the master weight(s) the panel was cut from
    unsigned *amax_out;           // optional: max |out| of this launch is folded in (eval plans: the consumer's amax_in)
    unsigned long long *phase_prof;   // measurement aid (-DMC_PHASE_TIMERS, mc_bench_conv): per workgroup and wave, cycles spent in
                                      // [0] staging, [1] barriers, [2] MFMA phases, [3] epilogue, [4] total
};

// ---- power-of-two operand scaling of the fp16-split mode ---------------------------------------------------------
// scale exponent of a tensor whose max |x| has the bit pattern `amax_bits`: max |x| * 2^e lands in [2^14, 2^15) (fp16
// overflows at 65504).  An all-zero (or denormal-max) tensor keeps e = 0; e stays within what exp2i() can represent.
// The output rescale is formed as the PRODUCT 2^-e_a * 2^-e_w, which leaves the float range only where the true
// result does.
__host__ __device__ __forceinline__ int f16_scale_exp(unsigned amax_bits) {
    const int E = (int)((amax_bits >> 23) & 0xffu);
    if (E == 0) return 0;
    const int e = 141 - E;
    return e > 126 ? 126 : e;
}
__host__ __device__ __forceinline__ float exp2i(int e) {     // 2^e, -126 <= e <= 127
    const unsigned u = (unsigned)(127 + e) << 23;
    return __builtin_bit_cast(float, u);
}
// ---- the max-|x| slots ------------------------------------------------------------------------------------------
// A tensor's slot is AMAX_SUB words, AMAX_STRIDE words (256 bytes) apart; its maximum is the largest of them.  Every
// workgroup of a producing launch folds its own maximum into sub-slot (blockIdx.x % AMAX_SUB), and only when that can
// still raise it (a relaxed read first; a stale value merely costs a redundant atomic).  Why not one word and one
// atomic per lane: measured (round 3, rocprofv3) -- ~32 k same-address requests per launch serialise in one L2 channel
// at ~1.5 ns each: +55 us on a 55 us element-wise pass.  One request per workgroup, spread over 16 channels, is noise.
// Inf / NaN never enter a slot (they propagate through the data instead).  Weight slots are single words (amax_w).
constexpr int AMAX_SUB = 16, AMAX_STRIDE = 64, AMAX_WORDS = AMAX_SUB * AMAX_STRIDE;
__device__ __forceinline__ void amax_commit(unsigned *slot, unsigned bits) {       // call from ONE lane per workgroup / wave
    unsigned *p = slot + (blockIdx.x % AMAX_SUB) * AMAX_STRIDE;
    if (bits < 0x7f800000u && bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
}
// every thread of the workgroup calls this (partially filled waves are fine: the reduction goes through LDS atomics)
__device__ __forceinline__ void amax_update_block(unsigned *slot, float v) {
    __shared__ unsigned s_amax;
    if (threadIdx.x == 0) s_amax = 0u;
    __syncthreads();
    const unsigned bits = __builtin_bit_cast(unsigned, v);
    if (bits < 0x7f800000u && bits != 0u) atomicMax(&s_amax, bits);
    __syncthreads();
    if (threadIdx.x == 0) amax_commit(slot, s_amax);
}
// full 64-lane waves only (the MFMA kernels' epilogues): one commit per wave
__device__ __forceinline__ void amax_update_wave(unsigned *slot, float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) amax_commit(slot, __builtin_bit_cast(unsigned, v));
}
// the tensor's maximum (bit pattern), wave-uniform; full 64-lane waves only
__device__ __forceinline__ unsigned amax_read(const unsigned *slot) {
    const int lane = threadIdx.x & 63;
    unsigned v = lane < AMAX_SUB ? slot[lane * AMAX_STRIDE] : 0u;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)v, o); v = t > v ? t : v; }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// Filter window code KS: 3 = 3x3 (pad 1), 1 = 1x1; 12 / 21 / 22 = 1x2, 2x1, 2x2 windows without padding --
// the four output-parity classes of a stride-2 3x3 data gradient (dX[2i+py][2j+px] only sees the taps
// of matching parity, see mc_train_plan.hip emit_dgrad), launched with a strided output mapping.
constexpr int win_h(int ks) { return ks >= 10 ? ks / 10 : ks; }
constexpr int win_w(int ks) { return ks >= 10 ? ks % 10 : ks; }
constexpr int win_pad(int ks) { return ks == 3 ? 1 : 0; }

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
struct ConvCfg {
    static constexpr int PB = WM * WTM;
    static constexpr int BNT = WN * WTN * 32;
    static constexpr int NT = 64 * WM * WN;
    static constexpr int KH = win_h(KS), KW = win_w(KS);
    static constexpr int PAD = win_pad(KS);
    static constexpr int IH = 3 * S + KH;
    static constexpr int IW = 7 * S + KW;
    static constexpr int NPIX = IH * IW;
    static constexpr int CKP = CK + 4;
    static constexpr int LDS_FLOATS = PB * NPIX * CKP + PB * 4;
    static constexpr size_t LDS_BYTES = sizeof(float) * LDS_FLOATS;
};

// Buffer-addressed memory access: every global access of the kernel goes through a buffer
// resource (SGPR descriptor) + one lane offset kept in a VGPR + a wave-uniform SGPR offset, so
// the K loop issues no vector-ALU address arithmetic at all -- on CDNA the MFMA shares the VALU
// issue port, and every v_mad/v_cndmask of ANY wave on the SIMD displaces matrix issue slots
// (measured: a staging wave next to an MFMA wave advances ~10x slower than alone).  Offsets
// >= num_records read as zero / drop the store, which also replaces the halo and edge branches.
constexpr int BUF_OOB = (int)0x80000000;
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}

// Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement": a speed
// hint, never relied on for correctness).  xcd_order() renumbers the grid so that every XCD walks ONE contiguous
// range of logical workgroups: neighbouring pixel chunks (shared halo rows) and the column tiles of a chunk (same
// input) then meet in the same 4 MB L2 instead of being fetched once per XCD.  Bijective for any grid size.
// MEASURED (A/B on one box, B=32; scratch/traffic_ab.sh): HBM read traffic per launch 458 -> 264 MB for the conv /
// data-gradient kernels (algorithmic: ~240 MB of reads), 700 -> 373 MB for the weight gradients, 1421 -> 1080 MB for
// the 16-channel row kernel -- i.e. the halo and column-tile re-reads that used to go out to the fabric once per
// XCD now hit in L2.  These kernels are MFMA-bound, so the time barely moves (train step 105.3-106.0 vs 105.6-105.9
// ms, eval forward B=32 +1 %, B=8 -1 %); the remap is kept for the traffic (-DMC_XCD_ORDER=0 compiles it out).
#ifndef MC_XCD_ORDER
#define MC_XCD_ORDER 1
#endif
__device__ __forceinline__ int xcd_order(int b, int n) {
    if (!MC_XCD_ORDER) return b;
    const int xcd = b & 7, idx = b >> 3, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Epilogue shared by both kernel variants.  C/D layout of v_mfma_f32_32x32x2: column = lane&31,
// row m = (r&3) + 8*(r>>2) + 4*(lane>>5); row m of a 4x8 patch is pixel (oy0 + (m>>3), ox0 + (m&7))
// = (oy0 + (r>>2), ox0 + (r&3) + 4*(lane>>5)).
// per-column coefficients of the epilogue.  Loaded BEFORE the K loop (conv_epi_coef): they are plain global loads, and
// fetched at the top of the epilogue each workgroup paid a full memory round trip for them with nothing to overlap
// (measured with the phase timers of mc_bench_conv, round 3: the epilogue took half as long as the workgroup's own MFMAs).
template <int WTN>
struct EpiCoef {
    float sc[WTN], bi[WTN], sh[WTN], ma[WTN], mb[WTN];
};
template <int WN, int WTN, bool BM>
__device__ __forceinline__ EpiCoef<WTN> conv_epi_coef(const ConvArgs &a, int n0, int wn, int li) {
    EpiCoef<WTN> c;
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) {
        const int n = n0 + (wn * WTN + tn) * 32 + li;
        const bool nok = n < a.Cout;
        c.sc[tn] = (a.scale && nok) ? a.scale[n] : 1.f;
        c.bi[tn] = (a.bias && nok) ? a.bias[n] : 0.f;
        c.sh[tn] = (a.stat_shift && nok) ? a.stat_shift[n] : 0.f;
        c.ma[tn] = (BM && a.bm_relu == 2 && nok) ? a.bm_a[n] : 0.f;
        c.mb[tn] = (BM && a.bm_relu == 2 && nok) ? a.bm_b[n] : 0.f;
    }
    return c;
}

template <int WM, int WN, int WTM, int WTN, int BNT, bool BM = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, f32x16 (&acc)[WTM][WTN], const int *pinfo,
                                              int patch0, int img, int n0, int wm, int wn, int g, int li,
                                              const EpiCoef<WTN> &coef, float omul = 1.f, float *vmax_acc = nullptr) {
    // Every multiply-add below is written as the fma it is meant to be, and nothing else may be contracted: with
    // -ffp-contract=fast the choice is the optimiser's, PER INSTANTIATION -- and the kernels that share this epilogue
    // promise bit-identical results (round 5: one copy of `v = acc * sc + bi; v += res` came out differently for one r).
#pragma clang fp contract(off)
    // omul: power-of-two rescale of the accumulator (fp16-split mode: undoes the operand scaling, exact); 1 otherwise
    // vmax_acc: a persistent caller (conv_wres.hip) collects this lane's max |stored value| over its calls here and commits
    // ConvArgs::amax_out once itself (amax_commit waits for a returned load: once per workgroup, not once per tile)
    const bool do_stats = a.stats != nullptr;
    float vmax = 0.f;                            // max |stored value| of this lane (ConvArgs::amax_out)
    const bool has_res = a.res != nullptr;
    constexpr bool bm = BM;                      // backward-statistics mode (see ConvArgs): its own instantiation
    const int bm_relu = a.bm_relu;
    const float floor_v = a.relu ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out + (size_t)img * a.o_img, (unsigned)a.o_img * 4u);
    const __amdgpu_buffer_rsrc_t r_res =
        make_rsrc(has_res ? a.res + (size_t)img * a.r_img : a.out, has_res ? (unsigned)a.r_img * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_y = make_rsrc(bm ? a.bm_y + (size_t)img * a.o_img : a.out, bm ? (unsigned)a.o_img * 4u : 0u);
    // the stored mask: floats (dense layout of the output) or, when bm_zbits is given, one word per pixel and 32 channels
    const bool zbits = bm && bm_relu == 1 && a.bm_zbits != nullptr;
    const int zwords = a.Cout >> 5;                                     // words per pixel of the bit-packed mask
    const unsigned zimg = zbits ? (unsigned)(a.Hout * a.Wout * zwords) * 4u : (unsigned)a.o_img * 4u;      // bytes per image
    const __amdgpu_buffer_rsrc_t r_z =
        make_rsrc(bm && bm_relu == 1 ? (zbits ? reinterpret_cast<const float *>(a.bm_zbits) + (size_t)img * a.Hout * a.Wout * zwords
                                              : a.bm_z + (size_t)img * a.o_img)
                                     : a.out,
                  bm && bm_relu == 1 ? zimg : 0u);
    // statistics partials are per 4x8 PATCH and channel: stats[b][patch][CoutP][2].  A patch's 32 values are
    // summed in an order fixed by the MFMA layout, so the partials -- and with them train-mode BN -- do not depend
    // on the workgroup shape the autotuner picked.
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) {
        const int n = n0 + (wn * WTN + tn) * 32 + li;
        const bool nok = n < a.Cout;
        const float sc = coef.sc[tn] * omul, bi = coef.bi[tn], sh = coef.sh[tn], ma = coef.ma[tn], mb = coef.mb[tn];
        const int v_out = nok ? (4 * g * a.o_px + a.out_coff + n) * 4 : BUF_OOB;
        const int v_res = nok ? (4 * g * a.r_px + n) * 4 : BUF_OOB;
        const int v_bm = nok ? (4 * g * a.o_px + n) * 4 : BUF_OOB;      // y / z share the dense layout of the output
        const int v_zb = nok ? (4 * g * zwords + (n >> 5)) * 4 : BUF_OOB;   // bit-packed mask: word n / 32 of pixel (ox0 + 4g + ..)
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm) {
            const int p = wm * WTM + tm;
            const int oy0 = __builtin_amdgcn_readfirstlane(pinfo[p * 4 + 1]);
            const int ox0 = __builtin_amdgcn_readfirstlane(pinfo[p * 4 + 2]);
            const int pv = __builtin_amdgcn_readfirstlane(pinfo[p * 4 + 3]);
            if (!pv) continue;
            float ssum = 0.f, ssq = 0.f;
            if (oy0 + 4 <= a.Hout && ox0 + 8 <= a.Wout) {
                // whole patch inside the image: wave-uniform offsets only.  One straight-line copy of the body per
                // (residual, stored-mask) combination: with the optional loads under run-time conditions in ONE body, hipcc
                // places the s_waitcnt vmcnt(N) of the loaded values behind the join -- and on the paths WITHOUT those loads
                // the same waits then count the patch's own stores (gfx950 retires stores on vmcnt too): every second store
                // waited for the write acknowledgement of an earlier one (round 5, found in the ISA of conv_wres_kernel).
                const int s_out = (oy0 * a.o_row + ox0 * a.o_px) * 4;
                const int s_res = (oy0 * a.r_row + ox0 * a.r_px) * 4;
                auto body = [&](auto res_c, auto zmask_c) {
                    constexpr bool RES = decltype(res_c)::value, ZMASK = decltype(zmask_c)::value;
                    float rv[16], yv[16], zv[16];
                    if constexpr (RES) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            rv[r] = buf_load1(r_res, v_res, s_res + ((r >> 2) * a.r_row + (r & 3) * a.r_px) * 4);
                    }
                    if constexpr (bm) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            yv[r] = buf_load1(r_y, v_bm, s_out + ((r >> 2) * a.o_row + (r & 3) * a.o_px) * 4);
                        if constexpr (ZMASK) {
                            if (zbits) {       // (wave-uniform) one word per row, the same for the 32 lanes of a half wave
                                const int s_zb = (oy0 * a.Wout + ox0) * zwords * 4;
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const unsigned w = __builtin_bit_cast(unsigned, buf_load1(r_z, v_zb, s_zb + ((r >> 2) * a.Wout + (r & 3)) * zwords * 4));
                                    zv[r] = ((w >> (n & 31)) & 1u) ? 1.f : 0.f;
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    zv[r] = buf_load1(r_z, v_bm, s_out + ((r >> 2) * a.o_row + (r & 3) * a.o_px) * 4);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = __builtin_fmaf(acc[tm][tn][r], sc, bi);
                        if constexpr (RES) v += rv[r];
                        if constexpr (bm) {
                            const bool on = ZMASK ? zv[r] > 0.f : (bm_relu == 0 || fmaf(yv[r], ma, mb) > 0.f);
                            v = on ? v : 0.f;
                            ssum += v;
                            ssq = fmaf(v, yv[r], ssq);
                        } else if (do_stats) {
                            const float d = v - sh;
                            ssum += d;
                            ssq += d * d;      // (two roundings: contraction is off in this function)
                        }
                        v = fmaxf(v, floor_v);
                        vmax = fmaxf(vmax, fabsf(v));
                        buf_store1(v, r_out, v_out, s_out + ((r >> 2) * a.o_row + (r & 3) * a.o_px) * 4);
                    }
                };
                using T = std::true_type;
                using Fz = std::false_type;
                const bool zmask = bm && bm_relu == 1;
                if (has_res) { if (zmask) body(T{}, T{}); else body(T{}, Fz{}); }
                else { if (zmask) body(Fz{}, T{}); else body(Fz{}, Fz{}); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int y = oy0 + (r >> 2), x = ox0 + (r & 3) + 4 * g;
                    if (nok && y < a.Hout && x < a.Wout) {
                        float v = __builtin_fmaf(acc[tm][tn][r], sc, bi);
                        if (has_res) v += buf_load1(r_res, (y * a.r_row + x * a.r_px + n) * 4, 0);
                        if (bm) {
                            const float yy = buf_load1(r_y, (y * a.o_row + x * a.o_px + n) * 4, 0);
                            bool on = true;
                            if (bm_relu == 1)
                                on = zbits ? ((__builtin_bit_cast(unsigned, buf_load1(r_z, ((y * a.Wout + x) * zwords + (n >> 5)) * 4, 0)) >> (n & 31)) & 1u) != 0u
                                           : buf_load1(r_z, (y * a.o_row + x * a.o_px + n) * 4, 0) > 0.f;
                            else if (bm_relu == 2) on = fmaf(yy, ma, mb) > 0.f;
                            v = on ? v : 0.f;
                            ssum += v;
                            ssq = fmaf(v, yy, ssq);
                        } else {
                            const float d = v - sh;
                            ssum += d;
                            ssq += d * d;      // (two roundings: contraction is off in this function)
                        }
                        v = fmaxf(v, floor_v);
                        vmax = fmaxf(vmax, fabsf(v));
                        buf_store1(v, r_out, (y * a.o_row + x * a.o_px + a.out_coff + n) * 4, 0);
                    }
                }
            }
            if (do_stats) {
                ssum += __shfl_xor(ssum, 32);
                ssq += __shfl_xor(ssq, 32);
                if (g == 0 && nok) {
                    float *dst = a.stats + (((size_t)img * a.ppi + patch0 + p) * a.CoutP + n) * 2;
                    dst[0] = ssum;
                    dst[1] = ssq;
                }
            }
        }
    }
    if (vmax_acc) *vmax_acc = fmaxf(*vmax_acc, vmax);
    else if (a.amax_out) amax_update_wave(a.amax_out, vmax);
}

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN, bool BM = false>
__global__ __launch_bounds__(64 * WM * WN, 3) void conv_mfma_kernel(const ConvArgs a) {
    using Cfg = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    constexpr int PB = Cfg::PB, BNT = Cfg::BNT, NT = Cfg::NT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CKP = Cfg::CKP;
    constexpr int C4 = CK / 4;
    static_assert(NT % C4 == 0, "a thread keeps one channel group across its staging elements");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    int *pinfo = reinterpret_cast<int *>(lds + PB * NPIX * CKP);   // [PB][4] = b, oy0, ox0, valid

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

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- staging plan: element e = tid + NT*i of the [PB][NPIX][C4] halo tile.  Its input pixel
    //      does not depend on the K-chunk, so the lane offsets are resolved once per source; inside
    //      the K loop a chunk is NIT buffer loads (SGPR chunk offset) + NIT ds_write_b128 at
    //      immediate offsets.
    constexpr int TOTAL = PB * NPIX * C4;
    constexpr int NIT = (TOTAL + NT - 1) / NT;
    const int c4 = tid % C4;
    float *stage_dst = lds + (tid / C4) * CKP + c4 * 4;

    // A-fragment base offsets (floats) inside the LDS image for this lane
    int a_off[WTM];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
        a_off[tm] = ((wm * WTM + tm) * NPIX + ((li >> 3) * S) * IW + (li & 7) * S) * CKP + 4 * g;

    const int Cin4 = a.Cin >> 2;
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpk, (unsigned)(Cfg::KH * Cfg::KW * a.Cin * a.CoutP) * 4u);
    const int w_lane = (g * a.CoutP + n0 + wn * WTN * 32 + li) * 16;   // bytes

    // B fragment of step s (= tap * CK/8 + k8) of the K-chunk starting at concat channel kc
    constexpr int K8 = CK / 8, NS = Cfg::KH * Cfg::KW * K8;
    auto load_b = [&](f32x4(&dst)[WTN], int kc, int s) {
        const int tap = s / K8, k8 = s % K8;
        const int soff = (tap * Cin4 + ((kc + k8 * 8) >> 2)) * a.CoutP * 16;
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn) dst[tn] = buf_load4(r_w, w_lane + tn * 32 * 16, soff);
    };
    auto load_a = [&](f32x4(&dst)[WTM], int s) {
        const int tap = s / K8, k8 = s % K8;
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
            dst[tm] = *reinterpret_cast<const f32x4 *>(
                &lds[a_off[tm] + ((tap / Cfg::KW) * IW + (tap % Cfg::KW)) * CKP + k8 * 8]);
    };
    f32x4 bcur[WTN];
    load_b(bcur, 0, 0);   // weights do not depend on the staged tile: in flight across the barriers
    const EpiCoef<WTN> coef = conv_epi_coef<WN, WTN, BM>(a, n0, wn, li);

    int kbase = 0;   // channel offset of the current source inside the virtual concat
    for (int si = 0; si < a.nsrc; ++si) {
        const int Cs = a.src[si].C;
        const __amdgpu_buffer_rsrc_t r_in =
            make_rsrc(a.src[si].p + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
        int voff[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int e = tid + NT * i;
            const int t = e / C4;
            const int pix = t % NPIX;
            const int p = (t / NPIX) % PB;
            const int iy = pix / IW, ix = pix % IW;
            const int y = pinfo[p * 4 + 1] * S - PAD + iy;
            const int x = pinfo[p * 4 + 2] * S - PAD + ix;
            const bool ok = e < TOTAL && pinfo[p * 4 + 3] && y >= 0 && y < a.Hin && x >= 0 && x < a.Win;
            voff[i] = ok ? ((y * a.Win + x) * Cs + c4 * 4) * 4 : BUF_OOB;
        }
        for (int c0 = 0; c0 < Cs; c0 += CK) {
            if (kbase + c0 > 0) __syncthreads();   // previous chunk's fragment reads done
            // ---- stage [PB][NPIX][CK] input halo, zero-filled outside the image
            constexpr int UB = NIT > 8 ? 8 : NIT;   // loads in flight per batch
#pragma unroll
            for (int i0 = 0; i0 < NIT; i0 += UB) {
                f32x4 v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (i0 + u < NIT) v[u] = buf_load4(r_in, voff[i0 + u], c0 * 4);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = i0 + u;
                    if (i < NIT && (NT * (i + 1) <= TOTAL || tid + NT * i < TOTAL))
                        *reinterpret_cast<f32x4 *>(stage_dst + i * (NT / C4) * CKP) = v[u];
                }
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

    conv_epilogue<WM, WN, WTM, WTN, BNT, BM>(a, acc, pinfo, chunk * PB, img, n0, wm, wn, g, li, coef);
}

// ---- wave-specialised variant -------------------------------------------------------------
// Same math, same operand layouts and the same accumulation order as conv_mfma_kernel (results are
// bit-identical), but the workgroup carries one extra PRODUCER wave that stages K-chunk i+1 into
// the second half of a double-buffered LDS tile while the WM*WN consumer waves run the MFMA steps
// of chunk i.  One barrier per chunk instead of two, and no MFMA wave ever waits on HBM.
template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
struct ConvCfgWS : ConvCfg<KS, S, CK, WM, WN, WTM, WTN> {
    using Base = ConvCfg<KS, S, CK, WM, WN, WTM, WTN>;
    static constexpr int NT = 64 * (WM * WN + 1);
    static constexpr int TILE = Base::PB * Base::NPIX * Base::CKP;
    static constexpr int LDS_FLOATS = 2 * TILE + Base::PB * 4;
    static constexpr size_t LDS_BYTES = sizeof(float) * LDS_FLOATS;
};

template <int KS, int S, int CK, int WM, int WN, int WTM, int WTN>
__global__ __launch_bounds__(64 * (WM * WN + 1), 3) void conv_mfma_ws_kernel(const ConvArgs a) {
    using Cfg = ConvCfgWS<KS, S, CK, WM, WN, WTM, WTN>;
    constexpr int PB = Cfg::PB, BNT = Cfg::BNT, PAD = Cfg::PAD;
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CKP = Cfg::CKP, TILE = Cfg::TILE;
    constexpr int C4 = CK / 4;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    int *pinfo = reinterpret_cast<int *>(lds + 2 * TILE);   // [PB][4] = b, oy0, ox0, valid

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == WM * WN;
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

    if (producer) {
        // ---- producer wave: element e = lane + 64*i of the [PB][NPIX][C4] tile (see
        //      conv_mfma_kernel: lane offsets once per source, then loads + LDS writes only)
        constexpr int TOTAL = PB * NPIX * C4;
        constexpr int NIT = (TOTAL + 63) / 64;
        constexpr int UB = NIT > 8 ? 8 : NIT;   // loads kept in flight per batch
        static_assert(64 % C4 == 0, "a lane keeps one channel group across its elements");
        const int c4 = lane % C4;
        int ci = 0;
        for (int si = 0; si < a.nsrc; ++si) {
            const int Cs = a.src[si].C;
            const __amdgpu_buffer_rsrc_t r_in =
                make_rsrc(a.src[si].p + (size_t)img * a.Hin * a.Win * Cs, (unsigned)(a.Hin * a.Win * Cs) * 4u);
            int voff[NIT];
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
                voff[i] = ok ? ((y * a.Win + x) * Cs + c4 * 4) * 4 : BUF_OOB;
            }
            for (int c0 = 0; c0 < Cs; c0 += CK, ++ci) {
                float *dst = lds + (ci & 1) * TILE + (lane / C4) * CKP + c4 * 4;
#pragma unroll
                for (int i0 = 0; i0 < NIT; i0 += UB) {
                    f32x4 v[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        if (i0 + u < NIT) v[u] = buf_load4(r_in, voff[i0 + u], c0 * 4);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int i = i0 + u;
                        if (i < NIT && (64 * (i + 1) <= TOTAL || lane + 64 * i < TOTAL))
                            *reinterpret_cast<f32x4 *>(dst + i * (64 / C4) * CKP) = v[u];
                    }
                }
                __syncthreads();   // chunk ci is published; consumers are done with chunk ci-1
            }
        }
    } else {
        f32x16 acc[WTM][WTN];
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

        int a_off[WTM];
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm)
            a_off[tm] = ((wm * WTM + tm) * NPIX + ((li >> 3) * S) * IW + (li & 7) * S) * CKP + 4 * g;

        const int Cin4 = a.Cin >> 2;
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpk, (unsigned)(Cfg::KH * Cfg::KW * a.Cin * a.CoutP) * 4u);
        const int w_lane = (g * a.CoutP + n0 + wn * WTN * 32 + li) * 16;   // bytes

        constexpr int K8 = CK / 8, NS = Cfg::KH * Cfg::KW * K8;
        auto load_b = [&](f32x4(&dst)[WTN], int kc, int s) {
            const int tap = s / K8, k8 = s % K8;
            const int soff = (tap * Cin4 + ((kc + k8 * 8) >> 2)) * a.CoutP * 16;
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn) dst[tn] = buf_load4(r_w, w_lane + tn * 32 * 16, soff);
        };
        f32x4 bcur[WTN];
        load_b(bcur, 0, 0);
        const EpiCoef<WTN> coef = conv_epi_coef<WN, WTN, false>(a, n0, wn, li);
        const int nch = a.Cin / CK;
        __syncthreads();   // chunk 0 staged
        for (int ci = 0; ci < nch; ++ci) {
            const float *tile = lds + (ci & 1) * TILE;
            auto load_a = [&](f32x4(&dst)[WTM], int s) {
                const int tap = s / K8, k8 = s % K8;
#pragma unroll
                for (int tm = 0; tm < WTM; ++tm)
                    dst[tm] = *reinterpret_cast<const f32x4 *>(
                        &tile[a_off[tm] + ((tap / Cfg::KW) * IW + (tap % Cfg::KW)) * CKP + k8 * 8]);
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
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                acur[tm][j], bcur[tn][j], acc[tm][tn], 0, 0, 0);
                if (s + 1 < NS) {
#pragma unroll
                    for (int tm = 0; tm < WTM; ++tm) acur[tm] = anext[tm];
                }
#pragma unroll
                for (int tn = 0; tn < WTN; ++tn) bcur[tn] = bnext[tn];
            }
            if (ci + 1 < nch) __syncthreads();   // chunk ci+1 staged, chunk ci released
        }
        conv_epilogue<WM, WN, WTM, WTN, BNT>(a, acc, pinfo, chunk * PB, img, n0, wm, wn, g, li, coef);
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
    CFG_256x64 = 2,     // (retired: 8-patch shapes never won and spill under the 3-waves/SIMD cap)
    CFG_256x32 = 3,
    CFG_128x64 = 4,     // 2x2 waves, 2x1 tiles : 128 px x  64 ch
    CFG_128x64m = 5,    // 4x1 waves, 1x2 tiles : 128 px x  64 ch (waves split M only)
    CFG_128x32 = 6,     // 4x1 waves, 1x1 tiles : 128 px x  32 ch
    CFG_64x128 = 7,     // 1x4 waves, 2x1 tiles :  64 px x 128 ch
    CFG_64x64 = 8,      // 2x2 waves, 1x1 tiles :  64 px x  64 ch
    CFG_COUNT = 9,
    CFG_WS = 16,        // flag: wave-specialised kernel (producer wave + double-buffered LDS)
    CFG_SMALL = 32,     // LDS-free 16x16x4 kernel for 16/32-channel 3x3 layers (conv_small.hip)
    CFG_WRES = 64       // flag: weight-resident persistent kernel (conv_wres.hip) where the launch is eligible; the shape
                        // bits name the tiling every other launch with this id takes (results are bit-identical)
};
inline ConvShape conv_shape(int cfg) {
    switch (cfg & 15) {
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
// statistics partials per image a launch with this shape writes (ConvArgs::chunks)
inline int conv_chunks_per_image(int cfg, int Hout, int Wout) {
    if (cfg == CFG_SMALL) return Hout;                    // the row kernel: one partial per output row
    return ((Wout + 7) / 8) * ((Hout + 3) / 4);           // every tiling: one partial per 4x8 patch
}
bool conv_small_ok(const ConvArgs &a, int ks, int stride);
bool conv_small_lazy_ok(const ConvArgs &a, int ks, int stride);     // conv_small_kernel<2, 16, 2, LZ>: lazy source (ConvSrc::la)
bool conv_bf16_ok(const ConvArgs &a, int ks, int stride);
hipError_t launch_conv_bf16(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved);
// nsplit: 1 = bf16, 3 = three bf16 pieces, 2 = two fp16 pieces of w * 2^e_w (amax: the weight tensor's max |w|, see ConvArgs)
hipError_t launch_pack_conv_w_bf16(const float *w, int Cout, int Cin, int k, void *dst, int CinPanel, int CoutP, int n_off,
                                   int c_off, int nsplit, hipStream_t st, const unsigned *amax = nullptr);
hipError_t launch_pack_conv_w_dgrad_bf16(const float *w, int Cout, int CinTotal, int k, int c_off, int Cs, int CsP, int CoutPad,
                                         int cls, int nsplit, void *dst, hipStream_t st, const unsigned *amax = nullptr);
// max |x| of a dense fp32 tensor folded into *slot (bit pattern; the caller zeroes the slot): for tensors that enter the
// fp16-split mode from outside the plans' own producers (op-level entry points, stage-level forwards)
// (single_word: a weight slot -- one word; otherwise a tensor slot of AMAX_WORDS words)
hipError_t launch_absmax(const float *x, size_t n, unsigned *slot, hipStream_t st, bool single_word = false);
bool conv_wres_ok(const ConvArgs &a, int ks, int stride);             // conv_wres.hip: mode 3, 3x3 stride 1, one 64-channel source
hipError_t launch_conv_wres(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved);
hipError_t launch_conv_small(const ConvArgs &a, int stride, hipStream_t st);
bool conv_thin_ok(const ConvArgs &a, int ks, int stride);            // conv_thin.hip
hipError_t launch_conv_thin(const ConvArgs &a, hipStream_t st);
// the four parity classes of a thin stride-2 data gradient (dY 32 -> 16 or 64 -> 32 channels) in one pass (conv_thin.hip)
bool dgrad_s2_thin_ok(int prec, int ks, int stride, int dyC, int srcC, int CinTotal, int c_off, const unsigned *amax_dy,
                      const unsigned *amax_w, int H, int W);
hipError_t launch_dgrad_s2_thin(const float *dy, int B, int H, int W, int dyC, const float *w_master, int CinTotal, int c_off, float *out,
                                int accumulate, const unsigned *amax_dy, const unsigned *amax_w, hipStream_t st);

hipError_t launch_conv(const ConvArgs &a, int ks, int stride, hipStream_t st, ConvArgs *resolved = nullptr);
// lazy sources (ConvSrc::la): does launch_conv() have a kernel for these arguments that forms them on load?  (The fp16-pipe
// kernels of mode 3: conv_thin16, conv_bf16<SPL = 2> for 3x3 stride 1 / 2 and 1x1, conv_wres; a launch with lazy sources
// that reaches any other kernel fails.)  Independent of ConvArgs::cfg among the tilings of conv_bf16 / conv_wres.
bool conv_lazy_capable(const ConvArgs &a, int ks, int stride);

// Measurement aid (mc_profile_train): the last launch_conv / launch_wgrad on this host thread notes
// its kernel family (1 = fused conv incl. dgrad, 2 = wgrad) and algorithmic FLOPs here.
struct ProfLast { int kind; double flops, bytes; };   // bytes: algorithmic (inputs + outputs + weights, each once)
extern thread_local ProfLast prof_last;

}  // namespace mc
