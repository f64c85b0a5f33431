// Prototype of the round-4 convolution for the fp16-split mode ("f16x2"): BOTH operands reach LDS by DMA
// (buffer_load ... lds, 16 bytes per lane), in a software pipeline with ONE barrier per K-step.
//
// What makes that possible is the activation format "P16": an fp32-sized element stored as its two fp16 pieces of
// x * 2^e (e = the tensor's exponent, chosen by the producer), octet-planar: for every pixel and every 8 channels
// [h0..h7][l0..l7] (16 + 16 bytes).  One 16-byte run IS the A operand of one v_mfma_f32_32x32x16_f16 lane, so the
// staging is a copy -- no convert / split work on the VALU that shares its issue port with the matrix pipe, no
// staging registers -- and weights (already packed as piece planes [piece][tap][Cin/8][CoutP][8]) take the same road.
//
// Pipeline (K-step s = one filter tap of one 16-channel chunk = 3 * WTM * WTN MFMAs per wave):
//   top of step s   : issue the DMA of the weight slice of step s + 3 (ring of 3 slices), and a share of the halo
//                     tile of the NEXT chunk (two tile buffers)
//                     read the fragments of step s + 1 into the second register set
//                     MFMAs of step s
//   bottom of step s: s_waitcnt vmcnt(N) -- N = what was issued at the top of this step (plus the halo pieces of the
//                     previous step, which come after the weights in issue order): everything step s + 2 needs has
//                     landed -- lgkmcnt(0), s_barrier
// A DMA therefore has two steps (>= 600 MFMA cycles) to land, and never occupies a register.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace p16 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BUF_OOB = (int)0x80000000;

struct Src {
    const void *p;      // P16 NHWC
    int C;
};
struct Args {
    Src src[4];
    int nsrc;
    int B, H, W;        // stride 1: input size == output size
    int Cin, Cout, CoutP;
    const void *wpk16;  // [2][taps][Cin/8][CoutP][8] fp16
    float *out;         // fp32 NHWC, ld = Cout
    float *stats;       // optional [B][ppi][CoutP][2]
    float omul;         // 2^-(e_a + e_w)
    unsigned long long *prof;   // optional [grid][3]: cycles prologue / K loop / epilogue of wave 0
    int ppr, ppi, chunks;
};

constexpr int win_h(int ks) { return ks >= 10 ? ks / 10 : ks; }
constexpr int win_w(int ks) { return ks >= 10 ? ks % 10 : ks; }
constexpr int win_pad(int ks) { return ks == 3 ? 1 : 0; }

template <int KS, int WM, int WN, int WTM, int WTN>
struct Cfg {
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int PB = WM * WTM, NPAIR = (PB + 1) / 2, BNT = WN * WTN * 32;
    static constexpr int KH = win_h(KS), KW = win_w(KS), PAD = win_pad(KS), NTAP = KH * KW;
    static constexpr int IH = 3 + KH, IW = 7 + KW, RS = 24;
    static_assert(2 * IW <= RS, "two patches per 24-slot row");
    static constexpr int PLANE_SLOTS = NPAIR * IH * RS;                 // one (octet, piece) plane of the halo tile
    static constexpr int A_SLOTS = 4 * PLANE_SLOTS;                      // 2 octets x 2 pieces
    static constexpr int NAI = (A_SLOTS + 63) / 64;                      // DMA instructions per tile
    static constexpr int NA_W = (NAI + NW - 1) / NW;                     // ... per wave (the tile buffer is padded to this)
    static constexpr int A_BYTES = NA_W * NW * 1024;
    static constexpr int B_SLOTS = 4 * BNT;
    static_assert(B_SLOTS % (64 * NW) == 0, "weight slice: whole DMA instructions per wave");
    static constexpr int NB_W = B_SLOTS / (64 * NW);
    static constexpr int B_BYTES = B_SLOTS * 16;
    // halo pieces of the next chunk are spread over the first NA_STEPS steps of a chunk, AQ per step and wave
    static constexpr int AQ = (NA_W + NTAP - 1) / NTAP;
    static constexpr int NA_STEPS = (NA_W + AQ - 1) / AQ;
    static constexpr int NABUF = NTAP >= 4 ? 2 : 4;                      // few-tap windows: the tile of chunk c + 3 is issued in chunk c
    static constexpr int LDS_BYTES = NABUF * A_BYTES + 3 * B_BYTES + PB * 16;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int xcd_order(int b, int n) {
    const int xcd = b & 7, idx = b >> 3, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
template <int N> __device__ __forceinline__ void wait_vm_lgkm() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int KS, int WM, int WN, int WTM, int WTN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_p16_kernel(const Args a) {
    using C = Cfg<KS, WM, WN, WTM, WTN>;
    constexpr int NW = C::NW, PB = C::PB, BNT = C::BNT, NTAP = C::NTAP, IW = C::IW, IH = C::IH, RS = C::RS, PAD = C::PAD;
    constexpr int PLANE_B = C::PLANE_SLOTS * 16, A_BYTES = C::A_BYTES, B_BYTES = C::B_BYTES, NA_W = C::NA_W, NB_W = C::NB_W;
    constexpr int AQ = C::AQ, NA_STEPS = C::NA_STEPS, NABUF = C::NABUF;
    static_assert(NABUF == 2, "prototype: 3x3 windows (few-tap windows need the deeper tile ring)");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const abuf = lds;
    unsigned char *const bbuf = lds + NABUF * A_BYTES;
    int *pinfo = reinterpret_cast<int *>(lds + NABUF * A_BYTES + 3 * B_BYTES);   // [PB][4] = img, oy0, ox0, valid

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

    const unsigned long long tp0 = __builtin_readcyclecounter();
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

    // ---- DMA plan.  Halo tile: flat slot f = 64 * j + lane of instruction j = wave + NW * i; slot order IS the LDS order
    //      [octet o][piece][pair][halo row][24 slots]; the lane's source is pixel (iy, ix) of patch 2 * pair + (slot >= IW).
    int a_pix[NA_W], a_sub[NA_W];     // pixel index inside the image (or -1), byte offset of (octet, piece) inside the chunk
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
            ok = pinfo[p * 4 + 3] && y >= 0 && y < a.H && x >= 0 && x < a.W;
            pix = y * a.W + x;
        }
        a_pix[i] = ok ? pix : -1;
        a_sub[i] = (plane >> 1) * 32 + (plane & 1) * 16;
    }
    // weight slice: flat slot f -> (octet o, piece, column n): [o][piece][BNT] x 16 bytes
    const int w_plane = NTAP * a.Cin * a.CoutP * 2;            // bytes of one piece of the panel
    const int Cin8 = a.Cin >> 3;
    int b_voff[NB_W];
#pragma unroll
    for (int i = 0; i < NB_W; ++i) {
        const int f = 64 * (wave + NW * i) + lane;
        const int o = f / (2 * BNT), pc = (f / BNT) & 1, n = f % BNT;
        b_voff[i] = pc * w_plane + (o * a.CoutP + n0 + n) * 16;
    }
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpk16, (unsigned)(2 * w_plane));

    // ---- fragment addresses (bytes)
    int a_off[WTM], b_off[WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm) {
        const int p = wm * WTM + tm;
        a_off[tm] = (2 * g) * PLANE_B + ((p >> 1) * IH * RS + (li >> 3) * RS + (li & 7) + IW * (p & 1)) * 16;
    }
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) b_off[tn] = ((2 * g) * BNT + (wn * WTN + tn) * 32 + li) * 16;

    f32x16 acc[WTM][WTN], accm[WTM][WTN];
#pragma unroll
    for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WTN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accm[tm][tn][r] = 0.f; }

    // ---- chunk walk over the virtual concat (16 channels per chunk)
    const int nch = a.Cin >> 4;
    int si_n = 0, c0_n = 0;                                     // source / channel offset of the chunk the NEXT tile DMA is for
    int Cs_n = a.src[0].C;
    __amdgpu_buffer_rsrc_t r_in = make_rsrc(static_cast<const char *>(a.src[0].p) + (size_t)img * a.H * a.W * Cs_n * 4,
                                            (unsigned)(a.H * a.W * Cs_n) * 4u);
    int a_voff[NA_W];
    auto lane_offsets = [&](int cs) {
#pragma unroll
        for (int i = 0; i < NA_W; ++i) a_voff[i] = a_pix[i] >= 0 ? a_pix[i] * cs * 4 + a_sub[i] : BUF_OOB;
    };
    lane_offsets(Cs_n);
    auto advance_next = [&]() {        // (si_n, c0_n) -> the chunk after it; stays on the last chunk at the end (harmless re-load)
        int s2 = si_n, c2 = c0_n + 16;
        if (c2 >= Cs_n) { ++s2; c2 = 0; }
        if (s2 >= a.nsrc) return;
        if (s2 != si_n) {
            si_n = s2;
            Cs_n = a.src[s2].C;
            r_in = make_rsrc(static_cast<const char *>(a.src[s2].p) + (size_t)img * a.H * a.W * Cs_n * 4, (unsigned)(a.H * a.W * Cs_n) * 4u);
            lane_offsets(Cs_n);
        }
        c0_n = c2;
    };
    auto dma_a = [&](int i, int buf) {       // piece i of this wave's share of the tile of chunk (si_n, c0_n)
        dma16(r_in, abuf + buf * A_BYTES + (wave + NW * i) * 1024, a_voff[i], c0_n * 4);
    };
    auto dma_b = [&](int slot, int tap, int ch) {     // weight slice of (tap, chunk ch) into ring slot `slot`
        const int soff = (tap * Cin8 + 2 * ch) * a.CoutP * 16;
#pragma unroll
        for (int i = 0; i < NB_W; ++i) dma16(r_w, bbuf + slot * B_BYTES + (wave + NW * i) * 1024, b_voff[i], soff);
    };
    auto load_frags = [&](f16x8(&fa)[2][WTM], f16x8(&fb)[2][WTN], int abuf_i, int tap, int slot) {
        const int ty = tap / C::KW, tx = tap % C::KW;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int tm = 0; tm < WTM; ++tm)
                fa[q][tm] = *reinterpret_cast<const f16x8 *>(abuf + abuf_i * A_BYTES + a_off[tm] + q * PLANE_B + (ty * RS + tx) * 16);
#pragma unroll
            for (int tn = 0; tn < WTN; ++tn)
                fb[q][tn] = *reinterpret_cast<const f16x8 *>(bbuf + slot * B_BYTES + b_off[tn] + q * BNT * 16);
        }
    };

    // ---- prologue: tile of chunk 0, weight slices of steps 0..2
#pragma unroll
    for (int i = 0; i < NA_W; ++i) dma_a(i, 0);
    advance_next();
#pragma unroll
    for (int s = 0; s < 3; ++s) dma_b(s % 3, s % NTAP, s / NTAP < nch ? s / NTAP : nch - 1);
    wait_vm_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    f16x8 fa[2][2][WTM], fb[2][2][WTN];      // [register set][piece][tile]
    load_frags(fa[0], fb[0], 0, 0, 0);

    // one chunk = NTAP steps, fully unrolled; two chunks per loop trip so that the tile buffer index is static
    auto chunk_steps = [&](auto PAR, int ch) {
        constexpr int par = decltype(PAR)::value;       // tile buffer of this chunk
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int cur = (par * NTAP + t) & 1;        // register set holding step s (NTAP odd: alternates across chunks too)
            // ---- top: DMA of step s + 3's weights, then this step's share of the next chunk's tile
            {
                const int t3 = (t + 3) % NTAP, ch3 = ch + (t + 3) / NTAP;
                dma_b(t % 3, t3, ch3 < nch ? ch3 : nch - 1);
            }
            if (t < NA_STEPS) {
#pragma unroll
                for (int q = 0; q < AQ; ++q)
                    if (t * AQ + q < NA_W) dma_a(t * AQ + q, par ^ 1);
            }
            if (t == NA_STEPS - 1) advance_next();      // (after this chunk's last piece) the tile after the next one
            // ---- fragments of step s + 1
            if (t + 1 < NTAP) load_frags(fa[cur ^ 1], fb[cur ^ 1], par, t + 1, (t + 1) % 3);
            else load_frags(fa[cur ^ 1], fb[cur ^ 1], par ^ 1, 0, (t + 1) % 3);
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMAs of step s: l*h, h*l into the minor accumulator, h*h into the main one
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int tm = 0; tm < WTM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < WTN; ++tn) {
                        const int qa = pp == 0 ? 1 : 0, qb = pp == 1 ? 1 : 0;
                        if (pp < 2)
                            accm[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][qa][tm], fb[cur][qb][tn], accm[tm][tn], 0, 0, 0);
                        else
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][0][tm], fb[cur][0][tn], acc[tm][tn], 0, 0, 0);
                    }
            // ---- bottom: everything step s + 2 reads has landed (weights of this step's issue and the halo pieces of this
            //      and the previous step may stay in flight), fragment reads retired, barrier
            __builtin_amdgcn_sched_barrier(0);
            // static counts: halo pieces issued at step t
#define P16_NA_AT(T) (((T) >= 0 && (T) < NA_STEPS) ? ((((T) + 1) * AQ <= NA_W) ? AQ : (NA_W - (T) * AQ)) : 0)
            switch (t) {      // t is a compile-time constant after unrolling; the switch folds
#define P16_CASE(T) case T: wait_vm_lgkm<P16_NA_AT(T - 1) + NB_W + P16_NA_AT(T)>(); break;
                P16_CASE(0) P16_CASE(1) P16_CASE(2) P16_CASE(3) P16_CASE(4) P16_CASE(5) P16_CASE(6) P16_CASE(7) P16_CASE(8)
#undef P16_CASE
                default: wait_vm_lgkm<0>(); break;
            }
            __builtin_amdgcn_s_barrier();
        }
    };
    const unsigned long long tp1 = __builtin_readcyclecounter();
    for (int ch = 0; ch < nch; ch += 2) {
        chunk_steps(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < nch) chunk_steps(std::integral_constant<int, 1>{}, ch + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped look-ahead DMAs of the last steps
    const unsigned long long tp2 = __builtin_readcyclecounter();

    // ---- epilogue (prototype): scale, optional statistics partials, fp32 store
    const float omul = a.omul;
    const bool do_stats = a.stats != nullptr;
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out + (size_t)img * a.H * a.W * a.Cout, (unsigned)(a.H * a.W * a.Cout) * 4u);
#pragma unroll
    for (int tn = 0; tn < WTN; ++tn) {
        const int n = n0 + (wn * WTN + tn) * 32 + li;
        const bool nok = n < a.Cout;
#pragma unroll
        for (int tm = 0; tm < WTM; ++tm) {
            const int p = wm * WTM + tm;
            const int oy0 = __builtin_amdgcn_readfirstlane(pinfo[p * 4 + 1]);
            const int ox0 = __builtin_amdgcn_readfirstlane(pinfo[p * 4 + 2]);
            const int pv = __builtin_amdgcn_readfirstlane(pinfo[p * 4 + 3]);
            if (!pv) continue;
            float ssum = 0.f, ssq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int y = oy0 + (r >> 2), x = ox0 + (r & 3) + 4 * g;
                const float v = (acc[tm][tn][r] + accm[tm][tn][r]) * omul;
                const bool ok = nok && y < a.H && x < a.W;
                if (ok) { ssum += v; ssq += v * v; }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r_out, ok ? ((y * a.W + x) * a.Cout + n) * 4 : BUF_OOB, 0, 0);
            }
            if (do_stats) {
                ssum += __shfl_xor(ssum, 32);
                ssq += __shfl_xor(ssq, 32);
                if (g == 0 && nok) {
                    float *dst = a.stats + (((size_t)img * a.ppi + chunk * PB + p) * a.CoutP + n) * 2;
                    dst[0] = ssum;
                    dst[1] = ssq;
                }
            }
        }
    }
    if (a.prof && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tp3 = __builtin_readcyclecounter();
        a.prof[blockIdx.x * 3 + 0] = tp1 - tp0; a.prof[blockIdx.x * 3 + 1] = tp2 - tp1; a.prof[blockIdx.x * 3 + 2] = tp3 - tp2;
    }
}

}  // namespace p16
