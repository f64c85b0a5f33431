// The 7x7 stem (NCHW image, 3 channels -> NHWC 16 channels; reference model/backbone/dla.py:231-234) on the fp16 matrix
// pipe -- precision mode 3, fp32 emulated by the 2-way fp16 split (conv_bf16.hip).  Same contract as stem_kernel
// (kernels_misc.hip), which stays the path of the other modes: the VALU kernel spends 0.8 ms on 74 GFLOP, three times
// the HBM time of its 1.2 GB.
//
//   * K = 7 filter rows x (8 taps x 4 channels): a filter row is one K-step of v_mfma_f32_16x16x32_f16.  Channels are
//     padded 3 -> 4 and taps 7 -> 8 with zero WEIGHTS, so that the 8 operand values of a lane -- two horizontally
//     adjacent pixels x 4 channels -- are 16 contiguous bytes of the staged tile [row][pixel][4 x fp16]:
//     lane l: pixel l % 16 of the 16-pixel tile, taps 2 * (l / 16) and 2 * (l / 16) + 1.
//   * no global maxima needed.  The image scale is per WORKGROUP: the tile's own max |x| (an LDS reduction over the
//     values the threads have just loaded) picks the power of two, and the epilogue multiplies by its exact inverse --
//     any power of two gives the same products.  The filter's scale comes from the wave's own fragments, which together
//     cover all 16 x 147 weights.
//   * a workgroup = a band of 8 rows of one image, walked in strips of 64 pixels (14 x 72 staged pixels, 16 KB of LDS
//     for both pieces); the filter registers are built once per band (a first version with one 8 x 64 tile per workgroup
//     spent a third of its time on them), and the next strip's loads are issued before the MFMAs of the current one.
#include "conv_mfma.h"
#include "kernels.h"

#ifndef STEM_MT_UNROLL
#define STEM_MT_UNROLL 1
#endif
namespace mc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int SR = 8, SW = 64;                   // output tile
constexpr int SIR = SR + 6, SIP = 72;            // staged rows, pixel slots per row (64 + 6 halo + 2 padding)
constexpr int SPIX = SIR * SIP;                  // 1008 staged pixels
constexpr int SNI = (SPIX + 255) / 256;          // 4 pixels per thread
}  // namespace

__global__ __launch_bounds__(256) void stem_f16_kernel(const float *__restrict__ img, int B, int H, int W,
                                                       const float *__restrict__ wpk /*[c][r][s][16]*/,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       float *__restrict__ out, int relu, unsigned *__restrict__ amax_out,
                                                       float *__restrict__ stats /*[B][H][16][2] or null*/,
                                                       const float *__restrict__ stat_shift) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * SPIX * 8];     // [piece][row][pixel] x (4 x fp16)
    __shared__ unsigned s_max;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int tiles_y = (H + SR - 1) / SR, nstrips = (W + SW - 1) / SW;
    const int bt = xcd_order(blockIdx.x, gridDim.x);
    const int b = bt / tiles_y;
    const int ty0 = (bt % tiles_y) * SR;

    // ---- the filter: bw[r][piece] = W[n = li][c][r][s = 2*kq + p] in the order [p][c] (c = 3 and s = 7: zero)
    float wv[7][8];
    float wmax = 0.f;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int s = 2 * kq + p;
                const float w = (c < 3 && s < 7) ? wpk[((c * 7 + r) * 7 + s) * 16 + li] : 0.f;
                wv[r][p * 4 + c] = w;
                wmax = fmaxf(wmax, fabsf(w));
            }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    const int ew = f16_scale_exp(__builtin_bit_cast(unsigned, wmax));
    const float w_scale = exp2i(ew);
    f16x8 bw[7][2];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ws = wv[r][j] * w_scale;
            const _Float16 hh = (_Float16)ws;
            bw[r][0][j] = hh;
            bw[r][1][j] = (_Float16)(ws - (float)hh);
        }

    const float sc = scale[li], sf = shift[li];
    const float sh = (stats && stat_shift) ? stat_shift[li] : 0.f;
    float vmax = 0.f;
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};     // train mode: sum (v - shift), sum (v - shift)^2 of this wave's two rows

    // ---- a workgroup walks its band of 8 rows in strips of 64 pixels (the filter registers are loaded once per band);
    //      thread = staged pixel (its three channel planes); the next strip's loads are in flight during the MFMAs
    float xin[SNI][3];
    auto fetch = [&](int strip) {
        const int tx0 = strip * SW;
#pragma unroll
        for (int i = 0; i < SNI; ++i) {
            const int e = tid + 256 * i;
            const int ly = e / SIP, lx = e - ly * SIP;
            const int y = ty0 - 3 + ly, x = tx0 - 3 + lx;
            const bool ok = e < SPIX && y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
            for (int c = 0; c < 3; ++c) xin[i][c] = ok ? img[(((size_t)b * 3 + c) * H + y) * W + x] : 0.f;
        }
    };
    fetch(0);
    for (int strip = 0; strip < nstrips; ++strip) {
        const int tx0 = strip * SW;
        // max |x| of the strip's tile -> its power-of-two scale (the barrier also ends the fragment reads of the last strip)
        float xmax = 0.f;
#pragma unroll
        for (int i = 0; i < SNI; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) xmax = fmaxf(xmax, fabsf(xin[i][c]));
        if (tid == 0) s_max = 0u;
        __syncthreads();
        {
            const unsigned bits = __builtin_bit_cast(unsigned, xmax);
            if (bits != 0u && bits < 0x7f800000u) atomicMax(&s_max, bits);
        }
        __syncthreads();
        const int ex = f16_scale_exp(s_max);
        const float x_scale = exp2i(ex);
        const float omul = exp2i(-ex) * exp2i(-ew);
#pragma unroll
        for (int i = 0; i < SNI; ++i) {
            const int e = tid + 256 * i;
            if (e >= SPIX) continue;
            f16x4 h4, l4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xs = c < 3 ? xin[i][c] * x_scale : 0.f;
                const _Float16 hh = (_Float16)xs;
                h4[c] = hh;
                l4[c] = (_Float16)(xs - (float)hh);
            }
            *reinterpret_cast<f16x4 *>(lds + e * 8) = h4;
            *reinterpret_cast<f16x4 *>(lds + SPIX * 8 + e * 8) = l4;
        }
        __syncthreads();
        if (strip + 1 < nstrips) fetch(strip + 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int trow = 2 * wave + q, y = ty0 + trow;
            if (y >= H) continue;
            float *orow = out + ((size_t)b * H + y) * W * 16;
#pragma unroll STEM_MT_UNROLL
            for (int mt = 0; mt < SW / 16; ++mt) {
                const int x0 = tx0 + mt * 16;
                if (x0 >= W) break;
                f32x4v acc = {0.f, 0.f, 0.f, 0.f}, accm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    // two adjacent staged pixels = 16 bytes at an 8-byte aligned address: two 8-byte reads
                    const unsigned char *p = lds + ((trow + r) * SIP + mt * 16 + li + 2 * kq) * 8;
                    const f16x4 h0 = *reinterpret_cast<const f16x4 *>(p), h1 = *reinterpret_cast<const f16x4 *>(p + 8);
                    const f16x4 l0 = *reinterpret_cast<const f16x4 *>(p + SPIX * 8), l1 = *reinterpret_cast<const f16x4 *>(p + SPIX * 8 + 8);
                    const f16x8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const f16x8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bw[r][0], accm, 0, 0, 0);
                    accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[r][1], accm, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[r][0], acc, 0, 0, 0);
                }
                // D layout: column (n) = lane & 15, row (pixel) = 4 * (lane >> 4) + e
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int x = x0 + 4 * kq + e;
                    float v = fmaf((accm[e] + acc[e]) * omul, sc, sf);
                    if (relu) v = fmaxf(v, 0.f);
                    if (x < W) {
                        orow[(size_t)x * 16 + li] = v;
                        vmax = fmaxf(vmax, fabsf(v));
                        const float d = v - sh;
                        ssum[q] += d;
                        ssq[q] = fmaf(d, d, ssq[q]);
                    }
                }
            }
        }
    }
    if (stats) {       // one partial per (image, output row): the layout of the row kernels (conv_small.hip)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int y = ty0 + 2 * wave + q;
            float s1 = ssum[q], s2 = ssq[q];
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (kq == 0 && y < H) {
                float *dst = stats + (((size_t)b * H + y) * 16 + li) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
    }
    if (amax_out) amax_update_wave(amax_out, vmax);
}

hipError_t launch_stem_f16(const float *img, int B, int H, int W, const float *wpk, const float *scale, const float *shift,
                           float *out, hipStream_t st, int relu, unsigned *amax_out, float *stats, const float *stat_shift) {
    const int tiles = (H + SR - 1) / SR;          // one workgroup per band of 8 rows
    hipLaunchKernelGGL(stem_f16_kernel, dim3(B * tiles), dim3(256), 0, st, img, B, H, W, wpk, scale, shift, out, relu, amax_out, stats,
                       stat_shift);
    return hipGetLastError();
}

}  // namespace mc
