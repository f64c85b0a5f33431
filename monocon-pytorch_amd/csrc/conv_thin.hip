// 3x3 stride-1 convolution of the 16-channel full-resolution layers (DLA level0 and its data gradient) in precision
// mode 3 (fp32 emulated by the 2-way fp16 split, see conv_bf16.hip) on v_mfma_f32_16x16x32_f16.
//
// Same contract as conv_small_kernel (conv_small.hip), which it replaces in mode 3 -- launch_conv_small() routes here when
// the layer qualifies (conv_thin_ok) -- and the same output mapping (D row = pixel, D column = output channel), but
// the operands come through LDS: on the fp32 pipe the row kernel is MFMA-bound (36 v_mfma_f32_16x16x4_f32 of 32 cycles
// per 16 pixels = 0.46 ms of matrix time for 15.7 M pixels, as much as the 2 GB of HBM traffic cost); on the fp16 pipe the
// same 16 pixels take 15 MFMAs of 16 cycles, IF every input element is scaled and split into its two fp16 pieces once
// and not once per filter tap -- hence the staging:
//   * a workgroup (4 waves) owns 8 output rows of one image and walks them in strips of 64 pixels.  The 10 x 66 halo
//     tile of a strip is converted once (x * 2^e -> h = fp16(x), l = fp16(x - h)) and stored as four planes
//     [piece h / l][channels 0-7 / 8-15][row][pixel] of 16 bytes per pixel.
//   * K = 9 taps x 16 channels = 144 runs as five K-steps of 32 = two taps each (the tenth half-step is zero).  The A
//     operand of a lane -- pixel l % 16 of the tile, tap 2 * step + (l / 32), channel half (l / 16) % 2 -- is ONE
//     ds_read_b128: the 16 lanes of a pass read 16 consecutive pixels of one plane, 256 contiguous bytes, conflict-free.
//   * the filter lives in registers for the lifetime of the wave (5 steps x 2 pieces x 16 bytes per 16 output channels),
//     converted from the fp32 panel the row kernel uses -- no extra pack pass.
//   * three partial products per step (l*wh and h*wl into a minor accumulator, h*wh into the main one), summed at the
//     end and multiplied by the exact inverse of both scales; epilogue as in conv_small_kernel (scale / bias / residual /
//     statistics per output row / ReLU / max |out|).
//   * the next strip's global loads are issued before the MFMAs of the current one (register prefetch, as in the
//     weight-gradient kernels).
#include "conv_mfma.h"

#ifndef THIN_OCC
#define THIN_OCC 2
#endif
#ifndef THIN_MT_UNROLL
#define THIN_MT_UNROLL 1
#endif
namespace mc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int TR = 8, TW = 64;                 // output rows per workgroup, strip width
constexpr int IR = TR + 2, IP = TW + 2;        // halo tile
constexpr int PLANE = IR * IP * 16;            // bytes of one [row][pixel] plane
constexpr int ITEMS = IR * IP * 4;             // staged float4 items per strip (4 channel quads per pixel)
constexpr int NI = (ITEMS + 255) / 256;
}  // namespace

template <int NTN>
__global__ __launch_bounds__(256, THIN_OCC) void conv_thin16_kernel(const ConvArgs a) {
    constexpr int CIN = 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * PLANE];      // [piece][half][row][pixel] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int half = kq & 1, tsel = kq >> 1;       // channel half and which of the step's two taps this lane feeds

    const int ex = f16_scale_exp(amax_read(a.amax_in[0])), ew = f16_scale_exp(*a.amax_w);
    const float x_scale = exp2i(ex), w_scale = exp2i(ew);
    const float omul = exp2i(-ex) * exp2i(-ew);

    // ---- the filter: bw[step][piece][nt] = W[n = nt*16 + li][channels 8*half .. +7][tap 2*step + tsel] * 2^ew, split
    f16x8 bw[5][2][NTN];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + tsel;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            f16x8 h8, l8;
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (tap < 9)
                    w = *reinterpret_cast<const f32x4 *>(a.wpk + ((size_t)(tap * (CIN / 4) + half * 2 + c4) * a.CoutP + nt * 16 + li) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ws = w[j] * w_scale;
                    const _Float16 hh = (_Float16)ws;
                    h8[c4 * 4 + j] = hh;
                    l8[c4 * 4 + j] = (_Float16)(ws - (float)hh);
                }
            }
            bw[s][0][nt] = h8;
            bw[s][1][nt] = l8;
        }
    }

    float sc[NTN], bi[NTN], sh[NTN];
    bool nok[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int n = nt * 16 + li;
        nok[nt] = n < a.Cout;
        sc[nt] = (a.scale && nok[nt]) ? a.scale[n] : 1.f;
        bi[nt] = (a.bias && nok[nt]) ? a.bias[n] : 0.f;
        sh[nt] = (a.stat_shift && nok[nt]) ? a.stat_shift[n] : 0.f;
    }
    const bool do_stats = a.stats != nullptr, has_res = a.res != nullptr;
    const float floor_v = a.relu ? 0.f : -__builtin_inff();
    float vmax = 0.f;

    // ---- staging plan: item e = (row, pixel, channel quad), quad fastest (4 lanes = the 64 bytes of one pixel).
    //      s_off: byte offset of the item relative to the tile's first halo pixel; rows outside the image need no test
    //      (their offsets fall outside the image's buffer descriptor), columns do (they would wrap into the next row)
    int s_off[NI], s_dst[NI];          // (s_dst < 0: no such item)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + 256 * i;
        const int c4 = e & 3, px = (e >> 2) % IP, row = (e >> 2) / IP;
        s_off[i] = ((row * a.Win + px) * CIN + c4 * 4) * 4;
        s_dst[i] = e < ITEMS ? ((c4 >> 1) * PLANE) + (row * IP + px) * 16 + (c4 & 1) * 8 : -1;
    }

    const int tiles_per_img = (a.Hout + TR - 1) / TR;
    const int nstrips = (a.Wout + TW - 1) / TW;
    for (int tile = xcd_order(blockIdx.x, gridDim.x); tile < a.B * tiles_per_img; tile += gridDim.x) {
        const int img = tile / tiles_per_img, oy0 = (tile - img * tiles_per_img) * TR;
        const __amdgpu_buffer_rsrc_t r_x = make_rsrc(a.src[0].p + (size_t)img * a.Hin * a.Win * CIN, (unsigned)(a.Hin * a.Win * CIN) * 4u);
        __amdgpu_buffer_rsrc_t r_out[2], r_res[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oy = oy0 + 2 * wave + q;
            const bool ok = oy < a.Hout;
            const size_t row = (size_t)img * a.Hout + (ok ? oy : 0);
            r_out[q] = make_rsrc(a.out + row * a.Wout * a.out_ld, ok ? (unsigned)(a.Wout * a.out_ld) * 4u : 0u);
            r_res[q] = make_rsrc(has_res ? a.res + row * a.Wout * a.res_ld : a.out, (has_res && ok) ? (unsigned)(a.Wout * a.res_ld) * 4u : 0u);
        }
        float ssum[2][NTN], ssq[2][NTN];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) ssum[q][nt] = ssq[q][nt] = 0.f;

        f32x4 pre[NI];
        const int band_off = ((oy0 - 1) * a.Win - 1) * CIN * 4;      // (row oy0 - 1, column -1) of this image
        auto fetch = [&](int strip) {
            const int x0 = strip * TW - 1, so = band_off + strip * TW * CIN * 4;
            if (strip > 0 && x0 + IP <= a.Win) {          // interior strip: every column is inside the image
#pragma unroll
                for (int i = 0; i < NI; ++i) pre[i] = buf_load4(r_x, s_dst[i] >= 0 ? s_off[i] + so : BUF_OOB, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int px = ((s_dst[i] % PLANE) >> 4) % IP;      // (recomputed: not worth 11 registers for 2 strips of 20)
                    const bool ok = s_dst[i] >= 0 && (unsigned)(x0 + px) < (unsigned)a.Win;
                    pre[i] = buf_load4(r_x, ok ? s_off[i] + so : BUF_OOB, 0);
                }
            }
        };
        auto store = [&]() {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (256 * (i + 1) > ITEMS && s_dst[i] < 0) continue;
                f16x4 h4, l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xs = pre[i][j] * x_scale;
                    const _Float16 hh = (_Float16)xs;
                    h4[j] = hh;
                    l4[j] = (_Float16)(xs - (float)hh);
                }
                *reinterpret_cast<f16x4 *>(lds + s_dst[i]) = h4;
                *reinterpret_cast<f16x4 *>(lds + 2 * PLANE + s_dst[i]) = l4;
            }
        };

        fetch(0);
        for (int strip = 0; strip < nstrips; ++strip) {
            __syncthreads();            // the fragment reads of the previous strip are done
            store();
            __syncthreads();
            if (strip + 1 < nstrips) fetch(strip + 1);
            const int sx0 = strip * TW;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int trow = 2 * wave + q;              // output row inside the tile
#pragma unroll THIN_MT_UNROLL
                for (int mt = 0; mt < TW / 16; ++mt) {
                    const int x0 = sx0 + mt * 16;
                    if (x0 >= a.Wout) break;
                    f32x4v acc[NTN], accm[NTN];
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) { acc[nt] = f32x4v{0.f, 0.f, 0.f, 0.f}; accm[nt] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int s = 0; s < 5; ++s) {
                        const int tap = 2 * s + tsel;       // (lane-dependent only through tsel)
                        const int r = tap / 3, c = tap - 3 * r;
                        f16x8 ah = {0, 0, 0, 0, 0, 0, 0, 0}, al = ah;
                        if (tap < 9) {
                            const unsigned char *p = lds + half * PLANE + ((trow + r) * IP + mt * 16 + li + c) * 16;
                            ah = *reinterpret_cast<const f16x8 *>(p);
                            al = *reinterpret_cast<const f16x8 *>(p + 2 * PLANE);
                        }
#pragma unroll
                        for (int nt = 0; nt < NTN; ++nt) {
                            accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bw[s][0][nt], accm[nt], 0, 0, 0);
                            accm[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[s][1][nt], accm[nt], 0, 0, 0);
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[s][0][nt], acc[nt], 0, 0, 0);
                        }
                    }
                    // D layout: column (n) = lane & 15, row (pixel) = 4 * (lane >> 4) + e
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) {
                        const int n = nt * 16 + li;
                        const int v_out = nok[nt] ? ((4 * kq) * a.out_ld + a.out_coff + n) * 4 : BUF_OOB;
                        const int v_res = nok[nt] ? ((4 * kq) * a.res_ld + n) * 4 : BUF_OOB;
                        float rv[4];
                        if (has_res) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) rv[e] = buf_load1(r_res[q], v_res, (x0 + e) * a.res_ld * 4);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = ((accm[nt][e] + acc[nt][e]) * omul) * sc[nt] + bi[nt];
                            if (has_res) v += rv[e];
                            if (do_stats) {
                                const float d = v - sh[nt];
                                ssum[q][nt] += d;
                                ssq[q][nt] += d * d;
                            }
                            v = fmaxf(v, floor_v);
                            vmax = fmaxf(vmax, (nok[nt] && oy0 + trow < a.Hout) ? fabsf(v) : 0.f);
                            buf_store1(v, r_out[q], v_out, (x0 + e) * a.out_ld * 4);
                        }
                    }
                }
            }
        }
        if (do_stats) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oy = oy0 + 2 * wave + q;
                if (oy >= a.Hout) continue;
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    float s1 = ssum[q][nt], s2 = ssq[q][nt];
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                    if (kq == 0 && nok[nt]) {
                        float *dst = a.stats + (((size_t)img * a.Hout + oy) * a.CoutP + nt * 16 + li) * 2;
                        dst[0] = s1;
                        dst[1] = s2;
                    }
                }
            }
        }
    }
    if (a.amax_out) amax_update_wave(a.amax_out, vmax);
}

// the layers the fp16-pipe kernel takes over from the fp32 row kernel (everything else stays there)
bool conv_thin_ok(const ConvArgs &a, int ks, int stride) {
#ifdef MC_NO_CONV_THIN
    return false;
#endif
    return a.prec == 3 && ks == 3 && stride == 1 && a.nsrc == 1 && a.Cin == 16 && a.Cout == 16 && a.amax_in[0] && a.amax_w &&
           a.Wout % 16 == 0 && a.Hout == a.Hin && a.Wout == a.Win && a.CoutP >= 16;
}

hipError_t launch_conv_thin(const ConvArgs &a, hipStream_t st) {
    const int tiles = a.B * ((a.Hout + TR - 1) / TR);
    hipLaunchKernelGGL((conv_thin16_kernel<1>), dim3(tiles < 4096 ? tiles : 4096), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mc
