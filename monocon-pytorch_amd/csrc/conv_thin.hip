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
#include "train.h"

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
    // lazy source (ConvSrc::la): the tile holds the producer's raw conv output; BatchNorm apply + ReLU happen in store()
    // below, with the operand scale folded into the coefficients.  A thread's items all belong to channel quad tid & 3.
    const bool lazy = a.src[0].la != nullptr;
    f32x4 lzA = {0.f, 0.f, 0.f, 0.f}, lzB = lzA;
    if (lazy) {
        lzA = reinterpret_cast<const f32x4 *>(a.src[0].la)[tid & 3];
        lzB = reinterpret_cast<const f32x4 *>(a.src[0].lb)[tid & 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) { lzA[j] *= x_scale; lzB[j] *= x_scale; }
    }
    const unsigned x_bytes = (unsigned)(a.Hin * a.Win * CIN) * 4u;      // one image: an offset below this is inside it
    unsigned okm = 0u;          // lazy: bit i = staged item i lies inside the image (padding must stay 0, not relu(b))

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
    // backward-statistics mode (ConvArgs::bm_y, as in conv_epilogue of conv_mfma.h): this launch completes the gradient of a
    // map; mask it with that map's ReLU and leave (sum d, sum d * y) per output row instead of (sum, sum of squares)
    const bool bm = a.bm_y != nullptr;
    const int bm_relu = a.bm_relu;
    float ma[NTN], mb[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        ma[nt] = (bm && bm_relu == 2 && nok[nt]) ? a.bm_a[nt * 16 + li] : 0.f;
        mb[nt] = (bm && bm_relu == 2 && nok[nt]) ? a.bm_b[nt * 16 + li] : 0.f;
    }

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
            okm = 0u;
            if (strip > 0 && x0 + IP <= a.Win) {          // interior strip: every column is inside the image
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int vo = s_dst[i] >= 0 ? s_off[i] + so : BUF_OOB;
                    pre[i] = buf_load4(r_x, vo, 0);
                    if (lazy) okm |= ((unsigned)vo < x_bytes ? 1u : 0u) << i;      // (columns are inside: in range <=> row inside)
                }
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int px = ((s_dst[i] % PLANE) >> 4) % IP;      // (recomputed: not worth 11 registers for 2 strips of 20)
                    const bool ok = s_dst[i] >= 0 && (unsigned)(x0 + px) < (unsigned)a.Win;
                    const int vo = ok ? s_off[i] + so : BUF_OOB;
                    pre[i] = buf_load4(r_x, vo, 0);
                    if (lazy) okm |= ((unsigned)vo < x_bytes ? 1u : 0u) << i;
                }
            }
        };
        auto store = [&]() {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (256 * (i + 1) > ITEMS && s_dst[i] < 0) continue;
                f16x4 h4, l4;
                const float cap = ((okm >> i) & 1u) ? __builtin_inff() : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xs = lazy ? lazy_act(pre[i][j], lzA[j], lzB[j], cap) : pre[i][j] * x_scale;
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
                    // backward-statistics mode: the y (and z) values of this tile's 64 outputs, requested BEFORE the MFMAs
                    // (loaded in the epilogue they cost one memory latency per tile: +1 ms per launch, measured)
                    const bool rok = oy0 + trow < a.Hout;
                    float ybm[NTN][4], zbm[NTN][4];
                    if (bm) {
#pragma unroll
                        for (int nt = 0; nt < NTN; ++nt) {
                            const size_t at0 = (((size_t)img * a.Hout + (rok ? oy0 + trow : 0)) * a.Wout + x0 + 4 * kq) * a.Cout + nt * 16 + li;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool ld = rok && nok[nt];
                                ybm[nt][e] = ld ? a.bm_y[at0 + (size_t)e * a.Cout] : 0.f;
                                zbm[nt][e] = (ld && bm_relu == 1) ? a.bm_z[at0 + (size_t)e * a.Cout] : 1.f;
                            }
                        }
                    }
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
                            if (bm) {
                                if (rok && nok[nt]) {
                                    const float yv = ybm[nt][e];
                                    const bool on = bm_relu == 0 || (bm_relu == 1 ? zbm[nt][e] > 0.f : fmaf(yv, ma[nt], mb[nt]) > 0.f);
                                    v = on ? v : 0.f;
                                    ssum[q][nt] += v;
                                    ssq[q][nt] = fmaf(v, yv, ssq[q][nt]);
                                }
                            } else if (do_stats) {
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


// ------------------------------------------------------------------------------------------------ stride-2 data gradient
// dX of the 16 -> 32 stride-2 3x3 layer (DLA level1, model/backbone/dla.py:280-298 under autograd), mode 3:
//   dX[2i + py][2j + px][c] = sum over the (1 + py)(1 + px) taps of the parity class of dY[i + dr][j + ds][0..31] . W[.][c][r][s],
//   r = 1 (py = 0) or 2 - 2 dr (py = 1), s likewise.
// The plan used to run the four parity classes as four launches of the generic tiled kernel (K = 32 per tap, 16 of its 32
// output columns padding, every launch writing every second pixel of the 1 GB map): 0.51 + 0.25 + 0.22 + 0.20 ms per step
// (rocprofv3, round 5) for 0.5 GB read and 1 GB written.  Here ONE pass: a workgroup stages a 5 x 33 pixel halo tile of dY
// (split once into its fp16 pieces, four 8-channel planes per piece), a wave owns one dY row = two output rows, a
// v_mfma_f32_16x16x32_f16 has M = 16 output pixels of one parity class, N = the 16 channels, K = the 32 channels of one tap;
// the filter (9 taps x 2 pieces) lives in registers.  Output pixels of both column parities are written by the same wave back to
// back: whole 128-byte lines.
struct DgradS2Args {
    const float *dy;            // (B, H, W, 32 * KS)
    const float *w;             // master weight OIHW (32 * KS, CinTotal, 3, 3); this launch: input channels [c_off, c_off + 16 * NT)
    float *out;                 // (B, 2H, 2W, 16 * NT)
    int B, H, W, CinTotal, c_off, accumulate;
    const unsigned *amax_dy, *amax_w;
};
namespace {
constexpr int DR = 4, DW = 32;                   // dY rows / pixels per tile strip (8 x 64 outputs)
constexpr int DIR_ = DR + 1, DIP = DW + 1;       // halo tile
constexpr int DPLANE = DIR_ * DIP * 16;          // bytes of one [row][pixel] plane of 8 channels
}  // namespace
// KS = K-steps per tap (dY channels / 32), NT = 16-channel groups of dX.  (1, 1): level1, a wave = one dY row -- the only shape
// the plan sends here.  (2, 2) -- level2's first conv, dY 64 ch at 96x320 -> dX 32 ch at 192x640, a wave = one channel group of
// two dY rows -- was built and measured (round 5): its filter (9 taps x 2 steps x 2 pieces = 144 registers) leaves one
// workgroup per CU, and the step did not move (one-session A/B 49.99 vs 49.66 ms with level1 alone); not instantiated.
template <int KS, int NT>
__global__ __launch_bounds__(256, KS == 1 ? 2 : 1) void dgrad_s2_thin_kernel(const DgradS2Args a) {
    constexpr int CY = 32 * KS, CX = 16 * NT, NOCT = 4 * KS;
    constexpr int DITEMS = DIR_ * DIP * (CY / 4), DNI = (DITEMS + 255) / 256;    // staged float4 items per strip
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * NOCT * DPLANE];       // [piece][channel octet][row][pixel] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int nt = wave % NT, row0 = (wave / NT) * NT;       // this wave's channel group and first dY row of the tile
    const int ex = f16_scale_exp(amax_read(a.amax_dy)), ew = f16_scale_exp(*a.amax_w);
    const float x_scale = exp2i(ex), w_scale = exp2i(ew), omul = exp2i(-ex) * exp2i(-ew);
    // filter: bw[tap][step][piece] = W[n = 32 step + 8 kq .. + 7][c = 16 nt + li][r][s] * 2^ew, split (B operand: K = n, N = c)
    f16x8 bw[9][KS][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 h8, l8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float ws = a.w[((size_t)(32 * ks + 8 * kq + q) * a.CinTotal + a.c_off + 16 * nt + li) * 9 + tap] * w_scale;
                const _Float16 hh = (_Float16)ws;
                h8[q] = hh;
                l8[q] = (_Float16)(ws - (float)hh);
            }
            bw[tap][ks][0] = h8; bw[tap][ks][1] = l8;
        }
    int s_off[DNI], s_dst[DNI];          // staging plan: item = (row, pixel, channel quad), quad fastest; s_dst < 0: none
#pragma unroll
    for (int i = 0; i < DNI; ++i) {
        const int e = tid + 256 * i;
        const int c4 = e % (CY / 4), px = (e / (CY / 4)) % DIP, row = (e / (CY / 4)) / DIP;
        s_off[i] = ((row * a.W + px) * CY + c4 * 4) * 4;
        s_dst[i] = e < DITEMS ? (c4 >> 1) * DPLANE + (row * DIP + px) * 16 + (c4 & 1) * 8 : -1;
    }
    const int tiles_per_img = (a.H + DR - 1) / DR, nstrips = (a.W + DW - 1) / DW;
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    for (int tile = xcd_order(blockIdx.x, gridDim.x); tile < a.B * tiles_per_img; tile += gridDim.x) {
        const int img = tile / tiles_per_img, i0 = (tile - img * tiles_per_img) * DR;
        // rows below the image fall outside the descriptor (-> zeros); columns are tested
        const __amdgpu_buffer_rsrc_t r_x = make_rsrc(a.dy + (size_t)img * a.H * a.W * CY, (unsigned)(a.H * a.W * CY) * 4u);
        __amdgpu_buffer_rsrc_t r_out[NT][2];
#pragma unroll
        for (int rr = 0; rr < NT; ++rr)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int i = i0 + row0 + rr;
                const bool ok = i < a.H;
                r_out[rr][py] = make_rsrc(a.out + ((size_t)img * Ho + (ok ? 2 * i + py : 0)) * Wo * CX, ok ? (unsigned)(Wo * CX) * 4u : 0u);
            }
        f32x4 pre[DNI];
        auto fetch = [&](int strip) {
            const int j0 = strip * DW, so = (i0 * a.W + j0) * CY * 4;
#pragma unroll
            for (int k = 0; k < DNI; ++k) {
                const int px = ((s_dst[k] % DPLANE) >> 4) % DIP;
                const bool ok = s_dst[k] >= 0 && j0 + px < a.W;
                pre[k] = buf_load4(r_x, ok ? s_off[k] + so : BUF_OOB, 0);
            }
        };
        fetch(0);
        for (int strip = 0; strip < nstrips; ++strip) {
            __syncthreads();            // the fragment reads of the previous strip are done
#pragma unroll
            for (int k = 0; k < DNI; ++k) {
                if (256 * (k + 1) > DITEMS && s_dst[k] < 0) continue;
                f16x4 h4, l4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xs = pre[k][q] * x_scale;
                    const _Float16 hh = (_Float16)xs;
                    h4[q] = hh;
                    l4[q] = (_Float16)(xs - (float)hh);
                }
                *reinterpret_cast<f16x4 *>(lds + s_dst[k]) = h4;
                *reinterpret_cast<f16x4 *>(lds + NOCT * DPLANE + s_dst[k]) = l4;
            }
            __syncthreads();
            if (strip + 1 < nstrips) fetch(strip + 1);
            const int j0 = strip * DW;
#pragma unroll
            for (int rr = 0; rr < NT; ++rr) {
#pragma unroll
                for (int mt = 0; mt < DW / 16; ++mt) {
                    if (j0 + mt * 16 >= a.W) break;
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            f32x4v acc = {0.f, 0.f, 0.f, 0.f}, accm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int dr = 0; dr <= py; ++dr) {
#pragma unroll
                                for (int ds = 0; ds <= px; ++ds) {
                                    const int r = py ? 2 - 2 * dr : 1, sx = px ? 2 - 2 * ds : 1, tap = r * 3 + sx;
#pragma unroll
                                    for (int ks = 0; ks < KS; ++ks) {
                                        const unsigned char *p = lds + (ks * 4 + kq) * DPLANE + ((row0 + rr + dr) * DIP + mt * 16 + li + ds) * 16;
                                        const f16x8 ah = *reinterpret_cast<const f16x8 *>(p), al = *reinterpret_cast<const f16x8 *>(p + NOCT * DPLANE);
                                        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bw[tap][ks][0], accm, 0, 0, 0);
                                        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[tap][ks][1], accm, 0, 0, 0);
                                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[tap][ks][0], acc, 0, 0, 0);
                                    }
                                }
                            }
                            // D: column (c) = lane & 15, row (pixel m) = 4 * (lane >> 4) + e -> output pixel 2 (j0 + 16 mt + m) + px
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int j = j0 + mt * 16 + 4 * kq + e;
                                const int voff = j < a.W ? ((2 * j + px) * CX + 16 * nt + li) * 4 : BUF_OOB;
                                float v = (accm[e] + acc[e]) * omul;
                                if (a.accumulate) v += buf_load1(r_out[rr][py], voff, 0);
                                buf_store1(v, r_out[rr][py], voff, 0);
                            }
                        }
                    }
                }
            }
        }
    }
}
bool dgrad_s2_thin_ok(int prec, int ks, int stride, int dyC, int srcC, int CinTotal, int c_off, const unsigned *amax_dy,
                      const unsigned *amax_w, int H, int W) {
    const bool shape = dyC == 32 && srcC == 16;
    return prec == 3 && ks == 3 && stride == 2 && shape && c_off >= 0 && c_off + srcC <= CinTotal && amax_dy && amax_w &&
           (size_t)H * W * dyC * 4 < (1ull << 31) && (size_t)4 * W * srcC * 4 < (1ull << 31);
}
hipError_t launch_dgrad_s2_thin(const float *dy, int B, int H, int W, int dyC, const float *w_master, int CinTotal, int c_off, float *out,
                                int accumulate, const unsigned *amax_dy, const unsigned *amax_w, hipStream_t st) {
    DgradS2Args a;
    a.dy = dy; a.w = w_master; a.out = out; a.B = B; a.H = H; a.W = W; a.CinTotal = CinTotal; a.c_off = c_off;
    a.accumulate = accumulate; a.amax_dy = amax_dy; a.amax_w = amax_w;
    const int tiles = B * ((H + DR - 1) / DR);
    const int srcC = dyC / 2;
    // (mc_profile_train: a data gradient -- 9 tap-MACs per output quad, dY read once, the map written once (+ read when accumulating))
    prof_last = {1, 2.0 * B * H * W * (double)dyC * srcC * 9.0, 4.0 * ((double)B * H * W * dyC + (double)B * 4 * H * W * srcC * (accumulate ? 2 : 1))};
    if (dyC != 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL((dgrad_s2_thin_kernel<1, 1>), dim3(tiles < 4096 ? tiles : 4096), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[n][c][r][s] = sum over pixels of dY[pixel][n] * X[pixel + (r - 1, s - 1)][c] of the same 16 -> 16 layer, mode 3, on
// v_mfma_f32_16x16x32_f16 (K = 32 consecutive pixels of a row).  Replaces wgrad_small_kernel<1, 1> (wgrad_mfma.hip: fp32
// 16x16x4, at its MFMA roof) when wgrad_thin_ok(); same partial layout [workgroup][tap][n][c] and the same split-K reduce.
// Structure of stem_wgrad_f16_kernel (stem_f16.hip): persistent workgroups walk 4-row x 128-pixel tiles, a wave owns a row;
//   * A (dY, NHWC): 8 consecutive pixels of a lane = 8 dword loads 64 bytes apart, scaled and split in registers;
//   * B (X): the 6 x 130 halo window goes through LDS TRANSPOSED, [piece][channel][row][pixel] fp16 -- a lane (channel c,
//     pixel octet g) reads the 16-pixel span of window row (row + r) once per piece (two aligned 16-byte reads) and forms
//     the operands of the three tap columns by register shifts (s = 1: four v_alignbit);
//   * both tensors are scaled by their global maxima (the slots of the fp16-split mode): one scale for all tiles, the MFMA
//     accumulators run across a workgroup's tiles.
namespace {
constexpr int WT_XT = 128, WT_P = 152, WT_ROWS = 6;       // tile width, staged pixel slots per window row, window rows
constexpr int WT_PLANE = 16 * WT_ROWS * WT_P * 2;         // bytes of one piece
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
}  // namespace

__global__ __launch_bounds__(256) void wgrad_thin16_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    unsigned char *win = wsm;                                           // [2][16][6][WT_P] fp16
    float (*red)[64] = reinterpret_cast<float (*)[64]>(wsm + 2 * WT_PLANE);   // [9 * 4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int ex = f16_scale_exp(amax_read(a.amax_x[0])), ed = f16_scale_exp(amax_read(a.amax_dy));
    const float x_scale = exp2i(ex), d_scale = exp2i(ed), omul = exp2i(-ex) * exp2i(-ed);
    const int H = a.Hout, W = a.Wout;
    // lazy X (ConvSrc::la in conv_mfma.h): raw conv output + BatchNorm coefficients; a thread's items are channel quad tid & 3
    const bool lazy = a.src[0].la != nullptr;
    f32x4 lzA = {0.f, 0.f, 0.f, 0.f}, lzB = lzA;
    if (lazy) {
        lzA = reinterpret_cast<const f32x4 *>(a.src[0].la)[tid & 3];
        lzB = reinterpret_cast<const f32x4 *>(a.src[0].lb)[tid & 3];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lzA[q] *= x_scale; lzB[q] *= x_scale; }
    }
    const unsigned x_bytes = (unsigned)(a.Hin * a.Win * 16) * 4u;
    unsigned okm = 0u;          // lazy: bits 2i, 2i + 1 = the two pixels of item i lie inside the image

    f32x4v acc[9], accm[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f}; accm[t] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    const int va = (8 * g * 16 + j) * 4;            // dY lane offset: pixel 8g of the 32-pixel group, channel j
    const int boff = ((j * WT_ROWS + wave) * WT_P + 8 * g) * 2;      // window row (wave + r) adds r * WT_P * 2

    // staging plan: item = (window row, pixel pair, channel quad), quad fastest
    constexpr int NITEM = WT_ROWS * (WT_P / 2) * 4, NIW = (NITEM + 255) / 256;
    int w_off[NIW], w_dst[NIW];                    // global byte offset relative to the window origin; LDS byte offset (< 0: none)
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int e = tid + 256 * i, c4 = e & 3, pr = (e >> 2) % (WT_P / 2), row = (e >> 2) / (WT_P / 2);
        w_off[i] = ((row * a.Win + 2 * pr) * 16 + c4 * 4) * 4;
        w_dst[i] = e < NITEM ? ((c4 * 4 * WT_ROWS + row) * WT_P + 2 * pr) * 2 : -1;
    }
    const int tiles_x = (W + WT_XT - 1) / WT_XT, tiles_y = (H + 3) / 4;
    const int ntiles = a.B * tiles_y * tiles_x;
    f32x4 pv[NIW][2];
    auto fetch = [&](int tl) {
        const int b = tl / (tiles_y * tiles_x), rem = tl - b * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * 4, x0 = (rem % tiles_x) * WT_XT;
        const __amdgpu_buffer_rsrc_t r_x = make_rsrc(a.src[0].p + (size_t)b * a.Hin * a.Win * 16, (unsigned)(a.Hin * a.Win * 16) * 4u);
        const int so = ((y0 - 1) * a.Win + x0 - 1) * 64;        // window origin (row y0 - 1, column x0 - 1); rows outside the
                                                                 // image fall out of the descriptor, columns are tested
        okm = 0u;
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int col = x0 - 1 + ((w_dst[i] >> 1) % WT_P);
            const bool live = w_dst[i] >= 0;
            const int vo0 = (live && col >= 0 && col < a.Win) ? w_off[i] + so : BUF_OOB;
            const int vo1 = (live && col + 1 >= 0 && col + 1 < a.Win) ? w_off[i] + so + 64 : BUF_OOB;
            pv[i][0] = buf_load4(r_x, vo0, 0);
            pv[i][1] = buf_load4(r_x, vo1, 0);
            if (lazy) okm |= (((unsigned)vo0 < x_bytes ? 1u : 0u) | ((unsigned)vo1 < x_bytes ? 2u : 0u)) << (2 * i);
        }
    };
    // dY of a tile (the wave's row: 4 groups of 32 pixels x 8 dword loads) is requested ONE TILE AHEAD, together with the next
    // X window: loaded where it is converted, every tile paid a full memory latency with one other wave per SIMD to cover it
    // (round 6; the kernel reached 3.3 TB/s alone).  Past the last tile the same tile is requested again and never used: the
    // loads stay unconditional (hipcc's wait counts are path-insensitive).
    float araw[WT_XT / 32][8];
    auto fetch_dy = [&](int tl, float (&dst)[WT_XT / 32][8]) {
        const int b = tl / (tiles_y * tiles_x), rem = tl - b * tiles_y * tiles_x;
        const int y = (rem / tiles_x) * 4 + wave, x0 = (rem % tiles_x) * WT_XT;
        const __amdgpu_buffer_rsrc_t r_dy =
            make_rsrc(a.dy + ((size_t)b * H + (y < H ? y : 0)) * W * a.dy_ld, y < H ? (unsigned)(W * a.dy_ld) * 4u : 0u);
#pragma unroll
        for (int G = 0; G < WT_XT / 32; ++G)
#pragma unroll
            for (int t = 0; t < 8; ++t) dst[G][t] = buf_load1(r_dy, va + t * 64, (x0 + G * 32) * 64);      // beyond the row: zero
    };
    if ((int)blockIdx.x < ntiles) { fetch(blockIdx.x); fetch_dy(blockIdx.x, araw); }
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        __syncthreads();                         // the previous tile's reads are done
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            if (w_dst[i] < 0) continue;
            const float cap0 = ((okm >> (2 * i)) & 1u) ? __builtin_inff() : 0.f, cap1 = ((okm >> (2 * i + 1)) & 1u) ? __builtin_inff() : 0.f;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float s0 = lazy ? lazy_act(pv[i][0][ch], lzA[ch], lzB[ch], cap0) : pv[i][0][ch] * x_scale;
                const float s1 = lazy ? lazy_act(pv[i][1][ch], lzA[ch], lzB[ch], cap1) : pv[i][1][ch] * x_scale;
                const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
                const f16x2 hp = {h0, h1}, lp = {(_Float16)(s0 - (float)h0), (_Float16)(s1 - (float)h1)};
                const int off = w_dst[i] + ch * WT_ROWS * WT_P * 2;
                *reinterpret_cast<f16x2 *>(win + off) = hp;
                *reinterpret_cast<f16x2 *>(win + WT_PLANE + off) = lp;
            }
        }
        __syncthreads();
        const int tnext = tl + (int)gridDim.x < ntiles ? tl + (int)gridDim.x : tl;
        if (tl + (int)gridDim.x < ntiles) fetch(tnext);
        float anext[WT_XT / 32][8];
        fetch_dy(tnext, anext);
#pragma unroll
        for (int G = 0; G < WT_XT / 32; ++G) {
            f16x8 ah, al;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float ds = araw[G][t] * d_scale;
                const _Float16 hh = (_Float16)ds;
                ah[t] = hh;
                al[t] = (_Float16)(ds - (float)hh);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                unsigned w[2][8];
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const unsigned char *p = win + z * WT_PLANE + boff + r * WT_P * 2 + G * 64;
                    const u32x4 lo = *reinterpret_cast<const u32x4 *>(p), hi = *reinterpret_cast<const u32x4 *>(p + 16);
#pragma unroll
                    for (int d = 0; d < 4; ++d) { w[z][d] = lo[d]; w[z][4 + d] = hi[d]; }
                }
#pragma unroll
                for (int sft = 0; sft < 3; ++sft) {
                    u32x4 bz[2];
#pragma unroll
                    for (int z = 0; z < 2; ++z)
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            bz[z][d] = (sft & 1) ? __builtin_amdgcn_alignbit(w[z][d + sft / 2 + 1], w[z][d + sft / 2], 16) : w[z][d + sft / 2];
                    const f16x8 bh = __builtin_bit_cast(f16x8, bz[0]), bl = __builtin_bit_cast(f16x8, bz[1]);
                    const int t = r * 3 + sft;
                    accm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accm[t], 0, 0, 0);
                    accm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accm[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int G = 0; G < WT_XT / 32; ++G)
#pragma unroll
            for (int t = 0; t < 8; ++t) araw[G][t] = anext[G][t];
    }
    // workgroup reduction, wave after wave (fixed order).  D: row (n) = 4 * (lane >> 4) + q, column (c) = lane & 15
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float *dst = &red[t * 4 + q][lane];
                    const float v = (accm[t][q] + acc[t][q]) * omul;
                    *dst = wv == 0 ? v : *dst + v;
                }
        }
        __syncthreads();
    }
    for (int e = tid; e < 9 * 4 * 64; e += 256) {
        const int l = e & 63, idx = e >> 6;
        const int q = idx & 3, t = idx >> 2;
        const int n = 4 * (l >> 4) + q, c = l & 15;
        a.partial[(((size_t)blockIdx.x * 9 + t) * 16 + n) * 16 + c] = red[idx][l];
    }
}

bool wgrad_thin_ok(const WgradArgs &a, int ks, int stride) {
#ifdef MC_NO_WGRAD_THIN
    return false;
#endif
    return a.prec == 3 && a.small && ks == 3 && stride == 1 && a.nsrc == 1 && a.Cin == 16 && a.Cout == 16 && a.dy_ld == 16 &&
           a.amax_x[0] && a.amax_dy && a.Wout % 4 == 0 && a.Hout == a.Hin && a.Wout == a.Win;
}

hipError_t launch_wgrad_thin(const WgradArgs &a, hipStream_t st) {
    const size_t lds = 2 * WT_PLANE + 9 * 4 * 64 * sizeof(float);
    static DynLdsOnce attr_set;
    {
        const hipError_t e = attr_set.ensure(reinterpret_cast<const void *>(wgrad_thin16_kernel), (int)(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(wgrad_thin16_kernel, dim3(a.ksplit), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace mc
