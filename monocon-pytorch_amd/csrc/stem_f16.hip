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
                                                       const float *__restrict__ stat_shift,
                                                       unsigned *__restrict__ img_amax /*max |image| slot or null*/) {
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
        if (img_amax && tid == 0) amax_commit(img_amax, s_max);      // the weight gradient scales the image by its global maximum
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
                           float *out, hipStream_t st, int relu, unsigned *amax_out, float *stats, const float *stat_shift,
                           unsigned *img_amax) {
    const int tiles = (H + SR - 1) / SR;          // one workgroup per band of 8 rows
    hipLaunchKernelGGL(stem_f16_kernel, dim3(B * tiles), dim3(256), 0, st, img, B, H, W, wpk, scale, shift, out, relu, amax_out, stats,
                       stat_shift, img_amax);
    return hipGetLastError();
}


// ---- the stem's weight gradient on the same pipe.  Structure of stem_wgrad_lds_kernel (kernels_head_train.hip): persistent
// workgroups walk 4-row x 128-pixel output tiles, the 3 x 10 x 134 image window of a tile goes through LDS, a wave owns one
// row, D[n][t] = sum over pixels of dY[pixel][n] * img[c(t)][y + r(t) - 3][x + s(t) - 3] over the 147 taps t = (c, r, s).
// With K = 32 pixels per v_mfma_f32_16x16x32_f16 instead of 4:
//   * A (dY, NHWC): the 8 consecutive pixels of a lane are 8 dword loads 64 bytes apart (16 lanes = one 64-byte pixel
//     row each) -- the same number of loads per pixel as before -- scaled and split in registers;
//   * B (image): a lane's 8 pixels start at x + 8g + s, s = 0..6 the tap's column.  Taps are tiled BY s: the 16 columns
//     of an MFMA are 16 of the 21 window rows (c, r), all with the same s -- then all seven s-tiles of a lane read the
//     same 16-pixel span of the same row: two aligned 16-byte LDS reads per piece, and the seven operands are that span
//     shifted by s elements in registers (nothing for even s, four v_alignbit for odd s; s is a compile-time constant of
//     the tile).  14 tiles (2 row groups x 7) instead of 10: 40 % more MFMAs, five times fewer LDS bytes.  (gfx950 does
//     serve unaligned ds_read_b128 -- scratch/ua/ua.hip -- but a first version built on 20 unaligned reads per K-step ran
//     at half the speed of the fp32 kernel.)
//   * the image is scaled by its GLOBAL maximum (folded into a slot by the forward stem kernel), dY by its tensor's slot:
//     partial sums of different tiles then share one scale and accumulate in the MFMA accumulators across tiles.
// 42 MFMAs of 16 cycles per 32 pixels instead of 80 of 32 cycles.
namespace {
constexpr int GW_XT = 128, GW_P = 144, GW_ROWS = 10;          // tile width, staged pixels per window row (134 used), rows
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
}  // namespace

// FUSED (round 4): dY is not read but formed on the fly, dY = P_c d + Q_c y + R_c, from the masked gradient d of the stem's
// activation map (left by the backward-statistics epilogue of level0's data gradient), the raw stem output y and the
// BatchNorm-backward coefficients -- the stem's dY has no other reader (no data gradient flows into the image), so its
// element-wise pass (3 GB of traffic, 0.77 ms, the LAST launch of the caller's stream in a backward) does not exist any more.
// Same fma chain as affine_bwd_kernel, i.e. the same fp32 dY; the operand scale comes from the sound bound
// max|P| max|d| + max|Q| max|y| + max|R| (slots `dy_amax` = max |d| and `y_amax`) instead of the exact maximum of dY.
template <bool FUSED>
__global__ __launch_bounds__(256) void stem_wgrad_f16_kernel(const float *__restrict__ img, const float *__restrict__ dy, int B,
                                                             int H, int W, float *__restrict__ partial,
                                                             const unsigned *__restrict__ img_amax,
                                                             const unsigned *__restrict__ dy_amax,
                                                             const float *__restrict__ ybn, const float *__restrict__ coef,
                                                             const unsigned *__restrict__ y_amax) {
    constexpr int PLANE_B = 3 * GW_ROWS * GW_P * 2;      // bytes of one piece
    __shared__ __attribute__((aligned(16))) unsigned char win[2 * PLANE_B];
    __shared__ float red[14 * 4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    float cP = 0.f, cQ = 0.f, cR = 0.f;            // FUSED: coefficients of this lane's channel j
    int ed;
    if constexpr (FUSED) {
        const f32x4v cf = reinterpret_cast<const f32x4v *>(coef)[j];
        cP = cf[0]; cQ = cf[1]; cR = cf[2];
        float mP = fabsf(cP), mQ = fabsf(cQ), mR = fabsf(cR);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {          // over the 16 channels (every 16-lane group holds all of them)
            mP = fmaxf(mP, __shfl_xor(mP, o)); mQ = fmaxf(mQ, __shfl_xor(mQ, o)); mR = fmaxf(mR, __shfl_xor(mR, o));
        }
        const float bound = mP * __builtin_bit_cast(float, amax_read(dy_amax)) + mQ * __builtin_bit_cast(float, amax_read(y_amax)) + mR;
        ed = f16_scale_exp(__builtin_bit_cast(unsigned, bound));
    } else {
        ed = f16_scale_exp(amax_read(dy_amax));
    }
    const int ex = f16_scale_exp(amax_read(img_amax));
    const float x_scale = exp2i(ex), d_scale = exp2i(ed), omul = exp2i(-ex) * exp2i(-ed);
    // window row (c, r) of this lane in row group 0 / 1 (21 rows: 16 + 5; the other 11 lanes of group 1 read row 0)
    int boff[2];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const int cr = grp * 16 + j, c = cr / 7, r = cr - 7 * c;
        boff[grp] = cr < 21 ? ((c * GW_ROWS + wave + r) * GW_P + 8 * g) * 2 : 0;
    }
    f32x4v acc[2][7], accm[2][7];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int sft = 0; sft < 7; ++sft) { acc[grp][sft] = f32x4v{0.f, 0.f, 0.f, 0.f}; accm[grp][sft] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    const int va = (8 * g * 16 + j) * 4;            // dY lane offset: pixel 8g of the 32-pixel group, channel j
    const int tiles_x = (W + GW_XT - 1) / GW_XT, tiles_y = (H + 3) / 4;
    const int ntiles = B * tiles_y * tiles_x;
    // staging plan: thread = pairs of horizontally adjacent window pixels (one packed dword per piece); the next tile's
    // pixels are loaded before the MFMAs of the current one
    constexpr int NPAIR = 3 * GW_ROWS * (GW_P / 2), NIW = (NPAIR + 255) / 256;
    int w_c[NIW], w_rr[NIW], w_col[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int e = tid + 256 * i, row = e / (GW_P / 2);
        w_c[i] = e < NPAIR ? row / GW_ROWS : -1;
        w_rr[i] = row % GW_ROWS;
        w_col[i] = (e - row * (GW_P / 2)) * 2;
    }
    float pv[NIW][2];
    auto fetch = [&](int tl) {
        const int b = tl / (tiles_y * tiles_x), rem = tl - b * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * 4, x0 = (rem % tiles_x) * GW_XT;
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int gy = y0 - 3 + w_rr[i], gx = x0 - 3 + w_col[i];
            const bool rok = w_c[i] >= 0 && gy >= 0 && gy < H;
            const float *src = img + (((size_t)b * 3 + (rok ? w_c[i] : 0)) * H + (rok ? gy : 0)) * W;
            pv[i][0] = (rok && gx >= 0 && gx < W) ? src[gx] : 0.f;
            pv[i][1] = (rok && gx + 1 >= 0 && gx + 1 < W) ? src[gx + 1] : 0.f;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int b = tl / (tiles_y * tiles_x), rem = tl - b * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * 4, x0 = (rem % tiles_x) * GW_XT;
        __syncthreads();                         // the previous tile's reads are done
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            if (w_c[i] < 0) continue;
            const float s0 = pv[i][0] * x_scale, s1 = pv[i][1] * x_scale;
            const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 hp = {h0, h1}, lp = {(_Float16)(s0 - (float)h0), (_Float16)(s1 - (float)h1)};
            const int off = ((w_c[i] * GW_ROWS + w_rr[i]) * GW_P + w_col[i]) * 2;
            *reinterpret_cast<f16x2 *>(win + off) = hp;
            *reinterpret_cast<f16x2 *>(win + PLANE_B + off) = lp;
        }
        __syncthreads();
        if (tl + (int)gridDim.x < ntiles) fetch(tl + gridDim.x);
        const int y = y0 + wave;
        const __amdgpu_buffer_rsrc_t r_dy =
            make_rsrc(dy + ((size_t)b * H + (y < H ? y : 0)) * W * 16, y < H ? (unsigned)(W * 16) * 4u : 0u);
        float araw[GW_XT / 32][8];      // all four 32-pixel groups of the row up front: the loads of group G + 1.. are in
                                        // flight under the MFMAs of group G (a first version loaded per group and waited)
#pragma unroll
        for (int G = 0; G < GW_XT / 32; ++G)
#pragma unroll
            for (int t = 0; t < 8; ++t) araw[G][t] = buf_load1(r_dy, va + t * 64, (x0 + G * 32) * 64);      // beyond the row: zero
        if constexpr (FUSED) {
            const __amdgpu_buffer_rsrc_t r_y =
                make_rsrc(ybn + ((size_t)b * H + (y < H ? y : 0)) * W * 16, y < H ? (unsigned)(W * 16) * 4u : 0u);
            float yraw[GW_XT / 32][8];
#pragma unroll
            for (int G = 0; G < GW_XT / 32; ++G)
#pragma unroll
                for (int t = 0; t < 8; ++t) yraw[G][t] = buf_load1(r_y, va + t * 64, (x0 + G * 32) * 64);
            // (pixels beyond the row / rows beyond the image: d = y = 0 gives R, not 0 -- mask them)
#pragma unroll
            for (int G = 0; G < GW_XT / 32; ++G)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bool in = y < H && x0 + G * 32 + 8 * g + t < W;
                    araw[G][t] = in ? fmaf(cP, araw[G][t], fmaf(cQ, yraw[G][t], cR)) : 0.f;
                }
        }
#pragma unroll
        for (int G = 0; G < GW_XT / 32; ++G) {
            f16x8 ah, al;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float ds = araw[G][t] * d_scale;
                const _Float16 hh = (_Float16)ds;
                ah[t] = hh;
                al[t] = (_Float16)(ds - (float)hh);
            }
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                // pixels 8g .. 8g + 15 of the lane's window row, both pieces: w[piece][0..7] dwords of two pixels each
                unsigned w[2][8];
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const u32x4 lo = *reinterpret_cast<const u32x4 *>(win + z * PLANE_B + boff[grp] + G * 64);
                    const u32x4 hi = *reinterpret_cast<const u32x4 *>(win + z * PLANE_B + boff[grp] + G * 64 + 16);
#pragma unroll
                    for (int d = 0; d < 4; ++d) { w[z][d] = lo[d]; w[z][4 + d] = hi[d]; }
                }
#pragma unroll
                for (int sft = 0; sft < 7; ++sft) {
                    u32x4 bz[2];
#pragma unroll
                    for (int z = 0; z < 2; ++z)
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            bz[z][d] = (sft & 1) ? __builtin_amdgcn_alignbit(w[z][d + sft / 2 + 1], w[z][d + sft / 2], 16) : w[z][d + sft / 2];
                    const f16x8 bh = __builtin_bit_cast(f16x8, bz[0]), bl = __builtin_bit_cast(f16x8, bz[1]);
                    accm[grp][sft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accm[grp][sft], 0, 0, 0);
                    accm[grp][sft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accm[grp][sft], 0, 0, 0);
                    acc[grp][sft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[grp][sft], 0, 0, 0);
                }
            }
        }
    }
    // workgroup reduction, wave after wave (fixed order); D: row (out channel n) = 4*(lane>>4) + q, column = window row j
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                for (int sft = 0; sft < 7; ++sft)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float *dst = &red[(grp * 7 + sft) * 4 + q][lane];
                        const float v = (accm[grp][sft][q] + acc[grp][sft][q]) * omul;
                        *dst = wv == 0 ? v : *dst + v;
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < 14 * 4 * 64; e += 256) {
        const int l = e & 63, idx = e >> 6;
        const int q = idx & 3, tile = idx >> 2, grp = tile / 7, sft = tile - 7 * grp;
        const int cr = grp * 16 + (l & 15), n = 4 * (l >> 4) + q;
        if (cr < 21) partial[((size_t)blockIdx.x * 147 + cr * 7 + sft) * 16 + n] = red[idx][l];     // t = c*49 + r*7 + s
    }
}

hipError_t launch_stem_wgrad_f16(const float *img, const float *dy, int B, int H, int W, float *partial, int nblocks,
                                 const unsigned *img_amax, const unsigned *dy_amax, hipStream_t st, const float *y,
                                 const float *coef, const unsigned *y_amax) {
    if (y) {        // dy = the masked gradient d of the stem's activation map; dY is formed on the fly
        if (!coef || !y_amax) return hipErrorInvalidValue;
        hipLaunchKernelGGL(stem_wgrad_f16_kernel<true>, dim3(nblocks), dim3(256), 0, st, img, dy, B, H, W, partial, img_amax, dy_amax, y,
                           coef, y_amax);
    } else {
        hipLaunchKernelGGL(stem_wgrad_f16_kernel<false>, dim3(nblocks), dim3(256), 0, st, img, dy, B, H, W, partial, img_amax, dy_amax,
                           nullptr, nullptr, nullptr);
    }
    return hipGetLastError();
}

}  // namespace mc
