// Non-GEMM kernels of the MonoCon forward: parameter packing, 7x7 stem, 2x2 max-pool, depthwise
// 4x4 transposed convolution, layout transposes and the attentive-norm dense-head passes.
// All activations are NHWC fp32; every kernel is bandwidth-bound and written for 16-byte
// per-lane accesses.
#include <algorithm>
#include "kernels.h"
#include "conv_mfma.h"

namespace mc {

// ============================================================================ packing
__global__ void pack_conv_w_kernel(const float *__restrict__ w, int Cout, int Cin, int kk, float *dst,
                                   int CinTotal, int CoutP, int n_off, int c_off) {
    const size_t total = (size_t)Cout * Cin * kk;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tap = e % kk;
        const int c = (e / kk) % Cin;
        const int n = e / ((size_t)kk * Cin);
        const int cg = c + c_off;
        dst[(((size_t)tap * (CinTotal >> 2) + (cg >> 2)) * CoutP + n + n_off) * 4 + (cg & 3)] = w[e];
    }
}

hipError_t launch_pack_conv_w(const float *w, int Cout, int Cin, int ks, float *dst, int CinTotal, int CoutP,
                              int n_off, int c_off, hipStream_t st) {
    const size_t total = (size_t)Cout * Cin * ks * ks;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_w_kernel, dim3(blocks), dim3(256), 0, st, w, Cout, Cin, ks * ks, dst, CinTotal,
                       CoutP, n_off, c_off);
    return hipGetLastError();
}

hipError_t launch_zero(float *p, size_t n, hipStream_t st) { return hipMemsetAsync(p, 0, n * sizeof(float), st); }

hipError_t launch_copy(const float *src, float *dst, size_t n, hipStream_t st) {
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
}

__global__ __launch_bounds__(256) void copy_batch_kernel(const CopyBatch cb) {
    const int seg = blockIdx.y;                    // wave-uniform: the table is read through scalar loads
    const float *s = cb.src[seg];
    float *d = cb.dst[seg];
    const int n = cb.n[seg];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}
hipError_t launch_copy_batch(const CopyBatch &cb, hipStream_t st) {
    if (cb.count <= 0) return hipSuccess;
    int mx = 1;
    for (int i = 0; i < cb.count; ++i) mx = cb.n[i] > mx ? cb.n[i] : mx;
    int gx = (mx + 1023) / 1024;                   // ~4 elements per thread for the largest segment
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(copy_batch_kernel, dim3(gx, cb.count), dim3(256), 0, st, cb);
    return hipGetLastError();
}

__global__ void fold_bn_kernel(const float *g, const float *b, const float *rm, const float *rv, float eps, int C,
                               float *scale, float *shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    const float s = (g ? g[c] : 1.f) * inv;
    scale[c] = s;
    shift[c] = (b ? b[c] : 0.f) - rm[c] * s;
}

hipError_t launch_fold_bn(const float *g, const float *b, const float *rm, const float *rv, float eps, int C,
                          float *scale, float *shift, hipStream_t st) {
    hipLaunchKernelGGL(fold_bn_kernel, dim3((C + 63) / 64), dim3(64), 0, st, g, b, rm, rv, eps, C, scale, shift);
    return hipGetLastError();
}

__global__ void pack_stem_w_kernel(const float *w, float *dst) {
    // (16,3,7,7) -> [c][r][s][16]
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 16 * 147) return;
    const int o = e / 147, rem = e % 147;
    dst[rem * 16 + o] = w[e];
}
hipError_t launch_pack_stem_w(const float *w, float *dst, hipStream_t st) {
    hipLaunchKernelGGL(pack_stem_w_kernel, dim3((16 * 147 + 255) / 256), dim3(256), 0, st, w, dst);
    return hipGetLastError();
}

__global__ void pack_deconv_w_kernel(const float *w, int C, float *dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * 16) return;
    const int c = e / 16, k = e % 16;
    dst[k * C + c] = w[e];
}
hipError_t launch_pack_deconv_w(const float *w, int C, float *dst, hipStream_t st) {
    hipLaunchKernelGGL(pack_deconv_w_kernel, dim3((C * 16 + 255) / 256), dim3(256), 0, st, w, C, dst);
    return hipGetLastError();
}

// ============================================================================ 7x7 stem
// NCHW (B,3,H,W) -> NHWC (B,H,W,16), conv7x7 pad 3 + folded BN + ReLU.  VALU direct convolution:
// a 16x64-pixel tile per workgroup, 4 x-adjacent pixels x 16 channels per thread; weights are
// wave-uniform and come through the scalar cache.  (reference model/backbone/dla.py:231-234)
constexpr int ST_TH = 16, ST_TW = 64, ST_LW = 72;
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ img, int B, int H, int W,
                                                   const float *__restrict__ wpk, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, float *__restrict__ out, int relu) {
    __shared__ __attribute__((aligned(16))) float tile[3][ST_TH + 6][ST_LW];
    const int tiles_x = (W + ST_TW - 1) / ST_TW, tiles_y = (H + ST_TH - 1) / ST_TH;
    const int bt = blockIdx.x;
    const int b = bt / (tiles_x * tiles_y);
    const int ty0 = ((bt / tiles_x) % tiles_y) * ST_TH, tx0 = (bt % tiles_x) * ST_TW;
    const int tid = threadIdx.x;
    for (int e = tid; e < 3 * (ST_TH + 6) * ST_LW; e += 256) {
        const int lx = e % ST_LW, ly = (e / ST_LW) % (ST_TH + 6), c = e / (ST_LW * (ST_TH + 6));
        const int y = ty0 - 3 + ly, x = tx0 - 3 + lx;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((size_t)b * 3 + c) * H + y) * W + x];
        tile[c][ly][lx] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[p][o] = 0.f;
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 7; ++r) {
            float in[12];
            const f32x4 *row = reinterpret_cast<const f32x4 *>(&tile[c][ty + r][tx * 4]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 v = row[q];
                in[q * 4 + 0] = v[0]; in[q * 4 + 1] = v[1]; in[q * 4 + 2] = v[2]; in[q * 4 + 3] = v[3];
            }
            const float *wr = wpk + (c * 7 + r) * 7 * 16;
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int o = 0; o < 16; ++o) {
                    const float wv = wr[s * 16 + o];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p][o] = fmaf(in[p + s], wv, acc[p][o]);
                }
        }
    }
    const int y = ty0 + ty;
    if (y >= H) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int x = tx0 + tx * 4 + p;
        if (x >= W) continue;
        f32x4 *dst = reinterpret_cast<f32x4 *>(out + (((size_t)b * H + y) * W + x) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = fmaf(acc[p][q * 4 + j], scale[q * 4 + j], shift[q * 4 + j]);
                v[j] = relu ? fmaxf(t, 0.f) : t;
            }
            dst[q] = v;
        }
    }
}

bool stem_f16_enabled() {
#ifdef MC_NO_STEM_F16
    return false;
#else
    return true;
#endif
}
hipError_t launch_stem(const float *img, int B, int H, int W, const float *wpk, const float *scale,
                       const float *shift, float *out, hipStream_t st, int relu, int prec, unsigned *amax) {
#ifndef MC_NO_STEM_F16
    if (prec == 3) return launch_stem_f16(img, B, H, W, wpk, scale, shift, out, st, relu, amax);
#endif
    const int tiles = ((W + ST_TW - 1) / ST_TW) * ((H + ST_TH - 1) / ST_TH);
    hipLaunchKernelGGL(stem_kernel, dim3(B * tiles), dim3(256), 0, st, img, B, H, W, wpk, scale, shift, out, relu);
    return hipGetLastError();
}

// ============================================================================ pool / deconv / layout
__global__ void maxpool2_kernel(const f32x4 *__restrict__ in, int B, int H, int W, int C4, f32x4 *__restrict__ out,
                                const f32x4 *__restrict__ la, const f32x4 *__restrict__ lb) {
    // la / lb: lazy input (ConvSrc::la in conv_mfma.h) -- the window is pooled over max(fma(y, la, lb), 0)
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = e % C4;
        const size_t p = e / C4;
        const int x = p % Wo, y = (p / Wo) % Ho, b = p / ((size_t)Wo * Ho);
        const f32x4 *r0 = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C4 + c;
        const f32x4 *r1 = r0 + (size_t)W * C4;
        f32x4 a = r0[0], bb = r0[C4], cc = r1[0], d = r1[C4];
        if (la) {
            const f32x4 av = la[c], bv = lb[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = fmaxf(fmaf(a[j], av[j], bv[j]), 0.f); bb[j] = fmaxf(fmaf(bb[j], av[j], bv[j]), 0.f);
                cc[j] = fmaxf(fmaf(cc[j], av[j], bv[j]), 0.f); d[j] = fmaxf(fmaf(d[j], av[j], bv[j]), 0.f);
            }
        }
        f32x4 m;
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a[j], bb[j]), fmaxf(cc[j], d[j]));
        out[e] = m;
    }
}
static inline int grid_for(size_t total, int bs) {
    size_t g = (total + bs - 1) / bs;
    return (int)(g > 16384 ? 16384 : (g == 0 ? 1 : g));
}
hipError_t launch_maxpool2(const float *in, int B, int H, int W, int C, float *out, hipStream_t st, const float *la,
                           const float *lb) {
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st,
                       reinterpret_cast<const f32x4 *>(in), B, H, W, C / 4, reinterpret_cast<f32x4 *>(out),
                       reinterpret_cast<const f32x4 *>(la), reinterpret_cast<const f32x4 *>(lb));
    return hipGetLastError();
}

// depthwise ConvTranspose2d(k=4, s=2, p=1): out[oy,ox] = sum over the <=2x2 inputs with
// oy = 2*iy - 1 + ky (reference model/backbone/dla_neck.py:58-65)
// grid = (ceil(Wo * C4 / 256), ceil(Ho / 4), B): the rows and the image come from the block index, the column / channel quad from one
// 32-bit division (round 5: as a flat grid-stride loop every output paid four 64-bit divisions -- 3.3 TB/s)
__global__ __launch_bounds__(256) void deconv4_kernel(const f32x4 *__restrict__ in, int B, int H, int W, int C4,
                                                      const f32x4 *__restrict__ wpk, f32x4 *__restrict__ out,
                                                      unsigned *__restrict__ amax, const f32x4 *__restrict__ la,
                                                      const f32x4 *__restrict__ lb) {
    // Round 6: the four output rows of a block share four input rows (h - 1 .. h + 2, h = oy0 / 2) and two input columns:
    // all 8 inputs and 8 filter taps of the thread are requested up front (one output at a time, behind `continue`
    // branches, the kernel had 16 dependent-ish loads per thread: 118 us for 315 MB at B = 32), then the four outputs are
    // formed in the ORIGINAL order of their <= 4 terms (dy = 0: dx = 0, 1; dy = 1: dx = 0, 1) -- an absent input is a zero
    // term, which leaves the sum's bits unchanged.
    const int Wo = 2 * W;
    float vmax = 0.f;          // max |out| of this thread (amax != null, see ConvArgs::amax_in)
    const unsigned ex = blockIdx.x * 256u + threadIdx.x;
    const int b = blockIdx.z;
    const bool live = ex < (unsigned)(Wo * C4);
    const int c = (int)(ex % (unsigned)C4), ox = (int)(ex / (unsigned)C4);
    if (live) {
        const int oy0 = blockIdx.y * 4, h = oy0 >> 1;
        const int ix1 = (ox + 1) >> 1, kx1 = ox + 1 - 2 * ix1;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 v[4][2], w[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = h - 1 + r;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ix = ix1 - dx;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                v[r][dx] = ok ? in[(((size_t)b * H + iy) * W + ix) * C4 + c] : zero;
                w[r][dx] = wpk[(r * 4 + kx1 + 2 * dx) * C4 + c];          // filter row ky = r (used below by its own ky), column kx1 + 2 dx
            }
        }
        if (la) {               // lazy input (ConvSrc::la in conv_mfma.h): formed on load; positions outside stay 0
            const f32x4 lav = la[c], lbv = lb[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iy = h - 1 + r;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int ix = ix1 - dx;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[r][dx][j] = fmaxf(fmaf(v[r][dx][j], lav[j], lbv[j]), 0.f);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = oy0 + q;
            if (oy >= 2 * H) break;
            // oy = 2h + q: input row iy1 = h + (q + 1) / 2, filter row ky1 = 1 - q % 2 (compile-time: v / w stay registers)
            f32x4 acc = zero;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int r = ((q + 1) >> 1) - dy + 1, ky = (1 - (q & 1)) + 2 * dy;      // row of v (input row h - 1 + r), filter row
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[r][dx][j], w[ky][dx][j], acc[j]);
                }
            }
            out[(((size_t)b * (2 * H) + oy) * Wo) * C4 + ex] = acc;
#pragma unroll
            for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(acc[j]));
        }
    }
    if (amax) amax_update_block(amax, vmax);
}
hipError_t launch_deconv4(const float *in, int B, int H, int W, int C, const float *wpk, float *out, hipStream_t st,
                          unsigned *amax, const float *la, const float *lb) {
    if (B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(deconv4_kernel, dim3((unsigned)((2 * W * (C / 4) + 255) / 256), (unsigned)((2 * H + 3) / 4), (unsigned)B), dim3(256), 0, st,
                       reinterpret_cast<const f32x4 *>(in), B, H, W, C / 4, reinterpret_cast<const f32x4 *>(wpk),
                       reinterpret_cast<f32x4 *>(out), amax, reinterpret_cast<const f32x4 *>(la), reinterpret_cast<const f32x4 *>(lb));
    return hipGetLastError();
}

// Plan-build aid: ReLU-shaped pseudo-random fill (half zeros, the rest uniform in (0, 4)) of a buffer the autotuner is
// about to time kernels on.  A freshly allocated plan buffer is all zeros, and on this part a kernel that multiplies zeros
// clocks 20-40 % higher than the same kernel on real activations (DVFS, profiles/r4_mfma_f16_power_ceiling.txt): shapes
// were being ranked in a regime the step never runs in.
__global__ void noise_fill_kernel(float *__restrict__ p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (h & 1u) ? 0.f : (float)(h >> 8) * (4.0f / 16777216.0f);
    }
}
hipError_t launch_noise_fill(float *p, size_t n, unsigned seed, hipStream_t st) {
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)g), dim3(256), 0, st, p, n, seed);
    return hipGetLastError();
}

// max |x| of a dense tensor -> *slot (bit pattern of a non-negative float; see amax_update): the operand-scale input of
// the fp16-split mode for tensors that come from outside the plans' own producers
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, unsigned *__restrict__ slot, int single) {
    float vmax = 0.f;
    const size_t n4 = n / 4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = x4[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(v[j]));
    }
    for (size_t e = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        vmax = fmaxf(vmax, fabsf(x[e]));
    if (!single) { amax_update_block(slot, vmax); return; }
    __shared__ unsigned s_max;                   // single-word slot (a weight's maximum)
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
    const unsigned bits = __builtin_bit_cast(unsigned, vmax);
    if (bits < 0x7f800000u && bits != 0u) atomicMax(&s_max, bits);
    __syncthreads();
    if (threadIdx.x == 0 && s_max != 0u) atomicMax(slot, s_max);
}
hipError_t launch_absmax(const float *x, size_t n, unsigned *slot, hipStream_t st, bool single_word) {
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(absmax_kernel, dim3(std::min(grid_for(n / 4 + 1, 256), single_word ? 64 : 2048)), dim3(256), 0, st, x, n, slot,
                       single_word ? 1 : 0);
    return hipGetLastError();
}

// tiled transposes between (B,C,HW) and (B,HW,C) through LDS, 32x32 tiles
__global__ void transpose_kernel(const float *__restrict__ in, int B, int R, int Cn, float *__restrict__ out) {
    // in: (B, R, Cn) row-major -> out: (B, Cn, R)
    __shared__ float t[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        t[i][tx] = (r < R && c < Cn) ? in[((size_t)b * R + r) * Cn + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cn) out[((size_t)b * Cn + c) * R + r] = t[tx][i];
    }
}
hipError_t launch_nchw_to_nhwc(const float *in, int B, int C, int H, int W, float *out, hipStream_t st) {
    const int HW = H * W;   // (B, C, HW) -> (B, HW, C)
    hipLaunchKernelGGL(transpose_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, st, in, B, C, HW, out);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(const float *in, int B, int C, int H, int W, float *out, hipStream_t st) {
    const int HW = H * W;   // (B, HW, C) -> (B, C, HW)
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (HW + 31) / 32, B), dim3(256), 0, st, in, B, HW, C, out);
    return hipGetLastError();
}

// ============================================================================ dense heads
// Row table: the 65 output channels grouped by producing head (head order = conv column order of
// the fused 64->576 3x3): heatmap, wh, offset, center2kpt_offset, kpt_heatmap, kpt_heatmap_offset,
// dim, depth, dir_feat(cls 12 + reg 12).   (reference monocon_heads.py:76-88,165-200)
static HeadRow g_rows[NUM_OUT_ROWS];
static int g_row_begin[NUM_HEADS + 1];
static bool g_rows_init = false;
static void init_rows() {
    if (g_rows_init) return;
    // head -> (pred index, channels, epilogue)
    const int hp[8][3] = {{0, 3, 1}, {2, 2, 0}, {3, 2, 0}, {5, 18, 0}, {1, 9, 1}, {4, 2, 0}, {6, 3, 0}, {7, 2, 0}};
    int r = 0;
    for (int h = 0; h < 8; ++h) {
        g_row_begin[h] = r;
        for (int c = 0; c < hp[h][1]; ++c) {
            int epi = hp[h][2];
            if (h == 7) epi = (c == 0) ? 2 : 0;
            g_rows[r++] = HeadRow{h, hp[h][0], c, epi};
        }
    }
    g_row_begin[8] = r;
    for (int c = 0; c < 12; ++c) g_rows[r++] = HeadRow{8, 8, c, 0};
    for (int c = 0; c < 12; ++c) g_rows[r++] = HeadRow{8, 9, c, 0};
    g_row_begin[9] = r;
    g_rows_init = true;
}
const HeadRow *head_rows() { init_rows(); return g_rows; }
const int *head_row_begin() { init_rows(); return g_row_begin; }

// AttnBN attention path (reference model/norm/attentive_norm.py:79-91,154-164), eval mode.
// One 64-lane workgroup per (image, head); lane = channel.
__global__ __launch_bounds__(256) void head_attn_kernel(const float *__restrict__ stats, int chunks, int HW,
                                                         HeadAttnParams p, float *__restrict__ scale,
                                                         float *__restrict__ shift) {
    const int b = blockIdx.x, h = blockIdx.y, c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int CP = NUM_HEADS * HEAD_CH;
    // per-patch partials (conv epilogue) summed in fp64: four waves take interleaved quarters, fixed order
    double s1 = 0.0, s2 = 0.0;
    const float2 *q = reinterpret_cast<const float2 *>(stats) + (size_t)b * chunks * CP + h * HEAD_CH + c;
    int k = part;
    for (; k + 28 < chunks; k += 32) {       // eight loads in flight (one at a time, every partial was a full round trip: 140 us);
        float2 v[8];                         // summed in the order of the plain loop: results unchanged
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q[(size_t)(k + 4 * u) * CP];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
    }
    for (; k < chunks; k += 4) {
        const float2 v = q[(size_t)k * CP];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    __shared__ double red[2][4][64];
    red[0][part][c] = s1;
    red[1][part][c] = s2;
    __syncthreads();
    const bool lead = part == 0;     // wave 0 finishes; the others only keep the barrier below company
    s1 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    s2 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    const double n = (double)HW;
    const float rm = p.rm[h][c], rv = p.rv[h][c];
    const double mean = (double)rm + s1 / n;
    const double var = (s2 - s1 * s1 / n) / (n - 1.0);          // unbiased (torch.var_mean)
    const float sstat = (float)(mean / sqrt(var + 1e-3));
    __shared__ float y[NUM_AFFINE];
    if (lead) {
        for (int k = 0; k < NUM_AFFINE; ++k) {
            float v = sstat * p.att_w[h][k * HEAD_CH + c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (c == 0) {
                const float t = v * p.att_scale[h][k] + p.att_shift[h][k];
                y[k] = fminf(fmaxf(t + 3.f, 0.f), 6.f) / 6.f;
            }
        }
    }
    __syncthreads();
    if (!lead) return;
    float gam = 0.f, bet = 0.f;
#pragma unroll
    for (int k = 0; k < NUM_AFFINE; ++k) {
        gam = fmaf(y[k], p.weight_[h][k * HEAD_CH + c], gam);
        bet = fmaf(y[k], p.bias_[h][k * HEAD_CH + c], bet);
    }
    const float inv = 1.0f / sqrtf(rv + 1e-3f);
    const float sc = gam * inv;
    const size_t o = ((size_t)b * NUM_HEADS + h) * HEAD_CH + c;
    scale[o] = sc;
    shift[o] = bet - rm * sc;
}
hipError_t launch_head_attn(const float *stats, int B, int chunks, int HW, const HeadAttnParams &p, float *scale,
                            float *shift, hipStream_t st) {
    hipLaunchKernelGGL(head_attn_kernel, dim3(B, NUM_HEADS), dim3(256), 0, st, stats, chunks, HW, p, scale, shift);
    return hipGetLastError();
}

// Second head pass: AttnBN apply + ReLU + 1x1 conv + output non-linearity, NCHW stores.
// One wave per (64-pixel tile, head): the [64 px][64 ch] hidden block is normalised while being
// staged into LDS, then lane = pixel accumulates the head's output rows against wave-uniform
// weights (scalar loads of the transposed [64][65] panel).  The head -> (prediction tensor,
// channel, epilogue) routing is a compile-time table so that no kernel-argument array is ever
// indexed dynamically (that would be demoted to scratch memory).
// (reference monocon_heads.py:114-120,165-200)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// head index -> first row, row count, and per-row destination; dir_feat (head 8) feeds two tensors
template <int H> struct HeadMap;
template <> struct HeadMap<0> { static constexpr int RB = 0, NR = 3; };
template <> struct HeadMap<1> { static constexpr int RB = 3, NR = 2; };
template <> struct HeadMap<2> { static constexpr int RB = 5, NR = 2; };
template <> struct HeadMap<3> { static constexpr int RB = 7, NR = 18; };
template <> struct HeadMap<4> { static constexpr int RB = 25, NR = 9; };
template <> struct HeadMap<5> { static constexpr int RB = 34, NR = 2; };
template <> struct HeadMap<6> { static constexpr int RB = 36, NR = 3; };
template <> struct HeadMap<7> { static constexpr int RB = 39, NR = 2; };
template <> struct HeadMap<8> { static constexpr int RB = 41, NR = 24; };

template <int H>
__device__ __forceinline__ constexpr int row_pred(int r) {
    constexpr int hp[8] = {0, 2, 3, 5, 1, 4, 6, 7};
    return H < 8 ? hp[H < 8 ? H : 0] : (r < 12 ? 8 : 9);
}
template <int H>
__device__ __forceinline__ constexpr int row_ch(int r) { return H < 8 ? r : (r < 12 ? r : r - 12); }
template <int H>
__device__ __forceinline__ constexpr int row_epi(int r) {
    return (H == 0 || H == 4) ? 1 : ((H == 7 && r == 0) ? 2 : 0);
}

// The 64 channels of a tile pass through LDS in two halves of 32 (`stage(half)` writes [64 px][33] and synchronises): half the
// LDS per wave, twice the workgroups per CU -- with one tile per wave and no loads in flight during the dot products, the
// memory pipe is kept busy by the OTHER waves (round 5: 3 workgroups per CU by LDS, 3.7 TB/s).  Same fma order as before.
template <int H, class Stage>
__device__ __forceinline__ void head_rows_compute(const HeadApplyArgs &a, const float *hl, int lane, int b, int hw,
                                                  bool ok, Stage stage) {
    constexpr int NR = HeadMap<H>::NR, RB = HeadMap<H>::RB;
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.f;
    // transposed weights [64][65]: rows of one head are contiguous.  Read through the constant
    // address space so the wave-uniform loads go down the scalar path (s_load) instead of VMEM.
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat *w = (cfloat *)(uintptr_t)(a.w + RB);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        stage(half);
#pragma unroll 4
        for (int c = 0; c < HEAD_CH / 2; ++c) {
            const float v = hl[lane * 33 + c];
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = fmaf(v, w[(half * (HEAD_CH / 2) + c) * NUM_OUT_ROWS + r], acc[r]);
        }
    }
    if (!ok) return;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        constexpr int dummy = 0;
        (void)dummy;
        float v = acc[r] + a.b[RB + r];
        const int epi = row_epi<H>(r);
        if (epi == 1) {
            v = fminf(fmaxf(sigmoidf_(v), 1e-4f), 1.0f - 1e-4f);
        } else if (epi == 2) {
            v = 1.0f / (sigmoidf_(v) + 1e-12f) - 1.0f;
        }
        const int p = row_pred<H>(r);
        a.pred[p][((size_t)b * a.pred_c[p] + row_ch<H>(r)) * a.HW + hw] = v;
    }
}

__global__ __launch_bounds__(192) void head_apply_kernel(const HeadApplyArgs a) {
    __shared__ float hlds[3][64 * 33];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.y * 3 + wave;
    const int tiles = (a.HW + 63) / 64;
    const int b = blockIdx.x / tiles, hw0 = (blockIdx.x % tiles) * 64;
    float *hl = hlds[wave];
    // stage + normalise: 16 lanes cover one pixel's 64 channels (float4 each)
    const int c4 = lane & 15;
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + ((size_t)b * NUM_HEADS + h) * HEAD_CH + c4 * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.shift + ((size_t)b * NUM_HEADS + h) * HEAD_CH + c4 * 4);
    f32x4 v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {      // all 16 loads in flight before the first use
        const int hw = hw0 + it * 4 + (lane >> 4);
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hw < a.HW)
            v[it] = *reinterpret_cast<const f32x4 *>(a.hidden + ((size_t)b * a.HW + hw) * (NUM_HEADS * HEAD_CH) +
                                                     h * HEAD_CH + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int px = it * 4 + (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[it][j] = fmaxf(fmaf(v[it][j], sc[j], sh[j]), 0.f);
        if (a.z_out && hw0 + px < a.HW)
            *reinterpret_cast<f32x4 *>(a.z_out + ((size_t)b * a.HW + hw0 + px) * (NUM_HEADS * HEAD_CH) + h * HEAD_CH + c4 * 4) = v[it];
    }
    // channel half `half` of the normalised tile -> LDS: the lanes holding it are c4 in [8 * half, 8 * half + 8)
    // hl = hlds[wave] is PRIVATE to the wave, and `stage` runs inside the per-head switch below: a workgroup barrier there
    // would be a barrier in wave-divergent control flow.  A wave's own LDS accesses execute in order; what is
    // needed is that the compiler keeps them in order: a wavefront-scope fence pair around a wave barrier.
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto stage = [&](int half) {
        if (half) wave_sync();                 // the first half is no longer read
        if ((c4 >> 3) == half) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int px = it * 4 + (lane >> 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) hl[px * 33 + (c4 & 7) * 4 + j] = v[it][j];
            }
        }
        wave_sync();
    };
    const int hw = hw0 + lane;
    const bool ok = hw < a.HW;
    switch (h) {
        case 0: head_rows_compute<0>(a, hl, lane, b, hw, ok, stage); break;
        case 1: head_rows_compute<1>(a, hl, lane, b, hw, ok, stage); break;
        case 2: head_rows_compute<2>(a, hl, lane, b, hw, ok, stage); break;
        case 3: head_rows_compute<3>(a, hl, lane, b, hw, ok, stage); break;
        case 4: head_rows_compute<4>(a, hl, lane, b, hw, ok, stage); break;
        case 5: head_rows_compute<5>(a, hl, lane, b, hw, ok, stage); break;
        case 6: head_rows_compute<6>(a, hl, lane, b, hw, ok, stage); break;
        case 7: head_rows_compute<7>(a, hl, lane, b, hw, ok, stage); break;
        default: head_rows_compute<8>(a, hl, lane, b, hw, ok, stage); break;
    }
}

hipError_t launch_head_apply(const HeadApplyArgs &a, hipStream_t st) {
    init_rows();
    const int tiles = (a.HW + 63) / 64;
    hipLaunchKernelGGL(head_apply_kernel, dim3(a.B * tiles, 3), dim3(192), 0, st, a);
    return hipGetLastError();
}

// ============================================================================ input pipeline
// Normalize + Pad + ToTensor of the reference's default transforms (transforms/default_transforms.py:375-433,
// 436-452) for one image already in HBM: HWC (uint8 or float32) -> CHW float32, (x - mean[c]) / std[c] evaluated
// in float64 and rounded once (numpy broadcasts the float64 mean / std arrays, torch.Tensor() then rounds to
// float32), zero padding to (Hp, Wp).  One thread per output pixel; HBM-bound (3 B or 12 B in, 12 B out).
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T *__restrict__ in, int H, int W, double m0, double m1,
                                                         double m2, double s0, double s1, double s2, int Hp, int Wp,
                                                         float *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wp) return;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (y < H && x < W) {
        const T *p = in + ((size_t)y * W + x) * 3;
        v0 = (float)(((double)p[0] - m0) / s0);
        v1 = (float)(((double)p[1] - m1) / s1);
        v2 = (float)(((double)p[2] - m2) / s2);
    }
    const size_t plane = (size_t)Hp * Wp, o = (size_t)y * Wp + x;
    out[o] = v0;
    out[plane + o] = v1;
    out[2 * plane + o] = v2;
}
// The random train augmentations' IMAGE work (transforms/augmentations.py of this package = the reference's
// transforms/default_transforms.py:27-372 PhotometricDistortion, RandomShift, RandomHorizontalFlip, RandomCrop3D without cv2)
// + Normalize + Pad + ToTensor, for a whole batch of raw uint8 frames in one launch.  The loader's workers draw the random
// numbers, move the labels and the calibration, and ship the decoded frame with 24 numbers; what costs them ~80 ms per frame
// (the float32 HSV round trip over 1.4 M values) is 12 bytes in / 12 bytes out per pixel here.
//   * geometry is an index map: out(y, x) = inside the crop window ? in(y - sy, (flipped ? W - 1 - x : x) - sx) : 0, and 0
//     wherever the source falls outside the frame (the shifted canvas is zero-filled BEFORE Normalize: those pixels come out as
//     -mean / std, the Pad region as 0);
//   * the colour chain is the numpy one, float32 operation by operation (every product and sum rounded on its own: no fma
//     contraction), so the frame is bit-identical to the host pipeline's (tests/test_augment_device.py).
// prm (24 floats per image; integers are exact in float32): 0 H, 1 W, 2 flags, 3 brightness delta, 4 contrast before the HSV
// stage, 5 saturation, 6 hue delta, 7 contrast after it, 8-10 channel permutation (on BGR), 11 sx, 12 sy, 13-16 window
// x0 y0 x1 y1.  flags: 1 colour stage present, 2 brightness, 4 contrast before, 8 saturation, 16 hue, 32 contrast after,
// 64 permutation, 128 shift, 256 flip, 512 window.
namespace {
// float32 division through float64: the float64 quotient of two floats rounds to what a correctly rounded float32 division
// gives (53 >= 2 * 24 + 2 bits) -- independent of how the compiler is told to divide floats.
__device__ __forceinline__ float div_rn(float a, float b) { return (float)((double)a / (double)b); }
__device__ __forceinline__ void aug_colour(float &c0, float &c1, float &c2, const float *__restrict__ prm, int flags) {
// numpy rounds every product and every sum: no a * b + c may become an fma here.  (HIP's __fmul_rn / __fadd_rn are plain
// operators and do not stop the contraction -- a first version built on them was 5 ulp off in 2 % of the pixels.)
#pragma clang fp contract(off)
    float b = c2, g = c1, r = c0;                                   // RGB -> BGR
    if (flags & 2) { const float d = prm[3]; b = b + d; g = g + d; r = r + d; }
    if (flags & 4) { const float a = prm[4]; b = b * a; g = g * a; r = r * a; }
    const float eps = 1.1920928955078125e-07f;
    float v = fmaxf(fmaxf(b, g), r);
    const float diff = v - fminf(fminf(b, g), r);
    float s = div_rn(diff, fabsf(v) + eps);
    const float k = div_rn(60.0f, diff + eps);
    float h;
    if (v == r) { h = (g - b) * k; }
    else if (v == g) { const float t0 = (b - r) * k; h = t0 + 120.0f; }
    else { const float t0 = (r - g) * k; h = t0 + 240.0f; }
    if (h < 0.f) h = h + 360.0f;
    if (flags & 8) s = s * prm[5];
    if (flags & 16) {
        float hue = h + prm[6];
        if (hue > 360.f) hue = hue - 360.f;
        h = hue < 0.f ? hue + 360.f : hue;
    }
    const float h6 = div_rn(h, 60.0f);
    float m = fmodf(h6, 6.0f);                                      // numpy's mod: the sign of the divisor
    if (m != 0.f) { if (m < 0.f) m = m + 6.0f; } else m = 0.f;
    const float sec = floorf(m), f = m - sec;
    const int sector = ((int)sec) % 6;
    const float one_s = 1.0f - s, sf = s * f, one_f = 1.0f - f;
    const float one_sf = 1.0f - sf, s1f = s * one_f;
    const float one_s1f = 1.0f - s1f;
    const float pp = v * one_s, q = v * one_sf, t = v * one_s1f;
    switch (sector) {                                               // (b, g, r) per sector
        case 0: b = pp; g = t; r = v; break;
        case 1: b = pp; g = v; r = q; break;
        case 2: b = t; g = v; r = pp; break;
        case 3: b = v; g = q; r = pp; break;
        case 4: b = v; g = pp; r = t; break;
        default: b = q; g = pp; r = v; break;
    }
    if (s == 0.f) { b = v; g = v; r = v; }
    if (flags & 32) { const float a = prm[7]; b = b * a; g = g * a; r = r * a; }
    if (flags & 64) {
        const float src[3] = {b, g, r};
        const int p0 = (int)prm[8], p1 = (int)prm[9], p2 = (int)prm[10];
        b = p0 == 0 ? src[0] : (p0 == 1 ? src[1] : src[2]);
        g = p1 == 0 ? src[0] : (p1 == 1 ? src[1] : src[2]);
        r = p2 == 0 ? src[0] : (p2 == 1 ? src[1] : src[2]);
    }
    c0 = r; c1 = g; c2 = b;                                         // BGR -> RGB
}
}  // namespace

__global__ __launch_bounds__(256) void preprocess_aug_kernel(const unsigned char *__restrict__ in, const float *__restrict__ prm_all,
                                                             int Hs, int Ws, double m0, double m1, double m2, double s0, double s1,
                                                             double s2, int Hp, int Wp, float *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
    if (x >= Wp) return;
    const float *prm = prm_all + (size_t)img * 24;
    const int H = (int)prm[0], W = (int)prm[1], flags = (int)prm[2];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (y < H && x < W) {
        bool live = true;
        if (flags & 512) live = x >= (int)prm[13] && x < (int)prm[15] && y >= (int)prm[14] && y < (int)prm[16];
        int xs = (flags & 256) ? W - 1 - x : x, ys = y;
        if (flags & 128) { xs -= (int)prm[11]; ys -= (int)prm[12]; }
        live = live && xs >= 0 && xs < W && ys >= 0 && ys < H;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (live) {
            const unsigned char *p = in + (((size_t)img * Hs + ys) * Ws + xs) * 3;
            c0 = (float)p[0]; c1 = (float)p[1]; c2 = (float)p[2];
            if (flags & 1) aug_colour(c0, c1, c2, prm, flags);
        }
        v0 = (float)(((double)c0 - m0) / s0);
        v1 = (float)(((double)c1 - m1) / s1);
        v2 = (float)(((double)c2 - m2) / s2);
    }
    const size_t plane = (size_t)Hp * Wp, o = (size_t)img * 3 * plane + (size_t)y * Wp + x;
    out[o] = v0;
    out[plane + o] = v1;
    out[2 * plane + o] = v2;
}
hipError_t launch_preprocess_aug(const unsigned char *frames, const float *prm, int B, int Hs, int Ws, const double mean[3],
                                 const double std[3], int Hp, int Wp, float *out, hipStream_t st) {
    hipLaunchKernelGGL(preprocess_aug_kernel, dim3((Wp + 255) / 256, Hp, B), dim3(256), 0, st, frames, prm, Hs, Ws, mean[0], mean[1],
                       mean[2], std[0], std[1], std[2], Hp, Wp, out);
    return hipGetLastError();
}

hipError_t launch_preprocess(const void *img_hwc, int is_u8, int H, int W, const double mean[3], const double std[3], int Hp,
                             int Wp, float *out_chw, hipStream_t st) {
    dim3 grid((Wp + 255) / 256, Hp);
    if (is_u8)
        hipLaunchKernelGGL(preprocess_kernel<unsigned char>, grid, dim3(256), 0, st, static_cast<const unsigned char *>(img_hwc),
                           H, W, mean[0], mean[1], mean[2], std[0], std[1], std[2], Hp, Wp, out_chw);
    else
        hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, st, static_cast<const float *>(img_hwc), H, W, mean[0],
                           mean[1], mean[2], std[0], std[1], std[2], Hp, Wp, out_chw);
    return hipGetLastError();
}

}  // namespace mc
