// Precision mode 4: the element-wise passes that PRODUCE and CONSUME P16 tensors (p16.h) in the train plan.
//
// A thread owns one channel OCTET (8 channels = the 32-byte [h0..h7][l0..l7] group of a pixel): both pieces move as
// 16-byte accesses, and a pass that rewrites a tensor in place (dY over dZ) touches exactly the bytes it read.
// Every producer chooses its tensor's exponent BEFORE writing, from a sound bound of max |x| built out of quantities an
// earlier launch left behind (max-|x| slots of the conv outputs, the coefficient vectors of the BatchNorm finalise
// kernels) -- every workgroup derives the same value, workgroup 0 publishes it for the consumers.
//   BatchNorm apply   z = act(a_c y + b_c (+ res))     |z|  <= max|a| max|y| + max|b| (+ 2^(16 - e_res))
//   BatchNorm backward dy = P_c d + Q_c y + R_c        |dy| <= max|P| max|d| + max|Q| max|y| + max|R|
//   2x2 max-pool      shares its input's exponent (a selection)
//   depthwise deconv  out = sum of <= 4 products       |out| <= 2^(16 - e_in) * max_c sum_taps |w|   (bound from the packer)
// Replaces, in mode 4, affine_act / affine_bwd / maxpool2 / deconv4 of kernels_train.hip / kernels_misc.hip (reference:
// nn.BatchNorm2d + ReLU + residual add, nn.MaxPool2d, nn.ConvTranspose2d of model/backbone/dla.py:34-51,178-179 and
// dla_neck.py:58-65, and what autograd builds for them).
#include "conv_mfma.h"
#include "p16.h"
#include "train.h"

namespace mc {

struct f32x8 { float v[8]; };

__device__ __forceinline__ f32x8 ld8(const float *row, int c8) {
    const f32x4 *p = reinterpret_cast<const f32x4 *>(row) + 2 * c8;
    const f32x4 a = p[0], b = p[1];
    return {{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}
__device__ __forceinline__ void st8(float *row, int c8, const f32x8 &v) {
    f32x4 *p = reinterpret_cast<f32x4 *>(row) + 2 * c8;
    p[0] = f32x4{v.v[0], v.v[1], v.v[2], v.v[3]};
    p[1] = f32x4{v.v[4], v.v[5], v.v[6], v.v[7]};
}
__device__ __forceinline__ float &at(f32x8 &v, int j) { return v.v[j]; }
__device__ __forceinline__ float at(const f32x8 &v, int j) { return v.v[j]; }

// P16 octet <-> 8 floats.  `inv` / `s`: exact powers of two.
__device__ __forceinline__ f32x8 p16_ld8(const void *row, int c8, float inv) {
    const f16x8_t *p = reinterpret_cast<const f16x8_t *>(static_cast<const char *>(row) + c8 * 32);
    const f16x8_t h = p[0], l = p[1];
    f32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) at(v, j) = ((float)h[j] + (float)l[j]) * inv;
    return v;
}
__device__ __forceinline__ f16x8_t p16_ld8_hi(const void *row, int c8) {
    return *reinterpret_cast<const f16x8_t *>(static_cast<const char *>(row) + c8 * 32);
}
__device__ __forceinline__ void p16_st8(void *row, int c8, const f32x8 &v, float s) {
    f16x8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = at(v, j) * s;
        h[j] = (_Float16)r;
        l[j] = (_Float16)(r - (float)h[j]);
    }
    f16x8_t *p = reinterpret_cast<f16x8_t *>(static_cast<char *>(row) + c8 * 32);
    p[0] = h; p[1] = l;
}

// max over the workgroup (every thread calls it; values >= 0)
__device__ __forceinline__ float block_max(float v, float *sh /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = sh[0];
    for (int w = 1; w < (int)(blockDim.x + 63) / 64; ++w) m = fmaxf(m, sh[w]);
    return m;
}
__device__ __forceinline__ float slot_max(const unsigned *slot) {      // max |x| a producer left in a slot
    return __builtin_bit_cast(float, amax_read(slot));
}

struct RowSplit8 { int threads, rg, blocks_per_img, rows_per_block; };
static RowSplit8 row_split8(int B, size_t rows_per_img, int C8) {
    RowSplit8 r;
    r.rg = 256 / C8 > 0 ? 256 / C8 : 1;
    r.threads = C8 * r.rg;
    int want = 8192 / (B > 0 ? B : 1);
    if (want < 1) want = 1;
    const size_t min_rows = (size_t)r.rg * 4;
    size_t bpi = (rows_per_img + min_rows - 1) / min_rows;
    if (bpi > (size_t)want) bpi = want;
    if (bpi < 1) bpi = 1;
    r.blocks_per_img = (int)bpi;
    r.rows_per_block = (int)((rows_per_img + bpi - 1) / bpi);
    return r;
}

// ------------------------------------------------------------------ z = act(a*y + b (+ res))  ->  P16
__global__ __launch_bounds__(256) void affine_act_p16_kernel(const float *__restrict__ y, const float *__restrict__ a,
                                                             const float *__restrict__ bb, const void *__restrict__ res16,
                                                             const int *__restrict__ e_res, int C8, int RG, int rows_per_img,
                                                             int blocks_per_img, int rows_per_block, int relu,
                                                             void *__restrict__ z16, int *__restrict__ e_out,
                                                             const unsigned *__restrict__ y_amax) {
    __shared__ float sh[8];
    const int c8 = threadIdx.x % C8, rg = threadIdx.x / C8;
    const int b = blockIdx.x / blocks_per_img, rb = blockIdx.x % blocks_per_img;
    const int r0 = rb * rows_per_block, r1 = min(rows_per_img, r0 + rows_per_block);
    const f32x8 av = ld8(a, c8), bv = ld8(bb, c8);
    float ma = 0.f, mb = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ma = fmaxf(ma, fabsf(at(av, j))); mb = fmaxf(mb, fabsf(at(bv, j))); }
    ma = block_max(ma, sh);
    mb = block_max(mb, sh + 4);
    float rinv = 0.f, bound = ma * slot_max(y_amax) + mb;
    if (res16) { const int er = *e_res; rinv = exp2i(-er); bound += 65536.f * rinv; }
    const int e = p16_exp_of_bound(bound);
    const float s = exp2i(e);
    if (blockIdx.x == 0 && threadIdx.x == 0) *e_out = e;
    const float fl = relu ? 0.f : -__builtin_inff();
    const size_t rowb = (size_t)C8 * 32;
    const size_t img = (size_t)b * rows_per_img;
    constexpr int U = 2;
    int r = r0 + rg;
    for (; r + (U - 1) * RG < r1; r += U * RG) {
        f32x8 v[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld8(y + (img + r + u * RG) * (size_t)C8 * 8, c8);
        if (res16) {
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = p16_ld8(static_cast<const char *>(res16) + (img + r + u * RG) * rowb, c8, rinv);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = fmaf(at(v[u], j), at(av, j), at(bv, j));
                if (res16) t += at(q[u], j);
                at(v[u], j) = fmaxf(t, fl);
            }
            p16_st8(static_cast<char *>(z16) + (img + r + u * RG) * rowb, c8, v[u], s);
        }
    }
    for (; r < r1; r += RG) {
        f32x8 v = ld8(y + (img + r) * (size_t)C8 * 8, c8);
        f32x8 q{};
        if (res16) q = p16_ld8(static_cast<const char *>(res16) + (img + r) * rowb, c8, rinv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = fmaf(at(v, j), at(av, j), at(bv, j));
            if (res16) t += at(q, j);
            at(v, j) = fmaxf(t, fl);
        }
        p16_st8(static_cast<char *>(z16) + (img + r) * rowb, c8, v, s);
    }
}
hipError_t launch_affine_act_p16(const float *y, const float *a, const float *b, const void *res16, const int *e_res, int B,
                                 size_t rows_per_img, int C, int relu, void *z16, int *e_out, const unsigned *y_amax, hipStream_t st) {
    if (C % 8 || C / 8 > 256 || !e_out || !y_amax || (res16 && !e_res)) return hipErrorInvalidValue;
    const RowSplit8 rs = row_split8(B, rows_per_img, C / 8);
    hipLaunchKernelGGL(affine_act_p16_kernel, dim3(B * rs.blocks_per_img), dim3(rs.threads), 0, st, y, a, b, res16, e_res, C / 8, rs.rg,
                       (int)rows_per_img, rs.blocks_per_img, rs.rows_per_block, relu, z16, e_out, y_amax);
    return hipGetLastError();
}

// ------------------------------------------------------------------ dy = P*d + Q*y + R,  d = relu ? dz*[z>0] : dz  ->  P16
// dy16 may alias dz (the train plan writes dY in place over dZ): a thread reads and then rewrites the 32 bytes of ITS octet.
// relu: 0 none (d already masked), 1 mask from the stored activation z16 (hi piece > 0; an activation below 2^-40 of its
// tensor's bound counts as zero), 2 mask recomputed from y with the forward coefficients.  gres_mode as affine_bwd_kernel.
__global__ __launch_bounds__(256) void affine_bwd_p16_kernel(const float *dz, const void *__restrict__ z16, const float *__restrict__ y,
                                                             const float *__restrict__ coef, int C8, int RG, int rows_per_img,
                                                             int blocks_per_img, int rows_per_block, int relu, void *dy16,
                                                             float *__restrict__ gres, int gres_mode, const float *__restrict__ fa,
                                                             const float *__restrict__ fb, int *__restrict__ e_out,
                                                             const unsigned *__restrict__ d_amax, const unsigned *__restrict__ y_amax) {
    __shared__ float sh[12];
    const int c8 = threadIdx.x % C8, rg = threadIdx.x / C8;
    const int b = blockIdx.x / blocks_per_img, rb = blockIdx.x % blocks_per_img;
    const int r0 = rb * rows_per_block, r1 = min(rows_per_img, r0 + rows_per_block);
    float cp[8], cq[8], cr[8];
    float mp = 0.f, mq = 0.f, mr = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x4 cf = reinterpret_cast<const f32x4 *>(coef)[c8 * 8 + j];
        cp[j] = cf[0]; cq[j] = cf[1]; cr[j] = cf[2];
        mp = fmaxf(mp, fabsf(cf[0])); mq = fmaxf(mq, fabsf(cf[1])); mr = fmaxf(mr, fabsf(cf[2]));
    }
    mp = block_max(mp, sh); mq = block_max(mq, sh + 4); mr = block_max(mr, sh + 8);
    const float bound = mp * slot_max(d_amax) + mq * slot_max(y_amax) + mr;
    const int e = p16_exp_of_bound(bound);
    const float s = exp2i(e);
    if (blockIdx.x == 0 && threadIdx.x == 0) *e_out = e;
    f32x8 ma{}, mb{};
    if (relu == 2) { ma = ld8(fa, c8); mb = ld8(fb, c8); }
    const size_t rowb = (size_t)C8 * 32, rowf = (size_t)C8 * 8;
    const size_t img = (size_t)b * rows_per_img;
    for (int r = r0 + rg; r < r1; r += RG) {
        const size_t row = img + r;
        f32x8 d = ld8(dz + row * rowf, c8);
        const f32x8 yv = ld8(y + row * rowf, c8);
        if (relu == 1) {
            const f16x8_t zh = p16_ld8_hi(static_cast<const char *>(z16) + row * rowb, c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) at(d, j) = (float)zh[j] > 0.f ? at(d, j) : 0.f;
        } else if (relu == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) at(d, j) = fmaf(at(yv, j), at(ma, j), at(mb, j)) > 0.f ? at(d, j) : 0.f;
        }
        f32x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) at(o, j) = fmaf(cp[j], at(d, j), fmaf(cq[j], at(yv, j), cr[j]));
        if (gres_mode == 1) st8(gres + row * rowf, c8, d);
        else if (gres_mode == 2) {
            f32x8 gv = ld8(gres + row * rowf, c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) at(gv, j) += at(d, j);
            st8(gres + row * rowf, c8, gv);
        }
        p16_st8(static_cast<char *>(dy16) + row * rowb, c8, o, s);
    }
}
hipError_t launch_affine_bwd_p16(const float *dz, const void *z16, const float *y, const float *coef, int B, size_t rows_per_img, int C,
                                 int relu, void *dy16, float *gres, int gres_mode, const float *fa, const float *fb, int *e_out,
                                 const unsigned *d_amax, const unsigned *y_amax, hipStream_t st) {
    if (C % 8 || C / 8 > 256 || !e_out || !d_amax || !y_amax) return hipErrorInvalidValue;
    if ((relu == 2 && (!fa || !fb)) || (relu == 1 && !z16)) return hipErrorInvalidValue;
    const RowSplit8 rs = row_split8(B, rows_per_img, C / 8);
    hipLaunchKernelGGL(affine_bwd_p16_kernel, dim3(B * rs.blocks_per_img), dim3(rs.threads), 0, st, dz, z16, y, coef, C / 8, rs.rg,
                       (int)rows_per_img, rs.blocks_per_img, rs.rows_per_block, relu, dy16, gres, gres_mode, fa, fb, e_out, d_amax, y_amax);
    return hipGetLastError();
}

// ------------------------------------------------------------------ 2x2 max-pool on P16 (same exponent in and out)
static inline int grid_for16(size_t total, int bs) {
    size_t g = (total + bs - 1) / bs;
    return (int)(g > 16384 ? 16384 : (g == 0 ? 1 : g));
}
__global__ void maxpool2_p16_kernel(const char *__restrict__ in, int B, int H, int W, int C8, char *__restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C8, rowb = (size_t)C8 * 32;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = e % C8;
        const size_t p = e / C8;
        const int x = p % Wo, yy = (p / Wo) % Ho;
        const size_t b = p / ((size_t)Wo * Ho);
        const char *r0 = in + ((b * H + 2 * yy) * W + 2 * x) * rowb;
        const char *r1 = r0 + (size_t)W * rowb;
        const f32x8 a = p16_ld8(r0, c, 1.f), bq = p16_ld8(r0 + rowb, c, 1.f), cq = p16_ld8(r1, c, 1.f), d = p16_ld8(r1 + rowb, c, 1.f);
        f32x8 m;
#pragma unroll
        for (int j = 0; j < 8; ++j) at(m, j) = fmaxf(fmaxf(at(a, j), at(bq, j)), fmaxf(at(cq, j), at(d, j)));
        p16_st8(out + p * rowb, c, m, 1.f);      // (re-splitting hi + lo reproduces the value exactly)
    }
}
hipError_t launch_maxpool2_p16(const void *in16, int B, int H, int W, int C, void *out16, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_p16_kernel, dim3(grid_for16(total, 256)), dim3(256), 0, st, static_cast<const char *>(in16), B, H, W,
                       C / 8, static_cast<char *>(out16));
    return hipGetLastError();
}
// backward: x stored as P16 (only the ordering of the four candidates matters: no exponent needed), dout / dx fp32
__global__ void maxpool2_bwd_p16_kernel(const char *__restrict__ x16, const float *__restrict__ dout, int B, int H, int W, int C8,
                                        float *__restrict__ dx, int accumulate) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C8, rowb = (size_t)C8 * 32, rowf = (size_t)C8 * 8;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = e % C8;
        const size_t p = e / C8;
        const int ox = p % Wo, oy = (p / Wo) % Ho;
        const size_t b = p / ((size_t)Wo * Ho);
        const size_t i00 = (b * H + 2 * oy) * W + 2 * ox, i01 = i00 + 1, i10 = i00 + W, i11 = i10 + 1;
        const f32x8 v00 = p16_ld8(x16 + i00 * rowb, c, 1.f), v01 = p16_ld8(x16 + i01 * rowb, c, 1.f);
        const f32x8 v10 = p16_ld8(x16 + i10 * rowb, c, 1.f), v11 = p16_ld8(x16 + i11 * rowb, c, 1.f);
        const f32x8 g = ld8(dout + p * rowf, c);
        f32x8 g00, g01, g10, g11;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float m = fmaxf(fmaxf(at(v00, j), at(v01, j)), fmaxf(at(v10, j), at(v11, j)));
            const int k = at(v00, j) == m ? 0 : (at(v01, j) == m ? 1 : (at(v10, j) == m ? 2 : 3));
            at(g00, j) = k == 0 ? at(g, j) : 0.f; at(g01, j) = k == 1 ? at(g, j) : 0.f;
            at(g10, j) = k == 2 ? at(g, j) : 0.f; at(g11, j) = k == 3 ? at(g, j) : 0.f;
        }
        auto put = [&](size_t i, const f32x8 &gv) {
            if (accumulate) {
                f32x8 t = ld8(dx + i * rowf, c);
#pragma unroll
                for (int j = 0; j < 8; ++j) at(t, j) += at(gv, j);
                st8(dx + i * rowf, c, t);
            } else {
                st8(dx + i * rowf, c, gv);
            }
        };
        put(i00, g00); put(i01, g01); put(i10, g10); put(i11, g11);
    }
}
hipError_t launch_maxpool2_bwd_p16(const void *x16, const float *dout, int B, int H, int W, int C, float *dx, int accumulate, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_bwd_p16_kernel, dim3(grid_for16(total, 256)), dim3(256), 0, st, static_cast<const char *>(x16), dout, B, H, W,
                       C / 8, dx, accumulate);
    return hipGetLastError();
}

// ------------------------------------------------------------------ depthwise ConvTranspose2d(k=4, s=2, p=1) on P16
// wbound: device float, max over the channels of sum_taps |w| (>= the sum over the <= 4 taps that meet in one output)
__global__ void deconv4_p16_kernel(const char *__restrict__ in16, const int *__restrict__ e_in, int B, int H, int W, int C8,
                                   const float *__restrict__ wpk, const float *__restrict__ wbound, char *__restrict__ out16,
                                   int *__restrict__ e_out) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)B * Ho * Wo * C8, rowb = (size_t)C8 * 32;
    const int ei = *e_in;
    const int eo = p16_exp_of_bound(65536.f * exp2i(-ei) * *wbound);
    const float s = exp2i(eo - ei > 100 ? 100 : (eo - ei < -100 ? -100 : eo - ei));      // (scaled domain of the input -> of the output)
    if (blockIdx.x == 0 && threadIdx.x == 0) *e_out = eo;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = e % C8;
        const size_t p = e / C8;
        const int ox = p % Wo, oy = (p / Wo) % Ho;
        const size_t b = p / ((size_t)Wo * Ho);
        const int iy1 = (oy + 1) >> 1, ky1 = oy + 1 - 2 * iy1;
        const int ix1 = (ox + 1) >> 1, kx1 = ox + 1 - 2 * ix1;
        f32x8 acc{};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int iy = iy1 - dy, ky = ky1 + 2 * dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ix = ix1 - dx, kx = kx1 + 2 * dx;
                if (ix < 0 || ix >= W) continue;
                const f32x8 v = p16_ld8(in16 + ((b * H + iy) * W + ix) * rowb, c, 1.f);
                const f32x8 w = ld8(wpk + (size_t)(ky * 4 + kx) * C8 * 8, c);
#pragma unroll
                for (int j = 0; j < 8; ++j) at(acc, j) = fmaf(at(v, j), at(w, j), at(acc, j));
            }
        }
        p16_st8(out16 + p * rowb, c, acc, s);
    }
}
hipError_t launch_deconv4_p16(const void *in16, const int *e_in, int B, int H, int W, int C, const float *wpk, const float *wbound,
                              void *out16, int *e_out, hipStream_t st) {
    if (C % 8 || !e_in || !e_out || !wbound) return hipErrorInvalidValue;
    const size_t total = (size_t)B * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(deconv4_p16_kernel, dim3(grid_for16(total, 256)), dim3(256), 0, st, static_cast<const char *>(in16), e_in, B, H, W,
                       C / 8, wpk, wbound, static_cast<char *>(out16), e_out);
    return hipGetLastError();
}
// max_c sum_taps |w| of a packed deconv weight [16][C] -> *out (one small launch per layer at pack time)
__global__ __launch_bounds__(256) void deconv_wbound_kernel(const float *__restrict__ wpk, int C, float *__restrict__ out) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float sacc = 0.f;
        for (int k = 0; k < 16; ++k) sacc += fabsf(wpk[k * C + c]);
        m = fmaxf(m, sacc);
    }
    m = block_max(m, sh);
    if (threadIdx.x == 0) *out = m;
}
hipError_t launch_deconv_wbound(const float *wpk, int C, float *out, hipStream_t st) {
    hipLaunchKernelGGL(deconv_wbound_kernel, dim3(1), dim3(256), 0, st, wpk, C, out);
    return hipGetLastError();
}

}  // namespace mc
