// Train-mode element-wise / reduction kernels (NHWC fp32, 16-byte accesses, all HBM-bound):
// BatchNorm batch statistics + running-stat update, normalise(+residual)+ReLU, the masked
// reductions and the affine back-substitution of BatchNorm / AttnBN backward, max-pool and
// depthwise-deconv backward, layout packing.
//
// Replaces the autograd graph torch builds for nn.BatchNorm2d / ReLU / residual add / MaxPool2d /
// ConvTranspose2d in the reference's BasicBlock, Root, Tree, Conv2dBlock, IDAUp
// (model/backbone/dla.py:34-51,124-132,187-205; dla_neck.py:34-38,94-106).
#include "conv_mfma.h"
#include "train.h"
#include <cstdlib>
#include <cstring>

namespace mc {

static inline int grid_for(size_t total, int bs, int cap = 16384) {
    size_t g = (total + bs - 1) / bs;
    return (int)(g > (size_t)cap ? cap : (g == 0 ? 1 : g));
}

// ------------------------------------------------------------------ per-(image,row-block,channel) sums
// mode 0: (sum(y - shift), sum((y - shift)^2));  mode 1: d = relu ? dz*[z>0] : dz -> (sum d, sum d*y)
// rows per workgroup: 256, halved (down to 32) while the launch would not fill the chip
static int red_rows(int B, int rows_per_img) {
    int r = 256;
    while (r > 32 && (long long)B * ((rows_per_img + r - 1) / r) < 1024) r >>= 1;
    return r;
}
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float *__restrict__ y, const float *__restrict__ dz,
                                                          const float *__restrict__ z, const float *__restrict__ shift,
                                                          int rows_per_img, int RED_ROWS, int C, int mode, int relu,
                                                          float *__restrict__ partial, int Cstride,
                                                          const float *__restrict__ fa, const float *__restrict__ fb) {
    const int C4 = C >> 2;
    const int RG = 256 / C4 > 0 ? 256 / C4 : 1;
    const int tid = threadIdx.x;
    const int c4 = tid % C4, rg = tid / C4;
    const int rb_per_img = (rows_per_img + RED_ROWS - 1) / RED_ROWS;
    const int b = blockIdx.x / rb_per_img, rb = blockIdx.x % rb_per_img;
    const int r0 = rb * RED_ROWS, r1 = min(rows_per_img, r0 + RED_ROWS);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const bool active = tid < C4 * RG;
    f32x4 sh = {0.f, 0.f, 0.f, 0.f};
    if (active && mode == 0 && shift) sh = *reinterpret_cast<const f32x4 *>(shift + c4 * 4);
    // relu == 2: the ReLU mask is recomputed from y with the forward coefficients (z = max(fma(y,a,b),0),
    // no residual) instead of reading z -- one pass less over the activation
    f32x4 ma = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
    if (active && relu == 2) { ma = *reinterpret_cast<const f32x4 *>(fa + c4 * 4); mb = *reinterpret_cast<const f32x4 *>(fb + c4 * 4); }
    if (active) {
        auto one = [&](const f32x4 yv, f32x4 d, const f32x4 zv) {
            if (mode == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float t = yv[j] - sh[j]; s1[j] += t; s2[j] += t * t; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (relu == 1) d[j] = zv[j] > 0.f ? d[j] : 0.f;
                    else if (relu == 2) d[j] = fmaf(yv[j], ma[j], mb[j]) > 0.f ? d[j] : 0.f;
                    s1[j] += d[j]; s2[j] += d[j] * yv[j];
                }
            }
        };
        const f32x4 *y4 = reinterpret_cast<const f32x4 *>(y), *d4 = reinterpret_cast<const f32x4 *>(dz);
        const f32x4 *z4 = reinterpret_cast<const f32x4 *>(z);
        const unsigned base = ((unsigned)b * rows_per_img) * C4 + c4, step = (unsigned)RG * C4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        int r = r0 + rg;
        for (; r + 3 * RG < r1; r += 4 * RG) {   // four independent rows in flight
            const unsigned e = base + (unsigned)r * C4;
            f32x4 yv[4], dv[4], zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) yv[u] = y4[e + u * step];
            if (mode != 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) dv[u] = d4[e + u * step];
                if (relu == 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) zv[u] = z4[e + u * step];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(yv[u], mode != 0 ? dv[u] : zero, (mode != 0 && relu == 1) ? zv[u] : zero);
        }
        for (; r < r1; r += RG) {
            const unsigned e = base + (unsigned)r * C4;
            one(y4[e], mode != 0 ? d4[e] : zero, (mode != 0 && relu == 1) ? z4[e] : zero);
        }
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s1[j]; red[tid * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    if (tid < C4) {
        float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1[j] += red[(g * C4 + tid) * 8 + j]; a2[j] += red[(g * C4 + tid) * 8 + 4 + j]; }
        float *dst = partial + ((size_t)blockIdx.x * Cstride + tid * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[j * 2] = a1[j]; dst[j * 2 + 1] = a2[j]; }
    }
}

// Measurement aid, compiled in ONLY with -DMC_DEBUG_HOOKS (never in the shipped library: a stray environment
// variable must not be able to turn launches into silent no-ops): MONOCON_HIP_DEBUG_SKIP=fold,fin,bfin,aact,abwd,cred
// skips the named launches -- results are then WRONG; the step time without a kernel family bounds what fusing it
// away can gain (scratch/skip_bounds.sh builds such a library).
#ifdef MC_DEBUG_HOOKS
static bool dbg_skip(const char *what) {
    static const char *e = std::getenv("MONOCON_HIP_DEBUG_SKIP");
    if (!e) return false;
    const char *p = std::strstr(e, what);
    if (!p) return false;
    const char c = p[std::strlen(what)];
    return (p == e || p[-1] == ',') && (c == 0 || c == ',');
}
#else
static constexpr bool dbg_skip(const char *) { return false; }
#endif
int chan_reduce_blocks(int B, int rows_per_img) {
    const int r = red_rows(B, rows_per_img);
    return B * ((rows_per_img + r - 1) / r);
}
hipError_t launch_chan_reduce(const float *y, const float *dz, const float *z, const float *shift, int B, int rows_per_img,
                              int C, int mode, int relu, float *partial, int Cstride, hipStream_t st, const float *fa,
                              const float *fb) {
    if (C % 4 || C / 4 > 256 || (size_t)B * rows_per_img * (C / 4) >= (1ull << 32)) return hipErrorInvalidValue;
    if (relu == 2 && (!fa || !fb)) return hipErrorInvalidValue;
    if (dbg_skip("cred")) return hipSuccess;
    hipLaunchKernelGGL(chan_reduce_kernel, dim3(chan_reduce_blocks(B, rows_per_img)), dim3(256), 0, st, y, dz, z, shift,
                       rows_per_img, red_rows(B, rows_per_img), C, mode, relu, partial, Cstride, fa, fb);
    return hipGetLastError();
}

// ------------------------------------------------------------------ partial-sum fold
// The conv epilogues leave one (s1, s2) pair per 4x8 patch and channel: tens of thousands of rows for the
// 96x320 maps.  Walking them channel by channel (one workgroup per channel, 8 bytes out of every Cstride*8) uses a
// sliver of each cache line; this pre-pass reads whole rows (thread = channel, consecutive 8-byte pairs) and folds
// FOLD_ROWS of them into one row of doubles, which the finalise kernels then finish.  Fixed order throughout.
// (round 4: 64 rows per workgroup, thread = (row group, channel PAIR) with 16-byte loads, eight rows in flight -- the
//  128-row version walked its rows four 8-byte loads at a time from 60 workgroups: 38 us per launch for 8 MB, 28 launches
//  a step)
constexpr int FOLD_ROWS = 64, FOLD_MIN_NB = 2048;
__global__ __launch_bounds__(256) void partial_fold_kernel(const float *__restrict__ partial, int nb, int Cstride, int C,
                                                           double *__restrict__ out /*[blocks][C][2]*/) {
    __shared__ double red[256][4];
    const int r0 = blockIdx.x * FOLD_ROWS, r1 = min(nb, r0 + FOLD_ROWS);
    const int C2 = C >> 1, S4 = Cstride >> 1;                 // channel pairs; float4s per row (C and Cstride are even)
    const int CG = C2 < 256 ? C2 : 256, RG = 256 / CG;        // thread = (row group, channel pair)
    const int cl = threadIdx.x % CG, rg = threadIdx.x / CG;
    const f32x4 *base = reinterpret_cast<const f32x4 *>(partial);
    for (int c0 = 0; c0 < C2; c0 += CG) {
        const int cp = c0 + cl;
        double s[4] = {0, 0, 0, 0};
        if (rg < RG && cp < C2) {
            const f32x4 *p = base + (size_t)(r0 + rg) * S4 + cp;
            const size_t step = (size_t)RG * S4;
            int r = r0 + rg;
            for (; r + 7 * RG < r1; r += 8 * RG, p += 8 * step) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[u * step];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += v[u][j];
            }
            for (; r < r1; r += RG, p += step) {
                const f32x4 v = *p;
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = s[j];
        __syncthreads();
        if (rg == 0 && cp < C2) {
            for (int g = 1; g < RG; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += red[g * CG + cl][j];
            double *o = out + ((size_t)blockIdx.x * C + 2 * cp) * 2;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
        }
        __syncthreads();
    }
}
size_t partial_fold_doubles(int nb, int C) { return (nb >= FOLD_MIN_NB && C % 2 == 0) ? (size_t)((nb + FOLD_ROWS - 1) / FOLD_ROWS) * C * 2 : 0; }

// ------------------------------------------------------------------ BatchNorm (train) finalise
// partial: [nb][Cstride][2] sums of (y - shift), (y - shift)^2 over n = nb * rows values per channel.
template <typename T>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const T *__restrict__ partial, int nb, int Cstride, double n,
                                                          const float *shift, const float *gamma, const float *beta,
                                                          float eps, float momentum, float *running_mean,
                                                          float *running_var, long long *nbt, float *a_out, float *b_out,
                                                          float *mean_out, float *rstd_out, const unsigned *ymax,
                                                          unsigned *zmax, int zrelu) {
    const int c = blockIdx.x;
    double s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        const T *p = partial + ((size_t)i * Cstride + c) * 2;
        s1 += p[0]; s2 += p[1];
    }
    __shared__ double sh[32];
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((threadIdx.x & 63) == 0) { sh[(threadIdx.x >> 6) * 2] = s1; sh[(threadIdx.x >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = s2 = 0;
        for (int w = 0; w < (int)blockDim.x / 64; ++w) { s1 += sh[w * 2]; s2 += sh[w * 2 + 1]; }
        const double m0 = s1 / n;
        const double mean = (shift ? (double)shift[c] : 0.0) + m0;
        double var = s2 / n - m0 * m0;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        const float a = g * rstd;
        if (a_out) { a_out[c] = a; b_out[c] = be - (float)mean * a; }
        if (zmax && ymax) {
            // the activation z = act(a * y + b) is never stored (a "lazy" tensor, ConvSrc::la): its consumers scale their
            // fp16-split operand by a power of two derived from THIS bound of max |z| -- |a| * max |y| + b (ReLU: only the
            // positive side counts) or + |b| -- from the exact max |y| the producing conv left in `ymax`.  A bound looser than
            // the true maximum by a factor 2^L costs L of the split's 22 bits in absolute terms; measured L <= 1.
            unsigned my = 0u;
            for (int i = 0; i < AMAX_SUB; ++i) { const unsigned v = ymax[i * AMAX_STRIDE]; my = v > my ? v : my; }
            const float b0 = be - (float)mean * a;
            float zb = fabsf(a) * __builtin_bit_cast(float, my) + (zrelu ? b0 : fabsf(b0));
            zb = fmaxf(zb, 0.f);
            const unsigned bits = __builtin_bit_cast(unsigned, zb);
            unsigned *q = zmax + (c % AMAX_SUB) * AMAX_STRIDE;
            if (bits < 0x7f800000u && bits != 0u) atomicMax(q, bits);
        }
        mean_out[c] = (float)mean;
        rstd_out[c] = rstd;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * n / (n - 1.0));
        }
        if (nbt && c == 0) *nbt += 1;
    }
}
hipError_t launch_bn_finalize(const float *partial, int nb, int Cstride, double n, int C, const float *shift,
                              const float *gamma, const float *beta, float eps, float momentum, float *rm, float *rv,
                              long long *nbt, float *a, float *b, float *mean, float *rstd, hipStream_t st, double *fold,
                              const unsigned *ymax, unsigned *zmax, int zrelu) {
    if (dbg_skip("fin")) return hipSuccess;
    if (fold && nb >= FOLD_MIN_NB) {
        const int nb2 = (nb + FOLD_ROWS - 1) / FOLD_ROWS;
        if (!dbg_skip("fold")) hipLaunchKernelGGL(partial_fold_kernel, dim3(nb2), dim3(256), 0, st, partial, nb, Cstride, C, fold);
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(C), dim3(256), 0, st, fold, nb2, C, n, shift, gamma, beta, eps,
                           momentum, rm, rv, nbt, a, b, mean, rstd, ymax, zmax, zrelu);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(C), dim3(nb >= 2048 ? 1024 : 256), 0, st, partial, nb, Cstride, n, shift, gamma, beta, eps,
                       momentum, rm, rv, nbt, a, b, mean, rstd, ymax, zmax, zrelu);
    return hipGetLastError();
}

// ------------------------------------------------------------------ z = act(a*y + b (+ res))
// per_sample: coefficient index = b*C + c (AttnBN) instead of c (BatchNorm)
// Thread layout of the element-wise passes: a thread owns ONE group of 4 channels (its coefficients
// stay in registers) and walks rows of ONE image with 32-bit float4 indices, four independent rows
// in flight -- no per-element div/mod, no coefficient reloads.
struct RowSplit { int threads, rg, blocks_per_img, rows_per_block; };
static RowSplit row_split(int B, size_t rows_per_img, int C4) {
    RowSplit r;
    r.rg = 256 / C4 > 0 ? 256 / C4 : 1;
    r.threads = C4 * r.rg;
    int want = 8192 / (B > 0 ? B : 1);                       // ~8k workgroups per launch
    if (want < 1) want = 1;
    size_t min_rows = (size_t)r.rg * 8;                       // at least 8 rows per thread
    size_t bpi = (rows_per_img + min_rows - 1) / min_rows;
    if (bpi > (size_t)want) bpi = want;
    if (bpi < 1) bpi = 1;
    r.blocks_per_img = (int)bpi;
    r.rows_per_block = (int)((rows_per_img + bpi - 1) / bpi);
    return r;
}

__global__ __launch_bounds__(256) void affine_act_kernel(const f32x4 *__restrict__ y, const float *__restrict__ a,
                                                         const float *__restrict__ bb, const f32x4 *__restrict__ res,
                                                         int C4, int RG, int rows_per_img, int blocks_per_img,
                                                         int rows_per_block, int per_sample, int relu,
                                                         f32x4 *__restrict__ z, unsigned *__restrict__ amax,
                                                         const float *__restrict__ ra, const float *__restrict__ rbc, int rrelu,
                                                         unsigned *__restrict__ zbits) {
    // zbits: also leave the ReLU mask bit-packed, [row][C / 32] words (ConvArgs::bm_zbits; C % 32 == 0: the eight threads
    // of a word are eight neighbouring lanes of one wave)
    // ra / rb: the residual is a LAZY tensor (ConvSrc::la): res holds its producer's raw conv output, its value is
    // act(ra * res + rb), formed here (a Tree's `project` branch, BatchNorm without ReLU: model/backbone/dla.py:181-185,198)
    const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
    const int b = blockIdx.x / blocks_per_img, rb = blockIdx.x % blocks_per_img;
    const int r0 = rb * rows_per_block, r1 = min(rows_per_img, r0 + rows_per_block);
    const int ci = (per_sample ? b * C4 : 0) + c4;
    float vmax = 0.f;            // max |z| of this thread (amax != null: the consumers' fp16-split operand scale)
    const f32x4 av = reinterpret_cast<const f32x4 *>(a)[ci], bv = reinterpret_cast<const f32x4 *>(bb)[ci];
    const float fl = relu ? 0.f : -__builtin_inff();
    f32x4 rav = {1.f, 1.f, 1.f, 1.f}, rbv = {0.f, 0.f, 0.f, 0.f};
    if (ra) { rav = reinterpret_cast<const f32x4 *>(ra)[c4]; rbv = reinterpret_cast<const f32x4 *>(rbc)[c4]; }
    const float rfl = (ra && rrelu) ? 0.f : -__builtin_inff();
    auto lazy_res = [&](f32x4 q) {
        if (ra) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = fmaxf(fmaf(q[j], rav[j], rbv[j]), rfl);
        }
        return q;
    };
    const unsigned base = ((unsigned)b * rows_per_img) * C4 + c4;
    const unsigned step = (unsigned)RG * C4;
    auto put_bits = [&](const f32x4 zv, int row) {      // the word of channels 32 (c4 / 8) .. + 31 of this row
        unsigned m = (zv[0] > 0.f ? 1u : 0u) | (zv[1] > 0.f ? 2u : 0u) | (zv[2] > 0.f ? 4u : 0u) | (zv[3] > 0.f ? 8u : 0u);
        m <<= 4 * (c4 & 7);
        m |= (unsigned)__shfl_xor((int)m, 1);
        m |= (unsigned)__shfl_xor((int)m, 2);
        m |= (unsigned)__shfl_xor((int)m, 4);
        if ((c4 & 7) == 0) zbits[((size_t)b * rows_per_img + row) * (C4 >> 3) + (c4 >> 3)] = m;
    };
    int r = r0 + rg;
    for (; r + 3 * RG < r1; r += 4 * RG) {
        const unsigned e = base + (unsigned)r * C4;
        f32x4 v[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = y[e + u * step];
        if (res) {
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = lazy_res(res[e + u * step]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = fmaf(v[u][j], av[j], bv[j]);
                if (res) t += q[u][j];
                v[u][j] = fmaxf(t, fl);
                vmax = fmaxf(vmax, fabsf(v[u][j]));
            }
            z[e + u * step] = v[u];
            if (zbits) put_bits(v[u], r + u * RG);
        }
    }
    for (; r < r1; r += RG) {
        const unsigned e = base + (unsigned)r * C4;
        f32x4 v = y[e];
        f32x4 q1 = {0.f, 0.f, 0.f, 0.f};
        if (res) q1 = lazy_res(res[e]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = fmaf(v[j], av[j], bv[j]);
            if (res) t += q1[j];
            v[j] = fmaxf(t, fl);
            vmax = fmaxf(vmax, fabsf(v[j]));
        }
        z[e] = v;
        if (zbits) put_bits(v, r);
    }
    if (amax) amax_update_block(amax, vmax);
}
hipError_t launch_affine_act(const float *y, const float *a, const float *b, const float *res, int B, size_t rows_per_img,
                             int C, int per_sample, int relu, float *z, hipStream_t st, unsigned *amax, const float *res_a,
                             const float *res_b, int res_relu, unsigned *zbits) {
    if (C % 4 || C / 4 > 256 || (size_t)B * rows_per_img * (C / 4) >= (1ull << 32)) return hipErrorInvalidValue;
    if (zbits && (C % 32 || !relu)) return hipErrorInvalidValue;
    if ((res_a != nullptr) != (res_b != nullptr) || (res_a && !res)) return hipErrorInvalidValue;
    if (dbg_skip("aact")) return hipSuccess;
    const RowSplit rs = row_split(B, rows_per_img, C / 4);
    hipLaunchKernelGGL(affine_act_kernel, dim3(B * rs.blocks_per_img), dim3(rs.threads), 0, st,
                       reinterpret_cast<const f32x4 *>(y), a, b, reinterpret_cast<const f32x4 *>(res), C / 4, rs.rg,
                       (int)rows_per_img, rs.blocks_per_img, rs.rows_per_block, per_sample, relu, reinterpret_cast<f32x4 *>(z), amax,
                       res_a, res_b, res_relu, zbits);
    return hipGetLastError();
}

// ------------------------------------------------------------------ BatchNorm backward finalise
// partial: [nb][Cstride][2] = (sum d, sum d*y).  dy = P*d + Q*y + R with
//   P = a, Q = -a*r*S2/n, R = -a*S1/n + a*r*mean*S2/n,  S2 = r*(sum d*y - mean*sum d);  dgamma = S2, dbeta = S1.
template <typename T>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const T *__restrict__ partial, int nb, int Cstride,
                                                              double n, const float *gamma, const float *mean,
                                                              const float *rstd, float *dgamma, float *dbeta,
                                                              float *coef /*[C][4]*/) {
    const int c = blockIdx.x;
    double s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        const T *p = partial + ((size_t)i * Cstride + c) * 2;
        s1 += p[0]; s2 += p[1];
    }
    __shared__ double sh[32];
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((threadIdx.x & 63) == 0) { sh[(threadIdx.x >> 6) * 2] = s1; sh[(threadIdx.x >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = s2 = 0;
        for (int w = 0; w < (int)blockDim.x / 64; ++w) { s1 += sh[w * 2]; s2 += sh[w * 2 + 1]; }
        const double r = rstd[c], mu = mean[c], g = gamma ? gamma[c] : 1.0;
        const double S2 = r * (s2 - mu * s1);
        if (dgamma) dgamma[c] = (float)S2;
        if (dbeta) dbeta[c] = (float)s1;
        const double a = g * r;
        coef[c * 4 + 0] = (float)a;
        coef[c * 4 + 1] = (float)(-a * r * S2 / n);
        coef[c * 4 + 2] = (float)(-a * s1 / n + a * r * mu * S2 / n);
        coef[c * 4 + 3] = 0.f;
    }
}
hipError_t launch_bn_bwd_finalize(const float *partial, int nb, int Cstride, double n, int C, const float *gamma,
                                  const float *mean, const float *rstd, float *dgamma, float *dbeta, float *coef,
                                  hipStream_t st, double *fold) {
    if (dbg_skip("bfin")) return hipSuccess;
    if (fold && nb >= FOLD_MIN_NB) {
        const int nb2 = (nb + FOLD_ROWS - 1) / FOLD_ROWS;
        if (!dbg_skip("fold")) hipLaunchKernelGGL(partial_fold_kernel, dim3(nb2), dim3(256), 0, st, partial, nb, Cstride, C, fold);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3(C), dim3(256), 0, st, fold, nb2, C, n, gamma, mean, rstd, dgamma,
                           dbeta, coef);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<float>, dim3(C), dim3(nb >= 2048 ? 1024 : 256), 0, st, partial, nb, Cstride, n, gamma, mean, rstd, dgamma,
                       dbeta, coef);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ partial, int nb, int Cstride,
                                                           float *__restrict__ out);

// ------------------------------------------------------------------ dy = P*d + Q*y + R,  d = relu ? dz*[z>0] : dz
// gres_mode: 0 none, 1 gres = d, 2 gres += d (gradient of the residual input)
// (dz and dy are NOT restrict-qualified: the train plan writes dy in place over dz; every element is read and then
//  written once, by the same thread)
__global__ __launch_bounds__(256) void affine_bwd_kernel(const f32x4 *dz, const f32x4 *__restrict__ z,
                                                         const f32x4 *__restrict__ y, const float *__restrict__ coef,
                                                         int C4, int RG, int rows_per_img, int blocks_per_img,
                                                         int rows_per_block, int per_sample, int relu,
                                                         f32x4 *dy, f32x4 *__restrict__ gres, int gres_mode,
                                                         const float *__restrict__ fa, const float *__restrict__ fb,
                                                         float *__restrict__ csum /*[blocks][4*C4][2] or null*/,
                                                         unsigned *__restrict__ amax /*max |dy| or null*/) {
    __shared__ f32x4 cred[256];
    float vmax = 0.f;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};      // column sums of dy over this thread's rows (conv bias gradient), csum != null
    const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
    const int b = blockIdx.x / blocks_per_img, rb = blockIdx.x % blocks_per_img;
    const int r0 = rb * rows_per_block, r1 = min(rows_per_img, r0 + rows_per_block);
    const int ci = ((per_sample ? b * C4 : 0) + c4) * 4;
    float cp[4], cq[4], cr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 cf = reinterpret_cast<const f32x4 *>(coef)[ci + j];
        cp[j] = cf[0]; cq[j] = cf[1]; cr[j] = cf[2];
    }
    f32x4 ma = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};   // relu == 2: mask from y (see chan_reduce_kernel)
    if (relu == 2) { ma = reinterpret_cast<const f32x4 *>(fa)[c4]; mb = reinterpret_cast<const f32x4 *>(fb)[c4]; }
    const unsigned base = ((unsigned)b * rows_per_img) * C4 + c4;
    const unsigned step = (unsigned)RG * C4;
    auto one = [&](unsigned e, f32x4 d, const f32x4 zv, const f32x4 yv) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (relu == 1) d[j] = zv[j] > 0.f ? d[j] : 0.f;
            else if (relu == 2) d[j] = fmaf(yv[j], ma[j], mb[j]) > 0.f ? d[j] : 0.f;
            o[j] = fmaf(cp[j], d[j], fmaf(cq[j], yv[j], cr[j]));
            cs[j] += o[j];
            vmax = fmaxf(vmax, fabsf(o[j]));
        }
        dy[e] = o;
        if (gres_mode == 1) gres[e] = d;
        else if (gres_mode == 2) {
            f32x4 gv = gres[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] += d[j];
            gres[e] = gv;
        }
    };
    int r = r0 + rg;
    for (; r + 3 * RG < r1; r += 4 * RG) {
        const unsigned e = base + (unsigned)r * C4;
        f32x4 d[4], zv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { d[u] = dz[e + u * step]; yv[u] = y[e + u * step]; }
        if (relu == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) zv[u] = z[e + u * step];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(e + u * step, d[u], zv[u], yv[u]);
    }
    for (; r < r1; r += RG) {
        const unsigned e = base + (unsigned)r * C4;
        f32x4 zv = {1.f, 1.f, 1.f, 1.f};
        if (relu == 1) zv = z[e];
        one(e, dz[e], zv, y[e]);
    }
    if (csum) {       // row groups folded in order through LDS: one (sum, -) pair per block and channel (colsum_final_kernel's format)
        cred[threadIdx.x] = cs;
        __syncthreads();
        if (rg == 0) {
            for (int g = 1; g < RG; ++g) {
                const f32x4 o = cred[g * C4 + c4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cs[j] += o[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[((size_t)blockIdx.x * (4 * C4) + c4 * 4 + j) * 2] = cs[j];
        }
    }
    if (amax) amax_update_block(amax, vmax);
}
int affine_bwd_blocks(int B, size_t rows_per_img, int C) { return B * row_split(B, rows_per_img, C / 4).blocks_per_img; }
hipError_t launch_affine_bwd(const float *dz, const float *z, const float *y, const float *coef, int B, size_t rows_per_img,
                             int C, int per_sample, int relu, float *dy, float *gres, int gres_mode, hipStream_t st,
                             const float *fa, const float *fb, float *csum, float *csum_out, unsigned *amax) {
    if (C % 4 || C / 4 > 256 || (size_t)B * rows_per_img * (C / 4) >= (1ull << 32)) return hipErrorInvalidValue;
    if (relu == 2 && (!fa || !fb || per_sample)) return hipErrorInvalidValue;
    if (dbg_skip("abwd")) return hipSuccess;
    const RowSplit rs = row_split(B, rows_per_img, C / 4);
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(B * rs.blocks_per_img), dim3(rs.threads), 0, st,
                       reinterpret_cast<const f32x4 *>(dz), reinterpret_cast<const f32x4 *>(z),
                       reinterpret_cast<const f32x4 *>(y), coef, C / 4, rs.rg, (int)rows_per_img, rs.blocks_per_img,
                       rs.rows_per_block, per_sample, relu, reinterpret_cast<f32x4 *>(dy), reinterpret_cast<f32x4 *>(gres),
                       gres_mode, fa, fb, csum, amax);
    if (csum && csum_out)     // column sums of dy: the partial rows of the launch above, finished per column
        hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, st, csum, B * rs.blocks_per_img, C, csum_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ small helpers
__global__ void add_kernel(f32x4 *__restrict__ a, const f32x4 *__restrict__ b, size_t n4) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        f32x4 x = a[e];
        const f32x4 yv = b[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] += yv[j];
        a[e] = x;
    }
}
hipError_t launch_add(float *a, const float *b, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, st, reinterpret_cast<f32x4 *>(a),
                       reinterpret_cast<const f32x4 *>(b), n / 4);
    return hipGetLastError();
}

// column sums of a [rows][ld] matrix -> out[C] (conv bias gradients): coalesced row-block partial sums
// (chan_reduce, mode 0) followed by a per-column reduction of the partials
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ partial, int nb, int Cstride,
                                                           float *__restrict__ out) {
    const int c = blockIdx.x;
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[((size_t)i * Cstride + c) * 2];
    __shared__ double sh[4];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}
hipError_t launch_colsum_final(const float *partial, int nb, int C, float *out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, st, partial, nb, C, out);
    return hipGetLastError();
}
size_t colsum_partial_floats(size_t rows, int ld) { return (size_t)chan_reduce_blocks(1, (int)rows) * ld * 2; }
hipError_t launch_colsum(const float *x, size_t rows, int C, int ld, float *partial, float *out, hipStream_t st) {
    hipError_t e = launch_chan_reduce(x, nullptr, nullptr, nullptr, 1, (int)rows, ld, 0, 0, partial, ld, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, st, partial, chan_reduce_blocks(1, (int)rows), ld, out);
    return hipGetLastError();
}

// 2x2/2 max-pool backward: gradient goes to the first maximum in window scan order (torch)
// la / lb: x is a lazy tensor (ConvSrc::la) -- the window is compared on max(fma(y, la, lb), 0), the values the forward pooled.
// STATS (round 6; lazy x, accumulate): this launch COMPLETES the gradient of x, a BatchNorm + ReLU output, so it also does what
// the BatchNorm backward's reduction pass would: it applies the ReLU mask (the formed value > 0) to the total gradient it
// writes and leaves the (sum d, sum d * y) partials per workgroup and channel in `partial` [gridDim.x][C][2] -- the
// three-tensor chan_reduce pass over the map disappears, and the element-wise pass behind it needs no mask.  A thread keeps
// ONE channel quad (the stride of the grid-stride loop is a multiple of C4); a workgroup's threads of a quad are folded in
// thread order.
template <bool STATS>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const f32x4 *__restrict__ x, const f32x4 *__restrict__ dout, int B, int H, int W,
                                    int C4, f32x4 *__restrict__ dx, int accumulate, const f32x4 *__restrict__ la,
                                    const f32x4 *__restrict__ lb, float *__restrict__ partial) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    [[maybe_unused]] f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = e % C4;
        const size_t p = e / C4;
        const int ox = p % Wo, oy = (p / Wo) % Ho;
        const size_t b = p / ((size_t)Wo * Ho);
        const size_t i00 = ((b * H + 2 * oy) * W + 2 * ox) * C4 + c, i01 = i00 + C4, i10 = i00 + (size_t)W * C4, i11 = i10 + C4;
        f32x4 v00 = x[i00], v01 = x[i01], v10 = x[i10], v11 = x[i11];
        const f32x4 g = dout[e];
        [[maybe_unused]] const f32x4 y00 = v00, y01 = v01, y10 = v10, y11 = v11;      // (STATS: the raw conv outputs)
        if (la) {
            const f32x4 av = la[c], bv = lb[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v00[j] = fmaxf(fmaf(v00[j], av[j], bv[j]), 0.f); v01[j] = fmaxf(fmaf(v01[j], av[j], bv[j]), 0.f);
                v10[j] = fmaxf(fmaf(v10[j], av[j], bv[j]), 0.f); v11[j] = fmaxf(fmaf(v11[j], av[j], bv[j]), 0.f);
            }
        }
        f32x4 g00, g01, g10, g11;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = fmaxf(fmaxf(v00[j], v01[j]), fmaxf(v10[j], v11[j]));
            const int k = v00[j] == m ? 0 : (v01[j] == m ? 1 : (v10[j] == m ? 2 : 3));
            g00[j] = k == 0 ? g[j] : 0.f; g01[j] = k == 1 ? g[j] : 0.f;
            g10[j] = k == 2 ? g[j] : 0.f; g11[j] = k == 3 ? g[j] : 0.f;
        }
        if (accumulate) {
            f32x4 t;
            t = dx[i00]; for (int j = 0; j < 4; ++j) g00[j] += t[j];
            t = dx[i01]; for (int j = 0; j < 4; ++j) g01[j] += t[j];
            t = dx[i10]; for (int j = 0; j < 4; ++j) g10[j] += t[j];
            t = dx[i11]; for (int j = 0; j < 4; ++j) g11[j] += t[j];
        }
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g00[j] = v00[j] > 0.f ? g00[j] : 0.f; g01[j] = v01[j] > 0.f ? g01[j] : 0.f;
                g10[j] = v10[j] > 0.f ? g10[j] : 0.f; g11[j] = v11[j] > 0.f ? g11[j] : 0.f;
                s1[j] += g00[j]; s2[j] += g00[j] * y00[j];
                s1[j] += g01[j]; s2[j] += g01[j] * y01[j];
                s1[j] += g10[j]; s2[j] += g10[j] * y10[j];
                s1[j] += g11[j]; s2[j] += g11[j] * y11[j];
            }
        }
        dx[i00] = g00; dx[i01] = g01; dx[i10] = g10; dx[i11] = g11;
    }
    if constexpr (STATS) {
        __shared__ float red[256 * 8];
        const int tid = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s1[j]; red[tid * 8 + 4 + j] = s2[j]; }
        __syncthreads();
        if (tid < C4) {          // (thread t holds quad t % C4: the loop's stride is a multiple of C4)
            float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
            for (int t = tid; t < 256; t += C4)
#pragma unroll
                for (int j = 0; j < 4; ++j) { a1[j] += red[t * 8 + j]; a2[j] += red[t * 8 + 4 + j]; }
            float *dst = partial + ((size_t)blockIdx.x * (4 * C4) + tid * 4) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j * 2] = a1[j]; dst[j * 2 + 1] = a2[j]; }
        }
    }
}
int maxpool2_bwd_blocks(int B, int H, int W, int C) { return grid_for((size_t)B * (H / 2) * (W / 2) * (C / 4), 256); }
hipError_t launch_maxpool2_bwd(const float *x, const float *dout, int B, int H, int W, int C, float *dx, int accumulate,
                               hipStream_t st, const float *la, const float *lb, float *stats_partial) {
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    const int blocks = grid_for(total, 256);
    if (stats_partial) {       // the fused BatchNorm-backward statistics: lazy x, a gradient to complete, a fixed quad per thread
        if (!la || !lb || !accumulate || C % 4 || 256 % (C / 4)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(maxpool2_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(x),
                           reinterpret_cast<const f32x4 *>(dout), B, H, W, C / 4, reinterpret_cast<f32x4 *>(dx), accumulate,
                           reinterpret_cast<const f32x4 *>(la), reinterpret_cast<const f32x4 *>(lb), stats_partial);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(x),
                       reinterpret_cast<const f32x4 *>(dout), B, H, W, C / 4, reinterpret_cast<f32x4 *>(dx), accumulate,
                       reinterpret_cast<const f32x4 *>(la), reinterpret_cast<const f32x4 *>(lb), nullptr);
    return hipGetLastError();
}

// depthwise ConvTranspose2d(k4,s2,p1) backward wrt input: din[iy,ix] = sum_{ky,kx} dout[2iy-1+ky, 2ix-1+kx] * w[ky,kx]
__global__ void deconv4_bwd_data_kernel(const f32x4 *__restrict__ dout, int B, int H, int W, int C4,
                                        const f32x4 *__restrict__ wpk, f32x4 *__restrict__ din) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = e % C4;
        const size_t p = e / C4;
        const int ix = p % W, iy = (p / W) % H;
        const size_t b = p / ((size_t)W * H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int oy = 2 * iy - 1 + ky;
            if (oy < 0 || oy >= 2 * H) continue;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int ox = 2 * ix - 1 + kx;
                if (ox < 0 || ox >= 2 * W) continue;
                const f32x4 g = dout[((b * 2 * H + oy) * 2 * W + ox) * C4 + c], w = wpk[(ky * 4 + kx) * C4 + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(g[j], w[j], acc[j]);
            }
        }
        din[e] = acc;
    }
}
hipError_t launch_deconv4_bwd_data(const float *dout, int B, int H, int W, int C, const float *wpk, float *din,
                                   hipStream_t st) {
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(deconv4_bwd_data_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st,
                       reinterpret_cast<const f32x4 *>(dout), B, H, W, C / 4, reinterpret_cast<const f32x4 *>(wpk),
                       reinterpret_cast<f32x4 *>(din));
    return hipGetLastError();
}

// ... and wrt the (C,1,4,4) weights: dw[c,ky,kx] = sum_{b,iy,ix} in[b,iy,ix,c] * dout[b,2iy-1+ky,2ix-1+kx,c].
// One workgroup per (image row-block); partial [blocks][16][C] then reduced by colsum-like pass.
// FUSED (round 6): with din / wpk given the same pass also forms the gradient wrt the input -- the two kernels read the same
// 4x4 windows of dout (measured at B = 32: 502 + 556 MB for the 64-channel layers where 315 MB are algorithmic); the terms of
// din are added in deconv4_bwd_data_kernel's order (ky outer, kx inner), so its bits do not change.
// STATS (FUSED, lazy input): `in` is a BatchNorm + ReLU output with this deconv as its only consumer, so din IS its complete
// gradient: it is masked here (formed value > 0) and the BatchNorm-backward partials (sum d, sum d * y) per workgroup and
// channel go to `stats` [gridDim.x][C][2], summed like two more filter taps -- no reduction pass over the map.
template <bool FUSED, bool STATS = false>
__global__ __launch_bounds__(256) void deconv4_bwd_w_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                            int B, int H, int W, int C, float *__restrict__ partial,
                                                            const float *__restrict__ la, const float *__restrict__ lb,
                                                            const f32x4 *__restrict__ wpk, f32x4 *__restrict__ din,
                                                            float *__restrict__ stats) {
    static_assert(!STATS || FUSED, "statistics of the gradient this pass forms");
    // one workgroup per (b, iy) input row; a thread owns 4 channels and every XG-th column, 16-byte
    // loads; the column groups are summed by wave shuffles + one LDS image (fixed order)
    constexpr int NA = STATS ? 18 : 16;          // accumulators: the 16 taps (+ the two statistics sums)
    __shared__ float red[NA][256];
    const int C4 = C >> 2, XG = 256 / C4;
    const int tid = threadIdx.x, c4 = tid % C4, xg = tid / C4;
    const int b = blockIdx.x / H, iy = blockIdx.x % H;
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(in), *do4 = reinterpret_cast<const f32x4 *>(dout);
    f32x4 acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 lav = {1.f, 1.f, 1.f, 1.f}, lbv = {0.f, 0.f, 0.f, 0.f};       // lazy input (ConvSrc::la)
    if (la) { lav = reinterpret_cast<const f32x4 *>(la)[c4]; lbv = reinterpret_cast<const f32x4 *>(lb)[c4]; }
    f32x4 wk[FUSED ? 16 : 1];
    if constexpr (FUSED) {
#pragma unroll
        for (int k = 0; k < 16; ++k) wk[k] = wpk[k * C4 + c4];
    }
    for (int ix = xg; ix < W; ix += XG) {
        f32x4 v = in4[(((size_t)b * H + iy) * W + ix) * C4 + c4];
        [[maybe_unused]] f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
        [[maybe_unused]] const f32x4 yraw = v;
        if (la) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], lav[j], lbv[j]), 0.f);
        }
        // the 16 dout values of the window in two batches of eight, requested before any is used (behind `continue` branches the
        // loads were issued and waited for one at a time); a tap outside the map contributes an exact zero term
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            f32x4 dv[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oy = 2 * iy - 1 + 2 * kh + q;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ox = 2 * ix - 1 + kx;
                    const bool ok = oy >= 0 && oy < 2 * H && ox >= 0 && ox < 2 * W;
                    dv[q][kx] = ok ? do4[(((size_t)b * 2 * H + oy) * 2 * W + ox) * C4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int k = (2 * kh + q) * 4 + kx;
                    const f32x4 d = dv[q][kx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(v[j], d[j], acc[k][j]);
                    if constexpr (FUSED) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) gacc[j] = fmaf(d[j], wk[k][j], gacc[j]);
                    }
                }
        }
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gacc[j] = v[j] > 0.f ? gacc[j] : 0.f;
                acc[16][j] += gacc[j];
                acc[17][j] = fmaf(gacc[j], yraw[j], acc[17][j]);
            }
        }
        if constexpr (FUSED) din[(((size_t)b * H + iy) * W + ix) * C4 + c4] = gacc;
    }
    // lanes c4 + C4*j of a wave hold the same channels
    for (int o = C4; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] += __shfl_xor(acc[k][j], o);
    const int wave = tid >> 6, lane = tid & 63;
    for (int w = 0; w < 4; ++w) {
        if (wave == w && lane < C4 && lane < 64) {
            // with C4 = 64 a wave covers all channel groups once; otherwise lanes < C4 hold the wave's sum
#pragma unroll
            for (int k = 0; k < NA; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float *dst = &red[k][(tid % C4) * 4 + j];
                    *dst = (w == 0 ? 0.f : *dst) + acc[k][j];
                }
        }
        __syncthreads();
    }
    for (int e = tid; e < 16 * C; e += 256) partial[(size_t)blockIdx.x * 16 * C + e] = red[e / C][e % C];
    if constexpr (STATS) {
        for (int e = tid; e < 2 * C; e += 256) stats[((size_t)blockIdx.x * C + e % C) * 2 + e / C] = red[16 + e / C][e % C];
    }
}
__global__ __launch_bounds__(256) void deconv4_bwd_w_reduce_kernel(const float *__restrict__ partial, int nblocks, int C,
                                                                   float *__restrict__ dw /*(C,1,4,4)*/) {
    const int c = blockIdx.x / 16, k = blockIdx.x % 16;
    double s = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[((size_t)i * 16 + k) * C + c];
    __shared__ double sh[4];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dw[c * 16 + k] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}
size_t deconv4_bwd_w_partial_floats(int B, int H, int C) { return (size_t)B * H * 16 * C; }
hipError_t launch_deconv4_bwd_w(const float *in, const float *dout, int B, int H, int W, int C, float *partial, float *dw,
                                hipStream_t st, const float *la, const float *lb, const float *wpk, float *din, float *stats) {
    if (C % 4 || C > 256 || 256 % (C / 4) || (wpk != nullptr) != (din != nullptr)) return hipErrorInvalidValue;
    if (stats && (!din || !la || !lb)) return hipErrorInvalidValue;
    if (stats)
        hipLaunchKernelGGL((deconv4_bwd_w_kernel<true, true>), dim3(B * H), dim3(256), 0, st, in, dout, B, H, W, C, partial, la, lb,
                           reinterpret_cast<const f32x4 *>(wpk), reinterpret_cast<f32x4 *>(din), stats);
    else if (din)
        hipLaunchKernelGGL((deconv4_bwd_w_kernel<true, false>), dim3(B * H), dim3(256), 0, st, in, dout, B, H, W, C, partial, la, lb,
                           reinterpret_cast<const f32x4 *>(wpk), reinterpret_cast<f32x4 *>(din), nullptr);
    else
        hipLaunchKernelGGL((deconv4_bwd_w_kernel<false, false>), dim3(B * H), dim3(256), 0, st, in, dout, B, H, W, C, partial, la, lb, nullptr, nullptr,
                           nullptr);
    hipLaunchKernelGGL(deconv4_bwd_w_reduce_kernel, dim3(C * 16), dim3(256), 0, st, partial, B * H, C, dw);
    return hipGetLastError();
}

}  // namespace mc
