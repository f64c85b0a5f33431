// Training-side kernels: target rasterisation, the ten MonoCon losses and their gradients with
// respect to the prediction maps.
//
// Replaces reference utils/target_generator.py:30-177 (+ utils/tensor_ops.py:62-125 gaussian
// helpers), model/dense_heads/monocon_heads.py:203-310 (_get_losses) and losses/*.py.
// The reference builds targets with a Python loop over batch x objects x 9 keypoints that issues
// tiny device ops with implicit syncs; here one workgroup per (image, label slot) does the same
// arithmetic -- including the float32 / python-float mix of gaussian_radius and the truncations
// that decide which pixel receives target 1.0 -- and max-splats with integer atomics (order
// independent, bit-reproducible).
#include "kernels.h"
#include "train.h"

namespace mc {

// ------------------------------------------------------------------ target generator
__device__ __forceinline__ float py_sqrt_f32(float v) { return (float)sqrt((double)v); }  // math.sqrt(tensor) -> fp32 operand

// reference utils/tensor_ops.py:77-99 evaluated exactly as torch evaluates it on 0-dim fp32 tensors
// with python scalars (scalars are rounded to fp32; math.sqrt runs in double on the fp32 value)
__device__ float gaussian_radius_ref(float height, float width) {
    const float mo = 0.3f;
    (void)mo;
    const float b1 = height + width;
    const float c1 = width * height * 0.7f / 1.3f;
    const float sq1 = py_sqrt_f32(b1 * b1 - 4.0f * c1);
    const float r1 = (b1 - sq1) / 2.0f;
    const float b2 = 2.0f * (height + width);
    const float c2 = 0.7f * width * height;
    const float sq2 = py_sqrt_f32(b2 * b2 - 16.0f * c2);
    const float r2 = (b2 - sq2) / 8.0f;
    const float b3 = -0.6f * (height + width);
    const float c3 = -0.7f * width * height;
    const float sq3 = py_sqrt_f32(b3 * b3 - 4.8f * c3);
    const float r3 = (b3 + sq3) / 2.4f;
    return fminf(r1, fminf(r2, r3));
}

// max-splat of exp(-(x^2+y^2)/(2 sigma^2)) (sigma = (2r+1)/6; entries < eps zeroed) clipped to the map;
// all values are >= 0 so the IEEE bit pattern orders like an unsigned integer
__device__ void splat(float *plane, int H, int W, int cx, int cy, int r, int tid, int nthr) {
    const int d = 2 * r + 1;
    const float sigma = (float)d / 6.0f;
    const float two_s2 = (float)(2.0 * ((double)d / 6.0) * ((double)d / 6.0));
    (void)sigma;
    const int left = min(cx, r), right = min(W - cx, r + 1);
    const int top = min(cy, r), bottom = min(H - cy, r + 1);
    const int ww = left + right, hh = top + bottom;
    for (int e = tid; e < ww * hh; e += nthr) {
        const int dx = e % ww - left, dy = e / ww - top;
        float g = expf(-(float)(dx * dx + dy * dy) / two_s2);
        if (g < 1.1920929e-07f) g = 0.f;     // h[h < eps * h.max()] = 0 with h.max() == 1
        atomicMax(reinterpret_cast<unsigned *>(&plane[(cy + dy) * W + (cx + dx)]), __float_as_uint(g));
    }
}

__device__ __forceinline__ float py_fmod_pos(float a, float m) {   // python/torch `%` with positive modulus
    float r = fmodf(a, m);
    if (r != 0.f && r < 0.f) r += m;
    return r;
}

__global__ __launch_bounds__(256) void make_targets_kernel(const TargetArgs a) {
    const int b = blockIdx.x / a.max_objs, slot = blockIdx.x % a.max_objs;
    const int tid = threadIdx.x;
    const float *mask = a.mask + (size_t)b * a.max_objs;
    if (mask[slot] == 0.f) return;
    int o = 0;                                   // rank among the valid objects of this image
    for (int i = 0; i < slot; ++i) o += mask[i] != 0.f;
    const int HW = a.fh * a.fw;
    const float wr = a.w_ratio, hr = a.h_ratio;
    const float *bb = a.gt_bboxes + ((size_t)b * a.max_objs + slot) * 4;
    const float ctx = (bb[0] + bb[2]) * wr / 2.0f;
    const float cty = (bb[1] + bb[3]) * hr / 2.0f;
    const int xi = (int)ctx, yi = (int)cty;       // .int(): truncation toward zero
    const float bh = (bb[3] - bb[1]) * hr, bw = (bb[2] - bb[0]) * wr;
    const int rad = max(0, (int)gaussian_radius_ref(bh, bw));
    const int cls = (int)(long long)a.gt_labels[(size_t)b * a.max_objs + slot];
    // A centre outside the map or a class id outside [0, num_classes) makes the reference fail with an index error
    // (target_generator.py:70-75); the Python boundary raises before launching (hipmonocon/train.py).  For direct
    // C-ABI callers the kernel stays memory-safe: no splat for such an object, gather index clamped into the map.
    const bool centre_ok = xi >= 0 && xi < a.fw && yi >= 0 && yi < a.fh && cls >= 0 && cls < a.num_classes;
    if (centre_ok)
        splat(a.center_heatmap + ((size_t)b * a.num_classes + cls) * HW, a.fh, a.fw, xi, yi, rad, tid, blockDim.x);
    const size_t row = (size_t)b * a.max_objs + o;
    const float *b3 = a.gt_bboxes_3d + ((size_t)b * a.max_objs + slot) * 7;
    if (tid == 0) {
        a.indices[row] = (long long)min(max(yi, 0), a.fh - 1) * a.fw + min(max(xi, 0), a.fw - 1);
        a.wh[row * 2 + 0] = bw; a.wh[row * 2 + 1] = bh;
        a.offset[row * 2 + 0] = ctx - (float)xi; a.offset[row * 2 + 1] = cty - (float)yi;
        a.dim[row * 3 + 0] = b3[3]; a.dim[row * 3 + 1] = b3[4]; a.dim[row * 3 + 2] = b3[5];
        a.depth[row] = a.depths[(size_t)b * a.max_objs + slot];
        // _convert_angle_to_class (target_generator.py:141-149), fp32 tensor arithmetic
        const float two_pi = 6.283185307179586f, per = 0.5235987755982988f;
        const float ang = py_fmod_pos(b3[6], two_pi);
        const float shifted = py_fmod_pos(ang + 0.2617993877991494f, two_pi);
        const int cid = (int)(shifted / per);
        a.alpha_cls[row] = (float)cid;
        a.alpha_offset[row] = shifted - (float)((double)cid * 0.5235987755982988 + 0.2617993877991494);
        a.mask_target[row] = 1;
    }
    // keypoints (one thread each for the scalars, whole block for the splats)
    const float *kp = a.gt_kpts_2d + ((size_t)b * a.max_objs + slot) * a.num_kpt * 2;
    const float *kv = a.gt_kpts_valid + ((size_t)b * a.max_objs + slot) * a.num_kpt;
    for (int k = 0; k < a.num_kpt; ++k) {
        if (kv[k] < 1.f) continue;
        const float kx = kp[2 * k] * wr, ky = kp[2 * k + 1] * hr;
        const int kxi = (int)kx, kyi = (int)ky;
        const bool inside = kxi >= 0 && kxi < a.fw && kyi >= 0 && kyi < a.fh;
        if (tid == 0) {
            a.c2k[row * 18 + 2 * k] = kx - (float)xi;
            a.c2k[row * 18 + 2 * k + 1] = ky - (float)yi;
            a.mask_c2k[row * 18 + 2 * k] = 1.f; a.mask_c2k[row * 18 + 2 * k + 1] = 1.f;
            if (inside) {
                a.indices_kpt[row * 9 + k] = (long long)kyi * a.fw + kxi;
                a.kho[row * 18 + 2 * k] = kx - (float)kxi;
                a.kho[row * 18 + 2 * k + 1] = ky - (float)kyi;
                a.mask_kho[row * 18 + 2 * k] = 1.f; a.mask_kho[row * 18 + 2 * k + 1] = 1.f;
            }
        }
        if (inside) splat(a.kpt_heatmap + ((size_t)b * a.num_kpt + k) * HW, a.fh, a.fw, kxi, kyi, rad, tid, blockDim.x);
    }
}

hipError_t launch_make_targets(const TargetArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(make_targets_kernel, dim3(a.B * a.max_objs), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ Gaussian focal loss
// reference losses/focal_loss.py:21-44.  Streaming reduction (the one HBM-bound loss: two maps,
// pred + target read once).  Per-block partial (pos, neg, npos) then a one-block finalise.
__global__ __launch_bounds__(256) void focal_partial_kernel(const float *__restrict__ p, const float *__restrict__ t,
                                                            size_t n, float *partial) {
    float pos = 0.f, neg = 0.f, np = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pv = p[i], tv = t[i];
        if (tv == 1.f) {
            pos += logf(pv + 1e-12f) * (1.f - pv) * (1.f - pv);
            np += 1.f;
        } else if (tv < 1.f) {
            const float q = 1.f - tv, q2 = q * q;
            neg += logf(1.f - pv + 1e-12f) * pv * pv * (q2 * q2);
        }
    }
    __shared__ float red[3][4];
    float v[3] = {pos, neg, np};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o);
        if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        partial[blockIdx.x * 3 + j] = red[j][0] + red[j][1] + red[j][2] + red[j][3];
    }
}

__global__ void focal_final_kernel(const float *partial, int nblocks, float *loss_out, float *aux /*[2]: npos, scale*/) {
    double pos = 0, neg = 0, np = 0;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        pos += partial[i * 3 + 0]; neg += partial[i * 3 + 1]; np += partial[i * 3 + 2];
    }
    for (int o = 32; o > 0; o >>= 1) { pos += __shfl_xor(pos, o); neg += __shfl_xor(neg, o); np += __shfl_xor(np, o); }
    if (threadIdx.x == 0) {
        *loss_out = (np == 0.0) ? (float)(-neg) : (float)(-(pos + neg) / np);
        aux[0] = (float)np;
    }
}

// d loss / d logit through clamp(sigmoid(x), 1e-4, 1-1e-4):  p is the clamped prediction
__global__ __launch_bounds__(256) void focal_grad_kernel(const float *__restrict__ p, const float *__restrict__ t,
                                                         size_t n, const float *aux, const float *gscale, int gidx,
                                                         float *__restrict__ dlogit, int wrt_pred) {
    const float np = aux[0];
    const float k = -gscale[gidx] / (np == 0.f ? 1.f : np);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pv = p[i], tv = t[i];
        float d = 0.f;
        if (tv == 1.f) {
            d = (1.f - pv) * (1.f - pv) / (pv + 1e-12f) - 2.f * (1.f - pv) * logf(pv + 1e-12f);
        } else if (tv < 1.f) {
            const float q = 1.f - tv, q2 = q * q;
            d = (-pv * pv / (1.f - pv + 1e-12f) + 2.f * pv * logf(1.f - pv + 1e-12f)) * (q2 * q2);
        }
        const bool inside = pv > 1e-4f && pv < 1.f - 1e-4f;
        dlogit[i] = wrt_pred ? k * d : (inside ? k * d * pv * (1.f - pv) : 0.f);
    }
}

constexpr int FOCAL_BLOCKS = 1024;
hipError_t launch_focal(const float *p, const float *t, size_t n, float *partial, float *loss_out, float *aux,
                        hipStream_t st) {
    hipLaunchKernelGGL(focal_partial_kernel, dim3(FOCAL_BLOCKS), dim3(256), 0, st, p, t, n, partial);
    hipLaunchKernelGGL(focal_final_kernel, dim3(1), dim3(64), 0, st, partial, FOCAL_BLOCKS, loss_out, aux);
    return hipGetLastError();
}
hipError_t launch_focal_grad(const float *p, const float *t, size_t n, const float *aux, const float *gscale, int gidx,
                             float *dlogit, hipStream_t st, int wrt_pred) {
    hipLaunchKernelGGL(focal_grad_kernel, dim3(2048), dim3(256), 0, st, p, t, n, aux, gscale, gidx, dlogit, wrt_pred);
    return hipGetLastError();
}
int focal_partial_floats() { return FOCAL_BLOCKS * 3; }

// ------------------------------------------------------------------ gathered regression losses
// reference monocon_heads.py:219-298 + losses/{l1,dim,depth,cross_entropy}_loss.py.  One workgroup:
// rows = B*max_objs label slots (valid ones are the first n_b of each image).  mode 0: forward
// (writes the 8 scalar losses); mode 1: gradient scatter into the (zeroed) d-prediction maps.
__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) - (v < 0.f); }

__device__ double block_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    return s;
}

// Three launches instead of one workgroup (round 5: one CU issuing all ~60 gathers of all B * max_objs label rows took 80 us,
// twice per step): `phase` 0 = the twelve sums over this workgroup's rows -> a.partial[workgroup][12]; 1 = one wave adds the
// workgroups' rows in order -> totals (kept behind the partials), and in mode 0 writes the losses; 2 = mode 1's gradient
// scatter of this workgroup's rows, from the totals.
constexpr int GL_MAXWG = 64, GL_NSUM = 12;
__global__ __launch_bounds__(256) void gathered_loss_kernel(const GatherLossArgs a, int mode, int phase) {
    __shared__ double sh[16];
    const int rows_all = a.B * a.max_objs, HW = a.HW;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int rows_per_wg = (rows_all + (int)gridDim.x - 1) / (int)gridDim.x;
    const int row0 = phase == 1 ? 0 : (int)blockIdx.x * rows_per_wg;
    const int rows = phase == 1 ? 0 : min(rows_all, row0 + rows_per_wg);       // this workgroup's rows: [row0, rows)
    double *totals = a.partial + GL_MAXWG * GL_NSUM;
    auto P = [&](int p, int nch, int b, int ch, long long ind) -> size_t { (void)p; return ((size_t)b * nch + ch) * HW + ind; };
    // ---- pass 1: sums
    double s_n = 0, s_wh = 0, s_off = 0, s_l1dim = 0, s_ldim = 0, s_dep = 0, s_c2k = 0, s_mc2k = 0, s_kho = 0, s_mkho = 0,
           s_bce = 0, s_areg = 0;
    for (int r = row0 + tid; phase == 0 && r < rows; r += nthr) {
        if (!a.mask_target[r]) continue;
        const int b = r / a.max_objs;
        const long long ind = a.indices[r];
        s_n += 1;
        for (int c = 0; c < 2; ++c) {
            s_wh += fabsf(a.pred[2][P(2, 2, b, c, ind)] - a.wh[r * 2 + c]);
            s_off += fabsf(a.pred[3][P(3, 2, b, c, ind)] - a.offset[r * 2 + c]);
        }
        for (int c = 0; c < 3; ++c) {
            const float p = a.pred[6][P(6, 3, b, c, ind)], t = a.dim[r * 3 + c];
            s_l1dim += fabsf(p - t);
            s_ldim += fabsf(p - t) / p;
        }
        {
            const float d = a.pred[7][P(7, 2, b, 0, ind)], s = a.pred[7][P(7, 2, b, 1, ind)];
            s_dep += 1.4142f * expf(-s) * fabsf(d - a.depth[r]) + s;
        }
        for (int c = 0; c < 18; ++c) {
            const float m = a.mask_c2k[r * 18 + c];
            s_c2k += fabsf(a.pred[5][P(5, 18, b, c, ind)] * m - a.c2k[r * 18 + c]);
            s_mc2k += m;
            const long long ik = a.indices_kpt[r * 9 + c / 2];
            s_kho += fabsf(a.pred[4][P(4, 2, b, c & 1, ik)] - a.kho[r * 18 + c]);
            s_mkho += a.mask_kho[r * 18 + c];
        }
        const int cls = (int)a.alpha_cls[r];
        for (int c = 0; c < 12; ++c) {
            const float x = a.pred[8][P(8, 12, b, c, ind)], y = (c == cls) ? 1.f : 0.f;
            s_bce += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        }
        s_areg += fabsf(a.pred[9][P(9, 12, b, cls, ind)] - a.alpha_offset[r]);
    }
    if (phase == 0) {
        const double v[GL_NSUM] = {block_sum(s_n, sh), block_sum(s_wh, sh), block_sum(s_off, sh), block_sum(s_l1dim, sh),
                                   block_sum(s_ldim, sh), block_sum(s_dep, sh), block_sum(s_c2k, sh), block_sum(s_mc2k, sh),
                                   block_sum(s_kho, sh), block_sum(s_mkho, sh), block_sum(s_bce, sh), block_sum(s_areg, sh)};
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < GL_NSUM; ++i) a.partial[(size_t)blockIdx.x * GL_NSUM + i] = v[i];
        }
        return;
    }
    if (phase == 1) {      // (launched with one workgroup; a.nwg = the workgroups of phase 0)
        if (tid < GL_NSUM) {
            double t = 0;
            for (int g = 0; g < a.nwg; ++g) t += a.partial[(size_t)g * GL_NSUM + tid];
            totals[tid] = t;
        }
        __syncthreads();
    }
    const double n = totals[0], wh = totals[1], off = totals[2], l1dim = totals[3], ldim = totals[4], dep = totals[5],
                 c2k = totals[6], mc2k = totals[7], kho = totals[8], mkho = totals[9], bce = totals[10], areg = totals[11];
    if (mode == 0) {
        if (tid == 0) {
            float *L = a.losses;   // LOSS order: 0 center_heatmap 1 wh 2 offset 3 dim 4 c2k 5 kpt_heatmap 6 kho 7 alpha_cls 8 alpha_reg 9 depth
            const double nn = n > 0 ? n : 1;
            L[1] = (float)(0.1 * wh / (2 * nn));
            L[2] = (float)(off / (2 * nn));
            // mean(l * comp) with comp = l1_mean / l_mean  ==  l1_mean (up to rounding)
            const double l1m = l1dim / (3 * nn), lm = ldim / (3 * nn);
            L[3] = (float)(lm * (float)(l1m / lm));
            L[4] = (float)(c2k / (mc2k + 1e-12));
            L[6] = (float)(kho / (mkho + 1e-12));
            L[7] = n > 0 ? (float)(bce / (12 * nn)) : 0.f;
            L[8] = (float)(areg / nn);
            L[9] = (float)(dep / nn);
            a.aux[0] = (float)n;
        }
        return;
    }
    // ---- pass 2: gradients, scattered with atomics (two objects may share a pixel)
    if (phase != 2 || n <= 0) return;
    const float g_wh = a.gscale[1] * 0.1f / (float)(2 * n), g_off = a.gscale[2] / (float)(2 * n);
    const float comp = (float)((l1dim / (3 * n)) / (ldim / (3 * n)));
    const float g_dim = a.gscale[3] * comp / (float)(3 * n);
    const float g_c2k = a.gscale[4] / (float)(mc2k + 1e-12), g_kho = a.gscale[6] / (float)(mkho + 1e-12);
    const float g_bce = a.gscale[7] / (float)(12 * n), g_areg = a.gscale[8] / (float)n, g_dep = a.gscale[9] / (float)n;
    for (int r = row0 + tid; r < rows; r += nthr) {
        if (!a.mask_target[r]) continue;
        const int b = r / a.max_objs;
        const long long ind = a.indices[r];
        for (int c = 0; c < 2; ++c) {
            atomicAdd(&a.dpred[2][P(2, 2, b, c, ind)], g_wh * sgnf(a.pred[2][P(2, 2, b, c, ind)] - a.wh[r * 2 + c]));
            atomicAdd(&a.dpred[3][P(3, 2, b, c, ind)], g_off * sgnf(a.pred[3][P(3, 2, b, c, ind)] - a.offset[r * 2 + c]));
        }
        for (int c = 0; c < 3; ++c) {
            const float p = a.pred[6][P(6, 3, b, c, ind)];
            atomicAdd(&a.dpred[6][P(6, 3, b, c, ind)], g_dim * sgnf(p - a.dim[r * 3 + c]) / p);
        }
        {
            const float d = a.pred[7][P(7, 2, b, 0, ind)], s = a.pred[7][P(7, 2, b, 1, ind)];
            const float e = 1.4142f * expf(-s);
            const float gd = g_dep * e * sgnf(d - a.depth[r]);
            // d = 1/(sigmoid(x)+eps) - 1  =>  dd/dx = -sig(1-sig)/(sig+eps)^2 with sig = 1/(d+1) - eps
            const float sig = 1.f / (d + 1.f) - 1e-12f;
            atomicAdd(&a.dpred[7][P(7, 2, b, 0, ind)],
                      a.wrt_pred ? gd : gd * (-sig * (1.f - sig) / ((sig + 1e-12f) * (sig + 1e-12f))));
            atomicAdd(&a.dpred[7][P(7, 2, b, 1, ind)], g_dep * (1.f - e * fabsf(d - a.depth[r])));
        }
        for (int c = 0; c < 18; ++c) {
            const float m = a.mask_c2k[r * 18 + c];
            const float p = a.pred[5][P(5, 18, b, c, ind)];
            atomicAdd(&a.dpred[5][P(5, 18, b, c, ind)], g_c2k * sgnf(p * m - a.c2k[r * 18 + c]) * m);
            const long long ik = a.indices_kpt[r * 9 + c / 2];
            atomicAdd(&a.dpred[4][P(4, 2, b, c & 1, ik)],
                      g_kho * sgnf(a.pred[4][P(4, 2, b, c & 1, ik)] - a.kho[r * 18 + c]));
        }
        const int cls = (int)a.alpha_cls[r];
        for (int c = 0; c < 12; ++c) {
            const float x = a.pred[8][P(8, 12, b, c, ind)], y = (c == cls) ? 1.f : 0.f;
            atomicAdd(&a.dpred[8][P(8, 12, b, c, ind)], g_bce * (1.f / (1.f + expf(-x)) - y));
        }
        atomicAdd(&a.dpred[9][P(9, 12, b, cls, ind)], g_areg * sgnf(a.pred[9][P(9, 12, b, cls, ind)] - a.alpha_offset[r]));
    }
}

size_t gathered_loss_ws_doubles() { return (size_t)GL_MAXWG * GL_NSUM + GL_NSUM; }
hipError_t launch_gathered_losses(const GatherLossArgs &a_in, int mode, hipStream_t st) {
    if (!a_in.partial) return hipErrorInvalidValue;
    GatherLossArgs a = a_in;
    const int rows = a.B * a.max_objs;
    a.nwg = rows < 64 * GL_MAXWG ? (rows + 63) / 64 : GL_MAXWG;       // 64 label rows per workgroup up to the table's size
    if (a.nwg < 1) a.nwg = 1;
    hipLaunchKernelGGL(gathered_loss_kernel, dim3(a.nwg), dim3(256), 0, st, a, mode, 0);
    hipLaunchKernelGGL(gathered_loss_kernel, dim3(1), dim3(64), 0, st, a, mode, 1);
    if (mode == 1) hipLaunchKernelGGL(gathered_loss_kernel, dim3(a.nwg), dim3(256), 0, st, a, mode, 2);
    return hipGetLastError();
}

}  // namespace mc
