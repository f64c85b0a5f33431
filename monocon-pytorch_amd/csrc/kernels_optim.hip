// Fused gradient-norm clip + AdamW over a table of parameter tensors.
// Replaces torch.nn.utils.clip_grad_norm_(35, L2) + torch.optim.AdamW.step of the reference's
// train loop (engine/monocon_engine.py:94-102): two launches per step instead of several
// hundred per-tensor kernels; 16 B/param of traffic (p, g, m, v read; p, m, v written).
#include <stdint.h>
#include "conv_mfma.h"
#include "train.h"

namespace mc {

// Every tensor of the table starts 16-byte aligned (FlatGrads pads each gradient to a multiple of 4 floats, torch
// allocations are 256-byte aligned) and every chunk starts at a multiple of 65536 elements, so a chunk is walked as
// float4 with a scalar tail: the passes run at the HBM roof instead of at the dword-load issue rate.
__device__ __forceinline__ bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const OptTensor *tab, const OptChunk *chunks, int nchunks,
                                                               float *partial) {
    float s = 0.f;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const OptChunk ck = chunks[c];
        const float *g = tab[ck.tensor].g + ck.begin;
        int i0 = 0;
        if (aligned16(g)) {
            const int n4 = ck.count >> 2;
            const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
            for (int i = threadIdx.x; i < n4; i += 256) {
                const f32x4 v = g4[i];
                s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            }
            i0 = n4 << 2;
        }
        for (int i = i0 + threadIdx.x; i < ck.count; i += 256) s += g[i] * g[i];
    }
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void gradnorm_final_kernel(const float *partial, int n, float max_norm, float *out /*[2]: norm, coef*/) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s);
        out[0] = norm;
        const float coef = max_norm / (norm + 1e-6f);
        out[1] = (max_norm > 0.f && coef < 1.f) ? coef : 1.f;
    }
}

__device__ __forceinline__ void adamw_one(float &p, float g, float &m, float &v, float coef, const AdamHyper &hp) {
    g *= coef;
    p *= hp.decay;                               // p *= 1 - lr*wd   (decoupled weight decay)
    m = m * hp.beta1 + g * hp.one_minus_beta1;
    v = v * hp.beta2 + g * g * hp.one_minus_beta2;
    const float denom = sqrtf(v) / hp.sqrt_bc2 + hp.eps;
    p -= hp.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(const OptTensor *tab, const OptChunk *chunks, int nchunks,
                                                    const float *normcoef, AdamHyper hp) {
    const float coef = normcoef[1];
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const OptChunk ck = chunks[c];
        const OptTensor t = tab[ck.tensor];
        float *p = t.p + ck.begin, *m = t.m + ck.begin, *v = t.v + ck.begin;
        const float *g = t.g + ck.begin;
        int i0 = 0;
        if (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) {
            const int n4 = ck.count >> 2;
            for (int i = threadIdx.x; i < n4; i += 256) {
                f32x4 p4 = reinterpret_cast<f32x4 *>(p)[i], m4 = reinterpret_cast<f32x4 *>(m)[i], v4 = reinterpret_cast<f32x4 *>(v)[i];
                const f32x4 g4 = reinterpret_cast<const f32x4 *>(g)[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pj = p4[j], mj = m4[j], vj = v4[j];
                    adamw_one(pj, g4[j], mj, vj, coef, hp);
                    p4[j] = pj; m4[j] = mj; v4[j] = vj;
                }
                reinterpret_cast<f32x4 *>(p)[i] = p4;
                reinterpret_cast<f32x4 *>(m)[i] = m4;
                reinterpret_cast<f32x4 *>(v)[i] = v4;
            }
            i0 = n4 << 2;
        }
        for (int i = i0 + threadIdx.x; i < ck.count; i += 256) adamw_one(p[i], g[i], m[i], v[i], coef, hp);
    }
}

constexpr int OPT_BLOCKS = 2048;
hipError_t launch_clip_adamw(const OptTensor *tab, const OptChunk *chunks, int nchunks, float *partial, float *normcoef,
                             float max_norm, const AdamHyper &hp, hipStream_t st) {
    const int blocks = nchunks < OPT_BLOCKS ? nchunks : OPT_BLOCKS;
    hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(blocks), dim3(256), 0, st, tab, chunks, nchunks, partial);
    hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(64), 0, st, partial, blocks, max_norm, normcoef);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, st, tab, chunks, nchunks, normcoef, hp);
    return hipGetLastError();
}
int opt_partial_floats() { return OPT_BLOCKS; }

}  // namespace mc
