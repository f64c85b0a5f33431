// Fused gradient-norm clip + AdamW over a table of parameter tensors.
// Replaces torch.nn.utils.clip_grad_norm_(35, L2) + torch.optim.AdamW.step of the reference's
// train loop (engine/monocon_engine.py:94-102): two launches per step instead of several
// hundred per-tensor kernels; 16 B/param of traffic (p, g, m, v read; p, m, v written).
#include "train.h"

namespace mc {

__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const OptTensor *tab, const OptChunk *chunks, int nchunks,
                                                               float *partial) {
    float s = 0.f;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const OptChunk ck = chunks[c];
        const float *g = tab[ck.tensor].g + ck.begin;
        for (int i = threadIdx.x; i < ck.count; i += 256) s += g[i] * g[i];
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

__global__ __launch_bounds__(256) void adamw_kernel(const OptTensor *tab, const OptChunk *chunks, int nchunks,
                                                    const float *normcoef, AdamHyper hp) {
    const float coef = normcoef[1];
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const OptChunk ck = chunks[c];
        const OptTensor t = tab[ck.tensor];
        for (int i = threadIdx.x; i < ck.count; i += 256) {
            const int j = ck.begin + i;
            const float g = t.g[j] * coef;
            float p = t.p[j] * hp.decay;                 // p *= 1 - lr*wd   (decoupled weight decay)
            const float m = t.m[j] * hp.beta1 + g * hp.one_minus_beta1;
            const float v = t.v[j] * hp.beta2 + g * g * hp.one_minus_beta2;
            const float denom = sqrtf(v) / hp.sqrt_bc2 + hp.eps;
            p -= hp.step_size * (m / denom);
            t.p[j] = p; t.m[j] = m; t.v[j] = v;
        }
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
