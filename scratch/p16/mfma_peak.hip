// What does the fp16 matrix pipe sustain when NOTHING else runs?  Pure v_mfma_f32_32x32x16_f16 loops on register operands:
// zero / constant / random data, 1 or 2 waves per SIMD, and the f16x2 pattern (3 MFMAs sharing operands, 2 accumulator sets).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void peak(const f16x8 *src, float *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[t] = s;
}

int main() {
    const int N = 1 << 16;
    std::vector<_Float16> h((size_t)N * 8);
    f16x8 *d; float *o;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 4 << 20));
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::normal_distribution<float> G(0.f, 1.f);
    const char *fills[] = {"zero", "const 1.0", "uniform [-1,1)", "relu(normal)", "relu(normal) hi / lo pieces alternating"};
    for (int f = 0; f < 5; ++f) {
        for (size_t i = 0; i < h.size(); ++i) {
            float v = 0.f;
            if (f == 1) v = 1.f;
            if (f == 2) v = U(rng);
            if (f == 3) { v = G(rng); v = v > 0 ? v : 0.f; }
            if (f == 4) { v = G(rng); v = v > 0 ? v : 0.f; if ((i >> 3) & 1) { _Float16 hi = (_Float16)(v * 1024.f); v = v * 1024.f - (float)hi; } else v *= 1024.f; }
            h[i] = (_Float16)v;
        }
        CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int wps = 1; wps <= 2; ++wps) {
            const int grid = 256 * wps, iters = 20000;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(peak<8>, dim3(grid), dim3(256), 0, 0, d, o, 1000);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(peak<8>, dim3(grid), dim3(256), 0, 0, d, o, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = (double)grid * 4 * iters * 8 * 32 * 32 * 16 * 2;
            printf("%-42s %d wave(s)/SIMD: %7.1f ms  %7.1f TF\n", fills[f], wps, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
