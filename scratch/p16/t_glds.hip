#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// probe: buffer_load ... lds (16 B per lane) -- does an out-of-range lane write ZERO to LDS?
extern "C" __global__ void probe(const float *src, int nbytes, float *out, int oob_lane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    // poison
    reinterpret_cast<float4 *>(lds)[lane] = make_float4(-7.f, -7.f, -7.f, -7.f);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
    int voff = lane * 16;
    if (lane == oob_lane) voff = (int)0x80000000;
    if (lane == oob_lane + 1) voff = nbytes;          // just past the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 v = reinterpret_cast<float4 *>(lds)[lane];
    out[lane * 4 + 0] = v.x; out[lane * 4 + 1] = v.y; out[lane * 4 + 2] = v.z; out[lane * 4 + 3] = v.w;
}
int main() {
    const int n = 64 * 4;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, d, n * 4, o, 5);
    std::vector<float> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float exp0 = (l == 5 || l == 6) ? 0.f : h[l * 4];
        if (r[l * 4] != exp0) { ++bad; }
    }
    printf("lane4: %g %g  lane5(oob): %g %g %g %g  lane6(past end): %g  lane7: %g  mismatches %d\n", r[16], r[17], r[20], r[21], r[22], r[23], r[24], r[28], bad);
    return 0;
}
