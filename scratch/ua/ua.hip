#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(2))) U8 { _Float16 v[8]; };
__global__ void k(const _Float16* in, _Float16* out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
    __syncthreads();
    const U8 u = *reinterpret_cast<const U8*>(&lds[threadIdx.x * 3 + 1]);   // 2-byte aligned, lane-dependent
    for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = u.v[j];
}
int main() {
    _Float16 h[1024], *d, *o, res[512];
    for (int i = 0; i < 1024; ++i) h[i] = (_Float16)i;
    hipMalloc(&d, 2048); hipMalloc(&o, 1024); hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); hipMemcpy(res, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 8; ++j) if ((float)res[t * 8 + j] != (float)(t * 3 + 1 + j)) ++bad;
    printf("packed-struct read: %d mismatches\n", bad);
    return 0;
}
