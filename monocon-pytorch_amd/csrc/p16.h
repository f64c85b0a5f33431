// The "P16" activation format of precision mode 4 (round 4).
//
// An fp32-sized element holds the two fp16 pieces of x * 2^e -- hi = fp16(x * 2^e), lo = fp16(x * 2^e - hi): the same split
// the f16x2 convolutions used to make on the fly (conv_bf16.hip, SPL = 2), made ONCE by the tensor's producer.  Layout:
// NHWC with 4 bytes per element, octet-planar inside a pixel: for channels 8o .. 8o+7 the 32 bytes at offset 32*o are
// [h0..h7][l0..l7].  One 16-byte run is exactly one lane's A operand of v_mfma_f32_32x32x16_f16, so a convolution stages
// its input with a plain copy -- here a DMA straight into LDS (conv_p16.hip) -- with no convert / split work on the VALU.
//
// e is ONE exponent per tensor, chosen by the producer BEFORE it writes (a consumer reads it from the tensor's exponent
// word): from a sound UPPER BOUND of max |x| (e.g. |a_c| * max|y| + |b_c| for a BatchNorm output), not from the exact
// maximum.  A loose bound costs nothing in relative precision: hi and lo are floating-point numbers, so every element
// within 2^-3 * 2^-e of ... down to the fp16 normal range keeps 22 significant bits; only elements below 2^-18 of the
// BOUND fall back to an absolute error of 2^-39 of the bound (a bound 2^L too large: 2^(L-39) of the true maximum, still
// far below fp32's own 2^-24 for any L a one-layer bound produces).  What the bound must never do is UNDER-estimate
// (fp16 overflows at 65504): bounds are built from triangle inequalities only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.h"

namespace mc {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// exponent of a tensor whose |x| is bounded by `bound` (>= 0, finite): bound * 2^e lands in [2^14, 2^15)
__host__ __device__ __forceinline__ int p16_exp_of_bound(float bound) {
    return f16_scale_exp(__builtin_bit_cast(unsigned, bound));
}
// byte offset of the hi pieces of channel quad q (channels 4q .. 4q+3) inside a pixel; the lo pieces sit 16 bytes further
__host__ __device__ __forceinline__ int p16_quad_off(int q) { return (q >> 1) * 32 + (q & 1) * 8; }

__device__ __forceinline__ void p16_split4(const f32x4 v, float s, f16x4 &hi, f16x4 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float r = v[j] * s;
        hi[j] = (_Float16)r;
        lo[j] = (_Float16)(r - (float)hi[j]);
    }
}
__device__ __forceinline__ f32x4 p16_join4(const f16x4 hi, const f16x4 lo, float inv) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((float)hi[j] + (float)lo[j]) * inv;     // hi + lo is exact in fp32 (<= 23 bits)
    return v;
}
// load / store of one channel quad of pixel row `row` (row pointer = tensor + pixel * C * 4 bytes)
__device__ __forceinline__ f32x4 p16_load4(const void *row, int q, float inv) {
    const char *p = static_cast<const char *>(row) + p16_quad_off(q);
    return p16_join4(*reinterpret_cast<const f16x4 *>(p), *reinterpret_cast<const f16x4 *>(p + 16), inv);
}
__device__ __forceinline__ void p16_store4(void *row, int q, const f32x4 v, float s) {
    f16x4 hi, lo;
    p16_split4(v, s, hi, lo);
    char *p = static_cast<char *>(row) + p16_quad_off(q);
    *reinterpret_cast<f16x4 *>(p) = hi;
    *reinterpret_cast<f16x4 *>(p + 16) = lo;
}

// fp32 NHWC -> P16 with the exponent of the tensor's max |x| (op-level entry points and tests: tensors that enter mode 4
// from outside the plans' own producers); *e_out receives the exponent
hipError_t launch_p16_encode(const float *x, size_t pixels, int C, const unsigned *amax_slot, void *dst, int *e_out, hipStream_t st);
hipError_t launch_p16_decode(const void *src, size_t pixels, int C, const int *e, float *dst, hipStream_t st);

}  // namespace mc
