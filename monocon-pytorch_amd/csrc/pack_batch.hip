// Batched weight packing: every conv weight of the model -> its K-major MFMA panel (fp32 and, in the bf16-pipe modes,
// the bf16 piece planes), for the forward panels and for the data-gradient panels, in ONE launch each.
//
// The panels are re-derived from the master weights before every train step (the optimizer just changed them):
// per layer that was one launch for the fp32 panel, one for the bf16 planes and one BatchNorm fold -- ~200 launches of
// ~5 us per step in fp32 mode, ~400 in the bf16x3 mode, i.e. 1-2 ms of a 80-100 ms step spent on launch cadence.
// A job table (one entry per weight tensor, block ranges by prefix sum) turns each family into a single grid.
// Layouts are those of pack_conv_w_kernel / pack_conv_w_bf16_kernel / pack_conv_w_dgrad_kernel /
// pack_conv_w_dgrad_bf16_kernel (kernels_misc.hip, conv_bf16.hip, kernels_head_train.hip), element for element.
#include "kernels.h"
#include "train.h"
#include "conv_mfma.h"
#include <type_traits>

namespace mc {

// nsplit == 2: two fp16 pieces of r (already multiplied by the tensor's power-of-two scale), else nsplit bf16 pieces
__device__ __forceinline__ void store_pieces_b(float r, unsigned short *dst, size_t plane, int nsplit) {
    if (nsplit == 2) {
        const _Float16 hi = (_Float16)r;
        const _Float16 lo = (_Float16)(r - (float)hi);
        dst[0] = __builtin_bit_cast(unsigned short, hi);
        dst[plane] = __builtin_bit_cast(unsigned short, lo);
    } else {
        for (int q = 0; q < nsplit; ++q) {
            const __bf16 piece = (__bf16)r;
            dst[q * plane] = __builtin_bit_cast(unsigned short, piece);
            r -= (float)piece;
        }
    }
}

// max |w| of every forward job's master weight folded into its slot (fp16-split mode; slots zeroed by the caller).
// Jobs that share a slot (the nine head convs of the fused 64 -> 576 panel) end up with their common maximum.
__global__ __launch_bounds__(256) void weight_amax_batch_kernel(const PackJobDesc *__restrict__ tab, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJobDesc j = tab[lo];
    if (!j.amax || j.kind != 0) return;
    // at most WA_BLOCKS workgroups per job take part (the grid is the pack kernel's): one atomic each on the job's word.
    // (8 made the 2.4 M weights of a 512 x 512 layer a 1 150-element serial walk per thread: 0.39 ms per step for 78 MB;
    //  with 64 the pass is bandwidth-bound)
    constexpr int WA_BLOCKS = 256;      // (round 5: every block of the job, 16-byte loads, four in flight: 77 -> ~25 us)
    const int lb = blockIdx.x - j.block_begin, nb = j.nblocks < WA_BLOCKS ? j.nblocks : WA_BLOCKS;
    if (lb >= nb) return;
    const size_t total = (size_t)j.Cout * j.Cin * j.k * j.k;
    float vmax = 0.f;
    if ((reinterpret_cast<uintptr_t>(j.w) & 15) == 0) {
        const size_t n4 = total / 4, stride = (size_t)nb * 256;
        const float4 *w4 = reinterpret_cast<const float4 *>(j.w);
        size_t e = (size_t)lb * 256 + threadIdx.x;
        for (; e + 3 * stride < n4; e += 4 * stride) {
            const float4 a = w4[e], b = w4[e + stride], c = w4[e + 2 * stride], d = w4[e + 3 * stride];
            vmax = fmaxf(vmax, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                     fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            vmax = fmaxf(vmax, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                                     fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
        }
        for (; e < n4; e += stride) {
            const float4 a = w4[e];
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        for (size_t t = n4 * 4 + (size_t)lb * 256 + threadIdx.x; t < total; t += stride) vmax = fmaxf(vmax, fabsf(j.w[t]));
    } else {
        for (size_t e = (size_t)lb * 256 + threadIdx.x; e < total; e += (size_t)nb * 256) vmax = fmaxf(vmax, fabsf(j.w[e]));
    }
    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
    const unsigned bits = __builtin_bit_cast(unsigned, vmax);
    if (bits < 0x7f800000u && bits != 0u) atomicMax(&s_max, bits);
    __syncthreads();
    if (threadIdx.x == 0 && s_max != 0u) atomicMax(j.amax, s_max);
}

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJobDesc *__restrict__ tab, int njobs) {
    // job of this block: last entry whose block_begin <= blockIdx.x (wave-uniform binary search)
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJobDesc j = tab[lo];
    const int lb = blockIdx.x - j.block_begin;
    const int k = j.k, kk = k * k;
    unsigned short *d16 = static_cast<unsigned short *>(j.dst16);
    const float wscale = (j.nsplit == 2 && d16) ? exp2i(f16_scale_exp(*j.amax)) : 1.f;
    // Output order (round 5): a work item is one 8-channel group of the panel's K dimension x one column, items numbered
    // column-fastest, so that a wave's stores are consecutive 16-byte runs (a piece's eight fp16 values, half a fp32 quad
    // pair).  Element order -- consecutive threads on consecutive taps of the master weight -- sent every 2- or 4-byte store
    // to a different tap plane: 1.35 GB written per step for 0.3 GB of panels (rocprofv3 WRITE_SIZE), 0.45 ms.
    auto put8 = [&](const float (&v)[8], size_t i32a, size_t i32b, size_t i16, size_t plane) {
        if (j.dst32) {
            *reinterpret_cast<float4 *>(j.dst32 + i32a) = float4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4 *>(j.dst32 + i32b) = float4{v[4], v[5], v[6], v[7]};
        }
        if (d16) {
            if (j.nsplit == 2) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 hi, lo;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float r = v[q] * wscale;
                    hi[q] = (_Float16)r;
                    lo[q] = (_Float16)(r - (float)hi[q]);
                }
                *reinterpret_cast<h8 *>(d16 + i16) = hi;
                *reinterpret_cast<h8 *>(d16 + plane + i16) = lo;
            } else {
                typedef unsigned short u8v __attribute__((ext_vector_type(8)));
                float r[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) r[q] = v[q] * wscale;
                for (int pz = 0; pz < j.nsplit; ++pz) {
                    u8v o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const __bf16 piece = (__bf16)r[q];
                        o[q] = __builtin_bit_cast(unsigned short, piece);
                        r[q] -= (float)piece;
                    }
                    *reinterpret_cast<u8v *>(d16 + pz * plane + i16) = o;
                }
            }
        }
    };
    if (j.kind == 0) {
        // forward panel: [tap][CinPanel/4][CoutP][4] fp32, [piece][tap][CinPanel/8][CoutP][8] bf16
        const size_t total = (size_t)j.Cout * j.Cin * kk;
        const size_t plane = (size_t)kk * j.CinTotal * j.CoutP;
        if (j.Cin % 8 == 0 && j.c_off % 8 == 0 && (kk == 9 || kk == 1) && (reinterpret_cast<uintptr_t>(j.w) & 15) == 0) {      // (16-byte loads of the master)
            const int items = j.Cout * (j.Cin >> 3);
            auto run = [&](auto kkc) {
                constexpr int KK = decltype(kkc)::value;
                for (int it = lb * 256 + threadIdx.x; it < items; it += j.nblocks * 256) {
                    const int n = it % j.Cout, c8 = it / j.Cout;
                    const int cc = c8 * 8 + j.c_off, nn = n + j.n_off;
                    const float *src = j.w + ((size_t)n * j.Cin + c8 * 8) * KK;      // 8 channels x KK taps, contiguous
                    float buf[8 * KK];
#pragma unroll
                    for (int q = 0; q < 2 * KK; ++q) {
                        const float4 t = reinterpret_cast<const float4 *>(src)[q];
                        buf[4 * q] = t.x; buf[4 * q + 1] = t.y; buf[4 * q + 2] = t.z; buf[4 * q + 3] = t.w;
                    }
#pragma unroll
                    for (int tap = 0; tap < KK; ++tap) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = buf[q * KK + tap];
                        const size_t i32a = (((size_t)tap * (j.CinTotal >> 2) + (cc >> 2)) * j.CoutP + nn) * 4;
                        put8(v, i32a, i32a + (size_t)j.CoutP * 4, (((size_t)tap * (j.CinTotal >> 3) + (cc >> 3)) * j.CoutP + nn) * 8, plane);
                    }
                }
            };
            if (kk == 9) run(std::integral_constant<int, 9>{}); else run(std::integral_constant<int, 1>{});
            return;
        }
        for (size_t e = (size_t)lb * 256 + threadIdx.x; e < total; e += (size_t)j.nblocks * 256) {
            const int tap = e % kk;
            const int c = (e / kk) % j.Cin;
            const int n = e / ((size_t)kk * j.Cin);
            const int cc = c + j.c_off, nn = n + j.n_off;
            float r = j.w[e];
            if (j.dst32) j.dst32[(((size_t)tap * (j.CinTotal >> 2) + (cc >> 2)) * j.CoutP + nn) * 4 + (cc & 3)] = r;
            if (d16)
                store_pieces_b(r * wscale, d16 + (((size_t)tap * (j.CinTotal >> 3) + (cc >> 3)) * j.CoutP + nn) * 8 + (cc & 7), plane,
                               j.nsplit);
        }
    } else {
        // data-gradient panel of one source (channels [c_off, c_off + Cs) of the forward weight): transposed + flipped,
        // or one output-parity class of a stride-2 3x3 data gradient (cls = 2*py + px)
        const int Cs = j.Cin, cls = j.cls, py = cls >> 1, px = cls & 1;
        const size_t total = (size_t)j.Cout * Cs * kk;
        const size_t plane = (size_t)(cls < 0 ? kk : (1 + py) * (1 + px)) * j.CoutP * j.CsP;
        if (j.Cout % 8 == 0 && (kk == 9 || kk == 1)) {
            // item = (source channel cl, group of 8 output channels n8), cl fastest: K of the data gradient is the forward's Cout
            const int items = Cs * (j.Cout >> 3);
            auto run = [&](auto kc) {
                constexpr int K = decltype(kc)::value, KK = K * K;
                for (int it = lb * 256 + threadIdx.x; it < items; it += j.nblocks * 256) {
                    const int cl = it % Cs, n8 = it / Cs;
                    const float *src = j.w + ((size_t)(n8 * 8) * j.CinTotal + j.c_off + cl) * KK;      // row q: + q * CinTotal * KK
#pragma unroll
                    for (int tap = 0; tap < KK; ++tap) {
                        const int r = tap / K, sx = tap % K;
                        int tapd;
                        if (cls < 0) {
                            tapd = (K - 1 - r) * K + (K - 1 - sx);
                        } else {
                            if ((py == 0) != (r == 1) || (px == 0) != (sx == 1)) continue;   // tap of the other parity
                            const int dr = py ? (2 - r) / 2 : 0, ds = px ? (2 - sx) / 2 : 0;
                            tapd = dr * (1 + px) + ds;
                        }
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = src[(size_t)q * j.CinTotal * KK + tap];
                        const size_t i32a = (((size_t)tapd * (j.CoutP >> 2) + n8 * 2) * j.CsP + cl) * 4;
                        put8(v, i32a, i32a + (size_t)j.CsP * 4, (((size_t)tapd * (j.CoutP >> 3) + n8) * j.CsP + cl) * 8, plane);
                    }
                }
            };
            if (k == 3) run(std::integral_constant<int, 3>{}); else run(std::integral_constant<int, 1>{});
            return;
        }
        for (size_t e = (size_t)lb * 256 + threadIdx.x; e < total; e += (size_t)j.nblocks * 256) {
            const int tap = e % kk;
            const int cl = (e / kk) % Cs;
            const int n = e / ((size_t)kk * Cs);
            const int r = tap / k, s = tap % k;
            int tapd;
            if (cls < 0) {
                tapd = (k - 1 - r) * k + (k - 1 - s);
            } else {
                if ((py == 0) != (r == 1) || (px == 0) != (s == 1)) continue;   // tap of the other parity
                const int dr = py ? (2 - r) / 2 : 0, ds = px ? (2 - s) / 2 : 0;
                tapd = dr * (1 + px) + ds;
            }
            float rem = j.w[(((size_t)n * j.CinTotal + j.c_off + cl) * k + r) * k + s];
            if (j.dst32) j.dst32[(((size_t)tapd * (j.CoutP >> 2) + (n >> 2)) * j.CsP + cl) * 4 + (n & 3)] = rem;
            if (d16)
                store_pieces_b(rem * wscale, d16 + (((size_t)tapd * (j.CoutP >> 3) + (n >> 3)) * j.CsP + cl) * 8 + (n & 7), plane,
                               j.nsplit);
        }
    }
}

void PackBatch::clear() {
    jobs.clear();
    total_blocks = 0;
    uploaded = false;
}
void PackBatch::add(PackJobDesc j) {
    const size_t total = (size_t)j.Cout * j.Cin * j.k * j.k;
    size_t nb = (total + 1023) / 1024;          // ~4 elements per thread
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    j.nblocks = (int)nb;
    j.block_begin = total_blocks;
    total_blocks += (int)nb;
    jobs.push_back(j);
    uploaded = false;
}
hipError_t PackBatch::launch(hipStream_t st, bool with_amax) {
    if (jobs.empty()) return hipSuccess;
    for (const PackJobDesc &j : jobs)
        if (j.nsplit == 2 && j.dst16 && !j.amax) return hipErrorInvalidValue;
    if (!uploaded) {
        if (dev) (void)hipFree(dev);
        dev = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&dev), jobs.size() * sizeof(PackJobDesc));
        if (e != hipSuccess) return e;
        e = hipMemcpy(dev, jobs.data(), jobs.size() * sizeof(PackJobDesc), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        uploaded = true;
    }
    if (with_amax)      // the maxima first: the pack kernel scales by them
        hipLaunchKernelGGL(weight_amax_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, dev, (int)jobs.size());
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, dev, (int)jobs.size());
    return hipGetLastError();
}
PackBatch::~PackBatch() {
    if (dev) (void)hipFree(dev);
}

// ---- BatchNorm folding (eval mode) of many layers in one launch
__global__ void fold_bn_batch_kernel(const FoldJobDesc *__restrict__ tab) {
    const FoldJobDesc j = tab[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= j.C) return;
    const float inv = 1.0f / sqrtf(j.rv[c] + j.eps);
    const float s = (j.g ? j.g[c] : 1.f) * inv;
    j.scale[c] = s;
    j.shift[c] = (j.b ? j.b[c] : 0.f) - j.rm[c] * s;
}
void FoldBatch::clear() {
    jobs.clear();
    uploaded = false;
    maxC = 0;
}
void FoldBatch::add(const FoldJobDesc &j) {
    jobs.push_back(j);
    if (j.C > maxC) maxC = j.C;
    uploaded = false;
}
hipError_t FoldBatch::launch(hipStream_t st) {
    if (jobs.empty()) return hipSuccess;
    if (!uploaded) {
        if (dev) (void)hipFree(dev);
        dev = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&dev), jobs.size() * sizeof(FoldJobDesc));
        if (e != hipSuccess) return e;
        e = hipMemcpy(dev, jobs.data(), jobs.size() * sizeof(FoldJobDesc), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        uploaded = true;
    }
    hipLaunchKernelGGL(fold_bn_batch_kernel, dim3((unsigned)((maxC + 63) / 64), (unsigned)jobs.size()), dim3(64), 0, st, dev);
    return hipGetLastError();
}
FoldBatch::~FoldBatch() {
    if (dev) (void)hipFree(dev);
}

}  // namespace mc
