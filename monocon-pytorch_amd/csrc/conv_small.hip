// 3x3 convolution for the thin full-resolution layers (16 or 32 input channels, 16 or 32 output
// channels: DLA level0 / level1 and their data gradients) on v_mfma_f32_16x16x4_f32, without LDS.
//
//   out[b,y,x,n] = act( scale[n] * sum_{c,r,s} in[b, y*S-1+r, x*S-1+s, c] * W[n,c,r,s] + bias[n] + residual )
//
// Same contract as conv_mfma_kernel (conv_mfma.h) -- it is selected through ConvArgs::cfg = CFG_SMALL
// -- but a different mapping, because at 16..32 channels the 32x32x2 tiling wastes half of every MFMA
// on padded columns and pays LDS staging for a K of only 144:
//   * M = 16 consecutive output pixels of one row, N = 16 output channels, K = 4 input channels per
//     MFMA.  A[pixel][k]: lane l holds pixel l%16, channels 4*(l/16)..+3 from ONE 16-byte load (a
//     pixel is a 64/128-byte NHWC row, so the 64 lanes of a tap read one contiguous 1-2 KB span that
//     stays in L1 across the 9 taps); register jj of that load is the A operand of MFMA jj.
//   * B[k][n] = W[n][4*(l/16)+jj][tap]: with the packed layout [tap][Cin/4][CoutP][4] these are again
//     16-byte loads; the whole filter (9 * Cin/4 * 4 registers per 16 columns) lives in registers for
//     the lifetime of the wave, which walks whole output rows.
//   * one buffer descriptor per input row: the top / bottom halo is a zero-length buffer, the left /
//     right halo an out-of-range lane offset in the two peeled edge groups of a row -- the main loop
//     has no vector address arithmetic (see the note on MFMA / VALU issue contention in conv_mfma.h).
//   * epilogue as in conv_mfma_kernel; the optional statistics partials are per (image, output row):
//     stats[b][y][CoutP][2], i.e. ConvArgs::chunks == Hout.
#include "conv_mfma.h"

namespace mc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef CS_PD
#define CS_PD 2          // groups in flight ahead of the one being computed (A/B in one session: 1 -> 2 = -0.23 ms per step, 3 = slower)
#endif
// LZ: the source is a lazy tensor (ConvSrc::la in conv_mfma.h: raw conv output + BatchNorm coefficients).  A lane always holds
// channels 4 * kq .. + 3 of its pixels, so the coefficients stay in registers and max(fma(y, la, lb), 0) is formed in front of
// the MFMAs (fp32 operands: no operand scale); padding stays 0 (`cap`: rows per descriptor, columns in the two edge groups).
template <int S, int CIN, int NTN, bool LZ = false>
__global__ __launch_bounds__(256) void conv_small_kernel(const ConvArgs a) {
    static_assert(!LZ || CIN == 16, "lazy source: the 16-channel layers");
    constexpr int KG = CIN / 16;          // 16-byte channel groups per lane and tap
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- the filter, once per wave: wreg[tap][kg][nt][jj] = W[n = nt*16 + li][c = kg*16 + 4*kq + jj][tap]
    f32x4 wreg[9][KG][NTN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
                wreg[t][kg][nt] = *reinterpret_cast<const f32x4 *>(
                    a.wpk + ((size_t)(t * (CIN / 4) + kg * 4 + kq) * a.CoutP + nt * 16 + li) * 4);

    float sc[NTN], bi[NTN], sh[NTN];
    bool nok[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int n = nt * 16 + li;
        nok[nt] = n < a.Cout;
        sc[nt] = (a.scale && nok[nt]) ? a.scale[n] : 1.f;
        bi[nt] = (a.bias && nok[nt]) ? a.bias[n] : 0.f;
        sh[nt] = (a.stat_shift && nok[nt]) ? a.stat_shift[n] : 0.f;
    }
    const bool do_stats = a.stats != nullptr, has_res = a.res != nullptr;
    const float floor_v = a.relu ? 0.f : -__builtin_inff();
    float vmax = 0.f;                 // max |out| of this lane (ConvArgs::amax_out)

    [[maybe_unused]] f32x4 lzA = {0.f, 0.f, 0.f, 0.f}, lzB = lzA;
    if constexpr (LZ) {
        lzA = reinterpret_cast<const f32x4 *>(a.src[0].la)[kq];
        lzB = reinterpret_cast<const f32x4 *>(a.src[0].lb)[kq];
    }
    const int vx = (li * S * CIN + kq * 4) * 4;                     // input lane offset inside a group
    constexpr int NR = (S == 1 && CIN == 16) ? 2 : 1;               // output rows per pass: a pair shares 2 of its 4 input rows
    constexpr int NI = (NR - 1) * S + 3;                            // input rows per pass
    const int hp = (a.Hout + NR - 1) / NR;
    const int RP = a.B * hp;
    for (int pr = xcd_order(blockIdx.x, gridDim.x) * 4 + wave; pr < RP; pr += gridDim.x * 4) {
        const int img = pr / hp, oy = (pr - img * hp) * NR;
        __amdgpu_buffer_rsrc_t r_x[NI], r_out[NR], r_res[NR];
        [[maybe_unused]] float rowcap[NI], ecap[3] = {0.f, 0.f, 0.f};      // LZ: 0 for padding rows / (edge groups) padding columns
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            const int iy = oy * S + r - 1;
            const bool ok = iy >= 0 && iy < a.Hin;
            r_x[r] = make_rsrc(a.src[0].p + ((size_t)img * a.Hin + (ok ? iy : 0)) * a.Win * CIN,
                               ok ? (unsigned)(a.Win * CIN) * 4u : 0u);
            rowcap[r] = ok ? __builtin_inff() : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const bool ok = oy + q < a.Hout;                       // the odd last row of an image: stores dropped
            const size_t row = (size_t)img * a.Hout + (ok ? oy + q : oy);
            r_out[q] = make_rsrc(a.out + row * a.Wout * a.out_ld, ok ? (unsigned)(a.Wout * a.out_ld) * 4u : 0u);
            r_res[q] = make_rsrc(has_res ? a.res + row * a.Wout * a.res_ld : a.out,
                                 (has_res && ok) ? (unsigned)(a.Wout * a.res_ld) * 4u : 0u);
        }
        float ssum[NR][NTN], ssq[NR][NTN];
#pragma unroll
        for (int q = 0; q < NR; ++q)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) ssum[q][nt] = ssq[q][nt] = 0.f;

        auto fetch = [&](int x0, bool edge, f32x4 (&av)[NI * 3][KG]) {
#pragma unroll
            for (int r = 0; r < NI; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) {
                        if (!edge) {
                            av[r * 3 + s][kg] = buf_load4(r_x[r], vx + (s * CIN + kg * 16) * 4, (x0 * S - 1) * CIN * 4);
                        } else {
                            const int px = (x0 + li) * S + s - 1;
                            av[r * 3 + s][kg] = buf_load4(
                                r_x[r], (px >= 0 && px < a.Win) ? (px * CIN + kg * 16 + kq * 4) * 4 : BUF_OOB, 0);
                            if constexpr (LZ) ecap[s] = (px >= 0 && px < a.Win) ? __builtin_inff() : 0.f;
                        }
                    }
        };
        auto compute = [&](int x0, const f32x4 (&av_in)[NI * 3][KG], bool edge = false) {
            f32x4 av[NI * 3][KG];
#pragma unroll
            for (int t = 0; t < NI * 3; ++t)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    if constexpr (LZ) {
                        const float cap = edge ? fminf(rowcap[t / 3], ecap[t % 3]) : rowcap[t / 3];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) av[t][kg][jj] = lazy_act(av_in[t][kg][jj], lzA[jj], lzB[jj], cap);
                    } else {
                        av[t][kg] = av_in[t][kg];
                    }
                }
            f32x4v acc[NR][NTN];
#pragma unroll
            for (int q = 0; q < NR; ++q)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) acc[q][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                            for (int q = 0; q < NR; ++q)
                                acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t + 3 * q * S][kg][jj], wreg[t][kg][nt][jj],
                                                                                  acc[q][nt], 0, 0, 0);
            // D layout: column (n) = lane & 15, row (pixel) = 4*(lane>>4) + q
#pragma unroll
            for (int q = 0; q < NR; ++q)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    const int n = nt * 16 + li;
                    const int v_out = nok[nt] ? ((4 * kq) * a.out_ld + a.out_coff + n) * 4 : BUF_OOB;
                    const int v_res = nok[nt] ? ((4 * kq) * a.res_ld + n) * 4 : BUF_OOB;
                    float rv[4];
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) rv[e] = buf_load1(r_res[q], v_res, (x0 + e) * a.res_ld * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[q][nt][e] * sc[nt] + bi[nt];
                        if (has_res) v += rv[e];
                        if (do_stats) {
                            const float d = v - sh[nt];
                            ssum[q][nt] += d;
                            ssq[q][nt] += d * d;
                        }
                        v = fmaxf(v, floor_v);
                        vmax = fmaxf(vmax, nok[nt] ? fabsf(v) : 0.f);
                        buf_store1(v, r_out[q], v_out, (x0 + e) * a.out_ld * 4);
                    }
                }
        };
        // the 16-pixel groups of a row as a software pipeline: the loads of group g + PD are in flight while group g runs
        // its MFMAs and stores (no branch between a fetch and the steady-state MFMAs: the compiler's wait counts are
        // path-insensitive)
        constexpr int PD = KG == 1 ? CS_PD : 1;           // (32 input channels: two groups ahead would not fit 256 registers)
        f32x4 ring[PD][NI * 3][KG], ev[NI * 3][KG];
        const int n = (a.Wout - 32) / 16;             // interior groups: output pixels 16 .. Wout - 17
        fetch(0, true, ev);
        int g = 0;
        if (n >= 2 * PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) fetch(16 + 16 * d, false, ring[d]);
            compute(0, ev, true);
            for (; g + 2 * PD <= n; g += PD) {
#pragma unroll
                for (int d = 0; d < PD; ++d) {
                    compute(16 + 16 * (g + d), ring[d]);
                    fetch(16 + 16 * (g + PD + d), false, ring[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < PD; ++d) compute(16 + 16 * (g + d), ring[d]);
            g += PD;
        } else {
            compute(0, ev, true);
        }
        for (; g < n; ++g) {
            fetch(16 + 16 * g, false, ev);
            compute(16 + 16 * g, ev);
        }
        if (a.Wout > 16) {
            fetch(a.Wout - 16, true, ev);
            compute(a.Wout - 16, ev, true);
        }

        if (do_stats) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                if (oy + q >= a.Hout) continue;
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    float s1 = ssum[q][nt], s2 = ssq[q][nt];
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                    if (kq == 0 && nok[nt]) {
                        float *dst = a.stats + ((((size_t)img * a.Hout + oy + q)) * a.CoutP + nt * 16 + li) * 2;
                        dst[0] = s1;
                        dst[1] = s2;
                    }
                }
            }
        }
    }
    if (a.amax_out) amax_update_wave(a.amax_out, vmax);
}

bool conv_small_ok(const ConvArgs &a, int ks, int stride) {
    if (ks != 3 || a.nsrc != 1 || (stride != 1 && stride != 2)) return false;
    if (a.Cin != 16 && a.Cin != 32) return false;
    if (a.Cout != 16 && a.Cout != 32) return false;
    if (a.Cin == 32 && a.Cout == 32) return false;            // filter would not fit the register budget
    if (a.Wout % 16 || a.Wout < 32) return false;
    if (a.Hout != (a.Hin + 2 - 3) / stride + 1 || a.Wout != (a.Win + 2 - 3) / stride + 1) return false;
    if (a.CoutP < a.Cout) return false;
    return true;
}

bool conv_small_lazy_ok(const ConvArgs &a, int ks, int stride) {
    return conv_small_ok(a, ks, stride) && stride == 2 && a.Cin == 16 && a.Cout == 32 && a.src[0].la && a.src[0].lb;
}

hipError_t launch_conv_small(const ConvArgs &a, int stride, hipStream_t st) {
    if (conv_thin_ok(a, 3, stride)) return launch_conv_thin(a, st);      // mode 3: the fp16-pipe kernel (conv_thin.hip)
    if (a.src[0].la) {          // lazy source: the stride-2 16 -> 32 layer (DLA level1) is the one that needs it
        if (!conv_small_lazy_ok(a, 3, stride)) return hipErrorInvalidValue;
        int lblocks = (a.B * a.Hout + 3) / 4;
        if (lblocks > 2048) lblocks = 2048;
        hipLaunchKernelGGL((conv_small_kernel<2, 16, 2, true>), dim3(lblocks), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    const int nr = (stride == 1 && a.Cin == 16) ? 2 : 1;             // rows per wave pass (see the kernel)
    const int rows = a.B * ((a.Hout + nr - 1) / nr);
    int blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
#define CS_LAUNCH(S_, CIN_, NTN_) hipLaunchKernelGGL((conv_small_kernel<S_, CIN_, NTN_>), dim3(blocks), dim3(256), 0, st, a)
    if (a.Cin == 16 && a.Cout == 16) { if (stride == 1) CS_LAUNCH(1, 16, 1); else CS_LAUNCH(2, 16, 1); }
    else if (a.Cin == 16) { if (stride == 1) CS_LAUNCH(1, 16, 2); else CS_LAUNCH(2, 16, 2); }
    else { if (stride == 1) CS_LAUNCH(1, 32, 1); else CS_LAUNCH(2, 32, 1); }
#undef CS_LAUNCH
    return hipGetLastError();
}

}  // namespace mc
