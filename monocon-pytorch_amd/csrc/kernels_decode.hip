// Heat-map decode: 3x3 local-maximum filter, exact per-image top-K over the flattened C*H*W map,
// gathers and 2D/3D box assembly.  Replaces reference utils/tensor_ops.py:17-31 and
// model/dense_heads/monocon_heads.py:313-329,379-558 (decode_heatmap, decode_alpha,
// calculate_roty, convert_pts2D_to_pts3D, _get_bboxes origin shift).
//
// Integer-exact contract: keep mask, top-K flat indices / classes and the threshold mask are
// bit-identical to the reference on identical float inputs; tie order is canonical
// (score descending, flat index ascending).  The selection is a 4-pass MSB-first radix select
// on the order-preserving integer image of the fp32 scores (no floating-point comparisons that
// could re-order equal keys), followed by an index-ordered pick of the ties at the threshold
// and a rank sort of the K winners.  The work is launch-latency bound (0.37 MB per image).
#include "kernels.h"

namespace mc {

__global__ void localmax_kernel(const float *__restrict__ heat, int B, int C, int H, int W,
                                float *__restrict__ filt, uint8_t *__restrict__ keep) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = e % W, y = (e / W) % H;
        const float *plane = heat + (e - (size_t)y * W - x);
        const float v = plane[y * W + x];
        float m = v;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                m = fmaxf(m, plane[yy * W + xx]);
            }
        }
        const bool k = (m == v);
        filt[e] = k ? v : 0.0f * v;      // heat * keep.float()
        if (keep) keep[e] = k ? 1 : 0;
    }
}

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int DEC_THREADS = 1024;
constexpr int DEC_MAXK = 1024;

__global__ __launch_bounds__(DEC_THREADS) void topk_decode_kernel(const DecodeArgs a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_krem, s_count, s_eq_taken;
    __shared__ unsigned wave_cnt[DEC_THREADS / 64];
    __shared__ unsigned ckey[DEC_MAXK];
    __shared__ int cidx[DEC_MAXK];
    __shared__ unsigned skey[DEC_MAXK];
    __shared__ int sidx[DEC_MAXK];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = a.H * a.W, N = a.C * HW, K = a.K;
    const float *f = a.filt + (size_t)b * N;

    // ---- radix select: key of the K-th largest element
    if (tid == 0) { s_prefix = 0; s_krem = K; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < N; i += DEC_THREADS) {
            const unsigned k = f2key(f[i]);
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned krem = s_krem, cum = 0;
            int bkt = 255;
            for (; bkt > 0; --bkt) {
                if (cum + hist[bkt] >= krem) break;
                cum += hist[bkt];
            }
            s_prefix = prefix | ((unsigned)bkt << shift);
            s_krem = krem - cum;
        }
        __syncthreads();
    }
    const unsigned T = s_prefix;          // key of the K-th largest
    const unsigned need_eq = s_krem;      // how many elements equal to T belong to the top-K
    if (tid == 0) { s_count = 0; s_eq_taken = 0; }
    __syncthreads();
    // ---- strictly greater: any order
    for (int i = tid; i < N; i += DEC_THREADS) {
        const unsigned k = f2key(f[i]);
        if (k > T) {
            const unsigned pos = atomicAdd(&s_count, 1u);
            ckey[pos] = k;
            cidx[pos] = i;
        }
    }
    __syncthreads();
    const unsigned ngt = s_count;         // == K - need_eq
    // ---- ties at T: smallest flat indices first (index-ordered block scan)
    for (int base = 0; base < N; base += DEC_THREADS) {
        if (s_eq_taken >= need_eq) break;
        const int i = base + tid;
        const bool eq = (i < N) && (f2key(f[i]) == T);
        const unsigned long long bal = __ballot(eq);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) wave_cnt[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = s_eq_taken;
        for (int w = 0; w < wv; ++w) before += wave_cnt[w];
        const unsigned my = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (eq && my < need_eq) {
            ckey[ngt + my] = T;
            cidx[ngt + my] = i;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < DEC_THREADS / 64; ++w) tot += wave_cnt[w];
            s_eq_taken += tot;
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- rank sort: key desc, index asc
    for (int t = tid; t < K; t += DEC_THREADS) {
        const unsigned kt = ckey[t];
        const int it = cidx[t];
        int rank = 0;
        for (int j = 0; j < K; ++j) {
            const unsigned kj = ckey[j];
            rank += (kj > kt) || (kj == kt && cidx[j] < it);
        }
        skey[rank] = kt;
        sidx[rank] = it;
    }
    __syncthreads();

    // ---- gathers + box assembly, one thread per detection
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    for (int t = tid; t < K; t += DEC_THREADS) {
        const int flat = sidx[t];
        const float score = f[flat];
        const int cls = flat / HW, ind = flat % HW;
        const int yi = ind / a.W, xi = ind % a.W;
        const float ys = (float)yi, xs = (float)xi;
        auto G = [&](int p, int nch, int ch) -> float {
            return a.pred[p][((size_t)b * nch + ch) * HW + ind];
        };
        const float w0 = G(2, 2, 0), w1 = G(2, 2, 1);
        const float o0 = G(3, 2, 0), o1 = G(3, 2, 1);
        const float tx = xs + o0, ty = ys + o1;
        const float sx = a.pad_w / (float)a.W, sy = a.pad_h / (float)a.H;
        const float dlog = G(7, 2, 1), z = G(7, 2, 0);
        const float sigma = expf(-dlog);
        const float sc2 = score * sigma;
        float *b2 = a.box2d + ((size_t)b * K + t) * 5;
        b2[0] = (tx - w0 / 2.f) * sx;
        b2[1] = (ty - w1 / 2.f) * sy;
        b2[2] = (tx + w0 / 2.f) * sx;
        b2[3] = (ty + w1 / 2.f) * sy;
        b2[4] = sc2;
        // alpha: arg-max bin (first maximum) + its offset, wrapped once into [-pi, pi]
        int bin = 0;
        float best = G(8, 12, 0);
        for (int k = 1; k < 12; ++k) {
            const float v = G(8, 12, k);
            if (v > best) { best = v; bin = k; }
        }
        float alpha = (float)bin * (float)(2.0 * 3.14159265358979323846 / 12.0) + G(9, 12, bin);
        if (alpha > PI_F) alpha -= TWO_PI_F;
        if (alpha < -PI_F) alpha += TWO_PI_F;
        const float u = (G(5, 18, 16) + xs) * sx;
        const float v = (G(5, 18, 17) + ys) * sy;
        const float *P = a.P2 + (size_t)b * 12;
        float roty = alpha + atan2f(u - P[2], P[0]);
        while (roty > PI_F) roty -= TWO_PI_F;
        while (roty < -PI_F) roty += TWO_PI_F;
        const float *Pi = a.P2inv + (size_t)b * 16;
        const float hx = u * z, hy = v * z;
        float xyz[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) xyz[i] = ((hx * Pi[i * 4 + 0] + hy * Pi[i * 4 + 1]) + z * Pi[i * 4 + 2]) + Pi[i * 4 + 3];
        const float d0 = G(6, 3, 0), d1 = G(6, 3, 1), d2 = G(6, 3, 2);
        float *b3 = a.box3d + ((size_t)b * K + t) * 7;
        b3[0] = xyz[0];
        b3[1] = xyz[1] + d1 * 0.5f;
        b3[2] = xyz[2];
        b3[3] = d0; b3[4] = d1; b3[5] = d2;
        b3[6] = roty;
        a.scores[(size_t)b * K + t] = score;
        a.flat_index[(size_t)b * K + t] = flat;
        a.cls[(size_t)b * K + t] = cls;
        if (a.keep_thr) a.keep_thr[(size_t)b * K + t] = sc2 > a.thr ? 1 : 0;
    }
}

hipError_t launch_decode(const DecodeArgs &a, hipStream_t st) {
    if (a.K > DEC_MAXK || a.K < 1 || a.K > a.C * a.H * a.W) return hipErrorInvalidValue;
    const size_t total = (size_t)a.B * a.C * a.H * a.W;
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(localmax_kernel, dim3((unsigned)g), dim3(256), 0, st, a.pred[0], a.B, a.C, a.H, a.W, a.filt,
                       a.keep_localmax);
    hipLaunchKernelGGL(topk_decode_kernel, dim3(a.B), dim3(DEC_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace mc
