// Heat-map decode: k x k local-maximum filter (k odd, 3 in every reference configuration), exact per-image top-K over the flattened C*H*W map,
// gathers and 2D/3D box assembly.  Replaces reference utils/tensor_ops.py:17-31 and
// model/dense_heads/monocon_heads.py:313-329,379-558 (decode_heatmap, decode_alpha,
// calculate_roty, convert_pts2D_to_pts3D, _get_bboxes origin shift).
//
// Integer-exact contract: keep mask, top-K flat indices / classes and the threshold mask are
// bit-identical to the reference on identical float inputs; tie order is canonical
// (score descending, flat index ascending).  The filter kernel compacts the surviving positive
// values into a per-image candidate list; the selection is a 4-pass MSB-first radix select on the
// order-preserving integer image of the fp32 scores (no floating-point comparisons that could
// re-order equal keys), a second radix select over the flat indices of the ties at the threshold
// when they do not all fit, and a rank sort of the K winners.  Launch-latency bound (0.37 MB per image).
#include "kernels.h"

namespace mc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// local-maximum filter (any odd window) + compaction: every surviving strictly positive value becomes a (key, flat index)
// candidate of its image (list order is arbitrary -- the selection below orders by key, then index).  After the
// filter at most ~1/9 of a heat map survives, so the selection reads ~10x less than the map.
constexpr int LM_THREADS = 1024, LM_PER_THREAD = 4, LM_CHUNK = LM_THREADS * LM_PER_THREAD;
__global__ __launch_bounds__(LM_THREADS) void localmax_compact_kernel(const float *__restrict__ heat, int C, int H, int W,
                                                                      int R,      // window radius: (kernel - 1) / 2
                                                                      float *__restrict__ filt, uint8_t *__restrict__ keep,
                                                                      unsigned *__restrict__ cand_key,
                                                                      int *__restrict__ cand_idx,
                                                                      unsigned *__restrict__ cand_count) {
    __shared__ unsigned s_n;
    const int N = C * H * W, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n = 0;
    __syncthreads();
    float v[LM_PER_THREAD];
    bool cand[LM_PER_THREAD];
    int pos[LM_PER_THREAD];                  // slot inside this workgroup's run of candidates, -1: not a candidate
    // all loads of the thread's elements first (independent, in flight together), the slot bookkeeping afterwards
#pragma unroll
    for (int it = 0; it < LM_PER_THREAD; ++it) {
        const int e = blockIdx.x * LM_CHUNK + it * LM_THREADS + tid;
        cand[it] = false;
        v[it] = 0.f;
        if (e < N) {
            const int x = e % W, y = (e / W) % H;
            const float *plane = heat + (size_t)b * N + (e - y * W - x);
            const float c = plane[y * W + x];
            float m = c;
            // (max_pool2d pads with -inf: positions outside the map do not take part)
            for (int dy = -R; dy <= R; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -R; dx <= R; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    m = fmaxf(m, plane[yy * W + xx]);
                }
            }
            const bool k = (m == c);
            filt[(size_t)b * N + e] = k ? c : 0.0f * c;      // heat * keep.float()
            if (keep) keep[(size_t)b * N + e] = k ? 1 : 0;
            cand[it] = k && c > 0.f;
            v[it] = c;
        }
    }
#pragma unroll
    for (int it = 0; it < LM_PER_THREAD; ++it) {
        // one LDS atomic per wave reserves the wave's slots inside the workgroup's run
        const unsigned long long bal = __ballot(cand[it]);
        unsigned base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, (unsigned)__popcll(bal));
        base = __shfl(base, 0);
        pos[it] = cand[it] ? (int)(base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))) : -1;
    }
    __syncthreads();
    // the workgroup's candidates go to the head of ITS chunk of the list; the count per (image, chunk) tells the
    // selection kernel how many of the LM_CHUNK slots are live -- no global atomics, nothing to zero between calls
    if (tid == 0) cand_count[(size_t)b * gridDim.x + blockIdx.x] = s_n;
    const size_t dst = (size_t)b * N + (size_t)blockIdx.x * LM_CHUNK;
#pragma unroll
    for (int it = 0; it < LM_PER_THREAD; ++it) {
        if (pos[it] < 0) continue;
        cand_key[dst + pos[it]] = f2key(v[it]);
        cand_idx[dst + pos[it]] = blockIdx.x * LM_CHUNK + it * LM_THREADS + tid;
    }
}

// The 3x3 filter for W % 4 == 0: a thread owns four consecutive pixels of a row -- three 16-byte loads (rows y-1, y,
// y+1) plus the two edge columns from the neighbouring lanes (v_mov_dpp-style shuffles; only a wave's first / last
// lane reloads them) instead of 36 dword loads, which is what bounds the scalar kernel (texture-address rate).
__global__ __launch_bounds__(LM_THREADS) void localmax_compact_v4_kernel(const float *__restrict__ heat, int C, int H, int W,
                                                                         float *__restrict__ filt, uint8_t *__restrict__ keep,
                                                                         unsigned *__restrict__ cand_key,
                                                                         int *__restrict__ cand_idx,
                                                                         unsigned *__restrict__ cand_count) {
    __shared__ unsigned s_n;
    const int N = C * H * W, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int e0 = blockIdx.x * LM_CHUNK + tid * 4;
    const bool live = e0 < N;
    const float NINF = -__builtin_inff();
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    bool cand[4] = {false, false, false, false};
    if (live) {
        const int x0 = e0 % W, y = (e0 / W) % H;
        const float *plane = heat + (size_t)b * N + (e0 - y * W - x0);
        float m[4] = {NINF, NINF, NINF, NINF};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            const bool rok = yy >= 0 && yy < H;
            const float *row = plane + (rok ? yy : y) * W;
            f32x4 r = *reinterpret_cast<const f32x4 *>(row + x0);
            // edge columns: the neighbouring lane holds them unless it sits in another row / wave
            float lft = __shfl_up(r[3], 1), rgt = __shfl_down(r[0], 1);
            if (lane == 0 && x0 > 0) lft = row[x0 - 1];
            if (lane == 63 && x0 + 4 < W) rgt = row[x0 + 4];
            if (x0 == 0) lft = NINF;
            if (x0 + 4 >= W) rgt = NINF;
            if (!rok) { r = f32x4{NINF, NINF, NINF, NINF}; lft = rgt = NINF; }
            m[0] = fmaxf(m[0], fmaxf(lft, fmaxf(r[0], r[1])));
            m[1] = fmaxf(m[1], fmaxf(r[0], fmaxf(r[1], r[2])));
            m[2] = fmaxf(m[2], fmaxf(r[1], fmaxf(r[2], r[3])));
            m[3] = fmaxf(m[3], fmaxf(r[2], fmaxf(r[3], rgt)));
            if (dy == 0) { c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = r[3]; }
        }
        f32x4 fo;
        unsigned kb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool k = (m[j] == c[j]);
            fo[j] = k ? c[j] : 0.0f * c[j];      // heat * keep.float()
            kb |= (k ? 1u : 0u) << (8 * j);
            cand[j] = k && c[j] > 0.f;
        }
        *reinterpret_cast<f32x4 *>(filt + (size_t)b * N + e0) = fo;
        if (keep) *reinterpret_cast<unsigned *>(keep + (size_t)b * N + e0) = kb;
    }
    // slots: one LDS atomic per wave for the wave's candidates, lanes ordered by (j, lane)
    unsigned long long bal[4];
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bal[j] = __ballot(cand[j]); tot += (unsigned)__popcll(bal[j]); }
    unsigned base = 0;
    if (lane == 0 && tot) base = atomicAdd(&s_n, tot);
    base = __shfl(base, 0);
    int pos[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pos[j] = cand[j] ? (int)(base + (unsigned)__popcll(bal[j] & ((1ull << lane) - 1ull))) : -1;
        base += (unsigned)__popcll(bal[j]);
    }
    __syncthreads();
    if (tid == 0) cand_count[(size_t)b * gridDim.x + blockIdx.x] = s_n;
    const size_t dst = (size_t)b * N + (size_t)blockIdx.x * LM_CHUNK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (pos[j] < 0) continue;
        cand_key[dst + pos[j]] = f2key(c[j]);
        cand_idx[dst + pos[j]] = e0 + j;
    }
}

constexpr int DEC_THREADS = 1024;
constexpr int DEC_MAXK = 1024;

// bucket holding the krem-th element when the 256 histogram bins are walked downwards (desc) or upwards:
// parallel scan by the first 256 threads; out[0] = bucket, out[1] = elements in the buckets walked before it
__device__ __forceinline__ void pick_bucket(const unsigned *hist, unsigned *wsum, unsigned krem, bool desc, unsigned *out) {
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned v = 0, s = 0;
    int bk = 0;
    if (tid < 256) {
        bk = desc ? 255 - tid : tid;
        v = hist[bk];
        s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(s, o);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[tid >> 6] = s;
    }
    __syncthreads();
    if (tid < 256) {
        for (int w = 0; w < (tid >> 6); ++w) s += wsum[w];
        if (s >= krem && s - v < krem) { out[0] = (unsigned)bk; out[1] = s - v; }
    }
    __syncthreads();
}

// Exact top-K into ckey / cidx (unordered): key descending, ties by ascending flat index.
// MSB-first radix select for the key of the K-th largest element; if more elements tie at that key than fit,
// a second radix select over their flat indices finds the index threshold.  COMPACT: the elements are the chunked
// candidate list (gk, gi) -- chunk c holds off[c+1] - off[c] live entries at c * LM_CHUNK -- whose keys are first
// staged densely in LDS (lkey) when they fit, so the four histogram passes never leave the CU; the flat indices are
// only fetched for the few elements at or above the threshold.  Otherwise the elements are the filtered map f itself.
constexpr int DEC_LK = 10240;            // candidate keys staged in LDS (40 KB); longer lists are read from global
constexpr int DEC_MAXCHUNK = 256;
template <bool COMPACT>
__device__ void select_topk(const float *f, const unsigned *gk, const int *gi, const unsigned *off, int nchunks, int n,
                            int K, unsigned *lkey, unsigned *hist, unsigned *wsum, unsigned *sh, unsigned *ckey,
                            int *cidx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = DEC_THREADS / 64;
    const bool staged = COMPACT && n <= DEC_LK;
    if (staged) {
        for (int c = wave; c < nchunks; c += NW) {
            const int nc = (int)(off[c + 1] - off[c]);
            const unsigned *src = gk + c * LM_CHUNK;
            unsigned *dst = lkey + off[c];
            int j = lane;
            for (; j + 192 < nc; j += 256) {       // four independent loads in flight
                const unsigned k0 = src[j], k1 = src[j + 64], k2 = src[j + 128], k3 = src[j + 192];
                dst[j] = k0; dst[j + 64] = k1; dst[j + 128] = k2; dst[j + 192] = k3;
            }
            for (; j < nc; j += 64) dst[j] = src[j];
        }
        __syncthreads();
    }
    // fn(key) over every element
    auto visit_keys = [&](auto fn) {
        if (!COMPACT) {
            for (int i = tid; i < n; i += DEC_THREADS) fn(f2key(f[i]));
        } else if (staged) {
            for (int i = tid; i < n; i += DEC_THREADS) fn(lkey[i]);
        } else {
            for (int c = wave; c < nchunks; c += NW)
                for (int j = lane, nc = (int)(off[c + 1] - off[c]); j < nc; j += 64) fn(gk[c * LM_CHUNK + j]);
        }
    };
    // fn(key, slot): slot -> flat index through idx_of (a global load for the candidate list: only call it when needed)
    auto idx_of = [&](int slot) -> unsigned { return COMPACT ? (unsigned)gi[slot] : (unsigned)slot; };
    auto visit_all = [&](auto fn) {
        if (!COMPACT) {
            for (int i = tid; i < n; i += DEC_THREADS) fn(f2key(f[i]), i);
        } else {
            for (int c = wave; c < nchunks; c += NW)
                for (int j = lane, nc = (int)(off[c + 1] - off[c]); j < nc; j += 64)
                    fn(staged ? lkey[off[c] + j] : gk[c * LM_CHUNK + j], c * LM_CHUNK + j);
        }
    };
    if (tid == 0) { sh[0] = 0; sh[1] = K; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = sh[0], krem = sh[1];
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        visit_keys([&](unsigned k) {
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        });
        __syncthreads();
        pick_bucket(hist, wsum, krem, true, sh + 4);
        if (tid == 0) {
            sh[0] = prefix | (sh[4] << shift);
            sh[1] = krem - sh[5];
            sh[3] = hist[sh[4]];          // after the last pass: how many elements carry exactly this key
        }
        __syncthreads();
    }
    const unsigned T = sh[0], need_eq = sh[1], eq_total = sh[3];
    unsigned I = 0xFFFFFFFFu;             // ties with flat index <= I are taken
    if (eq_total > need_eq) {
        __syncthreads();
        if (tid == 0) { sh[0] = 0; sh[1] = need_eq; }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = sh[0], krem = sh[1];
            const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            visit_all([&](unsigned k, int slot) {
                if (k != T) return;
                const unsigned ix = idx_of(slot);
                if ((ix & himask) == prefix) atomicAdd(&hist[(ix >> shift) & 255u], 1u);
            });
            __syncthreads();
            pick_bucket(hist, wsum, krem, false, sh + 4);
            if (tid == 0) {
                sh[0] = prefix | (sh[4] << shift);
                sh[1] = krem - sh[5];
            }
            __syncthreads();
        }
        I = sh[0];
    }
    __syncthreads();
    if (tid == 0) sh[2] = 0;
    __syncthreads();
    visit_all([&](unsigned k, int slot) {
        if (k < T) return;
        const unsigned ix = idx_of(slot);
        if (k > T || ix <= I) {
            const unsigned pos = atomicAdd(&sh[2], 1u);
            ckey[pos] = k;
            cidx[pos] = (int)ix;
        }
    });
    __syncthreads();
}

__global__ __launch_bounds__(DEC_THREADS) void topk_decode_kernel(const DecodeArgs a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[8], wsum[4];
    __shared__ unsigned off[DEC_MAXCHUNK + 1];
    __shared__ unsigned lkey[DEC_LK];
    __shared__ unsigned ckey[DEC_MAXK];
    __shared__ int cidx[DEC_MAXK];
    __shared__ unsigned skey[DEC_MAXK];
    __shared__ int sidx[DEC_MAXK];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = a.H * a.W, N = a.C * HW, K = a.K;
    const float *f = a.filt + (size_t)b * N;

    // K or more positive local maxima (always, for real heat maps): select among the compacted candidates;
    // otherwise zeros / negatives reach the top-K and the selection runs over the whole filtered map
    const int nchunks = (N + LM_CHUNK - 1) / LM_CHUNK;
    int M = 0;
    if (nchunks <= DEC_MAXCHUNK) {
        if (tid < nchunks) off[tid + 1] = a.cand_count[(size_t)b * nchunks + tid];
        __syncthreads();
        if (tid == 0) {
            unsigned s = 0;
            off[0] = 0;
            for (int c = 1; c <= nchunks; ++c) { s += off[c]; off[c] = s; }
        }
        __syncthreads();
        M = (int)off[nchunks];
    }
    if (M >= K)
        select_topk<true>(f, a.cand_key + (size_t)b * N, a.cand_idx + (size_t)b * N, off, nchunks, M, K, lkey, hist, wsum, sh,
                          ckey, cidx);
    else
        select_topk<false>(f, nullptr, nullptr, off, nchunks, N, K, lkey, hist, wsum, sh, ckey, cidx);   // chunks = slices of f
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

int decode_chunks(int n_per_image) { return (n_per_image + LM_CHUNK - 1) / LM_CHUNK; }

hipError_t launch_decode(const DecodeArgs &a, hipStream_t st) {
    if (a.K > DEC_MAXK || a.K < 1 || a.K > a.C * a.H * a.W) return hipErrorInvalidValue;
    if (!a.cand_key || !a.cand_idx || !a.cand_count) return hipErrorInvalidValue;
    const int N = a.C * a.H * a.W;
    const dim3 grid((unsigned)((N + LM_CHUNK - 1) / LM_CHUNK), (unsigned)a.B);
    if (a.lm_kernel < 1 || a.lm_kernel % 2 == 0) return hipErrorInvalidValue;
    if (a.W % 4 == 0 && a.lm_kernel == 3)
        hipLaunchKernelGGL(localmax_compact_v4_kernel, grid, dim3(LM_THREADS), 0, st, a.pred[0], a.C, a.H, a.W, a.filt,
                           a.keep_localmax, a.cand_key, a.cand_idx, a.cand_count);
    else
        hipLaunchKernelGGL(localmax_compact_kernel, grid, dim3(LM_THREADS), 0, st, a.pred[0], a.C, a.H, a.W, a.lm_kernel / 2, a.filt,
                           a.keep_localmax, a.cand_key, a.cand_idx, a.cand_count);
    hipLaunchKernelGGL(topk_decode_kernel, dim3(a.B), dim3(DEC_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace mc
